set -e
python scratch/x32_debug.py 2>&1 | grep -v "max err 0 " | tail -5
python -m pytest tests/test_f16_gpu.py -m gpu -q -x > gpurun_out/r05_x32_test.txt 2>&1 || { tail -30 gpurun_out/r05_x32_test.txt; exit 1; }
tail -2 gpurun_out/r05_x32_test.txt
B="--dtype f16 --size 512 --steps 30 --warmup 8 --no-cpu-baseline --no-roofline --no-unet-step --no-config5 --no-dist-leg"
P='import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], j["ms_per_step"], j["ms_per_step_min"], j["ms_per_step_median"])'
for i in 1 2; do
  timeout -k 10 300 python scratch/run_with_lib.py scratch/lib_x16/libsmsut_hip.so bench.py $B 2>/dev/null | python -c "$P" "c5 x16"
  timeout -k 10 300 python bench.py $B 2>/dev/null | python -c "$P" "c5 x32+wgrad"
done > gpurun_out/r05_x32_c5b.txt 2>&1
cat gpurun_out/r05_x32_c5b.txt
for i in 1 2; do
  timeout -k 10 300 python scratch/run_with_lib.py scratch/lib_x16/libsmsut_hip.so bench.py --workload unet $B 2>/dev/null | python -c "$P" "unet512 x16"
  timeout -k 10 300 python bench.py --workload unet $B 2>/dev/null | python -c "$P" "unet512 x32+wgrad"
done > gpurun_out/r05_x32_unetb.txt 2>&1
cat gpurun_out/r05_x32_unetb.txt
