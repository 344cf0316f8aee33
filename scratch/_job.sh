set -e
python -m pytest tests/test_modules_gpu.py -m gpu -q -x > gpurun_out/r05_tp_test.txt 2>&1 || { tail -40 gpurun_out/r05_tp_test.txt; exit 1; }
tail -3 gpurun_out/r05_tp_test.txt
python -m pytest tests/test_trainer_gpu.py tests/test_graph_gpu.py tests/test_siblings_gpu.py -m gpu -q -x > gpurun_out/r05_tp_test2.txt 2>&1 || { tail -40 gpurun_out/r05_tp_test2.txt; exit 1; }
tail -3 gpurun_out/r05_tp_test2.txt
B="--steps 30 --warmup 8 --no-cpu-baseline --no-roofline --no-unet-step --no-config5 --no-dist-leg"
P='import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], j["ms_per_step"], j["ms_per_step_min"], j["ms_per_step_median"])'
for i in 1 2; do
  SMSUT_TAIL_AVGPOOL=0 timeout -k 10 300 python bench.py $B 2>/dev/null | python -c "$P" "headline tp=0"
  SMSUT_TAIL_AVGPOOL=1 timeout -k 10 300 python bench.py $B 2>/dev/null | python -c "$P" "headline tp=1"
done > gpurun_out/r05_tp_head.txt 2>&1
cat gpurun_out/r05_tp_head.txt
for i in 1 2; do
  SMSUT_TAIL_AVGPOOL=0 timeout -k 10 300 python bench.py --dtype f16 --size 512 $B 2>/dev/null | python -c "$P" "c5 tp=0"
  SMSUT_TAIL_AVGPOOL=1 timeout -k 10 300 python bench.py --dtype f16 --size 512 $B 2>/dev/null | python -c "$P" "c5 tp=1"
done > gpurun_out/r05_tp_c5.txt 2>&1
cat gpurun_out/r05_tp_c5.txt
