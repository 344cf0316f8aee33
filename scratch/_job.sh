set -e
python -m pytest tests/test_modules_gpu.py tests/test_winograd_evidence_gpu.py tests/test_trainer_gpu.py tests/test_graph_gpu.py -m gpu -q -x > gpurun_out/r05_pool_test.txt 2>&1 || { tail -40 gpurun_out/r05_pool_test.txt; exit 1; }
tail -3 gpurun_out/r05_pool_test.txt
B="--steps 30 --warmup 8 --no-cpu-baseline --no-roofline --no-unet-step --no-config5 --no-dist-leg"
P='import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], j["ms_per_step"], j["ms_per_step_min"], j["ms_per_step_median"])'
for i in 1 2; do
  SMSUT_IN_ACT_POOL=0 timeout -k 10 300 python bench.py $B 2>/dev/null | python -c "$P" "headline pool=0"
  SMSUT_IN_ACT_POOL=1 timeout -k 10 300 python bench.py $B 2>/dev/null | python -c "$P" "headline pool=1"
done > gpurun_out/r05_pool_head.txt 2>&1
cat gpurun_out/r05_pool_head.txt
for i in 1 2; do
  SMSUT_IN_ACT_POOL=0 timeout -k 10 300 python bench.py --dtype f16 --size 512 $B 2>/dev/null | python -c "$P" "c5 pool=0"
  SMSUT_IN_ACT_POOL=1 timeout -k 10 300 python bench.py --dtype f16 --size 512 $B 2>/dev/null | python -c "$P" "c5 pool=1"
done > gpurun_out/r05_pool_c5.txt 2>&1
cat gpurun_out/r05_pool_c5.txt
