#!/bin/bash
# whole-step A/B of an environment switch: scratch/env_ab.sh VAR  (interleaved 0/1 runs, U-Net then uganConsis)
V=$1
for w in unet ugan; do
  for g in 0 1 0 1; do
    env $V=$g timeout -k 10 200 python bench.py --workload $w --warmup 10 --steps 40 --no-step-profile --no-unet-step --no-cpu-baseline --no-roofline 2>/dev/null \
      | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w $V=$g', d['ms_per_step'])" || exit 1
  done
done
