#!/bin/bash
# resident-workgroup target of the fp16-operand persistent conv (SMSUT_P_WGS_PER_CU) vs the time of bench.py's f16 roofline kernel
for o in 0 1 2 3 4 5 6 8; do
  SMSUT_P_WGS_PER_CU=$o python3 bench.py --dtype f16 --size 512 --steps 3 --warmup 2 --no-cpu-baseline --no-unet-step --no-config5 --no-step-profile 2>/dev/null > /tmp/occ_$o.json
  python3 -c "
import json,sys
d=json.loads(open('/tmp/occ_$o.json').read().strip().splitlines()[-1]); r=d['roofline']; print('wgs/cu', $o, r['avg_launch_ms'], r['achieved'])"
done
