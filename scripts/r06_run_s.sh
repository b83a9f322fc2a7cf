#!/bin/bash
# round 6, GPU call S: the driver's bench command eight times in a row on one box (the capture-time abort of profiles/README.md's r06_o note must not come back)
set -u
mkdir -p gpurun_out/r06_s
for i in 1 2 3 4 5 6 7 8; do
  timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_s/line_$i.json 2> gpurun_out/r06_s/err_$i.txt
  echo "run $i rc=$? $(python3 -c "
import json,sys
try:
    l=json.loads(open('gpurun_out/r06_s/line_$i.json').read().strip().splitlines()[-1]); c=l['config']
    print('f64 %.3f M  f32 %.3f M  td3 %.3f M  td3_ref %.0f  policy %.3f M' % (l['value']/1e6, c['f32_value']/1e6, c['td3_value']/1e6, c['td3_reference_updates_per_s'], c['policy_value']/1e6))
except Exception as e: print('NO LINE', e)
")"
done | tee gpurun_out/r06_s/summary.txt
