#!/bin/bash
# round 6, GPU call Y: the final tree -- trainer soak (pipelined TD3 from the shipped policy, 40 s), then the driver's bench command four times in a row
set -u
OUT=gpurun_out/r06_y
mkdir -p $OUT
timeout 600 python scripts/archive/gpu_trainer_soak.py 40 2>&1 | grep -v amdgpu.ids | tail -6 | tee $OUT/trainer_soak.txt
for i in 1 2 3 4; do
  timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/line_$i.json 2> $OUT/err_$i.txt
  echo "run $i rc=$? $(python3 -c "
import json
try:
    l=json.loads(open('$OUT/line_$i.json').read().strip().splitlines()[-1]); c=l['config']
    print('f64 %.3f M  f32 %.3f M  td3 %.3f M / %.0f  in-loop %.3f alone %.3f  ratio100 %.0f  td3_ref %.0f  policy %.3f M  dr %.3f M' % (l['value']/1e6, c['f32_value']/1e6, c['td3_value']/1e6, c['td3_grad_steps_per_s'], c['td3_roofline_frac'], c['td3_roofline_alone_frac'], c['td3_ratio100_grad_steps_per_s'], c['td3_reference_updates_per_s'], c['policy_value']/1e6, c['dr_value']/1e6))
except Exception as e: print('NO LINE', e)
")"
done | tee $OUT/summary.txt
