#!/bin/bash
# round 6, GPU call V: the shipped TD3 library (eight-wave k_critic_block, no release fence in any hand-off incl. k_store) against csrc/variants/td3_old.so
# (-DBLK_CRITIC_NW=4 -DTD3_LIGHT_HANDOFF=0: round 5's forms): TD3 test files, then the td3 / td3_reference / policy legs alternating, one box
set -u
OUT=gpurun_out/r06_v
mkdir -p $OUT
VD=$(pwd)/plen_ml_walk_amd/csrc/variants
echo "== TD3 tests, shipped"; timeout 1200 python -m pytest tests/test_block_gpu.py tests/test_robustness_gpu.py tests/test_td3_golden.py -q -x -m gpu 2>&1 | tail -2
for i in 1 2 3; do
  for V in old shipped; do
    L=""; [ $V = old ] && L=$VD/td3_old.so
    PLENTD3_LIB=$L timeout 600 python bench.py --gpus 1 --dtype f32 --steps 20 --warmup 5 --no-cpu-baseline --no-parity --legs td3,td3_reference,policy > $OUT/leg_${V}_$i.json 2> $OUT/leg_${V}_$i.err
    python3 -c "
import json
l=json.loads(open('$OUT/leg_${V}_$i.json').read().strip().splitlines()[-1]); c=l['config']
print('$V run $i: td3 %.3f M env-steps/s, %.0f grad steps/s, alone %.3f, in-loop %.3f | ratio100 %.0f | td3_reference %.0f updates/s | policy %.3f M' % (c['td3_value']/1e6, c['td3_grad_steps_per_s'], c['td3_roofline_alone_frac'], c['td3_roofline_frac'], c['td3_ratio100_grad_steps_per_s'], c['td3_reference_updates_per_s'], c['policy_value']/1e6))"
  done
done
