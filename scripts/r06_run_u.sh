#!/bin/bash
# round 6, GPU call U: the loss-partial hand-off without its release fence (TD3_LIGHT_HANDOFF, td3_kernels.hip: handoff_last) in all three critic passes, and
# k_critic_block as eight waves: TD3 test files per build, the batch-100 update, the td3 and td3_reference legs -- alternating, one box
set -u
OUT=gpurun_out/r06_u
mkdir -p $OUT
VD=$(pwd)/plen_ml_walk_amd/csrc/variants
VARIANTS="old nw4 nw8"
for V in $VARIANTS; do
  echo "== TD3 tests, $V"; PLENTD3_LIB=$VD/td3_$V.so timeout 1200 python -m pytest tests/test_block_gpu.py tests/test_robustness_gpu.py tests/test_td3_golden.py -q -x -m gpu 2>&1 | tail -2
done
for i in 1 2; do for V in $VARIANTS; do
  echo "== batch 100 / 256, $V ($i)"; PLENTD3_LIB=$VD/td3_$V.so PLEN_SMALL_BATCH_ONLY=team timeout 300 python scripts/gpu_td3_small_batch.py 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-150
done; done
for i in 1 2 3; do
  for V in $VARIANTS; do
    PLENTD3_LIB=$VD/td3_$V.so timeout 600 python bench.py --gpus 1 --dtype f32 --steps 20 --warmup 5 --no-cpu-baseline --no-parity --legs td3,td3_reference > $OUT/leg_${V}_$i.json 2> $OUT/leg_${V}_$i.err
    python3 -c "
import json
l=json.loads(open('$OUT/leg_${V}_$i.json').read().strip().splitlines()[-1]); c=l['config']
print('$V run $i: td3 %.3f M env-steps/s, %.0f grad steps/s, alone %.3f, in-loop %.3f | ratio100 %.0f | td3_reference %.0f updates/s' % (c['td3_value']/1e6, c['td3_grad_steps_per_s'], c['td3_roofline_alone_frac'], c['td3_roofline_frac'], c['td3_ratio100_grad_steps_per_s'], c['td3_reference_updates_per_s']))"
  done
done
