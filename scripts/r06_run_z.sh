#!/bin/bash
# round 6, GPU call Z: k_actor_block (the collectors' / the policy leg's actor forward) with its exploration noise drawn at the kernel's start by all threads, as four
# and as eight waves, against the library at HEAD (csrc/variants/td3_head.so): TD3 test files per build, then the td3 and policy legs alternating, one box
set -u
OUT=gpurun_out/r06_z
mkdir -p $OUT
VD=$(pwd)/plen_ml_walk_amd/csrc/variants
VARIANTS="head actor_nw4 actor_nw8"
for V in actor_nw4 actor_nw8; do
  echo "== TD3 tests, $V"; PLENTD3_LIB=$VD/td3_$V.so timeout 1200 python -m pytest tests/test_block_gpu.py tests/test_robustness_gpu.py tests/test_td3_golden.py -q -x -m gpu 2>&1 | tail -2
done
for i in 1 2 3; do
  for V in $VARIANTS; do
    PLENTD3_LIB=$VD/td3_$V.so timeout 600 python bench.py --gpus 1 --dtype f32 --steps 20 --warmup 5 --no-cpu-baseline --no-parity --legs td3,policy > $OUT/leg_${V}_$i.json 2> $OUT/leg_${V}_$i.err
    python3 -c "
import json
l=json.loads(open('$OUT/leg_${V}_$i.json').read().strip().splitlines()[-1]); c=l['config']
print('$V run $i: td3 %.3f M env-steps/s, %.0f grad steps/s | policy %.3f M' % (c['td3_value']/1e6, c['td3_grad_steps_per_s'], c['policy_value']/1e6))"
  done
done
