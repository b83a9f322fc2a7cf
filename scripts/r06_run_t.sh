#!/bin/bash
# round 6, GPU call T: k_critic_block variants against the shipped kernel -- eight waves of 32 features (-DBLK_CRITIC_NW=8: two waves per SIMD in the register footprint of
# one), loss partials counted before the last flush (-DBLK_EARLY_SUMS=1), partials handed over by write-through stores instead of a release fence (-DBLK_LIGHT_HANDOFF=1):
# parity tests, phase stamps, the pass alone, the td3 leg -- alternating, one box
set -u
OUT=gpurun_out/r06_t
mkdir -p $OUT
VD=$(pwd)/plen_ml_walk_amd/csrc/variants
VARIANTS="default nw4_e0_l1 nw8_e0_l0 nw8_e0_l1 nw8_e1_l1"
for V in $VARIANTS; do
  L=""; [ $V != default ] && L=$VD/td3_$V.so
  echo "== parity, $V"; PLENTD3_LIB=$L timeout 900 python -m pytest tests/test_block_gpu.py -q -x 2>&1 | tail -1
done
for V in "4 0 0" "4 0 1" "8 0 0" "8 0 1" "8 1 1"; do set -- $V; echo "== stamps, $1 waves, early sums $2, light hand-off $3"; BLK_CRITIC_NW=$1 BLK_EARLY_SUMS=$2 BLK_LIGHT_HANDOFF=$3 timeout 300 python scripts/gpu_td3_block_stamps.py 2>&1 | grep -v amdgpu.ids | tail -3; done
for V in $VARIANTS; do
  L=""; [ $V != default ] && L=$VD/td3_$V.so
  echo "== pass alone, $V"; PLENTD3_LIB=$L timeout 600 python scripts/gpu_td3_block_bench.py 2>&1 | tail -4 | cut -c1-150; cp gpurun_out/r05_td3_block_bench.json $OUT/block_bench_$V.json
done
for i in 1 2 3; do
  for V in $VARIANTS; do
    L=""; [ $V != default ] && L=$VD/td3_$V.so
    PLENTD3_LIB=$L timeout 600 python bench.py --gpus 1 --dtype f32 --steps 20 --warmup 5 --no-cpu-baseline --no-parity --legs td3 > $OUT/leg_${V}_$i.json 2> $OUT/leg_${V}_$i.err
    python3 -c "
import json
l=json.loads(open('$OUT/leg_${V}_$i.json').read().strip().splitlines()[-1]); c=l['config']
print('$V run $i: td3 %.3f M env-steps/s, %.0f grad steps/s, in-loop kernel %.1f us (frac %.3f), alone %.3f' % (c['td3_value']/1e6, c['td3_grad_steps_per_s'], c['td3_roofline_kernel_us'], c['td3_roofline_frac'], c['td3_roofline_alone_frac']))"
  done
done
