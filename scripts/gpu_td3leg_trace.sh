#!/bin/bash
set -u
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_td3leg_r05
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $REPO/bench.py --gpus 1 --steps 20 --warmup 5 --dtype f32 --legs td3 --no-cpu-baseline --no-parity --td3-steps 400 > $OUT/bench.json 2> $OUT/bench.err
ls $OUT/trace/*/ | head
