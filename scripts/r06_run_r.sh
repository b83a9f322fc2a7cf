#!/bin/bash
# round 6, GPU call R: kernel trace of the td3 leg alone (configs[2]: 4096 envs + the pipelined TD3 loop, batch 4096) -- what each kernel of the update costs
# INSIDE the loop, beside resident env launches (profiles/r06_r_td3_leg_kernel_stats.csv); the alone figures are profiles/r05_td3_block_kernel_stats.csv
set -u
REPO=$(pwd)
OUT=$REPO/gpurun_out/r06_r
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $REPO/bench.py --gpus 1 --dtype f32 --steps 20 --warmup 5 --no-cpu-baseline --no-parity --legs td3 --td3-steps 1000 > $OUT/bench_line.json 2> $OUT/bench_err.txt
cd $REPO
cut -c1-300 $OUT/bench_line.json
F=$(find $OUT/stats -name '*kernel_stats.csv' | head -1)
[ -n "$F" ] && cp $F $OUT/td3_leg_kernel_stats.csv && head -14 $OUT/td3_leg_kernel_stats.csv | cut -c1-160
find $OUT/stats -name '*kernel_trace.csv' -delete
