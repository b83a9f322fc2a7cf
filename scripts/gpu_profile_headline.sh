#!/bin/bash
# Profile of the mode bench.py HEADLINES: f64, 4 x 1024-env sub-batches on 4 HIP streams (f32: 2 x 2048), through gpurun.
#   1. rocprofv3 --kernel-trace --stats of the exact driver command with the extra legs off (program directly after `--`);
#   2. PMC passes (FETCH_SIZE | WRITE_SIZE | SQ_*; separate runs, no tracing) of the same sub-batch schedule.
# (each rocprofv3 run sits under its own `timeout`: a counter group the hardware cannot collect makes rocprofv3 abort and then wait forever -- r04_b lost 36 GPU-minutes to that)
# usage: bash scripts/gpu_profile_headline.sh <tag>  -> gpurun_out/prof_<tag>/..., condensed by scripts/make_headline_summary.py into profiles/<tag>_groups*
set -u
TAG=${1:-r04_a}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for DT in f64 f32; do
  G=4; [ $DT = f32 ] && G=2
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$DT -- python3 $REPO/bench.py --gpus 1 --dtype $DT --steps 100 --warmup 10 --no-cpu-baseline --no-parity --legs "" > $OUT/bench_$DT.json 2> $OUT/bench_$DT.err
  for C in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_BRANCH"; do
    NAME=$(echo $C | cut -d' ' -f1)
    timeout 300 rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_${DT}_$NAME -- python3 $REPO/scripts/gpu_pmc_target_groups.py 4096 $G $DT > $OUT/pmc_${DT}_$NAME.log 2>&1
  done
done
cd $REPO
python3 scripts/make_headline_summary.py $OUT $TAG
