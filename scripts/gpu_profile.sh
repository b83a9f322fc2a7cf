#!/bin/bash
# Profiles of one build on the GPU box (run through gpurun): rocprofv3 kernel stats of the bench command and PMC passes
# (FETCH_SIZE | WRITE_SIZE | SQ_* in separate runs, never combined with tracing) for the f32 and f64 env kernels.
# usage: bash scripts/gpu_profile.sh <tag>      -> gpurun_out/prof_<tag>/..., condensed by scripts/make_pmc_summary.py into profiles/
set -u
TAG=${1:-r02_x}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for DT in f64 f32; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$DT -- python3 $REPO/bench.py --dtype $DT --legs "" --groups 1 --steps 100 --warmup 10 --no-cpu-baseline --no-parity > $OUT/bench_$DT.json 2> $OUT/bench_$DT.err
  for C in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_BRANCH"; do
    NAME=$(echo $C | cut -d' ' -f1)
    rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_${DT}_$NAME -- python3 $REPO/scripts/gpu_pmc_target.py 50 4096 $DT > $OUT/pmc_${DT}_$NAME.log 2>&1
  done
done
cd $REPO
python3 scripts/make_pmc_summary.py $OUT $TAG
