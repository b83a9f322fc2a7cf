#!/bin/bash
# round 6, GPU call M: td3 tests on the packed small-batch weights, then the per-kernel times of a batch-100 update with and without them
set -u
OUT=$PWD/gpurun_out/r06_m
mkdir -p $OUT
timeout 1500 python -m pytest tests/test_robustness_gpu.py tests/test_block_gpu.py tests/test_td3_golden.py tests/test_cabi_gpu.py -m gpu -q > $OUT/gputest_td3.txt 2>&1; echo "pytest rc $?" >> $OUT/gputest_td3.txt
tail -6 $OUT/gputest_td3.txt
export TMPDIR=/tmp PLEN_SMALL_BATCH_ONLY=team
REPO=$PWD
cd /tmp
for pk in 0 1; do
  export PLEN_TD3_TEAM_PACKED=$pk
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_pk$pk -- python3 $REPO/scripts/gpu_td3_small_batch.py 100 > $OUT/log_pk$pk.txt 2>&1
  f=$(find $OUT/stats_pk$pk -name "*kernel_stats.csv" | head -1)
  echo "== packed=$pk"; head -6 "$f" | cut -c1-160
  cp "$f" $OUT/kernel_stats_pk$pk.csv
done
