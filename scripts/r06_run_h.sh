#!/bin/bash
# round 6, GPU call H: the batch-100 update's floor -- the shipped small-batch kernels against builds WITHOUT their trips to memory for weights (results wrong, time right)
set -u
OUT=gpurun_out/r06_h
mkdir -p $OUT
for lib in "" td3_noload.so td3_noload_nopark.so; do
  if [ -n "$lib" ]; then export PLENTD3_LIB=$PWD/plen_ml_walk_amd/csrc/variants/$lib; else unset PLENTD3_LIB; fi
  PLEN_SMALL_BATCH_ONLY=team timeout 300 python scripts/gpu_td3_small_batch.py 100 > $OUT/log_${lib:-shipped}.txt 2>&1
  cp gpurun_out/r04_td3_small_batch.json $OUT/small_batch_${lib:-shipped}.json 2>/dev/null
  echo "== ${lib:-shipped}"; tail -3 $OUT/log_${lib:-shipped}.txt; cat $OUT/small_batch_${lib:-shipped}.json
done
