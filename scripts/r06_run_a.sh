#!/bin/bash
# round 6, GPU call A: the whole -m gpu suite on the new build, bitwise comparison against the round-5 build (reference configuration + limit-violating targets),
# A/B of the builds on the headline legs, one default bench line.
set -u
OUT=gpurun_out/r06_a
mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/gputest.txt 2>&1; echo "pytest rc $?" >> $OUT/gputest.txt
timeout 600 python scripts/gpu_same_bits.py r05.so > $OUT/same_bits.txt 2>&1
timeout 900 python scripts/gpu_ab64.py r05.so r06a.so - > $OUT/ab_f64.txt 2>&1
AB_DTYPE=f32 timeout 900 python scripts/gpu_ab64.py r05.so r06a.so - > $OUT/ab_f32.txt 2>&1
timeout 900 python bench.py > $OUT/bench_line.json 2> $OUT/bench_err.txt
tail -3 $OUT/gputest.txt; cat $OUT/same_bits.txt $OUT/ab_f64.txt $OUT/ab_f32.txt; cut -c1-600 $OUT/bench_line.json
