#!/bin/bash
# round 6, GPU call N: what the driver runs at round end, on the final tree -- smoke(), the whole -m gpu suite, the bench command
set -u
OUT=gpurun_out/r06_n
mkdir -p $OUT
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; echo "smoke rc $?" >> $OUT/smoke.txt; tail -4 $OUT/smoke.txt
timeout 2400 python -m pytest tests -m gpu -q > $OUT/gputest.txt 2>&1; echo "pytest rc $?" >> $OUT/gputest.txt; tail -4 $OUT/gputest.txt
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_line.json 2> $OUT/bench_err.txt; cut -c1-300 $OUT/bench_line.json
