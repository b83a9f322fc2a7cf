#!/bin/bash
# round 6, GPU call K: the round's record -- the driver's own bench command, then the headline-schedule kernel trace + PMC passes (profiles/r06_k_*)
set -u
mkdir -p gpurun_out/r06_k
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_k/bench_line.json 2> gpurun_out/r06_k/bench_err.txt
cut -c1-400 gpurun_out/r06_k/bench_line.json
bash scripts/gpu_profile_headline.sh r06_k > gpurun_out/r06_k/profile_log.txt 2>&1
tail -5 gpurun_out/r06_k/profile_log.txt
ls profiles | grep r06_k
cp profiles/r06_k_* gpurun_out/r06_k/ 2>/dev/null
