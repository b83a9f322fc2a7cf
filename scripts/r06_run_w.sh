#!/bin/bash
# round 6, GPU call W: the record of the final TD3 library -- phase stamps (shipped form and round 5's), PMC passes + kernel trace of the stand-alone large-batch update
# (profiles/r06_w_td3_block_*), then the driver's bench command
set -u
OUT=gpurun_out/r06_w
mkdir -p $OUT
echo "== stamps, shipped (eight waves, no release fence)"; timeout 300 python scripts/gpu_td3_block_stamps.py 2>&1 | grep -v amdgpu.ids | tail -18
echo "== stamps, round 5's form (four waves, fence)"; BLK_CRITIC_NW=4 TD3_LIGHT_HANDOFF=0 timeout 300 python scripts/gpu_td3_block_stamps.py 2>&1 | grep -v amdgpu.ids | tail -18
bash scripts/gpu_td3_block_pmc.sh > $OUT/pmc_log.txt 2>&1; tail -16 $OUT/pmc_log.txt | cut -c1-160
cp gpurun_out/r05_td3_block_pmc.json $OUT/td3_block_pmc.json
find gpurun_out/trace_td3_block -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/td3_block_kernel_stats.csv
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_line.json 2> $OUT/bench_err.txt; echo "bench rc=$?"
cut -c1-300 $OUT/bench_line.json
