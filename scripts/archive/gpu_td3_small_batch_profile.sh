#!/bin/bash
# kernel trace of the small-batch (team) update at batch 100: gpurun_out/prof_td3_team/ -> summarised by scripts/summarise_kernel_db.py
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export PLEN_SMALL_BATCH_ONLY=${1:-team}
timeout 300 rocprofv3 --kernel-trace -d gpurun_out/prof_td3_team -o t -- python3 scripts/gpu_td3_small_batch.py 100 > gpurun_out/prof_td3_team.log 2>&1
tail -2 gpurun_out/prof_td3_team.log
