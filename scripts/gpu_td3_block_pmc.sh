#!/bin/bash
# PMC passes over the large-batch TD3 update (scripts/gpu_td3_block_run.py): separate runs per counter group, no tracing, each under its own timeout.
# -> gpurun_out/pmc_td3_block/<group>/..., summarised by scripts/summarise_pmc_by_kernel.py into gpurun_out/r05_td3_block_pmc.json; then a kernel trace with
# --stats of the same program -> gpurun_out/trace_td3_block/
set -u
REPO=$(pwd)
OUT=$REPO/gpurun_out/pmc_td3_block
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
i=0
for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
         "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_WAIT_INST_LDS" \
         "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum" \
         "GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 150 rocprofv3 --pmc $C --output-format csv -d $OUT/g$i -- python3 $REPO/scripts/gpu_td3_block_run.py 4096 20 > $OUT/g$i.log 2>&1
  echo "group $i ($C): exit $?"
done
cd $REPO
python3 scripts/summarise_pmc_by_kernel.py $OUT gpurun_out/r05_td3_block_pmc.json k_critic_block k_policy_block k_wgrad k_adam k_pack k_colsum
cd /tmp
rm -rf $REPO/gpurun_out/trace_td3_block
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/trace_td3_block -- python3 $REPO/scripts/gpu_td3_block_run.py 4096 200 > $REPO/gpurun_out/trace_td3_block.log 2>&1
cd $REPO
find gpurun_out/trace_td3_block -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'head -14 {}'
find gpurun_out/trace_td3_block -name "*kernel_trace.csv" -delete
find gpurun_out/pmc_td3_block -name "*counter_collection.csv" -size +2M -delete
