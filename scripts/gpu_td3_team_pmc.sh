#!/bin/bash
# PMC passes over the small-batch TD3 update (scripts/gpu_td3_small_batch.py 100, team path only): separate runs per counter group, no tracing, each under
# its own timeout (a group the hardware cannot collect makes rocprofv3 abort and then wait).  -> gpurun_out/pmc_td3_team/<group>/..., summarised by
# scripts/summarise_pmc_by_kernel.py into gpurun_out/r04_td3_team_pmc.json
set -u
REPO=$(pwd)
OUT=$REPO/gpurun_out/pmc_td3_team
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp PLEN_SMALL_BATCH_ONLY=team
cd /tmp
timeout 60 rocprofv3 --list-avail 2>/dev/null | grep -oE "\b(SQ_[A-Z0-9_]+|TCP_[A-Z0-9_a-z]+|TCC_[A-Z0-9_a-z]+|FETCH_SIZE|WRITE_SIZE)\b" | sort -u > $OUT/avail.txt
i=0
for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
         "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM" \
         "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum" \
         "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 150 rocprofv3 --pmc $C --output-format csv -d $OUT/g$i -- python3 $REPO/scripts/gpu_td3_small_batch.py 100 > $OUT/g$i.log 2>&1
  echo "group $i ($C): exit $?"
done
cd $REPO
python3 scripts/summarise_pmc_by_kernel.py $OUT gpurun_out/r04_td3_team_pmc.json
