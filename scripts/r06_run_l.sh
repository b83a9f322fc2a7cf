#!/bin/bash
# round 6, GPU call L: the small-batch kernels on packed weights -- tests, then the batch-100 update's GPU time with and without them, then the reference-recipe leg
set -u
OUT=gpurun_out/r06_l
mkdir -p $OUT
timeout 1200 python -m pytest tests/test_robustness_gpu.py tests/test_block_gpu.py tests/test_td3_golden.py -m gpu -x -q > $OUT/gputest_td3.txt 2>&1; echo "pytest rc $?" >> $OUT/gputest_td3.txt
tail -6 $OUT/gputest_td3.txt
for pk in 0 1 0 1; do
  PLEN_TD3_TEAM_PACKED=$pk PLEN_SMALL_BATCH_ONLY=team timeout 300 python scripts/gpu_td3_small_batch.py 100 256 2>&1 | grep -v amdgpu.ids | sed "s/^/packed=$pk /" | grep "team"
done
for pk in 0 1; do
  PLEN_TD3_TEAM_PACKED=$pk timeout 600 python bench.py --legs td3_reference --no-cpu-baseline --no-parity --steps 20 > $OUT/ref_pk$pk.json 2> $OUT/ref_pk$pk.err
  python - <<PY
import json
d=json.loads([l for l in open("$OUT/ref_pk$pk.json") if l.startswith("{")][-1]); t=d["legs"]["td3_reference"]
print("packed=$pk td3_reference %.0f updates/s, agent.train %.1f us per call" % (t["grad_steps_per_s"], t["agent_train_call"]["us_per_call"]))
PY
done
