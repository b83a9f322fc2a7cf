#!/bin/bash
# round 6, GPU call I: does the six-step graph's serialisation of its branches yield to the runtime's graph-queue knob?
set -u
OUT=gpurun_out/r06_i
mkdir -p $OUT
for q in default 2 4 8 16; do
  if [ $q = default ]; then unset DEBUG_HIP_FORCE_GRAPH_QUEUES; else export DEBUG_HIP_FORCE_GRAPH_QUEUES=$q; fi
  timeout 600 python bench.py --legs td3 --td3-block-graph 1 --td3-parts 2 --no-cpu-baseline --no-parity --steps 50 > $OUT/td3_q$q.json 2> $OUT/td3_q$q.err
  python - <<PY
import json
try:
    d=json.loads([l for l in open("$OUT/td3_q$q.json") if l.startswith("{")][-1])
    t=d["legs"]["td3"]; print("DEBUG_HIP_FORCE_GRAPH_QUEUES=$q", "td3 %.3f M env-steps/s, %.0f grad steps/s, %.4f ms/step" % (t["value"]/1e6, t["grad_steps_per_s"], t["ms_per_step"]), flush=True)
except Exception as ex:
    print("q=$q FAILED", repr(ex), open("$OUT/td3_q$q.err").read()[-400:])
PY
done
unset DEBUG_HIP_FORCE_GRAPH_QUEUES
bash scripts/r06_run_h.sh
