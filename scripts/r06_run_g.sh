#!/bin/bash
# round 6, GPU call G: the td3 leg with three graph replays per vector step against six vector steps per replay, two and four collectors, twice each
set -u
OUT=gpurun_out/r06_g
mkdir -p $OUT
for rnd in 1 2; do for cfg in "0 2" "1 2" "1 4" "0 4"; do set -- $cfg
  timeout 600 python bench.py --legs td3 --td3-block-graph $1 --td3-parts $2 --no-cpu-baseline --no-parity --steps 50 > $OUT/td3_bg$1_p$2_$rnd.json 2> $OUT/td3_bg$1_p$2_$rnd.err
  python - <<PY
import json
try:
    d=json.loads([l for l in open("$OUT/td3_bg$1_p$2_$rnd.json") if l.startswith("{")][-1])
    t=d["legs"]["td3"]; print("block_graph=$1 parts=$2", "td3 %.3f M env-steps/s, %.0f grad steps/s, %.4f ms/step" % (t["value"]/1e6, t["grad_steps_per_s"], t["ms_per_step"]), "roofline", {k: round(v,3) for k,v in t.get("roofline",{}).items() if k in ("frac","kernel_us","update_us","alone_frac","alone_kernel_us")}, flush=True)
except Exception as ex:
    print("block_graph=$1 parts=$2 FAILED", repr(ex), open("$OUT/td3_bg$1_p$2_$rnd.err").read()[-600:])
PY
done; done
