#!/bin/bash
# round 6, GPU call F: the whole -m gpu suite on the returned-to-dense kernel + the six-step TD3 graphs; td3 leg with and without the block graphs, twice each
set -u
OUT=gpurun_out/r06_f
mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q -s > $OUT/gputest.txt 2>&1; echo "pytest rc $?" >> $OUT/gputest.txt
tail -4 $OUT/gputest.txt; grep -E "first step, reference|envs whose first step|f32 loop copies" $OUT/gputest.txt
for rnd in 1 2; do for bg in 0 1; do
  timeout 600 python bench.py --legs td3 --td3-block-graph $bg --no-cpu-baseline --no-parity --steps 50 > $OUT/td3_bg${bg}_$rnd.json 2> $OUT/td3_bg${bg}_$rnd.err
  python - <<PY
import json
d=json.loads([l for l in open("$OUT/td3_bg${bg}_$rnd.json") if l.startswith("{")][-1])
t=d["legs"]["td3"]; print("block_graph=$bg", "td3 %.3f M env-steps/s, %.0f grad steps/s, %.4f ms/step" % (t["value"]/1e6, t["grad_steps_per_s"], t["ms_per_step"]), "roofline", {k: round(v,3) for k,v in t.get("roofline",{}).items() if k in ("frac","kernel_us","update_us","alone_frac","alone_kernel_us")})
PY
done; done
