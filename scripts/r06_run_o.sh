#!/bin/bash
# round 6, GPU call O: the driver's bench command three times (the intermittent abort of call N: a hipGraph destroyed by the garbage collector during a later capture)
set -u
OUT=gpurun_out/r06_o
mkdir -p $OUT
for k in 1 2 3; do
  timeout 1200 python -X faulthandler bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_line_$k.json 2> $OUT/bench_err_$k.txt; echo "run $k rc $?"
  python - <<PY
import json
try:
    d=json.loads([l for l in open("$OUT/bench_line_$k.json") if l.startswith("{")][-1]); c=d["config"]
    print({k_: (round(v) if isinstance(v, float) and v > 100 else v) for k_, v in c.items() if k_.endswith("value") or "updates" in k_ or "grad" in k_})
except Exception as ex:
    print("FAILED", repr(ex)); print(open("$OUT/bench_err_$k.txt").read()[-1500:])
PY
done
