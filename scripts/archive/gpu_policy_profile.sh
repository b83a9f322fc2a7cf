#!/bin/bash
# rocprofv3 kernel stats of the policy leg alone (bench.py --legs policy)
REPO=$(pwd); OUT=$REPO/gpurun_out/prof_policy; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $REPO/bench.py --dtype f32 --legs policy --steps 5 --warmup 2 --td3-steps 200 --no-cpu-baseline --no-parity > $OUT/bench.json 2> $OUT/bench.err
cd $REPO
python3 - <<'PY'
import csv, glob
rows=[]
for f in glob.glob("gpurun_out/prof_policy/stats/**/*kernel_stats.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time ms", tot/1e6, "distinct kernels", len(rows), "calls", sum(int(r["Calls"]) for r in rows))
for r in rows[:12]:
    print("%8d calls %9.1f us avg %6.2f%%  %s" % (int(r["Calls"]), float(r["AverageNs"])/1e3, 100*float(r["TotalDurationNs"])/tot, r["Name"][:100]))
PY
