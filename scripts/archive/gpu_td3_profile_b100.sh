#!/bin/bash
# kernel stats of the pipelined trainer at batch 100 (why is it slower than batch 4096?)
REPO=$(pwd); OUT=$REPO/gpurun_out/prof_td3_b100; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $REPO/scripts/gpu_pipeline_probe.py 100 > $OUT/log.txt 2>&1
cd $REPO
python3 - <<'PY'
import csv, glob
rows=[]
for f in glob.glob("gpurun_out/prof_td3_b100/stats/**/*kernel_stats.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time ms", tot/1e6)
for r in rows[:14]:
    print("%8d calls %9.1f us avg %6.2f%%  %s" % (int(r["Calls"]), float(r["AverageNs"])/1e3, 100*float(r["TotalDurationNs"])/tot, r["Name"][:100]))
PY
