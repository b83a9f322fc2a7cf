"""Sum rocprofv3 --pmc counter_collection.csv per kernel and counter (per-dispatch averages)."""
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = (r["Kernel_Name"][:40], r["Counter_Name"])
        acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
for (kn, cn), (v, n) in sorted(acc.items()):
    print("%-42s %-26s dispatches %5d  avg/dispatch %.4g" % (kn, cn, n, v / n))
