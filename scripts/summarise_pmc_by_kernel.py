"""Mean counter values per dispatch, by kernel, over every *counter_collection.csv under a directory of rocprofv3 --pmc runs.
usage: python scripts/summarise_pmc_by_kernel.py <dir> <out.json> [kernel-name substrings ...]"""
import csv, glob, json, os, sys
from collections import defaultdict
root, out = sys.argv[1], sys.argv[2]
want = sys.argv[3:] or ["k_critic_team", "k_policy_team", "k_wgrad_adam_group"]
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        name = row.get("Kernel_Name", "")
        key = next((w for w in want if w in name), None)
        if key is None:
            continue
        a = acc[key][row["Counter_Name"]]
        a[0] += float(row["Counter_Value"]); a[1] += 1
res = {k: {c: {"mean_per_dispatch": v[0] / v[1], "dispatches": v[1]} for c, v in sorted(cs.items())} for k, cs in acc.items()}
json.dump(res, open(out, "w"), indent=1)
for k, cs in res.items():
    print(k)
    for c, v in cs.items():
        print("   %-34s %16.1f   (%d dispatches)" % (c, v["mean_per_dispatch"], v["dispatches"]))
