"""Condense a rocprofv3 --kernel-trace --stats output directory into a small CSV for profiles/."""
import csv, glob, os, sys
src, dst = sys.argv[1], sys.argv[2]
rows = []
for f in glob.glob(os.path.join(src, "**", "*_kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        r["Name"] = r["Name"][:120]
        rows.append(r)
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
with open(dst, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
    for r in rows[:12]:
        w.writerow([r[k] for k in ["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"]])
# first dispatch of our kernel: resources as the runtime saw them
for f in glob.glob(os.path.join(src, "**", "*_kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "plen_env_kernel" in r["Kernel_Name"]:
            with open(dst, "a") as g:
                g.write("# dispatch: LDS_Block_Size=%s Scratch_Size=%s VGPR_Count=%s Accum_VGPR_Count=%s SGPR_Count=%s Workgroup=%s Grid=%s\n" % (
                    r["LDS_Block_Size"], r["Scratch_Size"], r["VGPR_Count"], r["Accum_VGPR_Count"], r["SGPR_Count"], r["Workgroup_Size_X"], r["Grid_Size_X"]))
            break
