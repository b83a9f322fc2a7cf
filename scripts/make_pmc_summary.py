"""Condense gpurun_out/prof_<tag>/ (scripts/gpu_profile.sh) into the small files kept under profiles/:
  profiles/<tag>_kernel_stats_<dtype>.csv   rocprofv3 --kernel-trace --stats of `bench.py --dtype <dtype> --legs "" --groups 1 --steps 100`
  profiles/<tag>_pmc_summary[_f64].json     per-dispatch averages of the PMC passes + the per-launch figures bench.py reports
FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE is doubled (MI355X_MICROARCH.md, HBM section: gfx950 tallies 128-B read requests at 64 B)."""
import collections, csv, glob, json, os, sys
src, tag = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for dt in ("f64", "f32"):
    rows = []
    for f in glob.glob(os.path.join(src, "stats_" + dt, "**", "*kernel_stats.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            r["Name"] = r["Name"][:120]; rows.append(r)
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    if rows:
        with open(os.path.join(ROOT, "profiles", "%s_kernel_stats_%s.csv" % (tag, dt)), "w", newline="") as f:
            w = csv.writer(f)
            cols = ["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"]
            w.writerow(cols)
            for r in rows[:12]:
                w.writerow([r[k] for k in cols])
            for tf in glob.glob(os.path.join(src, "stats_" + dt, "**", "*kernel_trace.csv"), recursive=True):
                for r in csv.DictReader(open(tf)):
                    if "plen_env_kernel" in r["Kernel_Name"]:
                        f.write("# dispatch: LDS_Block_Size=%s Scratch_Size=%s VGPR_Count=%s Accum_VGPR_Count=%s SGPR_Count=%s Workgroup=%s Grid=%s\n" % (
                            r["LDS_Block_Size"], r["Scratch_Size"], r["VGPR_Count"], r["Accum_VGPR_Count"], r["SGPR_Count"], r["Workgroup_Size_X"], r["Grid_Size_X"]))
                        break
                break
            try:
                f.write("# bench line: " + open(os.path.join(src, "bench_%s.json" % dt)).read().strip()[:1500] + "\n")
            except OSError:
                pass
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(os.path.join(src, "pmc_%s_*" % dt, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = (r["Kernel_Name"][:60], r["Counter_Name"])
            acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
    if not acc:
        continue
    counters = collections.defaultdict(dict)
    for (kn, cn), (v, n) in sorted(acc.items()):
        counters[kn][cn] = {"dispatches": n, "avg_per_dispatch": v / n}
    envk = [k for k in counters if "plen_env_kernel" in k][0]
    c = {k: v["avg_per_dispatch"] for k, v in counters[envk].items()}
    n = 4096
    per = {"envs_per_launch": n,
           "fetch_bytes_corrected": c.get("FETCH_SIZE", 0) * 1024 * 2, "write_bytes": c.get("WRITE_SIZE", 0) * 1024,
           "valu_insts_per_env_step": c.get("SQ_INSTS_VALU", 0) / n, "salu_insts_per_env_step": c.get("SQ_INSTS_SALU", 0) / n,
           "lds_insts_per_env_step": c.get("SQ_INSTS_LDS", 0) / n, "branch_insts_per_env_step": c.get("SQ_INSTS_BRANCH", 0) / n}
    per["hbm_traffic_bytes"] = per["fetch_bytes_corrected"] + per["write_bytes"]
    if c.get("SQ_WAVE_CYCLES"):
        per["valu_active_frac_of_wave_cycles"] = c.get("SQ_ACTIVE_INST_VALU", 0) / c["SQ_WAVE_CYCLES"]
        per["wait_any_frac"] = c.get("SQ_WAIT_ANY", 0) / c["SQ_WAVE_CYCLES"]
        per["wait_inst_any_frac"] = c.get("SQ_WAIT_INST_ANY", 0) / c["SQ_WAVE_CYCLES"]
    out = {"command": "rocprofv3 --pmc <counters> -- python3 scripts/gpu_pmc_target.py 50 4096 %s (one 4096-env launch per step, 30 steps incl. the first ones after reset; "
                      "one pass per counter group: FETCH_SIZE | WRITE_SIZE | SQ_*)" % dt,
           "build": tag, "dtype": dt, "counters": counters, "env_kernel_per_launch": per}
    with open(os.path.join(ROOT, "profiles", "%s_pmc_summary%s.json" % (tag, "" if dt == "f32" else "_f64")), "w") as f:
        json.dump(out, f, indent=1)
    print(dt, json.dumps(per))
