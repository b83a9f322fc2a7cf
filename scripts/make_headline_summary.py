"""Condense gpurun_out/prof_<tag>/ (scripts/gpu_profile_headline.sh) into profiles/:
  profiles/<tag>_groups{4,2}_kernel_stats_<dtype>.csv   rocprofv3 --kernel-trace --stats of the driver's command (`bench.py --gpus 1 --steps 100 --warmup 10`, extra legs
                                                       off) in the HEADLINE schedule: per-launch durations of the sub-batch launches, their concurrency, and the
                                                       throughput the trace itself implies (env-steps of the traced launches / the span they cover)
  profiles/<tag>_groups_pmc_summary_<dtype>.json        PMC passes of the same schedule, per LAUNCH (1024 or 2048 envs) and per env-step
FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE is doubled (MI355X_MICROARCH.md, HBM section: gfx950 tallies 128-B read requests at 64 B)."""
import collections, csv, glob, json, os, sys
import numpy as np
src, tag = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for dt, groups in (("f64", 4), ("f32", 2)):
    n_sub = 4096 // groups
    rows = []
    for f in glob.glob(os.path.join(src, "stats_" + dt, "**", "*kernel_stats.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            r["Name"] = r["Name"][:120]; rows.append(r)
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    trace = []
    for tf in glob.glob(os.path.join(src, "stats_" + dt, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(tf)):
            if "plen_env_kernel" in r["Kernel_Name"] and int(r["Grid_Size_X"]) == n_sub * 64:
                trace.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r))
    if not rows:
        continue
    out_csv = os.path.join(ROOT, "profiles", "%s_groups%d_kernel_stats_%s.csv" % (tag, groups, dt))
    with open(out_csv, "w", newline="") as f:
        w = csv.writer(f)
        cols = ["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"]
        w.writerow(cols)
        for r in rows[:12]:
            w.writerow([r[k] for k in cols])
        if trace:
            trace.sort()
            # the timed region's launches: drop the first 15 % (reset builds, warm-up) and the single-launch comparison loop (other grid size, already filtered)
            tr = trace[int(0.15 * len(trace)):]
            st = np.array([t[0] for t in tr], dtype=np.float64); en = np.array([t[1] for t in tr], dtype=np.float64)
            dur = en - st
            span = en.max() - st.min()
            # concurrency: time-weighted number of env launches in flight over the span
            ev = sorted([(s, 1) for s in st] + [(e, -1) for e in en])
            cur, last, acc = 0, ev[0][0], 0.0
            hist = collections.Counter()
            for t, d in ev:
                acc += cur * (t - last); hist[cur] += t - last; cur += d; last = t
            r0 = tr[0][2]
            f.write("# headline schedule: %d x %d envs on %d HIP streams; %d sub-batch launches of the timed region traced\n" % (groups, n_sub, groups, len(tr)))
            f.write("# per-launch duration ns: mean %.0f median %.0f min %.0f max %.0f\n" % (dur.mean(), np.median(dur), dur.min(), dur.max()))
            f.write("# mean launches in flight %.2f; fraction of the span with k in flight: %s\n" % (acc / span, json.dumps({int(k): round(v / span, 3) for k, v in sorted(hist.items())})))
            f.write("# throughput implied by the trace: %d launches x %d envs / %.3f ms = %.3f M env-steps/s; per vector step of 4096 envs: %.4f ms (span of the traced launches incl. the barriers between the 100-step blocks; the same run's host-clock ms_per_step is in the bench line below)\n" % (
                len(tr), n_sub, span / 1e6, len(tr) * n_sub / (span / 1e9) / 1e6, span / 1e6 / (len(tr) / groups)))
            f.write("# dispatch: LDS_Block_Size=%s Scratch_Size=%s VGPR_Count=%s Accum_VGPR_Count=%s SGPR_Count=%s Workgroup=%s Grid=%s\n" % (
                r0["LDS_Block_Size"], r0["Scratch_Size"], r0["VGPR_Count"], r0["Accum_VGPR_Count"], r0["SGPR_Count"], r0["Workgroup_Size_X"], r0["Grid_Size_X"]))
        try:
            f.write("# bench line: " + open(os.path.join(src, "bench_%s.json" % dt)).read().strip()[:1800] + "\n")
        except OSError:
            pass
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(os.path.join(src, "pmc_%s_*" % dt, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "plen_env_kernel" not in r["Kernel_Name"] or int(r["Grid_Size"]) != n_sub * 64:
                continue
            k = r["Counter_Name"]
            acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
    if not acc:
        continue
    c = {k: v[0] / v[1] for k, v in acc.items()}
    per = {"schedule": "%d x %d envs on %d streams" % (groups, n_sub, groups), "envs_per_launch": n_sub, "dispatches": {k: v[1] for k, v in acc.items()},
           "fetch_bytes_corrected": c.get("FETCH_SIZE", 0) * 1024 * 2, "write_bytes": c.get("WRITE_SIZE", 0) * 1024,
           "valu_insts_per_env_step": c.get("SQ_INSTS_VALU", 0) / n_sub, "salu_insts_per_env_step": c.get("SQ_INSTS_SALU", 0) / n_sub,
           "lds_insts_per_env_step": c.get("SQ_INSTS_LDS", 0) / n_sub, "branch_insts_per_env_step": c.get("SQ_INSTS_BRANCH", 0) / n_sub}
    per["hbm_traffic_bytes"] = per["fetch_bytes_corrected"] + per["write_bytes"]
    per["hbm_traffic_bytes_per_4096_env_step"] = per["hbm_traffic_bytes"] * groups
    if c.get("SQ_WAVE_CYCLES"):
        per["valu_active_frac_of_wave_cycles"] = c.get("SQ_ACTIVE_INST_VALU", 0) / c["SQ_WAVE_CYCLES"]
        per["wait_any_frac"] = c.get("SQ_WAIT_ANY", 0) / c["SQ_WAVE_CYCLES"]
        per["wait_inst_any_frac"] = c.get("SQ_WAIT_INST_ANY", 0) / c["SQ_WAVE_CYCLES"]
    out = {"command": "rocprofv3 --pmc <counters> -- python3 scripts/gpu_pmc_target_groups.py 4096 %d %s (the headline schedule, 40 steps incl. the first ones after reset; one pass per counter group; "
                      "PMC serialises the dispatches, so concurrency effects are in the kernel trace, not here)" % (groups, dt),
           "build": tag, "dtype": dt, "counters_avg_per_dispatch": c, "env_kernel_per_launch": per}
    with open(os.path.join(ROOT, "profiles", "%s_groups_pmc_summary_%s.json" % (tag, dt)), "w") as f:
        json.dump(out, f, indent=1)
    print(dt, json.dumps(per))
