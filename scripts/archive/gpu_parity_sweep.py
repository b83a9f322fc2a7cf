"""Parity sweep of the final build: all 4096 envs of a launch against the f64 oracle for several seeds and configurations (the quantities
tests/test_full_size_gpu.py asserts for one seed), written to gpurun_out/parity_sweep.json.   usage: python scripts/gpu_parity_sweep.py [seeds]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from oracle import oracle
oracle.build()
from test_full_size_gpu import _run

seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
out = {}
for name, dtype, T, kw in (("f64, rolling friction off, 12 steps", torch.float64, 12, dict(rolling=0.0)),
                           ("f64, rolling friction off, DR, 12 steps", torch.float64, 12, dict(rolling=0.0, dr=True)),
                           ("f64, reference configuration, 1 step", torch.float64, 1, dict()),
                           ("f64, reference configuration, DR, 1 step", torch.float64, 1, dict(dr=True)),
                           ("f32, rolling friction off, 12 steps", torch.float32, 12, dict(rolling=0.0)),
                           ("f32, reference configuration, 1 step", torch.float32, 1, dict())):
    rows = []
    for seed in range(seeds):
        err, rerr, flags, contacts = _run(dtype, T, seed=seed, **kw)
        rows.append(dict(seed=seed, obs_err_median_first=float(np.median(err[0])), obs_err_median_last=float(np.median(err[-1])), obs_err_p90_last=float(np.quantile(err[-1], .9)),
                         obs_err_max_first=float(err[0].max()), within_1e4_last=float((err[-1] <= 1e-4).mean()), reward_err_median_last=float(np.median(rerr[-1])),
                         done_flags_equal=float(flags.mean()), contact_flags_equal=float(contacts.mean())))
    out[name] = rows
    print("%-42s" % name, " | ".join("s%d med %.1e/%.1e <=1e-4 %.3f flags %.4f" % (r["seed"], r["obs_err_median_first"], r["obs_err_median_last"], r["within_1e4_last"], r["done_flags_equal"]) for r in rows), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "parity_sweep.json"), "w"), indent=1)
