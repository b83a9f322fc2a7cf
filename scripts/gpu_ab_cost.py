import json, os, subprocess, sys
ROOT = "/root/repo" if os.path.exists("/root/repo/bench.py") else os.environ["GRAFT_REPO_ROOT"]
lib = sys.argv[1]
for dtype in ("f64", "f32"):
    for cost in sys.argv[2].split(";"):
        env = dict(os.environ, PLENVEC_LIB=os.path.join(ROOT, "plen_ml_walk_amd", "csrc", "variants", lib), PLENVEC_COST=cost)
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dtype", dtype, "--legs", "", "--no-cpu-baseline", "--no-parity", "--steps", "200", "--groups", "1" if dtype == "f64" else "0"],
                             env=env, capture_output=True, text=True, timeout=300)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        d = json.loads(line[-1])
        print(dtype, cost, "%.3f M (%.4f ms/step)" % (d["value"] / 1e6, d["ms_per_step"]), flush=True)
