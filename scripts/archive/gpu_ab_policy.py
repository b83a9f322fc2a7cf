"""A/B of libplenvec builds on the policy leg (bench.py --legs policy: the shipped walking policy in the loop, a contact-rich workload) inside one gpurun call.
usage: python scripts/gpu_ab_policy.py libA.so libB.so ... ("-" = in-tree)"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for rnd in range(2):
    for lib in sys.argv[1:]:
        env = dict(os.environ)
        if lib != "-":
            env["PLENVEC_LIB"] = os.path.join(ROOT, "plen_ml_walk_amd", "csrc", "variants", lib)
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dtype", "f32", "--legs", "policy,td3", "--steps", "200", "--warmup", "20", "--td3-steps", os.environ.get("AB_TD3_STEPS", "1000"), "--no-cpu-baseline", "--no-parity"],
                             env=env, capture_output=True, text=True, timeout=600)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        if not line:
            print(lib, "FAILED", out.stderr[-300:]); continue
        d = json.loads(line[-1])
        print("%-14s policy leg %.3f M env-steps/s (%.4f ms/step)   td3 leg %.3f M   random actions f32 %.3f M" % (lib, d["legs"]["policy"]["value"] / 1e6, d["legs"]["policy"]["ms_per_step"], d["legs"]["td3"]["value"] / 1e6, d["value"] / 1e6), flush=True)
