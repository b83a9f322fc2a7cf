"""A/B of libplenvec builds on the TD3 leg (bench.py --legs td3) inside one gpurun call.  usage: python scripts/gpu_ab_td3.py libA.so libB.so ... ("-" = in-tree)"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for rnd in range(2):
    for lib in sys.argv[1:]:
        env = dict(os.environ)
        if lib != "-":
            env["PLENVEC_LIB"] = os.path.join(ROOT, "plen_ml_walk_amd", "csrc", "variants", lib)
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dtype", "f32", "--legs", "td3", "--steps", "5", "--warmup", "2", "--no-cpu-baseline", "--no-parity"],
                             env=env, capture_output=True, text=True, timeout=600)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        if not line:
            print(lib, "FAILED", out.stderr[-300:]); continue
        d = json.loads(line[-1])["legs"]["td3"]
        print("%-14s td3 %.3f M env-steps/s  %.0f grad steps/s  %.4f ms/step   batch-100 %.3f M" % (lib, d["value"] / 1e6, d["grad_steps_per_s"], d["ms_per_step"], d["reference_batch_100"]["value"] / 1e6), flush=True)
