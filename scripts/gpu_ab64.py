"""A/B of kernel builds on the f64 HEADLINE workload inside ONE gpurun call (boxes differ by a few percent between calls): bench.py's f64 leg
(pipelined 4 x 1024 and one launch per step) with PLENVEC_LIB pointing at each build, two rounds.
usage: python scripts/gpu_ab64.py libA.so libB.so ...   (paths relative to plen_ml_walk_amd/csrc/variants/; "-" = the in-tree libplenvec.so)"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if __name__ == "__main__":
    libs = sys.argv[1:]
    dtype = os.environ.get("AB_DTYPE", "f64")
    for rnd in range(2):
        for lib in libs:
            env = dict(os.environ)
            if lib != "-":
                env["PLENVEC_LIB"] = os.path.join(ROOT, "plen_ml_walk_amd", "csrc", "variants", lib)
            out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dtype", dtype, "--legs", "", "--no-cpu-baseline", "--no-parity", "--steps", "200"],
                                 env=env, capture_output=True, text=True, timeout=300)
            line = [l for l in out.stdout.splitlines() if l.startswith("{")]
            if not line:
                print(lib, "FAILED", out.stderr[-300:]); continue
            d = json.loads(line[-1])
            print("%-28s pipelined %.3f M env-steps/s (%.4f ms/step)   one launch %.4f ms   kernel %.4f ms" % (
                lib, d["value"] / 1e6, d["ms_per_step"], d["config_detail"]["one_launch_per_step"]["ms_per_step"], d["kernel_ms_per_launch"]), flush=True)
