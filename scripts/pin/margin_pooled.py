#!/usr/bin/env python3
"""The one model-table hypothesis round 4 left open: does the URDF importer's per-link btCompoundShape carry its own collision margin
(gUrdfDefaultCollisionMargin) into the AABB the link's inertia is computed from?  It cannot be flipped by a run-time switch (it changes the link inertias in the
generated tables), so this script, per margin value m:
  1. regenerates the model tables with PLEN_COMPOUND_MARGIN=m into a scratch directory (tools/extract_model.py, PLEN_MODEL_OUT),
  2. builds the C oracle on them there (a copy of oracle/plen_oracle.c next to the scratch plen_model_raw.h),
  3. evaluates, in a child process bound to that library (PLEN_ORACLE_LIB_DIR), the SAME objectives every run-time variant of round 4 was judged on:
     spawn pins R_0, R_1 and the robust sums over the seven perturbations (scripts/pin/ablation_r04.py: run), the zero-pose stance against init_height
     (scripts/pin/hypothesis_ablation.py: stance_raw), and the POOLED seven-actor closed-loop ensemble at sigma = 0.1 (scripts/pin/ablation_r04_pooled.py:
     7 x 384 episodes, W1 distance to the reference's last 1000 training returns).
Build container only (reads plen.urdf, the foot STLs and the seven shipped actors from /root/reference).  -> profiles/r05_margin_pooled.json
usage: margin_pooled.py [margins ...]      (default 0 0.0005 0.001 0.0015)"""
import glob, json, os, shutil, subprocess, sys, tempfile, time
sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def evaluate():
    """Child: every objective on the oracle library PLEN_ORACLE_LIB_DIR points at.  Prints one JSON line."""
    import numpy as np, torch
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from oracle import oracle as O
    import pybullet_pin as P
    from ablation_r04 import run as spawn_pins
    from hypothesis_ablation import stance_raw
    out = spawn_pins(("margin", {}))
    out["stance"] = stance_raw({})
    REF = os.environ.get("PLEN_REFERENCE", "/root/reference")
    Ls, Rs, per_actor = [], [], []
    for i, f in enumerate(sorted(glob.glob(os.path.join(REF, "plen_bullet/models/*_actor")))):
        sd = torch.load(f, map_location="cpu", weights_only=True)
        A = {k: v.double().numpy() for k, v in sd.items()}
        L, R = O.ensemble(384, actor=A, sigma=0.1, seed=10 + i)
        Ls.append(L); Rs.append(R)
        per_actor.append(dict(actor=os.path.basename(f), early_falls_lt50=float((L < 50).mean()), full_length=float((L >= 500).mean()), ret_mean=float(R.mean())))
    out["pooled_sigma_0.1"] = P.closed_loop_summary(np.concatenate(Ls), np.concatenate(Rs), 0.1)
    out["per_actor"] = per_actor
    # the deterministic episode of actor 3229999 and its low-noise ensemble (the evidence round 4 held against the margin; kept for comparison, not for the verdict)
    L, R = O.ensemble(512, actor=P.SD, sigma=1e-3, seed=2)
    out["actor_3229999_sigma_1e-3"] = P.closed_loop_summary(L, R, 1e-3)
    print(json.dumps(out))


def main():
    margins = [float(x) for x in sys.argv[1:]] or [0.0, 0.0005, 0.001, 0.0015]
    rows = []
    for m in margins:
        t0 = time.time()
        d = tempfile.mkdtemp(prefix="plen_margin_")
        env = dict(os.environ, PLEN_COMPOUND_MARGIN=repr(m), PLEN_MODEL_OUT=d, PYTHONDONTWRITEBYTECODE="1")
        subprocess.check_call([sys.executable, "-B", os.path.join(ROOT, "tools", "extract_model.py")], env=env, stdout=subprocess.DEVNULL)
        shutil.copy(os.path.join(ROOT, "oracle", "plen_oracle.c"), d)
        subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-fvisibility=hidden", "-fno-fast-math", "-ffp-contract=off", "-DORACLE_REAL=double",
                               "-o", os.path.join(d, "libplen_oracle_f64.so"), os.path.join(d, "plen_oracle.c"), "-lm", "-lpthread"])
        r = subprocess.run([sys.executable, "-B", os.path.abspath(__file__), "--evaluate"], env=dict(env, PLEN_ORACLE_LIB_DIR=d), capture_output=True, text=True)
        if r.returncode != 0:
            raise SystemExit(r.stderr[-2000:])
        row = json.loads(r.stdout.strip().splitlines()[-1])
        row["compound_margin_m"] = m
        row["seconds"] = round(time.time() - t0, 1)
        rows.append(row)
        p = row["pooled_sigma_0.1"]
        print("margin %.4f  R0 %.4f R1 %.3f robust %.4f / %.3f | stance %+.3f mm | pooled: early %.2f full %.2f mean %+4.0f W1 %5.1f | 3229999 s=1e-3: len %.0f full %.2f | %.0f s" % (
            m, row["R0"], row["R1"], row["robust_mean_R0"], row["robust_mean_sum_R1_R4"], row["stance"]["init_height_error_mm_at_2s"], p["early_falls_lt50"], p["full_length"],
            p["ret_mean"], p["w1_to_reference_last1000"], row["actor_3229999_sigma_1e-3"]["mean_length"], row["actor_3229999_sigma_1e-3"]["full_length"], row["seconds"]), flush=True)
        shutil.rmtree(d, ignore_errors=True)
    json.dump(dict(what=__doc__, reference=dict(last1000_mean=50.4, quantiles_5_25_50_75_95=[-113, -15, 55, 119, 200], init_height=0.160178937611),
                   sampling_noise_w1="+-3 (round 4: repeated baselines)", rows=rows), open(os.path.join(ROOT, "profiles", "r05_margin_pooled.json"), "w"), indent=1)


if __name__ == "__main__":
    evaluate() if "--evaluate" in sys.argv else main()
