#!/usr/bin/env python3
"""The round-4 ablation's closed-loop column on the POOLED sample of the reference's seven shipped actors (scripts/pin/seven_actors.py explains why: the reference's
last 1000 training episodes were collected by the policies of that period, and actor 3229999 alone is the most fragile of the seven at the start of an episode).
Per variant: 7 actors x 384 episodes at sigma = 0.1 on the CPU oracle -> early falls, full-length fraction, return quantiles, W1 distance to the reference's last-1000
returns.  Build container only (reads the checkpoints from /root/reference).  -> profiles/r04_ablation_pooled.json"""
import glob, json, os, sys, time
import numpy as np, torch
sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from oracle import oracle as O
import pybullet_pin as P
from ablation_r04 import VARIANTS
REF = os.environ.get("PLEN_REFERENCE", "/root/reference")
actors = []
for f in sorted(glob.glob(os.path.join(REF, "plen_bullet/models/*_actor"))):
    sd = torch.load(f, map_location="cpu", weights_only=True)
    actors.append({k: v.double().numpy() for k, v in sd.items()})
out = []
for name, hyp in VARIANTS:
    t0 = time.time()
    hk = {P.HYP[k]: float(v) for k, v in hyp.items()}
    Ls, Rs = [], []
    for i, A in enumerate(actors):
        L, R = O.ensemble(384, actor=A, sigma=0.1, seed=10 + i, hyp=hk)
        Ls.append(L); Rs.append(R)
    s = P.closed_loop_summary(np.concatenate(Ls), np.concatenate(Rs), 0.1)
    s["name"] = name; s["hyp"] = hyp
    out.append(s)
    print("%-64s early %.2f full %.2f mean %+4.0f med %+4.0f q95 %+4.0f W1 %5.1f | %.0fs" % (name[:64], s["early_falls_lt50"], s["full_length"], s["ret_mean"], s["ret_q_5_25_50_75_95"][2],
                                                                                             s["ret_q_5_25_50_75_95"][4], s["w1_to_reference_last1000"], time.time() - t0), flush=True)
json.dump(dict(what=__doc__, reference=dict(last1000_mean=50.4, quantiles_5_25_50_75_95=[-113, -15, 55, 119, 200]), variants=out),
          open(os.path.join(ROOT, "profiles", "r04_ablation_pooled.json"), "w"), indent=1)
