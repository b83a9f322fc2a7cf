#!/usr/bin/env python3
"""The reference ships SEVEN actor checkpoints of the end of its training run (plen_bullet/models/plen_walk_gazebo_3189999 .. 3249999_actor); the last 1000 training
episodes of its log (results/plen_walk_gazebo_.npy) were collected by the policies of exactly that period under N(0, 0.1) exploration noise (plen_td3.py:101-104).
This script evaluates each of the seven on the CPU oracle under the same noise (768 episodes each, oracle_ensemble) and the pooled sample against the reference's
last-1000 return distribution.  Build container only (reads the checkpoints from /root/reference with weights_only=True).  -> profiles/r04_seven_actors.json"""
import glob, json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.dont_write_bytecode = True
from oracle import oracle as O
import pybullet_pin as P
REF = os.environ.get("PLEN_REFERENCE", "/root/reference")
ref = P.reference_last1000_returns()
out = dict(what=__doc__, reference_last1000=dict(mean=float(ref.mean()), q_5_25_50_75_95=[float(v) for v in np.quantile(ref, [.05, .25, .5, .75, .95])],
                                                  max_over_all_24832_episodes=328.0), actors={})
allL, allR = [], []
for f in sorted(glob.glob(os.path.join(REF, "plen_bullet/models/*_actor"))):
    sd = torch.load(f, map_location="cpu", weights_only=True)
    A = {k: v.double().numpy() for k, v in sd.items()}
    L, R = O.ensemble(768, actor=A, sigma=0.1, seed=3)
    s = P.closed_loop_summary(L, R, 0.1)
    out["actors"][os.path.basename(f)] = s
    allL.append(L); allR.append(R)
    print("%-34s len %3.0f early %.2f full %.2f ret mean %+4.0f q %s W1 %.0f" % (os.path.basename(f), s["mean_length"], s["early_falls_lt50"], s["full_length"], s["ret_mean"],
                                                                                     np.round(s["ret_q_5_25_50_75_95"]), s["w1_to_reference_last1000"]))
L, R = np.concatenate(allL), np.concatenate(allR)
out["pooled"] = P.closed_loop_summary(L, R, 0.1)
s = out["pooled"]
print("%-34s len %3.0f early %.2f full %.2f ret mean %+4.0f q %s W1 %.0f" % ("pooled (7 x 768 episodes)", s["mean_length"], s["early_falls_lt50"], s["full_length"], s["ret_mean"],
                                                                                 np.round(s["ret_q_5_25_50_75_95"]), s["w1_to_reference_last1000"]))
json.dump(out, open(os.path.join(ROOT, "profiles", "r04_seven_actors.json"), "w"), indent=1)
