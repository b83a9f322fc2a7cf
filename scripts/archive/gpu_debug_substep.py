"""GPU bring-up: one physics substep on the HIP path vs the NumPy statement of the same formulation
(tests/np_model.py) and vs the C oracle.  Prints max abs differences per intermediate quantity."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import np_model as nm
from oracle.oracle import OracleEnv, agent_to_env
from plen_ml_walk_amd.vec_env import PlenVecEnv

def collect_states(n, seed=3):
    rng = np.random.default_rng(seed)
    e = OracleEnv(); e.reset()
    S, T = [], []
    tgt = np.zeros(18)
    t = 0
    while len(S) < n:
        if t % 4 == 0:
            a = rng.uniform(-1, 1, 18); tgt = np.array([agent_to_env(j, a[j]) for j in range(18)])
        e.set_targets(tgt)
        S.append(e.get_state()); T.append(tgt.copy())
        e.substep(); t += 1
        s = e.get_state()
        if s[2] < 0.06 or t % 97 == 0:
            e.reset()
    return np.array(S), np.array(T)

def main():
    n = 64
    S, T = collect_states(n)
    for dtype in (torch.float64, torch.float32):
        env = PlenVecEnv(n, dtype=dtype)
        env.set_state(torch.tensor(S))
        dump = env.debug_substeps(torch.tensor(T), nsub=1, dump=True).cpu().numpy().astype(np.float64)
        out = env.get_state().cpu().numpy().astype(np.float64)
        aux = env.get_aux().cpu().numpy()
        worst = {}
        for i in range(n):
            info = {}
            ref = nm.substep(S[i], T[i], info=info)
            o = OracleEnv(); o.set_state(S[i]); o.set_targets(T[i]); o.substep(); oref = o.get_state(); oc = o.contacts()
            d = dump[i]
            q = dict(M=np.abs(d[:576].reshape(24, 24) - info["M"]).max(),
                     tau=np.abs(d[576:600] - info["tau"]).max(),
                     L=np.abs(np.tril(d[640:1216].reshape(24, 24)) - info["L"]).max(),
                     A=np.abs(d[1216:1216 + 2304].reshape(48, 48) - info["A"]).max() / np.abs(info["A"]).max(),
                     b=np.abs(d[3520:3568] - info["b"]).max(),
                     dist=np.abs(d[3568 + 18:3568 + 48].reshape(2, 15)[:, 3::3] - info["dist"]).max(),
                     iters=abs(d[3700] - info["iterations"]),
                     state_np=np.abs(out[i] - ref).max(), state_oracle=np.abs(out[i] - oref).max(),
                     contacts=abs(aux[i, 4] - oc["right"]) + abs(aux[i, 5] - oc["left"]))
            for k, v in q.items():
                worst[k] = max(worst.get(k, 0), float(v))
            if i < 4 or q["state_oracle"] > (1e-6 if dtype == torch.float64 else 1e-2):
                print(dtype, "env", i, {k: "%.2e" % v for k, v in q.items()}, "ncp", oc["ncp"])
        print("WORST", dtype, {k: "%.3e" % v for k, v in worst.items()})
        env.close()

if __name__ == "__main__":
    main()
