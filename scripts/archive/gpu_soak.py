"""Soak: 4096 envs x many steps of random actions in several configurations; every 500 steps the whole state and the outputs must be
finite, quaternions normalised, heights bounded.  usage: python scripts/gpu_soak.py [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from plen_ml_walk_amd.vec_env import PlenVecEnvPipelined
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
n = 4096
for name, kw, dr in (("reference config", {}, False), ("joint_act", {"joint_act": True}, False), ("v0 head + DR", {"cfg_overrides": {"reward_head": 1}}, True),
                     ("f64", {"dtype": torch.float64}, False)):
    k = steps if "f64" not in name else steps // 4
    env = PlenVecEnvPipelined(n, groups=2, **kw)
    if dr:
        g0 = torch.Generator(device="cuda").manual_seed(7)
        env.set_params(0.8 + 0.4 * torch.rand(n, generator=g0, device="cuda"), 0.4 + 0.6 * torch.rand(n, generator=g0, device="cuda"))
    env.reset()
    g = torch.Generator(device="cuda").manual_seed(11)
    amp = 1.7 if kw.get("joint_act") else 1.0            # joint_act takes joint angles: exercise the +-1.7 limits too
    bad, ends, t0 = 0, 0, time.time()
    for t in range(k):
        a = (torch.rand(n, 18, device="cuda", generator=g) * 2 - 1) * amp
        obs, rew, done, info = env.step(a)
        ends += int((done != 0).sum()) if t % 50 == 0 else 0
        if t % 500 == 499:
            st = env.get_state()
            qn = st[:, 3:7].norm(dim=1)
            ok = torch.isfinite(st).all() and torch.isfinite(obs).all() and torch.isfinite(rew).all() and torch.isfinite(info["obs"]).all() \
                and bool(((qn - 1).abs() < 1e-3).all()) and bool((st[:, 2].abs() < 1.0).all()) and bool((st[:, 7:13].abs() <= 100.0).all())
            bad += 0 if ok else 1
    torch.cuda.synchronize()
    print("%-18s %6d steps x %d envs: %s  (%.1f s, episode ends sampled %d, non-finite guard resets %d)" % (
        name, k, n, "OK" if bad == 0 else "%d BAD CHECKS" % bad, time.time() - t0, ends, env.nonfinite_count()), flush=True)
    env.close()
