"""Why the TD3 leg's env launches are slower than the random-action benchmark's: the per-env cost estimate (aux[7], issue slots of the last step) and the contact flags
under uniform random actions and under a freshly initialised actor + N(0, 0.1) noise (what the TD3 leg's collectors play)."""
import os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from plen_ml_walk_amd.vec_env import PlenVecEnv
from plen_ml_walk_amd.td3 import TD3Agent
dev = torch.device("cuda:0")
torch.manual_seed(0)
agent = TD3Agent(26, 18, 1.0, device=dev)
n = 4096
for mode in ("uniform random actions", "random-init actor + N(0, 0.1)"):
    env = PlenVecEnv(n, device=dev)
    obs = env.reset().to(torch.float32)
    g = torch.Generator(device=dev).manual_seed(1)
    acc = torch.zeros(3, device=dev); steps = 0
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    tt = 0.0
    for t in range(300):
        if mode.startswith("uniform"):
            a = torch.rand(n, 18, generator=g, device=dev) * 2 - 1
        else:
            with torch.no_grad():
                a = (agent.actor(obs) + 0.1 * torch.randn(n, 18, generator=g, device=dev)).clamp(-1, 1)
        ev0.record()
        o, r, d, info = env.step(a)
        ev1.record()
        obs = info["obs"].to(torch.float32).clone() if isinstance(info, dict) and "obs" in info else o.to(torch.float32).clone()
        if t >= 100:
            torch.cuda.synchronize(); tt += ev0.elapsed_time(ev1)
            ax = env.get_aux().to(torch.float32)
            acc += torch.stack([ax[:, 7].mean(), ax[:, 4].mean(), ax[:, 5].mean()]); steps += 1
    m = (acc / steps).tolist()
    print("%-32s one launch of %d envs: %.3f ms; cost estimate %.0f issue slots per env-step; right / left foot touching in %.2f / %.2f of the env-steps" % (mode, n, tt / steps, m[0], m[1], m[2]), flush=True)
    env.close()
