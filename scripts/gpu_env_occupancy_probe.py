"""f32 env throughput against the number of env wavefronts kept resident: n envs as `groups` sub-batches on their own streams, back to back (PlenVecEnvPipelined).
4096 envs fill every wave slot of the chip (4 per SIMD at 128 VGPRs); 3072 leave one slot per SIMD (and 41 KB of LDS per compute unit) free -- what a learner
that is to run BESIDE the envs needs.  usage: python scripts/gpu_env_occupancy_probe.py -> gpurun_out/r05_env_occupancy.json"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import torch
from plen_ml_walk_amd.vec_env import PlenVecEnvPipelined
dev = torch.device("cuda", 0)
out = []
for dtype in (torch.float32, torch.float64):
    for n, groups in ((4096, 2), (4096, 4), (3072, 3), (3072, 2), (3072, 1), (2048, 2), (2048, 1), (1024, 1), (3584, 2), (2560, 2)):
        env = PlenVecEnvPipelined(n, groups=groups, device=dev, dtype=dtype)
        env.reset()
        g = torch.Generator(device=dev).manual_seed(0)
        acts = torch.rand(16, n, 18, generator=g, device=dev) * 2 - 1
        for t in range(20):
            env.step_async(acts[t % 16])
        env.sync(); torch.cuda.synchronize()
        steps = 300
        t0 = time.perf_counter()
        for t in range(steps):
            env.step_async(acts[t % 16])
        env.sync(); torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        row = {"dtype": str(dtype), "envs": n, "groups": groups, "ms_per_step": dt / steps * 1e3, "env_steps_per_s": n * steps / dt, "launch_ms": dt / steps * 1e3}
        out.append(row)
        print("%s n %4d groups %d: %.3f ms per step of all envs, %.2f M env-steps/s" % (str(dtype)[-7:], n, groups, row["ms_per_step"], row["env_steps_per_s"] / 1e6), flush=True)
        env.close()
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r05_env_occupancy.json"), "w"), indent=1)
