"""Latency of ONE TD3 update at the reference's batch size (100, plen_td3.py:28) for the three shapes of the same arithmetic: layer by layer (library
GEMMs, ~35 launches), one wave per 16 rows (csrc/td3_rows.hip), a team of 8 waves per 4 rows + grouped weight gradients (csrc/td3_team.hip).
Each is captured as a hipGraph of 32 updates (policy_freq 2: 16 with the delayed policy update) and replayed.
usage: python scripts/gpu_td3_small_batch.py [batch ...]   -> gpurun_out/r04_td3_small_batch.json"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from plen_ml_walk_amd import td3 as T
from plen_ml_walk_amd.td3_fused import FusedTD3

batches = [int(x) for x in sys.argv[1:]] or [100, 256]
out = {}
for B in batches:
    for name, kw in (("layers", dict(rows=False, team=False)), ("rows", dict(rows=True, team=False)), ("team", dict(rows=False, team=True))):
        if os.environ.get("PLEN_SMALL_BATCH_ONLY", name) != name:
            continue
        torch.manual_seed(0)
        ag = T.TD3Agent(26, 18, 1.0, data_parallel=False)
        fz = FusedTD3(ag, seed=1, **kw)
        fz.enable_flat_adam()
        data = torch.randn(100000, 72, device="cuda")
        data[:, 70] = torch.rand(100000, device="cuda"); data[:, 71] = (torch.rand(100000, device="cuda") > 0.02).float()
        tot = torch.tensor(100000, dtype=torch.long, device="cuda")

        def run(n):
            for k in range(n):
                fz.update(data, B, with_policy=(k % 2 == 1), all_reduce=False, total=tot)
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            run(4)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            run(32)
        for _ in range(5):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 40
        for _ in range(reps):
            g.replay()
        torch.cuda.synchronize()
        us = (time.perf_counter() - t0) / (reps * 32) * 1e6
        out["%s_B%d" % (name, B)] = {"us_per_update": us, "updates_per_s": 1e6 / us, "critic_loss": float(ag.last_critic_loss)}
        print("%-7s B %4d  %8.1f us per update  %9.0f updates/s   loss %.4f" % (name, B, us, 1e6 / us, float(ag.last_critic_loss)), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r04_td3_small_batch.json"), "w"), indent=1)
