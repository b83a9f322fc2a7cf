"""Worker for tests/test_distributed_cpu.py: run under torch.distributed.run with the gloo backend."""
import json
import os
import sys
import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from plen_ml_walk_amd import td3 as T          # noqa: E402
from plen_ml_walk_amd import sharding          # noqa: E402


def make_data(seed, n):
    rng = np.random.default_rng(seed)
    return (rng.normal(size=(n, 26)), rng.uniform(-1, 1, (n, 18)), rng.normal(size=(n, 26)), rng.normal(size=n),
            (rng.uniform(size=n) < 0.1).astype(np.float64))


def fill(buf, data):
    S, A, S2, R, D = data
    buf.add_batch(torch.as_tensor(S), torch.as_tensor(A), torch.as_tensor(S2), torch.as_tensor(R), torch.as_tensor(D))


def run_iterations(agent, buf, noises, B):
    real = torch.randn_like
    for k in range(len(noises)):
        torch.randn_like = lambda x, *a, **kw: noises[k]
        try:
            # deterministic batch: the first B entries of the rank-local buffer
            s = buf.sample
            buf.sample = lambda bs, ind=None: s(bs, ind=torch.arange(B))
            agent.train(buf, B)
            buf.sample = s
        finally:
            torch.randn_like = real


def main():
    out = sys.argv[1]
    dist.init_process_group("gloo")
    rank, world, _ = sharding.world_info()
    B = 32
    torch.manual_seed(100 + rank)               # DIFFERENT init per rank: the constructor must broadcast rank 0's
    agent = T.TD3Agent(26, 18, 1.0, device="cpu")
    buf = T.ReplayBuffer(1000, device="cpu")
    fill(buf, make_data(7 + rank, B))
    gn = torch.Generator().manual_seed(5)
    noises_all = [torch.randn(world * B, 18, generator=gn) for _ in range(2)]
    noises = [n[rank * B:(rank + 1) * B] for n in noises_all]
    init = [p.detach().clone() for p in list(agent.actor.parameters()) + list(agent.critic.parameters())]
    run_iterations(agent, buf, noises, B)
    flat = torch.cat([p.detach().reshape(-1) for p in list(agent.actor.parameters()) + list(agent.critic.parameters())])
    gathered = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    res = {"rank": rank, "world": world, "same_across_ranks": bool(all(torch.equal(gathered[0], g) for g in gathered)),
           "slice": sharding.rank_env_slice(4096 * world + 3, world, rank),
           "slice_exact": sharding.rank_env_slice(4096 * world, world, rank),          # configs[3]: 32768 envs over 8 ranks = 4096 each
           "slice_strong": sharding.rank_env_slice(4096, world, rank),                 # the metric's literal "@4096 envs" in total (strong scaling)
           "seed": sharding.rank_seed(1000, rank),
           "max_time": sharding.max_over_ranks(1.0 + rank, "cpu"), "sum_steps": sharding.sum_over_ranks(10 * (rank + 1), "cpu")}
    if rank == 0:
        # single-process reference: same initial parameters, the concatenated batch, no process group semantics needed
        ref = T.TD3Agent(26, 18, 1.0, device="cpu", data_parallel=False)
        with torch.no_grad():
            for p, q in zip(list(ref.actor.parameters()) + list(ref.critic.parameters()), init):
                p.copy_(q)
        ref.actor_target.load_state_dict(ref.actor.state_dict()); ref.critic_target.load_state_dict(ref.critic.state_dict())
        rbuf = T.ReplayBuffer(1000, device="cpu")
        for r in range(world):
            fill(rbuf, make_data(7 + r, B))
        run_iterations(ref, rbuf, noises_all, world * B)
        rflat = torch.cat([p.detach().reshape(-1) for p in list(ref.actor.parameters()) + list(ref.critic.parameters())])
        res["max_abs_diff_vs_single_process"] = float((rflat - flat).abs().max())
        res["moved"] = float((flat - torch.cat([q.reshape(-1) for q in init])).abs().max())
    with open("%s.rank%d.json" % (out, rank), "w") as f:
        json.dump(res, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
