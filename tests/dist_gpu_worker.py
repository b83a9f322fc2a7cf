"""Worker for test_two_ranks_on_one_gpu_*: run under torch.distributed.run with PLEN_DIST_BACKEND=gloo (two ranks sharing ONE GPU cannot use
RCCL; gloo carries the CUDA tensors).  Exercises the multi-rank code path of the hipGraph trainer: parameter broadcast, graph segments with the
two gradient all-reduces between them, per-rank env / replay / RNG."""
import json
import os
import sys
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out, fused = sys.argv[1], int(sys.argv[2])
    pipelined = len(sys.argv) > 3 and sys.argv[3].startswith("pipelined")
    batch = 1024 if (len(sys.argv) > 3 and sys.argv[3].endswith("_block")) else 256          # > 512: the large-batch kernels (csrc/td3_block.hip)
    from plen_ml_walk_amd import sharding
    from plen_ml_walk_amd.train_vec import setup_distributed, GraphedVecTD3Trainer, PipelinedVecTD3Trainer
    from plen_ml_walk_amd.vec_env import PlenVecEnv
    from plen_ml_walk_amd.td3 import ReplayBuffer, TD3Agent
    rank, world = setup_distributed()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    torch.manual_seed(100 + rank)                      # DIFFERENT initial parameters per rank: the constructor must broadcast rank 0's
    envs = [PlenVecEnv(128, device=dev), PlenVecEnv(128, device=dev)] if pipelined else [PlenVecEnv(256, device=dev)]
    env = envs[0]
    agent = TD3Agent(26, 18, 1.0, device=dev)
    init = torch.cat([p.detach().reshape(-1) for p in list(agent.actor.parameters()) + list(agent.critic.parameters())]).clone()
    replay = ReplayBuffer(20000, device=dev)
    if pipelined:
        tr = PipelinedVecTD3Trainer(envs, agent, replay, start_timesteps=512, batch_size=batch, seed=1000 + rank)
    else:
        tr = GraphedVecTD3Trainer(env, agent, replay, start_timesteps=512, batch_size=batch, updates_per_step=1, seed=1000 + rank, fused=bool(fused))
    for _ in range(14 if not pipelined else 24):
        tr.step()
    if pipelined:
        tr.sync()
    torch.cuda.synchronize()
    flat = torch.cat([p.detach().reshape(-1) for p in list(agent.actor.parameters()) + list(agent.critic.parameters())])
    tflat = torch.cat([p.detach().reshape(-1) for p in list(agent.actor_target.parameters()) + list(agent.critic_target.parameters())])
    gathered = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    tg = [torch.zeros_like(tflat) for _ in range(world)]
    dist.all_gather(tg, tflat)
    states = [torch.zeros(256, 26, device=dev) for _ in range(world)]
    dist.all_gather(states, (torch.cat(tr.state, 0) if pipelined else tr.state).contiguous())
    res = {"rank": rank, "world": world, "allreduce_mode": tr.allreduce_mode, "grad_steps": tr.grad_steps, "env_steps": tr.env_steps,
           "params_equal_across_ranks": bool(all(torch.equal(gathered[0], g) for g in gathered)),
           "targets_equal_across_ranks": bool(all(torch.equal(tg[0], g) for g in tg)),
           "init_equal_across_ranks_after_broadcast": None,
           "moved": float((flat - init).abs().max()), "finite": bool(torch.isfinite(flat).all() and torch.isfinite(agent.last_critic_loss)),
           "rank_local_env_states_differ": bool(not torch.equal(states[0], states[1])),
           "graphs": sorted(str(k) for k in tr._graphs), "block_pass": bool(getattr(tr.fused, "_block_pass", False)) if getattr(tr, "fused", None) is not None else False}
    with open("%s.rank%d.json" % (out, rank), "w") as f:
        json.dump(res, f)
    for e in envs:
        e.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
