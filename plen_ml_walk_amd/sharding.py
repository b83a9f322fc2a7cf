"""One process per GPU: which environments a rank owns, and the two tiny reductions the benchmarks
need.  Environments are independent (no cross-env term anywhere in PlenWalkEnv.step), so the env
step needs NO collective; only TD3 gradients are exchanged (td3._FlatGrads)."""
import os
import torch


def world_info():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def rank_env_slice(total_envs, world, rank):
    """Contiguous block partition: env i lives on rank i // ceil(total/world)."""
    per = (total_envs + world - 1) // world
    lo = min(total_envs, rank * per)
    return lo, min(total_envs, lo + per)


def rank_seed(seed, rank):
    return seed + rank


def max_over_ranks(value, device):
    """Timing rule of bench.py: the slowest rank defines the step time."""
    import torch.distributed as dist
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t[0])


def sum_over_ranks(value, device):
    import torch.distributed as dist
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t[0])
