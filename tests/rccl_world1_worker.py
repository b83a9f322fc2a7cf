"""Worker of test_rccl_world_size_one_*: ONE rank, backend nccl (= RCCL) on the box's one MI355X, PLEN_TD3_FORCE_COLLECTIVES=1 so that the
trainers take their multi-rank code path (graph segments with the two gradient all-reduces between them, or the collectives captured inside
the update graph with PLEN_TD3_CAPTURE_ALLREDUCE=1).  The reduction of one rank is the identity, so the run must end with the same
parameters as the plain single-process trainer from the same seeds; also times the critic-bucket all-reduce alone and beside two resident
2048-env launches (DESIGN.md section 11's worry about multi-wave collective kernels beside single-wave env workgroups)."""
import json
import os
import sys
import time
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(kind, collectives, steps, fused=True):
    from plen_ml_walk_amd.train_vec import GraphedVecTD3Trainer, PipelinedVecTD3Trainer
    from plen_ml_walk_amd.vec_env import PlenVecEnv
    from plen_ml_walk_amd.td3 import ReplayBuffer, TD3Agent
    os.environ["PLEN_TD3_FORCE_COLLECTIVES"] = "1" if collectives else "0"
    dev = torch.device("cuda", 0)
    torch.manual_seed(5)
    agent = TD3Agent(26, 18, 1.0, device=dev)
    replay = ReplayBuffer(20000, device=dev)
    replay.seed(0)
    if kind == "pipelined":
        envs = [PlenVecEnv(128, device=dev), PlenVecEnv(128, device=dev)]
        tr = PipelinedVecTD3Trainer(envs, agent, replay, start_timesteps=512, batch_size=256, seed=1000)
    else:
        envs = [PlenVecEnv(256, device=dev)]
        tr = GraphedVecTD3Trainer(envs[0], agent, replay, start_timesteps=512, batch_size=256, updates_per_step=1, seed=1000, fused=fused)
    for _ in range(steps):
        tr.step()
    if kind == "pipelined":
        tr.sync()
    torch.cuda.synchronize()
    flat = torch.cat([p.detach().reshape(-1) for p in list(agent.actor.parameters()) + list(agent.critic.parameters())]).clone()
    info = {"allreduce_mode": tr.allreduce_mode, "collectives": bool(tr.collectives), "grad_steps": tr.grad_steps, "graphs": sorted(str(k) for k in tr._graphs),
            "finite": bool(torch.isfinite(flat).all())}
    for e in envs:
        e.close()
    return flat, info


def latency(dev):
    """dist.all_reduce of the critic bucket (155138 f32 = 620 KB) on the update role stream: alone, and while two 2048-env launches loop on the
    collector streams."""
    from plen_ml_walk_amd.vec_env import PlenVecEnv, worker_stream
    bucket = torch.zeros(155138, device=dev)
    su = worker_stream(dev, "update")
    out = {}

    def timed(n=200):
        with torch.cuda.stream(su):
            for _ in range(20):
                dist.all_reduce(bucket)
            su.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter(); e0.record(su)
            for _ in range(n):
                dist.all_reduce(bucket)
            e1.record(su); su.synchronize()
            return {"host_us_per_call": (time.perf_counter() - t0) / n * 1e6, "device_us_per_call": e0.elapsed_time(e1) / n * 1e3}
    out["alone"] = timed()
    envs = [PlenVecEnv(2048, device=dev), PlenVecEnv(2048, device=dev)]
    acts = torch.rand(2048, 18, device=dev) * 2 - 1
    sc = [worker_stream(dev, 0), worker_stream(dev, 1)]
    for k in range(300):                       # ~60 ms of env launches queued on the collector streams
        for h in range(2):
            with torch.cuda.stream(sc[h]):
                envs[h].step(acts)
    out["beside_two_resident_2048_env_launches"] = timed(100)
    torch.cuda.synchronize()
    for e in envs:
        e.close()
    return out


def main():
    out = sys.argv[1]
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", device_id=dev)                   # RCCL
    res = {"backend": dist.get_backend(), "world": dist.get_world_size()}
    for name, kind, steps, fused in (("graphed_autograd", "graphed", 14, False), ("graphed", "graphed", 14, True), ("pipelined", "pipelined", 24, True)):
        ref, _ = run(kind, False, steps, fused)
        got, info = run(kind, True, steps, fused)
        info["max_abs_param_diff_vs_no_collectives"] = float((ref - got).abs().max())
        res[name] = info
    os.environ["PLEN_TD3_CAPTURE_ALLREDUCE"] = "1"
    ref, _ = run("graphed", False, 14)
    got, info = run("graphed", True, 14)
    info["max_abs_param_diff_vs_no_collectives"] = float((ref - got).abs().max())
    res["graphed_captured"] = info
    os.environ["PLEN_TD3_CAPTURE_ALLREDUCE"] = "0"
    res["allreduce_latency_620KB"] = latency(dev)
    with open(out, "w") as f:
        json.dump(res, f, indent=1)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
