"""Does what ran earlier on a HIP stream change how fast the pipelined TD3 trainer runs on it?  (bisecting a 0.74 -> 2.0 ms/step slowdown of
bench.py's td3 leg when it reuses the streams of a 4-sub-batch f64 env leg)

usage: python scripts/gpu_stream_reuse_probe.py CASE [ROLE_ORDER]      one process per case; ROLE_ORDER -> PLEN_STREAM_ROLE_ORDER
  fresh | f64g4 | f64g2 | f64g1 | f32g4 | f32g2 | f64g4-own (env on its own fresh streams) | f64g4-keep (env left open)
  touch:0,1,x,u,...  = first-use order of the role streams (k = collector / sub-batch k, u = update, x = throw-away), then the trainer alone"""
import os
import sys
import time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def trainer_ms(dev, n=4096, H=2, steps=100):
    from plen_ml_walk_amd.vec_env import PlenVecEnv
    from plen_ml_walk_amd.td3 import ReplayBuffer, TD3Agent
    from plen_ml_walk_amd.train_vec import PipelinedVecTD3Trainer
    torch.manual_seed(0)
    agent = TD3Agent(26, 18, 1.0, device=dev)
    replay = ReplayBuffer(1000000, device=dev)
    envs = [PlenVecEnv(n // H, device=dev) for _ in range(H)]
    tr = PipelinedVecTD3Trainer(envs, agent, replay, start_timesteps=10000, expl_noise=0.1, batch_size=4096, seed=1000)
    for _ in range(40):
        tr.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        tr.step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3, [s.stream_id for s in tr.streams] + [tr.su.stream_id]


def main():
    case = sys.argv[1]
    if len(sys.argv) > 2:
        os.environ["PLEN_STREAM_ROLE_ORDER"] = sys.argv[2]
    dev = torch.device("cuda", 0)
    torch.zeros(1, device=dev)
    from plen_ml_walk_amd.vec_env import PlenVecEnvPipelined
    keep = None
    env_ms = None
    if case.startswith("touch:"):
        # first use (= hardware queue creation) of the role streams in a given order: k = collector / sub-batch stream k, u = the update stream, x = a throw-away stream
        from plen_ml_walk_amd.vec_env import worker_stream
        ids = []
        for tok in case[6:].split(","):
            ids.append(tok)
        os.environ["PLEN_STREAM_ROLE_ORDER"] = ",".join({"u": "update"}.get(t, t) for t in ids)
        worker_stream(dev, 0)
    elif case != "fresh":
        dtype = torch.float64 if case.startswith("f64") else torch.float32
        g = int(case[4])
        env = PlenVecEnvPipelined(4096, groups=g, device=dev, dtype=dtype)
        if case.endswith("-own"):
            env.streams = [torch.cuda.Stream(device=dev) for _ in range(g)]
        env.reset()
        act = torch.rand(16, 4096, 18, device=dev) * 2 - 1
        for t in range(20):
            env.step_async(act[t % 16])
        env.sync()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for t in range(100):
            env.step_async(act[t % 16])
        env.sync()
        torch.cuda.synchronize()
        env_ms = (time.perf_counter() - t0) / 100 * 1e3
        ids = [s.stream_id for s in env.streams]
        if case.endswith("-keep"):
            keep = env
        else:
            env.close()
            del env
    else:
        ids = []
    ms, tids = trainer_ms(dev)
    print("%-12s env streams %s %s ms/step; trainer streams %s: %.3f ms/step" % (case, ids, "%.3f" % env_ms if env_ms else "-", tids, ms))


if __name__ == "__main__":
    main()
