"""Pipelined TD3 trainer with the update on CUs of its own (hipExtStreamCreateWithCUMask): does the chain of ~50 short update kernels run
faster when it never has to wait for an env wave to retire, and what does taking K CUs away cost the env kernels?

usage: python scripts/gpu_cumask_probe.py K [batch] [env_mask: comp|all]     K = CUs reserved for the update stream (0 = plain streams)"""
import ctypes as C
import os
import sys
import time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch


def masked_stream(hip, bits, dev):
    words = (C.c_uint32 * 8)(*[(bits >> (32 * i)) & 0xffffffff for i in range(8)])
    st = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(st), 8, words)
    assert rc == 0, "hipExtStreamCreateWithCUMask -> %d" % rc
    s = torch.cuda.ExternalStream(st.value, device=dev)
    with torch.cuda.stream(s):
        torch.zeros(1, device=dev)
    return s


def main():
    K = int(sys.argv[1])
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    env_mask = sys.argv[3] if len(sys.argv) > 3 else "comp"
    dev = torch.device("cuda", 0)
    torch.zeros(1, device=dev)
    from plen_ml_walk_amd import vec_env
    from plen_ml_walk_amd.vec_env import PlenVecEnv
    from plen_ml_walk_amd.td3 import ReplayBuffer, TD3Agent
    from plen_ml_walk_amd.train_vec import PipelinedVecTD3Trainer
    if K > 0:
        hip = C.CDLL("libamdhip64.so")
        full = (1 << 256) - 1
        upd = (1 << K) - 1
        env = full & ~upd if env_mask == "comp" else full
        vec_env.worker_stream(dev, 0)            # the registry's own streams first (pipe placement of the masked ones then follows)
        vec_env._WORKER_STREAMS[(0, "0")] = masked_stream(hip, env, dev)
        vec_env._WORKER_STREAMS[(0, "1")] = masked_stream(hip, env, dev)
        vec_env._WORKER_STREAMS[(0, "update")] = masked_stream(hip, upd, dev)
    n, H = 4096, 2
    torch.manual_seed(0)
    agent = TD3Agent(26, 18, 1.0, device=dev)
    replay = ReplayBuffer(1000000, device=dev)
    envs = [PlenVecEnv(n // H, device=dev) for _ in range(H)]
    tr = PipelinedVecTD3Trainer(envs, agent, replay, start_timesteps=10000, expl_noise=0.1, batch_size=batch, seed=1000)
    for _ in range(40):
        tr.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    steps = 200
    for _ in range(steps):
        tr.step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    print("update CUs %3d env mask %s batch %d: %.3f ms/step = %.2f M env-steps/s, %.0f grad steps/s" % (K, env_mask, batch, ms, n / ms / 1e3, 1e3 / ms))


if __name__ == "__main__":
    main()
