"""Why do the env launches take 430 us beside the learner and 303 us alone (scripts/gpu_td3_timeline.py)?  A -DPGS_STAMPS build of the env kernel stamps every wave's
lifetime in SHADER cycles (aux[6]); HIP events give the launch's wall time.  Two collectors' worth of f32 envs (2 x 2048 on two streams) step back to back while a third
stream replays large-batch TD3 updates, or not: if the waves' cycle counts stay and the wall time grows, the chip's clock dropped (the matrix cores' power draw);
if the cycle counts grow, the waves waited for each other.   usage: python scripts/gpu_clock_probe.py -> gpurun_out/r05_clock_probe.json"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
from plen_ml_walk_amd.build import build_variant
os.environ["PLENVEC_LIB"] = build_variant("stamps", ["-DPGS_STAMPS"])
import numpy as np, torch
from plen_ml_walk_amd.vec_env import PlenVecEnv, worker_stream
from plen_ml_walk_amd import td3 as T
from plen_ml_walk_amd.td3_fused import FusedTD3
dev = torch.device("cuda", 0)
n = 2048
envs = [PlenVecEnv(n, device=dev) for _ in range(2)]
streams = [worker_stream(dev, h) for h in range(2)]
su = worker_stream(dev, "update")
for e in envs:
    e.reset()
g = torch.Generator(device=dev).manual_seed(1)
acts = torch.rand(32, n, 18, device=dev, generator=g) * 2 - 1
torch.manual_seed(0)
ag = T.TD3Agent(26, 18, 1.0, data_parallel=False)
data = torch.randn(100000, 72, device=dev); data[:, 71] = 1.0
tot = torch.tensor(100000, dtype=torch.long, device=dev)
out = []


def learner_graph(B, kw):
    fz = FusedTD3(ag, seed=1, **kw)
    fz.enable_flat_adam()
    k = [0]

    def upd():
        fz.update(data, B, with_policy=(k[0] % 2 == 1), all_reduce=False, total=tot)
        k[0] += 1
    with torch.cuda.stream(su):
        upd(); upd()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, stream=su):
        for _ in range(8):
            upd()
    return gr


import ctypes as C
from plen_ml_walk_amd import td3_fused as F
_lib = F.load()


def spin_graph(wgs, iters):
    """~100 us of matrix-core instructions on `wgs` single-wave workgroups, 8 launches per graph"""
    sink = torch.zeros(wgs * 64, device=dev)
    with torch.cuda.stream(su):
        _lib.plentd3_dev_mfma_spin(wgs, iters, C.c_void_p(sink.data_ptr()), C.c_void_p(su.cuda_stream))
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, stream=su):
        for _ in range(8):
            _lib.plentd3_dev_mfma_spin(wgs, iters, C.c_void_p(sink.data_ptr()), C.c_void_p(su.cuda_stream))
    gr._keep = sink
    return gr


for label, B, kw in (("envs alone", 0, None), ("+ MFMA only, 1024 waves x ~100 us", -1, (1024, 200)), ("+ MFMA only, 256 waves x ~100 us", -1, (256, 200)), ("+ block updates, batch 4096", 4096, dict(rows=False, team=False, block=True)), ("+ block updates, batch 1024", 1024, dict(rows=False, team=False, block=True)),
                     ("+ row-kernel updates, batch 4096", 4096, dict(rows=True, team=False, block=False)), ("envs alone again", 0, None)):
    gr = (spin_graph(*kw) if B < 0 else learner_graph(B, kw)) if B else None
    steps = 120
    for t in range(20):
        for h in range(2):
            with torch.cuda.stream(streams[h]):
                envs[h].step(acts[(t + h) % 32])
    torch.cuda.synchronize()
    with torch.cuda.stream(streams[0]):
        envs[0].timing_begin()
    t0 = time.perf_counter()
    for t in range(steps):
        for h in range(2):
            with torch.cuda.stream(streams[h]):
                envs[h].step(acts[(t + h) % 32])
        if gr is not None and t % 3 == 0:
            with torch.cuda.stream(su):
                gr.replay()
    with torch.cuda.stream(streams[0]):
        km, ln = envs[0].timing_end()
    for s in streams:
        s.synchronize()
    wall = (time.perf_counter() - t0) / steps
    torch.cuda.synchronize()
    aux = envs[0].get_aux().cpu().numpy()
    d = aux[:, 6].astype(np.float64); d = d[d > 1000]
    row = {"case": label, "env_step_wall_us": wall * 1e6, "launch_us_events": km / max(ln, 1) * 1e3, "wave_cycles_median": float(np.median(d)), "wave_cycles_p90": float(np.percentile(d, 90)),
           "wave_cycles_max": float(d.max()), "implied_clock_GHz_of_longest_wave": float(d.max() / (km / max(ln, 1) * 1e3) / 1e3)}
    out.append(row)
    print("%-34s step %6.1f us  launch %6.1f us  wave cycles median %7.0f p90 %7.0f max %7.0f  => longest wave / launch time = %.2f GHz" % (
        label, row["env_step_wall_us"], row["launch_us_events"], row["wave_cycles_median"], row["wave_cycles_p90"], row["wave_cycles_max"], row["implied_clock_GHz_of_longest_wave"]), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r05_clock_probe.json"), "w"), indent=1)
