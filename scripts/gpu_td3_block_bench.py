"""GPU time of ONE TD3 update at the benchmark's batch (4096, BASELINE.json configs[2]) on an otherwise idle MI355X, for the shapes of the same arithmetic:
layer by layer (library GEMMs), one wave per 16 rows (csrc/td3_rows.hip), 16 rows per 4-wave workgroup with packed weights (csrc/td3_block.hip) -- whole
updates (graph of 8: 4 with the delayed policy update) and the critic / policy pass kernels alone (their own graphs), priced against the fp32 matrix peak.
usage: python scripts/gpu_td3_block_bench.py [batch]   -> gpurun_out/r05_td3_block_bench.json"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from plen_ml_walk_amd import td3 as T
from plen_ml_walk_amd.td3_fused import FusedTD3

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
CRITIC_MAC, POLICY_MAC = 516096, 26 * 256 + 256 * 256 + 256 * 18 + 44 * 256 + 256 * 256 + 256 * 256 + 256 * 18 + 18 * 256 + 256 * 256      # per batch row
PEAK = 157.3e12
out = {"batch": B, "critic_pass_flop": 2 * CRITIC_MAC * B, "policy_pass_flop": 2 * POLICY_MAC * B, "peak_flops": PEAK}


def timed(fn, inner, reps=30):
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        fn(); fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(inner):
            fn()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (reps * inner) * 1e6


for name, kw in (("layers", dict(rows=False, team=False, block=False)), ("rows", dict(rows=True, team=False, block=False)),
                 ("block", dict(rows=False, team=False, block=True)), ("block_rows_wgrad", dict(rows=True, team=False, block=True))):
    torch.manual_seed(0)
    ag = T.TD3Agent(26, 18, 1.0, data_parallel=False)
    fz = FusedTD3(ag, seed=1, **kw)
    fz.enable_flat_adam()
    data = torch.randn(100000, 72, device="cuda")
    data[:, 70] = torch.rand(100000, device="cuda"); data[:, 71] = (torch.rand(100000, device="cuda") > 0.02).float()
    tot = torch.tensor(100000, dtype=torch.long, device="cuda")
    k = [0]

    def update():
        fz.update(data, B, with_policy=(k[0] % 2 == 1), all_reduce=False, total=tot)
        k[0] += 1
    k[0] = 0
    us = timed(update, 8)
    row = {"us_per_update": us, "critic_loss": float(ag.last_critic_loss)}

    def critic():
        fz._zeroed = {"critic": True}
        fz.critic_backward(data, B, total=tot)
    row["us_critic_backward"] = timed(critic, 8)          # pass kernel(s) + weight gradients, no Adam

    def policy():
        fz._zeroed = {"actor": True}
        fz.policy_backward()
    row["us_policy_backward"] = timed(policy, 8)
    out[name] = row
    print("%-17s B %4d  update %7.1f us   critic_backward %7.1f   policy_backward %7.1f   loss %.4f" % (name, B, us, row["us_critic_backward"], row["us_policy_backward"],
                                                                                                         row["critic_loss"]), flush=True)

# the pass kernels alone (block): pack + k_critic_block, pack + k_policy_block
import ctypes as C
torch.manual_seed(0)
ag = T.TD3Agent(26, 18, 1.0, data_parallel=False)
fz = FusedTD3(ag, seed=1, rows=False, team=False, block=True)
data = torch.randn(100000, 72, device="cuda"); data[:, 71] = 1.0
tot = torch.tensor(100000, dtype=torch.long, device="cuda")
real = {n: getattr(fz.lib, n) for n in ("plentd3_wgrad", "plentd3_colsum", "plentd3_pack")}


class Skip(object):
    def __call__(self, *a):
        return 0


for label, skip in (("pass_with_pack", ("plentd3_wgrad", "plentd3_colsum")), ("pass_only", ("plentd3_wgrad", "plentd3_colsum", "plentd3_pack"))):
    fz.critic_backward(data, B, total=tot); fz.policy_backward(); torch.cuda.synchronize()           # packs exist
    lib = fz.lib

    class Proxy(object):
        def __getattr__(self, n):
            return Skip() if n in skip else getattr(lib, n)
    fz.lib = Proxy()

    def critic():
        fz._zeroed = {"critic": True}
        fz.critic_backward(data, B, total=tot)

    def policy():
        fz._zeroed = {"actor": True}
        fz.policy_backward()
    uc, up = timed(critic, 16), timed(policy, 16)
    fz.lib = lib
    out[label] = {"us_critic": uc, "us_policy": up, "critic_tflops": 2 * CRITIC_MAC * B / uc / 1e6, "policy_tflops": 2 * POLICY_MAC * B / up / 1e6,
                  "critic_frac_of_peak": 2 * CRITIC_MAC * B / (uc * 1e-6) / PEAK, "policy_frac_of_peak": 2 * POLICY_MAC * B / (up * 1e-6) / PEAK}
    print("%-15s critic %6.1f us = %5.1f TFLOP/s (%.2f of peak)   policy %6.1f us = %5.1f TFLOP/s (%.2f)" % (label, uc, out[label]["critic_tflops"], out[label]["critic_frac_of_peak"],
                                                                                                            up, out[label]["policy_tflops"], out[label]["policy_frac_of_peak"]), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r05_td3_block_bench.json"), "w"), indent=1)
