"""Time of FusedTD3.critic_backward (layer by layer) and critic_backward_rows (one row-block kernel) alone on the GPU, as captured graphs."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    from plen_ml_walk_amd import td3 as T
    from plen_ml_walk_amd.td3_fused import FusedTD3
    torch.manual_seed(0)
    ag = T.TD3Agent(26, 18, 1.0, data_parallel=False)
    fz = FusedTD3(ag, seed=1)
    data = torch.randn(100000, 72, device="cuda")
    tot = torch.tensor(100000, dtype=torch.long, device="cuda")
    s = torch.cuda.Stream()
    for name, rows in (("layer by layer", False), ("row blocks", True)):
        fz.rows = rows
        fn = lambda: fz.critic_backward(data, B, total=tot)
        with torch.cuda.stream(s):
            fn(); fn()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                fn()
            for _ in range(5):
                g.replay()
            s.synchronize()
            t0 = time.perf_counter()
            for _ in range(200):
                g.replay()
            s.synchronize()
            print("%-15s B=%d: %.1f us per critic backward" % (name, B, (time.perf_counter() - t0) / 200 * 1e6))


if __name__ == "__main__":
    main()
