"""Which of the TD3 update's GEMM shapes is slow in the library?  us per call for the forward (x W^T + b), weight-gradient (dH^T x) and input-gradient
(dH W) products at batch 4096, as torch issues them (rocBLAS / hipBLASLt)."""
import torch, time
dev = "cuda"
B = 4096
def bench(name, fn, n=100):
    """GPU time per call: n calls captured in one hipGraph (no host launch gaps), replayed 5 times."""
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(5): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): g.replay()
    torch.cuda.synchronize(); print("%-58s %7.1f us" % (name, (time.perf_counter() - t0) / (5 * n) * 1e6))
for K, N in ((256, 256), (44, 512), (26, 256), (256, 18), (256, 1)):
    x = torch.randn(B, K, device=dev); W = torch.randn(N, K, device=dev); b = torch.randn(N, device=dev); dH = torch.randn(B, N, device=dev)
    Wt = W.t().contiguous(); out = torch.empty(B, N, device=dev); gW = torch.empty(N, K, device=dev)
    bench("fwd  addmm(b, x[%d,%d], W[%d,%d].t())" % (B, K, N, K), lambda: torch.addmm(b, x, W.t()))
    bench("fwd  _addmm_activation (bias+relu epilogue)", lambda: torch._addmm_activation(b, x, W.t()))
    bench("fwd  addmm(b, x, Wt contiguous [K,N])", lambda: torch.addmm(b, x, Wt))
    bench("fwd  mm(x, W.t()) no bias", lambda: torch.mm(x, W.t()))
    bench("fwd  F.linear(x, W, b)", lambda: torch.nn.functional.linear(x, W, b))
    bench("wgrad mm(dH.t(), x) -> [N,K]", lambda: torch.mm(dH.t(), x, out=gW))
    bench("igrad mm(dH, W) -> [B,K]", lambda: torch.mm(dH, W))
    print()
