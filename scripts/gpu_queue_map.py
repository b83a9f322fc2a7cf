"""Which torch streams share a hardware queue?  Two streams on one queue serialise: time a spin kernel on each pair.

usage: python scripts/gpu_queue_map.py [n_normal] [n_high]      (GPU_MAX_HW_QUEUES from the environment, default 16 as the package sets)
Prints, for every stream, the list of streams it serialises with.  Findings are recorded in DESIGN.md section 10."""
import os
import sys
import time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import torch


def main():
    nn = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    nh = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    order = sys.argv[3] if len(sys.argv) > 3 else "normal-first"
    dev = torch.device("cuda", 0)
    torch.zeros(1, device=dev)
    if order == "high-first":
        hi = [("h%d" % i, torch.cuda.Stream(priority=-1)) for i in range(nh)]
        no = [("n%d" % i, torch.cuda.Stream()) for i in range(nn)]
    else:
        no = [("n%d" % i, torch.cuda.Stream()) for i in range(nn)]
        hi = [("h%d" % i, torch.cuda.Stream(priority=-1)) for i in range(nh)]
    streams = [("default", torch.cuda.current_stream())] + no + hi
    for _, s in streams:                       # first use in this order
        with torch.cuda.stream(s):
            torch.cuda._sleep(1000)
    torch.cuda.synchronize()
    cyc = 2_000_000                            # ~1 ms

    def t_pair(a, b):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.cuda.stream(a):
            torch.cuda._sleep(cyc)
        if b is not None:
            with torch.cuda.stream(b):
                torch.cuda._sleep(cyc)
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    base = min(t_pair(streams[1][1], None) for _ in range(3))
    print("GPU_MAX_HW_QUEUES=%s order=%s single %.3f ms" % (os.environ["GPU_MAX_HW_QUEUES"], order, base * 1e3))
    ids = [(n, s.stream_id, s.cuda_stream) for n, s in streams]
    print("streams:", [(n, i) for n, i, _ in ids])
    for i, (na, a) in enumerate(streams):
        coll = []
        for j, (nb, b) in enumerate(streams):
            if i == j:
                continue
            t = min(t_pair(a, b) for _ in range(2))
            if t > 1.6 * base:
                coll.append(nb)
        print("%-8s serialises with %s" % (na, coll))


if __name__ == "__main__":
    main()
