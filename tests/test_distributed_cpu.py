"""World-size-2 data-parallel TD3 on CPU (gloo): parameters start from rank 0's initialisation, the
flat-bucket gradient all-reduce makes two ranks with different local batches follow exactly the
single-process trajectory on the concatenated batch, and the rank bookkeeping of bench.py is right."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def test_two_rank_gradient_allreduce(tmp_path):
    out = str(tmp_path / "dist")
    env = dict(os.environ, OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "dist_worker.py"), out]
    subprocess.run(cmd, check=True, timeout=180, env=env, cwd=ROOT)
    r0 = json.load(open(out + ".rank0.json")); r1 = json.load(open(out + ".rank1.json"))
    assert r0["world"] == 2 and r0["same_across_ranks"] and r1["same_across_ranks"]
    assert r0["moved"] > 1e-5                                   # the optimiser really stepped
    assert r0["max_abs_diff_vs_single_process"] <= 1e-6         # mean of 2 half-batch gradients == full-batch gradient
    assert r0["slice"] == [0, 4098] and r1["slice"] == [4098, 8195]
    assert r0["max_time"] == 2.0 and r1["max_time"] == 2.0 and r0["sum_steps"] == 30.0
