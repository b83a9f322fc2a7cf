"""World-size-2 and world-size-8 data-parallel TD3 on CPU (gloo): parameters start from rank 0's initialisation, the
flat-bucket gradient all-reduce makes the ranks with different local batches follow exactly the
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


def test_eight_rank_gradient_allreduce_and_env_slices(tmp_path):
    """BASELINE.json configs[3]'s shape without the hardware: EIGHT ranks over gloo -- rank 0's parameters everywhere, eight different local
    batches, the flat-bucket all-reduce (mean) equal to one process on the concatenated batch, bitwise equal parameters on all eight ranks; 32768 envs split into
    eight contiguous slices of 4096, 4096 envs in total (strong scaling) into eight of 512 = one launch per rank, per-rank seeds distinct."""
    out = str(tmp_path / "dist8")
    env = dict(os.environ, OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "dist_worker.py"), out]
    subprocess.run(cmd, check=True, timeout=420, env=env, cwd=ROOT)
    rs = [json.load(open(out + ".rank%d.json" % r)) for r in range(8)]
    assert all(r["world"] == 8 and r["same_across_ranks"] for r in rs)
    # mean of 8 local gradients == the gradient of the 256-row batch, up to f32 summation order seen through two Adam steps (each moves a parameter by <= lr = 3e-4 whatever
    # the gradient's size, so an entry whose gradient is rounding noise can differ by a visible fraction of a step): <= 1e-5, i.e. 2 % of the 6e-4 the parameters moved
    assert rs[0]["moved"] > 1e-5 and rs[0]["max_abs_diff_vs_single_process"] <= 1e-5
    assert [r["slice_exact"] for r in rs] == [[4096 * k, 4096 * (k + 1)] for k in range(8)]
    assert [r["slice_strong"] for r in rs] == [[512 * k, 512 * (k + 1)] for k in range(8)]
    ragged = [r["slice"] for r in rs]                                                         # 32771 envs: contiguous, covering, the last rank takes the short slice
    assert ragged[0][0] == 0 and ragged[-1][1] == 32771 and all(ragged[k][1] == ragged[k + 1][0] for k in range(7)) and ragged[0][1] - ragged[0][0] == 4097
    assert sorted(r["seed"] for r in rs) == list(range(1000, 1008))
    assert all(r["max_time"] == 8.0 for r in rs) and rs[0]["sum_steps"] == 360.0
