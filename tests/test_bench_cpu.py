"""bench.py on a box without a GPU: it must import, state its contract, price the algorithmic bytes as SURVEY 8(d) does, and fail loudly (not fall
back to anything) when asked to run."""
import importlib.util
import json
import os
import subprocess
import sys
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
    return m


def test_algorithmic_bytes_are_the_surveys():
    b = _bench()
    assert b.algo_bytes_per_env_step(4) == 776 and b.algo_bytes_per_env_step(8) == 1476          # 368 + 408 B at 4-byte reals
    assert b.ENVS_PER_GPU == 4096 and b.HBM_PEAK_GBS == 8000.0 and b.MIN_TIMED_S >= 0.25
    pmc, src = b._pmc_summary("f64")
    assert pmc and src.startswith("profiles/r") and pmc["valu_insts_per_env_step"] > 5e4 and pmc["hbm_traffic_bytes"] > 6e6


def test_help_and_flags():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0
    for flag in ("--gpus", "--steps", "--warmup", "--scaling", "--no-parity", "--dtype", "--legs"):
        assert flag in out.stdout, flag


def test_no_gpu_means_no_number():
    if torch.cuda.is_available():
        return
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--legs", ""],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and not any(l.startswith("{") for l in out.stdout.splitlines())       # no JSON line from a machine without the MI355X
