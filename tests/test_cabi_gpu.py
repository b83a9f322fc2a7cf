"""The C ABI driven from C (tests/cabi_driver.c, no ctypes), plenvec_create_from_model and plenvec_step2."""
import ctypes as C
import os
import subprocess
import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "plen_ml_walk_amd", "csrc")


def _build_driver(out):
    from plen_ml_walk_amd.build import build_extension
    build_extension()
    # a C compiler, not hipcc: the header and the HIP runtime's C API are plain C99
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-x", "c", os.path.join(ROOT, "tests", "cabi_driver.c"), "-I", os.path.join(ROOT, "include"),
                           "-I/opt/rocm/include", "-D__HIP_PLATFORM_AMD__", "-L", CSRC, "-lplenvec", "-L/opt/rocm/lib", "-lamdhip64", "-lm",
                           "-Wl,-rpath," + CSRC, "-Wl,-rpath,/opt/rocm/lib", "-o", out])


def test_c_driver_compiles_against_the_header(tmp_path):
    """CPU: the header is valid C and the library links from a C program (nothing is run: no GPU here)."""
    _build_driver(str(tmp_path / "cabi_driver"))


@pytest.mark.gpu
def test_c_program_drives_the_library(tmp_path):
    exe = str(tmp_path / "cabi_driver")
    _build_driver(exe)
    p = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "cabi_driver ok" in p.stdout, (p.returncode, p.stdout, p.stderr)


@pytest.mark.gpu
def test_create_from_model_variant_equals_the_mass_scale_parameter():
    """A PlenModel with every mass and inertia scaled by 1.1 is the same robot as the default model under plenvec_set_params(mass_scale = 1.1);
    a model whose right arm is twice as heavy is a different one."""
    from plen_ml_walk_amd import _lib as L
    from plen_ml_walk_amd.vec_env import PlenVecEnv
    m = L.default_model()
    for b in range(L.NBODY):
        m.mass[b] *= 1.1
        for i in range(6):
            m.inertia[b][i] *= 1.1
        for i in range(L.MAXMEMB):
            m.member_mass[b][i] *= 1.1
    n = 16
    cfg = dict(rolling_friction=0.0)           # the well-conditioned configuration: rounding differences (m * 1.1 on the host vs in the kernel) stay at rounding level
    ea = PlenVecEnv(n, device="cuda:0", dtype=torch.float64, model=m, cfg_overrides=cfg)
    eb = PlenVecEnv(n, device="cuda:0", dtype=torch.float64, cfg_overrides=cfg)
    eb.set_params(mass_scale=torch.full((n,), 1.1, dtype=torch.float64))
    m2 = L.default_model()
    for b in (13, 14, 15):
        m2.mass[b] *= 2.0
    ec = PlenVecEnv(n, device="cuda:0", dtype=torch.float64, model=m2, cfg_overrides=cfg)
    oa, ob, oc = ea.reset().clone(), eb.reset().clone(), ec.reset().clone()
    assert float((oa - ob).abs().max()) < 1e-8, float((oa - ob).abs().max())
    assert float((oa - oc).abs().max()) > 1e-5, float((oa - oc).abs().max())
    g = torch.Generator(device="cuda").manual_seed(3)
    for t in range(6):
        a = torch.rand(n, 18, generator=g, device="cuda") * 0.4 - 0.2
        xa, ra, _, _ = ea.step(a); xb, rb, _, _ = eb.step(a)
        assert float((xa - xb).abs().max()) < 1e-6, t
    for e in (ea, eb, ec):
        e.close()


@pytest.mark.gpu
def test_step2_pair_matches_the_done_bits():
    from plen_ml_walk_amd import _lib as L
    from plen_ml_walk_amd.vec_env import PlenVecEnv
    n = 256
    e1 = PlenVecEnv(n, device="cuda:0", cfg_overrides={"max_episode_steps": 40})
    e2 = PlenVecEnv(n, device="cuda:0", cfg_overrides={"max_episode_steps": 40})
    e1.reset(); e2.reset()
    g = torch.Generator(device="cuda").manual_seed(0)
    seen_t = seen_d = 0
    for t in range(90):         # random actions: most episodes end by a fall within ~25 steps, the survivors at the 40-step limit
        a = torch.rand(n, 18, generator=g, device="cuda") * 2 - 1
        _, _, f, _ = e1.step(a)
        _, _, d, tr = e2.step2(a)
        f = f.clone()
        assert torch.equal(d, (((f & 1) != 0) & ((f & 2) == 0)).to(torch.uint8)) and torch.equal(tr, ((f & 2) != 0).to(torch.uint8))
        seen_t += int(tr.sum()); seen_d += int(d.sum())
    assert seen_t > 0 and seen_d > 0
    e1.close(); e2.close()
