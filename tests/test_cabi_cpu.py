"""The C-ABI shared object: loads on a CPU-only box, exports exactly what include/plenvec.h declares,
refuses to run without a GPU (no CPU fallback), and its config struct matches the ctypes mirror."""
import ctypes as C
import os
import re
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from plen_ml_walk_amd.build import build_extension
    from plen_ml_walk_amd import _lib
    build_extension()
    return _lib.load()


def test_header_symbols_are_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "plenvec.h")).read()
    names = sorted(set(re.findall(r"\b(plenvec_[a-z0-9_]+)\s*\(", hdr)))
    assert len(names) >= 16
    from plen_ml_walk_amd import _lib
    assert sorted(_lib.EXPORTS) == names
    for n in names:
        assert hasattr(lib, n), n


def test_td3_kernel_library_exports_its_header():
    """libplentd3.so (fused TD3 update kernels): loads without a GPU and exports exactly what include/plentd3.h declares."""
    from plen_ml_walk_amd.build import build_td3_kernels
    from plen_ml_walk_amd import td3_fused
    build_td3_kernels()
    hdr = open(os.path.join(ROOT, "include", "plentd3.h")).read()
    names = sorted(set(re.findall(r"\b(plentd3_[a-z0-9_]+)\s*\(", hdr)))
    assert sorted(td3_fused.EXPORTS) == names and len(names) == 34
    lib = td3_fused.load()
    for n in names:
        assert hasattr(lib, n), n
    if not torch.cuda.is_available():
        from plen_ml_walk_amd.td3 import TD3Agent
        with pytest.raises(td3_fused.PlenTd3Error):
            td3_fused.FusedTD3(TD3Agent(26, 18, 1.0, device="cpu"))


def test_default_cfg_matches_reference_constants(lib):
    from plen_ml_walk_amd import _lib
    c = _lib.default_cfg(False)
    assert (c.max_episode_steps, c.substeps, c.reset_substeps, c.num_iterations) == (500, 4, 8, 50)       # plen_env.py:15-19,40-42,569
    assert c.dt == 1.0 / 240.0 and c.gravity_z == -9.81 and c.motor_max_force == 0.15 and c.spawn_z == 0.158   # :41,:296,:753,:312
    assert abs(c.lateral_friction - 0.64) < 1e-15 and abs(c.rolling_friction - 0.08) < 1e-15 and c.linear_damping == 0.0
    j = _lib.default_cfg(True)
    assert abs(j.rolling_friction - 0.008) < 1e-15 and j.linear_damping == 0.1 and j.joint_act == 1      # :439-442,:472-475
    assert C.sizeof(_lib.PlenCfg) == 8 * 4 + 18 * 8 + 2 * 4 and c.nonfinite_guard == 1 and c.body_contacts == 1 and abs(c.box_lateral_friction - 0.4) < 1e-15


def test_error_paths_without_gpu(lib):
    from plen_ml_walk_amd import _lib
    h = C.c_void_p()
    cfg = _lib.default_cfg(False)
    assert lib.plenvec_create(C.byref(cfg), 0, 0, C.byref(h)) == -1            # PLENVEC_E_INVAL
    assert b"num_envs" in lib.plenvec_last_error()
    assert lib.plenvec_step(None, None, None, None, None, None, None) == -1
    if not torch.cuda.is_available():
        rc = lib.plenvec_create(C.byref(cfg), 4, 0, C.byref(h))
        assert rc == -3 and b"no CPU fallback" in lib.plenvec_last_error()      # PLENVEC_E_NODEV: fails loudly
        from plen_ml_walk_amd.vec_env import PlenVecEnv
        with pytest.raises(_lib.PlenvecError):
            PlenVecEnv(4)


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "plen_ml_walk_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(dp, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt and "plen_oracle" not in txt, f


def test_generated_motor_pass_header_is_current(tmp_path):
    """plen_motor_pass_gen.h (the solver's 18 motor rows as assembly text) is what tools/gen_motor_pass.py writes for the NC_ORDER_LIST of plenvec.hip."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("gen_motor_pass", os.path.join(ROOT, "tools", "gen_motor_pass.py"))
    g = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(g)
    committed = open(g.OUT).read()
    g.OUT = str(tmp_path / "gen.h")
    g.main()
    assert open(g.OUT).read() == committed
    order = g.nc_order()
    assert sorted(order) == list(range(18)) and "order: " + ", ".join(map(str, order)) in committed


def test_env_kernels_stay_inside_their_register_budgets(lib):
    """The compiler's resource report of the shipped library (build.py keeps it next to the .so): the env kernels run at the occupancy their design states with no scratch,
    and the scalar registers spilled into VGPR lanes stay where round 6 brought them (f64 705 -> < 100, f32 433 -> < 150; round 5's were the 18 per-row condition masks of the
    joint-limit rows, kept across all 50 limit-flavour copies of the solver loop).  A regression guard: a change that makes the compiler hold loop-invariant lane masks again
    shows up here before it shows up as v_readlane traffic in the solver loops."""
    import json
    from plen_ml_walk_amd.build import RESOURCES
    r = json.load(open(RESOURCES))
    k = {(("f64" if "IdLb" in n else "f32"), ("fast" if "Lb1EE" in n else "compiler_rows")): v for n, v in r.items() if "plen_env_kernel" in n}
    assert len(k) == 4, sorted(r)
    for (dt, path), v in k.items():
        assert v["ScratchSize"] == 0 and v["VGPRs Spill"] == 0, (dt, path, v)
        assert v["Occupancy"] == (2 if dt == "f64" else 4), (dt, path, v)
        assert v["LDS Size"] == (19872 if dt == "f64" else 9936), (dt, path, v)
    assert k[("f64", "fast")]["SGPRs Spill"] < 100, k[("f64", "fast")]
    assert k[("f32", "fast")]["SGPRs Spill"] < 250          # (193: prologue and limit flavours; no reload inside a hot iteration loop), k[("f32", "fast")]
