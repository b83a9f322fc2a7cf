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
        for i in range(L.MAXMEMB):
            m2.member_mass[b][i] *= 2.0           # (the member links' masses must sum to the body's: plenvec_create_from_model checks it)
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


@pytest.mark.gpu
def test_unused_box_slots_stay_out_of_contact_when_the_base_turns_over():
    """a PlenModel with num_boxes < 31 leaves unused box slots.  Round 3 parked them 1e6 m 'above' the base in the BASE frame -- a base
    tilted past 90 degrees (reachable with auto_reset off: compute_done is one-sided, plen_env.py:1082-1083) turned that into a phantom corner 1e6 m
    below the ground that took a foot's contact slot with a 1e6 m penetration.  Now the slot's breaking threshold is unreachable for any pose: a robot
    turned upside down in the air, with NO box collider at all, has no contact point (slot mask 0), falls freely for a substep and stays finite."""
    from plen_ml_walk_amd import _lib as L
    from plen_ml_walk_amd.vec_env import PlenVecEnv
    import numpy as np
    m = L.default_model()
    m.num_boxes = 0
    n = 8
    env = PlenVecEnv(n, device="cuda:0", dtype=torch.float64, model=m, auto_reset=False)
    env.reset()
    st = torch.zeros(n, 49, dtype=torch.float64, device="cuda")
    # base upside down (rotation by pi - 0.2 * e about x), 0.3 m above the ground: feet in the air, nothing near the ground
    for e in range(n):
        ang = np.pi - 0.2 * e / n
        st[e, 2] = 0.3; st[e, 3] = np.sin(ang / 2); st[e, 6] = np.cos(ang / 2)
    env.set_state(st)
    dump = env.debug_substeps(torch.zeros(n, 18, dtype=torch.float64, device="cuda"), 1)
    aux = env.get_aux().cpu().numpy()
    assert (aux[:, 7] == 0).all(), aux[:, 7]                      # no slot lent, no slot occupied
    s1 = env.get_state().cpu().numpy()
    assert np.isfinite(s1).all()
    assert np.allclose(s1[:, 12], -9.81 / 240.0, atol=1e-6), s1[:, 12]      # base v_z after one substep of free fall (the COM's, up to the joints' reaction)
    env.close()


@pytest.mark.gpu
def test_create_from_model_rejects_numbers_the_kernel_cannot_digest():
    from plen_ml_walk_amd import _lib as L
    import ctypes as C
    lib = L.load()
    cfg = L.default_cfg()

    def rc(mutate):
        m = L.default_model(); mutate(m)
        h = C.c_void_p()
        r = lib.plenvec_create_from_model(C.byref(m), C.byref(cfg), 4, 0, C.byref(h))
        if r == 0:
            lib.plenvec_destroy(h)
        return r
    assert rc(lambda m: None) == 0
    def nan_com(m): m.com[3][1] = float("nan")
    def skew_R(m): m.joint_R[2][1] += 0.01
    def neg_I(m): m.inertia[5][0] = -1e-6
    def bad_member(m): m.member_mass[0][0] *= 1.5
    def bad_rep(m): m.sole_rep[1][3] = 2
    def inf_box(m): m.box_t[0][2] = float("inf")
    for mut in (nan_com, skew_R, neg_I, bad_member, bad_rep, inf_box):
        assert rc(mut) != 0, mut.__name__


@pytest.mark.gpu
def test_facade_publishes_the_episode_reward_like_the_reference(capsys):
    """plen_env.py:574-579, 616-636: reset() prints the finished episode's cumulated reward and the 1000-episode moving average (NaN until 1000 episodes exist)."""
    import numpy as np
    from plen_ml_walk_amd.plen_env import PlenWalkEnv
    env = PlenWalkEnv()
    env.reset()
    first = capsys.readouterr().out
    assert "Episode #0" in first and "Reward: 0" in first and "MA Reward: nan" in first
    tot = 0.0
    for t in range(5):
        _, r, done, _ = env.step(np.zeros(18, dtype=np.float32)); tot += r
    env.reset()
    out = capsys.readouterr().out
    assert "Episode #1" in out and "Total Timesteps: 5" in out and ("Reward: %s" % tot) in out, (out, tot)
    assert env.moving_avg_buffer[1] == tot and env.episode_num == 2
    q = PlenWalkEnv(quiet=True); q.reset()
    assert capsys.readouterr().out == ""
    env.close(); q.close()
