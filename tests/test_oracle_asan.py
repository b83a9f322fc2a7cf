"""Sanitizer run of the CPU oracle (sanitizers belong on the CPU restatement; GPU AddressSanitizer is not available
on this pool).  Builds oracle/libplen_oracle_asan.so with -fsanitize=address,undefined (oracle/Makefile) and drives every exported
entry point the tests use in a child process with the ASan runtime preloaded; any report fails the test."""
import os
import subprocess
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

DRIVER = r"""
import sys, numpy as np
sys.path.insert(0, %(root)r)
from oracle import oracle
# oracle.py binds "libplen_oracle_<dtype>.so": dtype "asan" is the sanitizer build made by the test below
rng = np.random.default_rng(0)
for joint_act, head in ((False, 0), (True, 0), (False, 1)):
    e = oracle.OracleEnv(joint_act=joint_act, dtype="asan", reward_head=head)
    e.reset()
    obs, rew, flags = e.rollout(rng.uniform(-1, 1, (120, 18)).astype(np.float32))
    assert np.isfinite(obs).all()
    e.set_params(1.1, 0.5); e.set_world(10, 1e-7); e.set_friction(0.6, 0.05, 0.0)
    s = e.get_state(); e.set_state(s); e.script_reset()
    e.step(rng.uniform(-1, 1, 18)); e.get_aux(); e.contacts(); e.forward_dynamics(); e.minv_times(np.ones(24)); e.link_frames(); e.foot_forces()
    e.set_targets(np.zeros(18)); e.substep()
    e.script_step(np.zeros(18), .15, .1, 0., 0., 0., 0., 1, 0, 0., 0., 0., 0.)
    e.gazebo_script(.1, .15, 0., 0., 0., 0., 0., 10)
    del e
# long history: the gait arrays are bounded (HIST) and must not overflow
e = oracle.OracleEnv(dtype="asan"); e.script_reset()
for t in range(1300):
    e.script_step(np.full(18, 0.01 * (t %% 7)), .16, .0, 0., 0., 0., 0., 0, 0, 0., 0., 0., 0.)
print("ASAN-DRIVER-OK")
"""


def test_oracle_under_asan_and_ubsan(tmp_path):
    so = os.path.join(ROOT, "oracle", "libplen_oracle_asan.so")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "-B", "libplen_oracle_asan.so"])
    libasan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    if not os.path.isabs(libasan) or not os.path.exists(libasan):
        pytest.skip("gcc has no libasan runtime on this machine")
    env = dict(os.environ, LD_PRELOAD=libasan, ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:abort_on_error=0", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1",
               PYTHONDONTWRITEBYTECODE="1")
    drv = tmp_path / "drv.py"
    drv.write_text(DRIVER % dict(root=ROOT))
    p = subprocess.run([sys.executable, str(drv)], env=env, capture_output=True, text=True, timeout=600)
    out = p.stdout + p.stderr
    assert "AddressSanitizer" not in out and "runtime error" not in out, out[-3000:]
    assert p.returncode == 0 and "ASAN-DRIVER-OK" in p.stdout, out[-3000:]
