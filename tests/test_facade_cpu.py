"""Host-side mirror of the reference interface that needs no GPU: gym shim, spaces, registry, ranges."""
import os
import numpy as np
import pytest
import torch
from plen_ml_walk_amd import gym_compat


def test_box_and_timelimit():
    b = gym_compat.Box(np.ones(18) * -1, np.ones(18), dtype=np.float32)
    b.seed(0)
    s = b.sample()
    assert s.dtype == np.float32 and s.shape == (18,) and b.contains(s)

    class Dummy(gym_compat.Env):
        def reset(self): return 0
        def step(self, a): return 0, 1.0, False, {}
    e = gym_compat.TimeLimit(Dummy(), 3)
    e.reset()
    flags = [e.step(0)[2] for _ in range(3)]
    assert flags == [False, False, True] and e._max_episode_steps == 3


def test_registry_and_constants(golden_dir):
    from plen_ml_walk_amd import plen_env as pe
    spec = gym_compat._REGISTRY["PlenWalkEnv-v1"]
    assert spec["max_episode_steps"] == 500                                         # plen_env.py:15-19
    g = np.load(os.path.join(golden_dir, "a1_agent_to_env.npz"))
    assert np.array_equal(np.array(pe.ENV_RANGES), g["env_ranges"]) and np.array_equal(np.array(pe.REAL_RANGES), g["real_ranges"])
    assert pe.MOVING_JOINTS == list(g["moving_joints"]) and len(pe.JOINT_NAMES) == 18


@pytest.mark.skipif(torch.cuda.is_available(), reason="CPU-only behaviour")
def test_env_construction_fails_loudly_without_gpu():
    from plen_ml_walk_amd import plen_env as pe, _lib
    with pytest.raises(_lib.PlenvecError):
        pe.PlenWalkEnv()


def test_compat_shims_resolve():
    import importlib, sys
    compat = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "plen_ml_walk_amd", "compat")
    sys.path.insert(0, compat)
    try:
        td3 = importlib.import_module("plen_ros_helpers.td3")
        assert {"ReplayBuffer", "TD3Agent", "evaluate_policy"} <= set(dir(td3))
        pe = importlib.import_module("plen_bullet.plen_env")
        assert hasattr(pe, "PlenWalkEnv")
    finally:
        sys.path.remove(compat)
        for m in ("plen_ros_helpers.td3", "plen_ros_helpers", "plen_bullet.plen_env", "plen_bullet"):
            sys.modules.pop(m, None)
