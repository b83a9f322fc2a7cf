"""the gait generator + IK against known answers captured from the reference
(tests/golden/traj_gen.npz, made by tools/make_golden.py from trajectory_generator.py)."""
import os
import numpy as np
import pytest
from plen_ml_walk_amd.trajectory_generator import TrajectoryGenerator


@pytest.mark.parametrize("name,kw", [("default", {}), ("eval", dict(num_DoubleSupport=20, num_SingleSupport=20, height=20.0, stride=20.0))])
def test_foot_paths_and_joint_trajectories(golden_dir, name, kw):
    g = np.load(os.path.join(golden_dir, "traj_gen.npz"))
    gen = TrajectoryGenerator(**kw)
    gen.main()
    for attr in ("foot_walk_rfwd_r", "foot_walk_lfwd_r", "foot_walk_rfwd", "foot_walk_lfwd", "bend"):
        ref = g["%s.%s" % (name, attr)]
        got = getattr(gen, attr)
        assert got.shape == ref.shape
        assert np.abs(got - ref).max() <= 1e-12, attr
    p = g[name + ".params"]
    assert (gen.num_DoubleSupport, gen.num_SingleSupport) == (int(p[0]), int(p[1]))


def test_ik_known_answers(golden_dir):
    g = np.load(os.path.join(golden_dir, "traj_gen.npz"))
    gen = TrajectoryGenerator()
    assert np.abs(gen.IK(g["ik_points"], True) - g["ik_right"]).max() <= 1e-12
    assert np.abs(gen.IK(g["ik_points"], False) - g["ik_left"]).max() <= 1e-12
    acts = gen.walk_cycle_actions(cycles=2)
    assert acts.shape[1] == 18 and np.all(acts[:, 12:] == 0) and np.isfinite(acts).all()


def test_trajectory_eval_assembly_matches_the_committed_trajectories(golden_dir):
    """The reference commits what its trajectory_eval.py:180-271 assembles and saves (trajectories/<joint>_traj.npy, bend_traj.npy ->
    tests/golden/traj_eval.npz): generator + assembly here reproduce those 18 x 800 joint targets and the bend pose."""
    from plen_ml_walk_amd.trajectory_eval import assemble_joint_trajectories
    g = np.load(os.path.join(golden_dir, "traj_eval.npz"))
    walk, bend = assemble_joint_trajectories()
    assert walk.shape == g["walk"].shape == (800, 18)
    assert np.abs(walk - g["walk"]).max() <= 1e-12
    assert np.abs(bend - g["bend"]).max() <= 1e-12
