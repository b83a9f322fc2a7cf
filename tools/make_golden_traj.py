#!/usr/bin/env python3
"""Fixture from the reference's DATA files plen_bullet/trajectories/<joint>_traj.npy (18 x 800 float64, the joint-space walking
trajectory its trajectory_eval.py:180-271 assembles and saves) and bend_traj.npy -> tests/golden/traj_eval.npz.
Run in the build container only (needs /root/reference)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("PLEN_REFERENCE", "/root/reference")
JOINT_NAMES = ['rb_servo_r_hip', 'r_hip_r_thigh', 'r_thigh_r_knee', 'r_knee_r_shin', 'r_shin_r_ankle', 'r_ankle_r_foot',
               'lb_servo_l_hip', 'l_hip_l_thigh', 'l_thigh_l_knee', 'l_knee_l_shin', 'l_shin_l_ankle', 'l_ankle_l_foot',
               'torso_r_shoulder', 'r_shoulder_rs_servo', 're_servo_r_elbow', 'torso_l_shoulder', 'l_shoulder_ls_servo', 'le_servo_l_elbow']
d = os.path.join(REF, "plen_bullet/trajectories")
traj = np.stack([np.load(os.path.join(d, j + "_traj.npy")) for j in JOINT_NAMES], 1)
bend = np.load(os.path.join(d, "bend_traj.npy"))
assert traj.shape == (800, 18) and bend.shape == (18,)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "traj_eval.npz"), walk=traj, bend=bend)
print("traj_eval.npz", traj.shape, bend.shape)
