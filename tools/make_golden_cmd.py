#!/usr/bin/env python3
"""Fixture from the reference's DATA files: plen_bullet/trajectories/<joint>_cmd.npy, a recorded 500-step, 18-channel float32
policy action sequence (agent space, [-1, 1]), stacked in the reference's joint order (plen_env.py:718-743) ->
tests/golden/policy_cmd_sequence.npz.  Inputs only (the reference holds no paired outputs); used as a realistic open-loop
action script for the oracle-vs-kernel parity tests.  Run in the build container only (needs /root/reference)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("PLEN_REFERENCE", "/root/reference")
sys.path.insert(0, ROOT)
JOINT_NAMES = ['rb_servo_r_hip', 'r_hip_r_thigh', 'r_thigh_r_knee', 'r_knee_r_shin', 'r_shin_r_ankle', 'r_ankle_r_foot',
               'lb_servo_l_hip', 'l_hip_l_thigh', 'l_thigh_l_knee', 'l_knee_l_shin', 'l_shin_l_ankle', 'l_ankle_l_foot',
               'torso_r_shoulder', 'r_shoulder_rs_servo', 're_servo_r_elbow', 'torso_l_shoulder', 'l_shoulder_ls_servo', 'le_servo_l_elbow']
acts = np.stack([np.load(os.path.join(REF, "plen_bullet/trajectories", j + "_cmd.npy")) for j in JOINT_NAMES], 1).astype(np.float32)
assert acts.shape == (500, 18) and np.abs(acts).max() <= 1.0
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "policy_cmd_sequence.npz"), actions=acts, joint_names=np.array(JOINT_NAMES))
print("policy_cmd_sequence.npz", acts.shape, acts.dtype, float(acts.min()), float(acts.max()))

# ---- which of the shipped actors produced the log?  (walk_eval.py:48-54 loads 3229999; verified, not assumed)
# The first recorded action must be actor(reset observation).  Needs torch + the oracle's reset observation (test infrastructure).
try:
    import torch
    from oracle.oracle import OracleEnv
    obs0 = OracleEnv().reset()
    print("actor identification: max |actor_k(reset obs) - a_0| over the 18 channels")
    for k in (3189999, 3199999, 3209999, 3219999, 3229999, 3239999, 3249999):
        sd = torch.load(os.path.join(REF, "plen_bullet/models/plen_walk_gazebo_%d_actor" % k), map_location="cpu", weights_only=True)
        sd = {n: v.numpy().astype(np.float64) for n, v in sd.items()}
        h = np.maximum(sd["fc1.weight"] @ obs0 + sd["fc1.bias"], 0); h = np.maximum(sd["fc2.weight"] @ h + sd["fc2.bias"], 0)
        a0 = np.tanh(sd["fc3.weight"] @ h + sd["fc3.bias"])
        print("   %d  %.4f" % (k, np.abs(a0 - acts[0]).max()))
except Exception as e:            # the fixture above does not depend on this
    print("actor identification skipped:", e)
