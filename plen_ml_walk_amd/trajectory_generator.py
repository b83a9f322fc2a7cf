"""Open-loop walking gait generator + closed-form leg IK: same class, method and attribute names as the
reference's plen_bullet/src/plen_bullet/trajectory_generator.py (TrajectoryGenerator), units mm.

The reference builds a whole PlenWalkEnv(joint_act=True) only to read `real_ranges`
(trajectory_generator.py:52, :203-222); here the ranges come from plen_env.REAL_RANGES, so the generator
needs no GPU and no physics.  Feed `foot_walk_rfwd` / `foot_walk_lfwd` / `bend` to a joint_act
environment (PlenVecEnv(joint_act=True)) the way trajectory_eval.py:180-271 does."""
import math

import numpy as np

from .plen_env import REAL_RANGES


class _Ranges(object):
    real_ranges = [list(r) for r in REAL_RANGES]


class TrajectoryGenerator():
    def __init__(self, num_DoubleSupport=5, num_SingleSupport=10, height=30.0, stride=30.0, bend_distance=10.0,
                 body_sway=5.0, fwd_bias=10.0, sway_steps=5):
        self.l_hip_knee = 25.0            # trajectory_generator.py:34-37
        self.l_knee_foot = 40.0
        self.num_DoubleSupport = num_DoubleSupport
        self.num_SingleSupport = num_SingleSupport
        self.foot_lift_height = height
        self.stride_length = stride
        self._bend_distance = bend_distance
        self._body_sway = body_sway
        self.fwd_bias = fwd_bias
        self.env = _Ranges()

    def foot_path(self):
        """Cartesian foot trajectories relative to the hips (trajectory_generator.py:54-152)."""
        nD, nS = self.num_DoubleSupport, self.num_SingleSupport
        DS_support_foot = np.zeros((3, 2 * nD)); SS_support_foot = np.zeros((3, nS))
        DS_dominant_foot = np.zeros((3, 2 * nD)); SS_dominant_foot = np.zeros((3, nS))
        for i in range(nS):                                   # dominant foot, single support: lift + stride
            t = i / (nS - 1.0)
            SS_dominant_foot[0][i] = t * self.stride_length
            SS_dominant_foot[1][i] = np.sin(-np.pi * ((1 / 3.0) * (1 + t))) * self._body_sway
            SS_dominant_foot[2][i] = np.sin(t * np.pi) * self.foot_lift_height + self._bend_distance
        for i in range(2 * nD):                               # dominant foot, double support
            t = i / (2 * nD - 1.0)
            DS_dominant_foot[0][i] = -self.stride_length * (t / 2.0)
            DS_dominant_foot[1][i] = np.sin(-np.pi * ((-1.0 / 3.0) + (2 / 3.0) * t)) * self._body_sway
            DS_dominant_foot[2][i] = self._bend_distance
        for i in range(nS):                                   # support foot, single support
            t = i / (nS - 1.0)
            SS_support_foot[0][i] = self.stride_length * ((1.0 / 2.0) - t) / 2.0
            SS_support_foot[1][i] = np.sin(np.pi * ((1 / 3.0) * (1 + t))) * self._body_sway
            SS_support_foot[2][i] = self._bend_distance
        for i in range(2 * nD):                               # support foot, double support
            t = i / (2.0 * nD - 1.0)
            DS_support_foot[0][i] = self.stride_length * (1.0 - t) / 2.0
            DS_support_foot[1][i] = np.sin(-np.pi * ((2.0 / 3.0) + (2 / 3.0) * t)) * self._body_sway
            DS_support_foot[2][i] = self._bend_distance
        if self.fwd_bias != 0:
            DS_support_foot[0] = DS_support_foot[0] - self.fwd_bias
            SS_support_foot[0] = SS_support_foot[0] - self.fwd_bias
            DS_dominant_foot[0] = DS_dominant_foot[0] - self.fwd_bias
            SS_dominant_foot[0] = SS_dominant_foot[0] - self.fwd_bias
        self.foot_walk_rfwd_r = np.column_stack([DS_dominant_foot[:, nD:], SS_dominant_foot, DS_support_foot[:, :nD]])
        self.foot_walk_lfwd_r = np.column_stack([DS_support_foot[:, nD:], SS_support_foot, DS_dominant_foot[:, :nD]])
        self.SS_dominant_foot, self.DS_dominant_foot = SS_dominant_foot, DS_dominant_foot
        self.SS_support_foot, self.DS_support_foot = SS_support_foot, DS_support_foot

    def assemble_trajectories(self):
        """The other foot does the mirrored (y negated) trajectory (trajectory_generator.py:154-167)."""
        flip = np.array([[1], [-1], [1]])
        self.foot_walk_lfwd_l = self.foot_walk_rfwd_r * flip
        self.foot_walk_rfwd_l = self.foot_walk_lfwd_r * flip

    def IK(self, point, RightLeg):
        """5-DoF leg inverse kinematics relative to the hip (trajectory_generator.py:169-233).
        point: (3, n) array of foot positions; returns (n, 6) joint angles [0, th1..th5]."""
        point = np.asarray(point, dtype=np.float64)
        n = point[0].size
        joint_angles = np.zeros((n, 6))
        rr = self.env.real_ranges
        for i in range(n):
            lhip_knee, lknee_foot = self.l_hip_knee, self.l_knee_foot
            Zx = point[0][i]; Zy = point[1][i]
            Zz = self.l_hip_knee + self.l_knee_foot - point[2][i]
            th1 = math.atan2(-Zy, Zz) if RightLeg else math.atan2(Zy, Zz)
            th3 = math.acos((Zx ** 2 + Zy ** 2 + Zz ** 2 - lhip_knee ** 2 - lknee_foot ** 2) / (2.0 * lhip_knee * lknee_foot))
            sqrtyz = np.sqrt(Zy ** 2 + Zz ** 2)
            hok = lhip_knee / lknee_foot
            th2 = -math.atan2((sqrtyz * np.sin(th3) + Zx * np.cos(th3) + Zx * hok),
                              (sqrtyz * np.cos(th3) + sqrtyz * hok - Zx * np.sin(th3)))
            knee, thigh = (3, 2) if RightLeg else (9, 8)      # caps from the real joint ranges
            th3 = min(max(th3, rr[knee][0]), rr[knee][1])
            th2 = min(max(th2, rr[thigh][0]), rr[thigh][1])
            th4 = -(th2 + th3)
            th5 = -th1 if RightLeg else th1
            joint_angles[i] = np.array([0, th1, th2, th3, th4, th5])
        return joint_angles

    def joint_space_trajectories(self):
        """EE-space trajectories through the IK (trajectory_generator.py:235-270)."""
        bend_array = np.column_stack([np.array([0.0, 0.0, self._bend_distance])] * 3)
        self.bend = np.column_stack([self.IK(bend_array, True), self.IK(bend_array, False)])
        self.foot_walk_rfwd = np.column_stack([self.IK(self.foot_walk_rfwd_r, True), self.IK(self.foot_walk_rfwd_l, False)])
        self.foot_walk_lfwd = np.column_stack([self.IK(self.foot_walk_lfwd_r, True), self.IK(self.foot_walk_lfwd_l, False)])

    def main(self):
        self.foot_path()
        self.assemble_trajectories()
        self.joint_space_trajectories()

    def walk_cycle_actions(self, cycles=1):
        """[T, 18] joint targets for a joint_act environment: bend, then alternate right/left-forward steps;
        arms held at zero (the layout trajectory_eval.py:180-271 assembles)."""
        self.main()
        legs = [self.bend] + [self.foot_walk_rfwd, self.foot_walk_lfwd] * cycles
        legs = np.concatenate(legs, axis=0)
        out = np.zeros((legs.shape[0], 18))
        out[:, :12] = legs
        return out
