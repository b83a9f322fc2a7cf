"""Open-loop walking gait generator + closed-form leg IK: same class, method and attribute names as the
reference's plen_bullet/src/plen_bullet/trajectory_generator.py (TrajectoryGenerator), units mm.

The reference builds a whole PlenWalkEnv(joint_act=True) only to read `real_ranges`
(trajectory_generator.py:52, :203-222); here the ranges come from plen_env.REAL_RANGES, so the generator
needs no GPU and no physics.  Feed `foot_walk_rfwd` / `foot_walk_lfwd` / `bend` to a joint_act
environment (PlenVecEnv(joint_act=True)) the way trajectory_eval.py:180-271 does."""
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
        """Cartesian foot trajectories relative to the hips, as 3 x n arrays (x forward, y sideways, z up from the bent stance);
        phases as in trajectory_generator.py:54-152: the stepping ("dominant") foot lifts and strides during single support and slides
        back during double support, the support foot mirrors it with half the stride."""
        nD, nS = self.num_DoubleSupport, self.num_SingleSupport
        ts = np.arange(nS) / (nS - 1.0)                  # phase in [0, 1] over single support
        td = np.arange(2 * nD) / (2 * nD - 1.0)          # ... over double support
        L, sway, bend = self.stride_length, self._body_sway, self._bend_distance
        third = 1 / 3.0
        ss_dom = np.stack([ts * L,
                           np.sin(-np.pi * (third * (1 + ts))) * sway,
                           np.sin(ts * np.pi) * self.foot_lift_height + bend])
        ds_dom = np.stack([-L * (td / 2.0),
                           np.sin(-np.pi * ((-1.0 / 3.0) + (2 / 3.0) * td)) * sway,
                           np.full(2 * nD, bend)])
        ss_sup = np.stack([L * ((1.0 / 2.0) - ts) / 2.0,
                           np.sin(np.pi * (third * (1 + ts))) * sway,
                           np.full(nS, bend)])
        ds_sup = np.stack([L * (1.0 - td) / 2.0,
                           np.sin(-np.pi * ((2.0 / 3.0) + (2 / 3.0) * td)) * sway,
                           np.full(2 * nD, bend)])
        if self.fwd_bias != 0:
            for path in (ds_sup, ss_sup, ds_dom, ss_dom):
                path[0] -= self.fwd_bias
        # one step = second half of double support, single support, first half of the next double support
        self.foot_walk_rfwd_r = np.column_stack([ds_dom[:, nD:], ss_dom, ds_sup[:, :nD]])
        self.foot_walk_lfwd_r = np.column_stack([ds_sup[:, nD:], ss_sup, ds_dom[:, :nD]])
        self.SS_dominant_foot, self.DS_dominant_foot = ss_dom, ds_dom
        self.SS_support_foot, self.DS_support_foot = ss_sup, ds_sup

    def assemble_trajectories(self):
        """The other foot does the mirrored (y negated) trajectory (trajectory_generator.py:154-167)."""
        mirror_y = np.array([[1], [-1], [1]])
        self.foot_walk_lfwd_l = self.foot_walk_rfwd_r * mirror_y
        self.foot_walk_rfwd_l = self.foot_walk_lfwd_r * mirror_y

    def IK(self, point, RightLeg):
        """Closed-form inverse kinematics of one leg relative to its hip (trajectory_generator.py:169-233), vectorised over the
        n foot positions in `point` (3 x n, mm).  Returns (n, 6): [0, hip roll, thigh pitch, knee, ankle pitch, ankle roll]."""
        pt = np.asarray(point, dtype=np.float64)
        a, b = self.l_hip_knee, self.l_knee_foot
        x, y = pt[0], pt[1]
        z = a + b - pt[2]                                             # height of the hip above the foot
        roll = np.arctan2(-y, z) if RightLeg else np.arctan2(y, z)
        knee = np.arccos((x ** 2 + y ** 2 + z ** 2 - a ** 2 - b ** 2) / (2.0 * a * b))
        ryz = np.sqrt(y ** 2 + z ** 2)
        ratio = a / b
        thigh = -np.arctan2(ryz * np.sin(knee) + x * np.cos(knee) + x * ratio,
                            ryz * np.cos(knee) + ryz * ratio - x * np.sin(knee))
        lim = self.env.real_ranges
        k_idx, t_idx = (3, 2) if RightLeg else (9, 8)                 # knee / thigh rows of the real joint ranges cap the solution
        knee = np.clip(knee, lim[k_idx][0], lim[k_idx][1])
        thigh = np.clip(thigh, lim[t_idx][0], lim[t_idx][1])
        ankle_pitch = -(thigh + knee)
        ankle_roll = -roll if RightLeg else roll
        return np.stack([np.zeros_like(roll), roll, thigh, knee, ankle_pitch, ankle_roll], axis=1)

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
