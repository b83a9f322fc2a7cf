"""GPU tests of the C-ABI semantics added in round 2: the per-env non-finite guard, parameter changes vs the reset cache, masked
resets, and the TD3 update on the device (eager and hipGraph-captured) against the reference's golden vectors."""
import os
import numpy as np
import pytest
import torch

from oracle.oracle import OracleEnv

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env(n, dtype=torch.float32, **kw):
    from plen_ml_walk_amd.vec_env import PlenVecEnv
    return PlenVecEnv(n, dtype=dtype, **kw)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("auto_reset", [True, False])
def test_nonfinite_guard_resets_and_counts(dtype, auto_reset):
    """A NaN / inf in an env's state must not reach the caller: that env is reset (even with auto_reset off), reported as a truncation
    to the reset observation with reward 0 and PLENVEC_DONE_NONFINITE set, and counted; its neighbours are untouched."""
    from plen_ml_walk_amd import _lib as L
    n = 8
    env = _env(n, dtype, auto_reset=auto_reset)
    ref = _env(n, dtype, auto_reset=auto_reset)
    obs0 = env.reset().clone(); ref.reset()
    s = env.get_state()
    s[2, 15] = float("nan")            # a joint angle
    s[5, 0] = float("inf")             # base position (an infinite VELOCITY would be clamped to +-100 by the engine and stay finite)
    env.set_state(s)
    s_ok = ref.get_state(); ref.set_state(s_ok)           # same bookkeeping reset as env.set_state, finite everywhere
    act = torch.full((n, 18), 0.1, device="cuda")
    o, r, d, info = env.step(act)
    o2, r2, d2, info2 = ref.step(act)
    bad = torch.tensor([2, 5], device="cuda")
    good = torch.tensor([0, 1, 3, 4, 6, 7], device="cuda")
    assert torch.isfinite(o).all() and torch.isfinite(r).all() and torch.isfinite(info["obs"]).all()
    assert (d[bad] == (L.DONE_NONFINITE | L.DONE_TIMELIMIT)).all() and (r[bad] == 0).all()
    assert torch.equal(o[bad], obs0[bad]) and torch.equal(info["obs"][bad], obs0[bad])
    assert info["nonfinite"][bad].all() and not info["nonfinite"][good].any() and not info["terminal"][bad].any()
    assert torch.equal(o[good], o2[good]) and torch.equal(r[good], r2[good]) and torch.equal(d[good], d2[good])
    assert env.nonfinite_count() == 2 and ref.nonfinite_count() == 0
    st = env.get_state()
    assert torch.isfinite(st).all()
    fresh = _env(n, dtype); fresh.reset()
    assert torch.equal(st[bad], fresh.get_state()[bad])    # back in the reset state
    # the next step is an ordinary first step of an episode
    o3, r3, d3, _ = env.step(act)
    f3, fr3, fd3, _ = fresh.step(act)
    assert torch.equal(o3[bad], f3[bad]) and torch.equal(r3[bad], fr3[bad]) and env.nonfinite_count() == 2
    for e in (env, ref, fresh):
        e.close()


def test_nonfinite_guard_can_be_switched_off():
    """cfg.nonfinite_guard = 0 keeps the reference's behaviour: NaNs propagate (plen_env.py has no guard)."""
    env = _env(4, torch.float32, cfg_overrides=dict(nonfinite_guard=0))
    env.reset()
    s = env.get_state(); s[1, 20] = float("nan"); env.set_state(s)
    o, r, d, _ = env.step(torch.zeros(4, 18, device="cuda"))
    assert not torch.isfinite(o[1]).all() and torch.isfinite(o[[0, 2, 3]]).all() and env.nonfinite_count() == 0
    env.close()


def test_set_params_reset_cache_and_masked_reset():
    """plenvec_set_params: the dynamics use the new parameters at once; auto-resets after it restore a stance settled with the NEW
    parameters (the reset records are re-simulated at the next step without touching live envs); a masked reset resets only its mask."""
    n = 16
    g = torch.Generator(device="cuda").manual_seed(0)
    ms = 0.8 + 0.4 * torch.rand(n, generator=g, device="cuda")
    mu = 0.4 + 0.6 * torch.rand(n, generator=g, device="cuda")
    env = _env(n, torch.float64)
    env.reset()
    acts = (torch.rand(3, n, 18, generator=g, device="cuda") * 2 - 1) * 0.3
    for t in range(3):
        env.step(acts[t])
    live = env.get_state().clone()
    env.set_params(mass_scale=ms.double(), lateral_friction=mu.double())
    # (1) masked reset right after set_params: only the masked envs move, and they get the NEW settled stance
    mask = torch.zeros(n, dtype=torch.uint8, device="cuda"); mask[[1, 4, 9]] = 1
    env.reset(mask)
    st = env.get_state()
    keep = mask == 0
    assert torch.equal(st[keep], live[keep])
    want = []
    for i in (1, 4, 9):
        o = OracleEnv(); o.set_params(float(ms[i]), float(mu[i])); o.reset(); want.append(o.get_state())
    assert np.abs(st[[1, 4, 9]].cpu().numpy() - np.array(want)).max() <= 1e-9
    # (2) set_params followed directly by step(): envs that end their episode inside that step are restored from records built with the new parameters
    env2 = _env(n, torch.float64); env2.reset()
    s = env2.get_state(); s[:, 2] = 0.3                               # in the air, rolled by 1.2 rad > pi/3: every env terminates at once
    s[:, 3] = float(np.sin(0.6)); s[:, 4] = 0; s[:, 5] = 0; s[:, 6] = float(np.cos(0.6)); env2.set_state(s)
    env2.set_params(mass_scale=ms.double(), lateral_friction=mu.double())
    o, r, d, info = env2.step(torch.zeros(n, 18, device="cuda"))
    assert (d & 1).bool().all()
    st2 = env2.get_state().cpu().numpy()
    for i in (0, 7, 15):
        o_ = OracleEnv(); o_.set_params(float(ms[i]), float(mu[i])); ob = o_.reset()
        assert np.abs(st2[i] - o_.get_state()).max() <= 1e-9 and np.abs(info["obs"][i].cpu().numpy() - ob).max() <= 1e-9
    env.close(); env2.close()


# ------------------------------------------------------------------------------------------------ TD3 on the device
def _golden_agent(golden_dir, device):
    from plen_ml_walk_amd import td3 as T
    g = np.load(os.path.join(golden_dir, "td3_train.npz"))
    a = T.TD3Agent(26, 18, 1.0, device=device, data_parallel=False)
    a.load_arrays({k[len("init."):]: g[k] for k in g.files if k.startswith("init.")})
    a.actor_target.load_state_dict(a.actor.state_dict()); a.critic_target.load_state_dict(a.critic.state_dict())
    buf = T.ReplayBuffer(1000, device=device)
    buf.add_batch(torch.as_tensor(g["S"]), torch.as_tensor(g["A"]), torch.as_tensor(g["S2"]), torch.as_tensor(g["R"]), torch.as_tensor(g["D"]))
    return g, a, buf


def _check_against_golden(g, a, k, tol=2e-5):
    for net, prefix in ((a.actor, "actor."), (a.critic, "critic."), (a.actor_target, "actor_target."), (a.critic_target, "critic_target.")):
        for name, v in net.state_dict().items():
            ref_sum = g["it%d.sum.%s%s" % (k + 1, prefix, name)]
            got = v.detach().cpu().numpy().astype(np.float64)
            assert abs(got.sum() - ref_sum[0]) <= tol * max(1.0, ref_sum[1]), (k, prefix + name)
            key = "it%d.%s%s" % (k + 1, prefix, name)
            if key in g.files:
                assert np.abs(v.detach().cpu().numpy() - g[key]).max() <= tol, key


def test_td3_two_iterations_on_gpu_eager(golden_dir, monkeypatch):
    """TD3Agent.train x2 ON THE GPU from the reference's initial parameters with the reference's sampled indices and smoothing noise:
    parameters and targets afterwards match the reference's (td3.py:259-356) to 2e-5."""
    g, a, buf = _golden_agent(golden_dir, "cuda")
    noise = [torch.as_tensor(n).cuda() for n in g["noise"]]
    real_sample = buf.sample
    for k in range(2):
        monkeypatch.setattr(torch, "randn_like", lambda x, *a_, _k=k, **k_: noise[_k])
        monkeypatch.setattr(buf, "sample", lambda bs, _k=k: real_sample(bs, ind=g["idx"][_k]))
        a.train(buf, int(g["batch"]))
        _check_against_golden(g, a, k)
    assert a.total_it == 2


def test_td3_two_iterations_on_gpu_in_hip_graphs(golden_dir):
    """The same two iterations through what GraphedVecTD3Trainer captures: td3.td3_update inside a hipGraph with the capturable fused
    Adam the trainer installs (critic-only graph for iteration 1, critic+actor+targets graph for iteration 2), batch and noise fed through
    static tensors.  Same golden vectors, same tolerance."""
    from plen_ml_walk_amd import td3 as T
    from plen_ml_walk_amd.train_vec import GraphedVecTD3Trainer
    g, a, buf = _golden_agent(golden_dir, "cuda")
    a.actor_optimizer = GraphedVecTD3Trainer._capturable_adam(a.actor_optimizer, a.actor)
    a.critic_optimizer = GraphedVecTD3Trainer._capturable_adam(a.critic_optimizer, a.critic)
    B = int(g["batch"])
    idx = torch.zeros(B, dtype=torch.long, device="cuda")
    nz = torch.zeros(B, 18, device="cuda")
    loss = torch.zeros((), device="cuda")

    def it(with_policy):
        loss.copy_(T.td3_update(a, buf.sample(B, ind=idx), with_policy, noise=nz, all_reduce=False))

    # snapshot: warm-up runs (allocator, lazy init) must not count as iterations
    snap = [p.detach().clone() for net in (a.actor, a.critic, a.actor_target, a.critic_target) for p in net.parameters()]
    osnap = [a.actor_optimizer.state_dict(), a.critic_optimizer.state_dict()]
    import copy
    osnap = copy.deepcopy(osnap)
    graphs = {}
    side = torch.cuda.Stream()
    for wp in (False, True):
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            it(wp)
        torch.cuda.current_stream().wait_stream(side)
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            it(wp)
        graphs[wp] = gr
    with torch.no_grad():
        i = 0
        for net in (a.actor, a.critic, a.actor_target, a.critic_target):
            for p in net.parameters():
                p.copy_(snap[i]); i += 1
    # Adam state after the warm-ups: back to "never stepped" (in place: the graphs hold these tensors)
    for opt in (a.actor_optimizer, a.critic_optimizer):
        for st in opt.state.values():
            st["step"].zero_(); st["exp_avg"].zero_(); st["exp_avg_sq"].zero_()
    for k in range(2):
        idx.copy_(torch.as_tensor(g["idx"][k]).cuda()); nz.copy_(torch.as_tensor(g["noise"][k]).cuda())
        graphs[(k + 1) % 2 == 0].replay()
        torch.cuda.synchronize()
        _check_against_golden(g, a, k)
    assert torch.isfinite(loss)


def test_graphed_trainer_cadence_and_resumed_optimizer_state(tmp_path):
    """(a) The graph trainer's warm-up runs are the loop's real iterations: after k steps it has done exactly k collects and the same
    number of updates, with the same policy_freq alternation, as the eager trainer.  (b) Building it on a resumed agent keeps Adam's
    moments, step counts and learning rate (they used to be replaced by fresh optimisers)."""
    from plen_ml_walk_amd.td3 import ReplayBuffer, TD3Agent
    from plen_ml_walk_amd.train_vec import GraphedVecTD3Trainer, VecTD3Trainer
    n = 64
    env = _env(n); agent = TD3Agent(26, 18, 1.0, lr=1e-4); replay = ReplayBuffer(4000)
    tr = VecTD3Trainer(env, agent, replay, start_timesteps=128, batch_size=64, updates_per_step=1, seed=0)
    for _ in range(7):
        tr.step()
    assert tr.grad_steps == 6 and agent.total_it == 6
    before = {k: {kk: vv.clone() if torch.is_tensor(vv) else vv for kk, vv in st.items()} for k, st in agent.critic_optimizer.state_dict()["state"].items()}
    a_before = {k: {kk: vv.clone() if torch.is_tensor(vv) else vv for kk, vv in st.items()} for k, st in agent.actor_optimizer.state_dict()["state"].items()}
    env.close()
    env2 = _env(n)
    tr2 = GraphedVecTD3Trainer(env2, agent, replay, start_timesteps=128, batch_size=64, updates_per_step=1, seed=1)
    tr2.restore_counters({"env_steps": tr.env_steps, "grad_steps": tr.grad_steps})
    for opt, ref in ((agent.critic_optimizer, before), (agent.actor_optimizer, a_before)):
        assert opt.param_groups[0]["lr"] == 1e-4 and opt.param_groups[0]["capturable"]
        sd = opt.state_dict()["state"]
        assert len(sd) == len(ref) > 0
        for k in ref:
            assert torch.equal(sd[k]["exp_avg"], ref[k]["exp_avg"]) and torch.equal(sd[k]["exp_avg_sq"], ref[k]["exp_avg_sq"])
            assert float(sd[k]["step"]) == float(ref[k]["step"]) and sd[k]["step"].is_cuda
    assert float(next(iter(agent.critic_optimizer.state_dict()["state"].values()))["step"]) == 6.0
    assert float(next(iter(agent.actor_optimizer.state_dict()["state"].values()))["step"]) == 3.0
    for k in range(9):
        tr2.step()
        assert tr2.env_steps == tr.env_steps + (k + 1) * n and tr2.grad_steps == tr.grad_steps + k + 1
    torch.cuda.synchronize()
    assert int(tr2.total_t) == tr2.host_total == replay.size == 16 * n
    # 9 more critic steps, and actor steps on the even iterations 8, 10, 12, 14 only
    assert float(next(iter(agent.critic_optimizer.state_dict()["state"].values()))["step"]) == 15.0
    assert float(next(iter(agent.actor_optimizer.state_dict()["state"].values()))["step"]) == 7.0
    assert torch.isfinite(agent.last_critic_loss)
    env2.close()


def test_worker_streams_are_one_per_role_and_device():
    """vec_env.worker_stream: the same role always gets the same stream (a second pipelined env / trainer must not walk on through torch's stream pool
    onto shared hardware queues), different roles get different streams, and the update stream is the high-priority one."""
    from plen_ml_walk_amd.vec_env import worker_stream, PlenVecEnvPipelined
    dev = torch.device("cuda", 0)
    roles = [0, 1, 2, 3, "update", "side"]
    st = {r: worker_stream(dev, r) for r in roles}
    assert len({s.cuda_stream for s in st.values()}) == len(roles)
    assert all(worker_stream(dev, r).cuda_stream == st[r].cuda_stream for r in roles)
    assert st["update"].priority < st[0].priority and all(s.cuda_stream != torch.cuda.default_stream(dev).cuda_stream for s in st.values())
    e1 = PlenVecEnvPipelined(64, groups=2, device=dev)
    e2 = PlenVecEnvPipelined(64, groups=4, device=dev)
    assert [s.cuda_stream for s in e1.streams] == [st[0].cuda_stream, st[1].cuda_stream]
    assert [s.cuda_stream for s in e2.streams] == [st[k].cuda_stream for k in range(4)]
    a = torch.zeros(64, 18, device=dev)
    o1 = e1.step(a)[0].clone(); o2 = e2.step(a)[0].clone()          # two envs sharing role streams still step independently and identically
    torch.cuda.synchronize()
    assert torch.equal(o1, o2)
    e1.close(); e2.close()


# ------------------------------------------------------------------------------------------------ fused TD3 update (td3_fused.py + csrc/td3_kernels.hip)
def test_fused_td3_update_matches_the_reference_golden_iterations(golden_dir):
    """The hand-derived update (library GEMMs + the HIP kernels of libplentd3.so) on the reference's golden problem: same initial parameters,
    sampled indices and smoothing noise as td3.py produced -> the same parameters and targets after iteration 1 (critic only) and
    iteration 2 (critic, actor, Polyak), to the tolerance the autograd path is held to."""
    from plen_ml_walk_amd.td3_fused import FusedTD3
    g, a, buf = _golden_agent(golden_dir, "cuda")
    fz = FusedTD3(a)
    for k in range(2):
        idx = torch.as_tensor(g["idx"][k]).cuda().long()
        noise = torch.as_tensor(g["noise"][k]).cuda()
        a.total_it += 1
        loss = fz.update(buf.data, idx, with_policy=a.total_it % a.policy_freq == 0, noise=noise, all_reduce=False)
        torch.cuda.synchronize()
        assert torch.isfinite(loss)
        _check_against_golden(g, a, k)


def test_reference_golden_iterations_through_the_small_batch_kernels(golden_dir):
    """The reference's own two recorded train() iterations (td3.py:259-356 at its batch 100: sampled indices and smoothing noise captured from the reference,
    tests/golden/td3_train.npz) replayed through k_critic_team / k_policy_team / k_wgrad_adam_group (explicit idx / noise in PlenTd3CriticRows): parameters and
    targets after iteration 1 (critic only) and iteration 2 (critic, actor, Polyak) match the reference's to the tolerance the autograd path is held to."""
    from plen_ml_walk_amd.td3_fused import FusedTD3
    g, a, buf = _golden_agent(golden_dir, "cuda")
    fz = FusedTD3(a, team=True)
    fz.enable_flat_adam()
    for k in range(2):
        idx = torch.as_tensor(g["idx"][k]).cuda().long().contiguous()
        noise = torch.as_tensor(g["noise"][k]).cuda().contiguous()
        a.total_it += 1
        loss = fz.update(buf.data, idx, with_policy=a.total_it % a.policy_freq == 0, noise=noise, all_reduce=False)
        torch.cuda.synchronize()
        assert fz._team_pass and "critic" in fz._fused_done and torch.isfinite(loss)
        _check_against_golden(g, a, k)


@pytest.mark.parametrize("B,total", [(4096, 50000), (100, 700), (16, 40), (3, 5)])
def test_critic_rows_kernel_equals_the_layer_by_layer_update(B, total):
    """plentd3_critic_rows (one launch, 16 batch rows per wave through sampling, targets, critic forward and backward on the matrix cores) against
    FusedTD3.critic_backward (library GEMMs + one kernel per step) from the same random state: same sampled rows, same smoothing noise (both draw
    Philox numbers from the same counter), so the loss and every critic gradient agree to f32 summation order; the call counter advances once."""
    from plen_ml_walk_amd import td3 as T
    from plen_ml_walk_amd.td3_fused import FusedTD3
    torch.manual_seed(11)
    ag = T.TD3Agent(26, 18, 1.0, data_parallel=False)
    for net in (ag.actor_target, ag.critic_target, ag.critic):            # away from the all-equal initial targets
        for p_ in net.parameters():
            p_.data.add_(0.05 * torch.randn_like(p_))
    fz = FusedTD3(ag, seed=9)
    data = torch.randn(1000, 72, device="cuda")
    data[:, 70] = torch.rand(1000, device="cuda")                         # reward
    data[:, 71] = (torch.rand(1000, device="cuda") > 0.1).float()         # not_done
    tot = torch.tensor(total, dtype=torch.long, device="cuda")
    rng0 = fz.rng.clone()
    fz.rows, fz.team = False, False                                        # layer by layer
    loss_a = fz.critic_backward(data, B, total=tot, guard=64).clone()
    grads_a = ag._critic_grads.flat.clone()
    saved_a = [t.clone() for t in fz._saved[:2]]
    assert int(fz.rng[1]) == int(rng0[1]) + 1
    fz.rng.copy_(rng0)
    fz.rows = True
    loss_b = fz.critic_backward(data, B, total=tot, guard=64).clone()
    grads_b = ag._critic_grads.flat.clone()
    torch.cuda.synchronize()
    assert int(fz.rng[1]) == int(rng0[1]) + 1 and int(fz._done_count) == 0
    assert torch.equal(saved_a[0], fz._saved[0]) and torch.equal(saved_a[1][:, :26], fz._saved[1][:, :26])       # the same replay rows were drawn
    assert torch.isfinite(loss_b) and abs(float(loss_a) - float(loss_b)) <= 2e-5 * max(1.0, abs(float(loss_a)))
    scale = float(grads_a.abs().max())
    assert scale > 0 and float((grads_a - grads_b).abs().max()) <= 2e-5 * scale
    # and per layer, so that a wrong small block cannot hide behind a large one
    off = 0
    for p_ in T._flat_order(ag.critic):
        n = p_.numel()
        ga, gb = grads_a[off:off + n], grads_b[off:off + n]
        assert float((ga - gb).abs().max()) <= 1e-4 * max(float(ga.abs().max()), 1e-6), tuple(p_.shape)
        off += n


@pytest.mark.parametrize("B", [4096, 100, 3])
def test_policy_rows_kernel_equals_the_layer_by_layer_policy_gradient(B):
    """plentd3_policy_rows (actor forward, Q1 forward, the gradient of -mean Q1 back through critic and actor, one launch) + the three weight-gradient
    launches against FusedTD3.policy_backward's layer-by-layer path on the same batch: every actor gradient agrees to f32 summation order."""
    from plen_ml_walk_amd import td3 as T
    from plen_ml_walk_amd.td3_fused import FusedTD3
    torch.manual_seed(12)
    ag = T.TD3Agent(26, 18, 1.0, data_parallel=False)
    for net in (ag.actor, ag.critic):
        for p_ in net.parameters():
            p_.data.add_(0.05 * torch.randn_like(p_))
    fz = FusedTD3(ag, seed=3, team=False)
    data = torch.randn(500, 72, device="cuda")
    idx = torch.randint(0, 500, (B,), device="cuda")
    out = {}
    for rows in (False, True):
        fz.rows = False
        fz.critic_backward(data, idx, noise=torch.zeros(B, 18, device="cuda"))
        fz.rows = rows
        fz.policy_backward()
        torch.cuda.synchronize()
        out[rows] = ag._actor_grads.flat.clone()
    off = 0
    for p_ in T._flat_order(ag.actor):
        n = p_.numel()
        ga, gb = out[False][off:off + n], out[True][off:off + n]
        assert float(ga.abs().max()) > 0 and float((ga - gb).abs().max()) <= 1e-4 * float(ga.abs().max()), tuple(p_.shape)
        off += n


@pytest.mark.parametrize("B,total", [(100, 700), (512, 50000), (16, 40), (37, 900), (3, 5)])
def test_small_batch_team_kernels_equal_the_layer_by_layer_update(B, total):
    """The small-batch shape of the update (csrc/td3_team.hip: a team of 8 waves per 4 batch rows; every weight gradient of a pass in one
    plentd3_wgrad_group launch) against the layer-by-layer path (library GEMMs + one kernel per step) from the same random state: the reference's
    batch 100 (plen_td3.py:28), the largest batch that takes this path by default, one row block exactly, a ragged last block, fewer rows than a
    group of four.  Same sampled rows, same smoothing noise; loss, every critic gradient and every actor gradient agree to f32 summation order;
    the call counter advances once; and two runs of the team path are bitwise equal (no sum depends on the order workgroups finish in)."""
    from plen_ml_walk_amd import td3 as T
    from plen_ml_walk_amd.td3_fused import FusedTD3
    torch.manual_seed(21)
    ag = T.TD3Agent(26, 18, 1.0, data_parallel=False)
    for net in (ag.actor, ag.actor_target, ag.critic_target, ag.critic):
        for p_ in net.parameters():
            p_.data.add_(0.05 * torch.randn_like(p_))
    fz = FusedTD3(ag, seed=9)
    data = torch.randn(1000, 72, device="cuda")
    data[:, 70] = torch.rand(1000, device="cuda")
    data[:, 71] = (torch.rand(1000, device="cuda") > 0.1).float()
    tot = torch.tensor(total, dtype=torch.long, device="cuda")
    rng0 = fz.rng.clone()
    res = {}
    for name, team in (("layers", False), ("team", True), ("team again", True)):
        fz.rng.copy_(rng0)
        fz.team = team
        ag._critic_grads.zero(); ag._actor_grads.zero(); fz._zeroed = {}
        loss = fz.critic_backward(data, B, total=tot, guard=64).clone()
        assert fz._team_pass == team
        fz.policy_backward()
        torch.cuda.synchronize()
        assert int(fz.rng[1]) == int(rng0[1]) + 1
        res[name] = (loss, ag._critic_grads.flat.clone(), ag._actor_grads.flat.clone(), fz._saved[0].clone(), fz._saved[1].clone())
    assert int(fz._done_count) == 0
    la, ca, aa, sa_, pa = res["layers"]
    lb, cb, ab, sb, pb = res["team"]
    assert torch.equal(sa_, sb) and torch.equal(pa[:, :26], pb[:, :26])                         # the same replay rows were drawn
    assert float((pa[:, 26:] - pb[:, 26:]).abs().max()) <= 2e-5                                  # actor(s): the policy pass's actions
    assert torch.isfinite(lb) and abs(float(la) - float(lb)) <= 2e-5 * max(1.0, abs(float(la)))
    for net, ga_all, gb_all in ((ag.critic, ca, cb), (ag.actor, aa, ab)):
        assert float(ga_all.abs().max()) > 0 and float((ga_all - gb_all).abs().max()) <= 2e-5 * float(ga_all.abs().max())
        off = 0
        for p_ in T._flat_order(net):                   # per layer, so that a wrong small block cannot hide behind a large one
            n = p_.numel()
            ga, gb = ga_all[off:off + n], gb_all[off:off + n]
            assert float((ga - gb).abs().max()) <= 1e-4 * max(float(ga.abs().max()), 1e-6), tuple(p_.shape)
            off += n
    for x, y in zip(res["team"], res["team again"]):          # same bits every run: one workgroup per output tile, per-workgroup partial sums added in order
        assert torch.equal(x, y)


def test_adam_step_inside_the_weight_gradient_kernel_equals_the_separate_step():
    """plentd3_wgrad_adam_group (small batch, one rank: each gradient element is stepped by the workgroup that produced it, the bucket stays zero) against
    plentd3_wgrad_group + plentd3_adam: six updates of batch 100 from the same state leave bitwise equal parameters, targets, moments and step counts."""
    from plen_ml_walk_amd import td3 as T
    from plen_ml_walk_amd.td3_fused import FusedTD3
    data = torch.randn(5000, 72, device="cuda")
    data[:, 70] = torch.rand(5000, device="cuda"); data[:, 71] = (torch.rand(5000, device="cuda") > 0.1).float()
    tot = torch.tensor(5000, dtype=torch.long, device="cuda")
    out = {}
    for fuse in (False, True):
        torch.manual_seed(31)
        ag = T.TD3Agent(26, 18, 1.0, data_parallel=False)
        fz = FusedTD3(ag, seed=4, team=True)
        fz.enable_flat_adam()
        fz.fuse_adam = fuse
        for k in range(6):
            loss = fz.update(data, 100, with_policy=(k % 2 == 1), all_reduce=False, total=tot)
            assert ("critic" in fz._fused_done) == fuse and ("actor" in fz._fused_done) == (fuse and k % 2 == 1)
        torch.cuda.synchronize()
        assert float(ag._critic_grads.flat.abs().max()) == 0.0 and float(ag._actor_grads.flat.abs().max()) == 0.0
        out[fuse] = [loss.clone(), ag._critic_flat.flat.clone(), ag._actor_flat.flat.clone(), ag._critic_target_flat.flat.clone(), ag._actor_target_flat.flat.clone(),
                     fz._critic_adam.m.clone(), fz._critic_adam.v.clone(), fz._actor_adam.m.clone(), fz._actor_adam.v.clone(),
                     fz._critic_adam.step_t.clone(), fz._actor_adam.step_t.clone()]
        assert float(out[fuse][-2]) == 6.0 and float(out[fuse][-1]) == 3.0
    for a_, b_ in zip(out[False], out[True]):
        assert torch.equal(a_, b_)


def test_cached_eager_small_batch_iteration_equals_the_general_path():
    """FusedTD3._update_team_eager (persistent scratch tensors and argument blocks for the reference's eager train() call) against the general path that rebuilds
    everything per call: eight updates of batch 100 (every second one with the policy update) from the same state leave bitwise equal parameters, targets, moments,
    step counts and losses; a second batch size gets its own cache entry."""
    from plen_ml_walk_amd import td3 as T
    from plen_ml_walk_amd.td3_fused import FusedTD3
    data = torch.randn(5000, 72, device="cuda")
    data[:, 70] = torch.rand(5000, device="cuda"); data[:, 71] = (torch.rand(5000, device="cuda") > 0.1).float()
    tot = torch.tensor(5000, dtype=torch.long, device="cuda")
    out = {}
    for cached in (False, True):
        torch.manual_seed(35)
        ag = T.TD3Agent(26, 18, 1.0, data_parallel=False)
        fz = FusedTD3(ag, seed=4)
        fz.enable_flat_adam()
        fz.eager_cache = cached
        losses = []
        for k in range(8):
            losses.append(fz.update(data, 100 if k < 6 else 64, with_policy=(k % 2 == 1), all_reduce=False, total=tot).clone())
        torch.cuda.synchronize()
        assert (len(fz._eager) == 2) == cached and fz._team_pass
        out[cached] = losses + [ag._critic_flat.flat.clone(), ag._actor_flat.flat.clone(), ag._critic_target_flat.flat.clone(), ag._actor_target_flat.flat.clone(),
                                fz._critic_adam.m.clone(), fz._critic_adam.v.clone(), fz._actor_adam.m.clone(), fz._critic_adam.step_t.clone(), fz._actor_adam.step_t.clone()]
        assert float(out[cached][-2]) == 8.0 and float(out[cached][-1]) == 4.0
    for a_, b_ in zip(out[False], out[True]):
        assert torch.equal(a_, b_)


def test_team_order_pack_kernel_matches_its_definition():
    """plentd3_pack with PlenTd3PackJob.team = 1 (include/plentd3.h): dst float4 (((t NS + s) 8 + c) 64 + lane) = M[32 t + lane % 32][64 s + 32 (lane / 32) + 4 c + (0..3)],
    zero beyond the matrix -- for the shapes the small-batch kernels read (256 x 26, 512 x 44, 256 x 256, 18 x 256)."""
    import ctypes as C
    from plen_ml_walk_amd import td3_fused as F
    lib = F.load()
    for N, K in ((256, 26), (512, 44), (256, 256), (18, 256)):
        w = torch.randn(N, K, device="cuda")
        T_, NS = -(-N // 32), -(-K // 64)
        dst = torch.full((T_ * NS * 2048,), 7.0, device="cuda")
        G = F.PackGroup()
        J = G.job[0]
        J.src, J.dst, J.rs, J.cs, J.N, J.K, J.team = w.data_ptr(), dst.data_ptr(), K, 1, N, K, 1
        G.n_jobs = 1
        assert lib.plentd3_pack(C.byref(G), None) == 0
        torch.cuda.synchronize()
        wp = torch.zeros(T_ * 32, NS * 64, device="cuda"); wp[:N, :K] = w
        # [t][s][c][half][col][j] <- wp[32 t + col][64 s + 32 half + 4 c + j]
        ref = wp.view(T_, 32, NS, 2, 8, 4).permute(0, 2, 4, 3, 1, 5).contiguous().view(-1)
        assert torch.equal(dst, ref), (N, K)


def test_packed_small_batch_weights_stay_current_and_change_no_bit(monkeypatch):
    """Round 6: the small-batch kernels read their forward products' weights in operand order (no LDS parking), and the fused weight-gradient + Adam launches write every
    parameter they step -- and its Polyak target -- into the packed copies, so a chain of batch-100 updates needs no packing launch.  Same values into the same matrix
    instructions in the same order: twelve updates (every second one with the policy update; a torch-level write to the actor and an update whose Adam steps run as
    separate kernels in between, both of which must trigger a re-pack) leave bitwise the parameters, targets, moments and losses of PLEN_TD3_TEAM_PACKED=0; at the end every packed copy equals a fresh
    pack of its matrix."""
    from plen_ml_walk_amd import td3 as T
    from plen_ml_walk_amd import td3_fused as F
    data = torch.randn(5000, 72, device="cuda")
    data[:, 70] = torch.rand(5000, device="cuda"); data[:, 71] = (torch.rand(5000, device="cuda") > 0.1).float()
    tot = torch.tensor(5000, dtype=torch.long, device="cuda")
    out = {}
    for packed in ("0", "1"):
        monkeypatch.setenv("PLEN_TD3_TEAM_PACKED", packed)
        torch.manual_seed(41)
        ag = T.TD3Agent(26, 18, 1.0, data_parallel=False)
        fz = F.FusedTD3(ag, seed=9)
        fz.enable_flat_adam()
        assert fz._team_packed == (packed == "1")
        losses = []
        for k in range(12):
            if k == 5:
                with torch.no_grad():
                    ag.actor.fc2.weight.mul_(1.001)                  # a torch-level write (as load_state_dict would be): version counters move
            if k == 8:
                fz.fuse_adam = False                                # one update whose Adam steps are separate kernels (they know nothing of the packed copies; deterministic,
                fz.update(data, 100, with_policy=True, all_reduce=False, total=tot)          # unlike the layer-by-layer path's float atomics)
                fz.fuse_adam = True
            losses.append(fz.update(data, 100 if k % 3 else 64, with_policy=(k % 2 == 1), all_reduce=False, total=tot).clone())
            assert fz._team_pass and fz._tp_live == (packed == "1")
        torch.cuda.synchronize()
        out[packed] = losses + [ag._critic_flat.flat.clone(), ag._actor_flat.flat.clone(), ag._critic_target_flat.flat.clone(), ag._actor_target_flat.flat.clone(),
                                fz._critic_adam.m.clone(), fz._actor_adam.v.clone(), fz._critic_adam.step_t.clone()]
        if packed == "1":
            have = {k: v.clone() for k, v in fz._tpacks.items()}
            fz._tp_state = None
            assert fz._team_pack_sync()                              # a fresh pack of every matrix as it is now
            torch.cuda.synchronize()
            assert len(have) == 12
            for k, v in have.items():
                assert torch.equal(v, fz._tpacks[k]), k
    for a_, b_ in zip(out["0"], out["1"]):
        assert torch.equal(a_, b_)


def test_resumed_optimizer_keeps_its_step_count_through_the_fused_small_batch_update():
    """optimizer.load_state_dict() between updates (a resumed run) replaces the state tensors FlatAdam mirrors; the re-bind has to happen before the pass
    kernel counts the step, or the fused path's counter stays one behind the separate-Adam path's for good: after a reload at step 4, four
    more updates leave bitwise equal parameters, moments and step counts (8 / 4) on both paths."""
    from plen_ml_walk_amd import td3 as T
    from plen_ml_walk_amd.td3_fused import FusedTD3
    data = torch.randn(5000, 72, device="cuda")
    data[:, 70] = torch.rand(5000, device="cuda"); data[:, 71] = (torch.rand(5000, device="cuda") > 0.1).float()
    tot = torch.tensor(5000, dtype=torch.long, device="cuda")
    out = {}
    for fuse in (False, True):
        torch.manual_seed(33)
        ag = T.TD3Agent(26, 18, 1.0, data_parallel=False)
        fz = FusedTD3(ag, seed=4, team=True)
        fz.enable_flat_adam()
        fz.fuse_adam = fuse
        for k in range(4):
            fz.update(data, 100, with_policy=(k % 2 == 1), all_reduce=False, total=tot)
        torch.cuda.synchronize()
        for opt in (ag.critic_optimizer, ag.actor_optimizer):           # what TD3Agent.load() does with the two optimizer files
            sd = opt.state_dict()
            sd = {"state": {k_: {n: (v.clone() if torch.is_tensor(v) else v) for n, v in st.items()} for k_, st in sd["state"].items()}, "param_groups": sd["param_groups"]}
            opt.load_state_dict(sd)
        for k in range(4, 8):
            fz.update(data, 100, with_policy=(k % 2 == 1), all_reduce=False, total=tot)
        torch.cuda.synchronize()
        out[fuse] = [ag._critic_flat.flat.clone(), ag._actor_flat.flat.clone(), fz._critic_adam.m.clone(), fz._critic_adam.v.clone(), fz._actor_adam.m.clone(),
                     fz._critic_adam.step_t.clone(), fz._actor_adam.step_t.clone()]
        assert float(out[fuse][-2]) == 8.0 and float(out[fuse][-1]) == 4.0, (fuse, float(out[fuse][-2]), float(out[fuse][-1]))
    for a_, b_ in zip(out[False], out[True]):
        assert torch.equal(a_, b_)


def test_agent_given_an_index_less_device_still_takes_the_fused_iteration():
    """TD3Agent(device="cuda") (no index) and the replay tensor's cuda:0 used to compare unequal, which silently sent train() down the ~170-launch autograd
    iteration; the size scalar the fused update reads is one persistent device tensor, refilled in place."""
    from plen_ml_walk_amd import td3 as T
    a = T.TD3Agent(26, 18, 1.0, device="cuda", data_parallel=False)
    buf = T.ReplayBuffer(5000, device="cuda")
    assert a.device.index is not None and buf.device.index is not None
    buf.add_batch(torch.randn(300, 26), torch.rand(300, 18) * 2 - 1, torch.randn(300, 26), torch.randn(300), (torch.rand(300) < 0.05).float())
    a.train(buf, 100)
    assert a._fused is not None and a._fused._team_pass
    p = buf.size_on_device().data_ptr()
    buf.add_batch(torch.randn(10, 26), torch.rand(10, 18) * 2 - 1, torch.randn(10, 26), torch.randn(10), torch.zeros(10))
    assert buf.size_on_device().data_ptr() == p and int(buf.size_on_device()) == 310


def test_flat_adam_refuses_a_misaligned_view():
    """plentd3_adam moves four floats per access: a sub-view at an odd element offset is refused (-hipErrorInvalidValue = -1), not faulted on."""
    from plen_ml_walk_amd import td3_fused as F
    lib = F.load()
    n = 1000
    p, g, m, v = (torch.zeros(n + 4, device="cuda") for _ in range(4))
    step, done = torch.zeros((), device="cuda"), torch.zeros(1, dtype=torch.int32, device="cuda")
    st = F.C.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert lib.plentd3_adam(F._p(p[1:]), F._p(g), F._p(m), F._p(v), F._p(step), F._p(done), n, 3e-4, 0.9, 0.999, 1e-8, 0, None, 0.0, None, st) == -1
    assert lib.plentd3_adam(F._p(p), F._p(g), F._p(m), F._p(v), F._p(step), F._p(done), n, 3e-4, 0.9, 0.999, 1e-8, 0, None, 0.0, F._p(p[2:]), st) == -1
    assert lib.plentd3_adam(F._p(p), F._p(g), F._p(m), F._p(v), F._p(step), F._p(done), n, 3e-4, 0.9, 0.999, 1e-8, 0, None, 0.0, None, st) == 0
    torch.cuda.synchronize()
    assert float(step) == 1.0


def test_agent_train_takes_the_fused_iteration_on_the_device_and_honours_a_callers_sampling(tmp_path):
    """TD3Agent.train(replay_buffer, 100) -- the reference's call, plen_td3.py:119-120 -- on a HIP device with the device-resident ReplayBuffer runs the
    fused small-batch update (td3.py:259-356's arithmetic; indices and smoothing noise from the agent's Philox stream): step counts, Polyak cadence and
    checkpoints behave as with the autograd iteration; an overridden `sample` (instance or subclass) or fused_train = False takes the autograd path."""
    from plen_ml_walk_amd import td3 as T
    torch.manual_seed(3)
    a = T.TD3Agent(26, 18, 1.0, data_parallel=False)
    buf = T.ReplayBuffer(5000)
    buf.add_batch(torch.randn(600, 26), torch.rand(600, 18) * 2 - 1, torch.randn(600, 26), torch.randn(600), (torch.rand(600) < 0.05).float())
    p0 = a._critic_flat.flat.clone(); q0 = a._actor_flat.flat.clone(); t0 = a._actor_target_flat.flat.clone()
    a.train(buf, 100)
    assert a._fused is not None and "critic" in a._fused._fused_done and a._fused._team_pass
    torch.cuda.synchronize()
    assert torch.isfinite(a.last_critic_loss) and not torch.equal(p0, a._critic_flat.flat)
    assert torch.equal(q0, a._actor_flat.flat) and torch.equal(t0, a._actor_target_flat.flat)          # iteration 1: no policy update, no Polyak step
    a.train(buf, 100)
    torch.cuda.synchronize()
    assert not torch.equal(q0, a._actor_flat.flat) and not torch.equal(t0, a._actor_target_flat.flat) and a.total_it == 2
    for _ in range(4):
        a.train(buf, 100)
    sd_c, sd_a = a.critic_optimizer.state_dict(), a.actor_optimizer.state_dict()
    assert float(next(iter(sd_c["state"].values()))["step"]) == 6.0 and float(next(iter(sd_a["state"].values()))["step"]) == 3.0
    # checkpoints round-trip (td3.py:358-376's four files) and training continues from them
    a.save(str(tmp_path / "ck"))
    b = T.TD3Agent(26, 18, 1.0, data_parallel=False)
    b.load(str(tmp_path / "ck"))
    assert torch.equal(a._critic_flat.flat, b._critic_flat.flat) and torch.equal(a._actor_flat.flat, b._actor_flat.flat)
    b.train(buf, 100)
    torch.cuda.synchronize()
    assert float(next(iter(b.critic_optimizer.state_dict()["state"].values()))["step"]) == 7.0
    # a caller's own sampling is honoured: the autograd iteration runs on the batch it returns
    seen = []
    real = buf.sample
    buf.sample = lambda bs, ind=None: (seen.append(bs), real(bs, ind))[1]
    n_before = int(a._fused.rng[1])
    a.train(buf, 100)
    assert seen == [100] and int(a._fused.rng[1]) == n_before
    del buf.sample

    class Mine(T.ReplayBuffer):
        def sample(self, batch_size, ind=None):
            seen.append(-batch_size)
            return super().sample(batch_size, ind)
    mine = Mine(1000)
    mine.add_batch(torch.randn(300, 26), torch.rand(300, 18) * 2 - 1, torch.randn(300, 26), torch.randn(300), torch.zeros(300))
    a.train(mine, 64)
    assert seen[-1] == -64 and int(a._fused.rng[1]) == n_before
    a.fused_train = False
    a.train(buf, 100)
    torch.cuda.synchronize()
    assert int(a._fused.rng[1]) == n_before and torch.isfinite(a.last_critic_loss)


def test_single_transition_add_and_one_kernel_select_action_equal_the_general_paths(golden_dir):
    """The drop-in loop's per-step calls on a HIP device: ReplayBuffer.add (one packed row, one copy) writes what add_batch writes, ring wrap included;
    TD3Agent.select_action (one kernel, pinned buffers) returns the torch layers' action on the reference's shipped policy to f32 summation order."""
    from plen_ml_walk_amd import td3 as T
    rng = np.random.default_rng(5)
    a_buf, b_buf = T.ReplayBuffer(7), T.ReplayBuffer(7)
    for k in range(17):                                   # more than the capacity: the ring wraps twice
        s_, a_, s2 = rng.normal(size=26), rng.uniform(-1, 1, 18).astype(np.float32), rng.normal(size=26)
        r, d = np.float64(rng.normal()), float(k % 5 == 0)
        a_buf.add((s_, a_, s2, r, d))
        t = lambda x: torch.as_tensor(np.asarray(x, dtype=np.float32)).reshape(1, -1)
        b_buf.add_batch(t(s_), t(a_), t(s2), t(r), t(d))
        assert (a_buf.size, a_buf.ptr) == (b_buf.size, b_buf.ptr)
    assert torch.equal(a_buf.data, b_buf.data) and len(a_buf.storage) == 7
    ag = T.TD3Agent(26, 18, 1.0, data_parallel=False)
    ag.load_arrays(np.load(os.path.join(golden_dir, "policy_3229999.npz")))
    obs = rng.normal(size=(64, 26)) * np.array([1.0] * 18 + [0.1, 0.3, 0.3, 0.3, 0.3, 0.3, 1.0, 1.0])
    fast = np.stack([ag.select_action(o) for o in obs])
    assert ag._select_state is not None and fast.dtype == np.float32 and fast.shape == (64, 18)
    ag.fused_select = False
    slow = np.stack([ag.select_action(o) for o in obs])
    assert np.abs(fast - slow).max() <= 1e-4 and np.abs(slow).max() > 0.1          # (the shipped policy's pre-activations are O(10): 3e-5 observed)


@pytest.mark.parametrize("B", [100, 98, 101])
def test_small_batch_update_keeps_its_scratch_rows_inside_the_batch(B):
    """Canary rows behind every per-iteration scratch matrix of a team-path update stay untouched: the reference's batch 100 (25 whole blocks of 4 rows) and two
    batches whose last block is ragged (98 = 24 blocks + 2 rows, 101 = 25 + 1: the guarded stores of k_critic_team / k_policy_team)."""
    from plen_ml_walk_amd import td3 as T
    from plen_ml_walk_amd.td3_fused import FusedTD3
    torch.manual_seed(5)
    ag = T.TD3Agent(26, 18, 1.0, data_parallel=False)
    fz = FusedTD3(ag, seed=1, team=True)
    made = []

    def alloc(*shape):
        full = torch.full((shape[0] + 16,) + tuple(shape[1:]), 777.0, device="cuda")
        made.append((full, shape[0]))
        return full[:shape[0]]
    fz._alloc = alloc
    data = torch.randn(400, 72, device="cuda")
    fz.critic_backward(data, B, total=torch.tensor(400, dtype=torch.long, device="cuda"))
    fz.policy_backward()
    torch.cuda.synchronize()
    assert len(made) >= 10 and fz._team_pass
    for full, n in made:
        assert bool((full[n:] == 777.0).all()), tuple(full.shape)


def test_flat_adam_kernel_equals_torch_adam_over_ragged_sizes():
    """plentd3_adam (<= 128 workgroups of 256 lanes, 4 floats per lane and trip) against torch.optim.Adam for sizes that are not multiples of 4 or of a
    workgroup's span, three steps, with the zeroed gradient, Polyak target and parameter copy of the same pass."""
    from plen_ml_walk_amd import td3_fused as F
    lib = F.load()
    for n in (1, 3, 1023, 1025, 154114, 300001):
        torch.manual_seed(n)
        p = torch.randn(n, device="cuda"); ref = p.clone().requires_grad_(True)
        opt = torch.optim.Adam([ref], lr=3e-4)
        m, v = torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
        step, done = torch.zeros((), device="cuda"), torch.zeros(1, dtype=torch.int32, device="cuda")
        target, copy = torch.randn(n, device="cuda"), torch.empty(n, device="cuda")
        t_ref = target.clone()
        st = F.C.c_void_p(torch.cuda.current_stream().cuda_stream)
        for k in range(3):
            g = torch.randn(n, device="cuda") * (10.0 ** (k - 1))
            ref.grad = g.clone()
            opt.step()
            F._chk(lib.plentd3_adam(F._p(p), F._p(g), F._p(m), F._p(v), F._p(step), F._p(done), n, 3e-4, 0.9, 0.999, 1e-8, 1, F._p(target), 0.005, F._p(copy), st))
            t_ref = 0.005 * ref.detach() + 0.995 * t_ref
            torch.cuda.synchronize()
            assert float(step) == k + 1 and int(done) == 0 and float(g.abs().max()) == 0.0
            assert float((p - ref.detach()).abs().max()) <= 1e-6, (n, k)
            assert torch.equal(copy, p) and float((target - t_ref).abs().max()) <= 1e-6


@pytest.mark.parametrize("n", [2048, 37])
def test_actor_rows_kernel_equals_the_layer_by_layer_exploration_action(n):
    """plentd3_actor_rows (the collect phase's actor forward + exploration noise + clip in one launch) against FusedTD3.explore's GEMMs + plentd3_explore
    from the same random state: same Philox draws, so the actions agree to f32 summation order in the three layers."""
    from plen_ml_walk_amd import td3 as T
    from plen_ml_walk_amd.td3_fused import FusedTD3
    torch.manual_seed(13)
    ag = T.TD3Agent(26, 18, 1.0, data_parallel=False)
    fz = FusedTD3(ag, seed=4)
    rng = FusedTD3.new_rng("cuda", 77)
    state = torch.randn(n, 26, device="cuda")
    fz.rows = False
    a0 = fz.explore(state, 0.1, rng=rng).clone()
    fz.rows = True
    a1 = fz.explore(state, 0.1, rng=rng).clone()
    torch.cuda.synchronize()
    assert a0.shape == a1.shape == (n, 18) and float(a0.abs().max()) <= 1.0 and float(a1.abs().max()) <= 1.0
    assert float((a0 - a1).abs().max()) <= 2e-5


def test_flat_adam_kernel_equals_torch_adam():
    """plentd3_adam (one kernel over the flat buffers, with the fused gradient zeroing and Polyak update) against torch.optim.Adam on the same
    gradients for 5 steps: parameters, both moments and the step counter agree; optimizer.state_dict() still describes the state."""
    from plen_ml_walk_amd import td3 as T
    from plen_ml_walk_amd.td3_fused import FusedTD3
    from plen_ml_walk_amd.train_vec import GraphedVecTD3Trainer
    torch.manual_seed(2)
    ref = T.TD3Agent(26, 18, 1.0, data_parallel=False)
    new = T.TD3Agent(26, 18, 1.0, data_parallel=False)
    for dst, src in ((new.actor, ref.actor), (new.critic, ref.critic), (new.actor_target, ref.actor_target), (new.critic_target, ref.critic_target)):
        dst.load_state_dict(src.state_dict())
    new.critic_optimizer = GraphedVecTD3Trainer._capturable_adam(new.critic_optimizer, new.critic)
    new.actor_optimizer = GraphedVecTD3Trainer._capturable_adam(new.actor_optimizer, new.actor)
    fz = FusedTD3(new)
    fz.enable_flat_adam()
    tgt0 = new._critic_target_flat.flat.clone()
    for k in range(5):
        g = torch.randn_like(ref._critic_grads.flat) * (10.0 ** (k - 2))
        ref._critic_grads.flat.copy_(g); new._critic_grads.flat.copy_(g)
        ref.critic_optimizer.step()
        if k == 4:
            fz._critic_adam.step(zero_grad=True, target=new._critic_target_flat.flat, tau=0.005)
        else:
            new.critic_optimizer.step()                      # the patched step(): the same kernel, no extras
    torch.cuda.synchronize()
    a, b = ref._critic_flat.flat, new._critic_flat.flat
    assert float((a - b).abs().max()) <= 1e-6 * max(1.0, float(a.abs().max()))
    assert float(new._critic_grads.flat.abs().max()) == 0.0 and float(fz._critic_adam.step_t) == 5.0
    assert torch.allclose(new._critic_target_flat.flat, 0.005 * b + 0.995 * tgt0, atol=1e-7)
    # a checkpoint loaded into the live optimiser is carried over into the flat buffers at the next step
    sd = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in new.critic_optimizer.state_dict().items()}
    sd["state"] = {i: {k: v.clone() for k, v in st.items()} for i, st in new.critic_optimizer.state_dict()["state"].items()}
    m_before = fz._critic_adam.m.clone()
    fz._critic_adam.m.zero_(); fz._critic_adam.step_t.fill_(99.0)
    new.critic_optimizer.load_state_dict(sd)
    fz._critic_adam.bind()
    assert torch.equal(fz._critic_adam.m, m_before) and float(fz._critic_adam.step_t) == 5.0
    rs, ns = ref.critic_optimizer.state_dict()["state"], new.critic_optimizer.state_dict()["state"]
    # state entries are indexed by parameter order of module.parameters() in both optimisers
    for i in rs:
        assert float(ns[i]["step"]) == 5.0
        for k in ("exp_avg", "exp_avg_sq"):
            assert float((rs[i][k] - ns[i][k]).abs().max()) <= 1e-6 * float(rs[i][k].abs().max()), k


@pytest.mark.parametrize("B", [256, 4096])
def test_fused_td3_update_equals_autograd_update(B):
    """Same random problem through td3.td3_update (autograd) and FusedTD3.update: loss, every gradient and every parameter / target after a
    critic-only and a critic + policy iteration agree to f32 GEMM rounding (hence also the flat-buffer layout and the stacked first layers)."""
    from plen_ml_walk_amd import td3 as T
    from plen_ml_walk_amd.td3_fused import FusedTD3
    torch.manual_seed(3)
    ref = T.TD3Agent(26, 18, 1.0, data_parallel=False)
    fus = T.TD3Agent(26, 18, 1.0, data_parallel=False)
    for dst, src in ((fus.actor, ref.actor), (fus.critic, ref.critic), (fus.actor_target, ref.actor_target), (fus.critic_target, ref.critic_target)):
        dst.load_state_dict(src.state_dict())
    fz = FusedTD3(fus)
    buf = T.ReplayBuffer(3 * B)
    gen = torch.Generator(device="cuda").manual_seed(5)
    n = 2 * B
    buf.add_batch(torch.randn(n, 26, generator=gen, device="cuda"), torch.rand(n, 18, generator=gen, device="cuda") * 2 - 1,
                  torch.randn(n, 26, generator=gen, device="cuda"), torch.randn(n, generator=gen, device="cuda") * 3,
                  (torch.rand(n, generator=gen, device="cuda") < 0.1).float())
    for it in range(4):
        idx = torch.randint(0, n, (B,), generator=gen, device="cuda")
        noise = torch.randn(B, 18, generator=gen, device="cuda")
        wp = (it + 1) % 2 == 0
        lr_ = T.td3_update(ref, buf.sample(B, ind=idx), wp, noise=noise, all_reduce=False)
        lf_ = fz.update(buf.data, idx, wp, noise=noise, all_reduce=False)
        torch.cuda.synchronize()
        assert abs(float(lr_) - float(lf_)) <= 1e-4 * max(1.0, abs(float(lr_)))
        rel = lambda x, y: float((x - y).abs().max() / y.abs().max().clamp_min(1e-12))
        if not wp:          # (on a policy iteration autograd's actor_loss.backward() also adds into the critic's .grad; nothing reads that)
            for pr, pf in zip(ref.critic.parameters(), fus.critic.parameters()):
                assert rel(pf.grad, pr.grad) <= 2e-4, (it, "critic grad")
        if wp:
            for pr, pf in zip(ref.actor.parameters(), fus.actor.parameters()):
                assert rel(pf.grad, pr.grad) <= 2e-4, (it, "actor grad")
        for nr, nf in ((ref.actor, fus.actor), (ref.critic, fus.critic), (ref.actor_target, fus.actor_target), (ref.critic_target, fus.critic_target)):
            for (k, vr), (_, vf) in zip(nr.state_dict().items(), nf.state_dict().items()):
                assert float((vr - vf).abs().max()) <= 2e-5, (it, k)


def test_graph_trainer_fused_and_autograd_updates_learn_alike():
    """GraphedVecTD3Trainer with fused=True (default) and fused=False on the same seeds: both run, the ring fills identically, critic
    losses are finite and of the same size after 30 iterations (the two differ in RNG consumption, so no bitwise claim)."""
    from plen_ml_walk_amd.td3 import ReplayBuffer, TD3Agent
    from plen_ml_walk_amd.train_vec import GraphedVecTD3Trainer
    n = 256
    out = {}
    for fused in (True, False):
        torch.manual_seed(0)
        env = _env(n); agent = TD3Agent(26, 18, 1.0); replay = ReplayBuffer(20000)
        tr = GraphedVecTD3Trainer(env, agent, replay, start_timesteps=512, batch_size=256, updates_per_step=1, seed=0, fused=fused)
        for _ in range(32):
            tr.step()
        torch.cuda.synchronize()
        assert tr.env_steps == 32 * n and tr.grad_steps == 31 and replay.size == 32 * n
        assert torch.isfinite(agent.last_critic_loss) and torch.isfinite(replay.data[:replay.size]).all()
        out[fused] = float(agent.last_critic_loss)
        env.close()
    assert 0.2 <= out[True] / out[False] <= 5.0


@pytest.mark.parametrize("B,N,K", [(4096, 256, 256), (4096, 512, 44), (4096, 256, 26), (4096, 18, 256), (300, 256, 256), (77, 33, 45)])
def test_mfma_weight_gradient_kernel(B, N, K):
    """k_wgrad (v_mfma_f32_32x32x2_f32, batch split, atomics) against torch: dW = dH^T X and db = column sums of dH, on column-slice operands,
    ragged tile and batch sizes included.  f32 MFMA with f32 accumulation: agreement to summation-order rounding."""
    from plen_ml_walk_amd.td3_fused import FusedTD3
    from plen_ml_walk_amd.td3 import TD3Agent
    fz = FusedTD3(TD3Agent(26, 18, 1.0, data_parallel=False))
    g = torch.Generator(device="cuda").manual_seed(B + N + K)
    dh_full = torch.randn(B, N + 7, generator=g, device="cuda"); x_full = torch.randn(B, K + 5, generator=g, device="cuda")
    dh, x = dh_full[:, 3:3 + N], x_full[:, 2:2 + K]
    gw = torch.zeros(N, K, device="cuda"); gb = torch.zeros(N, device="cuda")
    fz._wgrad(dh, x, gw, gb)
    torch.cuda.synchronize()
    want_w = dh.double().t() @ x.double(); want_b = dh.double().sum(0)
    assert float((gw.double() - want_w).abs().max()) <= 2e-5 * float(want_w.abs().max()) + 1e-4
    assert float((gb.double() - want_b).abs().max()) <= 2e-5 * float(want_b.abs().max()) + 1e-4


@pytest.mark.parametrize("n,T,batch", [(512, 45, 512), (4096, 64, 4096)])
def test_pipelined_trainer_overlaps_without_races(n, T, batch):
    """(n = 4096, batch 4096: BASELINE.json configs[2] at full size -- 2 x 2048 envs + the whole TD3 loop, 64 vector steps.)
    PipelinedVecTD3Trainer (two half batches + the fused update on three streams, event dependencies only): counters, ring contents and
    ORDER are those of the synchronous loop -- every stored transition's next_state is the state stored one vector step later for the same
    env (unless its episode ended), which a torn or misplaced row would break; losses finite.  The collection side is bitwise reproducible
    (checked with learning off); the learner's float-atomic reductions (k_wgrad, k_colsum) are order-dependent in the last bits, like any
    split-K GEMM, so a learning run is not."""
    from plen_ml_walk_amd.td3 import ReplayBuffer, TD3Agent
    from plen_ml_walk_amd.train_vec import PipelinedVecTD3Trainer
    runs = []
    for start in (3 * n, 10 ** 9, 10 ** 9):
        torch.manual_seed(0)
        envs = [_env(n // 2), _env(n // 2)]
        agent = TD3Agent(26, 18, 1.0); replay = ReplayBuffer(40 * n)        # the ring wraps during the run: the sampling guard is exercised
        tr = PipelinedVecTD3Trainer(envs, agent, replay, start_timesteps=start, batch_size=batch, seed=7)
        for _ in range(T):
            tr.step()
        tr.sync(); torch.cuda.synchronize()
        learning = start < 10 ** 9
        assert tr.env_steps == T * n and tr.grad_steps == (T - 3 if learning else 0) and agent.total_it == tr.grad_steps
        assert int(tr.total_u) == T * n and int(tr.base[0]) == T * n and int(tr.base[1]) == T * n + n // 2
        assert replay.size == 40 * n and replay.ptr == (T * n) % (40 * n)
        d = replay.data
        assert torch.isfinite(d).all() and ((d[:, 71] == 0) | (d[:, 71] == 1)).all()
        if learning:
            assert torch.isfinite(agent.last_critic_loss) and float(agent.last_critic_loss) > 0
        # continuity: rows of steps T-35..T-7 (the ring holds the last 40 vector steps; the wrap is inside the checked range at T = 64)
        ok = tot = 0
        for t in range(T - 35, T - 7):
            a, b = d[(t % 40) * n:(t % 40 + 1) * n], d[((t + 1) % 40) * n:((t + 1) % 40 + 1) * n]
            same = (a[:, 44:70] == b[:, 0:26]).all(1)
            tot += n; ok += int(same.sum())
            ended = ~same
            # where the chain is broken the episode ended: the next stored state is the reset observation (identical for every env)
            if ended.any():
                assert (b[ended, 0:26] == b[ended][0, 0:26]).all()
        assert ok / tot >= 0.9
        assert float(d[:, 26:44].abs().max()) <= 1.0 and (d[(d[:, 71] == 0), 70] < -50).all()
        # device-side episode bookkeeping: every env-step is either in a finished episode or in a running one
        st = tr.episode_stats(reset=False)
        running = sum(float(x[:, 1].sum()) for x in tr.ep_ret)
        assert st["episodes"] > 0 and abs(st["mean_length"] * st["episodes"] + running - T * n) < 0.5
        assert -400 < st["mean_return"] < 100 and 5 < st["mean_length"] <= 500
        runs.append(d.clone())
        for e in envs:
            e.close()
    assert torch.equal(runs[1], runs[2])          # collection (RNG, env steps, ring writes on two streams): bitwise reproducible
    assert not torch.equal(runs[0], runs[1])      # ... and the learning run did act with a learned policy


def test_six_step_graph_equals_single_steps():
    """PipelinedVecTD3Trainer.step_block() replays SIX vector steps of the whole loop -- both collectors and the learner, with their event
    dependencies as graph edges -- as one hipGraph.  Same kernels on the same data in the same order per stream as six step() calls: from equal seeds, 24 warm-up
    steps through step() and then 36 more through step() in one trainer and through run() (six-step graphs) in the other end with the SAME ring contents, counters,
    random-stream positions and -- the large-batch kernels sum in a fixed order -- bitwise equal parameters."""
    from plen_ml_walk_amd.td3 import ReplayBuffer, TD3Agent
    from plen_ml_walk_amd.train_vec import PipelinedVecTD3Trainer
    n, batch = 1024, 1024
    out = []
    for blocks in (False, True):
        torch.manual_seed(0)
        envs = [_env(n // 2), _env(n // 2)]
        agent = TD3Agent(26, 18, 1.0); replay = ReplayBuffer(80 * n)
        tr = PipelinedVecTD3Trainer(envs, agent, replay, start_timesteps=3 * n, batch_size=batch, seed=7)
        for _ in range(24):
            tr.step()
        if blocks:
            tr.run(36 + 2)                       # six blocks, then two single steps behind them (the hand-over back to step())
            assert sum(1 for k in tr._graphs if k[0] == "block") >= 1
        else:
            for _ in range(36 + 2):
                tr.step()
        tr.sync(); torch.cuda.synchronize()
        assert tr.env_steps == 62 * n and tr.grad_steps == 62 - 3 and agent.total_it == tr.grad_steps and int(tr.total_u) == 62 * n
        assert int(tr.base[0]) == 62 * n and int(tr.base[1]) == 62 * n + n // 2
        out.append((replay.data[:62 * n].clone(), agent._critic_flat.flat.clone(), agent._actor_flat.flat.clone(), [r.clone() for r in tr.rngs], float(agent.last_critic_loss)))
        for e in envs:
            e.close()
    a, b = out
    assert torch.equal(a[0], b[0])                                       # every stored transition, in order
    assert torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])           # critic and actor parameters, bit for bit
    assert all(torch.equal(x, y) for x, y in zip(a[3], b[3])) and a[4] == b[4]


def test_in_kernel_philox_draws():
    """The kernels' own random numbers (Philox4x32-10 + Box-Muller; no library RNG call inside the captured graphs): uniform actions and
    exploration noise have the right moments, differ between calls (the following store() bumps the call counter) and between seeds, and
    are reproducible for equal (seed, call)."""
    from plen_ml_walk_amd.td3 import TD3Agent
    from plen_ml_walk_amd.td3_fused import FusedTD3
    ag = TD3Agent(26, 18, 1.0, data_parallel=False)
    with torch.no_grad():
        ag.actor.fc3.weight.zero_(); ag.actor.fc3.bias.zero_()          # actor(state) == 0: explore() returns the clipped noise itself
    fz = FusedTD3(ag)
    n = 4096
    state = torch.zeros(n, 26, device="cuda")
    data = torch.zeros(4 * n, 72, device="cuda"); total = torch.zeros((), dtype=torch.long, device="cuda")
    rew = torch.zeros(n, device="cuda"); done = torch.zeros(n, dtype=torch.uint8, device="cuda")

    def draws(seed, calls):
        rng = FusedTD3.new_rng("cuda", seed)
        out = []
        for _ in range(calls):
            u = fz.uniform_actions(n, rng)
            e = fz.explore(state, 0.2, rng=rng)                                 # clip at 5 sigma: moments of the plain normal
            fz.store(data, total, state, u, state, rew, done, rng=rng)          # bumps the call counter
            out.append((u.clone(), e.clone()))
        assert int(rng[1]) == calls
        return out
    a, b, c = draws(11, 2), draws(11, 2), draws(12, 1)
    for (u, e) in a:
        assert abs(float(u.mean())) < 0.01 and abs(float(u.var()) - 1 / 3) < 0.01 and float(u.min()) >= -1 and float(u.max()) < 1
        assert abs(float(e.mean())) < 0.005 and abs(float(e.std()) - 0.2) < 0.004 and float(e.abs().max()) <= 1.0
        z = e / 0.2
        assert abs(float((z ** 4).mean()) - 3.0) < 0.15 and abs(float((z ** 3).mean())) < 0.05      # normal kurtosis, no skew
    assert torch.equal(a[0][0], b[0][0]) and torch.equal(a[1][1], b[1][1])    # same (seed, call): same numbers
    assert not torch.equal(a[0][0], a[1][0]) and not torch.equal(a[0][0], c[0][0])
    # no visible correlation between neighbouring elements or between the two draws of one call
    u, e = a[0]
    assert abs(float((u[:, :-1] * u[:, 1:]).mean())) < 0.01 and abs(float((u * e).mean())) < 0.01


@pytest.mark.parametrize("fused", [1, 0, "pipelined", "pipelined_block"])
def test_two_ranks_on_one_gpu_graph_trainer(tmp_path, fused):
    """The multi-rank path of the hipGraph trainer with REAL collectives on GPU tensors (two ranks share this box's one GPU, so gloo instead of
    RCCL): parameters start from rank 0's, the update runs as graph segments with the critic / actor bucket all-reduces between them, and after
    13 updates on rank-local data both ranks hold bitwise identical parameters and targets, which moved and are finite."""
    import json, socket, subprocess, sys
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / "d")
    env = dict(os.environ, PLEN_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "tests", "dist_gpu_worker.py"), out] + (["1", fused] if str(fused).startswith("pipelined") else [str(fused)])
    subprocess.run(cmd, check=True, timeout=600, env=env, cwd=ROOT)
    r0, r1 = json.load(open(out + ".rank0.json")), json.load(open(out + ".rank1.json"))
    pipe = str(fused).startswith("pipelined")
    steps = 24 if pipe else 14          # (the pipelined trainer's 6 update-graph keys need 18 steps to be captured)
    for r in (r0, r1):
        assert r["world"] == 2 and r["allreduce_mode"] == "eager-between-graphs" and r["grad_steps"] == steps - (2 if pipe else 1)
        assert r["block_pass"] == (fused == "pipelined_block")       # batch 1024: the large-batch kernels, their partial gradients reduced into the bucket for the all-reduce
        assert r["env_steps"] == steps * 256
        assert r["params_equal_across_ranks"] and r["targets_equal_across_ranks"] and r["finite"] and r["moved"] > 1e-4
        assert r["rank_local_env_states_differ"]
    assert any("critic_backward" in g for g in r0["graphs"]) and any("actor_step" in g for g in r0["graphs"])


@pytest.mark.parametrize("B", [3, 37, 101])
def test_row_block_kernels_never_write_past_a_ragged_batch(B):
    """a 4-row store group straddling B (B % 4 != 0) must not write rows >= B.  Every per-iteration scratch matrix is allocated with
    canary rows behind it (FusedTD3._alloc hook); critic rows, policy rows and actor rows run at ragged sizes; the canaries stay untouched."""
    from plen_ml_walk_amd import td3 as T
    from plen_ml_walk_amd.td3_fused import FusedTD3
    torch.manual_seed(21)
    ag = T.TD3Agent(26, 18, 1.0, data_parallel=False)
    fz = FusedTD3(ag, seed=5, rows=True)
    pads = []

    def alloc(*shape):
        full = torch.full((shape[0] + 8,) + tuple(shape[1:]), 12345.0, device="cuda", dtype=torch.float32)
        pads.append(full[shape[0]:])
        return full[:shape[0]]
    fz._alloc = alloc
    data = torch.randn(400, 72, device="cuda")
    tot = torch.tensor(300, dtype=torch.long, device="cuda")
    loss = fz.critic_backward(data, B, total=tot, guard=0)
    fz.policy_backward()
    a = fz.explore(torch.randn(B, 26, device="cuda"), 0.1, rng=FusedTD3.new_rng("cuda", 3))
    torch.cuda.synchronize()
    assert torch.isfinite(loss) and a.shape == (B, 18)
    assert len(pads) >= 15
    for k, p_ in enumerate(pads):
        assert bool((p_ == 12345.0).all()), "scratch matrix %d of %d: rows past B = %d were written" % (k, len(pads), B)


def test_episode_statistics_survive_long_runs():
    """the ring-store kernel's episode statistics are float64 (count exact to 2^53): pre-loaded beyond float32's 2^24 they still
    absorb single episodes."""
    from plen_ml_walk_amd import td3 as T
    from plen_ml_walk_amd.td3_fused import FusedTD3
    ag = T.TD3Agent(26, 18, 1.0, data_parallel=False)
    fz = FusedTD3(ag, seed=1)
    n = 64
    data = torch.zeros(1000, 72, device="cuda")
    total = torch.zeros(1, dtype=torch.long, device="cuda")
    ep_ret = torch.zeros(n, 2, device="cuda")
    stats = torch.tensor([3.0e9, float(2 ** 24 + 1), float(2 ** 31)], dtype=torch.float64, device="cuda")
    s0 = stats.clone()
    done = torch.zeros(n, dtype=torch.uint8, device="cuda"); done[:5] = 1
    rew = torch.full((n,), 0.25, device="cuda")
    fz.store(data, total, torch.zeros(n, 26, device="cuda"), torch.zeros(n, 18, device="cuda"), torch.zeros(n, 26, device="cuda"), rew, done, episodes=(ep_ret, stats))
    torch.cuda.synchronize()
    d = (stats - s0).tolist()
    assert d == [5 * 0.25, 5.0, 5.0], d


def test_rccl_world_size_one_runs_the_multi_rank_code_path(tmp_path):
    """RCCL on the hardware there is: one rank, backend nccl, device_id given; with PLEN_TD3_FORCE_COLLECTIVES=1 both
    trainers cut their update at the two gradient all-reduces (eager between graph segments) or capture them (PLEN_TD3_CAPTURE_ALLREDUCE=1).
    One rank's reduction is the identity: the synchronous graph trainer on the autograd update (deterministic library GEMMs) ends bitwise where the
    collective-free run ends; the fused update's float-atomic weight gradients are reproducible to rounding only."""
    import json, socket, subprocess, sys
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / "rccl.json")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_world1_worker.py"), out], check=True, timeout=900, env=env, cwd=ROOT)
    r = json.load(open(out))
    dst = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(dst):
        json.dump(r, open(os.path.join(dst, "r03_rccl_world1.json"), "w"), indent=1)
    assert r["backend"] == "nccl" and r["world"] == 1
    g, p, c = r["graphed"], r["pipelined"], r["graphed_captured"]
    assert g["collectives"] and g["allreduce_mode"] == "eager-between-graphs" and any("critic_backward" in k for k in g["graphs"]) and any("actor_step" in k for k in g["graphs"])
    assert g["finite"] and g["max_abs_param_diff_vs_no_collectives"] < 1e-2 and g["grad_steps"] == 13
    ga = r["graphed_autograd"]
    assert ga["collectives"] and ga["finite"] and ga["max_abs_param_diff_vs_no_collectives"] == 0.0 and ga["grad_steps"] == 13
    assert c["allreduce_mode"] == "captured" and c["finite"] and c["max_abs_param_diff_vs_no_collectives"] < 1e-2
    assert p["collectives"] and p["allreduce_mode"] == "eager-between-graphs" and p["finite"] and p["grad_steps"] == 22
    # the fused learner's float-atomic weight gradients are not bitwise reproducible, and Adam turns a flipped sign of a tiny gradient into a full
    # step: two runs can differ by up to 2 * lr * steps = 7.8e-3 (13 steps) / 1.3e-2 (22 steps) in single parameters; the exact check is the autograd run above
    assert p["max_abs_param_diff_vs_no_collectives"] < 2e-2
    lat = r["allreduce_latency_620KB"]
    assert lat["alone"]["device_us_per_call"] < 500 and lat["beside_two_resident_2048_env_launches"]["device_us_per_call"] < 5000


@pytest.mark.gpu
def test_reference_recipe_two_thousand_updates():
    """the REFERENCE'S recipe -- batch 100, ONE update per env-step (plen_td3.py:119-120), policy_freq 2 (td3.py:329-345), exploration
    N(0, 0.1) -- through the hipGraph trainer: 16 envs step together, 16 updates follow.  After the random-action phase exactly one update per env-step has
    run, the actor has moved on every second of them (target networks follow), losses are finite, and the update-to-data ratio is the reference's 1."""
    from plen_ml_walk_amd.vec_env import PlenVecEnv
    from plen_ml_walk_amd.td3 import ReplayBuffer, TD3Agent
    from plen_ml_walk_amd.train_vec import GraphedVecTD3Trainer
    torch.manual_seed(0)
    n, start = 16, 1024
    env = PlenVecEnv(n, device="cuda:0")
    agent = TD3Agent(26, 18, 1.0, device=torch.device("cuda:0"))
    replay = ReplayBuffer(100000, device=torch.device("cuda:0")); replay.seed(0)
    tr = GraphedVecTD3Trainer(env, agent, replay, start_timesteps=start, expl_noise=0.1, batch_size=100, updates_per_step=n, seed=3)
    a0 = torch.cat([p.detach().flatten().clone() for p in agent.actor.parameters()])
    c0 = torch.cat([p.detach().flatten().clone() for p in agent.critic.parameters()])
    while tr.grad_steps < 2000:
        tr.step()
    torch.cuda.synchronize()
    warm_steps = start                                      # env-steps before the first update (start_timesteps is a multiple of n)
    assert tr.grad_steps == tr.env_steps - warm_steps + n and tr.grad_steps == 2000       # the step that reaches start_timesteps already updates: one update per env-step from there
    assert agent.total_it == 2000 and replay.size == tr.env_steps
    a1 = torch.cat([p.detach().flatten() for p in agent.actor.parameters()])
    c1 = torch.cat([p.detach().flatten() for p in agent.critic.parameters()])
    at = torch.cat([p.detach().flatten() for p in agent.actor_target.parameters()])
    assert torch.isfinite(a1).all() and torch.isfinite(c1).all() and float(agent.last_critic_loss) == float(agent.last_critic_loss)
    assert float((a1 - a0).abs().max()) > 1e-3 and float((c1 - c0).abs().max()) > 1e-3
    # Polyak tau 0.005 over 1000 policy updates: the target actor has moved most of the way from the initial actor towards the current one
    assert float((at - a0).abs().max()) > 1e-4 and float((at - a1).abs().mean()) < float((a0 - a1).abs().mean())
    env.close()


@pytest.mark.gpu
@pytest.mark.parametrize("n,with_advance", [(2048, True), (37, False), (1, True)])
def test_store_with_step_equals_store_then_add(n, with_advance):
    """plentd3_store_step (the ring position advanced inside the store kernel by the last block to arrive) against plentd3_store[_advance] followed by the caller's
    `total += step`: same ring rows over several wrapping steps, same state hand-over, same `total`, same episode statistics; the block counter is back at zero after
    every launch.  Reference: plen_td3.py:109-115."""
    from plen_ml_walk_amd import td3 as T
    from plen_ml_walk_amd.td3_fused import FusedTD3
    torch.manual_seed(3)
    ag = T.TD3Agent(26, 18, 1.0, data_parallel=False)
    fz = FusedTD3(ag, seed=1)
    cap, stride = 3 * n + 5, 2 * n                          # (another collector's rows lie between this one's: the ring wraps after two steps)
    out = {}
    for mode in ("separate", "fused"):
        g = torch.Generator(device="cuda").manual_seed(11)
        data = torch.zeros(cap, 72, device="cuda")
        total = torch.tensor(7, dtype=torch.long, device="cuda")
        state = torch.randn(n, 26, device="cuda", generator=g)
        ep, st = torch.zeros(n, 2, device="cuda"), torch.zeros(3, dtype=torch.float64, device="cuda")
        ctr = torch.zeros(1, dtype=torch.int32, device="cuda")
        rng = FusedTD3.new_rng(ag.device, 5)
        for _ in range(4):
            action, nxt, rew = torch.randn(n, 18, device="cuda", generator=g), torch.randn(n, 26, device="cuda", generator=g), torch.randn(n, device="cuda", generator=g)
            done = (torch.rand(n, device="cuda", generator=g) < 0.3).to(torch.uint8) * (1 + (torch.rand(n, device="cuda", generator=g) < 0.5).to(torch.uint8))
            obs = torch.randn(n, 26, device="cuda", generator=g)
            if mode == "fused":
                fz.store(data, total, state, action, nxt, rew, done, rng=rng, episodes=(ep, st), advance=obs if with_advance else None, step=(stride, ctr))
                assert int(ctr) == 0
            else:
                fz.store(data, total, state, action, nxt, rew, done, rng=rng, episodes=(ep, st), advance=obs if with_advance else None)
                total += stride
            if not with_advance:
                state.copy_(obs)
        torch.cuda.synchronize()
        out[mode] = (data.clone(), int(total), state.clone(), ep.clone(), st.clone(), rng.clone())
    a, b = out["separate"], out["fused"]
    assert a[1] == b[1] == 7 + 4 * stride
    for x, y in zip(a, b):
        if torch.is_tensor(x):
            assert torch.equal(x, y)
    assert float(a[0].abs().sum()) > 0
