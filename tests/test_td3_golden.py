"""TD3 (T1-T4 of SURVEY.md 8a) against golden vectors captured from the reference's td3.py."""
import os
import numpy as np
import pytest
import torch
from plen_ml_walk_amd import td3 as T


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def test_parameter_counts_and_state_dict_keys():
    a = T.TD3Agent(26, 18, 1.0, device="cpu")
    assert sum(p.numel() for p in a.actor.parameters()) == 77330
    assert sum(p.numel() for p in a.critic.parameters()) == 155138
    assert sorted(a.actor.state_dict()) == sorted(["fc%d.%s" % (i, k) for i in (1, 2, 3) for k in ("weight", "bias")])
    assert sorted(a.critic.state_dict()) == sorted(["fc%d.%s" % (i, k) for i in range(1, 7) for k in ("weight", "bias")])


def test_shipped_policy_forward(golden_dir):
    """select_action / critic of the reference's checkpoint 3229999 on 64 fixed inputs."""
    g = _load(golden_dir, "td3_forward.npz")
    pol = _load(golden_dir, "policy_3229999.npz")
    a = T.TD3Agent(26, 18, 1.0, device="cpu")
    a.load_arrays(pol)
    act = np.stack([a.select_action(o) for o in g["obs"]])
    assert act.dtype == np.float32 and act.shape == (64, 18)
    # (3e-5, not 1e-5: the shipped policy's pre-activations are O(10) and the host BLAS decides the summation order -- 1.1e-5 seen on the GPU box's CPU)
    assert np.abs(act - g["action"]).max() <= 3e-5
    batch = a.select_action_batch(torch.as_tensor(g["obs"], dtype=torch.float32)).numpy()
    assert np.abs(batch - g["action"]).max() <= 3e-5
    with torch.no_grad():
        q1, q2 = a.critic(torch.as_tensor(g["obs"], dtype=torch.float32), torch.as_tensor(g["action"]))
    assert np.abs(q1.numpy() - g["q1"]).max() <= 1e-4 * max(1, np.abs(g["q1"]).max())
    assert np.abs(q2.numpy() - g["q2"]).max() <= 1e-4 * max(1, np.abs(g["q2"]).max())


def test_two_train_iterations(golden_dir, monkeypatch):
    """TD3Agent.train x2 from the reference's initial parameters, with the reference's sampled indices
    and target-policy noise fed back in; parameters afterwards match to 1e-5."""
    g = _load(golden_dir, "td3_train.npz")
    a = T.TD3Agent(26, 18, 1.0, device="cpu")
    a.load_arrays({k[len("init."):]: g[k] for k in g.files if k.startswith("init.")})
    a.actor_target.load_state_dict(a.actor.state_dict()); a.critic_target.load_state_dict(a.critic.state_dict())
    buf = T.ReplayBuffer(1000, device="cpu")
    for i in range(len(g["R"])):
        buf.add((g["S"][i], g["A"][i], g["S2"][i], np.array(g["R"][i]), np.array(g["D"][i])))
    assert len(buf.storage) == 300
    noise = [torch.as_tensor(n) for n in g["noise"]]
    it = {"k": 0}
    monkeypatch.setattr(torch, "randn_like", lambda x, *a_, **k_: noise[it["k"]])
    real_sample = buf.sample
    for k in range(2):
        it["k"] = k
        monkeypatch.setattr(buf, "sample", lambda bs, _k=k: real_sample(bs, ind=g["idx"][_k]))
        a.train(buf, int(g["batch"]))
        for net, prefix in ((a.actor, "actor."), (a.critic, "critic."), (a.actor_target, "actor_target."), (a.critic_target, "critic_target.")):
            for name, v in net.state_dict().items():
                ref_sum = g["it%d.sum.%s%s" % (k + 1, prefix, name)]
                got = v.numpy().astype(np.float64)
                assert abs(got.sum() - ref_sum[0]) <= 1e-5 * max(1.0, ref_sum[1]), (k, prefix + name)
                key = "it%d.%s%s" % (k + 1, prefix, name)
                if key in g.files:
                    assert np.abs(v.numpy() - g[key]).max() <= 1e-5, key
    assert a.total_it == 2


def test_replay_buffer_ring_semantics():
    """Grows to max_size, then overwrites from ptr=0 round the ring (reference td3.py:143-147)."""
    b = T.ReplayBuffer(5, state_dim=2, action_dim=1, device="cpu")
    for i in range(7):
        b.add((np.full(2, i), np.full(1, i), np.full(2, i + 0.5), float(i), float(i % 2)))
    assert len(b.storage) == 5 and b.ptr == 2
    assert [int(b.storage[i][0][0]) for i in range(5)] == [5, 6, 2, 3, 4]
    s, a, s2, r, nd = b.sample(64)
    assert s.shape == (64, 2) and a.shape == (64, 1) and r.shape == (64, 1) and nd.shape == (64, 1)
    assert s.dtype == torch.float32 and set(np.unique(nd.numpy())) <= {0.0, 1.0}
    assert torch.all(nd == 1 - (r % 2))
    # batched insert crossing the end of the ring
    c = T.ReplayBuffer(5, state_dim=2, action_dim=1, device="cpu")
    c.add_batch(torch.arange(8.).reshape(4, 2), torch.zeros(4, 1), torch.zeros(4, 2), torch.arange(4.), torch.zeros(4))
    c.add_batch(torch.arange(8.).reshape(4, 2) + 100, torch.zeros(4, 1), torch.zeros(4, 2), torch.arange(4.) + 10, torch.zeros(4))
    assert c.size == 5 and c.ptr == 3
    assert c.reward[:, 0].tolist() == [11.0, 12.0, 13.0, 3.0, 10.0]


def test_replay_save_load_roundtrip(tmp_path):
    b = T.ReplayBuffer(50, device="cpu"); b.buffer_path = str(tmp_path)
    rng = np.random.default_rng(0)
    for i in range(20):
        b.add((rng.normal(size=26), rng.normal(size=18), rng.normal(size=26), rng.normal(), 0.0))
    b.save(7)
    c = T.ReplayBuffer(50, device="cpu"); c.buffer_path = str(tmp_path); c.load(7)
    assert c.size == 20 and torch.equal(c.state[:20], b.state[:20]) and torch.equal(c.reward[:20], b.reward[:20])


def test_checkpoint_roundtrip_and_target_quirk(tmp_path):
    """save/load use the reference's four file suffixes; load() leaves the targets alone (td3.py:366-376)."""
    a = T.TD3Agent(26, 18, 1.0, device="cpu")
    prefix = str(tmp_path / "plen_walk_gazebo_9")
    a.save(prefix)
    for sfx in ("_critic", "_critic_optimizer", "_actor", "_actor_optimizer"):
        assert os.path.exists(prefix + sfx)
    b = T.TD3Agent(26, 18, 1.0, device="cpu")
    tgt_before = [p.clone() for p in b.actor_target.parameters()]
    b.load(prefix)
    for p, q in zip(a.actor.parameters(), b.actor.parameters()):
        assert torch.equal(p, q)
    for p, q in zip(tgt_before, b.actor_target.parameters()):
        assert torch.equal(p, q)


def test_reference_training_log_numbers_quoted_in_the_docs(golden_dir):
    """DESIGN.md / the GPU tests quote the reference's own PyBullet training run (results/plen_walk_gazebo_.npy): 24 832 episodes, last
    1000 averaging +50, best single episode +328.  Pinned here against a summary of that data file."""
    import os
    g = np.load(os.path.join(golden_dir, "ref_training_log_summary.npz"))
    assert int(g["episodes"]) == 24832
    assert abs(float(g["last1000_mean"]) - 50.43) < 0.01 and abs(float(g["max_return"]) - 328.04) < 0.01
    assert float(g["first100_mean"]) < -150 and g["block_means_1000"].shape == (24,)
