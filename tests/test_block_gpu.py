"""The large-batch shape of the TD3 update (csrc/td3_block.hip: 16 batch rows per 256-thread workgroup, transposed products on the matrix cores, activations
in LDS, weights pre-packed in operand order) against the layer-by-layer path (library GEMMs + one kernel per step), the reference's golden iterations,
and itself (same bits every run).  Reference: plen_ros/src/plen_ros_helpers/td3.py:259-356."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _agent(seed):
    from plen_ml_walk_amd import td3 as T
    torch.manual_seed(seed)
    ag = T.TD3Agent(26, 18, 1.0, data_parallel=False)
    for net in (ag.actor, ag.actor_target, ag.critic_target, ag.critic):            # away from the all-equal initial targets
        for p_ in net.parameters():
            p_.data.add_(0.05 * torch.randn_like(p_))
    return ag


def test_pack_kernel_writes_matrix_core_operand_order():
    """plentd3_pack against its definition (include/plentd3.h): dst float4 ((t KS + s) 64 + lane) = M[16 t + lane % 16][16 s + 4 (lane / 16) + (0..3)], zero-padded,
    for a plain matrix, a transpose and a column block of a transpose (the three forms the passes use)."""
    from plen_ml_walk_amd.td3_fused import FusedTD3
    ag = _agent(1)
    fz = FusedTD3(ag, seed=1)
    w = torch.randn(18, 256, device="cuda")
    w14 = torch.randn(512, 44, device="cuda")
    fz._pack([fz._nt("a", w), fz._tr("b", w), ("c", w14, 18, 256, 1, 44, 26), fz._nt("d", w14)])
    torch.cuda.synchronize()

    def ref(M):
        N, K = M.shape
        T, KS = (N + 15) // 16, (K + 15) // 16
        P = torch.zeros(16 * T, 16 * KS, device="cuda")
        P[:N, :K] = M
        # [t][r][s][g][v] -> [t][s][g][r][v]
        return P.view(T, 16, KS, 4, 4).permute(0, 2, 3, 1, 4).contiguous().view(-1)
    assert torch.equal(fz._packs["a"], ref(w))
    assert torch.equal(fz._packs["b"], ref(w.t()))
    assert torch.equal(fz._packs["c"], ref(w14[:256, 26:44].t()))
    assert torch.equal(fz._packs["d"], ref(w14))


@pytest.mark.parametrize("B,total", [(4096, 50000), (1000, 700), (16, 40), (37, 900), (3, 5)])
def test_large_batch_block_kernels_equal_the_layer_by_layer_update(B, total):
    """k_critic_block / k_policy_block + the weight-gradient launches against the layer-by-layer path from the same random state: the benchmark's batch 4096
    (one workgroup per compute unit), a batch that is not a multiple of 16 (ragged last block), one block exactly, a ragged small batch, fewer rows than a
    quarter block.  Same sampled rows, same smoothing noise; loss, every critic gradient and every actor gradient agree to f32 summation order; the call
    counter advances once; the row-local outputs of two runs are bitwise equal (no sum inside the pass kernels depends on timing)."""
    from plen_ml_walk_amd import td3 as T
    from plen_ml_walk_amd.td3_fused import FusedTD3
    ag = _agent(21)
    fz = FusedTD3(ag, seed=9, team=False, rows=False)
    data = torch.randn(1000, 72, device="cuda")
    data[:, 70] = torch.rand(1000, device="cuda")
    data[:, 71] = (torch.rand(1000, device="cuda") > 0.1).float()
    tot = torch.tensor(total, dtype=torch.long, device="cuda")
    rng0 = fz.rng.clone()
    res, keep = {}, {}
    for name, block in (("layers", False), ("block", True), ("block again", True)):
        fz.rng.copy_(rng0)
        fz.block = block
        ag._critic_grads.zero(); ag._actor_grads.zero(); fz._zeroed = {}
        made = []
        fz._alloc = lambda *shape: (made.append(torch.full(shape, float("nan"), device="cuda")) or made[-1])
        loss = fz.critic_backward(data, B, total=tot, guard=64).clone()
        assert fz._block_pass == block
        fz.policy_backward()
        torch.cuda.synchronize()
        assert int(fz.rng[1]) == int(rng0[1]) + 1
        res[name] = (loss, ag._critic_grads.flat.clone(), ag._actor_grads.flat.clone(), fz._saved[0].clone(), fz._saved[1].clone())
        keep[name] = [t.clone() for t in made]
    fz._alloc = None
    assert int(fz._done_count) == 0
    la, ca, aa, sa_, pa = res["layers"]
    lb, cb, ab, sb, pb = res["block"]
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
    # the pass kernels' own outputs (activations and row-local gradients left for the weight-gradient kernels, the loss): same bits every run
    assert torch.equal(res["block"][0], res["block again"][0])
    for x, y in zip(keep["block"], keep["block again"]):
        assert torch.equal(torch.nan_to_num(x, nan=-7.0), torch.nan_to_num(y, nan=-7.0))


def test_block_kernels_do_not_write_past_a_ragged_batch():
    """Canary rows behind every per-iteration matrix: a batch of 37 (two full blocks + 5 rows) leaves them untouched, and every row inside is written."""
    from plen_ml_walk_amd.td3_fused import FusedTD3
    ag = _agent(5)
    fz = FusedTD3(ag, seed=2, team=False, rows=False, block=True)
    data = torch.randn(300, 72, device="cuda")
    data[:, 71] = 1.0
    tot = torch.tensor(300, dtype=torch.long, device="cuda")
    made = []

    def alloc(*shape):
        full = torch.full((shape[0] + 16,) + tuple(shape[1:]), 1234.5, device="cuda")
        made.append(full)
        return full[:shape[0]]
    fz._alloc = alloc
    B = 37
    fz.critic_backward(data, B, total=tot)
    fz.policy_backward()
    torch.cuda.synchronize()
    assert fz._block_pass and len(made) > 10
    for full in made:
        assert bool((full[-16:] == 1234.5).all()), tuple(full.shape)


def test_reference_golden_iterations_through_the_large_batch_kernels(golden_dir):
    """The reference's own two recorded train() iterations (td3.py:259-356: sampled indices and smoothing noise captured from the reference,
    tests/golden/td3_train.npz) replayed through k_critic_block / k_policy_block (explicit idx / noise): parameters and targets after iteration 1 (critic
    only) and iteration 2 (critic, actor, Polyak) match the reference's to the tolerance the autograd path is held to."""
    from test_robustness_gpu import _golden_agent, _check_against_golden
    from plen_ml_walk_amd.td3_fused import FusedTD3
    g, a, buf = _golden_agent(golden_dir, "cuda")
    fz = FusedTD3(a, team=False, rows=False, block=True)
    fz.enable_flat_adam()
    for k in range(2):
        idx = torch.as_tensor(g["idx"][k]).cuda().long().contiguous()
        noise = torch.as_tensor(g["noise"][k]).cuda().contiguous()
        a.total_it += 1
        loss = fz.update(buf.data, idx, with_policy=a.total_it % a.policy_freq == 0, noise=noise, all_reduce=False)
        torch.cuda.synchronize()
        assert fz._block_pass and torch.isfinite(loss)
        _check_against_golden(g, a, k)


@pytest.mark.parametrize("n", [2048, 37, 16])
def test_collect_phase_actor_forward_block_kernel_equals_the_row_kernel(n):
    """plentd3_actor_block (16 envs per workgroup, packed weights, activations in LDS) against plentd3_actor_rows on the same states, the same acting network
    and the same random state: the same exploration noise is drawn (noise index = element index), the actions agree to f32 summation order; repacking
    happens exactly when the acting network has changed."""
    from plen_ml_walk_amd.td3_fused import FusedTD3
    ag = _agent(8)
    state = torch.randn(n, 26, device="cuda") * torch.tensor([1.0] * 18 + [0.1, 0.3, 0.3, 0.3, 0.3, 0.3, 1.0, 1.0], device="cuda")
    out = {}
    for block in (False, True):
        fz = FusedTD3(ag, seed=4, rows=True, block=block)
        rng = FusedTD3.new_rng(ag.device, 77)
        out[block] = fz.explore(state, 0.1, rng=rng).clone()
    torch.cuda.synchronize()
    assert float((out[False] - out[True]).abs().max()) <= 2e-5 and float(out[True].abs().max()) <= 1.0 and float(out[True].std()) > 0.01
    # packed=True trusts the caller: stale packs after a raw change of the network, fresh ones after pack_actor() -- and without the promise explore() repacks itself
    fz = FusedTD3(ag, seed=4, rows=True, block=True)
    rng = FusedTD3.new_rng(ag.device, 77)
    a0 = fz.explore(state, 0.0, rng=rng).clone()
    with torch.no_grad():
        ag._actor_flat.flat.mul_(0.5)
    stale = fz.explore(state, 0.0, rng=rng, packed=True).clone()
    fresh = fz.explore(state, 0.0, rng=rng).clone()
    fz.pack_actor(ag.actor)
    again = fz.explore(state, 0.0, rng=rng, packed=True).clone()
    torch.cuda.synchronize()
    assert float((a0 - fresh).abs().max()) > 1e-3 and float((stale - fresh).abs().max()) > 1e-3 and torch.equal(fresh, again)       # (stale: old matrices, new biases)
    with torch.no_grad():
        ref = ag.max_action * torch.tanh(ag.actor.fc3(torch.relu(ag.actor.fc2(torch.relu(ag.actor.fc1(state))))))
    assert float((fresh - ref).abs().max()) <= 2e-5


@pytest.mark.parametrize("shape,B", [("block", 4096), ("block", 1000), ("team", 100), ("team", 256)])
def test_loss_partials_reach_the_last_workgroup_every_time(shape, B):
    """The workgroups' loss / head-bias partials are handed to the last workgroup WITHOUT a release fence (csrc/td3_kernels.hip: handoff_last -- write-through
    stores, waited for, then the count): 400 launches back to back, no host synchronisation in between, each from the same random state -- loss and both head-bias
    gradients come out with the same bits every time and the counter is back at zero.  (A partial read before it arrived would show as a different sum.)"""
    from plen_ml_walk_amd.td3_fused import FusedTD3
    ag = _agent(33)
    fz = FusedTD3(ag, seed=4, team=(shape == "team"), rows=False, block=(shape == "block"))
    data = torch.randn(5000, 72, device="cuda")
    data[:, 70] = torch.rand(5000, device="cuda")
    data[:, 71] = (torch.rand(5000, device="cuda") > 0.1).float()
    tot = torch.tensor(5000, dtype=torch.long, device="cuda")
    rng0 = fz.rng.clone()
    cr = ag.critic
    out = torch.empty(400, 3, device="cuda")
    for i in range(400):
        fz.rng.copy_(rng0)
        ag._critic_grads.zero(); fz._zeroed = {}
        loss = fz.critic_backward(data, B, total=tot)
        out[i, 0], out[i, 1], out[i, 2] = loss, cr.fc3.bias.grad[0], cr.fc6.bias.grad[0]
    torch.cuda.synchronize()
    assert (fz._block_pass, fz._team_pass) == (shape == "block", shape == "team")
    assert int(fz._done_count) == 0
    assert bool(torch.isfinite(out).all()) and float(out[0, 0]) > 0
    assert bool((out == out[0]).all()), out[(out != out[0]).any(dim=1)][:4]
