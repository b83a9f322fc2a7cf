"""One TD3 iteration (reference plen_ros/src/plen_ros_helpers/td3.py:259-356) with a hand-derived backward pass: library GEMMs for
the dense layers (torch.mm / addmm -> rocBLAS / hipBLASLt on the MI355X) and the hand-written HIP kernels of csrc/td3_kernels.hip
(C ABI include/plentd3.h, bound here with ctypes) for everything between them.

Why: the autograd formulation (td3.td3_update) is ~170 small kernels per iteration -- clamps, adds, fills, reductions, index ops, ~5 us
each -- and spends more GPU time in them than in its GEMMs.  This path issues 14 GEMMs + 12 fused kernels for a critic iteration and
26 + 22 for a critic + actor + targets iteration, on the SAME parameters, gradient buckets and optimisers as the autograd path
(TD3Agent keeps each network's parameters and gradients in one flat buffer; the twin critics' first layers are adjacent in it, so both
Q networks' first layers are one GEMM).  tests/test_robustness_gpu.py checks it against the reference's golden iteration and against
td3.td3_update.  GPU only: there is no CPU build of the kernels (the autograd path is the CPU / gloo implementation)."""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PLENTD3_LIB") or os.path.join(_HERE, "csrc", "libplentd3.so")          # (PLENTD3_LIB: an A/B build, scripts/ only)
EXPORTS = ["plentd3_gather", "plentd3_sample_gather", "plentd3_explore", "plentd3_uniform_actions", "plentd3_store", "plentd3_store_advance", "plentd3_store_step", "plentd3_target_action", "plentd3_q_heads", "plentd3_dh2", "plentd3_relu_mask", "plentd3_colsum", "plentd3_wgrad",
           "plentd3_tanh_out", "plentd3_dtanh", "plentd3_bias_relu", "plentd3_polyak", "plentd3_adam", "plentd3_critic_rows", "plentd3_policy_rows", "plentd3_actor_rows", "plentd3_critic_team", "plentd3_policy_team",
           "plentd3_wgrad_group", "plentd3_wgrad_adam_group", "plentd3_pack", "plentd3_critic_block", "plentd3_policy_block", "plentd3_wgrad_big", "plentd3_adam_big", "plentd3_actor_block", "plentd3_dev_mfma_spin", "plentd3_stamp", "plentd3_version"]
ROW, S, A, SA, H = 72, 26, 18, 44, 256
_lib = None


class PlenTd3Error(RuntimeError):
    pass


class CriticRowsArgs(C.Structure):
    """Mirror of PlenTd3CriticRows (include/plentd3.h)."""
    _fields_ = ([("data", C.c_void_p), ("rng", C.c_void_p), ("total", C.c_void_p), ("capacity", C.c_int64), ("guard", C.c_int64)]
                + [(n, C.c_void_p) for n in ("at_w1", "at_b1", "at_w2", "at_b2", "at_w3", "at_b3",
                                             "ct_w14", "ct_b14", "ct_w2", "ct_b2", "ct_w5", "ct_b5", "ct_w3", "ct_b3", "ct_w6", "ct_b6",
                                             "c_w14", "c_b14", "c_w2", "c_b2", "c_w5", "c_b5", "c_w3", "c_b3", "c_w6", "c_b6",
                                             "batch", "sa_pi", "t0", "t1", "sa2", "c1", "c2", "dh2", "dh1", "dq", "loss", "db3a", "db3b", "done_count", "rng_bump")]
                + [("sigma", C.c_float), ("clip", C.c_float), ("max_a", C.c_float), ("gamma", C.c_float), ("B", C.c_int), ("idx", C.c_void_p), ("noise", C.c_void_p), ("adam_step", C.c_void_p)]
                + [(n, C.c_void_p) for n in ("tp_at_w1", "tp_at_w2", "tp_at_w3", "tp_ct_w14", "tp_ct_w2", "tp_ct_w5", "tp_c_w14", "tp_c_w2", "tp_c_w5")])


class PolicyRowsArgs(C.Structure):
    """Mirror of PlenTd3PolicyRows (include/plentd3.h)."""
    _fields_ = ([(n, C.c_void_p) for n in ("a_w1", "a_b1", "a_w2", "a_b2", "a_w3", "a_b3", "c_w1", "c_b1", "c_w2", "c_b2", "c_w3",
                                           "sa_pi", "a_pi", "p1", "p2", "g1", "dg2", "dg1", "dz", "dp2", "dp1")]
                + [("max_a", C.c_float), ("B", C.c_int), ("adam_step", C.c_void_p), ("done_count", C.c_void_p)]
                + [(n, C.c_void_p) for n in ("tp_a_w1", "tp_a_w2", "tp_a_w3", "tp_c_w14", "tp_c_w2")])


class ActorRowsArgs(C.Structure):
    """Mirror of PlenTd3ActorRows (include/plentd3.h)."""
    _fields_ = ([(n, C.c_void_p) for n in ("a_w1", "a_b1", "a_w2", "a_b2", "a_w3", "a_b3", "state", "rng", "p1", "p2", "action")]
                + [("sigma", C.c_float), ("max_a", C.c_float), ("B", C.c_int)])


class WgradJob(C.Structure):
    """Mirror of PlenTd3WgradJob (include/plentd3.h)."""
    _fields_ = ([("dH", C.c_void_p), ("X", C.c_void_p), ("dW", C.c_void_p), ("db", C.c_void_p)] + [(n, C.c_int) for n in ("ds", "xs", "dws", "N", "K", "tile0")]
                + [("pack", C.c_void_p), ("pack_t", C.c_void_p), ("pack_ns", C.c_int)])


WGRAD_JOBS = 6


class WgradGroup(C.Structure):
    """Mirror of PlenTd3WgradGroup (include/plentd3.h)."""
    _fields_ = [("job", WgradJob * WGRAD_JOBS), ("n_jobs", C.c_int), ("B", C.c_int)]


ADAM_EXTRAS = 4


class AdamFusedArgs(C.Structure):
    """Mirror of PlenTd3AdamFused (include/plentd3.h)."""
    _fields_ = ([(n, C.c_void_p) for n in ("p", "g", "m", "v", "step", "target", "done_count")] + [("lr", C.c_double), ("beta1", C.c_double), ("beta2", C.c_double),
                ("eps", C.c_float), ("tau", C.c_float), ("n", C.c_int), ("n_extra", C.c_int), ("extra_off", C.c_int * ADAM_EXTRAS), ("step_advanced", C.c_int)])


PACK_JOBS = 16


class PackJob(C.Structure):
    """Mirror of PlenTd3PackJob (include/plentd3.h)."""
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p)] + [(n, C.c_int) for n in ("rs", "cs", "N", "K", "f4_0", "team")]


class PackGroup(C.Structure):
    """Mirror of PlenTd3PackGroup (include/plentd3.h)."""
    _fields_ = [("job", PackJob * PACK_JOBS), ("n_jobs", C.c_int)]


class CriticBlockArgs(C.Structure):
    """Mirror of PlenTd3CriticBlock (include/plentd3.h)."""
    _fields_ = [("rows", CriticRowsArgs)] + [(n, C.c_void_p) for n in ("p_at_w1", "p_at_w2", "p_at_w3", "p_ct_w14", "p_ct_w2", "p_ct_w5", "p_c_w14", "p_c_w2", "p_c_w5",
                                                                        "p_c_w2t", "p_c_w5t", "partials")]


class PolicyBlockArgs(C.Structure):
    """Mirror of PlenTd3PolicyBlock (include/plentd3.h)."""
    _fields_ = [("rows", PolicyRowsArgs)] + [(n, C.c_void_p) for n in ("p_a_w1", "p_a_w2", "p_a_w3", "p_c_w14", "p_c_w2", "p_c_w2t", "p_c_w1ta", "p_a_w3t", "p_a_w2t")]


WGRAD_BIG_JOBS = 8
WGRAD_BIG_CHUNKS = 8          # batch chunks of the large-batch weight gradients (512 rows each at batch 4096, a quarter per wave: ~660 workgroups per critic pass)


class WgradBigJob(C.Structure):
    """Mirror of PlenTd3WgradBigJob (include/plentd3.h)."""
    _fields_ = [("dH", C.c_void_p), ("X", C.c_void_p)] + [(n, C.c_int) for n in ("ds", "xs", "N", "K", "goff", "boff", "kind", "wg0")]


class WgradBig(C.Structure):
    """Mirror of PlenTd3WgradBig (include/plentd3.h)."""
    _fields_ = [("job", WgradBigJob * WGRAD_BIG_JOBS)] + [(n, C.c_int) for n in ("n_jobs", "B", "chunks", "rows_per_chunk", "stride")] + [("partial", C.c_void_p)]


class ActorBlockArgs(C.Structure):
    """Mirror of PlenTd3ActorBlock (include/plentd3.h)."""
    _fields_ = [("rows", ActorRowsArgs)] + [(n, C.c_void_p) for n in ("p_a_w1", "p_a_w2", "p_a_w3")]


TEAM_MAX_BATCH = 512          # FusedTD3(team=None): batches up to this size take the small-batch kernels (csrc/td3_team.hip)


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise PlenTd3Error("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` (hipcc --offload-arch=gfx950)" % LIB_PATH)
        lib = C.CDLL(LIB_PATH)
        for name in EXPORTS:
            getattr(lib, name)
        lib.plentd3_version.restype = C.c_char_p
        vp, i, f = C.c_void_p, C.c_int, C.c_float
        lib.plentd3_gather.argtypes = [vp, vp, vp, vp, vp, i, vp]
        lib.plentd3_sample_gather.argtypes = [vp, vp, vp, vp, C.c_int64, C.c_int64, vp, vp, vp, vp, i, vp]
        lib.plentd3_explore.argtypes = [vp, vp, vp, vp, f, f, i, vp]
        lib.plentd3_uniform_actions.argtypes = [vp, vp, i, vp]
        lib.plentd3_store.argtypes = [vp, vp, C.c_int64, vp, vp, vp, vp, vp, vp, vp, vp, i, vp]
        lib.plentd3_store_advance.argtypes = [vp, vp, C.c_int64, vp, vp, vp, vp, vp, vp, vp, vp, i, vp, vp]
        lib.plentd3_store_step.argtypes = [vp, vp, C.c_int64, vp, vp, vp, vp, vp, vp, vp, vp, i, vp, C.c_int64, vp, vp]
        lib.plentd3_target_action.argtypes = [vp, vp, vp, vp, vp, f, f, f, i, vp]
        lib.plentd3_q_heads.argtypes = [vp] * 12 + [f, i, i, vp]
        lib.plentd3_dh2.argtypes = [vp, vp, vp, vp, vp, i, i, i, vp]
        lib.plentd3_relu_mask.argtypes = [vp, vp, i, i, i, vp]
        lib.plentd3_colsum.argtypes = [vp, i, vp, i, vp, i, i, i, vp]
        lib.plentd3_wgrad.argtypes = [vp, i, vp, i, vp, i, vp, i, i, i, i, vp]
        lib.plentd3_tanh_out.argtypes = [vp, vp, vp, f, i, vp]
        lib.plentd3_dtanh.argtypes = [vp, vp, vp, f, i, vp]
        lib.plentd3_bias_relu.argtypes = [vp, vp, i, i, vp]
        lib.plentd3_polyak.argtypes = [vp, vp, f, i, vp]
        lib.plentd3_stamp.argtypes = [vp, vp, C.c_int64, i, i, i, vp]
        lib.plentd3_critic_rows.argtypes = [C.POINTER(CriticRowsArgs), vp]
        lib.plentd3_policy_rows.argtypes = [C.POINTER(PolicyRowsArgs), vp]
        lib.plentd3_actor_rows.argtypes = [C.POINTER(ActorRowsArgs), vp]
        lib.plentd3_critic_team.argtypes = [C.POINTER(CriticRowsArgs), vp]
        lib.plentd3_policy_team.argtypes = [C.POINTER(PolicyRowsArgs), vp]
        lib.plentd3_pack.argtypes = [C.POINTER(PackGroup), vp]
        lib.plentd3_critic_block.argtypes = [C.POINTER(CriticBlockArgs), vp]
        lib.plentd3_policy_block.argtypes = [C.POINTER(PolicyBlockArgs), vp]
        lib.plentd3_dev_mfma_spin.argtypes = [i, i, vp, vp]
        lib.plentd3_actor_block.argtypes = [C.POINTER(ActorBlockArgs), vp]
        lib.plentd3_wgrad_big.argtypes = [C.POINTER(WgradBig), vp]
        lib.plentd3_adam_big.argtypes = [vp, vp, vp, vp, vp, vp, i, C.c_double, C.c_double, C.c_double, f, vp, f, vp, vp, i, i, i, vp]
        lib.plentd3_wgrad_group.argtypes = [C.POINTER(WgradGroup), vp]
        lib.plentd3_wgrad_adam_group.argtypes = [C.POINTER(WgradGroup), C.POINTER(AdamFusedArgs), vp]
        lib.plentd3_adam.argtypes = [vp, vp, vp, vp, vp, vp, i, C.c_double, C.c_double, C.c_double, f, i, vp, f, vp, vp]
        _lib = lib
    return _lib


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _chk(rc):
    if rc != 0:
        raise PlenTd3Error("libplentd3 kernel launch failed: HIP error %d" % -rc)


class FlatAdam(object):
    """optimizer.step() of a torch.optim.Adam (td3.py:236-247: no weight decay, no amsgrad) as ONE kernel over the network's flat parameter /
    gradient buffers (plentd3_adam) instead of torch's two multi-tensor launches.  The moments live in flat buffers of the same layout; the
    optimizer's own state entries (exp_avg, exp_avg_sq, step) are re-pointed at views of them, so optimizer.state_dict() / checkpoints keep
    working, and optimizer.step is replaced by this step so that no code path can apply torch's update to the shared step counter."""

    def __init__(self, lib, optimizer, module, flat_params, flat_grads):
        from .td3 import _flat_order
        g = optimizer.param_groups[0]
        assert len(optimizer.param_groups) == 1 and g["weight_decay"] == 0 and not g.get("amsgrad", False) and not g.get("maximize", False)
        self.lib, self.opt, self.p, self.g = lib, optimizer, flat_params.flat, flat_grads.flat
        dev, n = self.p.device, self.p.numel()
        self.m, self.v = torch.zeros(n, device=dev), torch.zeros(n, device=dev)
        self.step_t = torch.zeros((), device=dev, dtype=torch.float32)
        self.done = torch.zeros(1, device=dev, dtype=torch.int32)
        self.params = _flat_order(module)
        self.epoch = 0           # steps taken by THIS object's own kernels (FusedTD3 compares it to know whether its packed copies of the weights are current)
        self.bind()
        optimizer.step = lambda closure=None: self.step()

    def bind(self):
        """Take over whatever state the optimiser holds (a fresh one: nothing; a resumed one / after load_state_dict: moments and step count are
        copied into the flat buffers) and point its state entries at views of the flat buffers."""
        off, steps = 0, set()
        for q in self.params:
            k = q.numel()
            mv, vv = self.m[off:off + k].view_as(q), self.v[off:off + k].view_as(q)
            st = self.opt.state.get(q)
            if st and st["exp_avg"].data_ptr() != mv.data_ptr():
                mv.copy_(st["exp_avg"]); vv.copy_(st["exp_avg_sq"]); steps.add(float(st["step"]))
            self.opt.state[q] = {"step": self.step_t, "exp_avg": mv, "exp_avg_sq": vv}
            off += k
        assert off == self.p.numel() and len(steps) <= 1
        if steps:
            self.step_t.fill_(steps.pop())

    def ensure_bound(self):
        """Re-bind if optimizer.load_state_dict() (a resumed run) has replaced the state tensors since the last step.  Must run BEFORE a pass kernel that
        advances step_t (PlenTd3CriticRows.adam_step): bind() refills step_t from the loaded state, and a refill after the kernel's increment would
        leave the counter one behind for good."""
        st0 = self.opt.state.get(self.params[0])
        if (not st0 or st0["exp_avg"].data_ptr() != self.m.data_ptr()) and not torch.cuda.is_current_stream_capturing():
            self.bind()

    def fused_args(self, target=None, tau=0.0, extras=(), step_advanced=False):
        """PlenTd3AdamFused for plentd3_wgrad_adam_group: this optimiser's step taken inside the weight-gradient kernel (hyper-parameters as they are now)."""
        if not step_advanced:        # (with step_advanced the pass kernel has already counted this step: ensure_bound() ran before it was launched)
            self.ensure_bound()
        g = self.opt.param_groups[0]
        a = AdamFusedArgs()
        a.p, a.g, a.m, a.v, a.step, a.done_count = (t.data_ptr() for t in (self.p, self.g, self.m, self.v, self.step_t, self.done))
        a.target = target.data_ptr() if target is not None else None
        a.lr, a.beta1, a.beta2, a.eps, a.tau = float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]), float(tau)
        a.n, a.n_extra, a.step_advanced = self.p.numel(), len(extras), int(bool(step_advanced))
        for k, e in enumerate(extras):
            a.extra_off[k] = int(e)
        return a

    def step(self, zero_grad=False, target=None, tau=0.0, copy_out=None):
        st0 = self.opt.state.get(self.params[0])
        if (not st0 or st0["exp_avg"].data_ptr() != self.m.data_ptr()) and not torch.cuda.is_current_stream_capturing():
            self.bind()              # optimizer.load_state_dict() replaced the state tensors: carry them over (a replayed graph never gets here:
                                     # load checkpoints before the trainers capture their graphs)
        st = C.c_void_p(torch.cuda.current_stream(self.p.device).cuda_stream)
        self.epoch += 1
        g = self.opt.param_groups[0]                       # hyper-parameters as they are now (a captured graph keeps the values of its capture)
        _chk(self.lib.plentd3_adam(_p(self.p), _p(self.g), _p(self.m), _p(self.v), _p(self.step_t), _p(self.done), self.p.numel(), float(g["lr"]), float(g["betas"][0]),
                                   float(g["betas"][1]), float(g["eps"]), int(zero_grad), _p(target), float(tau), _p(copy_out), st))


    def step_big(self, partial, chunks, stride, target=None, tau=0.0, copy_out=None, reduce_only=False):
        """step() on g = bucket + the per-chunk partial gradients of plentd3_wgrad_big (added in chunk order); leaves the bucket zero.  reduce_only: only
        bucket = that sum (several ranks: all-reduce it, then step())."""
        if not reduce_only:
            self.ensure_bound()
            self.epoch += 1
        st = C.c_void_p(torch.cuda.current_stream(self.p.device).cuda_stream)
        g = self.opt.param_groups[0]
        _chk(self.lib.plentd3_adam_big(_p(self.p), _p(self.g), _p(self.m), _p(self.v), _p(self.step_t), _p(self.done), self.p.numel(), float(g["lr"]), float(g["betas"][0]),
                                       float(g["betas"][1]), float(g["eps"]), _p(target), float(tau), _p(copy_out), _p(partial), int(chunks), int(stride), int(bool(reduce_only)), st))


class FusedTD3(object):
    """update(data, idx, with_policy) == td3.td3_update(agent, (data rows idx split into s, a, s2, r, not_done), with_policy)."""

    def __init__(self, agent, seed=0, rows=None, team=None, block=None):
        if agent.device.type != "cuda":
            raise PlenTd3Error("FusedTD3 needs the agent on a HIP device")
        self.agent = agent
        # counter-based RNG state of the update's draws (replay indices, target-smoothing noise): {seed, calls so far} on the device
        self.rng = self.new_rng(agent.device, seed)
        self.lib = load()
        self.dev = agent.device
        # does the GEMM library fuse bias + ReLU into the epilogue here?  (hipBLASLt: yes; verified numerically once)
        self.probe = None
        # row-block kernels (csrc/td3_rows.hip) + single-wave weight-gradient workgroups: what the update needs when it shares the chip with resident env
        # launches (PipelinedVecTD3Trainer: 0.73 -> 0.66 ms per step); on an otherwise idle GPU the library GEMMs are faster (critic pass 239 vs 315 us)
        self.rows = (os.environ.get("PLEN_TD3_ROWS", "0") == "1") if rows is None else bool(rows)
        # small batches (the reference's batch 100 with one update per env-step: a chain of dependent updates, so latency is what counts): the same
        # row-local passes with a TEAM of 8 waves per 4 batch rows (csrc/td3_team.hip) and all weight gradients of a pass in one launch
        # (plentd3_wgrad_group): 2 + 1 (Adam) launches per critic update instead of ~35.  None: chosen per call, batch <= TEAM_MAX_BATCH.
        self.team = (None if "PLEN_TD3_TEAM" not in os.environ else os.environ["PLEN_TD3_TEAM"] == "1") if team is None else bool(team)
        self._team_pass = False      # did the last critic pass take the team kernels (policy_backward follows it)
        # large batches (the benchmark's 4096): 16 batch rows per workgroup -- one per compute unit at batch 4096 --, its waves (critic pass and actor forward: eight, policy pass: four)
        # split every layer's output features, activations stay in LDS, weights are read pre-packed in matrix-core operand order (csrc/td3_block.hip).  None: chosen per
        # call, batch > TEAM_MAX_BATCH; the weight gradients stay what `rows` says (single-wave or 4-wave workgroups).
        self.block = (None if "PLEN_TD3_BLOCK" not in os.environ else os.environ["PLEN_TD3_BLOCK"] == "1") if block is None else bool(block)
        self._block_pass = False
        self._packs = {}             # name -> packed copy of a weight matrix (plentd3_pack), rewritten before every pass that reads it
        # small-batch kernels (round 6): the forward products' twelve weight matrices in the team kernels' operand order, packed ONCE (plentd3_pack, team layout) and then
        # kept current by the fused weight-gradient + Adam launch itself; re-packed only when something else has touched the parameters (_team_pack_sync)
        self._team_packed = os.environ.get("PLEN_TD3_TEAM_PACKED", "1") == "1"
        self._tpacks, self._tp_state, self._tp_epoch, self._tp_by_grad = {}, None, 0, {}
        self._tp_live = False        # the current iteration's team passes read the packed copies (so its fused Adam steps must write them)
        self._actor_packs = {}       # id(acting network) -> its packed matrices' names and the parameter versions they were made from
        self._partials = None
        self._big = {}               # "critic" / "actor" -> (partial gradients [chunks][stride], stride) of plentd3_wgrad_big
        self._big_pending = {}       # network -> chunks whose partial gradients still wait for their optimiser step (update() takes it with plentd3_adam_big)
        self._eager = {}             # batch size -> the small-batch iteration's persistent scratch tensors and argument blocks (_update_team_eager)
        self.eager_cache = os.environ.get("PLEN_TD3_EAGER_CACHE", "1") == "1"
        # one rank, small batch, flat Adam: the optimiser step is taken inside the grouped weight-gradient kernel (plentd3_wgrad_adam_group).
        # update() sets _fuse = {"critic": target-or-None, "actor": target} for the passes it is about to run and reads _fused_done back.
        self.fuse_adam = os.environ.get("PLEN_TD3_FUSE_ADAM", "1") == "1"
        self._fuse, self._fused_done = None, set()
        self._done_count = None
        self._policy_done = None
        self._alloc = None           # test hook: allocator of the per-iteration scratch matrices (tests put canary rows behind them)
        self._critic_adam = self._actor_adam = None
        self._zeroed = {}
        self.epilogue = False
        try:
            x = torch.randn(8, 12, device=self.dev); w = torch.randn(5, 12, device=self.dev); b = torch.randn(5, device=self.dev)
            y = torch._addmm_activation(b, x, w.t())
            self.epilogue = bool(torch.allclose(y, torch.relu(torch.addmm(b, x, w.t())), atol=1e-5))
        except Exception:
            self.epilogue = False

    def _wgrad(self, dh, x, gw, gb):
        """gw += dh^T x, gb += column sums of dh (gw, gb: zeroed views of a flat gradient bucket); dh, x may be column slices."""
        B, N = dh.shape
        K = x.shape[1]
        assert dh.stride(1) == 1 and x.stride(1) == 1 and gw.is_contiguous() and gw.shape == (N, K)
        _chk(self.lib.plentd3_wgrad(_p(dh), dh.stride(0), _p(x), x.stride(0), _p(gw), K, _p(gb), B, N, K, int(self.rows), self._stream()))

    def _use_team(self, B):
        """Small-batch kernels for this batch size?  An explicit team= wins; otherwise batches <= TEAM_MAX_BATCH take them unless the caller asked for the
        single-wave row kernels (rows=True: the pipelined trainer's update beside resident env launches, where a 512-thread workgroup would wait for
        eight free wave slots on one compute unit)."""
        return (not self.rows and B <= TEAM_MAX_BATCH) if self.team is None else self.team

    def _use_block(self, B):
        """Large-batch kernels for this batch size?  An explicit block= wins; otherwise every batch beyond the small-batch kernels' range takes them."""
        return (B > TEAM_MAX_BATCH) if self.block is None else self.block

    def _pack(self, jobs):
        """jobs: (name, tensor, N, K, rs, cs, element offset): M(i, k) = tensor.flat[offset + i rs + k cs] -> self._packs[name] in matrix-core operand order."""
        G = PackGroup()
        assert 1 <= len(jobs) <= PACK_JOBS
        for J, (name, t, N, K, rs, cs, off) in zip(G.job, jobs):
            n = ((N + 15) // 16) * ((K + 15) // 16) * 256
            dst = self._packs.get(name)
            if dst is None:
                dst = self._packs[name] = torch.empty(n, device=self.dev, dtype=torch.float32)
            assert t.dtype == torch.float32 and t.is_contiguous() and dst.numel() == n
            J.src, J.dst, J.rs, J.cs, J.N, J.K = t.data_ptr() + 4 * off, dst.data_ptr(), rs, cs, N, K
        G.n_jobs = len(jobs)
        _chk(self.lib.plentd3_pack(C.byref(G), self._stream()))

    def _team_pack_sync(self):
        """Make the team-order packed weights current; False if the feature is off.  They stay current through the small-batch path's own updates (the Adam step writes
        every element it changes into its packed slot, PlenTd3WgradJob.pack / pack_t); anything else that may have changed a parameter -- another update path of this object
        (FlatAdam steps, polyak()), torch-level writes to the flat buffers (load_state_dict, the autograd path: the buffers' version counters) -- makes this one launch."""
        if not self._team_packed or self._critic_adam is None:
            return False
        ag = self.agent
        at, ct, cr, ac = ag.actor_target, ag.critic_target, ag.critic, ag.actor
        flats = (ag._actor_flat.flat, ag._critic_flat.flat, ag._actor_target_flat.flat, ag._critic_target_flat.flat)
        # (the PARAMETERS' version counters: they are views of the flat buffers installed through .data, which does not share the buffer's counter -- a torch-level write
        #  such as load_state_dict or `p.mul_()` moves the parameter's, not the buffer's)
        watched = (at.fc1.weight, at.fc2.weight, at.fc3.weight, ct.fc1.weight, ct.fc4.weight, ct.fc2.weight, ct.fc5.weight,
                   cr.fc1.weight, cr.fc4.weight, cr.fc2.weight, cr.fc5.weight, ac.fc1.weight, ac.fc2.weight, ac.fc3.weight)
        state = (tuple(w._version for w in watched) + tuple(f._version for f in flats) + tuple(f.data_ptr() for f in flats)
                 + (self._critic_adam.epoch, self._actor_adam.epoch, self._tp_epoch))
        if state == self._tp_state:
            return True
        tv, cv, gv = ag._critic_target_flat.views, ag._critic_flat.views, ag._critic_grads.views
        mats = [("at_w1", at.fc1.weight), ("at_w2", at.fc2.weight), ("at_w3", at.fc3.weight), ("ct_w14", tv["W14"]), ("ct_w2", ct.fc2.weight), ("ct_w5", ct.fc5.weight),
                ("c_w14", cv["W14"]), ("c_w2", cr.fc2.weight), ("c_w5", cr.fc5.weight), ("a_w1", ac.fc1.weight), ("a_w2", ac.fc2.weight), ("a_w3", ac.fc3.weight)]
        G = PackGroup()
        for J, (name, w) in zip(G.job, mats):
            N, K = w.shape
            n = -(-N // 32) * -(-K // 64) * 2048
            dst = self._tpacks.get(name)
            if dst is None or dst.numel() != n:
                dst = self._tpacks[name] = torch.zeros(n, device=self.dev, dtype=torch.float32)
            assert w.is_contiguous()
            J.src, J.dst, J.rs, J.cs, J.N, J.K, J.team = w.data_ptr(), dst.data_ptr(), K, 1, N, K, 1
        G.n_jobs = len(mats)
        _chk(self.lib.plentd3_pack(C.byref(G), self._stream()))
        # gradient tensor -> (packed matrix, packed Polyak target, stages) for the fused weight-gradient + Adam launches
        T = self._tpacks
        self._tp_by_grad = {gv["W14"].data_ptr(): (T["c_w14"], T["ct_w14"], 1), cr.fc2.weight.grad.data_ptr(): (T["c_w2"], T["ct_w2"], 4), cr.fc5.weight.grad.data_ptr(): (T["c_w5"], T["ct_w5"], 4),
                            ac.fc1.weight.grad.data_ptr(): (T["a_w1"], T["at_w1"], 1), ac.fc2.weight.grad.data_ptr(): (T["a_w2"], T["at_w2"], 4), ac.fc3.weight.grad.data_ptr(): (T["a_w3"], T["at_w3"], 4)}
        self._tp_state = state
        return True

    def _tp_job(self, j, gw):
        """Attach the packed copies of gw's matrix to a fused weight-gradient + Adam job (heads and biases have none)."""
        ent = self._tp_by_grad.get(gw.data_ptr()) if self._team_packed else None
        if ent is not None:
            j.pack, j.pack_t, j.pack_ns = ent[0].data_ptr(), ent[1].data_ptr(), ent[2]

    def _tp_critic_args(self, a):
        for n_ in ("at_w1", "at_w2", "at_w3", "ct_w14", "ct_w2", "ct_w5", "c_w14", "c_w2", "c_w5"):
            setattr(a, "tp_" + n_, self._tpacks[n_].data_ptr())

    def _tp_policy_args(self, a):
        for n_ in ("a_w1", "a_w2", "a_w3", "c_w14", "c_w2"):
            setattr(a, "tp_" + n_, self._tpacks[n_].data_ptr())

    @staticmethod
    def _nt(name, w):
        """nn.Linear weight [out][in] as the A operand of Y^T = W X^T"""
        return (name, w, w.shape[0], w.shape[1], w.shape[1], 1, 0)

    @staticmethod
    def _tr(name, w):
        """its transpose (input gradients: dX^T = W^T dY^T)"""
        return (name, w, w.shape[1], w.shape[0], 1, w.shape[1], 0)

    def _wgrad_big(self, which, B, jobs):
        """Every weight gradient of a large-batch pass in one launch (plentd3_wgrad_big): jobs = (dh, x, gw, gb or None) as _wgrad; a dh of one column
        is a head row.  With update()'s consent (self._fuse holds `which`) the partial gradients stay where they are for plentd3_adam_big to add up and step on;
        otherwise they are added into the gradient bucket here (reduce_only), as every other path leaves them."""
        grads = self.agent._critic_grads if which == "critic" else self.agent._actor_grads
        n = grads.flat.numel()
        stride = (n + 3) // 4 * 4
        if which not in self._big:
            self._big[which] = torch.zeros(WGRAD_BIG_CHUNKS, stride, device=self.dev, dtype=torch.float32)
        partial = self._big[which]
        rpc = max(16, (-(-B // WGRAD_BIG_CHUNKS) + 15) // 16 * 16)
        chunks = -(-B // rpc)
        G = WgradBig()
        assert 1 <= len(jobs) <= WGRAD_BIG_JOBS and chunks <= WGRAD_BIG_CHUNKS
        base = grads.flat.data_ptr()
        for J, (dh, x, gw, gb) in zip(G.job, jobs):
            N, K = gw.shape
            assert dh.shape[0] == B and x.shape[0] == B and dh.stride(1) == 1 and x.stride(1) == 1 and gw.is_contiguous() and x.shape[1] == K and dh.shape[1] == N
            J.dH, J.X, J.ds, J.xs, J.N, J.K = dh.data_ptr(), x.data_ptr(), dh.stride(0), x.stride(0), N, K
            J.goff, J.boff, J.kind = (gw.data_ptr() - base) // 4, ((gb.data_ptr() - base) // 4 if gb is not None else -1), int(N == 1)
        G.n_jobs, G.B, G.chunks, G.rows_per_chunk, G.stride, G.partial = len(jobs), int(B), chunks, rpc, stride, partial.data_ptr()
        _chk(self.lib.plentd3_wgrad_big(C.byref(G), self._stream()))
        adam = self._critic_adam if which == "critic" else self._actor_adam
        if adam is not None and self._fuse is not None and which in self._fuse:
            self._big_pending[which] = chunks
            return
        st = self._stream()
        _chk(self.lib.plentd3_adam_big(None, _p(grads.flat), None, None, None, None, n, 0.0, 0.0, 0.0, 0.0, None, 0.0, None, _p(partial), chunks, stride, 1, st))

    def _wgrad_group(self, B, jobs, which=None, extras=()):
        """Every (dh, x, gw, gb) of `jobs` as _wgrad, in one launch (plentd3_wgrad_group: the batch is one reduction chunk).  With update()'s consent
        (self._fuse holds `which`) the network's Adam step (+ Polyak update) is taken in the same launch and the gradients are not stored."""
        G = WgradGroup()
        assert 1 <= len(jobs) <= WGRAD_JOBS
        for j, (dh, x, gw, gb) in zip(G.job, jobs):
            N, K = gw.shape
            assert dh.shape[0] == B and x.shape[0] == B and dh.stride(1) == 1 and x.stride(1) == 1 and gw.is_contiguous() and x.shape[1] == K and dh.shape[1] == N
            j.dH, j.X, j.dW, j.db = dh.data_ptr(), x.data_ptr(), gw.data_ptr(), (gb.data_ptr() if gb is not None else None)
            j.ds, j.xs, j.dws, j.N, j.K = dh.stride(0), x.stride(0), K, N, K
        G.n_jobs, G.B = len(jobs), int(B)
        if which is not None and self._fuse is not None and which in self._fuse:
            adam = self._critic_adam if which == "critic" else self._actor_adam
            if self._tp_live:          # this pass read the packed weights: the step keeps them current
                for j, (dh, x, gw, gb) in zip(G.job, jobs):
                    self._tp_job(j, gw)
            base = adam.g.data_ptr()
            a = adam.fused_args(target=self._fuse[which], tau=self.agent.tau, extras=[(t.data_ptr() - base) // 4 for t in extras], step_advanced=True)
            _chk(self.lib.plentd3_wgrad_adam_group(C.byref(G), C.byref(a), self._stream()))
            self._fused_done.add(which)
            return
        _chk(self.lib.plentd3_wgrad_group(C.byref(G), self._stream()))

    @staticmethod
    def new_rng(device, seed):
        """Device state of one random stream for the kernels' Philox draws: int64[2] = {seed, calls so far}."""
        return torch.tensor([int(seed) & 0x7fffffffffffffff, 0], dtype=torch.long, device=device)

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream)

    def enable_flat_adam(self):
        """Both optimisers' steps as single kernels over the flat buffers (FlatAdam), with the gradient zeroing and the Polyak updates folded in by
        update().  Call after the agent's optimisers are final (the trainers replace them with capturable ones first)."""
        ag = self.agent
        if os.environ.get("PLEN_TD3_FLAT_ADAM", "1") != "1":          # development: keep torch's fused Adam
            return
        self._critic_adam = FlatAdam(self.lib, ag.critic_optimizer, ag.critic, ag._critic_flat, ag._critic_grads)
        self._actor_adam = FlatAdam(self.lib, ag.actor_optimizer, ag.actor, ag._actor_flat, ag._actor_grads)
        ag._critic_grads.zero(); ag._actor_grads.zero()
        self._zeroed = {"critic": True, "actor": True}            # gradient buckets known to be zero (left so by the last fused Adam step)
        self._eager.clear()          # the cached small-batch argument blocks hold the PREVIOUS FlatAdam objects' step counters and moment pointers

    def _zero_grads(self, which):
        """Zero a gradient bucket before a backward pass unless the last Adam step already did.
        The zero-or-skip decision is a HOST decision: inside a captured hipGraph (GraphedVecTD3Trainer / PipelinedVecTD3Trainer) it is frozen at
        capture time.  After capture the fused graphs and the autograd path (TD3Agent.train, which also writes critic gradients) must therefore not
        be interleaved on one agent: a replay would skip a zeroing the eager path has made necessary.  The trainers never do; a caller who needs
        both re-captures (trainer.recapture()) after using the autograd path."""
        grads = self.agent._critic_grads if which == "critic" else self.agent._actor_grads
        if self._zeroed.get(which) and not grads.dirty:
            self._zeroed[which] = False
            return
        # the autograd path on the same agent (TD3Agent.train: td3_actor_backward also writes critic gradients) leaves the bucket dirty
        # whatever the last fused Adam step did
        self._zeroed[which] = False
        grads.zero()

    def _probe(self, k):
        """Timeline hook (train_vec.PipelinedVecTD3Trainer.enable_timeline): stamp point k of the update; nothing unless a probe is installed."""
        if self.probe is not None:
            self.probe(k)

    def stamp(self, table, counter, div, idx):
        """Timeline probe: table[(counter // div) % rows][idx] = device clock (100 MHz ticks) here on the current stream."""
        _chk(self.lib.plentd3_stamp(_p(table), _p(counter), int(div), int(table.shape[0]), int(table.shape[1]), int(idx), self._stream()))

    def _lin_relu(self, x, w, b, out=None):
        if self.epilogue:
            return torch._addmm_activation(b, x, w.t(), out=out) if out is not None else torch._addmm_activation(b, x, w.t())
        h = torch.addmm(b, x, w.t(), out=out) if out is not None else torch.addmm(b, x, w.t())
        return h.relu_() if out is None else h

    def _update_team_eager(self, data, B, with_policy, total, guard):
        """update() for the reference's own call -- TD3Agent.train(replay_buffer, 100), eagerly, once per env-step (plen_td3.py:119-120) -- with nothing rebuilt per
        call: the iteration's 26 scratch tensors and its four argument blocks are made once per batch size and kept (the addresses of parameters, moments and gradient
        buckets never change), so a call is four or two C calls and a handful of field updates instead of 26 torch.empty and ~150 ctypes field stores (100 -> ~70 us
        per call: the GPU's 62 us show through).  Same launches, same arguments, same results as the general path (tests compare them bit for bit)."""
        ag, lib, st, dev = self.agent, self.lib, self._stream(), self.dev
        self._tp_live = self._team_pack_sync()          # (before anything reads the weights; one launch only when something else has changed them)
        ent = self._eager.get(B)
        if ent is not None and ent[8] != self._tp_live:
            ent = None
        if ent is None:
            new = lambda *shape: torch.empty(*shape, device=dev, dtype=torch.float32)
            at, ct, cr, ac = ag.actor_target, ag.critic_target, ag.critic, ag.actor
            tv, cv, gv = ag._critic_target_flat.views, ag._critic_flat.views, ag._critic_grads.views
            batch, sa_pi, sa2, dq = new(B, ROW), new(B, SA), new(B, SA), new(B, 2)
            t0, t1, c1, c2, dh2, dh1 = (new(B, 2 * H) for _ in range(6))
            loss = new(2)
            if self._done_count is None:
                self._done_count = torch.zeros(1, device=dev, dtype=torch.int32)
            if self._policy_done is None:
                self._policy_done = torch.zeros(1, device=dev, dtype=torch.int32)
            a = CriticRowsArgs()
            a.rng, a.idx, a.noise = self.rng.data_ptr(), None, None
            a.at_w1, a.at_b1, a.at_w2, a.at_b2, a.at_w3, a.at_b3 = (t.data_ptr() for t in (at.fc1.weight, at.fc1.bias, at.fc2.weight, at.fc2.bias, at.fc3.weight, at.fc3.bias))
            a.ct_w14, a.ct_b14 = tv["W14"].data_ptr(), tv["b14"].data_ptr()
            a.ct_w2, a.ct_b2, a.ct_w5, a.ct_b5, a.ct_w3, a.ct_b3, a.ct_w6, a.ct_b6 = (t.data_ptr() for t in (ct.fc2.weight, ct.fc2.bias, ct.fc5.weight, ct.fc5.bias, ct.fc3.weight, ct.fc3.bias, ct.fc6.weight, ct.fc6.bias))
            a.c_w14, a.c_b14 = cv["W14"].data_ptr(), cv["b14"].data_ptr()
            a.c_w2, a.c_b2, a.c_w5, a.c_b5, a.c_w3, a.c_b3, a.c_w6, a.c_b6 = (t.data_ptr() for t in (cr.fc2.weight, cr.fc2.bias, cr.fc5.weight, cr.fc5.bias, cr.fc3.weight, cr.fc3.bias, cr.fc6.weight, cr.fc6.bias))
            a.batch, a.sa_pi, a.t0, a.t1, a.sa2, a.c1, a.c2, a.dh2, a.dh1, a.dq = (t.data_ptr() for t in (batch, sa_pi, t0, t1, sa2, c1, c2, dh2, dh1, dq))
            a.loss, a.db3a, a.db3b = loss.data_ptr(), cr.fc3.bias.grad.data_ptr(), cr.fc6.bias.grad.data_ptr()
            a.done_count, a.rng_bump, a.B = self._done_count.data_ptr(), self.rng.data_ptr(), int(B)
            a.adam_step = self._critic_adam.step_t.data_ptr()
            if self._tp_live:
                self._tp_critic_args(a)

            def group(jobs):
                G = WgradGroup()
                for j, (dh, x, gw, gb) in zip(G.job, jobs):
                    N, K = gw.shape
                    j.dH, j.X, j.dW, j.db = dh.data_ptr(), x.data_ptr(), gw.data_ptr(), (gb.data_ptr() if gb is not None else None)
                    j.ds, j.xs, j.dws, j.N, j.K = dh.stride(0), x.stride(0), K, N, K
                    if self._tp_live:
                        self._tp_job(j, gw)
                G.n_jobs, G.B = len(jobs), int(B)
                return G
            Gc = group([(dq[:, 0:1], c2[:, :H], cr.fc3.weight.grad, None), (dq[:, 1:2], c2[:, H:], cr.fc6.weight.grad, None),
                        (dh2[:, :H], c1[:, :H], cr.fc2.weight.grad, cr.fc2.bias.grad), (dh2[:, H:], c1[:, H:], cr.fc5.weight.grad, cr.fc5.bias.grad),
                        (dh1, batch[:, :SA], gv["W14"], gv["b14"])])
            a_pi, dz = new(B, A), new(B, A)
            p1, p2, g1, dg2, dg1, dp2, dp1 = (new(B, H) for _ in range(7))
            pp = PolicyRowsArgs()
            pp.a_w1, pp.a_b1, pp.a_w2, pp.a_b2, pp.a_w3, pp.a_b3 = (t.data_ptr() for t in (ac.fc1.weight, ac.fc1.bias, ac.fc2.weight, ac.fc2.bias, ac.fc3.weight, ac.fc3.bias))
            pp.c_w1, pp.c_b1, pp.c_w2, pp.c_b2, pp.c_w3 = (t.data_ptr() for t in (cr.fc1.weight, cr.fc1.bias, cr.fc2.weight, cr.fc2.bias, cr.fc3.weight))
            pp.sa_pi, pp.a_pi, pp.p1, pp.p2, pp.g1, pp.dg2, pp.dg1, pp.dz, pp.dp2, pp.dp1 = (t.data_ptr() for t in (sa_pi, a_pi, p1, p2, g1, dg2, dg1, dz, dp2, dp1))
            pp.B = int(B)
            pp.adam_step, pp.done_count = self._actor_adam.step_t.data_ptr(), self._policy_done.data_ptr()
            if self._tp_live:
                self._tp_policy_args(pp)
            Gp = group([(dz, p2, ac.fc3.weight.grad, ac.fc3.bias.grad), (dp2, p1, ac.fc2.weight.grad, ac.fc2.bias.grad), (dp1, batch[:, :S], ac.fc1.weight.grad, ac.fc1.bias.grad)])
            base = self._critic_adam.g.data_ptr()
            extras = [(cr.fc3.bias.grad.data_ptr() - base) // 4, (cr.fc6.bias.grad.data_ptr() - base) // 4]
            keep = (batch, sa_pi, sa2, dq, t0, t1, c1, c2, dh2, dh1, loss, a_pi, dz, p1, p2, g1, dg2, dg1, dp2, dp1)
            ent = self._eager[B] = (a, Gc, pp, Gp, extras, loss, (batch[:, :S], sa_pi, B), keep, self._tp_live)
        a, Gc, pp, Gp, extras, loss, saved, _, _ = ent
        self._fused_done = set()
        self._zero_grads("critic")
        self._critic_adam.ensure_bound()
        a.data, a.total, a.capacity, a.guard = data.data_ptr(), total.data_ptr(), int(data.shape[0]), int(guard)
        a.sigma, a.clip, a.max_a, a.gamma = float(ag.policy_noise), float(ag.noise_clip), float(ag.max_action), float(ag.discount)
        _chk(lib.plentd3_critic_team(C.byref(a), st))
        ad = self._critic_adam.fused_args(target=ag._critic_target_flat.flat if with_policy else None, tau=ag.tau, extras=extras, step_advanced=True)
        _chk(lib.plentd3_wgrad_adam_group(C.byref(Gc), C.byref(ad), st))
        self._fused_done.add("critic")
        self._zeroed["critic"] = True
        self._saved, self._team_pass, self._block_pass = saved, True, False
        # NOTE: a VIEW of this cache entry's persistent loss word, rewritten by the next small-batch call -- valid until then.  Callers that keep losses across
        # train() calls must copy (float(...) / .clone()); a clone here would put one more launch on a path that is host-bound at 63 us per call.
        ag.last_critic_loss = loss[0]
        if with_policy:
            self._zero_grads("actor")
            self._actor_adam.ensure_bound()
            pp.max_a = float(ag.max_action)
            _chk(lib.plentd3_policy_team(C.byref(pp), st))
            ad = self._actor_adam.fused_args(target=ag._actor_target_flat.flat, tau=ag.tau, step_advanced=True)
            _chk(lib.plentd3_wgrad_adam_group(C.byref(Gp), C.byref(ad), st))
            self._fused_done.add("actor")
            self._zeroed["actor"] = True
            ag.last_actor_loss = None
        return loss[0]

    def update(self, data, idx, with_policy, noise=None, all_reduce=True, total=None, guard=0):
        """The whole iteration; with torch.distributed initialised the two gradient buckets are averaged over ranks before their Adam steps."""
        ag = self.agent
        flat = self._critic_adam is not None
        fuse = flat and self.fuse_adam and not all_reduce         # (with ranks to average over, the gradients have to exist in the bucket)
        if (fuse and isinstance(idx, int) and noise is None and total is not None and self._alloc is None and self.probe is None and not self.rows and not self._use_block(idx)
                and self._use_team(idx) and self.eager_cache and data.dtype == torch.float32 and data.is_contiguous() and data.shape[1] == ROW
                and not torch.cuda.is_current_stream_capturing()):
            return self._update_team_eager(data, idx, with_policy, total, guard)
        self._fused_done = set()
        self._fuse = {"critic": ag._critic_target_flat.flat if with_policy else None} if fuse else None
        try:
            loss = self.critic_backward(data, idx, noise, total, guard)
        finally:
            self._fuse = None            # (consent to the in-kernel Adam step never outlives the pass it was given for)
        return self._after_critic_backward(loss, with_policy, all_reduce, flat, fuse)

    def _after_critic_backward(self, loss, with_policy, all_reduce, flat, fuse):
        ag = self.agent
        if all_reduce:
            ag._critic_grads.all_reduce_mean()
        if "critic" in self._big_pending:        # large batch: the step on bucket + partial gradients, the bucket left zero (+ the critic's Polyak update)
            big, chunks = self._big["critic"], self._big_pending.pop("critic")
            self._critic_adam.step_big(big, chunks, big.shape[1], target=ag._critic_target_flat.flat if with_policy else None, tau=ag.tau)
            self._zeroed["critic"] = True
        elif "critic" in self._fused_done:
            self._zeroed["critic"] = True      # nothing was written into the bucket (the head biases' sums were consumed and zeroed)
        elif flat:     # Adam + zeroed bucket (+ the critic's Polyak update, which nothing reads before the iteration's end: td3.py:348-352) in one pass
            self._critic_adam.step(zero_grad=True, target=ag._critic_target_flat.flat if with_policy else None, tau=ag.tau)
            self._zeroed["critic"] = True
        else:
            ag.critic_optimizer.step()
        ag.last_critic_loss = loss
        if with_policy:
            self._fuse = {"actor": ag._actor_target_flat.flat} if fuse else None
            try:
                self.policy_backward()
            finally:
                self._fuse = None
            if all_reduce:
                ag._actor_grads.all_reduce_mean()
            ag.last_actor_loss = None          # (-mean Q1 itself is not needed for the update; the autograd path reports it)
            if "actor" in self._big_pending:
                big, chunks = self._big["actor"], self._big_pending.pop("actor")
                self._actor_adam.step_big(big, chunks, big.shape[1], target=ag._actor_target_flat.flat, tau=ag.tau)
                self._zeroed["actor"] = True
            elif "actor" in self._fused_done:
                self._zeroed["actor"] = True
            elif flat:
                self._actor_adam.step(zero_grad=True, target=ag._actor_target_flat.flat, tau=ag.tau)
                self._zeroed["actor"] = True
            else:
                ag.actor_optimizer.step()
                self.polyak()
        return loss

    def explore(self, state, sigma, actor=None, rng=None, packed=False):
        """Collect-phase action (plen_td3.py:101-104): clamp(actor(state) + N(0, sigma), +-max_action) -- 3 GEMMs and one fused kernel that draws
        its own noise from `rng` (new_rng(); bumped by the store() that follows).  `actor`: the network to act with (default the online
        actor; the pipelined trainer passes a behaviour copy).  rng None: torch.randn (autograd-path compatible)."""
        ag = self.agent
        ac = ag.actor if actor is None else actor
        if self.rows and rng is not None and self._use_block(int(state.shape[0])) and (packed or self.block is True):
            # large batches: the block kernel on the acting network's packed weights.  packed: the caller has run pack_actor(actor) since the network last changed
            # (the pipelined trainer does, on the update stream, off the collectors' critical path); with block=True and no such promise they are packed here, every
            # call -- nothing observable says whether raw-pointer kernels (the Adam steps) have rewritten the parameters since.  Without either the four-wave row
            # kernel below runs: its workgroups need no LDS to speak of and start sooner beside a resident env launch (bench.py's policy leg: 11.88 against 11.74 M)
            n = int(state.shape[0])
            assert state.dtype == torch.float32 and state.is_contiguous() and state.shape[1] == S
            packs = self.pack_actor(ac, launch=not packed)
            act = (self._alloc or (lambda *shape: torch.empty(*shape, device=self.dev, dtype=torch.float32)))(n, A)
            pa = ActorBlockArgs()
            a = pa.rows
            a.a_b1, a.a_b2, a.a_b3 = ac.fc1.bias.data_ptr(), ac.fc2.bias.data_ptr(), ac.fc3.bias.data_ptr()
            a.state, a.rng, a.action = state.data_ptr(), rng.data_ptr(), act.data_ptr()
            a.sigma, a.max_a, a.B = float(sigma), float(ag.max_action), n
            pa.p_a_w1, pa.p_a_w2, pa.p_a_w3 = (t.data_ptr() for t in packs)
            _chk(self.lib.plentd3_actor_block(C.byref(pa), self._stream()))
            return act
        if self.rows and rng is not None:
            n = int(state.shape[0])
            assert state.dtype == torch.float32 and state.is_contiguous() and state.shape[1] == S
            new = self._alloc or (lambda *shape: torch.empty(*shape, device=self.dev, dtype=torch.float32))
            p1, p2, act = new(n, H), new(n, H), new(n, A)
            a = ActorRowsArgs()
            a.a_w1, a.a_b1, a.a_w2, a.a_b2, a.a_w3, a.a_b3 = (t.data_ptr() for t in (ac.fc1.weight, ac.fc1.bias, ac.fc2.weight, ac.fc2.bias, ac.fc3.weight, ac.fc3.bias))
            a.state, a.rng, a.p1, a.p2, a.action = state.data_ptr(), rng.data_ptr(), p1.data_ptr(), p2.data_ptr(), act.data_ptr()
            a.sigma, a.max_a, a.B = float(sigma), float(ag.max_action), n
            _chk(self.lib.plentd3_actor_rows(C.byref(a), self._stream()))
            return act
        with torch.no_grad():
            p2 = self._lin_relu(self._lin_relu(state, ac.fc1.weight, ac.fc1.bias), ac.fc2.weight, ac.fc2.bias)
            pre = torch.addmm(ac.fc3.bias, p2, ac.fc3.weight.t())
            noise = torch.randn_like(pre) if rng is None else None
            a = torch.empty_like(pre)
            _chk(self.lib.plentd3_explore(_p(pre), _p(noise), _p(rng), _p(a), float(sigma), float(ag.max_action), pre.numel(), self._stream()))
        return a

    def pack_actor(self, ac, launch=True):
        """The three weight matrices of an acting network (the online actor, a behaviour copy) in matrix-core operand order for plentd3_actor_block; one set of
        packed buffers per network object.  launch False: only look the buffers up (they must have been packed since the network last changed)."""
        names = self._actor_packs.get(id(ac))
        if names is None:
            names = self._actor_packs[id(ac)] = tuple("act%d_%s" % (len(self._actor_packs), n_) for n_ in ("w1", "w2", "w3"))
            launch = True
        if launch:
            self._pack([self._nt(names[0], ac.fc1.weight), self._nt(names[1], ac.fc2.weight), self._nt(names[2], ac.fc3.weight)])
        return tuple(self._packs[n_] for n_ in names)

    def uniform_actions(self, n, rng):
        """Warm-up actions U[-1, 1)^18 for n envs (plen_td3.py:91-92), drawn in-kernel from `rng` (bumped by the store() that follows)."""
        a = torch.empty(n, A, device=self.dev, dtype=torch.float32)
        _chk(self.lib.plentd3_uniform_actions(_p(rng), _p(a), n * A, self._stream()))
        return a

    def store(self, data, total, state, action, next_obs, reward, done, rng=None, episodes=None, advance=None, step=None):
        """One vector step into the packed replay ring at positions (total + e) % capacity (plen_td3.py:109-113); `total` = device int64 scalar.
        rng: the collect stream's random state, whose call counter this kernel advances.  episodes: (ep_ret [n, 2] float32, stats [3] float64) device
        tensors for the episode bookkeeping (running return / length per env; finished episodes summed into stats)."""
        n = int(state.shape[0])
        for t in (state, action, next_obs, reward):
            assert t.dtype == torch.float32 and t.is_contiguous()
        assert done.dtype == torch.uint8 and total.dtype == torch.long
        ep, st = episodes if episodes is not None else (None, None)
        if st is not None:
            # k_store adds into stats[0..2] with 8-byte atomics (float64 since round 3) and reads / writes ep_ret[e][0..1] as float32: a caller on the
            # round-2 contract (float32 stats, 12 bytes) would have its neighbouring allocation overwritten
            assert st.dtype == torch.float64 and st.numel() >= 3 and st.is_contiguous(), "episode stats must be a contiguous float64 tensor of >= 3 elements"
            assert ep is not None and ep.dtype == torch.float32 and tuple(ep.shape) == (n, 2) and ep.is_contiguous(), "ep_ret must be float32 [n, 2]"
        if step is not None:          # ... and total += step[0] in the same launch (step = (rows, zero-initialised uint32 device scalar))
            rows, ctr = step
            assert ctr.dtype in (torch.int32, torch.uint32) and ctr.numel() == 1 and ctr.device == total.device
            if advance is not None:
                assert advance.dtype == torch.float32 and advance.is_contiguous() and advance.shape == state.shape and advance.data_ptr() != state.data_ptr()
            _chk(self.lib.plentd3_store_step(_p(data), _p(total), int(data.shape[0]), _p(state), _p(action), _p(next_obs), _p(reward), _p(done), _p(rng), _p(ep), _p(st), n,
                                             _p(advance), int(rows), _p(ctr), self._stream()))
            return
        if advance is not None:       # ... and state <- advance (the observation to act on next) in the same launch
            assert advance.dtype == torch.float32 and advance.is_contiguous() and advance.shape == state.shape and advance.data_ptr() != state.data_ptr()
            _chk(self.lib.plentd3_store_advance(_p(data), _p(total), int(data.shape[0]), _p(state), _p(action), _p(next_obs), _p(reward), _p(done), _p(rng), _p(ep), _p(st), n,
                                                _p(advance), self._stream()))
            return
        _chk(self.lib.plentd3_store(_p(data), _p(total), int(data.shape[0]), _p(state), _p(action), _p(next_obs), _p(reward), _p(done), _p(rng), _p(ep), _p(st), n, self._stream()))

    def critic_backward(self, data, idx, noise=None, total=None, guard=0):
        """Sample, targets, critic forward / loss / backward: gradients land in the critic's flat bucket.  Returns the loss (device scalar).
        idx: LongTensor [B] of replay rows, or an int B with `total` (device int64 scalar: transitions written so far) to draw them here
        (then, with self.rows, everything up to the weight gradients runs as one row-block kernel: critic_backward_rows)."""
        self._team_pass = False
        self._block_pass = False
        if isinstance(idx, int) and noise is None and self._use_block(idx):
            return self.critic_backward_rows(data, idx, total, guard, block=True)
        if self.block is True and not isinstance(idx, int):            # explicit rows (and noise) through the large-batch kernels: the golden iterations
            assert idx.dtype == torch.long and idx.is_contiguous() and (noise is None or (noise.is_contiguous() and tuple(noise.shape) == (idx.shape[0], A)))
            return self.critic_backward_rows(data, int(idx.shape[0]), total, guard, block=True, idx=idx, noise=noise)
        if isinstance(idx, int) and noise is None and (self.rows or self._use_team(idx)):
            return self.critic_backward_rows(data, idx, total, guard, team=self._use_team(idx))
        if self.team is True and not isinstance(idx, int):          # explicit rows (and noise) through the small-batch kernels: the golden iterations
            assert idx.dtype == torch.long and idx.is_contiguous() and (noise is None or (noise.is_contiguous() and tuple(noise.shape) == (idx.shape[0], A)))
            return self.critic_backward_rows(data, int(idx.shape[0]), total, guard, team=True, idx=idx, noise=noise)
        ag, lib, st = self.agent, self.lib, self._stream()
        dev = self.dev
        assert data.dtype == torch.float32 and data.is_contiguous() and data.shape[1] == ROW
        new = self._alloc or (lambda *shape: torch.empty(*shape, device=dev, dtype=torch.float32))
        if isinstance(idx, int):
            B = idx
            batch, sa_pi, loss = new(B, ROW), new(B, SA), new(2)
            _chk(lib.plentd3_sample_gather(_p(data), None, _p(self.rng), _p(total), int(data.shape[0]), int(guard), None, _p(batch), _p(sa_pi), _p(loss), B, st))
        else:
            B = int(idx.shape[0])
            assert idx.dtype == torch.long
            batch, sa_pi, loss = new(B, ROW), new(B, SA), new(2)
            _chk(lib.plentd3_gather(_p(data), _p(idx), _p(batch), _p(sa_pi), _p(loss), B, st))
        s, sa, s2 = batch[:, :S], batch[:, :SA], batch[:, SA:SA + S]
        relu_both = not self.epilogue
        self._probe(1)
        with torch.no_grad():
            # ---- target policy smoothing + clipped double-Q target (td3.py:277-309) ----
            at, ct = ag.actor_target, ag.critic_target
            a1 = self._lin_relu(s2, at.fc1.weight, at.fc1.bias)
            a2 = self._lin_relu(a1, at.fc2.weight, at.fc2.bias)
            pre = torch.addmm(at.fc3.bias, a2, at.fc3.weight.t())
            sa2 = new(B, SA)          # noise None: the kernel draws torch.randn_like(action)'s equivalent itself (Philox, self.rng)
            _chk(lib.plentd3_target_action(_p(pre), _p(None if noise is None else noise.contiguous()), _p(self.rng), _p(batch), _p(sa2),
                                           float(ag.policy_noise), float(ag.noise_clip), float(ag.max_action), B, st))
            tv = ag._critic_target_flat.views
            h1 = self._lin_relu(sa2, tv["W14"], tv["b14"])                       # both target critics' first layers in one GEMM
            h2 = new(B, 2 * H)
            self._lin_relu(h1[:, :H], ct.fc2.weight, ct.fc2.bias, out=h2[:, :H])
            self._lin_relu(h1[:, H:], ct.fc5.weight, ct.fc5.bias, out=h2[:, H:])
            if relu_both:
                h2.relu_()
            y = new(B)
            _chk(lib.plentd3_q_heads(_p(h2), _p(ct.fc3.weight), _p(ct.fc3.bias), _p(ct.fc6.weight), _p(ct.fc6.bias), _p(batch), _p(y), None, None, None, None, None,
                                     float(ag.discount), B, 0, st))
            self._probe(2)
            # ---- critic forward, loss, backward (td3.py:312-331) ----
            cr = ag.critic
            cv, gv = ag._critic_flat.views, ag._critic_grads.views
            self._zero_grads("critic")
            c1 = self._lin_relu(sa, cv["W14"], cv["b14"])
            c2 = new(B, 2 * H)
            self._lin_relu(c1[:, :H], cr.fc2.weight, cr.fc2.bias, out=c2[:, :H])
            self._lin_relu(c1[:, H:], cr.fc5.weight, cr.fc5.bias, out=c2[:, H:])
            if relu_both:
                c2.relu_()
            dq = new(B, 2)
            _chk(lib.plentd3_q_heads(_p(c2), _p(cr.fc3.weight), _p(cr.fc3.bias), _p(cr.fc6.weight), _p(cr.fc6.bias), _p(batch), _p(y), _p(dq), _p(loss),
                                     _p(cr.fc3.bias.grad), _p(cr.fc6.bias.grad), _p(self.rng), float(ag.discount), B, 1, st))
            self._probe(3)
            # last layer weight gradients: dW3_c = h2_c^T dq_c
            _chk(lib.plentd3_colsum(_p(c2), 2 * H, _p(dq), 2, _p(cr.fc3.weight.grad), B, H, int(self.rows), st))
            _chk(lib.plentd3_colsum(C.c_void_p(c2.data_ptr() + 4 * H), 2 * H, C.c_void_p(dq.data_ptr() + 4), 2, _p(cr.fc6.weight.grad), B, H, int(self.rows), st))
            dh2 = new(B, 2 * H)
            _chk(lib.plentd3_dh2(_p(dq), _p(cr.fc3.weight), _p(cr.fc6.weight), _p(c2), _p(dh2), B, 2, 2 * H, st))
            self._wgrad(dh2[:, :H], c1[:, :H], cr.fc2.weight.grad, cr.fc2.bias.grad)
            self._wgrad(dh2[:, H:], c1[:, H:], cr.fc5.weight.grad, cr.fc5.bias.grad)
            dh1 = new(B, 2 * H)
            torch.mm(dh2[:, :H], cr.fc2.weight, out=dh1[:, :H])
            torch.mm(dh2[:, H:], cr.fc5.weight, out=dh1[:, H:])
            _chk(lib.plentd3_relu_mask(_p(dh1), _p(c1), B, 2 * H, 2 * H, st))
            self._wgrad(dh1, sa, gv["W14"], gv["b14"])
        self._saved = (s, sa_pi, B)
        self._probe(4)
        return loss[0]

    def critic_backward_rows(self, data, B, total, guard=0, team=False, idx=None, noise=None, block=False):
        """critic_backward() with everything between the sampling and the weight gradients in ONE launch of single-wave workgroups
        (plentd3_critic_rows, csrc/td3_rows.hip): 8 kernels per critic update instead of ~35, and none of them needs more than one free wave slot
        per workgroup to start, which is what the update lacks beside two resident env launches.  Draws its random numbers in-kernel
        (self.rng), so it takes no idx / noise arguments."""
        ag, lib, st = self.agent, self.lib, self._stream()
        dev = self.dev
        assert data.dtype == torch.float32 and data.is_contiguous() and data.shape[1] == ROW and (total is not None or idx is not None)
        assert team or block or (idx is None and noise is None)
        new = self._alloc or (lambda *shape: torch.empty(*shape, device=dev, dtype=torch.float32))
        with torch.no_grad():
            at, ct, cr = ag.actor_target, ag.critic_target, ag.critic
            tv, cv, gv = ag._critic_target_flat.views, ag._critic_flat.views, ag._critic_grads.views
            batch, sa_pi, sa2, dq = new(B, ROW), new(B, SA), new(B, SA), new(B, 2)
            t0, t1, c1, c2, dh2, dh1 = (new(B, 2 * H) for _ in range(6))
            loss = new(2) if (team or block) else torch.zeros(2, device=dev, dtype=torch.float32)         # (the team / block kernels store their loss, the row kernel adds to it)
            self._zero_grads("critic")
            if self._done_count is None:
                self._done_count = torch.zeros(1, device=dev, dtype=torch.int32)
            a = CriticRowsArgs()
            a.data, a.rng, a.total, a.capacity, a.guard = data.data_ptr(), self.rng.data_ptr(), (total.data_ptr() if total is not None else None), int(data.shape[0]), int(guard)
            a.idx, a.noise = (idx.data_ptr() if idx is not None else None), (noise.data_ptr() if noise is not None else None)
            fused_step = team and self._fuse is not None and "critic" in self._fuse        # the pass kernel advances the step counter for the launch that takes the step
            if fused_step:
                self._critic_adam.ensure_bound()          # a resumed optimiser's state is taken over BEFORE the kernel counts the step
            a.adam_step = self._critic_adam.step_t.data_ptr() if fused_step else None
            for pre, net in (("at", at),):
                a.at_w1, a.at_b1, a.at_w2, a.at_b2, a.at_w3, a.at_b3 = (t.data_ptr() for t in (net.fc1.weight, net.fc1.bias, net.fc2.weight, net.fc2.bias, net.fc3.weight, net.fc3.bias))
            a.ct_w14, a.ct_b14 = tv["W14"].data_ptr(), tv["b14"].data_ptr()
            a.ct_w2, a.ct_b2, a.ct_w5, a.ct_b5, a.ct_w3, a.ct_b3, a.ct_w6, a.ct_b6 = (t.data_ptr() for t in (ct.fc2.weight, ct.fc2.bias, ct.fc5.weight, ct.fc5.bias, ct.fc3.weight, ct.fc3.bias, ct.fc6.weight, ct.fc6.bias))
            a.c_w14, a.c_b14 = cv["W14"].data_ptr(), cv["b14"].data_ptr()
            a.c_w2, a.c_b2, a.c_w5, a.c_b5, a.c_w3, a.c_b3, a.c_w6, a.c_b6 = (t.data_ptr() for t in (cr.fc2.weight, cr.fc2.bias, cr.fc5.weight, cr.fc5.bias, cr.fc3.weight, cr.fc3.bias, cr.fc6.weight, cr.fc6.bias))
            a.batch, a.sa_pi, a.t0, a.t1, a.sa2, a.c1, a.c2, a.dh2, a.dh1, a.dq = (t.data_ptr() for t in (batch, sa_pi, t0, t1, sa2, c1, c2, dh2, dh1, dq))
            a.loss, a.db3a, a.db3b = loss.data_ptr(), cr.fc3.bias.grad.data_ptr(), cr.fc6.bias.grad.data_ptr()
            a.done_count, a.rng_bump = self._done_count.data_ptr(), self.rng.data_ptr()
            a.sigma, a.clip, a.max_a, a.gamma, a.B = float(ag.policy_noise), float(ag.noise_clip), float(ag.max_action), float(ag.discount), int(B)
            self._probe(1)
            if block:       # (the packing launch sits between probes 1 and 2, so that probes 2 -> 3 bracket the pass kernel alone: bench.py's legs.td3.roofline)
                nt, tr = self._nt, self._tr
                self._pack([nt("at_w1", at.fc1.weight), nt("at_w2", at.fc2.weight), nt("at_w3", at.fc3.weight),
                            nt("ct_w14", tv["W14"]), nt("ct_w2", ct.fc2.weight), nt("ct_w5", ct.fc5.weight),
                            nt("c_w14", cv["W14"]), nt("c_w2", cr.fc2.weight), nt("c_w5", cr.fc5.weight), tr("c_w2t", cr.fc2.weight), tr("c_w5t", cr.fc5.weight)])
            self._probe(2)
            if team:        # small batch: a team of 8 waves per row block, then every weight gradient in one launch (head rows as 1 x 256 products)
                # (packed weights only when this pass's Adam steps are taken inside the weight-gradient launches, which keep the packed copies current: with ranks
                #  to average over the separate Adam kernel would make every pass start with a packing launch)
                self._tp_live = self._fuse is not None and "critic" in self._fuse and self._team_pack_sync()
                if self._tp_live:
                    self._tp_critic_args(a)
                _chk(lib.plentd3_critic_team(C.byref(a), st))
                self._probe(3)
                self._wgrad_group(B, [(dq[:, 0:1], c2[:, :H], cr.fc3.weight.grad, None), (dq[:, 1:2], c2[:, H:], cr.fc6.weight.grad, None),
                                      (dh2[:, :H], c1[:, :H], cr.fc2.weight.grad, cr.fc2.bias.grad), (dh2[:, H:], c1[:, H:], cr.fc5.weight.grad, cr.fc5.bias.grad),
                                      (dh1, batch[:, :SA], gv["W14"], gv["b14"])], which="critic", extras=(cr.fc3.bias.grad, cr.fc6.bias.grad))
                self._saved = (batch[:, :S], sa_pi, B)
                self._team_pass = True
                self._probe(4)
                return loss[0]
            if block:       # large batch: 16 rows per 256-thread workgroup, packed weights (csrc/td3_block.hip), all weight gradients in one launch
                nb = (B + 15) // 16
                if self._partials is None or self._partials.numel() < 4 * nb + 128:
                    self._partials = torch.zeros(4 * nb + 128, device=dev, dtype=torch.float32)        # (+ room for a development build's phase stamps)
                pa = CriticBlockArgs()
                pa.rows = a
                for n_ in ("at_w1", "at_w2", "at_w3", "ct_w14", "ct_w2", "ct_w5", "c_w14", "c_w2", "c_w5", "c_w2t", "c_w5t"):
                    setattr(pa, "p_" + n_, self._packs[n_].data_ptr())
                pa.partials = self._partials.data_ptr()
                _chk(lib.plentd3_critic_block(C.byref(pa), st))
                self._block_pass = True
                self._probe(3)
                jobs = [(dq[:, 0:1], c2[:, :H], cr.fc3.weight.grad, None), (dq[:, 1:2], c2[:, H:], cr.fc6.weight.grad, None),
                        (dh2[:, :H], c1[:, :H], cr.fc2.weight.grad, cr.fc2.bias.grad), (dh2[:, H:], c1[:, H:], cr.fc5.weight.grad, cr.fc5.bias.grad),
                        (dh1, batch[:, :SA], gv["W14"], gv["b14"])]
                self._wgrad_big("critic", B, jobs)
                self._saved = (batch[:, :S], sa_pi, B)
                self._probe(4)
                return loss[0]
            _chk(lib.plentd3_critic_rows(C.byref(a), st))
            self._probe(3)
            # weight gradients (reductions over the batch): last layers, second layers, stacked first layers
            _chk(lib.plentd3_colsum(_p(c2), 2 * H, _p(dq), 2, _p(cr.fc3.weight.grad), B, H, int(self.rows), st))
            _chk(lib.plentd3_colsum(C.c_void_p(c2.data_ptr() + 4 * H), 2 * H, C.c_void_p(dq.data_ptr() + 4), 2, _p(cr.fc6.weight.grad), B, H, int(self.rows), st))
            self._wgrad(dh2[:, :H], c1[:, :H], cr.fc2.weight.grad, cr.fc2.bias.grad)
            self._wgrad(dh2[:, H:], c1[:, H:], cr.fc5.weight.grad, cr.fc5.bias.grad)
            self._wgrad(dh1, batch[:, :SA], gv["W14"], gv["b14"])
        self._saved = (batch[:, :S], sa_pi, B)
        self._probe(4)
        return loss[0]

    def policy_backward(self):
        """Delayed policy gradient through the (already updated) critic's Q1 (td3.py:334-345): gradients land in the actor's flat bucket."""
        ag, lib, st = self.agent, self.lib, self._stream()
        dev = self.dev
        s, sa_pi, B = self._saved
        new = self._alloc or (lambda *shape: torch.empty(*shape, device=dev, dtype=torch.float32))
        cr = ag.critic
        if self.rows or self._team_pass or self._block_pass:
            with torch.no_grad():
                ac = ag.actor
                self._zero_grads("actor")
                a_pi, dz = new(B, A), new(B, A)
                p1, p2, g1, dg2, dg1, dp2, dp1 = (new(B, H) for _ in range(7))
                a = PolicyRowsArgs()
                a.a_w1, a.a_b1, a.a_w2, a.a_b2, a.a_w3, a.a_b3 = (t.data_ptr() for t in (ac.fc1.weight, ac.fc1.bias, ac.fc2.weight, ac.fc2.bias, ac.fc3.weight, ac.fc3.bias))
                a.c_w1, a.c_b1, a.c_w2, a.c_b2, a.c_w3 = (t.data_ptr() for t in (cr.fc1.weight, cr.fc1.bias, cr.fc2.weight, cr.fc2.bias, cr.fc3.weight))
                a.sa_pi, a.a_pi, a.p1, a.p2, a.g1, a.dg2, a.dg1, a.dz, a.dp2, a.dp1 = (t.data_ptr() for t in (sa_pi, a_pi, p1, p2, g1, dg2, dg1, dz, dp2, dp1))
                a.max_a, a.B = float(ag.max_action), int(B)
                a.adam_step = a.done_count = None
                if self._team_pass and self._fuse is not None and "actor" in self._fuse:
                    if self._policy_done is None:
                        self._policy_done = torch.zeros(1, device=dev, dtype=torch.int32)
                    self._actor_adam.ensure_bound()
                    a.adam_step, a.done_count = self._actor_adam.step_t.data_ptr(), self._policy_done.data_ptr()
                if self._team_pass:
                    if self._tp_live:
                        self._tp_policy_args(a)
                    _chk(lib.plentd3_policy_team(C.byref(a), st))
                    self._wgrad_group(B, [(dz, p2, ac.fc3.weight.grad, ac.fc3.bias.grad), (dp2, p1, ac.fc2.weight.grad, ac.fc2.bias.grad),
                                          (dp1, s, ac.fc1.weight.grad, ac.fc1.bias.grad)], which="actor")
                    return
                if self._block_pass:
                    nt, tr = self._nt, self._tr
                    self._pack([nt("a_w1", ac.fc1.weight), nt("a_w2", ac.fc2.weight), nt("a_w3", ac.fc3.weight),
                                nt("c_w14", ag._critic_flat.views["W14"]), nt("c_w2", cr.fc2.weight), tr("c_w2t", cr.fc2.weight),
                                ("c_w1ta", ag._critic_flat.views["W14"], A, H, 1, SA, S),            # (i = action j, k = hidden) -> fc1.weight[k][26 + j]
                                tr("a_w3t", ac.fc3.weight), tr("a_w2t", ac.fc2.weight)])
                    pa = PolicyBlockArgs()
                    pa.rows = a
                    for n_ in ("a_w1", "a_w2", "a_w3", "c_w14", "c_w2", "c_w2t", "c_w1ta", "a_w3t", "a_w2t"):
                        setattr(pa, "p_" + n_, self._packs[n_].data_ptr())
                    _chk(lib.plentd3_policy_block(C.byref(pa), st))
                    self._wgrad_big("actor", B, [(dz, p2, ac.fc3.weight.grad, ac.fc3.bias.grad), (dp2, p1, ac.fc2.weight.grad, ac.fc2.bias.grad),
                                                 (dp1, s, ac.fc1.weight.grad, ac.fc1.bias.grad)])
                    return
                _chk(lib.plentd3_policy_rows(C.byref(a), st))
                self._wgrad(dz, p2, ac.fc3.weight.grad, ac.fc3.bias.grad)
                self._wgrad(dp2, p1, ac.fc2.weight.grad, ac.fc2.bias.grad)
                self._wgrad(dp1, s, ac.fc1.weight.grad, ac.fc1.bias.grad)
        else:
            with torch.no_grad():
                ac = ag.actor
                self._zero_grads("actor")
                p1 = self._lin_relu(s, ac.fc1.weight, ac.fc1.bias)
                p2 = self._lin_relu(p1, ac.fc2.weight, ac.fc2.bias)
                pre = torch.addmm(ac.fc3.bias, p2, ac.fc3.weight.t())
                a_pi = new(B, A)
                _chk(lib.plentd3_tanh_out(_p(pre), _p(a_pi), _p(sa_pi), float(ag.max_action), B, st))
                g1 = self._lin_relu(sa_pi, cr.fc1.weight, cr.fc1.bias)
                g2 = self._lin_relu(g1, cr.fc2.weight, cr.fc2.bias)
                dg2 = new(B, H)
                _chk(lib.plentd3_dh2(None, _p(cr.fc3.weight), None, _p(g2), _p(dg2), B, 1, H, st))       # d(-mean Q1)/d g2
                dg1 = torch.mm(dg2, cr.fc2.weight)
                _chk(lib.plentd3_relu_mask(_p(dg1), _p(g1), B, H, H, st))
                dsa = torch.mm(dg1, cr.fc1.weight)                                                        # [B, 44]; columns 26.. = d/d action
                dz = new(B, A)
                _chk(lib.plentd3_dtanh(_p(dsa), _p(a_pi), _p(dz), float(ag.max_action), B, st))
                self._wgrad(dz, p2, ac.fc3.weight.grad, ac.fc3.bias.grad)
                dp2 = torch.mm(dz, ac.fc3.weight)
                _chk(lib.plentd3_relu_mask(_p(dp2), _p(p2), B, H, H, st))
                self._wgrad(dp2, p1, ac.fc2.weight.grad, ac.fc2.bias.grad)
                dp1 = torch.mm(dp2, ac.fc2.weight)
                _chk(lib.plentd3_relu_mask(_p(dp1), _p(p1), B, H, H, st))
                self._wgrad(dp1, s, ac.fc1.weight.grad, ac.fc1.bias.grad)

    def polyak(self):
        self._tp_epoch += 1          # (targets change outside the fused small-batch step: its packed copies are stale)
        """target = tau * online + (1 - tau) * target for critic then actor (td3.py:348-356), one pass per network over the flat buffers."""
        ag, st = self.agent, self._stream()
        for flat, tflat in ((ag._critic_flat, ag._critic_target_flat), (ag._actor_flat, ag._actor_target_flat)):
            _chk(self.lib.plentd3_polyak(_p(tflat.flat), _p(flat.flat), float(ag.tau), flat.flat.numel(), st))
