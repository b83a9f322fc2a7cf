"""PlenVecEnv: N PLEN walking environments stepped by one HIP launch on one MI355X.

Torch is plumbing here (device memory + the current stream); the step itself is libplenvec.so.
The single-env, reference-shaped facade is plen_ml_walk_amd.plen_env.PlenWalkEnv."""
import ctypes as C
import os
import torch
from . import _lib as L


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


class StepInfo(dict):
    """Lazy view of the step flags (keeps tiny elementwise kernels out of the rollout loop):
    "obs" observation to act on next, "flags" raw PLENVEC_DONE_* bits, "terminal" / "time_limit" masks."""

    def __init__(self, flags, cur_obs):
        dict.__init__(self, obs=cur_obs, flags=flags)

    def __missing__(self, key):
        f = dict.__getitem__(self, "flags")
        if key == "time_limit":
            v = (f & L.DONE_TIMELIMIT) != 0
        elif key == "terminal":          # compute_done() fired and the time limit did not: plen_td3.py:109-110 done_bool
            v = ((f & L.DONE_TERMINAL) != 0) & ((f & L.DONE_TIMELIMIT) == 0)
        elif key == "nonfinite":         # the non-finite guard reset this env (reported as a truncation, never as a terminal)
            v = (f & L.DONE_NONFINITE) != 0
        else:
            raise KeyError(key)
        self[key] = v
        return v


class PlenVecEnv(object):
    """Vector form of PlenWalkEnv (plen_bullet/src/plen_bullet/plen_env.py:22): `reset()` and
    `step(action)` act on all `num_envs` environments; tensors stay on the GPU.

    step(action[N,18] float32) -> (next_obs[N,26], reward[N], done[N] uint8 (nonzero = episode ended), info)
      next_obs is the post-step observation (the terminal one if the episode ended);
      info["obs"] is the observation to act on next (reset observation for envs that ended, because
      the env auto-resets like the reference driver does, plen_td3.py:122-133);
      info["terminal"] marks compute_done() terminations with the time limit masked out, i.e. the
      `done_bool` the reference stores in the replay buffer (plen_td3.py:109-110);
      info["time_limit"] marks gym TimeLimit truncations (plen_env.py:15-19)."""

    def __init__(self, num_envs, device=None, dtype=torch.float32, joint_act=False, auto_reset=True, cfg_overrides=None, out_buffers=None, model=None):
        if not torch.cuda.is_available():
            raise L.PlenvecError("PlenVecEnv needs a ROCm GPU (MI355X); there is no CPU fallback")
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        if self.device.type != "cuda":
            raise L.PlenvecError("PlenVecEnv device must be a cuda (HIP) device")
        self.lib = L.load()
        self.num_envs = int(num_envs)
        self.dtype = dtype
        cfg = L.default_cfg(joint_act)
        cfg.dtype = L.DTYPE_F64 if dtype == torch.float64 else L.DTYPE_F32
        cfg.auto_reset = int(bool(auto_reset))
        for k, v in (cfg_overrides or {}).items():
            setattr(cfg, k, v)
        self.cfg = cfg
        self.max_episode_steps = cfg.max_episode_steps
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        h = C.c_void_p()
        with torch.cuda.device(idx):
            if model is None:
                L.check(self.lib.plenvec_create(C.byref(cfg), self.num_envs, idx, C.byref(h)))
            else:                         # a _lib.PlenModel: the same tree with other numbers (plenvec_create_from_model)
                L.check(self.lib.plenvec_create_from_model(C.byref(model), C.byref(cfg), self.num_envs, idx, C.byref(h)))
        self.h = h
        n, dev = self.num_envs, self.device
        if out_buffers is None:
            self._next_obs = torch.empty(n, L.OBS, dtype=dtype, device=dev)
            self._cur_obs = torch.empty(n, L.OBS, dtype=dtype, device=dev)
            self._reward = torch.empty(n, dtype=dtype, device=dev)
            self._done = torch.empty(n, dtype=torch.uint8, device=dev)
        else:                     # slices of a larger batch's outputs (PlenVecEnvPipelined)
            self._next_obs, self._cur_obs, self._reward, self._done = out_buffers
            assert all(t.is_contiguous() and t.shape[0] == n for t in out_buffers)

    def close(self):
        if getattr(self, "h", None):
            self.lib.plenvec_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    # ---- gym-like surface ---------------------------------------------------------------
    def reset(self, mask=None):
        m = None
        if mask is not None:
            m = mask.to(device=self.device, dtype=torch.uint8).contiguous()
        L.check(self.lib.plenvec_reset(self.h, _ptr(m), _ptr(self._cur_obs), self._stream()))
        return self._cur_obs

    def step(self, action, out=None):
        a = action
        if a.dtype != torch.float32 or not a.is_contiguous() or a.device != self.device:
            a = a.to(device=self.device, dtype=torch.float32).contiguous()
        assert a.shape == (self.num_envs, L.ACT)
        L.check(self.lib.plenvec_step(self.h, _ptr(a), _ptr(self._next_obs), _ptr(self._reward), _ptr(self._done),
                                      _ptr(self._cur_obs), self._stream()))
        flags = self._done
        return self._next_obs, self._reward, flags, StepInfo(flags, self._cur_obs)

    def step2(self, action):
        """plenvec_step2: (next_obs, reward, done, trunc) with done / trunc as separate 0/1 bytes (SURVEY 8(b)'s pair)."""
        a = action.to(device=self.device, dtype=torch.float32).contiguous()
        if getattr(self, "_trunc", None) is None:
            self._trunc = torch.empty(self.num_envs, dtype=torch.uint8, device=self.device)
        L.check(self.lib.plenvec_step2(self.h, _ptr(a), _ptr(self._next_obs), _ptr(self._reward), _ptr(self._done), _ptr(self._trunc),
                                       _ptr(self._cur_obs), self._stream()))
        return self._next_obs, self._reward, self._done, self._trunc

    # ---- state access (parity tests) ------------------------------------------------------
    def get_state(self):
        s = torch.empty(self.num_envs, L.STATE, dtype=self.dtype, device=self.device)
        L.check(self.lib.plenvec_get_state(self.h, _ptr(s), self._stream()))
        return s

    def set_state(self, state):
        s = state.to(device=self.device, dtype=self.dtype).contiguous()
        assert s.shape == (self.num_envs, L.STATE)
        L.check(self.lib.plenvec_set_state(self.h, _ptr(s), self._stream()))

    def get_aux(self):
        a = torch.empty(self.num_envs, 8, dtype=torch.int32, device=self.device)
        L.check(self.lib.plenvec_get_aux(self.h, _ptr(a), self._stream()))
        return a

    def debug_substeps(self, targets, nsub=1, dump=False):
        t = targets.to(device=self.device, dtype=self.dtype).contiguous()
        assert t.shape == (self.num_envs, L.ACT)
        d = torch.zeros(self.num_envs, L.DUMP, dtype=self.dtype, device=self.device) if dump else None
        L.check(self.lib.plenvec_debug_substeps(self.h, _ptr(t), int(nsub), _ptr(d), self._stream()))
        return d

    def set_params(self, mass_scale=None, lateral_friction=None):
        ms = mass_scale.to(device=self.device, dtype=self.dtype).contiguous() if mass_scale is not None else None
        mu = lateral_friction.to(device=self.device, dtype=self.dtype).contiguous() if lateral_friction is not None else None
        L.check(self.lib.plenvec_set_params(self.h, _ptr(ms), _ptr(mu), self._stream()))

    def nonfinite_count(self):
        """PLENVEC_DONE_NONFINITE events since construction (host-synchronous, diagnostic)."""
        c = C.c_int64(0)
        L.check(self.lib.plenvec_get_nonfinite_count(self.h, C.byref(c), self._stream()))
        return c.value

    def timing_begin(self):
        L.check(self.lib.plenvec_timing_begin(self.h, self._stream()))

    def timing_end(self):
        ms, n = C.c_double(0), C.c_int64(0)
        L.check(self.lib.plenvec_timing_end(self.h, self._stream(), C.byref(ms), C.byref(n)))
        return ms.value, n.value


_WORKER_STREAMS = {}
# first-use order of the role streams of a device (PLEN_STREAM_ROLE_ORDER overrides it for experiments; "x" = a throw-away stream)
_ROLE_ORDER = "2,3,0,1,update,side"
_ROLE_PRIORITY = {"update": -1}
if os.environ.get("PLEN_ROLE_PRIORITY"):          # experiments: "update:0,0:-1,1:-1" (HIP dispatch priorities per role; -1 = high)
    _ROLE_PRIORITY = {k: int(v) for k, v in (kv.split(":") for kv in os.environ["PLEN_ROLE_PRIORITY"].split(","))}


def worker_stream(device, role):
    """The process-wide HIP stream of one concurrent role: sub-batch k of a pipelined env / collector k of a trainer (role k = 0, 1, ...), the
    trainer's update ("update", dispatch priority -1: ~50 short kernels that must slip in beside two long env launches) and its side stream.

    Why roles instead of torch.cuda.Stream() per object (measured with scripts/gpu_queue_map.py and gpu_stream_reuse_probe.py):
    * the HIP runtime gives a stream its hardware queue at FIRST USE, GPU_MAX_HW_QUEUES (16 here, one of them the null stream's) in all, and streams
      that share a queue serialise; torch hands streams out round-robin from a pool of 32, so a process that builds several pipelined envs /
      trainers one after the other (bench.py's legs) ends up with two concurrently used streams on one queue (4 x 1024 f64 envs: 1.46 instead
      of 0.98 ms per step);
    * queues are spread over the 4 compute pipes in creation order, and a pipe serves its queues one at a time: the high-priority update queue on
      the pipe of a collector's queue takes the pipelined trainer from 0.74 to 1.9 ms per step, two collectors on one pipe cost 20%.
    So the role streams of a device are created and first used together, in an order that puts {0, 1, 2, 3} on four different pipes and
    {0, 1, update} on three (update shares 2's: a 4-sub-batch env and a trainer's update never run together), whatever was created before."""
    device = torch.device(device)
    di = device.index if device.index is not None else torch.cuda.current_device()
    device = torch.device("cuda", di)
    if (di, "0") not in _WORKER_STREAMS:
        import os
        with torch.cuda.stream(torch.cuda.default_stream(device)):
            torch.zeros(1, device=device)                  # the null stream's queue first (measured: created after the role streams' queues, it lands on a collector's pipe: 0.98 instead of 0.74 ms)
        torch.cuda.synchronize(device)
        for tok in os.environ.get("PLEN_STREAM_ROLE_ORDER", _ROLE_ORDER).split(","):
            st = torch.cuda.Stream(device=device, priority=_ROLE_PRIORITY.get(tok, 0))
            with torch.cuda.stream(st):
                torch.zeros(1, device=device)              # first use: the hardware queue is assigned now
            _WORKER_STREAMS[(di, tok)] = st if tok != "x" else None
        torch.cuda.synchronize(device)
    key = (di, str(role))
    st = _WORKER_STREAMS.get(key)
    if st is None:
        st = _WORKER_STREAMS[key] = torch.cuda.Stream(device=device, priority=_ROLE_PRIORITY.get(str(role), 0))
    return st


class PlenVecEnvPipelined(object):
    """`num_envs` environments as `groups` independent sub-batches, each with its own libplenvec handle and HIP stream.

    Why: one launch lasts as long as its slowest wave (both feet planted, 50 solver iterations: ~2x the mean wave), so a
    single-stream loop leaves ~30 % of the wave slots idle at every step boundary.  Environments are independent, so
    sub-batch A's step t+1 need not wait for sub-batch B's step t: with two streams the tail of one launch overlaps the
    body of the other (+17 % env-steps/s at 4096 envs, measured).  This is the usual asynchronous sub-batch mode of vector
    environments; each environment still advances one control step per `step` call with its own action.

    step_async(action) enqueues the sub-batches behind the caller's current stream (so `action` may have been produced on it)
    and returns immediately; the outputs (shared [N, ...] tensors, group g owning rows [g*N/G, (g+1)*N/G)) may be read after
    sync().  step() = step_async() + sync().  Outputs are overwritten by the next step: consume or copy them first."""

    def __init__(self, num_envs, groups=2, device=None, dtype=torch.float32, **kw):
        assert num_envs % groups == 0, "num_envs must be a multiple of groups"
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.num_envs, self.groups, self.dtype = int(num_envs), int(groups), dtype
        n, g, dev = self.num_envs, self.groups, self.device
        self.n_sub = n // g
        self._next_obs = torch.empty(n, L.OBS, dtype=dtype, device=dev)
        self._cur_obs = torch.empty(n, L.OBS, dtype=dtype, device=dev)
        self._reward = torch.empty(n, dtype=dtype, device=dev)
        self._done = torch.empty(n, dtype=torch.uint8, device=dev)
        sl = [slice(i * self.n_sub, (i + 1) * self.n_sub) for i in range(g)]
        self._slices = sl
        self.envs = [PlenVecEnv(self.n_sub, device=dev, dtype=dtype, out_buffers=(self._next_obs[s], self._cur_obs[s], self._reward[s], self._done[s]), **kw)
                     for s in sl]
        self.streams = [worker_stream(dev, k) for k in range(g)]
        self._ev_in = torch.cuda.Event()
        self._ev_out = [torch.cuda.Event() for _ in range(g)]
        self.max_episode_steps = self.envs[0].max_episode_steps

    def close(self):
        for e in self.envs:
            e.close()

    def reset(self, mask=None):
        self.sync()
        for e, s in zip(self.envs, self._slices):
            e.reset(None if mask is None else mask[s])
        return self._cur_obs

    def step_async(self, action):
        a = action
        if a.dtype != torch.float32 or not a.is_contiguous() or a.device != self.device:
            a = a.to(device=self.device, dtype=torch.float32).contiguous()
        assert a.shape == (self.num_envs, L.ACT)
        cur = torch.cuda.current_stream(self.device)
        self._ev_in.record(cur)
        for e, s, st, ev in zip(self.envs, self._slices, self.streams, self._ev_out):
            st.wait_event(self._ev_in)
            a.record_stream(st)
            with torch.cuda.stream(st):
                e.step(a[s])
                ev.record(st)

    def sync(self):
        cur = torch.cuda.current_stream(self.device)
        for ev in self._ev_out:
            cur.wait_event(ev)

    def step(self, action):
        self.step_async(action)
        self.sync()
        return self._next_obs, self._reward, self._done, StepInfo(self._done, self._cur_obs)

    def outputs(self):
        """(next_obs, reward, done, info) of the last step; valid after sync()."""
        return self._next_obs, self._reward, self._done, StepInfo(self._done, self._cur_obs)

    def set_params(self, mass_scale=None, lateral_friction=None):
        """Per-env domain randomisation (BASELINE.json configs[4]); tensors of length num_envs."""
        self.sync()
        for e, s in zip(self.envs, self._slices):
            e.set_params(None if mass_scale is None else mass_scale[s], None if lateral_friction is None else lateral_friction[s])

    def nonfinite_count(self):
        self.sync()
        return sum(e.nonfinite_count() for e in self.envs)

    def get_state(self):
        self.sync()
        return torch.cat([e.get_state() for e in self.envs], 0)

    def get_aux(self):
        self.sync()
        return torch.cat([e.get_aux() for e in self.envs], 0)
