"""TD3 on PyTorch-ROCm with the reference's surface.

Mirrors plen_ros/src/plen_ros_helpers/td3.py of the reference (same class / method names, argument
meaning and defaults), so `from plen_ros_helpers.td3 import ReplayBuffer, TD3Agent, evaluate_policy`
callers (plen_bullet/src/plen_td3.py:5) run unchanged through plen_ml_walk_amd/compat/.  What differs
is underneath:
  * ReplayBuffer keeps the transitions in device-resident ring tensors (reference: a Python list of
    tuples re-uploaded inside an O(B^2) loop, td3.py:166-193); `add` keeps the one-transition call,
    `add_batch` takes a whole vector step straight from PlenVecEnv without leaving the GPU;
  * TD3Agent runs on the rank-local device (reference hard-codes "cuda:1", td3.py:173,221), exposes
    a batched `select_action_batch`, and, when torch.distributed is initialised with world_size > 1,
    all-reduces the critic / actor gradients through one flat bucket per network (RCCL over xGMI on
    MI355X; gloo in the CPU tests) -- the data-parallel step named in BASELINE.json's north_star.
"""
import copy
import os
import pickle

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


def default_device():
    """The rank-local accelerator (one process per GPU), or the CPU when there is none."""
    if torch.cuda.is_available():
        return torch.device("cuda", int(os.environ.get("LOCAL_RANK", torch.cuda.current_device())))
    return torch.device("cpu")


class Actor(nn.Module):
    """26 -> 256 -> 256 -> 18, ReLU, ReLU, tanh * max_action (reference td3.py:19-57; same layer names)."""

    def __init__(self, state_dim, action_dim, max_action):
        super(Actor, self).__init__()
        self.fc1 = nn.Linear(state_dim, 256)
        self.fc2 = nn.Linear(256, 256)
        self.fc3 = nn.Linear(256, action_dim)
        self.max_action = max_action

    def forward(self, state):
        a = F.relu(self.fc1(state))
        a = F.relu(self.fc2(a))
        return self.max_action * torch.tanh(self.fc3(a))


class Critic(nn.Module):
    """Twin Q networks 44 -> 256 -> 256 -> 1 (reference td3.py:60-117; fc1..fc3 = Q1, fc4..fc6 = Q2)."""

    def __init__(self, state_dim, action_dim):
        super(Critic, self).__init__()
        self.fc1 = nn.Linear(state_dim + action_dim, 256)
        self.fc2 = nn.Linear(256, 256)
        self.fc3 = nn.Linear(256, 1)
        self.fc4 = nn.Linear(state_dim + action_dim, 256)
        self.fc5 = nn.Linear(256, 256)
        self.fc6 = nn.Linear(256, 1)

    def forward(self, state, action):
        sa = torch.cat([state, action], 1)
        q1 = F.relu(self.fc1(sa))
        q1 = F.relu(self.fc2(q1))
        q1 = self.fc3(q1)
        q2 = F.relu(self.fc4(sa))
        q2 = F.relu(self.fc5(q2))
        q2 = self.fc6(q2)
        return q1, q2

    def Q1(self, state, action):
        sa = torch.cat([state, action], 1)
        q1 = F.relu(self.fc1(sa))
        q1 = F.relu(self.fc2(q1))
        return self.fc3(q1)


class _StorageView(object):
    """`len(buffer.storage)` and `buffer.storage[i]` keep working (reference td3.py:130,143-147,175-179)."""

    def __init__(self, buf):
        self._b = buf

    def __len__(self):
        return self._b.size

    def __getitem__(self, i):
        b = self._b
        if not -b.size <= i < b.size:
            raise IndexError(i)
        i = i % b.size
        return (b.state[i].cpu().numpy(), b.action[i].cpu().numpy(), b.next_state[i].cpu().numpy(),
                float(b.reward[i, 0]), float(1.0 - b.not_done[i, 0]))


class ReplayBuffer(object):
    """Experience replay with the reference's interface (td3.py:122-193) on device ring tensors.

    Reference semantics kept: tuples are (state, action, next_state, reward, done); the buffer grows
    to `max_size`, after which `ptr` walks round it overwriting the oldest entry (td3.py:143-147);
    `sample` draws indices uniformly WITH replacement and returns
    (state, action, next_state, reward[B,1], not_done[B,1]) as float32 tensors on the device."""

    def __init__(self, max_size=1000000, state_dim=26, action_dim=18, device=None):
        self.max_size = int(max_size)
        self.ptr = 0
        self.size = 0
        self.device = _indexed(device) if device is not None else default_device()
        my_path = os.path.abspath(os.path.dirname(__file__))
        self.buffer_path = os.path.join(my_path, "../replay_buffer")
        n, dev = self.max_size, self.device
        # one packed row per transition: state | action | next_state | reward | not_done (72 floats for PLEN); the five tensors of the
        # reference's interface are column views of it, so a sampled batch is ONE gather (td3_fused / csrc/td3_kernels.hip k_gather)
        sd, ad = int(state_dim), int(action_dim)
        self.data = torch.empty(n, 2 * sd + ad + 2, dtype=torch.float32, device=dev)
        self.state = self.data[:, 0:sd]
        self.action = self.data[:, sd:sd + ad]
        self.next_state = self.data[:, sd + ad:2 * sd + ad]
        self.reward = self.data[:, 2 * sd + ad:2 * sd + ad + 1]
        self.not_done = self.data[:, 2 * sd + ad + 1:2 * sd + ad + 2]
        self.storage = _StorageView(self)
        self._gen = None

    # -- writes ---------------------------------------------------------------------------
    def add(self, data):
        """One transition tuple (state, action, next_state, reward, done), reference td3.py:136-147."""
        s, a, s2, r, d = data
        if self.device.type == "cuda":
            # one packed row (state | action | next_state | reward | not_done) built on the host and written with ONE copy: five little tensors
            # and five slice assignments were 130 us of a 750 us drop-in loop iteration
            w = self.data.shape[1]
            row = np.empty(w, dtype=np.float32)
            sd, ad = self.state.shape[1], self.action.shape[1]
            row[0:sd] = np.asarray(s, dtype=np.float32).reshape(-1)
            row[sd:sd + ad] = np.asarray(a, dtype=np.float32).reshape(-1)
            row[sd + ad:2 * sd + ad] = np.asarray(s2, dtype=np.float32).reshape(-1)
            row[w - 2] = np.float32(np.asarray(r, dtype=np.float64).reshape(-1)[0])
            row[w - 1] = np.float32(1.0) - np.float32(np.asarray(d, dtype=np.float64).reshape(-1)[0])
            start = self.size if self.size < self.max_size else self.ptr
            self.data[start].copy_(torch.from_numpy(row))
            if self.size < self.max_size:
                self.size += 1
                self.ptr = 0
            else:
                self.ptr = (self.ptr + 1) % self.max_size
            return
        t = lambda x: torch.as_tensor(np.asarray(x, dtype=np.float32)).reshape(1, -1)
        self.add_batch(t(s), t(a), t(s2), t(r), t(d))

    def add_batch(self, state, action, next_state, reward, done):
        """A whole vector step: tensors [N,26], [N,18], [N,26], [N] or [N,1], [N] or [N,1] (done = 1.0 terminal)."""
        n = state.shape[0]
        if n > self.max_size:
            raise ValueError("batch larger than the buffer")
        dev = self.device
        f = lambda x, w: x.to(device=dev, dtype=torch.float32).reshape(n, w)
        state, action, next_state = f(state, self.state.shape[1]), f(action, self.action.shape[1]), f(next_state, self.state.shape[1])
        reward, not_done = f(reward, 1), 1.0 - f(done, 1)
        # the reference appends until full, then overwrites at ptr: both are "write at (ptr_or_size)"
        start = self.size if self.size < self.max_size else self.ptr
        first = min(n, self.max_size - start)
        for dst, src in ((self.state, state), (self.action, action), (self.next_state, next_state),
                         (self.reward, reward), (self.not_done, not_done)):
            dst[start:start + first] = src[:first]
            if first < n:
                dst[0:n - first] = src[first:]
        if self.size < self.max_size:
            grown = min(self.max_size, self.size + n)
            overflow = self.size + n - grown
            self.size = grown
            self.ptr = overflow % self.max_size if grown == self.max_size else 0
        else:
            self.ptr = (self.ptr + n) % self.max_size

    def size_on_device(self):
        """Number of valid rows as an int64 device scalar (what the in-kernel sampling of the fused update reads; refreshed when the size changed)."""
        if getattr(self, "_size_dev", None) is None:
            self._size_dev = torch.zeros((), dtype=torch.long, device=self.device)
            self._size_dev_host = None
        if self._size_dev_host != self.size:           # one persistent scalar, refilled in place (no allocation and no pageable host-to-device copy per call
            self._size_dev.fill_(self.size)            # while the buffer is still filling: 1e6 train() calls in the reference's recipe)
            self._size_dev_host = self.size
        return self._size_dev

    # -- reads ----------------------------------------------------------------------------
    def sample(self, batch_size, ind=None):
        """Uniform with replacement (reference: np.random.randint(0, len, size=B), td3.py:175).
        `ind` (LongTensor / ndarray) overrides the draw; otherwise NumPy's global RNG is used exactly
        like the reference when `PLEN_TD3_NUMPY_RNG=1`, else a device generator (no host round trip)."""
        if self.size == 0:
            raise ValueError("cannot sample an empty buffer")
        if ind is None:
            if os.environ.get("PLEN_TD3_NUMPY_RNG") == "1":
                ind = torch.as_tensor(np.random.randint(0, self.size, size=batch_size))
            else:
                ind = torch.randint(0, self.size, (batch_size,), device=self.device, generator=self._gen)
        ind = torch.as_tensor(ind).to(self.device, dtype=torch.long)
        return (self.state[ind], self.action[ind], self.next_state[ind], self.reward[ind], self.not_done[ind])

    def seed(self, seed):
        self._gen = torch.Generator(device=self.device)
        self._gen.manual_seed(int(seed))

    # -- persistence (reference td3.py:149-164: pickle of the tuple list; here plain arrays) ---
    def save(self, iterations):
        if not os.path.exists(self.buffer_path):
            os.makedirs(self.buffer_path)
        path = self.buffer_path + "/" + "replay_buffer_" + str(iterations) + ".data"
        n = self.size
        payload = {"format": "plen_ml_walk_amd.replay.v1", "size": n, "ptr": self.ptr, "max_size": self.max_size,
                   "state": self.state[:n].cpu().numpy(), "action": self.action[:n].cpu().numpy(),
                   "next_state": self.next_state[:n].cpu().numpy(), "reward": self.reward[:n].cpu().numpy(),
                   "not_done": self.not_done[:n].cpu().numpy()}
        with open(path, "wb") as fh:
            np.savez(fh, **payload)

    def load(self, iterations):
        path = self.buffer_path + "/" + "replay_buffer_" + str(iterations) + ".data"
        with open(path, "rb") as fh:
            z = np.load(fh, allow_pickle=False)          # never unpickle an untrusted buffer file
            n = int(z["size"])
            if n > self.max_size:
                raise ValueError("saved buffer larger than max_size")
            self.state[:n] = torch.as_tensor(z["state"]); self.action[:n] = torch.as_tensor(z["action"])
            self.next_state[:n] = torch.as_tensor(z["next_state"]); self.reward[:n] = torch.as_tensor(z["reward"])
            self.not_done[:n] = torch.as_tensor(z["not_done"])
            self.size = n
            self.ptr = int(z["ptr"]) % self.max_size


def _flat_order(module):
    """Parameters of a network in the order they are laid out in its flat buffers.  Critic: the twin Q networks' first layers (fc1, fc4)
    and the biases of their first and second layers are adjacent, so that [fc1.weight; fc4.weight] is ONE [512, 44] matrix "W14" (one
    GEMM for both networks' first layers, td3_fused.py) and [fc1.bias; fc4.bias] = "b14", [fc2.bias; fc5.bias] = "b25" single vectors."""
    if isinstance(module, Critic):
        m = module
        return [m.fc1.weight, m.fc4.weight, m.fc1.bias, m.fc4.bias, m.fc2.weight, m.fc5.weight, m.fc2.bias, m.fc5.bias,
                m.fc3.weight, m.fc6.weight, m.fc3.bias, m.fc6.bias]
    return [p for p in module.parameters() if p.requires_grad]


def _stacked_views(module, flat):
    if not isinstance(module, Critic):
        return {}
    n1 = module.fc1.weight.numel()
    h, k = module.fc1.weight.shape
    o_b14 = 2 * n1
    o_b25 = o_b14 + 2 * h + 2 * module.fc2.weight.numel()
    return {"W14": flat[0:2 * n1].view(2 * h, k), "b14": flat[o_b14:o_b14 + 2 * h], "b25": flat[o_b25:o_b25 + 2 * h]}


def _indexed(device):
    """torch.device('cuda') -> torch.device('cuda', current index): tensors report an indexed device, so an index-less one never compares equal to
    theirs (TD3Agent(device="cuda") would silently take the autograd iteration in train())."""
    device = torch.device(device)
    if device.type == "cuda" and device.index is None and torch.cuda.is_available():
        return torch.device("cuda", torch.cuda.current_device())
    return device


class _FlatParams(object):
    """All parameters of one network as views into ONE contiguous buffer (layout: _flat_order): Polyak averaging is one pass over two flat
    buffers, and the critics' stacked first-layer views exist.  state_dict keys, shapes and values are unchanged."""

    def __init__(self, module):
        params = _flat_order(module)
        self.flat = torch.empty(sum(p.numel() for p in params), dtype=params[0].dtype, device=params[0].device)
        off = 0
        with torch.no_grad():
            for p in params:
                v = self.flat[off:off + p.numel()].view_as(p)
                v.copy_(p.data)
                p.data = v
                off += p.numel()
        self.views = _stacked_views(module, self.flat)


class _FlatGrads(object):
    """All gradients of one network as views into ONE contiguous buffer (same layout as _FlatParams), so a data-parallel step is
    a single in-place all-reduce (critic 155 138 floats, actor 77 330: latency-bound messages, one
    bucket each)."""

    def __init__(self, module, data_parallel=True):
        self.data_parallel = data_parallel
        params = _flat_order(module)
        self.params = params
        total = sum(p.numel() for p in params)
        self.flat = torch.zeros(total, dtype=params[0].dtype, device=params[0].device)
        off = 0
        for p in params:
            p.grad = self.flat[off:off + p.numel()].view_as(p)
            off += p.numel()
        self.views = _stacked_views(module, self.flat)
        self.dirty = False         # set by the autograd path whenever it may have left gradients in the bucket (FusedTD3._zero_grads reads it)

    def zero(self):
        self.flat.zero_()
        self.dirty = False

    def will_reduce(self):
        """Does all_reduce_mean() issue a collective in this process (several ranks, or the forced one-rank collective)?"""
        import torch.distributed as dist
        return bool(self.data_parallel and dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or os.environ.get("PLEN_TD3_FORCE_COLLECTIVES") == "1"))

    def all_reduce_mean(self):
        import torch.distributed as dist
        # PLEN_TD3_FORCE_COLLECTIVES=1: issue the collective at world size 1 too (one-GPU boxes: exercises RCCL's init, stream ordering and
        # graph capture on the real backend; the reduction of one rank is the identity)
        if self.data_parallel and dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or os.environ.get("PLEN_TD3_FORCE_COLLECTIVES") == "1"):
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
            self.flat.div_(dist.get_world_size())


def td3_critic_backward(agent, batch, noise=None):
    """First half of a TD3 iteration (reference td3.py:274-323): target-policy smoothing, clipped double-Q target, critic loss,
    gradients into the critic's flat bucket.  `noise` overrides the randn_like draw (golden tests feed the reference's)."""
    state, action, next_state, reward, not_done = batch
    with torch.no_grad():
        if noise is None:
            noise = torch.randn_like(action)
        noise = (noise * agent.policy_noise).clamp(-agent.noise_clip, agent.noise_clip)
        next_action = (agent.actor_target(next_state) + noise).clamp(-agent.max_action, agent.max_action)
        target_Q1, target_Q2 = agent.critic_target(next_state, next_action)
        target_Q = torch.min(target_Q1, target_Q2)
        target_Q = reward + not_done * agent.discount * target_Q
    current_Q1, current_Q2 = agent.critic(state, action)
    critic_loss = F.mse_loss(current_Q1, target_Q) + F.mse_loss(current_Q2, target_Q)
    agent._critic_grads.zero()
    critic_loss.backward()
    agent._critic_grads.dirty = True
    return critic_loss.detach()


def td3_actor_backward(agent, batch):
    """Delayed policy gradient (reference td3.py:334-341) into the actor's flat bucket."""
    state = batch[0]
    actor_loss = -agent.critic.Q1(state, agent.actor(state)).mean()
    agent._actor_grads.zero()
    actor_loss.backward()                          # also writes critic grads; they are zeroed before their next use
    agent._actor_grads.dirty = agent._critic_grads.dirty = True
    return actor_loss.detach()


def td3_polyak(agent):
    """tau*p + (1-tau)*t for critic then actor targets (reference td3.py:348-356)."""
    with torch.no_grad():
        for net, tgt in ((agent.critic, agent.critic_target), (agent.actor, agent.actor_target)):
            ps, ts = list(net.parameters()), list(tgt.parameters())
            torch._foreach_mul_(ts, 1 - agent.tau)
            torch._foreach_add_(ts, ps, alpha=agent.tau)


def td3_update(agent, batch, with_policy, noise=None, all_reduce=True):
    """THE TD3 iteration (reference td3.py:259-356) on an explicit batch: the one implementation behind TD3Agent.train, the eager vector
    trainer and the hipGraph-captured trainer (train_vec.py), so that one golden test covers them all.  With torch.distributed
    initialised the two flat gradient buckets are averaged over ranks before their optimiser steps."""
    loss = td3_critic_backward(agent, batch, noise)
    if all_reduce:
        agent._critic_grads.all_reduce_mean()
    agent.critic_optimizer.step()
    agent.last_critic_loss = loss
    if with_policy:
        aloss = td3_actor_backward(agent, batch)
        if all_reduce:
            agent._actor_grads.all_reduce_mean()
        agent.actor_optimizer.step()
        agent.last_actor_loss = aloss
        td3_polyak(agent)
    return loss


class TD3Agent(object):
    """Twin Delayed DDPG agent with the reference's constructor and methods (td3.py:196-376)."""

    def __init__(self, state_dim, action_dim, max_action, discount=0.99, tau=0.005, policy_noise=0.2, noise_clip=0.5,
                 policy_freq=2, device=None, lr=3e-4, data_parallel=True):
        self.data_parallel = data_parallel
        self.device = _indexed(device) if device is not None else default_device()
        self.actor = Actor(state_dim, action_dim, max_action).to(self.device)
        self.critic = Critic(state_dim, action_dim).to(self.device)
        self._broadcast_parameters()                       # every rank starts from rank 0's initialisation
        self.actor_target = copy.deepcopy(self.actor)
        self.critic_target = copy.deepcopy(self.critic)
        self._actor_flat, self._critic_flat = _FlatParams(self.actor), _FlatParams(self.critic)
        self._actor_target_flat, self._critic_target_flat = _FlatParams(self.actor_target), _FlatParams(self.critic_target)
        self.actor_optimizer = torch.optim.Adam(self.actor.parameters(), lr=lr)
        self.critic_optimizer = torch.optim.Adam(self.critic.parameters(), lr=lr)
        self._actor_grads = _FlatGrads(self.actor, data_parallel)
        self._critic_grads = _FlatGrads(self.critic, data_parallel)
        self.max_action = max_action
        self.discount = discount
        self.tau = tau
        self.policy_noise = policy_noise
        self.noise_clip = noise_clip
        self.policy_freq = policy_freq
        self.total_it = 0
        self.fused_train = None      # train(): None = fused iteration where possible (PLEN_TD3_FUSED_TRAIN=0 turns it off), False = always the autograd iteration
        self._fused = None
        self.fused_select = None     # select_action(): None = one-kernel forward on a HIP device (PLEN_TD3_FUSED_SELECT=0 turns it off), False = torch layers
        self._select_state = None
        self.last_critic_loss = None          # 0-d device tensor of the last train() call; on the fused small-batch path a view of a persistent word, valid until the next call (copy to keep)
        self.last_actor_loss = None

    def _broadcast_parameters(self):
        import torch.distributed as dist
        # PLEN_TD3_FORCE_COLLECTIVES=1: issue the collective at world size 1 too (one-GPU boxes: exercises RCCL's init, stream ordering and
        # graph capture on the real backend; the reduction of one rank is the identity)
        if self.data_parallel and dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or os.environ.get("PLEN_TD3_FORCE_COLLECTIVES") == "1"):
            for p in list(self.actor.parameters()) + list(self.critic.parameters()):
                dist.broadcast(p.data, src=0)

    # ---- acting -------------------------------------------------------------------------
    def select_action(self, state):
        """state: ndarray (state_dim,) -> ndarray (action_dim,), reference td3.py:243-257."""
        if self.device.type == "cuda" and self.fused_select is not False and os.environ.get("PLEN_TD3_FUSED_SELECT", "1") == "1" and self._plen_shaped() and int(np.size(state)) == 26:
            return self._select_action_kernel(state)
        state = torch.as_tensor(np.asarray(state, dtype=np.float32).reshape(1, -1), device=self.device)
        with torch.no_grad():
            return self.actor(state).cpu().numpy().flatten()

    def _plen_shaped(self):
        """The networks the HIP kernels are written for: actor 26-256-256-18, critics 44-256-256-1 (td3.py:19-117 at PLEN's dimensions)."""
        a, c = self.actor, self.critic
        return (a.fc1.in_features, a.fc1.out_features, a.fc2.out_features, a.fc3.out_features, c.fc1.in_features, c.fc1.out_features) == (26, 256, 256, 18, 44, 256)

    def _select_action_kernel(self, state):
        """select_action as one copy in, ONE kernel (plentd3_actor_rows: the three layers on the matrix cores, exploration sigma 0) and one copy out,
        through pinned host buffers: ~60 us instead of ~160 (six launches + pageable copies).  Same arithmetic up to f32 summation order."""
        from . import td3_fused as F
        st = self._select_state
        if st is None:
            lib = F.load()
            dev = self.device
            st = self._select_state = dict(lib=lib, h_in=torch.empty(1, 26, dtype=torch.float32).pin_memory(), h_out=torch.empty(1, 18, dtype=torch.float32).pin_memory(),
                                           d_in=torch.empty(1, 26, dtype=torch.float32, device=dev), d_out=torch.empty(1, 18, dtype=torch.float32, device=dev),
                                           p1=torch.empty(1, 256, dtype=torch.float32, device=dev), p2=torch.empty(1, 256, dtype=torch.float32, device=dev),
                                           rng=torch.zeros(2, dtype=torch.long, device=dev))
        st["h_in"].numpy()[0, :] = np.asarray(state, dtype=np.float32).reshape(-1)
        st["d_in"].copy_(st["h_in"], non_blocking=True)
        ac, a = self.actor, F.ActorRowsArgs()
        a.a_w1, a.a_b1, a.a_w2, a.a_b2, a.a_w3, a.a_b3 = (t.data_ptr() for t in (ac.fc1.weight, ac.fc1.bias, ac.fc2.weight, ac.fc2.bias, ac.fc3.weight, ac.fc3.bias))
        a.state, a.rng, a.p1, a.p2, a.action = st["d_in"].data_ptr(), st["rng"].data_ptr(), st["p1"].data_ptr(), st["p2"].data_ptr(), st["d_out"].data_ptr()
        a.sigma, a.max_a, a.B = 0.0, float(self.max_action), 1
        F._chk(st["lib"].plentd3_actor_rows(F.C.byref(a), F.C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)))
        st["h_out"].copy_(st["d_out"], non_blocking=True)
        torch.cuda.current_stream(self.device).synchronize()
        return st["h_out"].numpy()[0].copy()

    def select_action_batch(self, state):
        """state: tensor [N, state_dim] on the device -> tensor [N, action_dim]; no host round trip."""
        with torch.no_grad():
            return self.actor(state.to(self.device, dtype=torch.float32))

    # ---- learning -----------------------------------------------------------------------
    def train(self, replay_buffer, batch_size=100):
        """One TD3 iteration, reference td3.py:259-356.

        On a HIP device with this module's device-resident ReplayBuffer the iteration is the hand-fused one (td3_fused.FusedTD3.update: for the
        reference's batch 100 three launches per critic update and two more per policy update instead of ~170, csrc/td3_team.hip) -- same
        arithmetic, with the replay indices and the target-smoothing noise drawn inside the kernels from the agent's counter-based stream instead of
        torch's / NumPy's generators.  The autograd iteration (td3_update: same order of operations and RNG calls as the reference) runs
        wherever that is not possible or not wanted: on the CPU, with `fused_train = False` or PLEN_TD3_FUSED_TRAIN=0, with PLEN_TD3_NUMPY_RNG=1,
        or when the buffer's `sample` is not this module's own (a subclass or an instance override: the caller's sampling is honoured)."""
        self.total_it += 1
        with_policy = self.total_it % self.policy_freq == 0
        fz = self._fused_for(replay_buffer)
        if fz is not None:
            self.last_critic_loss = fz.update(replay_buffer.data, int(batch_size), with_policy, all_reduce=self._critic_grads.will_reduce(),
                                              total=replay_buffer.size_on_device())
            return
        batch = replay_buffer.sample(batch_size)
        td3_update(self, batch, with_policy=with_policy)

    def _fused_for(self, replay_buffer):
        """The FusedTD3 behind train() (created on first use), or None where train() has to take the autograd iteration."""
        if self.fused_train is False or (self.fused_train is None and os.environ.get("PLEN_TD3_FUSED_TRAIN", "1") != "1"):
            return None
        if self.device.type != "cuda" or os.environ.get("PLEN_TD3_NUMPY_RNG") == "1" or not self._plen_shaped():
            return None
        if not (isinstance(replay_buffer, ReplayBuffer) and type(replay_buffer).sample is ReplayBuffer.sample and "sample" not in vars(replay_buffer)):
            return None
        data = getattr(replay_buffer, "data", None)
        if data is None or data.device != self.device or data.dtype != torch.float32 or data.dim() != 2 or data.shape[1] != 72 or replay_buffer.size < 1:
            return None
        if self._fused is None:
            from .td3_fused import FusedTD3
            self._fused = FusedTD3(self, seed=int(torch.initial_seed()) & 0x7fffffff)
            self._fused.enable_flat_adam()
        return self._fused

    # ---- checkpoints: the reference's four files per checkpoint (td3.py:358-376) -----------------
    def save(self, filename):
        # (parameters are views of one flat buffer per network: clone, so that each file holds its own tensors only)
        torch.save({k: v.clone() for k, v in self.critic.state_dict().items()}, filename + "_critic")
        torch.save(self.critic_optimizer.state_dict(), filename + "_critic_optimizer")
        torch.save({k: v.clone() for k, v in self.actor.state_dict().items()}, filename + "_actor")
        torch.save(self.actor_optimizer.state_dict(), filename + "_actor_optimizer")

    def load(self, filename, load_optimizers=True):
        """Targets are NOT restored, exactly like the reference (td3.py:366-376).
        Files are read with weights_only=True; the reference's legacy optimizer pickles cannot be read
        that way, so a missing file or a legacy pickle is skipped with a warning instead of being
        force-unpickled; a readable file whose contents do not fit raises."""
        self.critic.load_state_dict(torch.load(filename + "_critic", map_location=self.device, weights_only=True))
        self.actor.load_state_dict(torch.load(filename + "_actor", map_location=self.device, weights_only=True))
        if load_optimizers:
            for opt, suffix in ((self.critic_optimizer, "_critic_optimizer"), (self.actor_optimizer, "_actor_optimizer")):
                try:
                    sd = torch.load(filename + suffix, map_location=self.device, weights_only=True)
                except (FileNotFoundError, pickle.UnpicklingError) as ex:       # absent file, or the reference's legacy pickle that weights_only refuses
                    print("TD3Agent.load: optimizer state %s not loaded (%s)" % (suffix, type(ex).__name__))
                    continue
                opt.load_state_dict(sd)       # a readable but mismatched optimizer file is an error, not a fresh start

    def load_arrays(self, arrays):
        """Load actor./critic. arrays (e.g. tests/golden/policy_3229999.npz, the reference's shipped policy)."""
        for net, prefix in ((self.actor, "actor."), (self.critic, "critic.")):
            sd = {k[len(prefix):]: torch.as_tensor(np.asarray(v)) for k, v in arrays.items() if k.startswith(prefix)}
            if sd:
                net.load_state_dict(sd)


def evaluate_policy(policy, env_name, seed, eval_episodes=10, render=False):
    """Average undiscounted return over `eval_episodes` episodes, reference td3.py:381-415."""
    from . import gym_compat
    eval_env = gym_compat.make(env_name, render=render)
    eval_env.seed(seed + 100)
    avg_reward = 0.0
    for _ in range(eval_episodes):
        state, done = eval_env.reset(), False
        while not done:
            action = policy.select_action(np.array(state))
            state, reward, done, _ = eval_env.step(action)
            avg_reward += reward
    avg_reward /= eval_episodes
    print("---------------------------------------")
    print("Evaluation over {} episodes: {}".format(eval_episodes, avg_reward))
    print("---------------------------------------")
    if render:
        eval_env.close()
    return avg_reward
