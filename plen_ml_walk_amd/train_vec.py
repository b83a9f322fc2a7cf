#!/usr/bin/env python
"""Vectorised TD3 training: N PLEN envs per GPU stepped by one HIP launch, actor/critic/replay on the
same device, one process per GPU, gradients all-reduced over RCCL (BASELINE.json configs[2], [3]).

What the reference's loop (plen_bullet/src/plen_td3.py:83-157) becomes with N envs per step:
  * warm-up: uniform random actions until the rank-local buffer holds >= start_timesteps transitions
    (reference: 1e4 single steps; at N=4096 that is 3 vector steps);
  * acting: actor(state) + N(0, expl_noise) clipped to [-1, 1]  (plen_td3.py:101-104), batched on device;
  * storing: (state, action, next_state, reward, done_bool) with done_bool = terminal AND NOT time-limit
    (plen_td3.py:109-110); next_state is the terminal observation, the next `state` is the reset
    observation of envs that ended (auto-reset inside the step kernel);
  * learning: `updates_per_step` TD3Agent.train() calls of `batch_size` per vector step.  The reference
    does 1 update of 100 per single env step (update-to-data 1 : 1); that ratio cannot be kept at
    millions of env-steps/s, so the default here is 1 update of batch 4096 per vector step of 4096 envs
    (same samples-per-env-step, 1/4096 of the optimiser steps) -- state the choice when reporting.
  * ranks: envs and replay are rank-local; the flat critic / actor gradient buckets are averaged over
    ranks every update (TD3Agent), parameters start from rank 0's initialisation.
"""
import argparse
import json
import os
import time

import torch
import contextlib
import gc


@contextlib.contextmanager
def capture_graph(g, **kw):
    """torch.cuda.graph(g, **kw) with the cyclic garbage collector held off for the duration of the capture.  The trainers (and their optimiser / probe lambdas) sit in
    reference cycles, so an earlier trainer's CUDAGraph objects are destroyed whenever the collector happens to run -- and hipGraphDestroy DURING a stream capture is an
    error ("operation not permitted when stream is capturing") that surfaces in a destructor and aborts the process (seen once in round 6: the driver's bench command,
    intermittent).  torch.cuda.graph collects garbage before it begins; nothing may be collected until it has ended."""
    was = gc.isenabled()
    gc.collect()
    gc.disable()
    try:
        with torch.cuda.graph(g, **kw):
            yield
    finally:
        if was:
            gc.enable()



def setup_distributed():
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    forced = os.environ.get("PLEN_TD3_FORCE_COLLECTIVES") == "1"          # a world-size-1 RCCL group (one-GPU boxes), see td3._FlatGrads
    if (world > 1 or forced) and not dist.is_initialized():
        os.environ.setdefault("MASTER_PORT", "29517"); os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # PLEN_DIST_BACKEND=gloo: development override (two ranks sharing ONE GPU cannot use RCCL; gloo moves CUDA tensors through the host)
        backend = os.environ.get("PLEN_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            lr = int(os.environ.get("LOCAL_RANK", "0")) % max(1, torch.cuda.device_count())
            torch.cuda.set_device(lr)
            dist.init_process_group(backend, device_id=torch.device("cuda", lr))
        else:
            dist.init_process_group(backend)
    from . import sharding
    return sharding.world_info()[0], world


def save_training(agent, trainer, prefix):
    """Checkpoint = the reference's four files (td3.py:358-366) + the two target networks (which the reference's
    save/load silently drops) + a small JSON with the loop counters."""
    agent.save(prefix)
    torch.save({k: v.clone() for k, v in agent.actor_target.state_dict().items()}, prefix + "_actor_target")
    torch.save({k: v.clone() for k, v in agent.critic_target.state_dict().items()}, prefix + "_critic_target")
    with open(prefix + "_trainer.json", "w") as fh:
        json.dump({"env_steps": int(trainer.env_steps), "grad_steps": int(trainer.grad_steps), "total_it": int(agent.total_it)}, fh)


def resume_training(agent, replay, prefix, replay_id=None):
    """Counterpart of plen_td3.py:57-69: load an existing policy (and optionally a replay buffer) if present."""
    counters = {"env_steps": 0, "grad_steps": 0, "total_it": 0}
    if os.path.exists(prefix + "_critic"):
        agent.load(prefix)
        for net, tgt, suffix in ((agent.actor, agent.actor_target, "_actor_target"), (agent.critic, agent.critic_target, "_critic_target")):
            if os.path.exists(prefix + suffix):
                tgt.load_state_dict(torch.load(prefix + suffix, map_location=agent.device, weights_only=True))
            else:                         # a reference-format checkpoint: start the targets at the loaded networks
                tgt.load_state_dict(net.state_dict())
        if os.path.exists(prefix + "_trainer.json"):
            with open(prefix + "_trainer.json") as fh:
                counters.update(json.load(fh))
        agent.total_it = counters["total_it"]
    if replay_id is not None and os.path.exists(replay.buffer_path + "/replay_buffer_" + str(replay_id) + ".data"):
        replay.load(replay_id)
    return counters


class VecTD3Trainer(object):
    def __init__(self, env, agent, replay, start_timesteps=10000, expl_noise=0.1, batch_size=4096, updates_per_step=1, seed=0):
        self.env, self.agent, self.replay = env, agent, replay
        self.start_timesteps = start_timesteps
        self.expl_noise = expl_noise
        self.batch_size = batch_size
        self.updates_per_step = updates_per_step
        dev = agent.device
        self.gen = torch.Generator(device=dev)
        self.gen.manual_seed(seed)
        self.state = env.reset().to(torch.float32).clone()
        self.env_steps = 0
        self.grad_steps = 0
        n = self.state.shape[0]
        self.ep_return = torch.zeros(n, device=dev)
        self.finished_returns = []

    def restore_counters(self, c):
        self.env_steps, self.grad_steps = int(c["env_steps"]), int(c["grad_steps"])

    def step(self):
        env, agent = self.env, self.agent
        n = self.state.shape[0]
        if self.replay.size < self.start_timesteps:
            action = torch.rand(n, 18, device=agent.device, generator=self.gen) * 2 - 1
        else:
            action = agent.select_action_batch(self.state)
            action = (action + torch.randn(action.shape, device=agent.device, generator=self.gen) * (agent.max_action * self.expl_noise)).clamp(-agent.max_action, agent.max_action)
        next_obs, reward, done, info = env.step(action)
        self.replay.add_batch(self.state, action, next_obs, reward, info["terminal"].to(torch.float32))
        self.ep_return += reward.to(torch.float32)
        ended = done != 0
        if len(self.finished_returns) < 64:                      # cheap running log, bounded
            self.finished_returns.append((self.ep_return * ended).sum() / ended.sum().clamp(min=1))
        self.ep_return = self.ep_return * (~ended)
        self.state = info["obs"].to(torch.float32).clone()
        self.env_steps += n
        if self.replay.size >= self.start_timesteps:
            for _ in range(self.updates_per_step):
                agent.train(self.replay, self.batch_size)
                self.grad_steps += 1


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=4096, help="environments per GPU")
    ap.add_argument("--steps", type=int, default=200, help="vector steps to run")
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--updates-per-step", type=int, default=1)
    ap.add_argument("--start-timesteps", type=int, default=10000)
    ap.add_argument("--replay", type=int, default=1000000)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--save", default=None, help="checkpoint prefix (reference 4-file layout)")
    ap.add_argument("--graphs", type=int, default=1, help="capture collect/update in hipGraphs (any number of ranks)")
    ap.add_argument("--schedule", default="sync", choices=["sync", "pipelined"], help="pipelined: actor/learner overlap on three HIP streams per rank "
                    "(PipelinedVecTD3Trainer: two half batches + the fused update; needs --graphs 1, --updates-per-step 1, an even --envs)")
    ap.add_argument("--resume", default=None, help="checkpoint prefix to continue from (4-file layout + <prefix>_trainer.json; plen_td3.py:57-69)")
    ap.add_argument("--buffer-path", default=None, help="directory of replay_buffer_<n>.data files (td3.py:128-131)")
    ap.add_argument("--save-replay", type=int, default=None, help="write the rank-local replay buffer as replay_buffer_<n>.data at the end")
    ap.add_argument("--load-replay", type=int, default=None, help="read replay_buffer_<n>.data before training (with --resume)")
    a = ap.parse_args(argv)
    from . import sharding
    from .vec_env import PlenVecEnv
    from .td3 import ReplayBuffer, TD3Agent
    import torch.distributed as dist
    rank, world = setup_distributed()
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")) % max(1, torch.cuda.device_count()))
    torch.cuda.set_device(dev)
    torch.manual_seed(a.seed)                                     # identical init on every rank (then broadcast anyway)
    pipelined = a.schedule == "pipelined" and a.graphs and a.updates_per_step == 1 and a.envs % 2 == 0
    envs = [PlenVecEnv(a.envs // 2, device=dev) for _ in range(2)] if pipelined else [PlenVecEnv(a.envs, device=dev)]
    env = envs[0]
    agent = TD3Agent(26, 18, 1.0, device=dev)
    replay = ReplayBuffer(a.replay, device=dev)
    replay.seed(a.seed + rank)
    if a.buffer_path:
        replay.buffer_path = a.buffer_path if world == 1 else os.path.join(a.buffer_path, "rank%d" % rank)
    resumed = None
    if a.resume:
        resumed = resume_training(agent, replay, a.resume, a.load_replay)
    torch.manual_seed(a.seed + 7919 * (rank + 1))                # from here on the global generator (target-policy smoothing noise) differs per rank
    if pipelined:
        tr = PipelinedVecTD3Trainer(envs, agent, replay, a.start_timesteps, 0.1, a.batch, seed=1000 + rank)
    elif a.graphs:
        tr = GraphedVecTD3Trainer(env, agent, replay, a.start_timesteps, 0.1, a.batch, a.updates_per_step, seed=1000 + rank)
    else:
        tr = VecTD3Trainer(env, agent, replay, a.start_timesteps, 0.1, a.batch, a.updates_per_step, seed=1000 + rank)
    if resumed:
        tr.restore_counters(resumed)
    # the pipelined loop has 6 update and 6 collect graph keys, each run eagerly twice and captured on its third use: ~22 steps pass before
    # everything replays, so a shorter warm-up would put eager runs, captures and device-wide syncs into the timed region
    for _ in range(max(a.warmup, 30) if pipelined else a.warmup):
        tr.step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    e0, g0, t0 = tr.env_steps, tr.grad_steps, time.perf_counter()
    for _ in range(a.steps):
        tr.step()
    if pipelined:
        tr.sync()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = sharding.max_over_ranks(time.perf_counter() - t0, dev)
    if rank == 0:
        print(json.dumps({"metric": "td3_env_steps_per_sec", "value": world * (tr.env_steps - e0) / dt, "unit": "env-steps/s",
                          "grad_steps_per_sec": (tr.grad_steps - g0) / dt, "n_gpus": world, "envs_per_gpu": a.envs, "batch": a.batch,
                          "updates_per_step": a.updates_per_step, "hip_graphs": bool(a.graphs), "allreduce_mode": getattr(tr, "allreduce_mode", None), "steps": a.steps, "ms_per_step": dt / a.steps * 1e3,
                          "critic_loss": float(agent.last_critic_loss) if agent.last_critic_loss is not None else None,
                          "schedule": "pipelined" if pipelined else "sync", "episodes": tr.episode_stats() if pipelined else None}))
        if a.save:
            save_training(agent, tr, a.save)
    if a.save_replay is not None:
        replay.save(a.save_replay)
    for e in envs:
        e.close()
    if world > 1:
        dist.destroy_process_group()



class GraphedVecTD3Trainer(object):
    """The same loop as VecTD3Trainer with the launch-bound parts captured in hipGraphs:
      collect graph : actor forward (+ exploration noise) or uniform random actions -> libplenvec step kernel
                      (launched on the capturing stream through the C ABI) -> ring write into the replay tensors
                      -> next state;
      update graphs : replay sampling on device -> td3.td3_update (the one TD3 iteration shared with TD3Agent.train),
                      Adam with capturable state.
    Write position / fill level of the ring live in device scalars that the graphs advance; the host mirrors the
    same arithmetic, so nothing synchronises.

    world_size > 1 (one rank per GPU): the update is cut at its two collectives into graph segments
        [sample + targets + critic backward] -> all-reduce(critic bucket) -> [critic Adam (+ actor backward)]
        -> all-reduce(actor bucket) -> [actor Adam + Polyak]
    with the RCCL all-reduces issued eagerly between the graph replays (`allreduce_mode` "eager-between-graphs"); set
    PLEN_TD3_CAPTURE_ALLREDUCE=1 to capture the collectives inside one graph instead ("captured").

    Every graph key is executed eagerly (on a side stream) the first two times it is needed and captured on the third: those
    eager runs ARE the loop's real iterations, so the collect/update cadence and policy_freq alternation are exactly the eager
    trainer's from the first step on."""

    def __init__(self, env, agent, replay, start_timesteps=10000, expl_noise=0.1, batch_size=4096, updates_per_step=1, seed=0, fused=True):
        import torch.distributed as dist
        from . import td3 as T
        assert agent.device.type == "cuda"
        # fused = the hand-derived update of td3_fused.py (library GEMMs + the HIP kernels of csrc/td3_kernels.hip); False = autograd (td3.td3_update)
        self.fused = None
        if fused:
            from .td3_fused import FusedTD3
            self.fused = FusedTD3(agent, seed=seed)
            self._collect_rng = FusedTD3.new_rng(agent.device, seed + 7919)
        self.env, self.agent, self.replay = env, agent, replay
        self.start_timesteps, self.expl_noise = start_timesteps, expl_noise
        self.batch_size, self.updates_per_step = batch_size, updates_per_step
        dev = agent.device
        n = env.num_envs
        self.n = n
        self.world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
        self.collectives = self.world > 1 or (dist.is_available() and dist.is_initialized() and os.environ.get("PLEN_TD3_FORCE_COLLECTIVES") == "1")
        self.allreduce_mode = None if not self.collectives else ("captured" if os.environ.get("PLEN_TD3_CAPTURE_ALLREDUCE") == "1" else "eager-between-graphs")
        torch.manual_seed(seed)
        # capturable Adam (step counters on device) that CONTINUES the agent's optimisers: same lr/betas/eps, moments and step counts
        # carried over (a resumed run must not restart Adam)
        agent.actor_optimizer = self._capturable_adam(agent.actor_optimizer, agent.actor)
        agent.critic_optimizer = self._capturable_adam(agent.critic_optimizer, agent.critic)
        if self.fused is not None:
            self.fused.enable_flat_adam()
        self.state = env.reset().to(torch.float32).clone()
        self.total_t = torch.zeros((), dtype=torch.long, device=dev)        # transitions written so far
        self.arange_n = torch.arange(n, device=dev)
        self.env_steps = 0
        self.grad_steps = 0
        self.host_total = 0
        self._graphs = {}
        self._eager_runs = {}
        from .vec_env import worker_stream
        self._side = worker_stream(dev, "side")
        self._critic_loss = torch.zeros((), device=dev)
        self._batch = None          # the sampled batch of the update in flight (static tensors once captured)

        def collect(random_actions):
            if random_actions and self.fused is not None:
                action = self.fused.uniform_actions(n, self._collect_rng)
            elif random_actions:
                action = torch.rand(n, 18, device=dev) * 2 - 1
            elif self.fused is not None:
                action = self.fused.explore(self.state, agent.max_action * expl_noise, rng=self._collect_rng)
            else:
                with torch.no_grad():
                    action = agent.actor(self.state)
                    action = (action + torch.randn_like(action) * (agent.max_action * expl_noise)).clamp(-agent.max_action, agent.max_action)
            next_obs, reward, done, info = env.step(action)
            if self.fused is not None and next_obs.dtype == torch.float32:
                self.fused.store(replay.data, self.total_t, self.state, action, next_obs, reward, done, rng=self._collect_rng)      # one kernel: packed rows into the ring
            else:
                idx = (self.total_t + self.arange_n) % replay.max_size
                terminal = ((done & 1) != 0) & ((done & 2) == 0)
                # one packed row per transition (ReplayBuffer.data): state | action | next_state | reward | not_done
                rows = torch.cat([self.state, action, next_obs.to(torch.float32), reward.to(torch.float32).reshape(n, 1),
                                  1.0 - terminal.to(torch.float32).reshape(n, 1)], 1)
                replay.data.index_copy_(0, idx, rows)
            self.total_t += n
            self.state.copy_(info["obs"])

        def sample_indices():
            size_t = torch.clamp(self.total_t, max=replay.max_size)
            ind = (torch.rand(batch_size, device=dev) * size_t).long().clamp_(max=replay.max_size - 1)
            return torch.minimum(ind, size_t - 1)

        def sample():
            ind = sample_indices()
            return (replay.state[ind], replay.action[ind], replay.next_state[ind], replay.reward[ind], replay.not_done[ind])

        fz = self.fused

        def update(with_policy, publish=True):         # one rank, or collectives captured: the whole iteration in one graph
            if fz is not None:
                loss = fz.update(replay.data, batch_size, with_policy, all_reduce=self.collectives, total=self.total_t)
            else:
                loss = T.td3_update(agent, sample(), with_policy, all_reduce=self.collectives)
            if publish:                                # (a 5 us device copy: a block of updates publishes its last loss only)
                self._critic_loss.copy_(loss)

        # --- segments for eager collectives between graph replays (world > 1) ---
        def seg_critic_backward():
            if fz is not None:
                self._critic_loss.copy_(fz.critic_backward(replay.data, batch_size, total=self.total_t))
            else:
                self._batch = sample()
                self._critic_loss.copy_(T.td3_critic_backward(agent, self._batch))

        def seg_critic_step(with_policy):
            agent.critic_optimizer.step()
            if with_policy:
                if fz is not None:
                    fz.policy_backward()
                else:
                    T.td3_actor_backward(agent, self._batch)

        def seg_actor_step():
            agent.actor_optimizer.step()
            if fz is not None:
                fz.polyak()
            else:
                T.td3_polyak(agent)

        self._collect_fn, self._update_fn = collect, update
        self._segs = (seg_critic_backward, seg_critic_step, seg_actor_step)

    @staticmethod
    def _capturable_adam(old, module):
        g = old.param_groups[0]
        new = torch.optim.Adam(module.parameters(), lr=g["lr"], betas=g["betas"], eps=g["eps"], weight_decay=g["weight_decay"],
                               capturable=True, fused=True)
        sd = old.state_dict()
        if sd["state"]:
            for st in sd["state"].values():           # capturable Adam keeps `step` as a float32 tensor on the parameter's device
                st["step"] = torch.as_tensor(float(st["step"]), dtype=torch.float32)
            for k in ("capturable", "fused"):
                for pg in sd["param_groups"]:
                    pg[k] = True
            new.load_state_dict(sd)
        return new

    def restore_counters(self, c):
        """Continue a run: the ring position comes from the (loaded) replay buffer, the update cadence from the counters."""
        r = self.replay
        self.host_total = r.size if r.size < r.max_size else r.max_size + r.ptr
        self.total_t.fill_(self.host_total)
        self.env_steps, self.grad_steps = int(c["env_steps"]), int(c["grad_steps"])

    def recapture(self):
        """Drop the captured graphs (they are re-captured on the next steps).  Needed after anything that changes a HOST decision frozen into them --
        e.g. the autograd path (TD3Agent.train) used on the same agent leaves the gradient buckets dirty, and a captured fused update would skip
        the zeroing (td3_fused.FusedTD3._zero_grads)."""
        torch.cuda.synchronize()
        self._graphs.clear(); self._eager_runs.clear()

    def _run(self, key, fn, *args):
        """Execute `fn(*args)` once: eagerly on the side stream the first two times this key is seen (allocator / lazy-init warm-up,
        and real work), from then on as a replay of its captured graph."""
        g = self._graphs.get(key)
        if g is not None:
            g.replay()
            return
        cur = torch.cuda.current_stream()
        runs = self._eager_runs.get(key, 0)
        if runs < 2:
            self._side.wait_stream(cur)
            with torch.cuda.stream(self._side):
                fn(*args)
            cur.wait_stream(self._side)
            self._eager_runs[key] = runs + 1
            return
        g = torch.cuda.CUDAGraph()
        # thread_local: another thread's HIP calls (the RCCL watchdog's event queries when world_size > 1) must not invalidate the capture
        with capture_graph(g, capture_error_mode="thread_local"):
            fn(*args)
        self._graphs[key] = g
        g.replay()                                   # the capture itself does not execute

    def _update_block(self, n):
        """n consecutive iterations starting at a multiple of policy_freq (td3.py:334: every policy_freq-th one carries the delayed policy update)."""
        for k in range(n):
            self._update_fn((k + 1) % self.agent.policy_freq == 0, publish=(k == n - 1))

    def _update(self, with_policy):
        if not self.collectives or self.allreduce_mode == "captured":
            self._run(("update", with_policy), self._update_fn, with_policy)
            return
        a, b, c = self._segs
        self._run(("critic_backward",), a)
        self.agent._critic_grads.all_reduce_mean()
        self._run(("critic_step", with_policy), b, with_policy)
        if with_policy:
            self.agent._actor_grads.all_reduce_mean()
            self._run(("actor_step",), c)

    def step(self):
        warm = self.host_total < self.start_timesteps
        self._run(("collect", warm), self._collect_fn, warm)
        self.host_total += self.n
        self.env_steps += self.n
        if self.host_total >= self.start_timesteps:
            n_up, pf = self.updates_per_step, self.agent.policy_freq
            if n_up > 1 and n_up % pf == 0 and self.grad_steps % pf == 0 and (not self.collectives or self.allreduce_mode == "captured"):
                # a whole vector step's updates as ONE graph (the reference's recipe: as many updates as env-steps, plen_td3.py:119-120): the pattern
                # of delayed policy updates repeats every policy_freq iterations, so the block is the same every step, and a chain of ~100 us
                # updates is not paced by one graph launch each
                self._run(("updates", n_up), self._update_block, n_up)
                self.grad_steps += n_up
                self.agent.total_it = self.grad_steps
            else:
                for _ in range(n_up):
                    with_policy = (self.grad_steps + 1) % pf == 0
                    self._update(with_policy)
                    self.grad_steps += 1
                    self.agent.total_it = self.grad_steps
        self.replay.size = min(self.host_total, self.replay.max_size)
        self.replay.ptr = self.host_total % self.replay.max_size if self.host_total >= self.replay.max_size else 0
        self.agent.last_critic_loss = self._critic_loss


class PipelinedVecTD3Trainer(object):
    """Actor / learner overlap on each rank's GPU: the envs step as two half batches on their own HIP streams and the fused TD3
    update runs on a third, all three replaying hipGraphs with only event dependencies between them.  With world_size > 1 the update is cut at
    its two gradient all-reduces into graph segments, the collectives issued on the update stream between the replays (like
    GraphedVecTD3Trainer); envs, replay and random streams are rank-local.

    Why: one 4096-env launch owns every wave slot of the chip for as long as its slowest waves run, and the update's ~50 small kernels
    cannot start beside it (measured, scripts/gpu_overlap_probe.py: 4096 envs 0.52 ms + update 0.46 ms = 0.99 ms together); a 2048-env
    launch leaves half the slots free: 0.41 ms + 0.46 ms run in 0.53 ms together.  So the loop is software-pipelined:
        collector h, step t : waits for update t-2;  acts with the behaviour actor written by update t-2 (a ring of 3 copies of the actor's
                              flat parameter buffer);  steps its 2048 envs;  stores the transitions into its rows of the ring
        update t            : waits for both collectors' step t-1;  samples rows written before step t (rows of steps t, t+1, which
                              collectors may be writing right now, are excluded once the ring has wrapped: k_sample_gather's guard);
                              at its end copies the actor into behaviour copy (t+1) % 3
    No tensor is written by one stream while another may read it (each dependency above is an event): the collection side is bitwise
    reproducible given its seeds (tested with learning off; the learner's float-atomic reductions are order-dependent in the last bits).
    The only semantic difference from VecTD3Trainer is that the acting policy is two updates old and the learner one vector step
    behind -- off-policy TD3 does not care.  One update of `batch_size` per vector step of `num_envs` env-steps, as in the synchronous loops."""

    def __init__(self, envs, agent, replay, start_timesteps=10000, expl_noise=0.1, batch_size=4096, seed=0):
        import copy
        import torch.distributed as dist
        from . import td3 as T
        from .td3_fused import FusedTD3
        assert len(envs) >= 2 and all(e.num_envs == envs[0].num_envs for e in envs) and agent.device.type == "cuda"
        self.world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
        self.collectives = self.world > 1 or (dist.is_available() and dist.is_initialized() and os.environ.get("PLEN_TD3_FORCE_COLLECTIVES") == "1")
        self.allreduce_mode = None if not self.collectives else "eager-between-graphs"
        self.envs, self.agent, self.replay = envs, agent, replay
        self.nh = envs[0].num_envs
        self.H = len(envs)                     # sub-batches ("halves" in the comments: 2 is the measured best, bench.py / DESIGN.md section 10)
        self.n = self.H * self.nh
        self.start_timesteps, self.expl_noise, self.batch_size = start_timesteps, expl_noise, batch_size
        dev = agent.device
        assert replay.max_size >= 4 * self.n, "the ring must hold more than the rows in flight"
        self.fused = FusedTD3(agent, seed=seed, rows=None if "PLEN_TD3_ROWS" in os.environ else True)      # beside resident env launches: single-wave workgroups
        self.rngs = [FusedTD3.new_rng(dev, seed + 7919 * (h + 1)) for h in range(self.H)]         # one random stream per collector
        torch.manual_seed(seed)
        agent.actor_optimizer = GraphedVecTD3Trainer._capturable_adam(agent.actor_optimizer, agent.actor)
        agent.critic_optimizer = GraphedVecTD3Trainer._capturable_adam(agent.critic_optimizer, agent.critic)
        self.fused.enable_flat_adam()
        # behaviour actors: 3 copies of the actor whose parameters are views of their own flat buffers
        self.behaviour = [copy.deepcopy(agent.actor) for _ in range(3)]
        self.bflat = [T._FlatParams(b) for b in self.behaviour]
        if self.fused._use_block(self.nh):         # (every later change of a behaviour copy is followed by its repacking: _finish)
            for b in self.behaviour:
                self.fused.pack_actor(b)
        from .vec_env import worker_stream
        self.streams = [worker_stream(dev, h) for h in range(self.H)]
        self.su = worker_stream(dev, "update")
        self.state = [e.reset().to(torch.float32).clone() for e in envs]
        self.base = [torch.tensor(h * self.nh, dtype=torch.long, device=dev) for h in range(self.H)]  # next ring row of each sub-batch
        self._store_ctr = [torch.zeros(1, dtype=torch.int32, device=dev) for _ in range(self.H)]     # plentd3_store_step's block counters
        self.total_u = torch.zeros((), dtype=torch.long, device=dev)                                  # rows complete before the current step
        self.ep_ret = [torch.zeros(self.nh, 2, device=dev) for _ in range(self.H)]                    # running return / length of every env
        self.ep_stats = torch.zeros(3, dtype=torch.float64, device=dev)                                                    # finished episodes: sum of returns, count, sum of lengths
        self._critic_loss = torch.zeros((), device=dev)
        self.t = 0
        self.learning = True                   # False: the collectors alone, acting with the (then frozen) policy -- what the loop costs without its learner
        self.env_steps = self.grad_steps = 0
        self._graphs, self._eager_runs = {}, {}
        self._ev_col = {}
        self._ev_upd = {}
        self.timeline = None
        torch.cuda.synchronize(dev)

    def enable_timeline(self, rows=64):
        """Development: device-clock stamps (td3_fused.stamp, 100 MHz ticks) at the ends of every collect and update, row = vector step % rows,
        columns = [collector h: start, env launch, env done, end] * H + [update: start, sampled, targets, critic forward, critic backward, end].
        Call before the graphs are captured."""
        assert not self._graphs, "enable_timeline() before the first captured step"
        self.timeline = torch.zeros(rows, 4 * self.H + 6, dtype=torch.long, device=self.agent.device)
        self.fused.probe = lambda k: self._stamp(self.total_u, 4 * self.H + k)
        return self.timeline

    def _stamp(self, counter, idx):
        if self.timeline is not None:
            self.fused.stamp(self.timeline, counter, self.n, idx)

    # one half-batch vector step: act, step the envs, store (all on the current stream)
    def _collect(self, h, random_actions, buf):
        env, nh = self.envs[h], self.nh
        dev = self.agent.device
        self._stamp(self.base[h], 4 * h)
        if random_actions:
            action = self.fused.uniform_actions(nh, self.rngs[h])
        else:
            action = self.fused.explore(self.state[h], self.agent.max_action * self.expl_noise, actor=self.behaviour[buf], rng=self.rngs[h], packed=True)
        self._stamp(self.base[h], 4 * h + 1)
        next_obs, reward, done, info = env.step(action)
        self._stamp(self.base[h], 4 * h + 2)
        obs = info["obs"]
        fold = obs.dtype == torch.float32 and obs.is_contiguous()          # (the state <- observation copy rides in the store kernel)
        self._stamp(self.base[h], 4 * h + 3)          # ("end" of the collect step: in front of the store kernel, which moves base on)
        # (the ring position of this collector's next step, base += rows of ALL collectors, rides in the store kernel: one launch less on the collector's chain)
        self.fused.store(self.replay.data, self.base[h], self.state[h], action, next_obs, reward, done, rng=self.rngs[h], episodes=(self.ep_ret[h], self.ep_stats),
                         advance=obs if fold else None, step=(self.n, self._store_ctr[h]))
        if not fold:
            self.state[h].copy_(obs)

    def _update(self, with_policy, buf_out):
        self._stamp(self.total_u, 4 * self.H)
        loss = self.fused.update(self.replay.data, self.batch_size, with_policy, all_reduce=False, total=self.total_u, guard=2 * self.n)
        self._critic_loss.copy_(loss)
        self._stamp(self.total_u, 4 * self.H + 5)
        self._finish(buf_out)

    def _finish(self, buf_out):
        self.total_u += self.n
        self.bflat[buf_out].flat.copy_(self.agent._actor_flat.flat)
        if self.fused._use_block(self.nh):     # ... and in matrix-core operand order for the collectors' actor forward, here on the update stream: off their critical path
            self.fused.pack_actor(self.behaviour[buf_out])

    # the same update cut at its collectives (world_size > 1)
    def _seg_critic_backward(self):
        self._critic_loss.copy_(self.fused.critic_backward(self.replay.data, self.batch_size, total=self.total_u, guard=2 * self.n))

    def _seg_critic_step(self, with_policy, buf_out):
        self.agent.critic_optimizer.step()
        if with_policy:
            self.fused.policy_backward()
        else:
            self._finish(buf_out)

    def _seg_actor_step(self, buf_out):
        self.agent.actor_optimizer.step()
        self.fused.polyak()
        self._finish(buf_out)

    def recapture(self):
        """Drop the captured graphs (they are re-captured on the next steps).  Needed after anything that changes a HOST decision frozen into them --
        e.g. the autograd path (TD3Agent.train) used on the same agent leaves the gradient buckets dirty, and a captured fused update would skip
        the zeroing (td3_fused.FusedTD3._zero_grads)."""
        torch.cuda.synchronize()
        self._graphs.clear(); self._eager_runs.clear()

    def _run(self, key, stream, fn, *args):
        """fn(*args) on `stream`: eagerly the first two times the key is seen, then as a replay of its graph captured on that stream."""
        with torch.cuda.stream(stream):
            g = self._graphs.get(key)
            if g is None:
                runs = self._eager_runs.get(key, 0)
                if runs < 2:
                    fn(*args)
                    self._eager_runs[key] = runs + 1
                    return
                g = torch.cuda.CUDAGraph()
                with capture_graph(g, stream=stream, capture_error_mode="thread_local"):
                    fn(*args)
                self._graphs[key] = g
            g.replay()

    def step(self):
        t, n = self.t, self.n
        warm = t * n < self.start_timesteps                    # uniform random actions until the ring holds start_timesteps transitions
        learn = self.learning and not warm and t >= 1
        for h in range(self.H):
            s = self.streams[h]
            ev = self._ev_upd.get(t - 2)
            if ev is not None:
                s.wait_event(ev)
            self._run(("collect", h, warm, (t - 1) % 3), s, self._collect, h, warm, (t - 1) % 3)
            e = torch.cuda.Event(); e.record(s); self._ev_col[(h, t)] = e
        su = self.su
        for h in range(self.H):
            ev = self._ev_col.get((h, t - 1))
            if ev is not None:
                su.wait_event(ev)
        if learn:
            with_policy = (self.grad_steps + 1) % self.agent.policy_freq == 0
            if not self.collectives:
                self._run(("update", with_policy, (t + 1) % 3), su, self._update, with_policy, (t + 1) % 3)
            else:
                self._run(("critic_backward",), su, self._seg_critic_backward)
                with torch.cuda.stream(su):
                    self.agent._critic_grads.all_reduce_mean()
                self._run(("critic_step", with_policy, (t + 1) % 3), su, self._seg_critic_step, with_policy, (t + 1) % 3)
                if with_policy:
                    with torch.cuda.stream(su):
                        self.agent._actor_grads.all_reduce_mean()
                    self._run(("actor_step", (t + 1) % 3), su, self._seg_actor_step, (t + 1) % 3)
            self.grad_steps += 1
            self.agent.total_it = self.grad_steps
        else:
            with torch.cuda.stream(su):
                self.total_u += n                               # invariant: total_u == t * n when update t samples (rows of steps < t are complete)
                self.bflat[(t + 1) % 3].flat.copy_(self.agent._actor_flat.flat)
                if self.fused._use_block(self.nh):
                    self.fused.pack_actor(self.behaviour[(t + 1) % 3])
        e = torch.cuda.Event(); e.record(su); self._ev_upd[t] = e
        self._step_tail(t, n)

    # ---- K vector steps of the whole pipeline as ONE hipGraph (round 6) -----------------------------------------------------------------------------------------
    BLOCK = 6          # lcm of the behaviour-copy ring (3) and policy_freq (2): the same graph serves every block of six steps

    def _block_body(self, t0, g0):
        """Six vector steps with the event structure of step(): collect(h, t) waits for update t - 2 and acts with copy (t - 1) % 3; update t waits for both collectors' step
        t - 1 and refreshes copy (t + 1) % 3.  Called under stream capture on the update stream: the collector streams fork from it and join it again, so every dependency
        becomes an edge of one graph (what lies before t0 is ordered by the graph launch itself)."""
        su, pf = self.su, self.agent.policy_freq
        ev0 = torch.cuda.Event(); ev0.record(su)
        for s in self.streams:
            s.wait_event(ev0)
        col, upd = {}, {}
        for k in range(self.BLOCK):
            t = t0 + k
            for h, s in enumerate(self.streams):
                if (t - 2) in upd:
                    s.wait_event(upd[t - 2])
                with torch.cuda.stream(s):
                    self._collect(h, False, (t - 1) % 3)
                e = torch.cuda.Event(); e.record(s); col[(h, t)] = e
            for h in range(self.H):
                if (h, t - 1) in col:
                    su.wait_event(col[(h, t - 1)])
            with torch.cuda.stream(su):
                self._update((g0 + k + 1) % pf == 0, (t + 1) % 3)
            e = torch.cuda.Event(); e.record(su); upd[t] = e
        for s in self.streams:
            su.wait_stream(s)

    def step_block(self):
        """BLOCK vector steps as one graph replay (single rank, steady state: past the random-action phase, learning on); falls back to BLOCK calls of step() otherwise.
        Why: step() launches three graphs per vector step, and every graph launch costs ~24 us of idle gap on the chain it
        belongs to -- on each collector's chain once per step.  Same kernels on the same data in the same order per stream as BLOCK calls of step()."""
        t, n = self.t, self.n
        warmed = (all(("collect", h, False, b) in self._graphs for h in range(self.H) for b in range(3))
                  and all(("update", wp, b) in self._graphs for wp in (False, True) for b in range(3)))          # every piece has run eagerly and been captured on its own: lazy initialisations are done
        if self.collectives or not self.learning or t * n < self.start_timesteps or t < 3 or not warmed:
            for _ in range(self.BLOCK):
                self.step()
            return
        su = self.su
        key = ("block", t % 3, self.grad_steps % self.agent.policy_freq)
        # everything enqueued by earlier step() calls on the collector streams is ordered before the block (its nodes run on the graph's own branches)
        for h in range(self.H):
            ev = self._ev_col.get((h, t - 1))
            if ev is not None:
                su.wait_event(ev)
        g = self._graphs.get(key)
        with torch.cuda.stream(su):
            if g is None:
                g = torch.cuda.CUDAGraph()
                with capture_graph(g, stream=su, capture_error_mode="thread_local"):
                    self._block_body(t, self.grad_steps)
                self._graphs[key] = g
            g.replay()
        # a later step() must see the whole block behind it: one event after the replay stands for every collect / update event of the block's last steps
        e = torch.cuda.Event(); e.record(su)
        for k in range(self.BLOCK):
            self._ev_upd[t + k] = e
            for h in range(self.H):
                self._ev_col[(h, t + k)] = e
        self.grad_steps += self.BLOCK
        self.agent.total_it = self.grad_steps
        for k in range(self.BLOCK):
            self._step_tail(t + k, n)

    def run(self, steps):
        """`steps` vector steps: whole blocks through step_block(), the remainder through step()."""
        while steps >= self.BLOCK:
            self.step_block(); steps -= self.BLOCK
        for _ in range(steps):
            self.step()

    def _step_tail(self, t, n):
        for k in [k for k in self._ev_upd if k < t - 3]:
            del self._ev_upd[k]
        for k in [k for k in self._ev_col if k[1] < t - 3]:
            del self._ev_col[k]
        self.t += 1
        self.env_steps += n
        done = self.t * n
        self.replay.size = min(done, self.replay.max_size)
        self.replay.ptr = done % self.replay.max_size if done >= self.replay.max_size else 0
        self.agent.last_critic_loss = self._critic_loss

    def sync(self):
        for s in self.streams + [self.su]:
            s.synchronize()

    def restore_counters(self, c):
        """Continue a run: ring position from the (loaded) replay buffer (rounded down to whole vector steps), update cadence from the counters."""
        r = self.replay
        done = r.size if r.size < r.max_size else r.max_size + r.ptr
        self.t = done // self.n
        self.total_u.fill_(self.t * self.n)
        for h in range(self.H):
            self.base[h].fill_(self.t * self.n + h * self.nh)
        self.env_steps, self.grad_steps = int(c["env_steps"]), int(c["grad_steps"])

    def episode_stats(self, reset=True):
        """Mean return and length of the episodes that finished since the last call (the reference prints them per episode, plen_env.py:616-636);
        accumulated on the device by the ring-store kernel.  Host-synchronous."""
        self.sync()
        s = self.ep_stats.tolist()
        if reset:
            # the collectors' next k_store adds into ep_stats on their own (non-blocking) streams: zero it on the update stream and make
            # both collectors wait for that, like every other cross-stream access of this class
            with torch.cuda.stream(self.su):
                self.ep_stats.zero_()
                ev = torch.cuda.Event(); ev.record(self.su)
            for st in self.streams:
                st.wait_event(ev)
        n = max(s[1], 1.0)
        return {"episodes": int(s[1]), "mean_return": s[0] / n, "mean_length": s[2] / n}


if __name__ == "__main__":
    main()
