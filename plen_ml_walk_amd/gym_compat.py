"""The sliver of the OpenAI-gym API the reference uses (gym.Env, spaces.Box, register / make with
max_episode_steps, TimeLimit), so the drop-in surfaces work whether or not `gym` is installed
(it is absent from the MI355X image).  If a real `gym` is importable the env is registered there too."""
import numpy as np


class Box(object):
    """gym.spaces.Box subset: low/high/shape/dtype, sample(), contains()."""

    def __init__(self, low, high, shape=None, dtype=np.float32):
        self.dtype = np.dtype(dtype)
        self.low = np.asarray(low, dtype=self.dtype) if shape is None else np.full(shape, low, dtype=self.dtype)
        self.high = np.asarray(high, dtype=self.dtype) if shape is None else np.full(shape, high, dtype=self.dtype)
        self.shape = self.low.shape
        self.np_random = np.random.RandomState()

    def seed(self, seed=None):
        self.np_random = np.random.RandomState(seed)
        return [seed]

    def sample(self):
        lo = np.where(np.isfinite(self.low), self.low, -1.0)
        hi = np.where(np.isfinite(self.high), self.high, 1.0)
        return self.np_random.uniform(lo, hi).astype(self.dtype)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))


class Env(object):
    metadata = {}
    reward_range = (-np.inf, np.inf)
    action_space = None
    observation_space = None

    def seed(self, seed=None):
        return [seed]

    def close(self):
        pass


class TimeLimit(object):
    """gym.wrappers.TimeLimit: done=True (info["TimeLimit.truncated"]) once `max_episode_steps` steps elapsed."""

    def __init__(self, env, max_episode_steps):
        self.env = env
        self._max_episode_steps = max_episode_steps
        self._elapsed_steps = None

    def __getattr__(self, name):
        return getattr(self.env, name)

    def step(self, action):
        obs, reward, done, info = self.env.step(action)
        self._elapsed_steps += 1
        if self._elapsed_steps >= self._max_episode_steps:
            info["TimeLimit.truncated"] = not done
            done = True
        return obs, reward, done, info

    def reset(self, **kw):
        self._elapsed_steps = 0
        return self.env.reset(**kw)


_REGISTRY = {}


def register(id, entry_point, max_episode_steps=None, **kwargs):     # noqa: A002 (gym's own argument name)
    _REGISTRY[id] = dict(entry_point=entry_point, max_episode_steps=max_episode_steps, kwargs=kwargs)


def make(id, **kwargs):     # noqa: A002
    if id not in _REGISTRY:
        raise KeyError("unknown environment id %r" % (id,))
    spec = _REGISTRY[id]
    ep = spec["entry_point"]
    if isinstance(ep, str):
        mod, cls = ep.split(":")
        ep = getattr(__import__(mod, fromlist=[cls]), cls)
    kw = dict(spec["kwargs"]); kw.update(kwargs)
    env = ep(**kw)
    if spec["max_episode_steps"]:
        env = TimeLimit(env, spec["max_episode_steps"])
    return env
