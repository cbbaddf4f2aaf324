"""Adapters for the reference's actual consumers -- optional imports, nothing here is needed by the
hot path.

* ``validate_batched_environment``: what ``tf_agents.environments.utils.validate_py_environment`` checks
  (the reference runs it on its Environment, environment/environment_test.py:723-764), without
  tf-agents: random actions inside ``action_spec`` for a few episodes; every TimeStep must match the
  batched time-step spec (shapes, dtypes, finite observations, discount in [0, 1]) and the step types
  must follow FIRST -> MID* -> LAST -> FIRST.
* ``tf_agents_environment(env)``: a ``tf_agents.environments.py_environment.PyEnvironment`` subclass
  with ``batched = True`` around a ``BatchedEnvironment`` (environment/environment.py:1165-1370 is a
  PyEnvironment; this is the vectorised drop-in).  tf-agents drivers take NumPy nests, so this adapter
  copies the [B, O] observation and the [B] reward / discount / step type to the host once per step --
  use ``BatchedEnvironment`` directly to keep a policy on the device.
* ``gymnasium_vector_env(env)``: a ``gymnasium.vector.VectorEnv`` subclass (next-step autoreset).

Both factories import their library when called and raise ImportError if it is absent."""
from __future__ import annotations

from typing import Callable, Optional

import numpy as np
import torch

from sbsim_amd.environment import STEP_FIRST, STEP_LAST, STEP_MID


def validate_batched_environment(env, episodes: int = 2, action_fn: Optional[Callable] = None, seed: int = 0,
                                 max_steps: int = 100000) -> dict:
  """Drives ``env`` (reset() / step(action) -> TimeStep, action_spec(), observation_spec(), batch_size)
  with random in-spec actions for ``episodes`` episodes and asserts the TimeStep contract.  Returns
  counts (steps, episodes, rewards seen)."""
  a_spec, o_spec = env.action_spec(), env.observation_spec()
  B = int(env.batch_size)
  gen = torch.Generator(device="cpu")
  gen.manual_seed(seed)
  lo = -1.0 if a_spec.minimum is None else float(a_spec.minimum)
  hi = 1.0 if a_spec.maximum is None else float(a_spec.maximum)

  def check(ts, expect_first: bool):
    obs, rew, disc, st = ts.observation, ts.reward, ts.discount, ts.step_type
    assert tuple(obs.shape) == (B,) + tuple(o_spec.shape), (tuple(obs.shape), o_spec.shape)
    assert obs.dtype == torch.float32 and rew.dtype == torch.float32 and disc.dtype == torch.float32
    assert tuple(rew.shape) == (B,) and tuple(disc.shape) == (B,) and tuple(st.shape) == (B,)
    assert st.dtype == torch.int32
    assert bool(torch.isfinite(obs).all()), "non-finite observation"
    assert bool(((disc >= 0) & (disc <= 1)).all()), "discount outside [0, 1]"
    kinds = set(int(k) for k in torch.unique(st).tolist())
    assert len(kinds) == 1, "the buildings share the simulator clock: one step type per TimeStep"
    k = kinds.pop()
    assert k in (STEP_FIRST, STEP_MID, STEP_LAST)
    if expect_first:
      assert k == STEP_FIRST, "an episode starts with FIRST"
    if k == STEP_FIRST:
      assert bool((rew == 0).all()), "reward of a FIRST step"
    else:
      assert not bool(torch.isnan(rew).any())
    return k

  n_steps, n_eps = 0, 0
  k = check(env.reset(), True)
  dev = _device_of(env)
  while n_eps < episodes:
    assert n_steps < max_steps, "no episode end"
    if action_fn is not None:
      act = action_fn(n_steps)
    else:
      act = torch.rand((B,) + tuple(a_spec.shape), generator=gen, dtype=torch.float32) * (hi - lo) + lo
    assert bool(((act >= lo) & (act <= hi)).all())
    prev = k
    k = check(env.step(act.to(dev)), prev == STEP_LAST)   # after LAST the next step must be a FIRST (auto-reset)
    if k == STEP_LAST:
      n_eps += 1
    n_steps += 1
  return {"steps": n_steps, "episodes": n_eps}


def _device_of(env):
  sim = getattr(env, "sim", None)
  return getattr(sim, "tdev", torch.device("cpu"))


def tf_agents_environment(env):
  """-> tf_agents PyEnvironment (batched) around ``env``.  ImportError without tf-agents."""
  from tf_agents.environments import py_environment
  from tf_agents.specs import array_spec
  from tf_agents.trajectories import time_step as ts_lib

  a_spec, o_spec = env.action_spec(), env.observation_spec()

  class TFAgentsBatchedEnv(py_environment.PyEnvironment):
    """environment/environment.py's PyEnvironment interface, B buildings per call."""

    def __init__(self):
      super().__init__()
      self._env = env
      self._action_spec = array_spec.BoundedArraySpec(shape=tuple(a_spec.shape), dtype=np.float32, minimum=-1.0, maximum=1.0,
                                                      name="action")
      self._observation_spec = array_spec.ArraySpec(shape=tuple(o_spec.shape), dtype=np.float32, name="observation")

    @property
    def batched(self) -> bool:
      return True

    @property
    def batch_size(self) -> int:
      return int(self._env.batch_size)

    def action_spec(self):
      return self._action_spec

    def observation_spec(self):
      return self._observation_spec

    def _convert(self, t):
      return ts_lib.TimeStep(step_type=t.step_type.cpu().numpy().astype(np.int32), reward=t.reward.cpu().numpy().astype(np.float32),
                             discount=t.discount.cpu().numpy().astype(np.float32), observation=t.observation.cpu().numpy())

    def _reset(self):
      return self._convert(self._env.reset())

    def _step(self, action):
      a = torch.as_tensor(np.asarray(action, dtype=np.float32)).to(_device_of(self._env))
      return self._convert(self._env.step(a))

    def close(self):
      self._env.close()

  return TFAgentsBatchedEnv()


def gymnasium_vector_env(env, time_limit_is_truncation: bool = True):
  """-> gymnasium.vector.VectorEnv around ``env`` (next-step autoreset: the step after a terminal one ignores
  its actions and returns the next episode's first observation).  ImportError without gymnasium.

  An episode here ends only because its time is up (`num_timesteps_in_episode`, environment.py:1286-1292) --
  in gymnasium's vocabulary a TRUNCATION: `truncated` is set at the last step and `terminated` stays False, so
  that a learner keeps bootstrapping from the last observation.  (The reference's own TF-Agents environment
  ends the episode with discount 0, i.e. as a termination; `time_limit_is_truncation=False` reports it that
  way: `terminated` set, `truncated` False -- round 3's behaviour.)"""
  import gymnasium as gym

  a_spec, o_spec = env.action_spec(), env.observation_spec()

  class SbsimVectorEnv(gym.vector.VectorEnv):
    def __init__(self):
      self._env = env
      self.num_envs = int(env.batch_size)
      self.single_observation_space = gym.spaces.Box(-np.inf, np.inf, shape=tuple(o_spec.shape), dtype=np.float32)
      self.single_action_space = gym.spaces.Box(-1.0, 1.0, shape=tuple(a_spec.shape), dtype=np.float32)
      self.observation_space = gym.spaces.Box(-np.inf, np.inf, shape=(self.num_envs,) + tuple(o_spec.shape), dtype=np.float32)
      self.action_space = gym.spaces.Box(-1.0, 1.0, shape=(self.num_envs,) + tuple(a_spec.shape), dtype=np.float32)

    def reset(self, *, seed=None, options=None):
      del seed, options   # the simulator is deterministic; inputs come from the host controllers
      t = self._env.reset()
      return t.observation.cpu().numpy(), {"step_type": t.step_type.cpu().numpy()}

    def step(self, actions):
      a = torch.as_tensor(np.asarray(actions, dtype=np.float32)).to(_device_of(self._env))
      t = self._env.step(a)
      st = t.step_type.cpu().numpy()
      last = st == STEP_LAST
      terminated, truncated = (np.zeros_like(last), last) if time_limit_is_truncation else (last, np.zeros_like(last))
      return (t.observation.cpu().numpy(), t.reward.cpu().numpy().astype(np.float32), terminated, truncated,
              {"step_type": st, "discount": t.discount.cpu().numpy()})

    def close(self, **kwargs):
      self._env.close()

  return SbsimVectorEnv()
