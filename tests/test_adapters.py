"""sbsim_amd/adapters.py: the TimeStep-contract check (what tf-agents' validate_py_environment checks,
environment/environment_test.py:723-764) and the optional tf-agents / gymnasium adapters.

CPU: the contract check and the adapters on a small stand-in environment with BatchedEnvironment's
interface (the adapters only relay TimeSteps); the real libraries are used when installed, otherwise the
adapters run against minimal stand-ins of the few classes they touch.  GPU (-m gpu): the contract check
on the real BatchedEnvironment."""
import sys
import types

import numpy as np
import pytest

torch = pytest.importorskip("torch")

from sbsim_amd import adapters  # noqa: E402
from sbsim_amd.environment import STEP_FIRST, STEP_LAST, STEP_MID, ArraySpec, TimeStep  # noqa: E402


class _ToyEnv:
  """BatchedEnvironment's interface on CPU tensors: episodes of `n` transitions + the terminal step, then auto-reset."""

  def __init__(self, B=3, O=5, n=4, bad=None):
    self.batch_size, self.O, self.n, self.bad, self.k, self.closed = B, O, n, bad, 0, False
    self._ended = False

  def action_spec(self):
    return ArraySpec((2,), np.dtype(np.float32), "action", -1.0, 1.0)

  def observation_spec(self):
    return ArraySpec((self.O,), np.dtype(np.float32), "observation")

  def _ts(self, kind, reward):
    B = self.batch_size
    obs = torch.full((B, self.O), float(self.k), dtype=torch.float64 if self.bad == "dtype" else torch.float32)
    return TimeStep(torch.full((B,), kind, dtype=torch.int32), torch.full((B,), reward, dtype=torch.float32),
                    torch.ones((B,), dtype=torch.float32), obs)

  def reset(self):
    self.k, self._ended = 0, False
    return self._ts(STEP_FIRST, 0.0)

  def step(self, action):
    assert tuple(action.shape) == (self.batch_size, 2)
    if self._ended and self.bad != "no_autoreset":
      return self.reset()
    self.k += 1
    if self.k > self.n:
      self._ended = True
      return self._ts(STEP_LAST, -0.5)
    return self._ts(STEP_MID, -0.25)

  def close(self):
    self.closed = True


def test_contract_check_accepts_a_conforming_environment_and_rejects_broken_ones():
  out = adapters.validate_batched_environment(_ToyEnv(), episodes=3)
  assert out == {"steps": 3 * 5 + 2, "episodes": 3}      # 4 transitions + the terminal step per episode, + the auto-resets
  with pytest.raises(AssertionError):
    adapters.validate_batched_environment(_ToyEnv(bad="dtype"), episodes=1)
  with pytest.raises(AssertionError):
    adapters.validate_batched_environment(_ToyEnv(bad="no_autoreset"), episodes=2)


def _fake_tf_agents(monkeypatch):
  """The four classes the adapter touches, when tf-agents itself is not installed."""
  root = types.ModuleType("tf_agents")
  envs = types.ModuleType("tf_agents.environments")
  pe = types.ModuleType("tf_agents.environments.py_environment")
  specs = types.ModuleType("tf_agents.specs")
  asp = types.ModuleType("tf_agents.specs.array_spec")
  traj = types.ModuleType("tf_agents.trajectories")
  tsl = types.ModuleType("tf_agents.trajectories.time_step")

  class PyEnvironment:
    def __init__(self):
      self._current = None

    def reset(self):
      self._current = self._reset()
      return self._current

    def step(self, action):
      self._current = self._step(action)
      return self._current

  class ArraySpecT:
    def __init__(self, shape, dtype, name=None):
      self.shape, self.dtype, self.name = tuple(shape), np.dtype(dtype), name

  class BoundedArraySpec(ArraySpecT):
    def __init__(self, shape, dtype, minimum, maximum, name=None):
      super().__init__(shape, dtype, name)
      self.minimum, self.maximum = minimum, maximum

  import collections
  tsl.TimeStep = collections.namedtuple("TimeStep", ["step_type", "reward", "discount", "observation"])
  pe.PyEnvironment, asp.ArraySpec, asp.BoundedArraySpec = PyEnvironment, ArraySpecT, BoundedArraySpec
  for name, mod in (("tf_agents", root), ("tf_agents.environments", envs), ("tf_agents.environments.py_environment", pe),
                    ("tf_agents.specs", specs), ("tf_agents.specs.array_spec", asp), ("tf_agents.trajectories", traj),
                    ("tf_agents.trajectories.time_step", tsl)):
    monkeypatch.setitem(sys.modules, name, mod)
  envs.py_environment, specs.array_spec, traj.time_step = pe, asp, tsl


def test_tf_agents_adapter_relays_batched_time_steps(monkeypatch):
  try:
    import tf_agents  # noqa: F401
  except ImportError:
    _fake_tf_agents(monkeypatch)
  toy = _ToyEnv(B=4, O=6, n=2)
  env = adapters.tf_agents_environment(toy)
  assert env.batched and env.batch_size == 4
  assert env.action_spec().shape == (2,) and env.action_spec().minimum == -1.0 and env.observation_spec().shape == (6,)
  ts = env.reset()
  assert ts.observation.shape == (4, 6) and ts.observation.dtype == np.float32 and (ts.step_type == STEP_FIRST).all()
  kinds = []
  for _ in range(5):
    ts = env.step(np.zeros((4, 2), np.float32))
    assert ts.reward.dtype == np.float32 and ts.discount.shape == (4,)
    kinds.append(int(ts.step_type[0]))
  assert kinds == [STEP_MID, STEP_MID, STEP_LAST, STEP_FIRST, STEP_MID]
  env.close()
  assert toy.closed


def test_gymnasium_adapter_autoresets_at_the_next_step(monkeypatch):
  try:
    import gymnasium  # noqa: F401
  except ImportError:
    gym = types.ModuleType("gymnasium")
    gym.vector = types.ModuleType("gymnasium.vector")
    gym.spaces = types.ModuleType("gymnasium.spaces")

    class VectorEnv:
      pass

    class Box:
      def __init__(self, low, high, shape, dtype):
        self.low, self.high, self.shape, self.dtype = low, high, tuple(shape), dtype

    gym.vector.VectorEnv, gym.spaces.Box = VectorEnv, Box
    for name, mod in (("gymnasium", gym), ("gymnasium.vector", gym.vector), ("gymnasium.spaces", gym.spaces)):
      monkeypatch.setitem(sys.modules, name, mod)
  toy = _ToyEnv(B=2, O=3, n=1)
  venv = adapters.gymnasium_vector_env(toy)
  assert venv.num_envs == 2 and venv.single_action_space.shape == (2,) and venv.observation_space.shape == (2, 3)
  obs, info = venv.reset(seed=1)
  assert obs.shape == (2, 3) and (info["step_type"] == STEP_FIRST).all()
  obs, rew, term, trunc, info = venv.step(np.zeros((2, 2), np.float32))
  assert not term.any() and not trunc.any() and rew.dtype == np.float32
  obs, rew, term, trunc, info = venv.step(np.zeros((2, 2), np.float32))
  assert trunc.all() and not term.any()          # the episode's time is up: gymnasium's truncation
  obs, rew, term, trunc, info = venv.step(np.zeros((2, 2), np.float32))   # next-step autoreset
  assert not term.any() and not trunc.any() and (info["step_type"] == STEP_FIRST).all() and (rew == 0).all()
  # the reference's own reading (discount 0 = a termination), on request
  venv = adapters.gymnasium_vector_env(_ToyEnv(B=2, O=3, n=1), time_limit_is_truncation=False)
  venv.reset()
  venv.step(np.zeros((2, 2), np.float32))
  _, _, term, trunc, _ = venv.step(np.zeros((2, 2), np.float32))
  assert term.all() and not trunc.any()


@pytest.mark.gpu
def test_batched_environment_meets_the_time_step_contract():
  """The check tf-agents' validate_py_environment makes (environment_test.py:723-764), on the HIP environment:
  three episodes of 6 transitions, random in-spec actions."""
  if not torch.cuda.is_available():
    pytest.skip("no GPU")
  from sbsim_amd.environment import BatchedEnvironment
  from tests.golden_util import load
  from tests.test_gpu_parity import _plan
  env = BatchedEnvironment(_plan(load("plan_r9_sb1.npz")), 8, num_days_in_episode=6 * 300 / 86400.0, holiday_calendar=None)
  out = adapters.validate_batched_environment(env, episodes=3)
  assert out["episodes"] == 3 and out["steps"] == 3 * 7 + 2
  env.close()
