"""What survives an episode reset and what does not, against two rollouts of the reference on one
simulator with SimulatorBuilding.reset() in between (tests/golden/h2_sb1_r9_episodes.npz from
oracle/gen_golden_episodes.py): Vav.reset() closes the reheat valve and sets the damper to 0.1
(vav.py:93-99) while the thermostat's mode and previous time stamp and the boiler's action time
stamp survive (thermostat.py:66-69, smart_device.py:71-72); episode 2 opens with rejected requests
(rejection_simulator_building.py:52-60), so its first steps run in HEAT mode with a closed valve, and
the boiler's tank lag sees a negative duration since its last action (boiler.py:158-217).

CPU: the oracle is bit-exact.  GPU (-m gpu): the HIP path through BatchedSimulator (C ABI)."""
import numpy as np
import pytest

from oracle import oracle as orc
from tests.golden_util import load, oracle_params, oracle_plan


def _ulp32(a, b):
  a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
  return np.abs(a.view(np.int32).astype(np.int64) - b.view(np.int32).astype(np.int64))


def test_oracle_restates_reset_between_episodes():
  g = load("h2_sb1_r9_episodes.npz")
  plan = oracle_plan(load("plan_r9_sb1.npz"))
  prm = oracle_params(g["params_json"])
  ob = orc.OracleBuilding(plan, prm, float(g["initial_temp"]))
  for ep in ("e1_", "e2_"):
    ob.reset()
    ob.observe_boiler(float(g[ep + "ts_seconds"][0]))   # Environment.reset()'s observation
    T = len(g[ep + "n_sweeps"])
    for t in range(T):
      out = ob.step(
          now_ts=float(g[ep + "ts_seconds"][t]), t_amb_now=float(g[ep + "t_amb_now"][t]), h_conv=float(g["h_conv"]),
          t_amb_next=float(g[ep + "t_amb_next"][t]), comfort_now=bool(g[ep + "comfort_now"][t]),
          comfort_prev=g[ep + "comfort_prev"][t] == 1, comfort_next=bool(g[ep + "comfort_next"][t]),
          occupancy=float(g[ep + "occupancy"][t]), e_price=float(g[ep + "e_price"][t]),
          e_carbon=float(g[ep + "e_carbon"][t]), g_price=float(g[ep + "g_price"][t]),
          g_carbon=float(g[ep + "g_carbon"][t]), action=g[ep + "action_native"][t], observe=True,
          reject=bool(g[ep + "rejected"][t]))
      assert out["n_sweeps"] == int(g[ep + "n_sweeps"][t]), (ep, t)
      assert np.array_equal(out["mode"], g[ep + "mode"][t]), (ep, t)
      assert np.array_equal(out["valve"], g[ep + "valve"][t]) and np.array_equal(out["damper"], g[ep + "damper"][t]), (ep, t)
      assert np.array_equal(out["zone_temp_post"], g[ep + "zone_temp_post"][t]), (ep, t)
      assert out["blr_tank_temp"] == g[ep + "blr_tank_temp"][t], (ep, t, out["blr_tank_temp"], g[ep + "blr_tank_temp"][t])
      rates = np.array([out["blower_rate"], out["ac_rate"], out["gas_rate"], out["pump_rate"]], np.float32)
      assert _ulp32(rates, g[ep + "rates"][t]).max() <= 1, (ep, t, rates, g[ep + "rates"][t])
      assert _ulp32(out["reward"], g[ep + "reward"][t]) <= 2, (ep, t)
    assert np.array_equal(ob.grid(), g[ep + "final_grid"])


@pytest.mark.gpu
@pytest.mark.parametrize("path", ["reg", "lds"])
def test_hip_reset_between_episodes_against_reference_golden(path, monkeypatch):
  torch = pytest.importorskip("torch")
  if not torch.cuda.is_available():
    pytest.skip("no GPU")
  from sbsim_amd import _ffi
  from sbsim_amd.environment import ACTION_REJECTION_REWARD, BatchedSimulator, SimConfig
  from tests.test_gpu_parity import T_TOL, _plan

  class _Episode:   # _step_in reads g[key][t]
    def __init__(self, g, ep):
      self.g, self.ep = g, ep

    def __getitem__(self, k):
      return self.g[self.ep + k]

  from tests.test_gpu_parity import _step_in
  if path == "lds":
    monkeypatch.setenv("SBSIM_FORCE_LDS_PATH", "1")
  g = load("h2_sb1_r9_episodes.npz")
  B = 5
  sim = BatchedSimulator(_plan(load("plan_r9_sb1.npz")), SimConfig.sb1(), B, float(g["h_conv"]))
  obs = torch.zeros((B, sim.O), dtype=torch.float32, device="cuda")
  rew = torch.zeros((B,), dtype=torch.float32, device="cuda")
  info = torch.zeros((B, _ffi.SB_INFO_STRIDE), dtype=torch.float32, device="cuda")
  aux = [0.0] * _ffi.SB_NUM_AUX
  for ep in ("e1_", "e2_"):
    e = _Episode(g, ep)
    sim.reset()
    sim.observe(aux, float(e["t_amb_now"][0]), obs)   # Environment.reset()'s observation (environment.py:1165-1176)
    assert np.allclose(obs.cpu().numpy()[0, :len(e["obs_reset"])], e["obs_reset"], rtol=1e-6, atol=1e-6), ep
    T = len(e["n_sweeps"])
    for t in range(T):
      act = torch.tensor(np.tile(e["actions_norm"][t], (B, 1)), dtype=torch.float32, device="cuda")
      rej = torch.full((B,), int(e["rejected"][t]), dtype=torch.uint8, device="cuda")
      si = _step_in(e, t)
      si.reject_dev = rej.data_ptr()
      sim.step(act, si, obs, rew, info)
      i = info.cpu().numpy().astype(np.float64)
      assert (i[:, 4] == e["n_sweeps"][t]).all(), (ep, t, i[:, 4], e["n_sweeps"][t])
      assert np.abs(sim.zone_temps().cpu().numpy() - e["zone_temp_post"][t]).max() < T_TOL, (ep, t)
      assert np.array_equal(sim.modes().cpu().numpy(), np.tile(e["mode"][t], (B, 1))), (ep, t)
      assert np.allclose(i[:, :4], e["rates"][t].astype(np.float64), rtol=2e-6, atol=1e-6), (ep, t, i[0, :4], e["rates"][t])
      sc = sim.scalars().cpu().numpy()
      assert np.allclose(sc[:, 8], e["blr_tank_temp"][t], atol=1e-9), (ep, t)
      assert np.allclose(sc[:, 5], e["blr_flow"][t], atol=1e-12), (ep, t)   # reheat flow = sum of the open valves
      r = rew.cpu().numpy().astype(np.float64)
      if e["rejected"][t]:
        assert (r == ACTION_REJECTION_REWARD).all(), (ep, t)
      else:
        assert np.abs(r - float(e["reward"][t])).max() < 1e-6, (ep, t)
    assert np.abs(sim.temps().cpu().numpy() - e["final_grid"]).max() < T_TOL, ep
  sim.close()
