"""The configurable action set (boiler / air-handler heating + cooling setpoints / VAV damper
command) and the action-rejection path, against a rollout of the reference itself
(tests/golden/h2_sb1_r9_actions.npz from oracle/gen_golden_actions.py: SimulatorBuilding +
the reward function in Environment._step's order; at some steps request_action is never
reached -- RejectionSimulatorBuilding -- and at others the damper command leaves [0, 1]).

CPU: the oracle's restatement of those paths is bit-exact.  GPU (-m gpu): the HIP path through
the C ABI (sb_params.act_kind / sb_step_in.reject_dev) within the usual tolerances, and the
reward is -inf exactly where the reference's Environment returns ACTION_REJECTION_REWARD."""
import numpy as np
import pytest

from oracle import oracle as orc
from tests.golden_util import load, oracle_params, oracle_plan


def _ulp32(a, b):
  a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
  return np.abs(a.view(np.int32).astype(np.int64) - b.view(np.int32).astype(np.int64))


def _damper_cmd(g, t, Z):
  cmd = np.full(Z, np.nan)
  cmd[int(g["action_zone"][3])] = float(g["action_native_all"][t][3])
  return cmd


def test_oracle_restates_rejections_and_the_extra_actions():
  g = load("h2_sb1_r9_actions.npz")
  plan = oracle_plan(load("plan_r9_sb1.npz"))
  prm = oracle_params(g["params_json"])
  ob = orc.OracleBuilding(plan, prm, float(g["initial_temp"]))
  T, Z = len(g["n_sweeps"]), len(plan.zones)
  assert g["rejected"].sum() == 6 and (g["action_accepted"] == 0).sum() == 9
  for t in range(T):
    nat = g["action_native_all"][t]
    out = ob.step(
        now_ts=float(g["ts_seconds"][t]), t_amb_now=float(g["t_amb_now"][t]), h_conv=float(g["h_conv"]),
        t_amb_next=float(g["t_amb_next"][t]), comfort_now=bool(g["comfort_now"][t]),
        comfort_prev=g["comfort_prev"][t] == 1, comfort_next=bool(g["comfort_next"][t]),
        occupancy=float(g["occupancy"][t]), e_price=float(g["e_price"][t]), e_carbon=float(g["e_carbon"][t]),
        g_price=float(g["g_price"][t]), g_carbon=float(g["g_carbon"][t]), action=nat[:2], observe=True,
        reject=bool(g["rejected"][t]), cool_sp=float(nat[2]), damper_cmd=_damper_cmd(g, t, Z))
    assert out["action_accepted"] == bool(g["action_accepted"][t]), t
    assert out["n_sweeps"] == int(g["n_sweeps"][t]), t
    assert np.array_equal(out["zone_temp_post"], g["zone_temp_post"][t]), t
    assert np.array_equal(out["mode"], g["mode"][t]) and np.array_equal(out["damper"], g["damper"][t]), t
    assert out["blr_tank_temp"] == g["blr_tank_temp"][t] and out["t_supply_air"] == g["t_supply_air"][t], t
    rates = np.array([out["blower_rate"], out["ac_rate"], out["gas_rate"], out["pump_rate"]], np.float32)
    assert _ulp32(rates, g["rates"][t]).max() <= 1, (t, rates, g["rates"][t])
    assert _ulp32(out["reward"], g["reward"][t]) <= 2, t
  assert np.array_equal(ob.grid(), g["final_grid"])


@pytest.mark.gpu
@pytest.mark.parametrize("path", ["reg", "lds"])
def test_hip_action_set_and_rejections_against_reference_golden(path, monkeypatch):
  torch = pytest.importorskip("torch")
  if not torch.cuda.is_available():
    pytest.skip("no GPU")
  from sbsim_amd import _ffi
  from sbsim_amd.environment import ACTION_REJECTION_REWARD, BatchedSimulator, SimConfig, observation_field_names
  from tests.test_gpu_parity import T_TOL, _plan, _step_in
  if path == "lds":
    monkeypatch.setenv("SBSIM_FORCE_LDS_PATH", "1")
  g = load("h2_sb1_r9_actions.npz")
  cfg = SimConfig.sb1()
  cfg.action_names = tuple(str(n) for n in g["action_names"])
  cfg.action_zones = tuple(int(z) for z in g["action_zone"])
  cfg.action_ranges = tuple((float(lo), float(hi)) for lo, hi in g["action_ranges"])
  B = 6
  sim = BatchedSimulator(_plan(load("plan_r9_sb1.npz")), cfg, B, float(g["h_conv"]))
  assert sim.n_actions == 4 and sim.launch_info["path"] == (1 if path == "reg" else 0)
  sim.reset()
  obs = torch.zeros((B, sim.O), dtype=torch.float32, device="cuda")
  rew = torch.zeros((B,), dtype=torch.float32, device="cuda")
  info = torch.zeros((B, _ffi.SB_INFO_STRIDE), dtype=torch.float32, device="cuda")
  col_aux = observation_field_names([str(z) for z in g["zone_names"]], True)[4]
  T = len(g["n_sweeps"])
  # buildings 0..3 follow the golden (its rejections included); 4 and 5 are never rejected and
  # must differ from it after the first rejected step (the mask is per building)
  for t in range(T):
    act = torch.tensor(np.tile(g["actions_norm"][t], (B, 1)), dtype=torch.float32, device="cuda")
    rej = torch.zeros((B,), dtype=torch.uint8, device="cuda")
    rej[:4] = int(g["rejected"][t])
    si = _step_in(g, t)
    si.reject_dev = rej.data_ptr()
    sim.step(act, si, obs, rew, info)
    i = info.cpu().numpy().astype(np.float64)
    zt = sim.zone_temps().cpu().numpy()
    assert (i[:4, 4] == g["n_sweeps"][t]).all(), (t, i[:, 4], g["n_sweeps"][t])
    assert np.abs(zt[:4] - g["zone_temp_post"][t]).max() < T_TOL, t
    assert np.array_equal(sim.modes().cpu().numpy()[:4], np.tile(g["mode"][t], (4, 1))), t
    assert np.allclose(i[:4, :4], g["rates"][t].astype(np.float64), rtol=2e-6, atol=1e-6), t
    r = rew.cpu().numpy().astype(np.float64)
    if g["action_accepted"][t]:
      assert np.abs(r[:4] - float(g["reward"][t])).max() < 1e-6, t
    else:   # environment.py:1301-1302
      assert (r[:4] == ACTION_REJECTION_REWARD).all(), (t, r)
      assert np.abs(i[:4, 7] - float(g["reward"][t])).max() < 1e-6   # what compute_reward returned
    if g["rejected"][t]:
      assert np.isfinite(r[4:]).all() or not g["action_accepted"][t]
    o = obs.cpu().numpy()
    assert np.allclose(o[:4, :col_aux], g["obs"][t], rtol=1e-6, atol=1e-6), t
  grid = sim.temps().cpu().numpy()
  assert np.abs(grid[:4] - g["final_grid"]).max() < T_TOL
  assert np.abs(grid[4:] - g["final_grid"]).max() > 1e-3   # the never-rejected buildings went another way
  sc = sim.scalars().cpu().numpy()
  assert np.allclose(sc[:4, 8], g["blr_tank_temp"][-1], atol=1e-9)
  sim.close()
