"""GPU parity: the HIP path (through the C ABI) against the reference-generated goldens
and against the CPU oracle on seeded inputs.

Tolerances (BASELINE.json north_star: "temperatures within 1e-4 C of reference"):
  * control-volume and zone temperatures: |dT| <= 1e-4 K is the contract; the kernel
    computes in float64 (reassociated sums, reciprocal multiply, fma) so the OBSERVED
    bound asserted here is 1e-8 K, and Gauss-Seidel sweep counts must be EQUAL;
  * fp32 proto fields (energy rates, reward): relative 2e-6 / absolute 1e-6;
  * cumulative energy over the day: relative 1e-6.
"""
import ctypes as C
import datetime as dt

import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from oracle import oracle as orc  # noqa: E402  (checker only)
from sbsim_amd import _ffi  # noqa: E402
from sbsim_amd.environment import (BatchedEnvironment, BatchedSimulator, SimConfig,  # noqa: E402
                                   observation_field_names)
from sbsim_amd.floorplan import FloorPlan, Material, Materials  # noqa: E402
from tests.golden_util import load, oracle_params, oracle_plan  # noqa: E402

T_TOL = 1e-8
TEST_SMALL = Materials(Material(50., 700., 1.), Material(2., 500., 1800.), Material(.05, 500., 3000.))


def _need_gpu():
  if not torch.cuda.is_available():
    pytest.skip("no GPU")


def _plan(p):
  return FloorPlan(conductivity=p["conductivity"], heat_capacity=p["heat_capacity"],
                   density=p["density"], exterior_space=p["exterior_space"],
                   zone_label=p["zone_label"], diffusers=p["diffusers"],
                   cv_size_cm=float(p["cv_size_cm"]), floor_height_cm=float(p["floor_height_cm"]),
                   zone_names=tuple(str(z) for z in p["zone_names"]))


def _step_in(g, t, has_action=True, occupancy=None):
  si = _ffi.StepIn()
  si.t_amb_now, si.t_amb_next = float(g["t_amb_now"][t]), float(g["t_amb_next"][t])
  si.comfort_now, si.comfort_prev, si.comfort_next = (int(g["comfort_now"][t]), int(g["comfort_prev"][t]),
                                                      int(g["comfort_next"][t]))
  si.has_action = int(has_action)
  si.occupancy = float(g["occupancy"][t]) if occupancy is None else occupancy
  si.e_price, si.e_carbon = float(g["e_price"][t]), float(g["e_carbon"][t])
  si.g_price, si.g_carbon = float(g["g_price"][t]), float(g["g_carbon"][t])
  return si


@pytest.mark.parametrize("tag,kernel", [("random", "reg"), ("const", "reg-pair"), ("const", "lds"),
                                        ("random", "lds-columns"), ("random", "reg-two"), ("random", "reg-band")])
def test_one_day_rollout_against_reference_golden(tag, kernel, monkeypatch):
  """BASELINE.json configs[0] semantics on the GPU: SB1 physics on R9, 288 steps."""
  _need_gpu()
  g = load(f"h2_sb1_r9_{tag}.npz")
  B = 5
  # "reg": the library's own choice for R9 -- rows as lanes, grid in registers, one wavefront
  # owns rows 0..63 and the last two (wall) rows are finished by a scan; "reg-pair": columns as
  # lanes, two wavefronts per building exchanging their seam rows through LDS; "lds": rows as
  # lanes on the LDS-grid kernel (two bands + seam); "lds-columns": the LDS-grid kernel on the
  # transposed grid; "reg-two": columns as lanes, one wavefront, two columns per lane, sweeps
  # overlapped in predicted blocks (step_two.hip); "reg-band": columns as lanes, two wavefronts (64 + 32
  # columns), sweeps overlapped in predicted blocks (step_band.hip).  All six wavefront schedules are checked
  # against the reference.
  if kernel.startswith("lds"):
    monkeypatch.setenv("SBSIM_FORCE_LDS_PATH", "1")
  if kernel == "reg-pair":
    monkeypatch.setenv("SBSIM_NO_TWO_ROW_PATH", "1")   # the library's own choice for 96 x 66 is "reg-two",
    monkeypatch.setenv("SBSIM_NO_BAND_PATH", "1")      # without it step_band.hip on two wavefronts ("reg-band")
  if kernel == "reg-band":
    monkeypatch.setenv("SBSIM_BAND_PATH", "1")
  sim = BatchedSimulator(_plan(load("plan_r9_sb1.npz")), SimConfig.sb1(), B, float(g["h_conv"]),
                         orientation={"reg": "auto", "reg-pair": "columns", "lds": "rows",
                                      "lds-columns": "columns", "reg-two": "columns", "reg-band": "columns"}[kernel])
  assert sim.transposed == (kernel in ("reg-pair", "lds-columns", "reg-two", "reg-band"))
  assert sim.launch_info["kernel"] == {"reg": 3, "reg-pair": 2, "lds": 0, "lds-columns": 0, "reg-two": 4, "reg-band": 5}[kernel]
  assert sim.launch_info["path"] == (1 if kernel.startswith("reg") else 0)
  assert sim.launch_info["waves_per_building"] == (2 if kernel in ("reg-pair", "reg-band") else 1)
  sim.reset()
  obs = torch.zeros((B, sim.O), dtype=torch.float32, device="cuda")
  rew = torch.zeros((B,), dtype=torch.float32, device="cuda")
  info = torch.zeros((B, _ffi.SB_INFO_STRIDE), dtype=torch.float32, device="cuda")
  T = len(g["n_sweeps"])
  col = observation_field_names([str(z) for z in g["zone_names"]], True)
  names, col_aux = col[0], col[4]
  assert [n for n in names[:col_aux]] == [str(n) for n in g["obs_names"]]
  worst_t, worst_r = 0.0, 0.0
  for t in range(T):
    act = torch.tensor(np.tile(g["actions_norm"][t], (B, 1)), dtype=torch.float32, device="cuda")
    sim.step(act, _step_in(g, t), obs, rew, info)
    i = info.cpu().numpy().astype(np.float64)
    zt = sim.zone_temps().cpu().numpy()
    assert (i[:, 4] == g["n_sweeps"][t]).all(), (t, i[:, 4], g["n_sweeps"][t])
    worst_t = max(worst_t, np.abs(zt - g["zone_temp_post"][t]).max())
    assert np.abs(zt - g["zone_temp_post"][t]).max() < T_TOL, t
    ref_rates = g["rates"][t].astype(np.float64)
    assert np.allclose(i[:, :4], ref_rates, rtol=2e-6, atol=1e-6), (t, i[0, :4], ref_rates)
    r = rew.cpu().numpy().astype(np.float64)
    worst_r = max(worst_r, np.abs(r - float(g["reward"][t])).max())
    assert np.abs(r - float(g["reward"][t])).max() < 1e-6, t
    # RewardResponse diagnostics (info[8:24] = proto fields 2..17) against the reference's response
    rr = np.array([g["rr_productivity"][t], g["rr_elec_cost"][t], g["rr_gas_cost"][t], g["rr_carbon"][t]], np.float64)
    assert np.allclose(i[:, 8:12], rr, rtol=3e-6, atol=1e-9), (t, i[0, 8:12], rr)
    nr = np.array([g["rr_norm_prod_regret"][t], g["rr_norm_energy_cost"][t], g["rr_norm_carbon"][t]], np.float64)
    assert np.allclose(i[:, 21:24], nr, rtol=3e-6, atol=1e-7), (t, i[0, 21:24], nr)
    assert (i[:, 12] == 0).all() and (i[:, 18] == 1).all() and (i[:, 19] == 0).all()
    assert np.allclose(i[:, 17], 9 * float(g["occupancy"][t]), rtol=1e-6) and np.allclose(i[:, 20], i[:, 8] - i[:, 16] * i[:, 17] * 300.0 / 3600.0, rtol=1e-4, atol=1e-4)
    # native observation values (identity normalisation) == the proto's fp32 values
    o = obs.cpu().numpy()
    assert np.allclose(o[:, :col_aux], g["obs"][t], rtol=1e-6, atol=1e-6), t
    if t + 1 in (1, 144):
      assert np.abs(sim.temps().cpu().numpy() - g[f"grid_{t + 1}"]).max() < T_TOL
  grid = sim.temps().cpu().numpy()
  assert np.abs(grid - g["final_grid"]).max() < T_TOL
  sc = sim.scalars().cpu().numpy()
  ref_cum = g["rates"].astype(np.float64).sum(axis=0) * 300.0
  assert np.allclose(sc[:, 12:16], ref_cum, rtol=1e-6)
  assert np.allclose(sc[:, 8], g["blr_tank_temp"][-1], atol=1e-9)
  print(f"[{tag}] max |dT_zone| = {worst_t:.3e} K, max |d reward| = {worst_r:.3e}")


@pytest.mark.parametrize("name,golden", [("small_test", "h1_small_test.npz"),
                                         ("weird_test", "h1_weird_test.npz"),
                                         ("r9_test", "h1_r9_test_cold200.npz")])
def test_thermostat_only_rollouts_on_reference_test_plans(name, golden):
  """Bare Simulator.step_sim() goldens: odd shapes (W odd, n<=1 cells), 100-sweep steps,
  and the reference KAT 301.895482 (simulator_flexible_floor_plan_test.py:1275-1312)."""
  _need_gpu()
  g = load(golden)
  p = load(f"plan_{name}.npz")
  import json
  prm = json.loads(str(g["params_json"]))
  cfg = SimConfig(
      time_step_sec=prm["dt"], convergence_threshold=prm["conv_threshold"], iteration_limit=prm["iter_limit"],
      comfort_temp_window=(prm["comfort_lo"], prm["comfort_hi"]), eco_temp_window=(prm["eco_lo"], prm["eco_hi"]),
      vav_max_air_flow_rate=prm["vav_max_air_flow"], vav_reheat_max_water_flow_rate=prm["vav_max_water_flow"],
      ahu_recirculation=prm["ahu_recirc"], ahu_heating_air_temp_setpoint=prm["ahu_heat_sp"],
      ahu_cooling_air_temp_setpoint=prm["ahu_cool_sp"], ahu_fan_differential_pressure=prm["ahu_dp"],
      ahu_fan_efficiency=prm["ahu_eff"], ahu_max_air_flow_rate=prm["ahu_max_flow"], ahu_has_weather_sensor=False,
      boiler_reheat_water_setpoint=prm["blr_setpoint"], boiler_water_pump_differential_head=prm["blr_head"],
      boiler_water_pump_efficiency=prm["blr_pump_eff"], boiler_heating_rate=prm["blr_heating_rate"],
      boiler_cooling_rate=prm["blr_cooling_rate"], initial_temp=float(g["initial_temp"]))
  B = 3
  sim = BatchedSimulator(_plan(p), cfg, B, float(g["h_conv"]))
  sim.reset()
  rew = torch.zeros((B,), dtype=torch.float32, device="cuda")
  info = torch.zeros((B, _ffi.SB_INFO_STRIDE), dtype=torch.float32, device="cuda")
  comfort = g["comfort"]
  for t in range(len(g["n_sweeps"])):
    si = _ffi.StepIn()
    si.t_amb_now = si.t_amb_next = float(g["t_amb"])
    si.comfort_now, si.comfort_next = int(comfort[t]), int(comfort[t + 1])
    si.comfort_prev = int(comfort[t - 1]) if t else -1
    si.has_action = 0
    si.occupancy, si.e_price, si.e_carbon, si.g_price, si.g_carbon = 1.0, 1e-8, 1e-8, 1e-8, 1e-8
    sim.step(None, si, None, rew, info)
    i = info.cpu().numpy()
    assert (i[:, 4] == g["n_sweeps"][t]).all(), (t, i[:, 4], g["n_sweeps"][t])
    assert np.abs(sim.zone_temps().cpu().numpy() - g["zone_temp_post"][t]).max() < T_TOL, t
    sc = sim.scalars().cpu().numpy()
    assert np.abs(sc[:, 7] - g["blr_return_temp"][t]).max() < 1e-9, t
    assert (sim.modes().cpu().numpy() == g["mode"][t]).all(), t
  assert np.abs(sim.temps().cpu().numpy() - g["final_grid"]).max() < T_TOL
  if name == "r9_test":
    assert abs(float(g["blr_return_temp"][0]) - 301.895482) < 1e-5


@pytest.mark.parametrize("mode", ["library", "exact-only", "redo-every-third", "redo-all"])
def test_divergent_buildings_against_oracle(mode, monkeypatch):
  """Config-2 style: per-building initial temperatures and per-building random actions;
  every building is checked against its own CPU-oracle twin.  k_sweep_roll decides a sweep on the high
  words of max |delta| and leaves the few buildings those 32 bits cannot decide to its float64
  instantiation (the redo list, step_roll.hip): the library's pair of launches, the float64 kernel alone,
  and the pair with every third / every building forced through the list."""
  _need_gpu()
  if mode == "exact-only":
    monkeypatch.setenv("SBSIM_ROLL_EXACT", "1")
  if mode.startswith("redo"):
    monkeypatch.setenv("SBSIM_DEBUG_FORCE_REDO", "3" if mode == "redo-every-third" else "1")
  g = load("h2_sb1_r9_random.npz")
  p = load("plan_r9_sb1.npz")
  B, T = 24, 40
  rs = np.random.RandomState(7)
  init = np.clip(294.0 + rs.randn(B, 1, 1) + 0.3 * rs.randn(B, 68, 98), 285.0, 305.0)
  acts = rs.uniform(-1, 1, size=(T, B, 2)).astype(np.float32)
  sim = BatchedSimulator(_plan(p), SimConfig.sb1(), B, float(g["h_conv"]))
  sim.reset(temps=torch.tensor(init.reshape(B, -1), dtype=torch.float64, device="cuda"))
  plan, prm = oracle_plan(p), oracle_params(g["params_json"])
  twins = [orc.OracleBuilding(plan, prm, 0.0, reset_temps=init[b].reshape(-1)) for b in range(B)]
  obs = torch.zeros((B, sim.O), dtype=torch.float32, device="cuda")
  rew = torch.zeros((B,), dtype=torch.float32, device="cuda")
  info = torch.zeros((B, _ffi.SB_INFO_STRIDE), dtype=torch.float32, device="cuda")
  rng_w, rng_a = (310.0, 355.0), (285.0, 300.0)
  for t in range(T):
    sim.step(torch.tensor(acts[t], device="cuda"), _step_in(g, t + 100), obs, rew, info)
    i = info.cpu().numpy().astype(np.float64)
    zt = sim.zone_temps().cpu().numpy()
    r = rew.cpu().numpy()
    for b in range(B):
      a = acts[t, b]
      native = [np.float32((float(a[0]) + 1.0) / 2.0 * (rng_w[1] - rng_w[0]) + rng_w[0]),
                np.float32((float(a[1]) + 1.0) / 2.0 * (rng_a[1] - rng_a[0]) + rng_a[0])]
      tt = t + 100
      o = twins[b].step(
          now_ts=300.0 * t, t_amb_now=float(g["t_amb_now"][tt]), h_conv=float(g["h_conv"]),
          t_amb_next=float(g["t_amb_next"][tt]), comfort_now=bool(g["comfort_now"][tt]),
          comfort_prev=g["comfort_prev"][tt] == 1, comfort_next=bool(g["comfort_next"][tt]),
          occupancy=float(g["occupancy"][tt]), e_price=float(g["e_price"][tt]),
          e_carbon=float(g["e_carbon"][tt]), g_price=float(g["g_price"][tt]),
          g_carbon=float(g["g_carbon"][tt]), action=native, observe=True)
      assert i[b, 4] == o["n_sweeps"], (t, b, i[b, 4], o["n_sweeps"])
      assert np.abs(zt[b] - o["zone_temp_post"]).max() < T_TOL, (t, b)
      ref = np.array([o["blower_rate"], o["ac_rate"], o["gas_rate"], o["pump_rate"]], np.float64)
      assert np.allclose(i[b, :4], ref, rtol=2e-6, atol=1e-6), (t, b)
      assert abs(float(r[b]) - o["reward"]) < 1e-6, (t, b)
  grid = sim.temps().cpu().numpy()
  for b in range(B):
    assert np.abs(grid[b] - twins[b].grid()).max() < T_TOL, b


@pytest.mark.parametrize("path", ["reg", "lds"])
def test_set_temps_with_a_grid_whose_sum_changes(path, monkeypatch):
  """sb_set_temps is a general `building.temp <- temps` (building.py:891-893), not only the seeded
  convection's within-room permutation: a grid with ANOTHER sum must move the next step's recirculation temperature
  (simulator.py:423-426 takes building.temp.mean() every step; the library caches it in scal[11]).  Two steps, a
  non-permutation edit of every building's interior, two more steps -- against oracle twins that get the same edit."""
  _need_gpu()
  if path == "lds":
    monkeypatch.setenv("SBSIM_FORCE_LDS_PATH", "1")
  g = load("h2_sb1_r9_random.npz")
  p = load("plan_r9_sb1.npz")
  B = 5
  rs = np.random.RandomState(11)
  init = np.clip(294.0 + rs.randn(B, 1, 1) + 0.3 * rs.randn(B, 68, 98), 285.0, 305.0)
  acts = rs.uniform(-1, 1, size=(4, B, 2)).astype(np.float32)
  sim = BatchedSimulator(_plan(p), SimConfig.sb1(), B, float(g["h_conv"]))
  sim.reset(temps=torch.tensor(init.reshape(B, -1), dtype=torch.float64, device="cuda"))
  plan, prm = oracle_plan(p), oracle_params(g["params_json"])
  twins = [orc.OracleBuilding(plan, prm, 0.0, reset_temps=init[b].reshape(-1)) for b in range(B)]
  obs = torch.zeros((B, sim.O), dtype=torch.float32, device="cuda")
  rew = torch.zeros((B,), dtype=torch.float32, device="cuda")
  info = torch.zeros((B, _ffi.SB_INFO_STRIDE), dtype=torch.float32, device="cuda")
  rng_w, rng_a = (310.0, 355.0), (285.0, 300.0)

  def both(t):
    sim.step(torch.tensor(acts[t], device="cuda"), _step_in(g, t + 100), obs, rew, info)
    i = info.cpu().numpy().astype(np.float64)
    zt = sim.zone_temps().cpu().numpy()
    for b in range(B):
      a, tt = acts[t, b], t + 100
      native = [np.float32((float(a[0]) + 1.0) / 2.0 * (rng_w[1] - rng_w[0]) + rng_w[0]),
                np.float32((float(a[1]) + 1.0) / 2.0 * (rng_a[1] - rng_a[0]) + rng_a[0])]
      o = twins[b].step(
          now_ts=300.0 * t, t_amb_now=float(g["t_amb_now"][tt]), h_conv=float(g["h_conv"]),
          t_amb_next=float(g["t_amb_next"][tt]), comfort_now=bool(g["comfort_now"][tt]),
          comfort_prev=g["comfort_prev"][tt] == 1, comfort_next=bool(g["comfort_next"][tt]),
          occupancy=float(g["occupancy"][tt]), e_price=float(g["e_price"][tt]), e_carbon=float(g["e_carbon"][tt]),
          g_price=float(g["g_price"][tt]), g_carbon=float(g["g_carbon"][tt]), action=native, observe=True)
      assert i[b, 4] == o["n_sweeps"], (t, b, i[b, 4], o["n_sweeps"])
      assert np.abs(zt[b] - o["zone_temp_post"]).max() < T_TOL, (t, b)
      assert abs(i[b, 6] - o["t_supply_air"]) < 1e-4, (t, b, i[b, 6], o["t_supply_air"])   # (an fp32 output)
      ref = np.array([o["blower_rate"], o["ac_rate"], o["gas_rate"], o["pump_rate"]], np.float64)
      assert np.allclose(i[b, :4], ref, rtol=2e-6, atol=1e-6), (t, b)

  both(0)
  both(1)
  grid = sim.temps().cpu().numpy().reshape(B, 68, 98)
  inside = ~np.asarray(p["exterior_space"], dtype=bool)
  edit = grid.copy()
  edit[:, inside] += 1.5 + 0.5 * rs.rand(B, int(inside.sum()))      # every interior cell warmer: the grid's sum changes
  edit[:, 10:30, 10:40] -= 2.0 * inside[10:30, 10:40]                # ... and one patch colder
  sim.set_temps(torch.tensor(edit, dtype=torch.float64, device="cuda"))
  for b in range(B):
    twins[b].temp[:] = edit[b].reshape(-1)      # the oracle reads its grid (and every mean of it) afresh each step
  assert abs(float(sim.scalars().cpu().numpy()[0, 11]) - edit[0].mean()) < 1e-9   # the recirculation temperature follows
  both(2)
  both(3)
  final = sim.temps().cpu().numpy()
  for b in range(B):
    assert np.abs(final[b].reshape(-1) - twins[b].temp).max() < T_TOL, b
  sim.close()


@pytest.mark.parametrize("limit", [1, 2, 5])
def test_iteration_limit_ends_the_step(limit):
  """simulator.py:348-368 with a limit that bites (the golden rollouts always converge): the
  register kernel starts sweep k+1 before it knows that sweep k was the last one allowed and has
  to undo it; sweep counts and grids against the oracle."""
  _need_gpu()
  import dataclasses
  g = load("h2_sb1_r9_random.npz")
  p = load("plan_r9_sb1.npz")
  B, T = 6, 8
  rs = np.random.RandomState(3)
  init = np.clip(294.0 + rs.randn(B, 1, 1) + 0.3 * rs.randn(B, 68, 98), 285.0, 305.0)
  acts = rs.uniform(-1, 1, size=(T, B, 2)).astype(np.float32)
  cfg = dataclasses.replace(SimConfig.sb1(), iteration_limit=limit)
  sim = BatchedSimulator(_plan(p), cfg, B, float(g["h_conv"]))
  assert sim.launch_info["path"] == 1
  sim.reset(temps=torch.tensor(init.reshape(B, -1), dtype=torch.float64, device="cuda"))
  plan, prm = oracle_plan(p), oracle_params(g["params_json"], iter_limit=limit)
  twins = [orc.OracleBuilding(plan, prm, 0.0, reset_temps=init[b].reshape(-1)) for b in range(B)]
  obs = torch.zeros((B, sim.O), dtype=torch.float32, device="cuda")
  rew = torch.zeros((B,), dtype=torch.float32, device="cuda")
  info = torch.zeros((B, _ffi.SB_INFO_STRIDE), dtype=torch.float32, device="cuda")
  rng_w, rng_a = (310.0, 355.0), (285.0, 300.0)
  hit = 0
  for t in range(T):
    sim.step(torch.tensor(acts[t], device="cuda"), _step_in(g, t + 100), obs, rew, info)
    i = info.cpu().numpy().astype(np.float64)
    for b in range(B):
      a = acts[t, b]
      native = [np.float32((float(a[0]) + 1.0) / 2.0 * (rng_w[1] - rng_w[0]) + rng_w[0]),
                np.float32((float(a[1]) + 1.0) / 2.0 * (rng_a[1] - rng_a[0]) + rng_a[0])]
      tt = t + 100
      o = twins[b].step(
          now_ts=300.0 * t, t_amb_now=float(g["t_amb_now"][tt]), h_conv=float(g["h_conv"]),
          t_amb_next=float(g["t_amb_next"][tt]), comfort_now=bool(g["comfort_now"][tt]),
          comfort_prev=g["comfort_prev"][tt] == 1, comfort_next=bool(g["comfort_next"][tt]),
          occupancy=float(g["occupancy"][tt]), e_price=float(g["e_price"][tt]),
          e_carbon=float(g["e_carbon"][tt]), g_price=float(g["g_price"][tt]),
          g_carbon=float(g["g_carbon"][tt]), action=native, observe=True)
      assert i[b, 4] == o["n_sweeps"] <= limit, (t, b, i[b, 4], o["n_sweeps"])
      hit += int(o["n_sweeps"] == limit)
  assert hit > 0
  grid = sim.temps().cpu().numpy()
  for b in range(B):
    assert np.abs(grid[b] - twins[b].grid()).max() < T_TOL, b


def test_per_building_weather_and_per_zone_occupancy():
  """The optional DEVICE inputs of sb_step_in: `t_amb_dev` [B][2] (every building its own
  ambient temperature now / next) and `occupancy_dev` [Z] (per-zone occupancy) -- BASELINE.json
  configs[2] semantics (per-instance weather).  Each building against its own oracle twin."""
  _need_gpu()
  g = load("h2_sb1_r9_random.npz")
  p = load("plan_r9_sb1.npz")
  B, T = 8, 12
  rs = np.random.RandomState(3)
  init = np.clip(294.0 + rs.randn(B, 1, 1) + 0.2 * rs.randn(B, 68, 98), 285.0, 305.0)
  acts = rs.uniform(-1, 1, size=(T, B, 2)).astype(np.float32)
  sim = BatchedSimulator(_plan(p), SimConfig.sb1(), B, float(g["h_conv"]))
  sim.reset(temps=torch.tensor(init.reshape(B, -1), dtype=torch.float64, device="cuda"))
  plan, prm = oracle_plan(p), oracle_params(g["params_json"])
  twins = [orc.OracleBuilding(plan, prm, 0.0, reset_temps=init[b].reshape(-1)) for b in range(B)]
  obs = torch.zeros((B, sim.O), dtype=torch.float32, device="cuda")
  rew = torch.zeros((B,), dtype=torch.float32, device="cuda")
  info = torch.zeros((B, _ffi.SB_INFO_STRIDE), dtype=torch.float32, device="cuda")
  rng_w, rng_a = (310.0, 355.0), (285.0, 300.0)
  offs = np.linspace(-12.0, 9.0, B)                       # building b lives in a climate offs[b] K away
  occ = np.array([0.0, 1.0, 2.5, 4.0, 0.5, 3.0, 1.5, 0.0, 2.0])[:sim.Z]
  occ_dev = torch.tensor(occ, dtype=torch.float64, device="cuda")
  for t in range(T):
    tt = t + 100
    amb = np.stack([float(g["t_amb_now"][tt]) + offs + 0.05 * t, float(g["t_amb_next"][tt]) + offs + 0.05 * (t + 1)], axis=1)
    amb_dev = torch.tensor(amb, dtype=torch.float64, device="cuda")
    si = _step_in(g, tt)
    si.t_amb_dev = amb_dev.data_ptr()
    si.occupancy_dev = occ_dev.data_ptr()
    sim.step(torch.tensor(acts[t], device="cuda"), si, obs, rew, info)
    i = info.cpu().numpy().astype(np.float64)
    zt = sim.zone_temps().cpu().numpy()
    r = rew.cpu().numpy()
    for b in range(B):
      a = acts[t, b]
      native = [np.float32((float(a[0]) + 1.0) / 2.0 * (rng_w[1] - rng_w[0]) + rng_w[0]),
                np.float32((float(a[1]) + 1.0) / 2.0 * (rng_a[1] - rng_a[0]) + rng_a[0])]
      o = twins[b].step(
          now_ts=300.0 * t, t_amb_now=float(amb[b, 0]), h_conv=float(g["h_conv"]),
          t_amb_next=float(amb[b, 1]), comfort_now=bool(g["comfort_now"][tt]),
          comfort_prev=g["comfort_prev"][tt] == 1, comfort_next=bool(g["comfort_next"][tt]),
          occupancy=occ, e_price=float(g["e_price"][tt]), e_carbon=float(g["e_carbon"][tt]),
          g_price=float(g["g_price"][tt]), g_carbon=float(g["g_carbon"][tt]), action=native, observe=True)
      assert i[b, 4] == o["n_sweeps"], (t, b, i[b, 4], o["n_sweeps"])
      assert np.abs(zt[b] - o["zone_temp_post"]).max() < T_TOL, (t, b)
      ref = np.array([o["blower_rate"], o["ac_rate"], o["gas_rate"], o["pump_rate"]], np.float64)
      assert np.allclose(i[b, :4], ref, rtol=2e-6, atol=1e-6), (t, b)
      assert abs(float(r[b]) - o["reward"]) < 1e-6, (t, b)
  grid = sim.temps().cpu().numpy()
  for b in range(B):
    assert np.abs(grid[b] - twins[b].grid()).max() < T_TOL, b
  # the exterior ring holds each building's own ambient temperature
  assert np.allclose(grid[:, 0, 0], amb[:, 0], atol=1e-12)
  assert len(set(np.round(grid[:, 0, 0], 6))) == B


def test_device_side_per_building_weather_through_the_environment_api():
  """SURVEY 8(f) rank 2 (weather half): BatchedSinusoidWeather -- every building its own
  (low, high), temperatures generated on the device from one host factor -- through
  BatchedEnvironment.reset()/step(); every building against an oracle twin driven by its own
  host WeatherController."""
  _need_gpu()
  import datetime as dt
  from sbsim_amd.environment import BatchedEnvironment
  from sbsim_amd.host_inputs import BatchedSinusoidWeather, WeatherController
  p = load("plan_r9_sb1.npz")
  g = load("h2_sb1_r9_random.npz")
  B, T = 6, 10
  rs = np.random.RandomState(9)
  low = np.array([262.0, 268.5, 273.0, 277.25, 281.0, 255.0])
  high = low + np.array([4.0, 9.5, 10.0, 12.0, 6.5, 20.0])
  weather = BatchedSinusoidWeather(low, high, convection_coefficient=float(g["h_conv"]))
  start = dt.datetime(2023, 7, 6, 9, 30, 0)
  env = BatchedEnvironment(_plan(p), B, config=SimConfig.sb1(), weather=weather, start_timestamp=start,
                           holiday_calendar=None, collect_info=True)
  ts0 = env.reset()
  names = env.field_names
  oat = names.index("air_handler_id/outside_air_temperature_sensor")
  assert np.allclose(ts0.observation.cpu().numpy()[:, oat], weather.temps(start).astype(np.float32))
  plan, prm = oracle_plan(p), oracle_params(g["params_json"])
  twins = [orc.OracleBuilding(plan, prm, float(env.config.initial_temp)) for _ in range(B)]
  ctrl = [WeatherController(lo, hi) for lo, hi in zip(low, high)]
  acts = rs.uniform(-1, 1, size=(T, B, 2)).astype(np.float32)
  rng_w, rng_a = env.config.action_ranges
  for t in range(T):
    now = env.current_simulation_timestamp
    si = env.make_step_in(now)
    out = env.step(torch.tensor(acts[t], device="cuda"))
    zt = env.sim.zone_temps().cpu().numpy()
    info = env.info.cpu().numpy()
    nxt = now + dt.timedelta(seconds=300)
    for b in range(B):
      a = acts[t, b]
      native = [np.float32((float(a[0]) + 1.0) / 2.0 * (rng_w[1] - rng_w[0]) + rng_w[0]),
                np.float32((float(a[1]) + 1.0) / 2.0 * (rng_a[1] - rng_a[0]) + rng_a[0])]
      o = twins[b].step(
          now_ts=300.0 * t, t_amb_now=ctrl[b].get_current_temp(now), h_conv=float(g["h_conv"]),
          t_amb_next=ctrl[b].get_current_temp(nxt), comfort_now=bool(si.comfort_now),
          comfort_prev=si.comfort_prev == 1, comfort_next=bool(si.comfort_next), occupancy=si.occupancy,
          e_price=si.e_price, e_carbon=si.e_carbon, g_price=si.g_price, g_carbon=si.g_carbon,
          action=native, observe=True)
      assert int(info[b, 4]) == o["n_sweeps"], (t, b, info[b, 4], o["n_sweeps"])
      assert np.abs(zt[b] - o["zone_temp_post"]).max() < T_TOL, (t, b)
      assert abs(float(out.reward[b]) - o["reward"]) < 1e-6, (t, b)
    assert np.allclose(out.observation.cpu().numpy()[:, oat], weather.temps(nxt).astype(np.float32))
  grid = env.sim.temps().cpu().numpy()
  for b in range(B):
    assert np.abs(grid[b] - twins[b].grid()).max() < T_TOL, b
  assert np.array_equal(grid[:, 0, 0], weather.temps(now))   # the exterior ring: the device's own T_ambient, bit for bit
  env.close()


def test_full_size_batch_register_and_lds_kernels_agree(monkeypatch):
  """BASELINE.json configs[1] at its full size (65,536 R9 buildings, random per-building
  initial temperatures and actions): the two independent sweep kernels -- grid in registers
  with the tail scan, grid in LDS with the banded DPP sweep -- must take the same number of
  Gauss-Seidel sweeps for EVERY building at every step and end within 1e-9 K of each other;
  sb_reset(temps) / sb_get_temps round-trips the caller's layout exactly."""
  _need_gpu()
  B, T = 65536, 10
  p = load("plan_r9_sb1.npz")
  g = load("h2_sb1_r9_random.npz")
  gen = torch.Generator(device="cuda")
  gen.manual_seed(5)
  t0 = (294.0 + torch.randn((B, 1), generator=gen, device="cuda", dtype=torch.float64)).clamp(285.0, 305.0)
  init = (t0 + 0.05 * torch.randn((B, 68 * 98), generator=gen, device="cuda", dtype=torch.float64)).contiguous()
  acts = torch.rand((T, B, 2), generator=gen, device="cuda", dtype=torch.float32) * 2.0 - 1.0
  sims = []
  for force_lds in (False, True):
    if force_lds:
      monkeypatch.setenv("SBSIM_FORCE_LDS_PATH", "1")
    sim = BatchedSimulator(_plan(p), SimConfig.sb1(), B, float(g["h_conv"]))
    assert sim.launch_info["path"] == (0 if force_lds else 1)
    sim.reset(temps=init)
    assert torch.equal(sim.temps().reshape(B, -1), init)
    sims.append(sim)
  obs = [torch.zeros((B, s.O), dtype=torch.float32, device="cuda") for s in sims]
  rew = [torch.zeros((B,), dtype=torch.float32, device="cuda") for _ in sims]
  info = [torch.zeros((B, _ffi.SB_INFO_STRIDE), dtype=torch.float32, device="cuda") for _ in sims]
  total = 0.0
  for t in range(T):
    si = _step_in(g, 100 + t)
    for k, sim in enumerate(sims):
      sim.step(acts[t], si, obs[k], rew[k], info[k])
    assert torch.equal(info[0][:, 4], info[1][:, 4]), t          # sweeps, every building
    assert bool((info[0][:, 5] == 1).all())                       # all converged
    total += float(info[0][:, 4].sum())
    assert float((sims[0].zone_temps() - sims[1].zone_temps()).abs().max()) < 1e-9, t
    assert float((rew[0] - rew[1]).abs().max()) < 1e-6, t
  assert float((sims[0].temps() - sims[1].temps()).abs().max()) < 1e-9
  assert 2.0 < total / (B * T) < 30.0
  for sim in sims:
    sim.close()


@pytest.mark.gpu
@pytest.mark.parametrize("B", [1, 3, 5, 1030])
def test_roll_kernel_workgroups_with_idle_and_drawing_wavefronts(B, monkeypatch):
  """k_sweep_roll runs four buildings = four wavefronts per workgroup that share the steps' class words
  in LDS: batches that leave wavefronts of the last workgroup without a building (1, 3, 5) and one just
  above the 1,024 resident wavefronts (the first draws from the device counter) against the LDS-grid
  kernel -- equal sweep counts for every building, grids within 1e-9 K."""
  _need_gpu()
  T = 8
  p = load("plan_r9_sb1.npz")
  g = load("h2_sb1_r9_random.npz")
  gen = torch.Generator(device="cuda")
  gen.manual_seed(11 + B)
  t0 = (294.0 + torch.randn((B, 1), generator=gen, device="cuda", dtype=torch.float64)).clamp(285.0, 305.0)
  init = (t0 + 0.05 * torch.randn((B, 68 * 98), generator=gen, device="cuda", dtype=torch.float64)).contiguous()
  acts = torch.rand((T, B, 2), generator=gen, device="cuda", dtype=torch.float32) * 2.0 - 1.0
  sims = []
  for force_lds in (False, True):
    if force_lds:
      monkeypatch.setenv("SBSIM_FORCE_LDS_PATH", "1")
    sim = BatchedSimulator(_plan(p), SimConfig.sb1(), B, float(g["h_conv"]))
    assert sim.launch_info["path"] == (0 if force_lds else 1)
    if not force_lds:
      assert sim.launch_info["kernel"] == 3 and sim.launch_info["waves_per_workgroup"] == 4
      assert sim.launch_info["workgroups"] == min((B + 3) // 4, 256)
    sim.reset(temps=init)
    sims.append(sim)
  obs = [torch.zeros((B, s.O), dtype=torch.float32, device="cuda") for s in sims]
  rew = [torch.zeros((B,), dtype=torch.float32, device="cuda") for _ in sims]
  info = [torch.zeros((B, _ffi.SB_INFO_STRIDE), dtype=torch.float32, device="cuda") for _ in sims]
  for t in range(T):
    si = _step_in(g, 100 + t)
    for k, sim in enumerate(sims):
      sim.step(acts[t], si, obs[k], rew[k], info[k])
    assert torch.equal(info[0][:, 4], info[1][:, 4]), t
    assert float((sims[0].zone_temps() - sims[1].zone_temps()).abs().max()) < 1e-9, t
  assert float((sims[0].temps() - sims[1].temps()).abs().max()) < 1e-9
  for sim in sims:
    sim.close()


@pytest.mark.gpu
@pytest.mark.parametrize("band", [False, True])
@pytest.mark.parametrize("rooms,room_shape,B", [((8, 5), (12, 14), 21845), ((14, 9), (8, 7), 21845)])
def test_mixed_classes_at_size_block_and_lds_kernels_agree(rooms, room_shape, B, band, monkeypatch):
  """BASELINE.json configs[2]'s larger classes ("SB2-synth" 107x78, "SB1-synth" 129x75 inside the
  ring) at `bench.py --config mixed`'s batch size per class: the register kernels that overlap sweeps in
  predicted blocks (step_two.hip; band: step_band.hip) and the LDS-grid kernel must take the same number
  of Gauss-Seidel sweeps for EVERY building at every step (tens of sweeps each, blocks overrunning now
  and then) and end within 1e-9 K of each other."""
  _need_gpu()
  if band:
    monkeypatch.setenv("SBSIM_BAND_PATH", "1")
  from sbsim_amd.floorplan import rectangular_floor_plan
  g = load("h2_sb1_r9_random.npz")
  plan = FloorPlan.from_file_input(rectangular_floor_plan(rooms, room_shape), Materials.sb1(), 10.0, 300.0)
  H, W = plan.shape
  T = 4
  gen = torch.Generator(device="cuda")
  gen.manual_seed(9)
  t0 = (294.0 + torch.randn((B, 1), generator=gen, device="cuda", dtype=torch.float64)).clamp(285.0, 305.0)
  init = (t0 + 0.05 * torch.randn((B, H * W), generator=gen, device="cuda", dtype=torch.float64)).contiguous()
  acts = torch.rand((T, B, 2), generator=gen, device="cuda", dtype=torch.float32) * 2.0 - 1.0
  sims = []
  for force_lds in (False, True):
    if force_lds:
      monkeypatch.setenv("SBSIM_FORCE_LDS_PATH", "1")
    sim = BatchedSimulator(plan, SimConfig.sb1(), B, float(g["h_conv"]))
    assert sim.launch_info["path"] == (0 if force_lds else 1)
    if not force_lds:
      assert sim.launch_info["kernel"] == (5 if band else 4)
    sim.reset(temps=init)
    assert torch.equal(sim.temps().reshape(B, -1), init)
    sims.append(sim)
  obs = [torch.zeros((B, s.O), dtype=torch.float32, device="cuda") for s in sims]
  rew = [torch.zeros((B,), dtype=torch.float32, device="cuda") for _ in sims]
  info = [torch.zeros((B, _ffi.SB_INFO_STRIDE), dtype=torch.float32, device="cuda") for _ in sims]
  total = 0.0
  for t in range(T):
    si = _step_in(g, 100 + t)
    for k, sim in enumerate(sims):
      sim.step(acts[t], si, obs[k], rew[k], info[k])
    assert torch.equal(info[0][:, 4], info[1][:, 4]), t          # sweeps, every building
    total += float(info[0][:, 4].sum())
    assert float((sims[0].zone_temps() - sims[1].zone_temps()).abs().max()) < 1e-9, t
  assert float((sims[0].temps() - sims[1].temps()).abs().max()) < 1e-9
  assert 5.0 < total / (B * T) <= 100.0
  for sim in sims:
    sim.close()


def test_full_size_isothermal_fixed_point():
  """A size-independent property at the full BASELINE batch: a building that is everywhere at
  the ambient temperature, with the air handler passing that air through (heating setpoint <=
  T_amb <= cooling setpoint) and every thermostat OFF, is a fixed point of the step: VAV power
  0, one Gauss-Seidel sweep, the grid unchanged to round-off -- for all 65,536 buildings, on
  both sweep kernels."""
  _need_gpu()
  B = 65536
  p = load("plan_r9_sb1.npz")
  g = load("h2_sb1_r9_random.npz")
  t_amb = 295.0
  for force_lds in (False, True):
    if force_lds:
      os.environ["SBSIM_FORCE_LDS_PATH"] = "1"
    try:
      sim = BatchedSimulator(_plan(p), SimConfig.sb1(), B, float(g["h_conv"]))
    finally:
      os.environ.pop("SBSIM_FORCE_LDS_PATH", None)
    sim.reset(initial_temp=t_amb)
    obs = torch.zeros((B, sim.O), dtype=torch.float32, device="cuda")
    rew = torch.zeros((B,), dtype=torch.float32, device="cuda")
    info = torch.zeros((B, _ffi.SB_INFO_STRIDE), dtype=torch.float32, device="cuda")
    si = _step_in(g, 100)            # mid-morning: comfort window 294..297 K around 295 K
    si.t_amb_now = si.t_amb_next = t_amb
    si.has_action = 0                 # setpoints stay at the constructor values (285 / 298 K)
    for _ in range(3):
      sim.step(None, si, obs, rew, info)
      assert bool((info[:, 4] == 1).all()) and bool((info[:, 5] == 1).all())
      assert float(sim.zone_power().abs().max()) < 1e-6
      assert float((sim.temps() - t_amb).abs().max()) < 1e-10
    assert bool((sim.modes() == 0).all())
    sim.close()


def test_environment_api_episode_bookkeeping():
  """environment.py:1165-1212,1311-1368: restart / N transitions / termination / auto-reset,
  plus observation normalisation and auxiliary features."""
  _need_gpu()
  g = load("h2_sb1_r9_const.npz")
  from sbsim_amd.environment import SB1_OBSERVATION_NORMALIZATION
  env = BatchedEnvironment(_plan(load("plan_r9_sb1.npz")), 4, num_days_in_episode=5 * 300 / 86400.0,
                           holiday_calendar=None, observation_normalization=SB1_OBSERVATION_NORMALIZATION,
                           collect_info=True)
  assert env.action_spec().shape == (2,) and env.observation_spec().shape == (46,)
  assert env.steps_per_episode == 5
  ts = env.reset()
  assert int(ts.step_type[0]) == 0 and ts.observation.shape == (4, 46)
  a = torch.tensor(np.tile(g["actions_norm"][0], (4, 1)), device="cuda")
  kinds = []
  for k in range(7):
    ts = env.step(a)
    kinds.append(int(ts.step_type[0]))
    if k == 0:
      o = ts.observation.cpu().numpy()[0]
      names = env.field_names
      j = names.index("boiler_id/supply_water_setpoint")
      assert abs(o[j] - (340.0 - 310.0) / 50.0) < 1e-6
      j = names.index("vav_room_1/zone_air_temperature_sensor")
      assert abs(o[j] - (np.float32(g["zone_temp_pre"][0][0]) - 190.0) / np.sqrt(np.float32(408.113303))) < 1e-5
      assert abs(float(ts.reward[0]) - float(g["reward"][0])) < 1e-6
      assert abs(o[names.index("hod_cos_000")] - np.cos(2 * np.pi * (7 * 3600 + 300) / 86400.0)) < 1e-6
  # 5 transitions, the terminal step, then the auto-reset (FIRST)
  assert kinds == [1, 1, 1, 1, 1, 2, 0], kinds
  env.close()


def test_gym_vector_view_autoresets_like_gymnasium():
  """GymVectorEnv: (obs, reward, terminated, truncated, info) over a short episode and the
  next-step autoreset; the same numbers as the TimeStep API."""
  _need_gpu()
  from sbsim_amd.environment import BatchedEnvironment, GymVectorEnv
  B = 4
  plan = _plan(load("plan_r9_sb1.npz"))
  steps = 5                                       # 5 transitions + the terminal step
  mk = lambda: BatchedEnvironment(plan, B, num_days_in_episode=steps * 300.0 / 86400.0, holiday_calendar=None)
  venv, ref = GymVectorEnv(mk()), mk()
  assert venv.num_envs == B and venv.single_action_shape == (2,) and venv.single_observation_shape == (ref.sim.O,)
  obs, info = venv.reset()
  ts = ref.reset()
  assert torch.equal(obs, ts.observation) and bool((info["step_type"] == 0).all())
  rs = np.random.RandomState(1)
  seen_terminal = 0
  for t in range(2 * (steps + 2)):
    a = torch.tensor(rs.uniform(-1, 1, size=(B, 2)).astype(np.float32), device="cuda")
    obs, rew, term, trunc, info = venv.step(a)
    ts = ref.step(a)
    assert torch.equal(obs, ts.observation) and torch.equal(rew, ts.reward)
    assert not bool(term.any())                     # the episode's time is up: a truncation, never a termination
    assert bool(trunc.all()) == bool((ts.step_type == 2).all()) and bool(trunc.all()) == bool(trunc.any())
    if bool(trunc.all()):
      seen_terminal += 1
      obs2, rew2, term2, trunc2, info2 = venv.step(a)    # next-step autoreset: first observation, zero reward
      ts2 = ref.step(a)
      assert bool((info2["step_type"] == 0).all()) and float(rew2.abs().max()) == 0.0 and not bool(term2.any() or trunc2.any())
      assert torch.equal(obs2, ts2.observation)
  assert seen_terminal == 2
  venv.close(); ref.close()


def test_abi_error_paths():
  _need_gpu()
  L = _ffi.load()
  assert L.sb_abi_version() == _ffi.SB_ABI_VERSION
  h = C.c_void_p()
  assert L.sb_create(None, None, None, 1, 0, C.byref(h)) == -1
  assert b"null" in L.sb_last_error()
  p = load("plan_small_test.npz")
  cfg = SimConfig(ahu_heating_air_temp_setpoint=300.0, ahu_cooling_air_temp_setpoint=290.0)
  with pytest.raises(_ffi.SbsimError, match="cooling_air_temp_setpoint"):
    BatchedSimulator(_plan(p), cfg, 2, 12.0)
  sim = BatchedSimulator(_plan(p), SimConfig(), 2, 12.0)
  with pytest.raises(ValueError):
    sim.step(torch.zeros((3, 2), device="cuda"), _ffi.StepIn(), None, torch.zeros(2, device="cuda"))


def _oracle_twin(plan, cfg, init_flat):
  oplan = orc.OraclePlan(plan.conductivity, plan.density, plan.heat_capacity, plan.exterior_space,
                         plan.zone_cell_lists(), plan.diffusers, plan.cv_size_cm, plan.floor_height_cm)
  c = cfg
  oprm = orc.OracleParams(
      dt=c.time_step_sec, conv_threshold=c.convergence_threshold, iter_limit=c.iteration_limit,
      vav_max_air_flow=c.vav_max_air_flow_rate, vav_max_water_flow=c.vav_reheat_max_water_flow_rate,
      ahu_recirc=c.ahu_recirculation, ahu_heat_sp=c.ahu_heating_air_temp_setpoint,
      ahu_cool_sp=c.ahu_cooling_air_temp_setpoint, ahu_dp=c.ahu_fan_differential_pressure,
      ahu_eff=c.ahu_fan_efficiency, blr_setpoint=c.boiler_reheat_water_setpoint,
      blr_head=c.boiler_water_pump_differential_head, blr_pump_eff=c.boiler_water_pump_efficiency,
      comfort_lo=c.comfort_temp_window[0], comfort_hi=c.comfort_temp_window[1],
      eco_lo=c.eco_temp_window[0], eco_hi=c.eco_temp_window[1],
      blr_heating_rate=c.boiler_heating_rate, blr_cooling_rate=c.boiler_cooling_rate, ahu_has_weather=1)
  return orc.OracleBuilding(oplan, oprm, 0.0, reset_temps=init_flat)


@pytest.mark.parametrize("rooms,room_shape,orientation,path", [
    ((8, 5), (12, 14), "auto", 0),     # "SB2-synth": 40 zones, 109x80 grid, 51 cell classes -> LDS grid (2 per CU, as many as the register variant)
    ((6, 4), (17, 21), "rows", 1),     # 111x91 inside the exterior ring, 24 zones, > 31 cell classes -> registers: step_band.hip on two wavefronts, 92 slots (until round 4: k_sweep_reg on two)
    ((4, 3), (25, 28), "rows", 0),     # 109x92: two of them fit a CU's LDS -> LDS grid preferred over the 96-slot pair variant
    ((14, 9), (8, 7), "auto", 0),      # "SB1-synth": 126 zones (the real SB1's VAV count), 131x78 grid, 3 bands
    ((2, 3), (9, 10), "rows", 1),      # small: 22x34 inside the ring -> registers, 1 wave, 66 slots
    ((5, 1), (12, 10), "generic", 0),  # H > 64, narrow grid, forced onto the generic (all-LDS) sweep
    ((2, 2), (5, 9), "rows", 1),       # 15x23 inside the ring -> registers, 1 wave, 32 slots
    ((2, 3), (20, 30), "rows", 30),    # 45x96 -> k_sweep_roll without tail rows (lanes 45..63 own pad rows); until round 4: k_sweep_reg, 96 slots
    ((2, 5), (30, 12), "columns", 1),  # transposed 68x65 -> registers, 2 waves (uneven split), 66 slots
    ((2, 3), (30, 30), "rows", 1),     # 65x96 -> registers, 1 wave + ONE tail row
    ((4, 5), (10, 8), "rows", 30),     # 47x48, 20 zones -> k_sweep_roll<64> (its 64-step period against a 112-step sweep of k_sweep_reg)
    ((4, 5), (10, 8), "rows", 32),     # ... and on k_sweep_roll<96> (SBSIM_NO_ROLL_64=1)
    ((2, 2), (30, 29), "rows", 30),    # 65x63 -> k_sweep_roll<64> + ONE tail row
    ((2, 3), (20, 24), "rows", 30),    # 45x78 -> k_sweep_roll<80>
    ((2, 3), (30, 24), "rows", 30),    # 65x78 -> k_sweep_roll<80> + ONE tail row
    ((2, 3), (20, 21), "rows", 30),    # 45x69 -> k_sweep_roll<72> (all of A in LDS: 70 + 2 slots)
    ((2, 3), (30, 27), "rows", 30),    # 65x87 -> k_sweep_roll<88> + ONE tail row
    ((4, 5), (10, 8), "rows", 31),     # ... and on k_sweep_reg (SBSIM_NO_ROLL_SMALL=1): zone reduce in two 16-zone passes
    ((4, 2), (30, 17), "columns", 0),  # the same family on the LDS-grid kernel, lanes = columns
    # 67..130 rows, <= 80 columns: step_two.hip (path 4: one wavefront, two rows per lane) is the library's
    # choice; step_band.hip (path 5: two wavefronts, one row per lane) sits behind SBSIM_BAND_PATH=1
    ((8, 5), (12, 14), "auto", 5),     # "SB2-synth" 107x78 inside the ring: 64 + 43 rows, 80 slots
    ((14, 9), (8, 7), "auto", 5),      # "SB1-synth" 129x75, 137 cell classes: 64 + 64 rows + ONE tail row, 76 slots
    ((5, 3), (24, 24), "rows", 5),     # 128x78: 64 + 64 rows, no tail row
    ((4, 4), (15, 17), "rows", 5),     # 67x75: 64 + 3 rows (wavefront 1's sweep ends early)
    ((14, 7), (8, 10), "rows", 5),     # 129x80: 80 slots + ONE tail row
    ((8, 5), (12, 14), "auto", 4),     # "SB2-synth": two rows per lane (54 lanes, the last with one row), 80 slots
    ((14, 9), (8, 7), "auto", 4),      # "SB1-synth": two rows per lane + ONE tail row, 76 slots
    ((5, 3), (24, 24), "rows", 4),     # 128x78: two rows per lane, all 64 lanes, no tail row
    ((4, 4), (15, 17), "rows", 4),     # 67x75: two rows per lane, 34 lanes (the sweep ends early)
    ((14, 7), (8, 10), "rows", 4),     # 129x80: two rows per lane, 80 slots + ONE tail row
    ((8, 5), (12, 10), "rows", 4),     # 107x58: two rows per lane, 64 slots
    ((14, 7), (8, 7), "rows", 4),      # 129x59: 64 slots + ONE tail row
    # 131..258 rows, <= 96 columns: step_band.hip with three or four wavefronts per building (the library's choice)
    ((9, 4), (16, 17), "rows", 53),    # 156x75 inside the exterior ring: 64 + 64 + 28 rows, 76 slots
    ((10, 4), (18, 20), "rows", 53),   # 193x87: 3 x 64 rows + ONE tail row, 88 slots (16 of them in AGPRs)
    ((10, 4), (19, 20), "rows", 54),   # 203x87, 40 zones (VERDICT r3 #4's 200 x 90 class): 3 x 64 + 11 rows, 88 slots
    ((12, 3), (20, 24), "rows", 54),   # 255x78: 3 x 64 + 63 rows, 80 slots
    ((4, 10), (20, 19), "columns", 54),  # the 203x87 plan transposed in the file: the same kernel, lanes = file columns
    ((9, 4), (16, 19), "rows", 53),    # 156x83: 84 slots
    ((9, 4), (16, 15), "rows", 53),    # 156x67: 68 slots
    ((12, 3), (20, 22), "rows", 54),   # 255x72: 72 slots (the class words' read-ahead runs on, as with 96)
    ((10, 4), (18, 21), "rows", 53),   # 193x91: 92 slots + ONE tail row
    ((10, 4), (19, 22), "rows", 54),   # 203x95: 96 slots (24 in AGPRs; the class words' read-ahead runs on across a period's end)
])
def test_mixed_floor_plans_against_oracle(rooms, room_shape, orientation, path, monkeypatch):
  """BASELINE.json configs[2] semantics: other floor-plan classes (different H x W and zone
  counts) through the same C ABI, each checked against its CPU-oracle twin.  path: 0 the LDS-grid
  kernel, 1 the library's choice among the register kernels, 4 / 5 that kernel (sb_sweep_kernel)."""
  from sbsim_amd.floorplan import rectangular_floor_plan
  kern, waves = None, None
  if path == 31:      # k_sweep_reg<NR,1> where the library would take k_sweep_roll
    monkeypatch.setenv("SBSIM_NO_ROLL_SMALL", "1")
    kern, path = 1, 1
  elif path == 30:    # k_sweep_roll by the library's own choice (a plan of <= 66 rows)
    kern, path = 3, 1
  elif path == 32:    # the 96-slot instantiation where the library would take the 64-slot one
    monkeypatch.setenv("SBSIM_NO_ROLL_64", "1")
    kern, path = 3, 1
  elif path >= 50:      # 53 / 54: step_band.hip by the library's own choice, three / four wavefronts per building
    kern, waves, path = 5, path - 50, 1
  elif path >= 4:
    kern, path = path, 1
    if kern == 5:
      monkeypatch.setenv("SBSIM_BAND_PATH", "1")
  _check_plan_against_oracle(rectangular_floor_plan(rooms, room_shape), rooms[0] * rooms[1], orientation, path, monkeypatch,
                             expect_kernel=kern, expect_waves=waves)


@pytest.mark.gpu
@pytest.mark.parametrize("kern", [5, 4])
@pytest.mark.parametrize("haste,slack", [(1.0, 1.0), (1.0, 0.0), (1.0, -100.0)])
def test_block_kernels_tail_rows_and_overrun_blocks(haste, slack, kern, monkeypatch):
  """step_band.hip / step_two.hip: a 130-row plan (SB1-synth with a 3-CV wall at the bottom: TWO tail
  rows), and the prediction switched off (slack 0 / -100: every period rolls, so every step runs past
  its last sweep and is run again from the stored grid) -- sweep counts and temperatures must not notice."""
  from sbsim_amd.floorplan import rectangular_floor_plan
  fp = rectangular_floor_plan((14, 9), (8, 7))
  fp = np.insert(fp, fp.shape[0] - 2, fp[-2], axis=0)
  assert fp.shape == (132, 77)
  monkeypatch.setenv("SBSIM_DEBUG_PRED_HASTE", str(haste))
  monkeypatch.setenv("SBSIM_DEBUG_PRED_SLACK", str(slack))
  if kern == 5:
    monkeypatch.setenv("SBSIM_BAND_PATH", "1")
  _check_plan_against_oracle(fp, 126, "rows", 1, monkeypatch, expect_steps=76 + 64 - 1 + 8 if kern == 4 else 76 + 8,
                             expect_kernel=kern)


@pytest.mark.gpu
@pytest.mark.parametrize("rooms,shape,zones", [((14, 9), (8, 7), 126), ((8, 5), (12, 14), 40)])
@pytest.mark.parametrize("kappa", ["off", "0.9", "2.5", "40"])
def test_two_rows_kernel_measure_free_periods(kappa, rooms, shape, zones, monkeypatch):
  """step_two_impl.h (round 6): rolling periods far from a step's last sweep run without max|delta| (max|delta| never
  grows from one sweep to the next, so one measurement above the threshold proves every sweep before it).  With the
  next measurement placed far too late (kappa 2.5 / 40: the first measurement behind a measure-free stretch is
  already at or below the threshold) the block is run again with every period measuring; switched off
  (SBSIM_TWO_NO_SKIP=1) the kernel is round 5's.  Sweep counts EQUAL to the oracle's and grids within 1e-8 K either
  way: the schedule decides the speed, never the result."""
  from sbsim_amd.floorplan import rectangular_floor_plan
  if kappa == "off":
    monkeypatch.setenv("SBSIM_TWO_NO_SKIP", "1")
  else:
    monkeypatch.setenv("SBSIM_DEBUG_SKIP_KAPPA", kappa)
  _check_plan_against_oracle(rectangular_floor_plan(rooms, shape), zones, "auto", 1, monkeypatch, expect_kernel=4, T=16)


@pytest.mark.parametrize("kern", [5, 4])
@pytest.mark.parametrize("limit", [1, 2, 3, 7, 12])
def test_block_kernels_iteration_limit(limit, kern, monkeypatch):
  """simulator.py:348-368 with a limit that bites on the kernels that overlap sweeps in blocks
  ("SB2-synth" needs ~18 sweeps per step): blocks are clipped to the sweeps that are left, the limit
  ends a step at a block's exact stop, sweep counts and grids against the oracle."""
  from sbsim_amd.floorplan import rectangular_floor_plan
  if kern == 5:
    monkeypatch.setenv("SBSIM_BAND_PATH", "1")
  _check_plan_against_oracle(rectangular_floor_plan((8, 5), (12, 14)), 40, "auto", 1, monkeypatch,
                             iteration_limit=limit, expect_kernel=kern)


@pytest.mark.gpu
@pytest.mark.parametrize("haste,slack", [(1.0, 1.0), (1.0, 0.0), (1.0, -100.0)])
def test_band_kernel_four_wavefronts_tail_rows_and_overrun_blocks(haste, slack, monkeypatch):
  """step_band.hip at its largest: 258 rows inside the exterior ring (4 x 64 + TWO tail rows; a 5-CV wall at the
  bottom of a 255-row plan), wavefront 0 deciding three sweeps before the last wavefront's part exists; and the
  prediction switched off (slack 0 / -100: every period rolls, every step runs past its last sweep and is run
  again from the stored grid).  Sweep counts and temperatures must not notice."""
  from sbsim_amd.floorplan import rectangular_floor_plan
  fp = rectangular_floor_plan((12, 3), (20, 24))
  for _ in range(3):
    fp = np.insert(fp, fp.shape[0] - 2, fp[-2], axis=0)
  assert fp.shape == (260, 80)
  monkeypatch.setenv("SBSIM_DEBUG_PRED_HASTE", str(haste))
  monkeypatch.setenv("SBSIM_DEBUG_PRED_SLACK", str(slack))
  _check_plan_against_oracle(fp, 36, "rows", 1, monkeypatch, expect_steps=80 + 8, expect_kernel=5, expect_waves=4, B=4, T=10)


@pytest.mark.gpu
@pytest.mark.parametrize("limit", [1, 2, 3, 5, 9])
def test_band_kernel_four_wavefronts_iteration_limit(limit, monkeypatch):
  """simulator.py:348-368 with a limit that bites, four wavefronts per building (203 x 87 inside the ring)."""
  from sbsim_amd.floorplan import rectangular_floor_plan
  _check_plan_against_oracle(rectangular_floor_plan((10, 4), (19, 20)), 40, "rows", 1, monkeypatch,
                             iteration_limit=limit, expect_kernel=5, expect_waves=4, B=4, T=8)


def _need_experimental_kernels(variant):
  """The multi-sweep / overlapped streaming kernels are exact but slower than k_sweep_stream: since round 6 they are in the
  library only when it is built with SBSIM_BUILD_EXPERIMENTAL=1 (sbsim_amd/build.py); the default library ignores the flags."""
  if variant and not _ffi.load().sb_has_experimental_kernels():
    pytest.skip("built without SBSIM_BUILD_EXPERIMENTAL=1: step_stream_ms.hip / k_sweep_stream_roll are not in this library")


def _check_plan_against_oracle(file_plan, n_zones, orientation, path, monkeypatch, expect_steps=None,
                               iteration_limit=None, expect_kernel=None, B=6, T=14, expect_waves=None):
  _need_gpu()
  g = load("h2_sb1_r9_random.npz")
  plan = FloorPlan.from_file_input(file_plan, Materials.sb1(), 10.0, 300.0)
  H, W = plan.shape
  rs = np.random.RandomState(11)
  init = np.clip(294.0 + 2.0 * rs.randn(B, 1) + 0.2 * rs.randn(B, H * W), 285.0, 305.0)
  acts = rs.uniform(-1, 1, size=(T, B, 2)).astype(np.float32)
  cfg = SimConfig.sb1()
  if iteration_limit is not None:
    import dataclasses
    cfg = dataclasses.replace(cfg, iteration_limit=iteration_limit)
  if orientation == "generic":
    monkeypatch.setenv("SBSIM_FORCE_GENERIC_SWEEP", "1")
    orientation = "rows"
  if path == 0:
    monkeypatch.setenv("SBSIM_FORCE_LDS_PATH", "1")
  if path == 2 and H * W < 100000:
    monkeypatch.setenv("SBSIM_FORCE_STREAM_PATH", "1")   # small plans reach the streaming kernel only when told to
  sim = BatchedSimulator(plan, cfg, B, float(g["h_conv"]), orientation=orientation)
  assert sim.Z == n_zones and sim.launch_info["path"] == path
  if expect_steps is not None:
    assert sim.launch_info["sweep_steps"] == expect_steps   # the two-rows kernel with two tail rows
  if expect_kernel is not None:
    assert sim.launch_info["kernel"] == expect_kernel
  if expect_waves is not None:
    assert sim.launch_info["waves_per_building"] == expect_waves
  sim.reset(temps=torch.tensor(init, dtype=torch.float64, device="cuda"))
  twins = [_oracle_twin(plan, cfg, init[b]) for b in range(B)]
  obs = torch.zeros((B, sim.O), dtype=torch.float32, device="cuda")
  rew = torch.zeros((B,), dtype=torch.float32, device="cuda")
  info = torch.zeros((B, _ffi.SB_INFO_STRIDE), dtype=torch.float32, device="cuda")
  lo, hi = cfg.action_ranges
  for t in range(T):
    tt = 96 + t   # mid-morning: occupied, comfort mode
    sim.step(torch.tensor(acts[t], device="cuda"), _step_in(g, tt), obs, rew, info)
    i = info.cpu().numpy().astype(np.float64)
    zt = sim.zone_temps().cpu().numpy()
    r = rew.cpu().numpy()
    for b in range(B):
      a = acts[t, b]
      native = [np.float32((float(a[0]) + 1.0) / 2.0 * (lo[1] - lo[0]) + lo[0]),
                np.float32((float(a[1]) + 1.0) / 2.0 * (hi[1] - hi[0]) + hi[0])]
      o = twins[b].step(
          now_ts=300.0 * t, t_amb_now=float(g["t_amb_now"][tt]), h_conv=float(g["h_conv"]),
          t_amb_next=float(g["t_amb_next"][tt]), comfort_now=bool(g["comfort_now"][tt]),
          comfort_prev=g["comfort_prev"][tt] == 1, comfort_next=bool(g["comfort_next"][tt]),
          occupancy=float(g["occupancy"][tt]), e_price=float(g["e_price"][tt]),
          e_carbon=float(g["e_carbon"][tt]), g_price=float(g["g_price"][tt]),
          g_carbon=float(g["g_carbon"][tt]), action=native, observe=True)
      assert i[b, 4] == o["n_sweeps"], (t, b, i[b, 4], o["n_sweeps"])
      assert np.abs(zt[b] - o["zone_temp_post"]).max() < T_TOL, (t, b)
      ref = np.array([o["blower_rate"], o["ac_rate"], o["gas_rate"], o["pump_rate"]], np.float64)
      assert np.allclose(i[b, :4], ref, rtol=2e-6, atol=1e-6), (t, b)
      assert abs(float(r[b]) - o["reward"]) < 2e-6, (t, b)
  grid = sim.temps().cpu().numpy()
  for b in range(B):
    assert np.abs(grid[b] - twins[b].grid()).max() < T_TOL, b


@pytest.mark.parametrize("rooms,room_shape,orientation,waves", [
    ((3, 3), (20, 30), "rows", 2),      # R9: 66 x 96 inside the ring -> two wavefronts (64 + 2 rows)
    ((3, 3), (20, 30), "columns", 2),   # ... transposed: 96 x 66 (64 + 32 rows), 72 slots for 66 columns
    ((8, 5), (12, 14), "rows", 2),      # "SB2-synth" 107 x 78, 40 zones
    ((14, 9), (8, 7), "rows", 3),       # "SB1-synth" 129 x 75: three wavefronts (the last with one row), 126 zones, 137 classes
    ((2, 2), (5, 9), "rows", 1),        # 15 x 23: one wavefront, no seam
    ((10, 2), (19, 40), "rows", 4),     # 203 x 85 inside the ring: four wavefronts
])
@pytest.mark.parametrize("multi_sweep", [False, True, "overlapped"])
def test_streaming_kernel_against_oracle(rooms, room_shape, orientation, waves, multi_sweep, monkeypatch):
  """step_stream.hip (the grid stays in global memory, up to 16 wavefronts per building, seam rows through
  LDS) forced onto floor plans the other kernels own: sweep counts EQUAL, temperatures within 1e-8 K of
  the CPU-oracle twins."""
  from sbsim_amd.floorplan import rectangular_floor_plan
  _need_experimental_kernels(multi_sweep)
  if multi_sweep == "overlapped":   # step_stream.hip's k_sweep_stream_roll: sweeps overlapped on two grids that take turns (opt-in)
    monkeypatch.setenv("SBSIM_STREAM_ROLL", "1")
  elif multi_sweep:   # step_stream_ms.hip: up to four sweeps per pass over the grid (parallelogram tiling; opt-in)
    monkeypatch.setenv("SBSIM_STREAM_MS", "1")
  _check_plan_against_oracle(rectangular_floor_plan(rooms, room_shape), rooms[0] * rooms[1], orientation, 2, monkeypatch,
                             expect_kernel=6, expect_waves=waves)


@pytest.mark.parametrize("multi_sweep", [False, True, "overlapped"])
def test_floor_plan_beyond_one_cu_against_oracle(multi_sweep, monkeypatch):
  """A floor plan no CU can hold (missing #1): 299 x 401 control volumes, 126 zones -- 0.96 MB of
  float64 state per building.  The library picks the streaming kernel by itself (five wavefronts per
  building); sweep counts EQUAL and temperatures within 1e-8 K of the CPU oracle."""
  from sbsim_amd.floorplan import rectangular_floor_plan
  fp = rectangular_floor_plan((14, 9), (20, 43))
  assert fp.shape == (299, 401)
  _need_experimental_kernels(multi_sweep)
  if multi_sweep == "overlapped":
    monkeypatch.setenv("SBSIM_STREAM_ROLL", "1")
  elif multi_sweep:
    monkeypatch.setenv("SBSIM_STREAM_MS", "1")
  _check_plan_against_oracle(fp, 126, "rows", 2, monkeypatch, expect_kernel=6, expect_waves=5, B=2, T=3)


@pytest.mark.parametrize("kernel", ["auto", "lds", "stream"])
def test_rectangular_building_without_exterior_ring(kernel, monkeypatch):
  """SURVEY 8c G7 on the HIP path: the deprecated rectangular ``Building`` (building.py:394-605) has no
  exterior-space ring -- its neighbour lists keep every in-bounds cell (building.py:529-545), so the
  trim box is the whole grid and the corner / edge formulas sit on its rim.  One cold thermostat-only
  step of the 3x3-room scenario: 100 sweeps, the reference's KAT 301.895482 (simulator_test.py:955-987)
  and its final grid (tests/golden/rect_building_cold200.npz, from the reference's own Building)."""
  _need_gpu()
  import json
  g = load("rect_building_cold200.npz")
  h = load("h1_r9_test_cold200.npz")
  H, W = g["conductivity"].shape
  rs, bs = g["room_shape"], g["building_shape"]
  label = np.full((H, W), -1, dtype=np.int64)
  for zx in range(bs[0]):
    for zy in range(bs[1]):
      x0, y0 = zx * (rs[0] + 1) + 2, zy * (rs[1] + 1) + 2  # building.py:167-180
      label[x0:x0 + rs[0], y0:y0 + rs[1]] = zx * bs[1] + zy
  plan = FloorPlan(conductivity=g["conductivity"], heat_capacity=g["heat_capacity"], density=g["density"],
                   exterior_space=np.zeros((H, W), bool), zone_label=label, diffusers=g["diffusers"],
                   cv_size_cm=float(g["cv_size_cm"]), floor_height_cm=float(g["floor_height_cm"]),
                   skip_exterior_neighbors=False)
  prm = json.loads(str(h["params_json"]))
  cfg = SimConfig(
      time_step_sec=prm["dt"], convergence_threshold=prm["conv_threshold"], iteration_limit=prm["iter_limit"],
      comfort_temp_window=(prm["comfort_lo"], prm["comfort_hi"]), eco_temp_window=(prm["eco_lo"], prm["eco_hi"]),
      vav_max_air_flow_rate=prm["vav_max_air_flow"], vav_reheat_max_water_flow_rate=prm["vav_max_water_flow"],
      ahu_recirculation=prm["ahu_recirc"], ahu_heating_air_temp_setpoint=prm["ahu_heat_sp"],
      ahu_cooling_air_temp_setpoint=prm["ahu_cool_sp"], ahu_fan_differential_pressure=prm["ahu_dp"],
      ahu_fan_efficiency=prm["ahu_eff"], ahu_max_air_flow_rate=prm["ahu_max_flow"], ahu_has_weather_sensor=False,
      boiler_reheat_water_setpoint=prm["blr_setpoint"], boiler_water_pump_differential_head=prm["blr_head"],
      boiler_water_pump_efficiency=prm["blr_pump_eff"], boiler_heating_rate=prm["blr_heating_rate"],
      boiler_cooling_rate=prm["blr_cooling_rate"], initial_temp=200.0)
  if kernel == "lds":
    monkeypatch.setenv("SBSIM_FORCE_LDS_PATH", "1")
  if kernel == "stream":
    monkeypatch.setenv("SBSIM_FORCE_STREAM_PATH", "1")
  B = 3
  sim = BatchedSimulator(plan, cfg, B, 12.0)
  assert sim.launch_info["path"] == {"auto": 1, "lds": 0, "stream": 2}[kernel]
  sim.reset()
  rew = torch.zeros((B,), dtype=torch.float32, device="cuda")
  info = torch.zeros((B, _ffi.SB_INFO_STRIDE), dtype=torch.float32, device="cuda")
  si = _ffi.StepIn()
  si.t_amb_now = si.t_amb_next = 296.0
  si.comfort_now = si.comfort_next = 0
  si.comfort_prev = -1
  si.has_action = 0
  si.occupancy, si.e_price, si.e_carbon, si.g_price, si.g_carbon = 1.0, 1e-8, 1e-8, 1e-8, 1e-8
  sim.step(None, si, None, rew, info)
  i = info.cpu().numpy()
  assert (i[:, 4] == h["n_sweeps"][0]).all(), i[:, 4]      # the same scenario on both reference building classes
  sc = sim.scalars().cpu().numpy()
  assert np.abs(sc[:, 7] - float(g["blr_return_temp"])).max() < 1e-9
  assert abs(float(sc[0, 7]) - 301.895482) < 1e-5
  assert np.abs(sim.temps().cpu().numpy().reshape(B, H, W) - g["final_grid"]).max() < T_TOL
  sim.close()


def _run_bench(args, timeout=600):
  import json, subprocess, sys
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  out = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, capture_output=True, text=True, timeout=timeout, cwd=root)
  assert out.returncode == 0, out.stderr[-2000:]
  lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
  assert len(lines) == 1, out.stdout[-2000:]
  return json.loads(lines[0])


def test_mixed_environment_gives_every_building_its_own_generator_stream():
  """a seeded convection shuffle handed to MixedBatchedEnvironment must not repeat its streams from class
  to class or from rank to rank -- class k's building i draws as building sum(totals[:k]) + (its index inside the
  class) of the whole mixed batch.  Two classes of the SAME plan: class 1 equals a BatchedEnvironment whose generator
  starts at building 4, and differs from class 0; rank 1 of 2 continues where rank 0 stopped."""
  _need_gpu()
  from sbsim_amd import host_inputs
  from sbsim_amd.environment import BatchedEnvironment, MixedBatchedEnvironment
  from sbsim_amd.floorplan import FloorPlan, Materials, rectangular_floor_plan
  plan = FloorPlan.from_file_input(rectangular_floor_plan((2, 2), (8, 9)), Materials.sb1(), 10.0, 300.0)
  conv = lambda first=0: host_inputs.StochasticConvectionSimulator(1.0, 3, 77, first_building=first)
  kw = dict(holiday_calendar=None)
  a = torch.zeros((4, 2), device="cuda")

  def grids(env, n=6):
    env.reset()
    for _ in range(n):
      env.step(a[:env.batch_size])
    return env.sim.temps()

  menv = MixedBatchedEnvironment([(plan, 4), (plan, 4)], convection_simulator=conv(), **kw)
  menv.reset()
  for _ in range(6):
    menv.step(torch.zeros((8, 2), device="cuda"))
  g0, g1 = menv.envs[0].sim.temps(), menv.envs[1].sim.temps()
  assert not torch.equal(g0, g1)
  ref0 = BatchedEnvironment(plan, 4, convection_simulator=conv(0), **kw)
  ref1 = BatchedEnvironment(plan, 4, convection_simulator=conv(4), **kw)
  assert torch.equal(g0, grids(ref0)) and torch.equal(g1, grids(ref1))
  menv.close()
  # rank 1 of 2: class 0's buildings 2..3 and class 1's buildings 2..3 = global buildings 2..3 and 6..7
  m1 = MixedBatchedEnvironment([(plan, 4), (plan, 4)], rank=1, world=2, convection_simulator=conv(), **kw)
  m1.reset()
  for _ in range(6):
    m1.step(torch.zeros((4, 2), device="cuda"))
  assert torch.equal(m1.envs[0].sim.temps(), g0[2:]) and torch.equal(m1.envs[1].sim.temps(), g1[2:])
  with pytest.raises(ValueError, match="at least one building per rank"):
    MixedBatchedEnvironment([(plan, 4), (plan, 1)], rank=0, world=2, **kw)
  m1.close()
  ref0.close()
  ref1.close()


def test_mixed_environment_equals_its_classes_stepped_alone():
  """MixedBatchedEnvironment (one env over several floor-plan classes, BASELINE.json configs[2]): every class's
  slice of the TimeStep equals what a BatchedEnvironment of that class alone returns for the same actions --
  observations in the first `observation_widths[k]` columns, zeros after -- over an episode end and the reset
  that follows; the classes share the clock."""
  _need_gpu()
  from sbsim_amd.environment import BatchedEnvironment, MixedBatchedEnvironment
  from sbsim_amd.floorplan import FloorPlan, Materials, rectangular_floor_plan
  plans = [FloorPlan.from_file_input(rectangular_floor_plan(r, sh), Materials.sb1(), 10.0, 300.0)
           for r, sh in (((3, 3), (20, 30)), ((2, 2), (5, 9)), ((8, 5), (12, 14)))]
  counts, steps = [5, 7, 3], 4
  kw = dict(num_days_in_episode=steps * 300.0 / 86400.0, holiday_calendar=None)
  menv = MixedBatchedEnvironment(list(zip(plans, counts)), **kw)
  refs = [BatchedEnvironment(p, n, **kw) for p, n in zip(plans, counts)]
  assert menv.batch_size == 15 and menv.slices == [(0, 5), (5, 12), (12, 15)]
  assert menv.observation_widths == [r.sim.O for r in refs] and menv.observation_spec().shape == (max(menv.observation_widths),)
  assert menv.class_of_building.cpu().tolist() == [0] * 5 + [1] * 7 + [2] * 3
  rs = np.random.RandomState(3)

  def compare(ts, rts):
    for k, ((lo, hi), r) in enumerate(zip(menv.slices, rts)):
      w = menv.observation_widths[k]
      assert torch.equal(ts.observation[lo:hi, :w], r.observation) and not bool(ts.observation[lo:hi, w:].any())
      assert torch.equal(ts.reward[lo:hi], r.reward) and torch.equal(ts.discount[lo:hi], r.discount)
      assert torch.equal(ts.step_type[lo:hi], r.step_type)

  compare(menv.reset(), [r.reset() for r in refs])
  for t in range(2 * steps + 3):                    # two episode ends and the resets that follow
    a = torch.tensor(rs.uniform(-1, 1, size=(15, 2)).astype(np.float32), device="cuda")
    compare(menv.step(a), [r.step(a[lo:hi].contiguous()) for (lo, hi), r in zip(menv.slices, refs)])
  with pytest.raises(ValueError, match="action must be"):
    menv.step(torch.zeros((14, 2), device="cuda"))
  menv.close()
  for r in refs:
    r.close()


def test_bench_mixed_config_line_and_its_twins():
  """`bench.py --config mixed` (BASELINE.json configs[2]) at the full batch, two timed rounds: the JSON line
  parses, names the three classes' kernels, and 64 buildings per class agree with their CPU-oracle twins
  (sweep counts EQUAL, zone temperatures within 1e-8 K) at every step of the run."""
  _need_gpu()
  d = _run_bench(["--config", "mixed", "--buildings", "65535", "--steps", "2", "--warmup", "1", "--check-buildings", "64"])
  assert d["n_gpus"] == 1 and d["steps"] == 2 and d["unit"] == "zone-updates/s" and d["value"] > 1e7
  assert d["timed_through"] == "MixedBatchedEnvironment.step()" and d["config"]["observation_width"] == 3 * 126 + 19
  cl = d["config"]["classes"]
  assert set(cl) == {"R9", "SB2-synth", "SB1-synth"}
  for name, c in cl.items():
    assert c["buildings"] == 21845 and c["sweep_kernel_ms_alone"] > 0 and c["step_ms_in_round"] > 0
    par = c["parity_vs_oracle"]
    assert par["buildings"] == 64 and par["steps"] == 6            # 1 warm-up + 2 timed + 3 one-class-at-a-time rounds
    assert par["sweep_count_mismatches"] == 0 and par["max_abs_dT_zone_K"] < T_TOL, (name, par)


def test_bench_policy_config_line_and_its_twins():
  """`bench.py --config policy` (BASELINE.json configs[4]: the batch driven by a SAC-shaped actor on the
  same GPU): the line parses and 64 buildings agree with their oracle twins fed the actor's actions."""
  _need_gpu()
  d = _run_bench(["--config", "policy", "--steps", "2", "--warmup", "2", "--check-buildings", "64"])
  assert d["n_gpus"] == 1 and d["gathered_returns"] == 65536 and d["value"] > 1e7
  par = d["config"]["parity_vs_oracle"]
  assert par["buildings"] == 64 and par["steps"] == 2 + 2 + 8   # warm-up + timed + the eight profiled steps behind them (sweep_kernel_ms)
  assert d["roofline"]["sweep_kernel_ms"] > 0 and 0.5 < d["roofline"]["hbm_TBps_real"] < 8.0
  assert par["sweep_count_mismatches"] == 0 and par["max_abs_dT_zone_K"] < T_TOL, par


def test_bench_two_ranks_sharing_one_gpu(monkeypatch):
  """The N > 1 code path of bench.py on a one-GPU box (SBSIM_BENCH_SHARE_GPU=1: both ranks on device 0, gloo
  with host-staged collectives instead of RCCL): self-launch under torch.distributed.run, per-rank shards
  and seeds, barrier-bracketed timing, max over ranks, the gathered returns of both ranks, one JSON line."""
  _need_gpu()
  monkeypatch.setenv("SBSIM_BENCH_SHARE_GPU", "1")
  d = _run_bench(["--gpus", "2", "--buildings", "8192", "--steps", "3", "--warmup", "2", "--no-cpu-baseline"])
  assert d["n_gpus"] == 2 and d["gathered_returns"] == 16384 and d["scaling"] == "weak"
  assert d["config"]["buildings_per_gpu"] == 8192 and d["value"] > 1e6 and "share a device" in d["config"]["parallelism"]
  # the preflight check (world size, device per rank, 256 KiB all_gather, all_reduce(MAX)) ran first; per-rank times
  assert d["rccl_ranks"] == 2 and d["preflight"]["ok"] is True and d["preflight"]["devices"] == ["cuda:0", "cuda:0"]
  assert len(d["per_rank_ms_per_step"]) == 2 and len(d["per_rank_sweep_kernel_ms"]) == 2
  p = _run_bench(["--config", "policy", "--gpus", "2", "--buildings", "4096", "--steps", "2", "--warmup", "1"])
  assert p["n_gpus"] == 2 and p["gathered_returns"] == 8192
  # configs[2] on two ranks: every class block-partitioned over the ranks on its own, so each rank holds the
  # same class mix; the gather puts the returns back into global class-major order
  m = _run_bench(["--config", "mixed", "--gpus", "2", "--buildings", "3072", "--steps", "2", "--warmup", "1", "--check-buildings", "4"])
  assert m["n_gpus"] == 2 and m["gathered_returns"] == 2 * 3 * 1024 and m["rccl_ranks"] == 2
  assert m["config"]["class_totals"] == [2048, 2048, 2048] and m["config"]["class_ranges_rank0"] == [[0, 1024]] * 3
  assert len(m["per_rank_ms_per_step"]) == 2
  for name, c in m["config"]["classes"].items():
    assert c["buildings"] == 1024, name
    par = c["parity_vs_oracle"]
    assert par["sweep_count_mismatches"] == 0 and par["max_abs_dT_zone_K"] < T_TOL, (name, par)


def test_bench_eight_ranks_sharing_one_gpu(monkeypatch):
  """The driver's 8-GPU line rehearsed on real HIP state: eight ranks of bench.py under
  torch.distributed.run, all on device 0 (SBSIM_BENCH_SHARE_GPU=1: gloo with host-staged collectives -- RCCL refuses two
  ranks on one device), 8,192 buildings per rank: eight library handles and eight streams on one device, per-rank
  shards and seeds, the preflight, barrier-bracketed timing, the gather of 65,536 returns -- and configs[2] with every
  class sharded eight ways and gathered by class.  What the first real 8-GPU run adds to this is RCCL and nothing else."""
  _need_gpu()
  monkeypatch.setenv("SBSIM_BENCH_SHARE_GPU", "1")
  d = _run_bench(["--gpus", "8", "--buildings", "8192", "--steps", "3", "--warmup", "2", "--no-cpu-baseline"])
  assert d["n_gpus"] == 8 and d["gathered_returns"] == 65536 and d["scaling"] == "weak" and d["rccl_ranks"] == 8
  assert d["preflight"]["ok"] is True and d["preflight"]["devices"] == ["cuda:0"] * 8
  assert len(d["per_rank_ms_per_step"]) == 8 and len(d["per_rank_sweep_kernel_ms"]) == 8 and d["value"] > 1e6
  m = _run_bench(["--config", "mixed", "--gpus", "8", "--buildings", "3072", "--steps", "2", "--warmup", "1", "--check-buildings", "2"])
  assert m["n_gpus"] == 8 and m["gathered_returns"] == 8 * 3 * 1024 and m["rccl_ranks"] == 8
  assert m["config"]["class_totals"] == [8192] * 3 and m["config"]["class_ranges_rank0"] == [[0, 1024]] * 3
  assert len(m["per_rank_ms_per_step"]) == 8
  for name, c in m["config"]["classes"].items():
    par = c["parity_vs_oracle"]
    assert c["buildings"] == 1024 and par["sweep_count_mismatches"] == 0 and par["max_abs_dT_zone_K"] < T_TOL, (name, par)


def test_bench_through_the_public_env_api_costs_the_same():
  """The headline line steps through BatchedSimulator.step(phases=...) so that HIP events can bracket the
  sweep kernel; `--through-env-api` times BatchedEnvironment.step() itself (host-side step inputs, sb_step,
  TimeStep): within 3 % of the phases path."""
  _need_gpu()
  import json
  common = ["--gpus", "1", "--steps", "20", "--warmup", "30", "--no-cpu-baseline", "--no-also"]
  seen = []
  for attempt in range(3):   # three interleaved pairs of runs, always: the verdict is on their medians, not on the luckiest pair
    a = _run_bench(common)
    b = _run_bench(common + ["--through-env-api"])
    assert b["timed_through"] == "BatchedEnvironment.step()" and a["timed_through"].startswith("BatchedSimulator.step")
    assert abs(b["config"]["mean_sweeps_per_env_step"] - a["config"]["mean_sweeps_per_env_step"]) < 1e-9
    seen.append((a["ms_per_step"], b["ms_per_step"]))
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
  with open(os.path.join(root, "gpurun_out", "env_api_pairs.json"), "w") as fh:   # (copied into profiles/ by the round's collection)
    json.dump({"command": "bench.py " + " ".join(common) + " [--through-env-api]", "ms_per_step_pairs_phases_vs_env_api": seen}, fh)
  med = lambda v: sorted(v)[len(v) // 2]
  assert med([y for _, y in seen]) < 1.03 * med([x for x, _ in seen]), seen


def test_bench_two_gpus_when_the_box_has_them():
  """`bench.py --gpus 2` on a box with two devices: two ranks over RCCL, 131,072 gathered returns, a
  per-GPU rate within 5 % ... of the one-GPU line's (weak scaling: independent shards, one gather)."""
  _need_gpu()
  if torch.cuda.device_count() < 2:
    pytest.skip("one GPU visible")
  one = _run_bench(["--gpus", "1", "--steps", "4", "--warmup", "2", "--no-cpu-baseline"])
  two = _run_bench(["--gpus", "2", "--steps", "4", "--warmup", "2", "--no-cpu-baseline"])
  assert two["n_gpus"] == 2 and two["gathered_returns"] == 131072
  assert two["rccl_ranks"] == 2 and two["preflight"]["backend"].startswith("rccl") and two["preflight"]["devices"] == ["cuda:0", "cuda:1"]
  assert len(two["per_rank_ms_per_step"]) == 2 and max(two["per_rank_ms_per_step"]) <= two["ms_per_step"] * 1.01
  assert two["value"] / 2 > 0.9 * one["value"]
