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


@pytest.mark.gpu
def test_hip_simulator_building_replays_the_reference_rollout():
  """``HipSimulatorBuilding`` -- BaseBuilding's request / response interface
  (simulator_building.py:151-315) on the HIP library -- driven exactly like the harness that made
  tests/golden/h2_sb1_r9_random.npz drove the reference's SimulatorBuilding: request_action ->
  wait_time -> request_observations -> reward_info, one simulated day."""
  torch = pytest.importorskip("torch")
  if not torch.cuda.is_available():
    pytest.skip("no GPU")
  import datetime as dt
  from sbsim_amd import building_adapter as ba
  from tests.test_gpu_parity import _plan
  g = load("h2_sb1_r9_random.npz")
  start = dt.datetime.fromisoformat(str(g["start_timestamp"]))
  b = ba.HipSimulatorBuilding(_plan(load("plan_r9_sb1.npz")), start_timestamp=start, holiday_calendar=None)
  assert b.time_step_sec == 300.0 and b.current_timestamp == start and len(b.zones) == 9 and len(b.devices) == 11
  req = b.observation_request_for_all_fields()
  names = [f"{r.device_id}/{r.measurement_name}" for r in req.single_observation_requests]
  assert [n.split("/")[1] for n in names] == [str(n).split("/")[-1] for n in g["obs_names"]]
  first = b.request_observations(req)
  assert np.allclose([r.continuous_value for r in first.single_observation_responses], g["obs_reset"], rtol=1e-6, atol=1e-6)
  T = len(g["n_sweeps"])
  for t in range(T):
    assert int(b.current_timestamp.replace(tzinfo=dt.timezone.utc).timestamp()) == int(g["ts_seconds"][t])
    areq = ba.ActionRequest(timestamp=b.current_timestamp, single_action_requests=[
        ba.SingleActionRequest("boiler_id", "supply_water_setpoint", float(g["action_native"][t][0])),
        ba.SingleActionRequest("air_handler_id", "supply_air_heating_temperature_setpoint", float(g["action_native"][t][1]))])
    resp = b.request_action(areq)
    assert all(r.response_type == ba.ActionResponseType.ACCEPTED for r in resp.single_action_responses)
    b.wait_time()
    obs = b.request_observations(req)
    vals = np.array([r.continuous_value for r in obs.single_observation_responses])
    assert all(r.observation_valid for r in obs.single_observation_responses)
    assert np.allclose(vals, g["obs"][t], rtol=1e-6, atol=1e-6), t
    ri = b.reward_info
    zt = np.array([z.zone_air_temperature for z in ri.zone_reward_infos.values()])
    assert np.abs(zt - g["ri_zone_temp"][t]).max() <= 4e-5, t   # fp32 fields of temperatures within 1e-8 K
    assert np.allclose([z.heating_setpoint_temperature for z in ri.zone_reward_infos.values()], g["ri_heat_sp"][t])
    assert np.allclose([z.cooling_setpoint_temperature for z in ri.zone_reward_infos.values()], g["ri_cool_sp"][t])
    assert np.allclose([z.average_occupancy for z in ri.zone_reward_infos.values()], g["ri_occ"][t])
    ah, bl = ri.air_handler_reward_infos["air_handler_id"], ri.boiler_reward_infos["boiler_id"]
    rates = [ah.blower_electrical_energy_rate, ah.air_conditioning_electrical_energy_rate,
             bl.natural_gas_heating_energy_rate, bl.pump_electrical_energy_rate]
    assert np.allclose(rates, g["rates"][t].astype(np.float64), rtol=2e-6, atol=1e-6), t
    assert abs(b.last_reward - float(g["reward"][t])) < 1e-6, t
    assert b.num_occupants == int(g["num_occupants"][t]), t
  # an unknown device, a field that is not settable, a damper command the handle does not drive
  resp = b.request_action(ba.ActionRequest(single_action_requests=[
      ba.SingleActionRequest("no_such_device", "supply_water_setpoint", 340.0),
      ba.SingleActionRequest("boiler_id", "supply_water_temperature_sensor", 340.0),
      ba.SingleActionRequest(b.devices[2].device_id, "supply_air_damper_percentage_command", 0.5)]))
  assert [r.response_type for r in resp.single_action_responses] == [
      ba.ActionResponseType.REJECTED_INVALID_DEVICE, ba.ActionResponseType.REJECTED_NOT_ENABLED_OR_AVAILABLE,
      ba.ActionResponseType.REJECTED_NOT_ENABLED_OR_AVAILABLE]
  bad = b.request_observations(ba.ObservationRequest(single_observation_requests=[ba.SingleObservationRequest("boiler_id", "nope")]))
  assert not bad.single_observation_responses[0].observation_valid
  b.close()


@pytest.mark.gpu
def test_hip_simulator_building_rejections_and_damper_commands():
  """The same interface on the four-column action set of h2_sb1_r9_actions.npz: a step whose request
  the building rejected is ``wait_time`` without ``request_action``; a damper command outside [0, 1]
  comes back REJECTED_NOT_ENABLED_OR_AVAILABLE while the other setpoints apply."""
  torch = pytest.importorskip("torch")
  if not torch.cuda.is_available():
    pytest.skip("no GPU")
  import datetime as dt
  from sbsim_amd import building_adapter as ba
  from sbsim_amd.environment import SimConfig
  from tests.test_gpu_parity import T_TOL, _plan
  g = load("h2_sb1_r9_actions.npz")
  cfg = SimConfig.sb1()
  cfg.action_names = tuple(str(n) for n in g["action_names"])
  cfg.action_zones = tuple(int(z) for z in g["action_zone"])
  cfg.action_ranges = tuple((float(lo), float(hi)) for lo, hi in g["action_ranges"])
  b = ba.HipSimulatorBuilding(_plan(load("plan_r9_sb1.npz")), config=cfg, holiday_calendar=None,
                              start_timestamp=dt.datetime.fromisoformat(str(g["start_timestamp"])))
  vav = b.devices[2 + int(g["action_zone"][3])].device_id
  for t in range(len(g["n_sweeps"])):
    nat = [float(x) for x in g["action_native_all"][t]]
    if not g["rejected"][t]:
      resp = b.request_action(ba.ActionRequest(timestamp=b.current_timestamp, single_action_requests=[
          ba.SingleActionRequest("boiler_id", "supply_water_setpoint", nat[0]),
          ba.SingleActionRequest("air_handler_id", "supply_air_heating_temperature_setpoint", nat[1]),
          ba.SingleActionRequest("air_handler_id", "supply_air_cooling_temperature_setpoint", nat[2]),
          ba.SingleActionRequest(vav, "supply_air_damper_percentage_command", nat[3])]))
      ok = all(r.response_type == ba.ActionResponseType.ACCEPTED for r in resp.single_action_responses)
      assert ok == bool(g["action_accepted"][t]), t
      assert [r.response_type for r in resp.single_action_responses[:3]] == [ba.ActionResponseType.ACCEPTED] * 3
    b.wait_time()
    assert int(b._info_row[4]) == int(g["n_sweeps"][t]), t
    zt = b.env.sim.zone_temps()[0].cpu().numpy()
    assert np.abs(zt - g["zone_temp_post"][t]).max() < T_TOL, t
    assert abs(b.last_reward - float(g["reward"][t])) < 1e-6, t
  # a damper command of -inf: the host says REJECTED and the device must reject the step too (reward -inf,
  # environment.py:1301-1302) -- the raw value would read as "field not mentioned" there (<= SB_ACTION_KEEP)
  resp = b.request_action(ba.ActionRequest(timestamp=b.current_timestamp, single_action_requests=[
      ba.SingleActionRequest("boiler_id", "supply_water_setpoint", 340.0),
      ba.SingleActionRequest(vav, "supply_air_damper_percentage_command", float("-inf"))]))
  assert resp.single_action_responses[1].response_type == ba.ActionResponseType.REJECTED_NOT_ENABLED_OR_AVAILABLE
  b.wait_time()
  assert float(b.env._reward[0]) == float("-inf")
  b.close()


@pytest.mark.gpu
def test_hip_simulator_building_on_a_four_wavefront_plan(monkeypatch):
  """The single-building adapter on a 205 x 89 / 40-zone plan (step_band.hip, four wavefronts per building; one
  building = one workgroup): request_action -> wait_time -> request_observations -> reward_info, against a second
  adapter on the LDS-grid kernel (SBSIM_NO_BAND_PATH=1) given the same requests: sweep counts EQUAL, observations and
  rewards equal to float32 rounding, grids within 1e-9 K."""
  torch = pytest.importorskip("torch")
  if not torch.cuda.is_available():
    pytest.skip("no GPU")
  from sbsim_amd import building_adapter as ba
  from sbsim_amd.environment import SimConfig
  from sbsim_amd.floorplan import FloorPlan, Materials, rectangular_floor_plan
  plan = FloorPlan.from_file_input(rectangular_floor_plan((10, 4), (19, 20)), Materials.sb1(), 10.0, 300.0)
  cfg = SimConfig.sb1()
  b = ba.HipSimulatorBuilding(plan, cfg, holiday_calendar=None)
  monkeypatch.setenv("SBSIM_NO_BAND_PATH", "1")
  ref = ba.HipSimulatorBuilding(plan, cfg, holiday_calendar=None)
  monkeypatch.delenv("SBSIM_NO_BAND_PATH")
  assert b.env.sim.launch_info["kernel"] == 5 and b.env.sim.launch_info["waves_per_building"] == 4
  assert ref.env.sim.launch_info["kernel"] == 0 and len(b.zones) == 40
  rs = np.random.RandomState(5)
  for t in range(6):
    sw, sa = float(rs.uniform(310.0, 350.0)), float(rs.uniform(285.0, 295.0))
    out = []
    for bld in (b, ref):
      resp = bld.request_action(ba.ActionRequest(timestamp=bld.current_timestamp, single_action_requests=[
          ba.SingleActionRequest("boiler_id", "supply_water_setpoint", sw),
          ba.SingleActionRequest("air_handler_id", "supply_air_heating_temperature_setpoint", sa)]))
      assert all(r.response_type == ba.ActionResponseType.ACCEPTED for r in resp.single_action_responses)
      bld.wait_time()
      obs = bld.request_observations(bld.observation_request_for_all_fields())
      out.append((np.array([r.continuous_value for r in obs.single_observation_responses]), bld.last_reward, bld._info_row[4]))
    assert out[0][2] == out[1][2] >= 1                       # Gauss-Seidel sweeps of the step
    assert np.allclose(out[0][0], out[1][0], rtol=1e-6, atol=1e-6) and abs(out[0][1] - out[1][1]) < 1e-6
  assert np.abs(b.env.sim.temps().cpu().numpy() - ref.env.sim.temps().cpu().numpy()).max() < 1e-9
  b.close()
  ref.close()


@pytest.mark.gpu
def test_action_vector_as_wide_as_the_building_has_setpoints():
  """ABI 8: one action column per settable field (environment/environment.py:591-653 admits every device setpoint with a
  normaliser): the boiler's, the air handler's two and one damper command per VAV of a 126-zone plan -- a 129-column
  vector (the 16-column limit of ABI <= 7 did not hold an SB1 agent that drives its dampers, vav.py:66-70).  Against
  oracle twins (their `damper_cmd` path, sb_oracle.c: vav.py:125-129): some columns left out of a request
  (SB_ACTION_KEEP), some commands outside [0, 1] (that action is rejected, the others apply, the reward is -inf)."""
  torch = pytest.importorskip("torch")
  if not torch.cuda.is_available():
    pytest.skip("no GPU")
  from oracle import oracle as orc   # checker only
  from sbsim_amd import _ffi
  from sbsim_amd.environment import ACTION_REJECTION_REWARD, BatchedSimulator, SimConfig
  from sbsim_amd.floorplan import FloorPlan, Materials, rectangular_floor_plan
  from tests.test_gpu_parity import T_TOL, _step_in
  plan = FloorPlan.from_file_input(rectangular_floor_plan((14, 9), (8, 7)), Materials.sb1(), 10.0, 300.0)
  Z = len(plan.zone_cell_lists())
  assert Z == 126
  cfg = SimConfig.sb1()
  cfg.action_names = ("supply_water_setpoint", "supply_air_heating_temperature_setpoint",
                      "supply_air_cooling_temperature_setpoint") + ("supply_air_damper_percentage_command",) * Z
  # the dampers are listed in reverse zone order: a column is not its zone's index
  cfg.action_zones = (0, 0, 0) + tuple(range(Z - 1, -1, -1))
  cfg.action_ranges = ((310.0, 355.0), (285.0, 295.0), (296.0, 305.0)) + ((-0.001, 1.001),) * Z   # a normaliser that can overshoot [0, 1] (one damper in 500)
  g = load("h2_sb1_r9_random.npz")
  B, T = 6, 8
  sim = BatchedSimulator(plan, cfg, B, float(g["h_conv"]))
  assert sim.n_actions == 3 + Z
  rs = np.random.RandomState(5)
  H, W = plan.shape
  init = np.clip(294.0 + rs.randn(B, 1) + np.zeros((B, H * W)), 285.0, 305.0)
  sim.reset(temps=torch.tensor(init, dtype=torch.float64, device="cuda"))
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
  twins = [orc.OracleBuilding(oplan, oprm, 0.0, reset_temps=init[b]) for b in range(B)]
  obs = torch.zeros((B, sim.O), dtype=torch.float32, device="cuda")
  rew = torch.zeros((B,), dtype=torch.float32, device="cuda")
  info = torch.zeros((B, _ffi.SB_INFO_STRIDE), dtype=torch.float32, device="cuda")
  n_rejected = 0
  for t in range(T):
    norm = rs.uniform(-1, 1, size=(B, 3 + Z)).astype(np.float32)
    native = np.empty((B, 3 + Z), dtype=np.float32)
    for i, (lo, hi) in enumerate(cfg.action_ranges):   # bounded_action_normalizer.py:93-98, then the proto's float
      native[:, i] = ((norm[:, i].astype(np.float64) + 1.0) / 2.0 * (hi - lo) + lo).astype(np.float32)
    keep = rs.rand(B, 3 + Z) < 0.3                      # the request does not mention these fields
    keep[:, :2] = False                                 # (the twins' interface always carries the two SB1 setpoints)
    if t % 2 == 0:                                      # every other step in native units with left-out columns
      act = native.copy()
      act[keep] = _ffi.SB_ACTION_KEEP
    else:
      act, keep = norm, np.zeros_like(keep)
    si = _step_in(g, t + 100)
    si.actions_native = int(t % 2 == 0)
    sim.step(torch.tensor(act, device="cuda"), si, obs, rew, info)
    i_ = info.cpu().numpy().astype(np.float64)
    zt, r = sim.zone_temps().cpu().numpy(), rew.cpu().numpy()
    dm = sim._get(sim._lib.sb_get_zone_power, (B, Z), torch.float64).cpu().numpy()
    for b in range(B):
      cmd = np.full(Z, np.nan)
      for col in range(3, 3 + Z):
        if not keep[b, col]:
          cmd[cfg.action_zones[col]] = float(native[b, col])
      tt = t + 100
      o = twins[b].step(
          now_ts=300.0 * t, t_amb_now=float(g["t_amb_now"][tt]), h_conv=float(g["h_conv"]),
          t_amb_next=float(g["t_amb_next"][tt]), comfort_now=bool(g["comfort_now"][tt]),
          comfort_prev=g["comfort_prev"][tt] == 1, comfort_next=bool(g["comfort_next"][tt]),
          occupancy=float(g["occupancy"][tt]), e_price=float(g["e_price"][tt]), e_carbon=float(g["e_carbon"][tt]),
          g_price=float(g["g_price"][tt]), g_carbon=float(g["g_carbon"][tt]),
          action=[native[b, 0], native[b, 1]], cool_sp=None if keep[b, 2] else float(native[b, 2]), damper_cmd=cmd)
      assert i_[b, 4] == o["n_sweeps"], (t, b, i_[b, 4], o["n_sweeps"])
      assert np.abs(zt[b] - o["zone_temp_post"]).max() < T_TOL, (t, b)
      assert np.allclose(dm[b], o["q_zone"], rtol=1e-12, atol=1e-9), (t, b)     # every zone's VAV power: every damper landed
      if o["action_accepted"]:
        assert abs(float(r[b]) - o["reward"]) < 1e-6, (t, b)
      else:
        n_rejected += 1
        assert r[b] == ACTION_REJECTION_REWARD and abs(i_[b, 7] - o["reward"]) < 1e-6, (t, b)
  assert 0 < n_rejected < B * T          # the overshooting normaliser rejected some steps, not all
  sim.close()
