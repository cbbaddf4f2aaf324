"""Generates the golden fixtures under tests/golden by RUNNING THE REFERENCE ITSELF.

TEST INFRASTRUCTURE -- runs only in the build container, where ``/root/reference``
exists (``python -m oracle.gen_golden``).  The reference is imported through
``oracle.refshim`` (stand-ins for missing third-party modules only; every line of
simulator / HVAC / reward arithmetic executed here is the reference's own code), driven
through fixed scenarios, and its inputs/outputs are written as small ``.npz`` / ``.json``
files.  The fixtures are data (inputs and expected outputs); no reference source text
is stored.  ``tests/test_oracle_golden.py`` pins ``oracle/sb_oracle.c`` to them, and
``tests/test_host_golden.py`` pins the product's host-side calendar / weather /
occupancy / tariff / floor-plan code to them.

Harness levels:
  H1  bare ``Simulator``:  step_sim() [+ reward_info()]
  H2  ``SimulatorBuilding``: request_action -> wait_time -> request_observations ->
      reward_info -> compute_reward, i.e. Environment._step minus tf-agents.
"""
from __future__ import annotations

import json
import os
import sys
import time

import numpy as np
import pandas as pd

from oracle import refshim

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def _mods():
  refshim.install()
  m = {}
  for name in ("simulator.building", "simulator.simulator", "simulator.simulator_flexible_floor_plan",
               "simulator.hvac_floorplan_based", "simulator.hvac", "simulator.boiler",
               "simulator.air_handler", "simulator.setpoint_schedule", "simulator.weather_controller",
               "simulator.step_function_occupancy", "simulator.simulator_building",
               "simulator.vav", "simulator.thermostat",
               "reward.setpoint_energy_carbon_regret", "reward.electricity_energy_cost",
               "reward.natural_gas_energy_cost", "utils.conversion_utils",
               "simulator.simulator_flexible_floor_plan_test", "simulator.simulator_test",
               "simulator.building_test", "proto.smart_control_building_pb2"):
    m[name.split(".")[-1]] = refshim.ref(name)
  return m


# --------------------------------------------------------------------------- dumps
def dump_building(b, floor_plan=None) -> dict:
  """Post-preprocessing arrays of a FloorPlanBasedBuilding (building.py:717-764)."""
  H, W = b.temp.shape
  zone_names = [z for z in b._room_dict.keys() if z.startswith("room")]
  label = np.full((H, W), -1, dtype=np.int16)
  for zi, z in enumerate(zone_names):
    cells = b._room_dict[z]
    # room_dict lists are in raster order (building_utils.py:406-414); assert it so
    # that the label map alone reproduces them.
    flat = [x * W + y for x, y in cells]
    assert flat == sorted(flat)
    for x, y in cells:
      label[x, y] = zi
  out = dict(
      H=H, W=W, cv_size_cm=float(b.cv_size_cm), floor_height_cm=float(b.floor_height_cm),
      conductivity=np.asarray(b.conductivity, dtype=np.float64),
      heat_capacity=np.asarray(b.heat_capacity, dtype=np.float64),
      density=np.asarray(b.density, dtype=np.float64),
      exterior_space=(np.asarray(b._exterior_space) == -1),
      exterior_walls=(np.asarray(b._exterior_walls) == -2),
      interior_walls=(np.asarray(b._interior_walls) == -3),
      len_neighbors=np.asarray(b.len_neighbors, dtype=np.int8),
      diffusers=np.asarray(b.diffusers, dtype=np.float64),
      zone_label=label, zone_names=np.array(zone_names),
  )
  if floor_plan is not None:
    out["floor_plan"] = np.asarray(floor_plan, dtype=np.int8)
  return out


class SweepCounter:
  """Counts calls of Simulator.update_temperature_estimates (one per sweep)."""

  def __init__(self, sim):
    self.n = 0
    orig = sim.update_temperature_estimates

    def wrapped(*a, **k):
      self.n += 1
      return orig(*a, **k)

    sim.update_temperature_estimates = wrapped

  def take(self) -> int:
    n, self.n = self.n, 0
    return n


def tariff_tables(m) -> dict:
  elec = m["electricity_energy_cost"].ElectricityEnergyCost()
  gas = m["natural_gas_energy_cost"].NaturalGasEnergyCost()
  return dict(
      weekday_price=np.asarray(elec._weekday_energy_prices.magnitude, dtype=np.float64),
      weekend_price=np.asarray(elec._weekend_energy_prices.magnitude, dtype=np.float64),
      carbon_rate=np.asarray(elec._carbon_emission_rates, dtype=np.float64),
      gas_price=np.asarray(gas._month_gas_price, dtype=np.float64),
      gas_carbon=float(gas._carbon_rate),
  ), elec, gas


# --------------------------------------------------------------------------- scenarios
SB1 = dict(  # configs/resources/sb1/sim_config.gin:23-225 (values are configuration data)
    cv_size_cm=10.0, floor_height_cm=300.0, initial_temp=294.0,
    air=(50.0, 700.0, 1.0),        # conductivity, heat_capacity, density
    wall=(50.0, 1.0, 700.0),       # gin swaps heat_capacity/density for interior walls
    ext=(0.05, 700.0, 1.0),
    schedule=dict(morning_start_hour=6, evening_start_hour=19,
                  comfort_temp_window=(294, 297), eco_temp_window=(289, 298)),
    boiler=dict(reheat_water_setpoint=360.0, water_pump_differential_head=6.0,
                water_pump_efficiency=0.98, heating_rate=0.5, cooling_rate=0.1),
    ahu=dict(recirculation=0.3, heating_air_temp_setpoint=285.0, cooling_air_temp_setpoint=298.0,
             fan_differential_pressure=10000.0, fan_efficiency=0.9, max_air_flow_rate=8.67),
    vav_max_air_flow_rate=0.035, vav_reheat_max_water_flow_rate=0.03,
    time_step_sec=300, convergence_threshold=0.1, iteration_limit=100, iteration_warning=30,
    weather=dict(default_low_temp=273.0, default_high_temp=283.0, convection_coefficient=100.0),
    reward=dict(max_productivity_personhour_usd=300.0, min_productivity_personhour_usd=100.0,
                max_electricity_rate=160000, max_natural_gas_rate=400000,
                productivity_midpoint_delta=0.5, productivity_decay_stiffness=4.3,
                productivity_weight=0.2, energy_cost_weight=0.4, carbon_emission_weight=0.4),
    action_ranges=dict(supply_water_setpoint=(310.0, 355.0),
                       supply_air_heating_temperature_setpoint=(285.0, 300.0)),
    default_actions=dict(supply_water_setpoint=340.0,
                         supply_air_heating_temperature_setpoint=285.0),
    occupancy=dict(work_start_h=9, work_end_h=17, work_occupancy=10.0, nonwork_occupancy=0.1),
)


def build_sb1(m, floor_plan, start, with_weather_on_ahu=True, initial_temp=None):
  bp = m["building"]
  mk = lambda t: bp.MaterialProperties(conductivity=t[0], heat_capacity=t[1], density=t[2])
  building = bp.FloorPlanBasedBuilding(
      cv_size_cm=SB1["cv_size_cm"], floor_height_cm=SB1["floor_height_cm"],
      initial_temp=SB1["initial_temp"] if initial_temp is None else initial_temp,
      inside_air_properties=mk(SB1["air"]), inside_wall_properties=mk(SB1["wall"]),
      building_exterior_properties=mk(SB1["ext"]), floor_plan=floor_plan,
      zone_map=floor_plan.copy())
  weather = m["weather_controller"].WeatherController(**SB1["weather"])
  boiler = m["boiler"].Boiler(device_id="boiler_id", **SB1["boiler"])
  ahu = m["air_handler"].AirHandler(device_id="air_handler_id",
                                    sim_weather_controller=weather if with_weather_on_ahu else None,
                                    **SB1["ahu"])
  schedule = m["setpoint_schedule"].SetpointSchedule(**SB1["schedule"])
  hvac = m["hvac_floorplan_based"].FloorPlanBasedHvac(
      air_handler=ahu, boiler=boiler, schedule=schedule,
      vav_max_air_flow_rate=SB1["vav_max_air_flow_rate"],
      vav_reheat_max_water_flow_rate=SB1["vav_reheat_max_water_flow_rate"])
  sim = m["simulator_flexible_floor_plan"].SimulatorFlexibleGeometries(
      building, hvac, weather, SB1["time_step_sec"], SB1["convergence_threshold"],
      SB1["iteration_limit"], SB1["iteration_warning"], start)
  return sim, building, hvac, weather, schedule


def obs_request(m, sb):
  """environment.py:543-553: devices sorted by id, fields sorted by name."""
  pb = m["smart_control_building_pb2"]
  req = pb.ObservationRequest()
  names = []
  for dev in sorted(sb.devices, key=lambda d: d.device_id):
    for field in sorted(dev.observable_fields):
      req.single_observation_requests.add(device_id=dev.device_id, measurement_name=field)
      names.append(f"{dev.device_id}/{field}")
  return req, names


def rollout_h2(m, sim, building, hvac, weather, schedule, n_steps, actions_norm, grid_steps,
               label, extra_actions=(), rejected_steps=(), prev_ts=None):
  """Environment._step ordering without tf-agents (environment.py:1228-1309).

  extra_actions: more columns of the action vector after the SB1 pair, as (device_id,
  setpoint_name, (lo, hi)) -- actions_norm then has 2 + len(extra_actions) columns.
  rejected_steps: steps at which the building rejects the whole request (a RuntimeError out of
  request_action, rejection_simulator_building.py:52-60): Environment catches it, skips nothing
  else, and returns reward -inf (environment.py:1266-1309).
  prev_ts: Thermostat._previous_timestamp as a previous episode on the same simulator left it
  (it survives Simulator.reset(), thermostat.py:66-69); only the recorded comfort_prev uses it."""
  pb = m["smart_control_building_pb2"]
  cu = m["conversion_utils"]
  occ_cfg = SB1["occupancy"]
  occupancy = m["step_function_occupancy"].StepFunctionOccupancy(
      pd.Timedelta(occ_cfg["work_start_h"], unit="h"), pd.Timedelta(occ_cfg["work_end_h"], unit="h"),
      occ_cfg["work_occupancy"], occ_cfg["nonwork_occupancy"])
  sb = m["simulator_building"].SimulatorBuilding(sim, occupancy)
  tables, elec, gas = tariff_tables(m)
  reward_fn = m["setpoint_energy_carbon_regret"].SetpointEnergyCarbonRegretFunction(
      electricity_energy_cost=elec, natural_gas_energy_cost=gas, **SB1["reward"])
  counter = SweepCounter(sim)
  req, obs_names = obs_request(m, sb)
  zones = list(hvac.vavs.keys())
  Z = len(zones)
  dt = pd.Timedelta(sim.time_step_sec, unit="s")
  rng_w = SB1["action_ranges"]["supply_water_setpoint"]
  rng_a = SB1["action_ranges"]["supply_air_heating_temperature_setpoint"]

  sb.reset()
  rec = {k: [] for k in (
      "action_native", "t_amb_now", "t_amb_next", "comfort_now", "comfort_prev", "comfort_next",
      "occupancy", "hour_utc", "month", "is_workday", "e_price", "e_carbon", "g_price", "g_carbon",
      "zone_temp_pre", "zone_temp_post", "n_sweeps", "recirc_pre", "t_supply_air", "zone_q_sum",
      "ahu_flow", "ahu_count", "blr_flow", "blr_count", "blr_return_temp", "blr_tank_temp",
      "damper", "valve", "mode", "rates", "ri_zone_temp", "ri_heat_sp", "ri_cool_sp", "ri_occ",
      "reward", "rr_productivity", "rr_norm_prod_regret", "rr_norm_energy_cost",
      "rr_norm_carbon", "rr_elec_cost", "rr_gas_cost", "rr_carbon", "obs", "ts_seconds",
      "num_occupants", "comfort_soon", "rejected", "action_accepted", "action_native_all")}
  grids = {}
  # first observation at reset (environment.py:1165-1176)
  first = sb.request_observations(req)
  obs0 = np.array([r.continuous_value for r in first.single_observation_responses], np.float32)
  t0 = time.time()
  for step in range(n_steps):
    ts = sim.current_timestamp
    a = actions_norm[step].astype(np.float32)
    # bounded_action_normalizer.py:73-98 evaluated in float64, then the proto float field
    native = [np.float32((float(a[0]) + 1.0) / 2.0 * (rng_w[1] - rng_w[0]) + rng_w[0]),
              np.float32((float(a[1]) + 1.0) / 2.0 * (rng_a[1] - rng_a[0]) + rng_a[0])]
    areq = pb.ActionRequest(timestamp=cu.pandas_to_proto_timestamp(ts))
    areq.single_action_requests.append(pb.SingleActionRequest(
        device_id="boiler_id", setpoint_name="supply_water_setpoint",
        continuous_value=float(native[0])))
    areq.single_action_requests.append(pb.SingleActionRequest(
        device_id="air_handler_id", setpoint_name="supply_air_heating_temperature_setpoint",
        continuous_value=float(native[1])))
    native_all = [float(native[0]), float(native[1])]
    for j, (dev_id, sp_name, (x_lo, x_hi)) in enumerate(extra_actions):
      v = np.float32((float(a[2 + j]) + 1.0) / 2.0 * (x_hi - x_lo) + x_lo)
      native_all.append(float(v))
      areq.single_action_requests.append(pb.SingleActionRequest(
          device_id=dev_id, setpoint_name=sp_name, continuous_value=float(v)))
    if step in rejected_steps:   # RejectionSimulatorBuilding.request_action raises before delegating
      accepted = False
    else:
      resp = sb.request_action(areq)
      accepted = all(r.response_type == 1 for r in resp.single_action_responses)   # environment.py:all_actions_accepted
      assert accepted or extra_actions
    rec["rejected"].append(int(step in rejected_steps))
    rec["action_accepted"].append(int(accepted))
    rec["action_native_all"].append(native_all)
    recirc_pre = building.temp.mean()
    t_amb_now = weather.get_current_temp(ts)
    t_sa = hvac.air_handler.get_supply_air_temp(recirc_pre, t_amb_now)
    sb.wait_time()
    n_sw = counter.take()
    new_ts = sim.current_timestamp
    obs = sb.request_observations(req)
    ri = sb.reward_info
    rr = reward_fn.compute_reward(ri)

    rec["action_native"].append([float(native[0]), float(native[1])])
    rec["ts_seconds"].append(int(ts.timestamp()))
    rec["t_amb_now"].append(t_amb_now)
    rec["t_amb_next"].append(weather.get_current_temp(new_ts))
    rec["comfort_now"].append(schedule.is_comfort_mode(ts))
    rec["comfort_prev"].append(-1 if prev_ts is None else int(schedule.is_comfort_mode(prev_ts)))
    rec["comfort_next"].append(schedule.is_comfort_mode(new_ts))
    rec["comfort_soon"].append(schedule.is_comfort_mode(new_ts + pd.Timedelta(60, unit="minute")))
    rec["num_occupants"].append(sb.num_occupants)
    rec["occupancy"].append(occupancy.average_zone_occupancy("z", new_ts, new_ts + dt))
    start_utc = cu.proto_to_pandas_timestamp(ri.start_timestamp)
    workday = cu.is_work_day(start_utc)
    rec["hour_utc"].append(start_utc.hour)
    rec["month"].append(start_utc.month)
    rec["is_workday"].append(workday)
    rec["e_price"].append((tables["weekday_price"] if workday else tables["weekend_price"])[start_utc.hour])
    rec["e_carbon"].append(tables["carbon_rate"][start_utc.hour])
    rec["g_price"].append(tables["gas_price"][start_utc.month - 1])
    rec["g_carbon"].append(tables["gas_carbon"])
    rec["zone_temp_pre"].append([hvac.vavs[z].zone_air_temperature for z in zones])
    post = building.get_zone_average_temps()
    rec["zone_temp_post"].append([post[z] for z in zones])
    rec["n_sweeps"].append(n_sw)
    rec["recirc_pre"].append(recirc_pre)
    rec["t_supply_air"].append(t_sa)
    rec["zone_q_sum"].append([building.get_zone_thermal_energy_rate(z) for z in zones])
    rec["ahu_flow"].append(hvac.air_handler.air_flow_rate)
    rec["ahu_count"].append(hvac.air_handler.cooling_request_count)
    rec["blr_flow"].append(hvac.boiler._total_flow_rate)
    rec["blr_count"].append(hvac.boiler.heating_request_count)
    rec["blr_return_temp"].append(hvac.boiler.return_water_temperature_sensor)
    rec["blr_tank_temp"].append(hvac.boiler._current_temperature)
    rec["damper"].append([hvac.vavs[z].damper_setting for z in zones])
    rec["valve"].append([hvac.vavs[z].reheat_valve_setting for z in zones])
    rec["mode"].append([hvac.vavs[z].thermostat._current_mode.value for z in zones])
    ah = ri.air_handler_reward_infos["air_handler_id"]
    bl = ri.boiler_reward_infos["boiler_id"]
    rec["rates"].append([ah.blower_electrical_energy_rate,
                         ah.air_conditioning_electrical_energy_rate,
                         bl.natural_gas_heating_energy_rate, bl.pump_electrical_energy_rate])
    zid = [cu.floor_plan_based_zone_identifier_to_id(z) for z in zones]
    rec["ri_zone_temp"].append([ri.zone_reward_infos[i].zone_air_temperature for i in zid])
    rec["ri_heat_sp"].append([ri.zone_reward_infos[i].heating_setpoint_temperature for i in zid])
    rec["ri_cool_sp"].append([ri.zone_reward_infos[i].cooling_setpoint_temperature for i in zid])
    rec["ri_occ"].append([ri.zone_reward_infos[i].average_occupancy for i in zid])
    rec["reward"].append(rr.agent_reward_value)
    rec["rr_productivity"].append(rr.productivity_reward)
    rec["rr_norm_prod_regret"].append(rr.normalized_productivity_regret)
    rec["rr_norm_energy_cost"].append(rr.normalized_energy_cost)
    rec["rr_norm_carbon"].append(rr.normalized_carbon_emission)
    rec["rr_elec_cost"].append(rr.electricity_energy_cost)
    rec["rr_gas_cost"].append(rr.natural_gas_energy_cost)
    rec["rr_carbon"].append(rr.carbon_emitted)
    rec["obs"].append([r.continuous_value for r in obs.single_observation_responses])
    if (step + 1) in grid_steps:
      grids[f"grid_{step + 1}"] = building.temp.copy()
    prev_ts = ts
  print(f"  {label}: {n_steps} steps in {time.time() - t0:.1f}s, "
        f"sweeps total {sum(rec['n_sweeps'])}, max {max(rec['n_sweeps'])}")
  out = {}
  f32 = {"rates", "ri_zone_temp", "ri_heat_sp", "ri_cool_sp", "ri_occ", "reward", "obs",
         "rr_productivity", "rr_norm_prod_regret", "rr_norm_energy_cost", "rr_norm_carbon",
         "rr_elec_cost", "rr_gas_cost", "rr_carbon", "action_native", "action_native_all"}
  ints = {"n_sweeps", "ahu_count", "blr_count", "mode", "comfort_now", "comfort_prev",
          "comfort_next", "comfort_soon", "hour_utc", "month", "is_workday", "ts_seconds",
          "num_occupants", "rejected", "action_accepted"}
  if not extra_actions and not rejected_steps:   # the original fixtures keep their key set
    for k in ("rejected", "action_accepted", "action_native_all"):
      rec.pop(k)
  for k, v in rec.items():
    if k in f32:
      out[k] = np.asarray(v, dtype=np.float32)
    elif k in ints:
      out[k] = np.asarray(v, dtype=np.int64)
    else:
      out[k] = np.asarray(v, dtype=np.float64)
  out.update(grids)
  out["obs_names"] = np.array(obs_names)
  out["obs_reset"] = obs0
  out["zone_names"] = np.array(zones)
  out["actions_norm"] = np.asarray(actions_norm[:n_steps], dtype=np.float32)
  out["final_grid"] = building.temp.copy()
  out["final_input_q"] = building.input_q.copy()
  return out


def rollout_h1(m, sim, building, hvac, n_steps, label, grid_steps=()):
  """Bare Simulator.step_sim() loop (simulator.py:578-594)."""
  counter = SweepCounter(sim)
  zones = list(hvac.vavs.keys())
  rec = {k: [] for k in ("zone_temp_post", "n_sweeps", "blr_return_temp", "ahu_flow", "blr_flow",
                         "damper", "valve", "mode", "zone_temp_pre")}
  grids = {}
  init = building.temp.copy()
  for step in range(n_steps):
    sim.step_sim()
    rec["n_sweeps"].append(counter.take())
    post = building.get_zone_average_temps()
    rec["zone_temp_post"].append([post[z] for z in zones])
    rec["zone_temp_pre"].append([hvac.vavs[z].zone_air_temperature for z in zones])
    rec["blr_return_temp"].append(hvac.boiler.return_water_temperature_sensor)
    rec["ahu_flow"].append(hvac.air_handler.air_flow_rate)
    rec["blr_flow"].append(hvac.boiler._total_flow_rate)
    rec["damper"].append([hvac.vavs[z].damper_setting for z in zones])
    rec["valve"].append([hvac.vavs[z].reheat_valve_setting for z in zones])
    rec["mode"].append([hvac.vavs[z].thermostat._current_mode.value for z in zones])
    if (step + 1) in grid_steps:
      grids[f"grid_{step + 1}"] = building.temp.copy()
  print(f"  {label}: sweeps {rec['n_sweeps']}")
  out = {k: np.asarray(v, dtype=np.int64 if k in ("n_sweeps", "mode") else np.float64)
         for k, v in rec.items()}
  out.update(grids)
  out["initial_grid"] = init
  out["final_grid"] = building.temp.copy()
  out["final_input_q"] = building.input_q.copy()
  out["zone_names"] = np.array(zones)
  return out


def hvac_params_of(hvac, sim, schedule=None) -> dict:
  """Scalar configuration of an Hvac + Simulator, read back from the live objects."""
  b, a = hvac.boiler, hvac.air_handler
  sch = schedule if schedule is not None else next(iter(hvac.vavs.values())).thermostat.get_setpoint_schedule()
  v = next(iter(hvac.vavs.values()))
  return dict(
      dt=float(sim.time_step_sec), conv_threshold=float(sim._convergence_threshold),
      iter_limit=int(sim._iteration_limit),
      vav_max_air_flow=float(v.max_air_flow_rate),
      vav_max_water_flow=float(v._reheat_max_water_flow_rate),
      ahu_recirc=float(a.recirculation), ahu_heat_sp=float(a._init_heating_air_temp_setpoint),
      ahu_cool_sp=float(a._init_cooling_air_temp_setpoint),
      ahu_dp=float(a.fan_differential_pressure), ahu_eff=float(a.fan_efficiency),
      ahu_max_flow=float(a.max_air_flow_rate),
      ahu_has_weather=int(a._sim_weather_controller is not None),
      blr_setpoint=float(b._init_reheat_water_setpoint),
      blr_head=float(b._init_water_pump_differential_head),
      blr_pump_eff=float(b._init_water_pump_efficiency),
      blr_heating_rate=float(b._heating_rate), blr_cooling_rate=float(b._cooling_rate),
      blr_conv=float(b._convection_coefficient), blr_len=float(b._tank_length),
      blr_radius=float(b._tank_radius), blr_capacity=float(b._water_capacity),
      blr_ins_k=float(b._insulation_conductivity), blr_ins_thick=float(b._insulation_thickness),
      comfort_lo=float(sch.comfort_temp_window[0]), comfort_hi=float(sch.comfort_temp_window[1]),
      eco_lo=float(sch.eco_temp_window[0]), eco_hi=float(sch.eco_temp_window[1]),
      morning_start_hour=int(sch.morning_start_hour), evening_start_hour=int(sch.evening_start_hour),
      holidays=sorted(int(h) for h in sch.holidays),
  )


def main() -> None:
  if not refshim.available():
    print("reference tree not present; nothing to do")
    return
  os.makedirs(OUT, exist_ok=True)
  m = _mods()
  ft = m["simulator_flexible_floor_plan_test"].FlexibleFloorplanSimulatorTest()
  meta = {"generated_by": "oracle/gen_golden.py", "numpy": np.__version__,
          "pandas": pd.__version__, "scenarios": {}}

  # ---- G0: floor-plan preprocessing on the reference tests' own plans -----------------
  print("G0 floor plans")
  plans = {}
  r9_plan = np.asarray(ft._create_scenario_floor_plan())
  small_plan = np.asarray(ft._create_dummy_floor_plan_small())
  weird_plan = np.asarray(ft._create_dummy_floor_plan_weird_shape())
  b_r9 = ft._create_scenario_building(initial_temp=200.0, match_old_diffusers=True)
  b_r9_native = ft._create_scenario_building(initial_temp=200.0, match_old_diffusers=False)
  b_small = ft._create_small_building(initial_temp=293.0)
  b_weird = ft._create_weirdly_shaped_building(initial_temp=293.0)
  for name, b, fp in (("r9_test", b_r9, r9_plan), ("r9_test_native_diffusers", b_r9_native, r9_plan),
                      ("small_test", b_small, small_plan), ("weird_test", b_weird, weird_plan)):
    d = dump_building(b, fp)
    plans[name] = d
    np.savez_compressed(os.path.join(OUT, f"plan_{name}.npz"), **d)
    print(f"  {name}: {d['H']}x{d['W']} zones={len(d['zone_names'])} "
          f"diffusers={(d['diffusers'] > 0).sum()}")

  # ---- G1/G2: single sweep and FD time step on seeded random fields ------------------
  print("G1/G2 sweeps")
  rs = np.random.RandomState(20240117)
  wc = m["weather_controller"]
  for name, b in (("r9_test", b_r9), ("small_test", b_small), ("weird_test", b_weird)):
    hv = ft._create_scenario_hvac(zone_identifier=list(b._room_dict.keys()))
    sim = m["simulator_flexible_floor_plan"].SimulatorFlexibleGeometries(
        b, hv, wc.WeatherController(280.0, 280.0), 300.0, 0.1, 100, 1000,
        pd.Timestamp("2012-12-21"))
    H, W = b.temp.shape
    prev = 285.0 + 10.0 * rs.rand(H, W)
    est = prev + 0.5 * (rs.rand(H, W) - 0.5)
    q = np.where(b.diffusers > 0, 2000.0 * (rs.rand(H, W) - 0.3), 0.0)
    b.temp = prev.copy()
    b.input_q = q.copy()
    est1 = est.copy()
    _, md1 = sim.update_temperature_estimates(est1, ambient_temperature=277.5,
                                              convection_coefficient=12.0)
    est2 = est1.copy()
    _, md2 = sim.update_temperature_estimates(est2, ambient_temperature=277.5,
                                              convection_coefficient=12.0)
    # G2: whole time step from `prev`
    b.temp = prev.copy()
    counter = SweepCounter(sim)
    conv = sim.finite_differences_timestep(ambient_temperature=277.5, convection_coefficient=12.0)
    np.savez_compressed(
        os.path.join(OUT, f"sweep_{name}.npz"), prev=prev, est=est, q=q, est_after_1=est1,
        max_delta_1=md1, est_after_2=est2, max_delta_2=md2, t_amb=277.5, h=12.0, dt=300.0,
        fd_grid=b.temp.copy(), fd_sweeps=counter.take(), fd_converged=conv, thr=0.1,
        iter_limit=100)
    print(f"  {name}: max_delta {md1:.6f} {md2:.6f}, fd sweeps -> converged={conv}")

  # ---- G3/G7: reference KAT (301.895482) on both simulator classes + H1 rollouts ------
  print("G3 KAT + H1 rollouts (reference test scenario)")
  for init_temp, n_steps, tag in ((200.0, 6, "cold200"), (292.0, 12, "mild292")):
    b = ft._create_scenario_building(initial_temp=init_temp, match_old_diffusers=True)
    hv = ft._create_scenario_hvac(zone_identifier=list(b._room_dict.keys()))
    sim = m["simulator_flexible_floor_plan"].SimulatorFlexibleGeometries(
        b, hv, wc.WeatherController(296.0, 296.0), 300.0, 0.1, 100, 10, pd.Timestamp("12-21-2012"))
    out = rollout_h1(m, sim, b, hv, n_steps, f"r9_test {tag}", grid_steps=(1,))
    out["params_json"] = np.array(json.dumps(hvac_params_of(hv, sim)))
    out["t_amb"] = 296.0
    out["h_conv"] = 12.0
    out["initial_temp"] = init_temp
    # calendar inputs for the thermostat: Friday 2012-12-21 00:00.. (naive -> UTC)
    ts = [pd.Timestamp("12-21-2012") + pd.Timedelta(300 * i, unit="s") for i in range(n_steps + 1)]
    sch = next(iter(hv.vavs.values())).thermostat.get_setpoint_schedule()
    out["comfort"] = np.array([sch.is_comfort_mode(t) for t in ts], dtype=np.int64)
    np.savez_compressed(os.path.join(OUT, f"h1_r9_test_{tag}.npz"), **out)
    if tag == "cold200":
      meta["kat_return_water_temp_after_1_step"] = float(out["blr_return_temp"][0])
      print("  KAT return water temp:", out["blr_return_temp"][0], "(reference test expects 301.895482)")

  # rectangular (deprecated) Building on the same scenario: simulator_test.py:955-987
  st = m["simulator_test"].SimulatorTest()
  b_old = st._create_scenario_building(initial_temp=200.0)
  hv_old = st._create_scenario_hvac()
  sim_old = m["simulator"].Simulator(b_old, hv_old, wc.WeatherController(296.0, 296.0), 300.0,
                                     0.1, 100, 10, pd.Timestamp("12-21-2012"))
  sim_old.step_sim()
  meta["kat_return_water_temp_rectangular_building"] = float(
      hv_old.boiler.return_water_temperature_sensor)
  np.savez_compressed(
      os.path.join(OUT, "rect_building_cold200.npz"),
      conductivity=b_old.conductivity, heat_capacity=b_old.heat_capacity, density=b_old.density,
      diffusers=np.asarray(b_old.diffusers, dtype=np.float64), final_grid=b_old.temp,
      room_shape=np.array(b_old.room_shape), building_shape=np.array(b_old.building_shape),
      blr_return_temp=hv_old.boiler.return_water_temperature_sensor,
      cv_size_cm=b_old.cv_size_cm, floor_height_cm=b_old.floor_height_cm)
  print("  rectangular Building KAT:", hv_old.boiler.return_water_temperature_sensor)

  # small + weird plans, H1, test materials (exercises n<=1 / n==2 / n==3 cells everywhere)
  for name, mk in (("small_test", lambda: ft._create_small_building(initial_temp=285.0)),
                   ("weird_test", lambda: ft._create_weirdly_shaped_building(initial_temp=285.0))):
    b = mk()
    hv = ft._create_scenario_hvac(zone_identifier=list(b._room_dict.keys()))
    sim = m["simulator_flexible_floor_plan"].SimulatorFlexibleGeometries(
        b, hv, wc.WeatherController(275.0, 275.0), 300.0, 0.01, 100, 1000,
        pd.Timestamp("2012-12-21 10:00"))
    out = rollout_h1(m, sim, b, hv, 8, f"{name} h1", grid_steps=(1,))
    out["params_json"] = np.array(json.dumps(hvac_params_of(hv, sim)))
    out["t_amb"] = 275.0
    out["h_conv"] = 12.0
    out["initial_temp"] = 285.0
    ts = [pd.Timestamp("2012-12-21 10:00") + pd.Timedelta(300 * i, unit="s") for i in range(9)]
    sch = next(iter(hv.vavs.values())).thermostat.get_setpoint_schedule()
    out["comfort"] = np.array([sch.is_comfort_mode(t) for t in ts], dtype=np.int64)
    np.savez_compressed(os.path.join(OUT, f"h1_{name}.npz"), **out)

  # ---- G4/G5/G6: SB1-physics R9, one simulated day, H2 harness ------------------------
  print("G4/G5 H2 rollouts (SB1 physics on R9)")
  start = pd.Timestamp("2023-07-06 07:00:00")
  n_day = 288
  da = SB1["default_actions"]
  rw, ra = SB1["action_ranges"]["supply_water_setpoint"], SB1["action_ranges"]["supply_air_heating_temperature_setpoint"]
  const = np.tile(np.array([[(da["supply_water_setpoint"] - rw[0]) / (rw[1] - rw[0]) * 2 - 1,
                             (da["supply_air_heating_temperature_setpoint"] - ra[0]) / (ra[1] - ra[0]) * 2 - 1]],
                           dtype=np.float32), (n_day, 1))
  rnd = np.random.RandomState(1234).uniform(-1.0, 1.0, size=(n_day, 2)).astype(np.float32)
  sb1_plan_dumped = False
  for tag, actions, n_steps, grid_steps in (("const", const, n_day, (1, 144)),
                                            ("random", rnd, n_day, (1, 144))):
    sim, building, hvac, weather, schedule = build_sb1(m, r9_plan, start)
    if not sb1_plan_dumped:
      np.savez_compressed(os.path.join(OUT, "plan_r9_sb1.npz"), **dump_building(building, r9_plan))
      sb1_plan_dumped = True
    out = rollout_h2(m, sim, building, hvac, weather, schedule, n_steps, actions, grid_steps,
                     f"sb1_r9 {tag}")
    prm = hvac_params_of(hvac, sim, schedule)
    prm.update({k: float(v) for k, v in dict(
        max_prod=SB1["reward"]["max_productivity_personhour_usd"],
        min_prod=SB1["reward"]["min_productivity_personhour_usd"],
        max_elec=SB1["reward"]["max_electricity_rate"], max_gas=SB1["reward"]["max_natural_gas_rate"],
        prod_delta=SB1["reward"]["productivity_midpoint_delta"],
        prod_stiff=SB1["reward"]["productivity_decay_stiffness"],
        w_prod=SB1["reward"]["productivity_weight"], w_cost=SB1["reward"]["energy_cost_weight"],
        w_carbon=SB1["reward"]["carbon_emission_weight"]).items()})
    out["params_json"] = np.array(json.dumps(prm))
    out["h_conv"] = SB1["weather"]["convection_coefficient"]
    out["initial_temp"] = SB1["initial_temp"]
    out["start_timestamp"] = np.array(str(start))
    np.savez_compressed(os.path.join(OUT, f"h2_sb1_r9_{tag}.npz"), **out)

  tables, _, _ = tariff_tables(m)
  np.savez_compressed(os.path.join(OUT, "tariffs.npz"), **tables)

  # ---- host-side generators: weather / schedule / occupancy traces over odd times -----
  print("host traces")
  weather = wc.WeatherController(273.0, 283.0, special_days={188: (270.0, 290.0), 189: (265.0, 275.0)})
  tss = [pd.Timestamp("2023-07-06 00:00:00") + pd.Timedelta(777 * i, unit="s") for i in range(400)]
  sch = m["setpoint_schedule"].SetpointSchedule(6, 19, (294, 297), (289, 298), holidays={188})
  occ = m["step_function_occupancy"].StepFunctionOccupancy(
      pd.Timedelta(9, unit="h"), pd.Timedelta(17, unit="h"), 10.0, 0.1)
  cu = m["conversion_utils"]
  np.savez_compressed(
      os.path.join(OUT, "host_traces.npz"),
      ts_seconds=np.array([int(t.timestamp()) for t in tss], dtype=np.int64),
      weather=np.array([weather.get_current_temp(t) for t in tss]),
      comfort=np.array([sch.is_comfort_mode(t) for t in tss], dtype=np.int64),
      occupancy=np.array([occ.average_zone_occupancy("z", t, t + pd.Timedelta(300, unit="s")) for t in tss]),
      occupancy_777=np.array([occ.average_zone_occupancy("z", t, t + pd.Timedelta(777, unit="s")) for t in tss]),
      is_workday=np.array([cu.is_work_day(t) for t in tss], dtype=np.int64),
      hod_rad=np.array([cu.get_radian_time(t, cu.TimeIntervalEnum.HOUR_OF_DAY) for t in tss]),
      dow_rad=np.array([cu.get_radian_time(t, cu.TimeIntervalEnum.DAY_OF_WEEK) for t in tss]),
      special_days=np.array([[188, 270.0, 290.0], [189, 265.0, 275.0]]),
      schedule_holidays=np.array([188]))

  meta["scenarios"]["SB1"] = {k: v for k, v in SB1.items()}
  with open(os.path.join(OUT, "meta.json"), "w") as fh:
    json.dump(meta, fh, indent=1, default=str)
  print("done ->", OUT)


if __name__ == "__main__":
  sys.exit(main())
