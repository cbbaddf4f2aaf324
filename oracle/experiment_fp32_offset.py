"""Precision experiment (TEST INFRASTRUCTURE, CPU only): does float32 state flip stopping decisions?

SURVEY.md section 7, hard part 1, option (b): keep every control-volume temperature as a float32
OFFSET from a reference temperature and do the arithmetic in float64.  That would halve the HBM
bytes and the registers of the sweep kernel.  The Gauss-Seidel loop stops when max|delta| <= 0.1 K
(simulator.py:362); a rounding error of a few 1e-6 K can flip that decision, and one sweep more or
less moves temperatures by up to ~0.1 K.  This script measures how often that happens on the bench
workload (BASELINE.json configs[1]: R9 buildings, random setpoint actions, 288 steps = one day):
the same buildings are stepped twice by the CPU oracle, once with float64 state (the reference's
arithmetic, bit for bit) and once with float32-offset state (sbo_set_state_mode), and compared.

  python -m oracle.experiment_fp32_offset --buildings 2048 --steps 288 --out profiles/r03_fp32_offset_flips.json
"""
from __future__ import annotations

import argparse
import datetime as dt
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)

from oracle import oracle as orc  # noqa: E402
from sbsim_amd import host_inputs  # noqa: E402
from sbsim_amd.environment import SimConfig  # noqa: E402
from sbsim_amd.floorplan import FloorPlan, Materials, rectangular_floor_plan  # noqa: E402


def main() -> None:
  ap = argparse.ArgumentParser()
  ap.add_argument("--buildings", type=int, default=2048)
  ap.add_argument("--steps", type=int, default=288)
  ap.add_argument("--t-ref", type=float, default=294.0, help="reference temperature of the float32 offsets (K)")
  ap.add_argument("--threads", type=int, default=0)
  ap.add_argument("--out", default="")
  args = ap.parse_args()

  plan = FloorPlan.from_file_input(rectangular_floor_plan((3, 3), (20, 30)), Materials.sb1(), 10.0, 300.0)
  c = SimConfig.sb1()
  oplan = orc.OraclePlan(plan.conductivity, plan.density, plan.heat_capacity, plan.exterior_space,
                         plan.zone_cell_lists(), plan.diffusers, plan.cv_size_cm, plan.floor_height_cm)
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
  weather = host_inputs.WeatherController(273.0, 283.0, convection_coefficient=100.0)
  occupancy = host_inputs.StepFunctionOccupancy(dt.timedelta(hours=9), dt.timedelta(hours=17), 10.0, 0.1,
                                                holiday_calendar="us")
  schedule = c.schedule()
  elec = host_inputs.ElectricityEnergyCost(holiday_calendar="us")
  gas = host_inputs.NaturalGasEnergyCost()
  start = host_inputs.as_datetime(dt.datetime(2023, 7, 6, 7, 0, 0))
  step = dt.timedelta(seconds=c.time_step_sec)

  nb, H, W = args.buildings, *plan.shape
  rs = np.random.RandomState(7)                       # bench.py: per-building initial temperature
  t_init = np.clip(294.0 + rs.randn(nb), 285.0, 305.0)
  init = np.broadcast_to(t_init[:, None], (nb, H * W)).copy()
  ra = np.random.RandomState(1234)                    # random setpoint actions U[-1, 1]^2 per building and step
  acts = ra.uniform(-1.0, 1.0, size=(args.steps, nb, 2)).astype(np.float32)
  lo, hi = c.action_ranges
  threads = args.threads or max(1, min(orc.lib().sbo_max_threads(), os.cpu_count() or 1))
  lib = orc.lib()
  lib.sbo_set_state_mode.argtypes = [orc.C.c_int32, orc.C.c_double]
  lib.sbo_set_state_mode.restype = None

  zone_cells = [np.asarray(cl, dtype=np.int64) for cl in plan.zone_cell_lists()]
  zone_cells = [cl if cl.ndim == 1 else cl[:, 0] * W + cl[:, 1] for cl in zone_cells]

  def rollout(mode):
    lib.sbo_set_state_mode(mode, args.t_ref)
    batch = orc.OracleBatch(oplan, oprm, init if mode == 0 else args.t_ref + (init - args.t_ref).astype(np.float32).astype(np.float64))
    sweeps = np.zeros((args.steps, nb), dtype=np.int32)
    zt = np.zeros((args.steps, nb, oplan.Z))
    ts, prev = start, None
    t0 = time.perf_counter()
    for k in range(args.steps):
      nxt = ts + step
      comfort_prev = 0 if prev is None else int(schedule.is_comfort_mode(prev))
      start_utc = host_inputs.reward_start_time_utc(nxt)
      ep, ec = elec.rates(start_utc)
      gp, gc = gas.rates(start_utc)
      occ = occupancy.average_zone_occupancy("", nxt, nxt + step)
      ins = []
      for b in range(nb):
        a = acts[k, b]
        native = [np.float32((float(a[0]) + 1.0) / 2.0 * (lo[1] - lo[0]) + lo[0]),
                  np.float32((float(a[1]) + 1.0) / 2.0 * (hi[1] - hi[0]) + hi[0])]
        ins.append(orc.make_step_in(
            now_ts=300.0 * k, t_amb_now=weather.get_current_temp(ts), h_conv=weather.convection_coefficient,
            t_amb_next=weather.get_current_temp(nxt), comfort_now=int(schedule.is_comfort_mode(ts)),
            comfort_prev=comfort_prev, comfort_next=int(schedule.is_comfort_mode(nxt)),
            occupancy=np.full(oplan.Z, occ), observe=1, e_price=ep, e_carbon=ec, g_price=gp, g_carbon=gc,
            action=native))
      outs = batch.step(ins, n_threads=threads)
      for b in range(nb):
        sweeps[k, b] = outs[b].n_sweeps
      g = np.stack([ob.temp for ob in batch.buildings])
      zt[k] = np.stack([g[:, cells].mean(axis=1) for cells in zone_cells], axis=1)
      prev, ts = ts, nxt
    lib.sbo_set_state_mode(0, 0.0)
    grids = np.stack([b.grid() for b in batch.buildings])
    return sweeps, zt, grids, time.perf_counter() - t0

  s64, z64, g64, t64 = rollout(0)
  s32, z32, g32, t32 = rollout(1)
  differs = s64 != s32
  first = np.where(differs.any(axis=0), differs.argmax(axis=0), -1)
  dz = np.abs(z64 - z32).max(axis=2)                  # [steps, nb] largest zone-temperature difference
  res = {
      "what": "float32-offset state (T - t_ref stored as float32, float64 arithmetic) vs the reference's float64 state, "
              "CPU oracle, bench workload (R9, random setpoint actions)",
      "buildings": nb, "steps": args.steps, "t_ref_K": args.t_ref,
      "building_steps": int(nb * args.steps),
      "mean_sweeps_per_step_f64": float(s64.mean()),
      "building_steps_with_a_different_sweep_count": int(differs.sum()),
      "buildings_whose_sweep_count_ever_differs": int(differs.any(axis=0).sum()),
      "fraction_of_buildings_affected": float(differs.any(axis=0).mean()),
      "first_differing_step_median": (float(np.median(first[first >= 0])) if (first >= 0).any() else None),
      "max_abs_zone_temperature_difference_K": float(dz.max()),
      "max_abs_cell_temperature_difference_after_last_step_K": float(np.abs(g64 - g32).max()),
      "fraction_of_building_steps_with_zone_difference_above_1e-4_K": float((dz > 1e-4).mean()),
      "fraction_of_buildings_ever_above_1e-4_K": float((dz > 1e-4).any(axis=0).mean()),
      "zone_difference_where_sweep_counts_never_differed_max_K":
          (float(dz[:, ~differs.any(axis=0)].max()) if (~differs.any(axis=0)).any() else None),
      "contract_K": 1e-4,
      "seconds": {"f64": t64, "f32_offset": t32, "threads": threads},
  }
  print(json.dumps(res, indent=1))
  if args.out:
    with open(args.out, "w") as fh:
      json.dump(res, fh, indent=1)
      fh.write("\n")


if __name__ == "__main__":
  main()
