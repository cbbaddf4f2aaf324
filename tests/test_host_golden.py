"""Pins the product's host-side code (floor-plan preprocessing, weather, schedule,
occupancy, tariffs, time features) to values produced by the reference itself
(tests/golden, written by oracle/gen_golden.py).  CPU only."""
import datetime as dt
import json
import os

import numpy as np
import pytest

from sbsim_amd import host_inputs as hi
from sbsim_amd.floorplan import FloorPlan, Material, Materials, rectangular_floor_plan
from tests.golden_util import GOLDEN, load

UTC = dt.timezone.utc
TEST_SMALL = Materials(Material(50., 700., 1.), Material(2., 500., 1800.), Material(.05, 500., 3000.))
TEST_R9 = Materials(Material(50., 700., 1.), Material(5., 800., 1800.), Material(5., 800., 3000.))


@pytest.mark.parametrize("name,mat,cv,buf", [
    ("r9_test_native_diffusers", TEST_R9, 20.0, 0), ("small_test", TEST_SMALL, 20.0, 0),
    ("weird_test", TEST_SMALL, 20.0, 0), ("r9_sb1", Materials.sb1(), 10.0, 3)])
def test_floor_plan_preprocessing_matches_reference(name, mat, cv, buf):
  g = load(f"plan_{name}.npz")
  fp = FloorPlan.from_file_input(g["floor_plan"], mat, cv, 300.0, buffer_from_walls=buf)
  for mine, ref in (("conductivity", "conductivity"), ("heat_capacity", "heat_capacity"),
                    ("density", "density"), ("exterior_space", "exterior_space"),
                    ("zone_label", "zone_label"), ("diffusers", "diffusers")):
    assert np.array_equal(getattr(fp, mine), g[ref]), (name, mine)
  assert np.array_equal(fp.neighbor_masks()[1], g["len_neighbors"])
  assert list(fp.zone_names) == list(g["zone_names"])


def test_rectangular_tiler_reproduces_reference_test_plan():
  assert np.array_equal(rectangular_floor_plan((3, 3), (20, 30)), load("plan_r9_sb1.npz")["floor_plan"])


def _ts(sec):
  return dt.datetime.fromtimestamp(int(sec), tz=UTC).replace(tzinfo=None)


def test_weather_schedule_occupancy_time_features():
  g = load("host_traces.npz")
  special = {int(r[0]): (float(r[1]), float(r[2])) for r in g["special_days"]}
  weather = hi.WeatherController(273.0, 283.0, special_days=special)
  sched = hi.SetpointSchedule(6, 19, (294, 297), (289, 298), holidays=set(int(h) for h in g["schedule_holidays"]))
  occ = hi.StepFunctionOccupancy(dt.timedelta(hours=9), dt.timedelta(hours=17), 10.0, 0.1,
                                 holiday_calendar=None)   # the golden harness had no holiday package
  for i, sec in enumerate(g["ts_seconds"]):
    t = _ts(sec)
    assert weather.get_current_temp(t) == g["weather"][i], i
    assert sched.is_comfort_mode(t) == bool(g["comfort"][i]), i
    assert occ.average_zone_occupancy("z", t, t + dt.timedelta(seconds=300)) == g["occupancy"][i], i
    assert occ.average_zone_occupancy("z", t, t + dt.timedelta(seconds=777)) == g["occupancy_777"][i], i
    assert hi.is_work_day(t, None) == bool(g["is_workday"][i])
    assert hi.get_radian_time(t, True) == g["hod_rad"][i]
    assert hi.get_radian_time(t, False) == g["dow_rad"][i]


def test_tariff_tables():
  g = load("tariffs.npz")
  e, n = hi.ElectricityEnergyCost(), hi.NaturalGasEnergyCost()
  assert np.array_equal(e.weekday_energy_prices, g["weekday_price"])
  assert np.array_equal(e.weekend_energy_prices, g["weekend_price"])
  assert np.array_equal(e.carbon_emission_rates, g["carbon_rate"])
  assert np.array_equal(n.month_gas_price, g["gas_price"])
  assert n.carbon_rate == float(g["gas_carbon"])


@pytest.mark.parametrize("tag", ["const", "random"])
def test_step_inputs_of_the_one_day_rollout(tag):
  """The per-step host inputs the environment would hand the kernel, against the values
  the reference objects produced during the H2 golden rollout."""
  g = load(f"h2_sb1_r9_{tag}.npz")
  weather = hi.WeatherController(273.0, 283.0, convection_coefficient=100.0)
  sched = hi.SetpointSchedule(6, 19, (294, 297), (289, 298))
  occ = hi.StepFunctionOccupancy(dt.timedelta(hours=9), dt.timedelta(hours=17), 10.0, 0.1, None)
  e, n = hi.ElectricityEnergyCost(holiday_calendar=None), hi.NaturalGasEnergyCost()
  step = dt.timedelta(seconds=300)
  for i, sec in enumerate(g["ts_seconds"]):
    t = _ts(sec)
    nxt = t + step
    assert weather.get_current_temp(t) == g["t_amb_now"][i]
    assert weather.get_current_temp(nxt) == g["t_amb_next"][i]
    assert sched.is_comfort_mode(t) == bool(g["comfort_now"][i])
    assert sched.is_comfort_mode(nxt) == bool(g["comfort_next"][i])
    assert sched.is_comfort_mode(nxt + dt.timedelta(minutes=60)) == bool(g["comfort_soon"][i])
    assert occ.average_zone_occupancy("", nxt, nxt + step) == g["occupancy"][i]
    su = hi.reward_start_time_utc(nxt)
    assert (su.hour, su.month) == (int(g["hour_utc"][i]), int(g["month"][i]))
    assert e.rates(su) == (g["e_price"][i], g["e_carbon"][i])
    assert n.rates(su) == (g["g_price"][i], g["g_carbon"][i])
    n_occ = int(sum(occ.average_zone_occupancy("", nxt - dt.timedelta(minutes=5), nxt) for _ in range(9)))
    assert n_occ == int(g["num_occupants"][i])


def test_us_federal_holidays():
  h = hi.us_federal_holidays(2023)
  for d in [(1, 2), (1, 16), (2, 20), (5, 29), (6, 19), (7, 4), (9, 4), (10, 9), (11, 10), (11, 23), (12, 25)]:
    assert dt.date(2023, *d) in h, d
  assert not hi.is_work_day(dt.datetime(2023, 7, 4, 12))
  assert hi.is_work_day(dt.datetime(2023, 7, 6, 12))
  assert not hi.is_work_day(dt.datetime(2023, 7, 8, 12))
  assert dt.date(2021, 12, 31) in hi.us_federal_holidays(2021)   # New Year 2022 observed


def test_schedule_timezone_conversion():
  s = hi.SetpointSchedule(6, 19, (294, 297), (289, 298), time_zone="US/Pacific")
  aware = dt.datetime(2023, 7, 6, 13, 30, tzinfo=UTC)      # 06:30 PDT, Thursday
  assert s.is_comfort_mode(aware)
  assert not s.is_comfort_mode(dt.datetime(2023, 7, 6, 12, 30, tzinfo=UTC))  # 05:30 PDT
  assert not s.is_comfort_mode(dt.datetime(2023, 7, 6, 5, 0))                # naive = UTC clock
  with pytest.raises(ValueError):
    hi.SetpointSchedule(20, 19, (294, 297), (289, 298))


def test_batched_sinusoid_weather_equals_one_controller_per_building():
  """BatchedSinusoidWeather (per-building bounds, generated on the device from one host factor)
  gives bit-identical temperatures to B separate WeatherControllers
  (weather_controller.py:93-123), whose traces are pinned to the reference above."""
  import datetime as dt
  from sbsim_amd.host_inputs import BatchedSinusoidWeather, WeatherController
  rs = np.random.RandomState(2)
  low = 250.0 + 40.0 * rs.rand(17)
  high = low + 25.0 * rs.rand(17)
  w = BatchedSinusoidWeather(low, high, convection_coefficient=100.0)
  for k in range(0, 2 * 288, 7):
    ts = dt.datetime(2023, 7, 6, 0, 0, 0) + dt.timedelta(seconds=300 * k)
    want = np.array([WeatherController(lo, hi).get_current_temp(ts) for lo, hi in zip(low, high)])
    assert np.array_equal(w.temps(ts), want)
    f = w.factor(ts)
    assert np.array_equal(f * (high - low) + low, want)       # what the device evaluates
  with pytest.raises(ValueError):
    BatchedSinusoidWeather([300.0], [290.0])
