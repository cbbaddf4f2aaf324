"""Per-building replay weather: one ReplayWeatherController per building
over the same CSV, shifted in time, interpolated on the device."""
import datetime as dt

import numpy as np
import pytest

from sbsim_amd import host_inputs
from tests.golden_util import load

UTC = dt.timezone.utc


def _csv(tmp_path, hours=96):
  rs = np.random.RandomState(4)
  t0 = dt.datetime(2023, 7, 5, 0, 0, 0, tzinfo=UTC)
  temps = 60.0 + 15.0 * np.sin(np.arange(hours) * 2 * np.pi / 24.0 - 2.0) + rs.randn(hours) * 2.0
  path = tmp_path / "weather.csv"
  with open(path, "w") as fh:
    fh.write("Time,TempF\n")
    for i in range(hours):
      stamp = (t0 + dt.timedelta(hours=i, minutes=53 if i % 7 == 3 else 0)).strftime("%Y%m%d-%H%M")
      fh.write(f"{(t0 + dt.timedelta(hours=i, minutes=53 if i % 7 == 3 else 0)).isoformat()},{temps[i]:.2f}\n")
  return str(path)


def test_batched_replay_weather_is_one_controller_per_building(tmp_path):
  path = _csv(tmp_path)
  offsets = np.array([0.0, 3600.0, 5400.0, 7.5 * 3600.0, 86400.0 - 1.0, 12345.678])
  w = host_inputs.BatchedReplayWeather(path, offsets, convection_coefficient=12.0)
  one = host_inputs.ReplayWeatherController(path, 12.0)
  for minutes in (0, 5, 55, 60, 61, 600, 1439, 1440 + 53):
    ts = dt.datetime(2023, 7, 6, 0, 0, 0, tzinfo=UTC) + dt.timedelta(minutes=minutes)
    want = [one.get_current_temp(ts + dt.timedelta(seconds=float(o))) for o in offsets]
    assert np.array_equal(w.temps(ts), np.array(want))
  with pytest.raises(ValueError, match="after the latest"):
    w.temps(dt.datetime(2023, 7, 8, 12, 0, 0, tzinfo=UTC))
  with pytest.raises(ValueError, match="before the latest"):
    w.temps(dt.datetime(2023, 7, 4, 12, 0, 0, tzinfo=UTC))


@pytest.mark.gpu
def test_device_replay_weather_through_the_environment_api(tmp_path):
  """Every building against an oracle twin fed by its own ReplayWeatherController; the exterior
  ring of the final grid is the device's own interpolation, compared bit for bit."""
  import torch
  if not torch.cuda.is_available():
    pytest.skip("needs an MI355X")
  from oracle import oracle as orc
  from sbsim_amd.environment import BatchedEnvironment, SimConfig
  from tests.golden_util import oracle_params, oracle_plan
  from tests.test_gpu_parity import T_TOL, _plan
  path = _csv(tmp_path)
  p = load("plan_r9_sb1.npz")
  g = load("h2_sb1_r9_random.npz")
  offsets = np.array([0.0, 1800.0, 3600.0 * 5 + 17.0, 3600.0 * 11, 86400.0, 40000.25])
  B, T = len(offsets), 14
  weather = host_inputs.BatchedReplayWeather(path, offsets, convection_coefficient=float(g["h_conv"]))
  one = host_inputs.ReplayWeatherController(path, float(g["h_conv"]))
  start = dt.datetime(2023, 7, 6, 9, 30, 0, tzinfo=UTC)
  env = BatchedEnvironment(_plan(p), B, config=SimConfig.sb1(), weather=weather, start_timestamp=start,
                           holiday_calendar=None, collect_info=True)
  ts0 = env.reset()
  oat = env.field_names.index("air_handler_id/outside_air_temperature_sensor")
  assert np.allclose(ts0.observation.cpu().numpy()[:, oat], weather.temps(start).astype(np.float32))
  plan, prm = oracle_plan(p), oracle_params(g["params_json"])
  twins = [orc.OracleBuilding(plan, prm, float(env.config.initial_temp)) for _ in range(B)]
  rs = np.random.RandomState(9)
  acts = rs.uniform(-1, 1, size=(T, B, 2)).astype(np.float32)
  rng_w, rng_a = env.config.action_ranges
  shift = [dt.timedelta(seconds=float(o)) for o in offsets]
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
          now_ts=300.0 * t, t_amb_now=one.get_current_temp(now + shift[b]), h_conv=float(g["h_conv"]),
          t_amb_next=one.get_current_temp(nxt + shift[b]), comfort_now=bool(si.comfort_now),
          comfort_prev=si.comfort_prev == 1, comfort_next=bool(si.comfort_next), occupancy=si.occupancy,
          e_price=si.e_price, e_carbon=si.e_carbon, g_price=si.g_price, g_carbon=si.g_carbon,
          action=native, observe=True)
      assert int(info[b, 4]) == o["n_sweeps"], (t, b, info[b, 4], o["n_sweeps"])
      assert np.abs(zt[b] - o["zone_temp_post"]).max() < T_TOL, (t, b)
      assert abs(float(out.reward[b]) - o["reward"]) < 1e-6, (t, b)
    assert np.allclose(out.observation.cpu().numpy()[:, oat], weather.temps(nxt).astype(np.float32))
  grid = env.sim.temps().cpu().numpy()
  assert np.array_equal(grid[:, 0, 0], weather.temps(now))   # the device's own interpolation, bit for bit
  env.close()
