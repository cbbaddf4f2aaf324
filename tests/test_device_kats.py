"""Known answers of the reference's own device tests, asked of the oracle's
device formulas -- the very functions sbo_step is composed of (oracle/sb_oracle.c: sbo_dev_*)."""
import datetime as dt

import pytest

from oracle import oracle as orc
from sbsim_amd import host_inputs

C_AIR, C_WATER = 1006.0, 4180.0   # utils/constants.py:21-24


def _params(**over):
  base = dict(dt=300.0, conv_threshold=0.1, iter_limit=100, vav_max_air_flow=0.6, vav_max_water_flow=0.4,
              ahu_recirc=0.3, ahu_heat_sp=270.0, ahu_cool_sp=288.0, ahu_dp=20000.0, ahu_eff=0.8,
              blr_setpoint=360.0, blr_head=3.0, blr_pump_eff=0.6, comfort_lo=292.0, comfort_hi=295.0,
              eco_lo=290.0, eco_hi=297.0)
  base.update(over)
  return orc.OracleParams(**base)


def test_boiler_thermal_energy_rate_kats():
  """boiler_test.py:131-146 (default boiler: 360 K, dissipation only, then 370 K) and :148-176."""
  p = _params()
  assert orc.device("boiler_gas_rate", 360.0, 0.0, 300.0, 280.0, 0.0, 0.0, params=p) == pytest.approx(500.066862, abs=1e-4)
  assert orc.device("boiler_gas_rate", 370.0, 0.0, 300.0, 280.0, 0.0, 300.0, params=p) == pytest.approx(562.57521, abs=1e-4)
  for setpoint, ret, outside, flow, want in ((340.0, 300.0, 280.0, 0.6, 100695.0501), (300.0, 300.0, 280.0, 0.6, 125.0167),
                                             (300.0, 300.0, 280.0, 0.01, 125.0167), (300.0, 300.0, 300.0, 0.01, 0.0)):
    got = orc.device("boiler_gas_rate", setpoint, flow, ret, outside, 0.0, 0.0, params=_params(blr_setpoint=setpoint))
    assert got == pytest.approx(want, abs=1e-3)


def test_boiler_dissipation_and_pump_kats():
  """boiler_test.py:431-439 (312.5418 W at 340 K / 290 K; zero at equal temperatures) and the
  pump formula of boiler.py:322-333 as boiler_test.py:399-416 states it."""
  p = _params()
  assert orc.boiler_dissipation(p, 340.0, 290.0) == pytest.approx(312.5418, abs=1e-4)
  assert orc.boiler_dissipation(p, 290.0, 290.0) == pytest.approx(0.0, abs=1e-4)
  assert orc.device("boiler_pump_power", 0.6, params=p) == 0.6 * 1000.0 * 9.8 * 3.0 / 0.6


def test_air_handler_kats():
  """air_handler_test.py:110-175: mixed air and supply air temperature (setpoints 270 / 288)."""
  for r, recirc, amb in ((0.3, 280, 240), (0.6, 244, 270), (0.1, 210, 316), (0.4, 250, 316), (0.4, 286, 266), (0.12, 198, 290)):
    assert orc.device("ahu_mixed", r, recirc, amb) == r * recirc + (1 - r) * amb
  for r, recirc, amb, want in ((0.3, 280, 240, 270), (0.6, 244, 270, 270), (0.1, 210, 316, 288), (0.4, 250, 316, 288),
                               (0.4, 286, 266, 0.4 * 286 + 0.6 * 266), (0.12, 198, 290, 0.12 * 198 + 0.88 * 290)):
    assert orc.device("ahu_supply", 270.0, 288.0, orc.device("ahu_mixed", r, recirc, amb)) == want
  # :335-371 compute_thermal_energy_rate; :373-420 fan powers (intake: all the air, exhaust: the fresh share)
  for flow, amb, recirc in ((100, 250, 210), (0.5, 280, 320), (1000, 155, 134), (2, 246, 290), (900, 50, 270)):
    mixed = orc.device("ahu_mixed", 0.3, recirc, amb)
    supply = orc.device("ahu_supply", 270.0, 288.0, mixed)
    assert orc.device("ahu_thermal_rate", flow, supply, mixed) == flow * C_AIR * (supply - mixed)
  p = _params()
  assert orc.device("ahu_blower_power", 5.0, params=p) == 5.0 * 20000.0 / 0.8 + (5.0 * (1.0 - 0.3)) * 20000.0 / 0.8


def test_vav_kats():
  """vav_test.py:198-262: zone supply temperature and the energy applied to the zone."""
  for valve, max_w, damper, max_a, t_sa, t_w in ((0.5, 0.8, 0.3, 0.3, 270, 360), (0.1, 0.1, 0.4, 0.4, 210, 32),
                                                 (0, 0.2, 0.2, 0.9, 260, 270), (0.9, 0.4, 0.1, 0.6, 270, 430)):
    reheat, air = valve * max_w, damper * max_a
    want = (t_sa * (C_AIR * air - C_WATER * reheat) + t_w * C_WATER * reheat) / air / C_AIR   # vav_test.py:55-75
    assert orc.device("vav_supply_temp", t_sa, t_w, air, reheat) == want
    assert orc.device("vav_energy", air, want, 293.0) == air * C_AIR * (want - 293.0)


@pytest.mark.parametrize("ts,zone_temp,damper,valve", [
    ("2021-05-09 14:00", 293, 0.1, 0.0), ("2021-05-10 09:00", 296, 1.0, 0.0), ("2021-05-12 09:00", 291, 1.0, 1.0),
    ("2021-05-12 17:59", 291, 1.0, 1.0), ("2021-05-11 03:00", 288, 1.0, 1.0), ("2021-05-11 03:00", 291, 0.1, 0.0),
    ("2021-05-11 22:00", 298, 1.0, 0.0), ("2021-05-11 22:00", 297, 0.1, 0.0)])
def test_vav_update_settings_table(ts, zone_temp, damper, valve):
  """vav_test.py:150-174: thermostat 9-18 h, comfort (292, 295), eco (290, 297), previous timestamp
  one hour earlier; the host schedule decides the mode flags, the oracle the control."""
  now = dt.datetime.fromisoformat(ts)
  sched = host_inputs.SetpointSchedule(9, 18, (292, 295), (290, 297), holidays=set())
  window = sched.get_temperature_window(now)
  _, got_damper, got_valve = orc.thermostat(0, float(zone_temp), window, sched.is_comfort_mode(now),
                                            int(sched.is_comfort_mode(now - dt.timedelta(minutes=60))))
  assert (got_damper, got_valve) == (damper, valve)


@pytest.mark.parametrize("day_of_year,seconds,expected", [
    (4, 0, 40.5), (4, 12 * 3600, 62.5), (4, 6 * 3600, 51.5), (110, 0, 30.0), (110, 12 * 3600, 70.0),
    (110, 6 * 3600, 50.0), (109, 18 * 3600, 46.25), (110, 18 * 3600, 55.25)])
def test_weather_controller_kats(day_of_year, seconds, expected):
  """weather_controller_test.py:97-121 (special day 110 with its own low / high): the host mirror."""
  w = host_inputs.WeatherController(40.5, 62.5, special_days={110: (30, 70)})
  ts = dt.datetime(2021, 1, 1) + dt.timedelta(days=day_of_year - 1, seconds=seconds)
  assert w.get_current_temp(ts) == expected


def test_replay_weather_controller_kats():
  """weather_controller_test.py:138-176 on the reference's own test data (Time / TempF columns of
  simulator/local_weather_test_data.csv, `YYYYMMDD-HHMM` time stamps): 298.15 K one second after
  03:00 UTC, ValueError outside the trace."""
  import os
  path = os.path.join(os.path.dirname(__file__), "golden", "local_weather_test_data.csv")
  w = host_inputs.ReplayWeatherController(path, 10.0)
  utc = dt.timezone.utc
  assert w.get_current_temp(dt.datetime(2023, 7, 1, 3, 0, 1, tzinfo=utc)) == pytest.approx(298.15, abs=1e-5)
  assert w.get_air_convection_coefficient(dt.datetime(2023, 7, 1, 3, tzinfo=utc)) == 10.0
  with pytest.raises(ValueError, match="before the latest"):
    w.get_current_temp(dt.datetime(2023, 6, 30, 23, 59, tzinfo=utc))
  with pytest.raises(ValueError, match="after the latest"):
    w.get_current_temp(dt.datetime(2023, 7, 1, 10, 0, 1, tzinfo=utc))
  b = host_inputs.BatchedReplayWeather(path, [0.0, 3600.0], 10.0)
  t = b.temps(dt.datetime(2023, 7, 1, 2, 30, tzinfo=utc))
  assert t[0] == w.get_current_temp(dt.datetime(2023, 7, 1, 2, 30, tzinfo=utc))
  assert t[1] == w.get_current_temp(dt.datetime(2023, 7, 1, 3, 30, tzinfo=utc)) == pytest.approx(298.15, abs=1e-9)


def test_boiler_adjust_temperature_kats():
  """boiler_test.py:197-230: (setpoint, actual, seconds, heating K/min, cooling K/min) -> tank."""
  for sp, actual, secs, h, c, want in ((330.0, 290.0, 60, 0.0, 0.0, 290.0), (330.0, 290.0, 60, 2.0, 0.0, 292.0),
                                       (300.0, 290.0, 600, 2.0, 0.0, 300.0), (320.0, 330.0, 60, 0.0, 0.5, 329.5),
                                       (320.0, 330.0, 600, 0.0, 2.0, 320.0)):
    assert orc.device("boiler_adjust", sp, actual, float(secs), h, c) == pytest.approx(want, abs=1e-7)


def test_thermostat_state_machine_kats():
  """thermostat_test.py:47-135 with its schedule (9-18 h, comfort (292, 295), eco (290, 297)).
  Modes: 0 OFF, 1 HEAT, 2 COOL, 3 PASSIVE_COOL."""
  sched = host_inputs.SetpointSchedule(9, 18, (292, 295), (290, 297), holidays=set())
  weekday, weekend = dt.datetime(2021, 5, 5, 11), dt.datetime(2021, 5, 8, 11)
  low, high = sched.get_temperature_window(weekday)
  mid = (low + high) / 2
  assert sched.is_comfort_mode(weekday) and not sched.is_comfort_mode(weekend)
  mode, seq = 0, []
  for tz in (low - 1, mid - 1, high + 1, mid + 1, mid - 1, mid + 1):      # test_update_comfort_mode / _default_control
    mode, _, _ = orc.thermostat(mode, tz, (low, high), True, 1)
    seq.append(mode)
  assert seq == [1, 1, 2, 2, 0, 0]
  eco = sched.get_temperature_window(weekend)
  m, _, _ = orc.thermostat(0, 0.0, (low, high), True, -1)                  # test_eco_transition
  m, damper, valve = orc.thermostat(m, 0.0, eco, False, 1)
  assert m == 3 and (damper, valve) == (0.1, 0.0)
  m, _, _ = orc.thermostat(m, (eco[0] + eco[1]) / 2, eco, False, 0)        # test_eco_mode: stays passive above the
  assert m == 3                                                            # heating setpoint, heats below it
  m, _, _ = orc.thermostat(m, 0.0, eco, False, 0)
  assert m == 1
