"""Host-side, per-step inputs of the batched step: weather, schedule, occupancy, tariffs.

These are the pieces SURVEY.md section 8(a) rows a14/a17/a18 marks "host-side; fed as a
per-step scalar": every building of a batch shares the simulator clock, so calendar logic
(time zones, holidays, schedules) runs once per step on the host and reaches the kernel as
a small struct (include/sbsim_amd.h ``sb_step_in``).  Each class mirrors the reference
class of the same name (same constructor arguments, same method names) and cites it.

Timestamps are ``datetime.datetime`` (``pandas.Timestamp`` is accepted: it is a subclass).
Naive timestamps are interpreted exactly as the reference does: as UTC wall-clock
(``setpoint_schedule.py:100-106``, ``conversion_utils.py:39-49``).
"""
from __future__ import annotations

import csv
import datetime as dt
import math
from typing import Dict, Iterable, Mapping, Optional, Sequence, Set, Tuple
from zoneinfo import ZoneInfo

import numpy as np

UTC = dt.timezone.utc
_SECONDS_IN_A_DAY = 24 * 3600
_DAYS_IN_A_YEAR = 365
_MIN_RADIANS = -math.pi / 2.0
_MAX_RADIANS = 3.0 * math.pi / 2.0


def as_datetime(ts) -> dt.datetime:
  if hasattr(ts, "to_pydatetime"):
    return ts.to_pydatetime()
  return ts


def _tz(time_zone) -> dt.tzinfo:
  if isinstance(time_zone, dt.tzinfo):
    return time_zone
  if time_zone in ("UTC", None):
    return UTC
  return ZoneInfo(str(time_zone))


def epoch_seconds(ts: dt.datetime) -> float:
  """pandas ``Timestamp.timestamp()``: naive values are taken as UTC."""
  ts = as_datetime(ts)
  if ts.tzinfo is None:
    ts = ts.replace(tzinfo=UTC)
  return ts.timestamp()


def seconds_since_midnight(ts: dt.datetime) -> float:
  return ts.hour * 3600 + ts.minute * 60 + ts.second + ts.microsecond * 1e-6


# --------------------------------------------------------------------------- weather
class WeatherController:
  """Sinusoid: low at midnight, high at noon (simulator/weather_controller.py:47-132)."""

  def __init__(self, default_low_temp: float, default_high_temp: float,
               special_days: Optional[Mapping[int, Tuple[float, float]]] = None,
               convection_coefficient: float = 12.0):
    if default_low_temp > default_high_temp:
      raise ValueError("default_low_temp cannot be greater than default_high_temp.")
    self.default_low_temp = default_low_temp
    self.default_high_temp = default_high_temp
    self.special_days = dict(special_days) if special_days else {}
    self.convection_coefficient = convection_coefficient
    for day, (low, high) in self.special_days.items():
      if low > high:
        raise ValueError(f"Low temp cannot be greater than high temp for special day: {day}.")

  def seconds_to_rads(self, seconds_in_day: float) -> float:
    return (seconds_in_day / _SECONDS_IN_A_DAY) * (_MAX_RADIANS - _MIN_RADIANS) + _MIN_RADIANS

  def get_current_temp(self, timestamp) -> float:
    ts = as_datetime(timestamp)
    today = ts.timetuple().tm_yday
    tomorrow = (today + 1) % _DAYS_IN_A_YEAR
    today_low, today_high = self.special_days.get(
        today, (self.default_low_temp, self.default_high_temp))
    tomorrow_low = self.special_days.get(tomorrow, (self.default_low_temp, None))[0]
    high = today_high
    low = today_low if ts.hour < 12 else tomorrow_low
    rad = self.seconds_to_rads(seconds_since_midnight(ts))
    return 0.5 * (math.sin(rad) + 1) * (high - low) + low

  def get_air_convection_coefficient(self, timestamp) -> float:
    return self.convection_coefficient


class BatchedSinusoidWeather:
  """One sinusoid WeatherController per building (per-building low / high bounds): the same
  formula as ``WeatherController.get_current_temp`` (weather_controller.py:93-123),
  ``T_b = 0.5*(sin(rad)+1) * (high_b - low_b) + low_b``.  The time-only factor is computed on
  the host, the per-building temperatures on the device (sb_step_in.weather_lohi_dev), so no
  per-building host work is left in a step.  Special days are not modelled."""

  def __init__(self, low_temps, high_temps, convection_coefficient: float = 12.0):
    import numpy as np
    self.low = np.ascontiguousarray(low_temps, dtype=np.float64)
    self.high = np.ascontiguousarray(high_temps, dtype=np.float64)
    if self.low.shape != self.high.shape or self.low.ndim != 1:
      raise ValueError("low_temps and high_temps must be 1-D arrays of the same length")
    if (self.low > self.high).any():
      raise ValueError("default_low_temp cannot be greater than default_high_temp.")
    self.convection_coefficient = convection_coefficient
    self._one = WeatherController(0.0, 1.0)

  def factor(self, timestamp) -> float:
    """0.5*(sin(rad(t))+1): what a (low=0, high=1) controller returns."""
    return self._one.get_current_temp(timestamp)

  def temps(self, timestamp):
    """[B] float64 ambient temperatures, bit-identical to B separate WeatherControllers."""
    f = self.factor(timestamp)
    return f * (self.high - self.low) + self.low

  def get_air_convection_coefficient(self, timestamp) -> float:
    return self.convection_coefficient


def _parse_weather_time(text: str) -> dt.datetime:
  """``pd.Timestamp(t, tz='UTC')`` on the CSV's ``Time`` column (weather_controller.py:183-185): the
  sbsim weather files write ``YYYYMMDD-HHMM`` (UTC); ISO strings are accepted as well."""
  text = text.strip()
  try:
    t = dt.datetime.strptime(text, "%Y%m%d-%H%M")
  except ValueError:
    t = dt.datetime.fromisoformat(text.replace("Z", "+00:00").replace(" UTC", "+00:00"))
  return t.replace(tzinfo=UTC) if t.tzinfo is None else t


class ReplayWeatherController:
  """Linear interpolation of an hourly ``Time,TempF`` CSV
  (simulator/weather_controller.py:166-218)."""

  def __init__(self, local_weather_path: str, convection_coefficient: float = 12.0):
    times, temps = [], []
    with open(local_weather_path, newline="") as fh:
      for row in csv.DictReader(fh):
        t = _parse_weather_time(row["Time"])
        times.append(t.timestamp())
        temps.append(float(row["TempF"]))
    self._times = np.asarray(times, dtype=np.float64)
    self._temps = np.asarray(temps, dtype=np.float64)
    self.convection_coefficient = convection_coefficient

  def get_current_temp(self, timestamp) -> float:
    ts = as_datetime(timestamp)
    if ts.tzinfo is None:
      raise TypeError("ReplayWeatherController needs a tz-aware timestamp (reference: tz_convert)")
    target = ts.timestamp()
    if target < self._times.min():
      raise ValueError(f"Attempting to get weather data at {ts}, before the latest timestamp.")
    if target > self._times.max():
      raise ValueError(f"Attempting to get weather data at {ts}, after the latest timestamp.")
    temp_f = float(np.interp(target, self._times, self._temps))
    return (temp_f - 32.0) * 5.0 / 9.0 + 273.15   # conversion_utils.py:155-171

  def get_air_convection_coefficient(self, timestamp) -> float:
    return self.convection_coefficient


class BatchedReplayWeather(ReplayWeatherController):
  """One ``ReplayWeatherController`` per building over the same CSV, building b reading the trace
  ``offsets_sec[b]`` seconds ahead (per-building weather diversity).  The
  interpolation runs on the device (``sb_step_in.weather_times_dev``): np.interp's arithmetic and
  the reference's Fahrenheit -> kelvin conversion, bit-identical to ``temps()`` below."""

  def __init__(self, local_weather_path: str, offsets_sec, convection_coefficient: float = 12.0):
    super().__init__(local_weather_path, convection_coefficient)
    self.offsets_sec = np.ascontiguousarray(offsets_sec, dtype=np.float64)
    if self.offsets_sec.ndim != 1:
      raise ValueError("offsets_sec must be one offset per building")

  @property
  def times(self) -> np.ndarray:
    return self._times

  @property
  def temps_f(self) -> np.ndarray:
    return self._temps

  def query_time(self, timestamp) -> float:
    ts = as_datetime(timestamp)
    if ts.tzinfo is None:
      raise TypeError("ReplayWeatherController needs a tz-aware timestamp (reference: tz_convert)")
    target = ts.timestamp()
    lo, hi = target + self.offsets_sec.min(), target + self.offsets_sec.max()
    if lo < self._times.min():   # weather_controller.py:196-205, for every building
      raise ValueError(f"Attempting to get weather data at {ts}, before the latest timestamp.")
    if hi > self._times.max():
      raise ValueError(f"Attempting to get weather data at {ts}, after the latest timestamp.")
    return target

  def temps(self, timestamp) -> np.ndarray:
    """[B] ambient temperatures, K."""
    x = self.query_time(timestamp) + self.offsets_sec
    temp_f = np.array([np.interp(v, self._times, self._temps) for v in x])   # scalar calls: arr_interp's scalar path
    return (temp_f - 32.0) * 5.0 / 9.0 + 273.15

  def get_current_temp(self, timestamp) -> float:
    raise TypeError("per-building weather: use temps() / BatchedEnvironment")


# --------------------------------------------------------------------------- schedule
class SetpointSchedule:
  """Day (comfort) / night (eco) temperature windows (simulator/setpoint_schedule.py:29-128)."""

  def __init__(self, morning_start_hour: int, evening_start_hour: int,
               comfort_temp_window: Tuple[float, float], eco_temp_window: Tuple[float, float],
               holidays: Optional[Set[int]] = None, time_zone="UTC"):
    if morning_start_hour > evening_start_hour:
      raise ValueError("morning_start_hour must be less than evening_start_hour")
    if comfort_temp_window[0] > comfort_temp_window[1]:
      raise ValueError("comfort_temp_window[0] must be less than comfort_temp_window[1]")
    if eco_temp_window[0] > eco_temp_window[1]:
      raise ValueError("eco_temp_window[0] must be less than eco_temp_window[1]")
    self.morning_start_hour = morning_start_hour
    self.evening_start_hour = evening_start_hour
    self.comfort_temp_window = comfort_temp_window
    self.eco_temp_window = eco_temp_window
    self._time_zone = _tz(time_zone)
    self.holidays = set(holidays) if holidays else set()

  def _localize_or_convert_timezone(self, ts: dt.datetime) -> dt.datetime:
    ts = as_datetime(ts)
    if ts.tzinfo is not None:
      return ts.astimezone(self._time_zone)
    return ts.replace(tzinfo=UTC)   # naive -> localized as UTC, NOT converted (:100-106)

  def is_weekend(self, ts) -> bool:
    return self._localize_or_convert_timezone(ts).weekday() >= 5

  def is_comfort_mode(self, ts) -> bool:
    local = self._localize_or_convert_timezone(ts)
    return (self.morning_start_hour <= local.hour < self.evening_start_hour
            and local.timetuple().tm_yday not in self.holidays
            and local.weekday() < 5)

  def get_temperature_window(self, ts) -> Tuple[float, float]:
    return self.comfort_temp_window if self.is_comfort_mode(ts) else self.eco_temp_window


# --------------------------------------------------------------------------- work days
def _nth_weekday(year: int, month: int, weekday: int, n: int) -> dt.date:
  d = dt.date(year, month, 1)
  d += dt.timedelta(days=(weekday - d.weekday()) % 7 + 7 * (n - 1))
  return d


def _last_weekday(year: int, month: int, weekday: int) -> dt.date:
  d = dt.date(year + (month == 12), month % 12 + 1, 1) - dt.timedelta(days=1)
  return d - dt.timedelta(days=(d.weekday() - weekday) % 7)


def us_federal_holidays(year: int) -> Set[dt.date]:
  """The default ``holidays.US()`` calendar the reference consults
  (utils/conversion_utils.py:62-70): federal holidays with Friday/Monday observance."""
  days = [dt.date(year, 1, 1), _nth_weekday(year, 1, 0, 3), _nth_weekday(year, 2, 0, 3),
          _last_weekday(year, 5, 0), dt.date(year, 7, 4), _nth_weekday(year, 9, 0, 1),
          _nth_weekday(year, 10, 0, 2), dt.date(year, 11, 11), _nth_weekday(year, 11, 3, 4),
          dt.date(year, 12, 25)]
  if year >= 2021:
    days.append(dt.date(year, 6, 19))
  out = set(days)
  for d in days:
    if d.weekday() == 5:
      out.add(d - dt.timedelta(days=1))
    elif d.weekday() == 6:
      out.add(d + dt.timedelta(days=1))
  if dt.date(year + 1, 1, 1).weekday() == 5:   # next New Year's Day observed on Dec 31
    out.add(dt.date(year, 12, 31))
  return out


_HOLIDAY_CACHE: Dict[int, Set[dt.date]] = {}


def is_work_day(ts, holiday_calendar: Optional[Iterable[dt.date]] = "us") -> bool:
  """conversion_utils.py:67-70: weekday < 5 and the date is not a US holiday.
  ``holiday_calendar=None`` disables the holiday test (what the golden harness ran with,
  because the ``holidays`` package is not installed here)."""
  ts = as_datetime(ts)
  if ts.weekday() >= 5:
    return False
  if holiday_calendar is None:
    return True
  if holiday_calendar == "us":
    cal = _HOLIDAY_CACHE.get(ts.year)
    if cal is None:   # (setdefault would rebuild the year's calendar on every call)
      cal = _HOLIDAY_CACHE[ts.year] = us_federal_holidays(ts.year)
  else:
    cal = holiday_calendar
  return ts.date() not in cal


# --------------------------------------------------------------------------- occupancy
class StepFunctionOccupancy:
  """Constant occupancy inside / outside working hours
  (simulator/step_function_occupancy.py:36-173)."""

  def __init__(self, work_start_time: dt.timedelta, work_end_time: dt.timedelta,
               work_occupancy: float, nonwork_occupancy: float,
               holiday_calendar: Optional[Iterable[dt.date]] = "us"):
    for t in (work_start_time, work_end_time):
      if t > dt.timedelta(hours=24) or t.total_seconds() < 0.0:
        raise ValueError("Time delta must be positive and less than one day.")
    self._work_start_time = work_start_time
    self._work_end_time = work_end_time
    self._work_occupancy = work_occupancy
    self._nonwork_occupancy = nonwork_occupancy
    self._holiday_calendar = holiday_calendar

  def _split_workday(self, start: dt.timedelta, end: dt.timedelta):
    if start > end:
      raise ValueError("Cannot have an end time before start time.")
    before = during = after = 0.0
    current = start
    interval_end = min(end, dt.timedelta(hours=24))
    nxt = min(interval_end, self._work_start_time)
    if current < nxt:
      before = (nxt - current).total_seconds()
      current = max(current, nxt)
    nxt = min(interval_end, self._work_end_time)
    if current < nxt:
      during = (nxt - current).total_seconds()
      current = nxt
    nxt = interval_end
    if current < nxt:
      after = (nxt - current).total_seconds()
    return before, during, after

  def average_zone_occupancy(self, zone_id, start_time, end_time) -> float:
    start_time, end_time = as_datetime(start_time), as_datetime(end_time)
    if start_time >= end_time:
      raise ValueError("End time may not occur before start time.")
    work = nonwork = 0.0
    day = start_time.replace(hour=0, minute=0, second=0, microsecond=0)
    if start_time.tzinfo is not None:
      day = day.replace(tzinfo=None)   # pd.Timestamp(year=, month=, day=) is naive (:88-90)
      start_time, end_time = start_time.replace(tzinfo=None), end_time.replace(tzinfo=None)
    current = start_time - day
    while day <= end_time:
      day_end = min(dt.timedelta(days=1), end_time - day)
      if is_work_day(day, self._holiday_calendar):
        b, d, a = self._split_workday(current, day_end)
        work += d
        nonwork += b + a
      else:
        nonwork += (day_end - current).total_seconds()
      day += dt.timedelta(days=1)
      current = dt.timedelta(0)
    return (work * self._work_occupancy + nonwork * self._nonwork_occupancy) / (work + nonwork)


class RandomizedArrivalDepartureOccupancy:
  """Occupants that arrive and leave at random times of a working day
  (simulator/randomized_arrival_departure_occupancy.py:41-218).  Host mirror: the same
  ``np.random.RandomState(seed)`` stream consumed in the same order, so one instance
  reproduces the reference call for call (tests/golden/occupancy_randomized.npz).  Every
  ``average_zone_occupancy`` call ADVANCES each occupant of the zone (one Bernoulli trial inside
  the arrival / departure window), exactly as in the reference -- the result depends on how
  often it is asked, not only on the time."""

  AWAY, WORK = 1, 2

  def __init__(self, zone_assignment: int, earliest_expected_arrival_hour: int,
               latest_expected_arrival_hour: int, earliest_expected_departure_hour: int,
               latest_expected_departure_hour: int, time_step_sec: int, seed: Optional[int] = 17321,
               time_zone="UTC", holiday_calendar: Optional[Iterable[dt.date]] = "us"):
    assert (earliest_expected_arrival_hour < latest_expected_arrival_hour
            < earliest_expected_departure_hour < latest_expected_departure_hour)   # :68-73
    self.zone_assignment = int(zone_assignment)
    self.hours = (int(earliest_expected_arrival_hour), int(latest_expected_arrival_hour),
                  int(earliest_expected_departure_hour), int(latest_expected_departure_hour))
    self.time_step_sec = float(time_step_sec)
    self.seed = seed
    self.p_arrival = self.event_probability(self.hours[0], self.hours[1], self.time_step_sec)
    self.p_departure = self.event_probability(self.hours[2], self.hours[3], self.time_step_sec)
    self._random_state = np.random.RandomState(seed)
    self._tz = _tz(time_zone)
    self._holiday_calendar = holiday_calendar
    self._zone_occupants: Dict[str, list] = {}

  @staticmethod
  def event_probability(start_hour: int, end_hour: int, time_step_sec: float) -> float:
    """:100-112: geometric distribution whose mean is half the window."""
    n_halfway = (end_hour - start_hour) * 3600.0 / time_step_sec / 2.0
    return 1.0 / n_halfway

  def local(self, ts) -> dt.datetime:
    ts = as_datetime(ts)
    return ts if ts.tzinfo is None else ts.astimezone(self._tz)   # :93-98

  def is_work_day(self, ts) -> bool:
    loc = self.local(ts)
    return is_work_day(dt.datetime(loc.year, loc.month, loc.day), self._holiday_calendar)   # :141-150

  def _peek(self, state: int, ts) -> int:
    """ZoneOccupant.peek (:138-160)."""
    hour = self.local(ts).hour
    e_arr, l_arr, e_dep, _ = self.hours
    if not self.is_work_day(ts):
      return self.AWAY
    if state == self.AWAY:
      if hour < e_arr or hour > l_arr:
        return state
      return self.WORK if self._random_state.rand() < self.p_arrival else state
    if hour < e_dep:
      return state
    return self.AWAY if self._random_state.rand() < self.p_departure else state

  def average_zone_occupancy(self, zone_id, start_time, end_time) -> float:
    """:183-218: the number of occupants at work at ``start_time``."""
    occ = self._zone_occupants.setdefault(zone_id, [self.AWAY] * self.zone_assignment)
    n = 0.0
    for i in range(len(occ)):
      occ[i] = self._peek(occ[i], start_time)
      if occ[i] == self.WORK:
        n += 1.0
    return n


class BatchedRandomizedArrivalDepartureOccupancy(RandomizedArrivalDepartureOccupancy):
  """One independent ``RandomizedArrivalDepartureOccupancy`` per building, generated on the
  device (``sb_occupancy_attach`` / ``sb_occupancy_peek``): the per-occupant state machine is the
  reference's, the Bernoulli trials come from a counter-based generator (Philox4x32-10 keyed by
  seed, counter = global building index, zone, query number), so the streams cannot be those of
  ``np.random.RandomState`` -- statistically equivalent, not bitwise --
  but they do not depend on how the batch is sharded over GPUs.  ``BatchedEnvironment`` asks
  twice per step like the reference's environment: the reward's query, then ``num_occupants``'
  query five minutes earlier.  ``first_building``: global index of the batch's building 0."""

  def __init__(self, *args, first_building: int = 0, **kwargs):
    super().__init__(*args, **kwargs)
    if not 1 <= self.zone_assignment <= 32:
      raise ValueError("the device generator keeps a zone's occupants in one 32-bit word")
    self.first_building = int(first_building)

  def average_zone_occupancy(self, zone_id, start_time, end_time) -> float:
    raise TypeError("per-building occupancy lives on the device; use BatchedEnvironment")


class StochasticConvectionSimulator:
  """Parameters of the reference's StochasticConvectionSimulator(p, distance, seed)
  (simulator/stochastic_convection_simulator.py:35-60; SB1: p = 1, distance = 5,
  sim_config.gin:36-39).  The shuffle itself runs on the device after every FD update
  (``sb_convection_attach``, ``BatchedEnvironment(convection_simulator=...)``): the reference's
  random process with counter-based draws per building -- statistically equivalent to the
  reference (which uses Python's global ``random``), independent of batch sharding."""

  def __init__(self, p: float, distance: int, seed: Optional[int], first_building: int = 0):
    self.p, self.distance, self.seed, self.first_building = float(p), int(distance), seed, int(first_building)


# --------------------------------------------------------------------------- tariffs
# Units: kg carbon / MWh (reward/electricity_energy_cost.py:39-64).
CARBON_EMISSION_BY_HOUR = (
    88.19666493, 87.79190866, 87.87607686, 87.83054163, 88.00279618, 88.19648183,
    89.70663283, 93.97947901, 98.85868291, 100.7853521, 101.3866866, 101.7795612,
    102.5919168, 103.4403736, 104.1380294, 104.7359292, 102.0714466, 97.04226176,
    93.57895651, 92.46355045, 91.72914657, 90.69209747, 89.76552213, 88.99950995)
# cents / kWh (reward/electricity_energy_cost.py:72-123).
WEEKDAY_PRICE_BY_HOUR = (16.0,) * 6 + (18.0,) * 6 + (20.0,) * 7 + (16.0,) * 5
WEEKEND_PRICE_BY_HOUR = (16.0,) * 24
# USD / 1000 ft3, 2020 (reward/natural_gas_energy_cost.py:31-44).
GAS_PRICE_BY_MONTH = (9.02, 8.35, 7.77, 7.26, 6.69, 6.86, 6.77, 6.76, 6.99, 7.19, 7.96, 8.98)
KWH_PER_KFT3_GAS = 293.07107   # utils/constants.py:40
JOULES_PER_KWH = 3.6e6         # utils/constants.py:29
GAS_CO2 = 53.12                # utils/constants.py:44


class ElectricityEnergyCost:
  """reward/electricity_energy_cost.py:125-224, reduced to the two rates the kernel needs."""

  def __init__(self, weekday_energy_prices: Sequence[float] = WEEKDAY_PRICE_BY_HOUR,
               weekend_energy_prices: Sequence[float] = WEEKEND_PRICE_BY_HOUR,
               carbon_emission_rates: Sequence[float] = CARBON_EMISSION_BY_HOUR,
               holiday_calendar: Optional[Iterable[dt.date]] = "us"):
    if len(weekday_energy_prices) != 24 or len(weekend_energy_prices) != 24:
      raise ValueError("Energy cost rates must have 24 entries.")
    if len(carbon_emission_rates) != 24:
      raise ValueError("Carbon emission rates must have 24 entries.")
    self.carbon_emission_rates = np.array(carbon_emission_rates) / 1.0e6 / 3600.0
    self.weekday_energy_prices = np.array(weekday_energy_prices) / 100.0 / 1000.0 / 3600.0
    self.weekend_energy_prices = np.array(weekend_energy_prices) / 100.0 / 1000.0 / 3600.0
    self._holiday_calendar = holiday_calendar

  def rates(self, start_time_utc: dt.datetime) -> Tuple[float, float]:
    """(USD per W per s, kg per W per s) at ``start_time.hour`` (UTC: SURVEY.md 7.8)."""
    hour = start_time_utc.hour
    if is_work_day(start_time_utc, self._holiday_calendar):
      price = self.weekday_energy_prices[hour]
    else:
      price = self.weekend_energy_prices[hour]
    return float(price), float(self.carbon_emission_rates[hour])


class NaturalGasEnergyCost:
  """reward/natural_gas_energy_cost.py:47-138."""

  def __init__(self, gas_price_by_month: Sequence[float] = GAS_PRICE_BY_MONTH):
    assert len(gas_price_by_month) == 12, "Gas price per month must have exactly 12 values."
    self.month_gas_price = np.array(gas_price_by_month) / KWH_PER_KFT3_GAS / JOULES_PER_KWH
    self.carbon_rate = GAS_CO2 / KWH_PER_KFT3_GAS / JOULES_PER_KWH

  def rates(self, start_time_utc: dt.datetime) -> Tuple[float, float]:
    return float(self.month_gas_price[start_time_utc.month - 1]), float(self.carbon_rate)


def reward_start_time_utc(ts: dt.datetime) -> dt.datetime:
  """conversion_utils.py:39-59: the reward sees pandas_to_proto -> proto_to_pandas, i.e. UTC."""
  return dt.datetime.fromtimestamp(int(epoch_seconds(ts)), tz=UTC)


# --------------------------------------------------------------------------- time features
def get_radian_time(ts, hour_of_day: bool) -> float:
  """conversion_utils.py:107-134 (in the timestamp's own zone)."""
  ts = as_datetime(ts)
  if hour_of_day:
    frac = seconds_since_midnight(ts) / _SECONDS_IN_A_DAY
  else:
    frac = float(ts.weekday()) / 7.0
  return 2.0 * np.pi * frac
