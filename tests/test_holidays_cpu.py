"""`us_federal_holidays` (sbsim_amd/host_inputs.py) decides `is_work_day` -- the tariffs' weekday / weekend table and the
step-function occupancy (utils/conversion_utils.py:62-70 consults `holidays.US()`, which is not installed here).  A second,
independent source for the same calendar is in the image: pandas' `USFederalHolidayCalendar`.  Every WEEKDAY it marks as a
federal holiday between 1990 and 2040 must be one of ours and the other way round (`is_work_day` never asks about a
Saturday or Sunday: `weekday() < 5` comes first, conversion_utils.py:69).  No difference in 51 years; the rules that could
have made one (Juneteenth from 2021 on, the Friday / Monday observance, New Year's Day observed on December 31) are asked
by name below."""
import datetime as dt

import pandas as pd
import pytest
from pandas.tseries.holiday import USFederalHolidayCalendar

from sbsim_amd import host_inputs

YEARS = range(1990, 2041)


def _pandas_weekdays(year):
  days = USFederalHolidayCalendar().holidays(start=f"{year}-01-01", end=f"{year}-12-31")
  return {d.date() for d in days if d.weekday() < 5}


def _ours_weekdays(year):
  return {d for d in host_inputs.us_federal_holidays(year) if d.weekday() < 5 and d.year == year}


def test_federal_holidays_agree_with_pandas_calendar_1990_2040():
  only_ours, only_pandas = [], []
  for y in YEARS:
    ours, theirs = _ours_weekdays(y), _pandas_weekdays(y)
    only_ours += sorted(ours - theirs)
    only_pandas += sorted(theirs - ours)
  # (holidays.US observes New Year's Day on the Friday BEFORE when January 1 is a Saturday -- December 31 of the old
  # year; pandas' nearest_workday rule reports the same Friday when asked about the old year's range)
  assert only_pandas == [], only_pandas
  assert only_ours == [], only_ours


@pytest.mark.parametrize("year,month,day,name", [
    (2021, 6, 18, "Juneteenth (observed: June 19, 2021 was a Saturday)"),
    (2022, 6, 20, "Juneteenth (observed: a Sunday)"),
    (2020, 6, 19, None),                      # not a federal holiday before 2021
    (2021, 12, 31, "New Year's Day 2022 (observed)"),
    (2023, 1, 2, "New Year's Day (observed: a Sunday)"),
    (2023, 11, 10, "Veterans Day (observed: a Saturday)"),
    (2024, 11, 28, "Thanksgiving"),
    (2024, 5, 27, "Memorial Day"),
    (2025, 1, 20, "Martin Luther King Jr. Day"),
    (2025, 2, 17, "Washington's Birthday"),
    (2025, 9, 1, "Labor Day"),
    (2025, 10, 13, "Columbus Day"),
    (2026, 7, 3, "Independence Day (observed: a Saturday)"),
    (2027, 12, 24, "Christmas Day (observed: a Saturday)"),
])
def test_named_days(year, month, day, name):
  d = dt.date(year, month, day)
  assert (d in host_inputs.us_federal_holidays(year)) == (name is not None)
  assert host_inputs.is_work_day(dt.datetime(year, month, day, 12, 0), "us") == (name is None and d.weekday() < 5)
  assert (pd.Timestamp(d) in USFederalHolidayCalendar().holidays(start=f"{year}-01-01", end=f"{year}-12-31")) == (name is not None)


def test_work_days_per_year_match_pandas_business_days():
  """The number of work days per year, the quantity the tariffs and the occupancy schedule integrate over."""
  from pandas.tseries.offsets import CustomBusinessDay
  bd = CustomBusinessDay(calendar=USFederalHolidayCalendar())
  for y in (1999, 2010, 2021, 2022, 2023, 2032):
    n_pandas = len(pd.date_range(f"{y}-01-01", f"{y}-12-31", freq=bd))
    n_ours = sum(host_inputs.is_work_day(dt.datetime(y, 1, 1) + dt.timedelta(days=k), "us") for k in range(366)
                 if (dt.datetime(y, 1, 1) + dt.timedelta(days=k)).year == y)
    assert n_ours == n_pandas, (y, n_ours, n_pandas)
