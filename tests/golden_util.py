"""Helpers shared by the golden-vector tests (oracle side only)."""
import json
import os

import numpy as np

from oracle import oracle as orc

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
  return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def zones_from_label(label):
  """room_dict lists are raster-ordered (building_utils.py:406-414)."""
  flat = np.asarray(label).reshape(-1)
  return [np.nonzero(flat == z)[0].astype(np.int32) for z in range(int(flat.max()) + 1)]


def oracle_plan(plan_npz, diffusers=None, skip_exterior=True):
  p = plan_npz
  return orc.OraclePlan(
      p["conductivity"], p["density"], p["heat_capacity"], p["exterior_space"],
      zones_from_label(p["zone_label"]), p["diffusers"] if diffusers is None else diffusers,
      float(p["cv_size_cm"]), float(p["floor_height_cm"]), skip_exterior=skip_exterior)


def oracle_params(params_json, **over):
  d = json.loads(str(params_json))
  for k in ("morning_start_hour", "evening_start_hour", "holidays"):
    d.pop(k, None)
  d.update(over)
  return orc.OracleParams(**d)
