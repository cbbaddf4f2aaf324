"""CPU check of the HIP sweep's schedule and class tables against the oracle."""
import numpy as np
import pytest

from oracle import oracle as orc
from sbsim_amd.floorplan import FloorPlan
from tests.golden_util import load, oracle_plan
from tests.kernel_model import FastSweepModel, model_fd_timestep


def _plan_from_golden(p):
  return FloorPlan(conductivity=p["conductivity"], heat_capacity=p["heat_capacity"],
                   density=p["density"], exterior_space=p["exterior_space"],
                   zone_label=p["zone_label"], diffusers=p["diffusers"],
                   cv_size_cm=float(p["cv_size_cm"]), floor_height_cm=float(p["floor_height_cm"]))


@pytest.mark.parametrize("name", ["small_test", "weird_test", "r9_test"])
def test_skewed_class_table_sweep_matches_oracle(name):
  g = load(f"sweep_{name}.npz")
  p = load(f"plan_{name}.npz")
  fp = _plan_from_golden(p)
  cp = fp.compile(float(g["dt"]), float(g["h"]))
  # per-zone power that reproduces the golden input_q (q = q_zone * diffuser)
  qz = np.zeros(cp.Z)
  q = g["q"]
  for z, cells in enumerate(fp.zone_cell_lists()):
    d = fp.diffusers.reshape(-1)[cells]
    sel = d > 0
    if sel.any():
      qz[z] = (q.reshape(-1)[cells][sel] / d[sel])[0]
  # rebuild q consistently (one power per zone) for both sides
  q2 = np.zeros_like(q)
  for z, cells in enumerate(fp.zone_cell_lists()):
    q2.reshape(-1)[cells] = qz[z] * fp.diffusers.reshape(-1)[cells]
  plan = oracle_plan(p)
  ref, n_ref, _ = orc.fd_timestep(plan, g["prev"], q2, float(g["t_amb"]), float(g["h"]),
                                  float(g["dt"]), 0.01, 60)
  got, n_got = model_fd_timestep(cp, g["prev"], float(g["t_amb"]), qz, 0.01, 60)
  assert n_got == n_ref
  assert np.abs(got - ref).max() < 1e-10
  fm = FastSweepModel(cp)
  assert fm.fast
  got2, n2 = fm.fd_timestep(g["prev"], float(g["t_amb"]), qz, 0.01, 60)
  assert n2 == n_ref
  assert np.abs(got2 - ref).max() < 1e-10
