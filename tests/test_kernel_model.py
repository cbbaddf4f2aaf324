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


# ---- register-resident sweep (step_reg.hip): schedule, tail recurrence, seam protocol ----
def _r9_like(rooms, room_shape):
  from sbsim_amd.floorplan import Materials, rectangular_floor_plan
  return FloorPlan.from_file_input(rectangular_floor_plan(rooms, room_shape), Materials.sb1(), 10.0, 300.0)


@pytest.mark.parametrize("rooms,room_shape,transpose,mode,schedule", [
    ((2, 2), (5, 9), False, 1, "tight"),      # 15x23 inside the ring: one wavefront, 32 slots
    ((2, 3), (9, 10), False, 1, "tight"),     # 23x36: one wavefront, 66 slots
    ((3, 3), (20, 30), False, 3, "tight"),    # R9, lanes = rows: 64 rows + 2 tail rows, 96 slots
    ((2, 3), (30, 30), False, 3, "tight"),    # one tail row
    ((3, 3), (20, 30), False, 3, "rolling"),  # R9 with overlapped sweeps (speculative start + undo)
    ((2, 3), (30, 30), False, 3, "rolling"),
    ((2, 3), (20, 30), False, 1, "rolling"),  # 43 x 94 on 96 slots: the schedule also holds with idle lanes (kernel: SB_ROLL_MODE1)
    ((3, 3), (20, 30), True, 2, "tight"),     # R9, lanes = columns: two wavefronts, wave 1 as early as allowed
    ((3, 3), (20, 30), True, 2, "late"),      # ... and as late as possible: same grid
    ((2, 5), (30, 12), True, 2, "tight"),     # uneven split
])
def test_register_sweep_schedule_matches_oracle(rooms, room_shape, transpose, mode, schedule):
  from tests.kernel_model import RegSweepModel
  fp0 = _r9_like(rooms, room_shape)
  fp = fp0.transposed() if transpose else fp0
  dt, h = 300.0, 100.0
  cp = fp.compile(dt, h)
  m = RegSweepModel(cp)
  assert m.mode == mode
  rs = np.random.RandomState(5)
  H, W = fp.shape
  prev = np.clip(293.0 + 1.5 * rs.randn(H, W), 285.0, 300.0)
  qz = rs.uniform(-400.0, 900.0, size=cp.Z)
  q = np.zeros((H, W))
  for z, cells in enumerate(fp.zone_cell_lists()):
    q.reshape(-1)[cells] = qz[z] * fp.diffusers.reshape(-1)[cells]
  plan = orc.OraclePlan(fp.conductivity, fp.density, fp.heat_capacity, fp.exterior_space,
                        fp.zone_cell_lists(), fp.diffusers, fp.cv_size_cm, fp.floor_height_cm)
  ref, n_ref, _ = orc.fd_timestep(plan, prev, q, 279.5, h, dt, 0.05, 40)
  got, n_got = m.fd_timestep(prev, 279.5, qz, 0.05, 40, schedule=schedule)
  assert n_got == n_ref and n_ref >= 3
  assert np.abs(got - ref).max() < 1e-10


def test_seam_lag_is_tight():
  """One chunk less lag and wave 1 would read a seam value before it is published."""
  from tests import kernel_model as km
  fp = _r9_like((3, 3), (20, 30)).transposed()
  cp = fp.compile(300.0, 100.0)
  m = km.RegSweepModel(cp)
  assert m.mode == 2 and m.lag == km.seam_lag(m.lw[0])
  H, W = fp.shape
  prev = np.full((H, W), 294.0)
  m.lag -= 1
  with pytest.raises(AssertionError, match="before wave 0 published"):
    m.fd_timestep(prev, 280.0, np.zeros(cp.Z), 0.05, 2, schedule="tight")


def test_overlapped_sweeps_stop_at_the_iteration_limit():
  """The started sweep is undone when the limit, not convergence, ends the step."""
  from tests.kernel_model import RegSweepModel
  fp = _r9_like((3, 3), (20, 30))
  dt, h = 300.0, 100.0
  cp = fp.compile(dt, h)
  m = RegSweepModel(cp)
  rs = np.random.RandomState(11)
  H, W = fp.shape
  prev = np.clip(293.0 + 1.5 * rs.randn(H, W), 285.0, 300.0)
  qz = rs.uniform(-400.0, 900.0, size=cp.Z)
  q = np.zeros((H, W))
  for z, cells in enumerate(fp.zone_cell_lists()):
    q.reshape(-1)[cells] = qz[z] * fp.diffusers.reshape(-1)[cells]
  plan = orc.OraclePlan(fp.conductivity, fp.density, fp.heat_capacity, fp.exterior_space,
                        fp.zone_cell_lists(), fp.diffusers, fp.cv_size_cm, fp.floor_height_cm)
  for limit in (1, 3):
    ref, n_ref, _ = orc.fd_timestep(plan, prev, q, 279.5, h, dt, 1e-9, limit)
    got, n_got = m.fd_timestep(prev, 279.5, qz, 1e-9, limit, schedule="rolling")
    assert n_got == n_ref == limit
    assert np.abs(got - ref).max() < 1e-10
