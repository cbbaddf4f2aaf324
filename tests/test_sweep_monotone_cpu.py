"""What `k_sweep_two`'s measure-free periods rest on: inside an env step the Gauss-Seidel sweeps are a
stationary iteration x' = L x' + U x + c with non-negative coefficients whose four sum to at most 1 per cell, so
delta_{k+1} = (I - L)^-1 U delta_k and max|delta| never grows from one sweep to the next.  Checked here on the CPU oracle
(oracle/sb_oracle.c, the reference's arithmetic): (i) the coefficient condition the planner tests (`Dev::two_skip`) holds for
the compiled class tables of the floor plans the benches and tests use; (ii) the oracle's max|delta| series is monotone
on them, from rough starts, with and without zone power, past the stopping sweep."""
import numpy as np
import pytest

from oracle import oracle as orc
from sbsim_amd.floorplan import FloorPlan, Materials, rectangular_floor_plan

PLANS = {
    "R9": ((3, 3), (20, 30)),
    "SB2-synth": ((8, 5), (12, 14)),
    "SB1-synth": ((14, 9), (8, 7)),
    "small": ((2, 2), (5, 9)),
}


def _plan(name):
  rooms, shape = PLANS[name]
  return FloorPlan.from_file_input(rectangular_floor_plan(rooms, shape), Materials.sb1(), 10.0, 300.0)


@pytest.mark.parametrize("name", list(PLANS))
def test_class_coefficients_are_a_contraction(name):
  plan = _plan(name)
  coef = plan.compile(300.0, 100.0).class_coef   # [classes][bU bD bL bR ap gc sc pad] at SB1's time step and convection coefficient
  b = np.asarray(coef)[:, :4]
  assert (b >= 0.0).all() and (b.sum(axis=1) <= 1.0).all()
  # the rest of a cell's weight is on its own previous temperature and on the ambient air: nothing is amplified
  assert (np.asarray(coef)[:, 4] >= 0.0).all()


@pytest.mark.parametrize("name,seed", [("R9", 0), ("R9", 1), ("SB2-synth", 2), ("SB1-synth", 3), ("small", 4)])
def test_max_delta_never_grows_between_sweeps(name, seed):
  plan = _plan(name)
  oplan = orc.OraclePlan(plan.conductivity, plan.density, plan.heat_capacity, plan.exterior_space,
                         plan.zone_cell_lists(), plan.diffusers, plan.cv_size_cm, plan.floor_height_cm)
  H, W = plan.shape
  rs = np.random.RandomState(seed)
  prev = np.clip(294.0 + 3.0 * rs.randn(H * W), 280.0, 310.0)           # a rough field: every mode is excited
  q = np.zeros(H * W)
  if seed % 2:                                                            # zone power on the diffusers, both signs
    for z, cells in enumerate(plan.zone_cell_lists()):
      q[np.asarray(cells)[:: 7]] = (-1.0) ** z * 40.0 * rs.rand()
  est = prev.copy()
  n = 60 if H * W < 9000 else 36
  md = [orc.sweep(oplan, prev, est, q, 281.5, 100.0, 300.0) for _ in range(n)]
  assert md[0] > 0.1                                                      # the series starts above the threshold ...
  assert all(b <= a * (1.0 + 1e-12) for a, b in zip(md, md[1:])), md      # ... and never grows
  assert md[-1] < md[0]
