"""Developer aid: how much does a building's sweep count change from one env step to the next?
(The two-rows-per-lane kernel predicts it to overlap sweeps.)  Prints the distribution of
n_t - n_{t-1} for the mixed-config classes under random setpoint actions."""
import numpy as np
import torch

import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
from sbsim_amd import _ffi
from sbsim_amd.environment import BatchedEnvironment
from sbsim_amd.floorplan import FloorPlan, Materials, rectangular_floor_plan

dev = torch.device("cuda", 0)
for name, rooms, shape in bench.MIXED_CLASSES:
  plan = FloorPlan.from_file_input(rectangular_floor_plan(rooms, shape), Materials.sb1(), 10.0, 300.0)
  B = 2048
  env = BatchedEnvironment(plan, B, device=0, holiday_calendar="us", collect_info=True, num_days_in_episode=3)
  rs = np.random.RandomState(7)
  H, W = plan.shape
  t_init = torch.tensor(np.clip(294.0 + rs.randn(B), 285.0, 305.0), dtype=torch.float64, device=dev)
  env.reset()
  env.sim.reset(temps=t_init[:, None].expand(B, H * W).contiguous())
  gen = torch.Generator(device=dev)
  gen.manual_seed(1234)
  prev, diffs = None, []
  for t in range(40):
    a = torch.rand((B, 2), generator=gen, device=dev, dtype=torch.float32) * 2.0 - 1.0
    env.step(a)
    n = env._info[:, 4].cpu().numpy().astype(int)
    if prev is not None and t >= 5:
      diffs.append(n - prev)
    prev = n
  d = np.concatenate(diffs)
  vals, cnt = np.unique(d, return_counts=True)
  print(name, "mean n", prev.mean(), "min/max", prev.min(), prev.max())
  print("  diff quantiles 0.1% 1% 5% 50% 95% 99%:", np.percentile(d, [0.1, 1, 5, 50, 95, 99]))
  print("  P(n_t < n_prev - 2) =", (d < -2).mean(), " P(< -3) =", (d < -3).mean(), " P(< -5) =", (d < -5).mean())
