"""Developer aid: which sweep kernel the planner picks over a grid of floor-plan shapes (2 x 2 rooms) and what it delivers:
ms per step of the sweep kernel and cell-sweeps/s at B buildings -- finds the planner's weak regions.
Usage (GPU box): python tools/scan_plan_shapes.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from sbsim_amd import _ffi  # noqa: E402
from sbsim_amd.environment import BatchedEnvironment  # noqa: E402
from sbsim_amd.floorplan import FloorPlan, Materials, rectangular_floor_plan  # noqa: E402

dev = torch.device("cuda", 0)
B, K, WARM = int(os.environ.get("B", "8192")), 5, 3
ROWS = [int(v) for v in os.environ.get("ROWS", "21,41,61,65,81,101,129,161,201,251").split(",")]
COLS = [int(v) for v in os.environ.get("COLS", "31,51,71,79,87,95").split(",")]
print("rows x cols inside the exterior ring: kernel (wavefronts per building), sweep kernel ms/step, sweeps, 1e11 cell-sweeps/s")
for R in ROWS:
  line = []
  for Cc in COLS:
    h, w = (R - 5) // 2, (Cc - 5) // 2
    plan = FloorPlan.from_file_input(rectangular_floor_plan((2, 2), (h, w)), Materials.sb1(), 10.0, 300.0)
    env = BatchedEnvironment(plan, B, device=0, holiday_calendar="us", collect_info=True, num_days_in_episode=3)
    H, W = plan.shape
    rs = np.random.RandomState(7)
    t_init = torch.tensor(np.clip(294.0 + rs.randn(B), 285.0, 305.0), dtype=torch.float64, device=dev)
    env.reset()
    env.sim.reset(temps=t_init[:, None].expand(B, H * W).contiguous())
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234)
    acts = torch.rand((K + WARM, B, 2), generator=gen, device=dev, dtype=torch.float32) * 2.0 - 1.0
    ev, sw = [], []
    for t in range(K + WARM):
      si = env.make_step_in(env.current_simulation_timestamp)
      a = (acts[t], si, env._obs, env._reward, env._info)
      env.sim.step(*a, phases=1)
      e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
      e0.record()
      env.sim.step(*a, phases=2)
      e1.record()
      env.sim.step(*a, phases=4)
      env._prev_thermostat_ts = env._now
      env._now = env._now + env._step_interval
      if t >= WARM:
        ev.append((e0, e1))
        sw.append(float(env._info[:, 4].mean()))
    torch.cuda.synchronize()
    ms = float(np.mean([x.elapsed_time(y) for x, y in ev]))
    li = env.sim.launch_info
    rate = B * H * W * np.mean(sw) / (ms * 1e-3) / 1e11
    line.append(f"{H - 2}x{W - 2}{'T' if env.sim.transposed else ''}: {_ffi.SWEEP_KERNELS[li['kernel']].replace('k_sweep_', '')}({li['waves_per_building']}) {ms:.2f} {np.mean(sw):.1f} {rate:.2f}")
    env.close()
  print(" | ".join(line), flush=True)
