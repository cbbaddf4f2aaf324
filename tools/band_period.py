"""Developer aid: step_band.hip's steady rolling period without cycle stamps -- a convergence threshold of 0 makes every
step run to the iteration limit (one block of limit - 1 rolling periods + the final one), one building per CU:
sweep-kernel time / limit = one period of NR steps on all of the building's wavefronts.  Usage (GPU box):
python tools/band_period.py"""
import dataclasses
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from sbsim_amd.environment import BatchedEnvironment, SimConfig  # noqa: E402
from sbsim_amd.floorplan import FloorPlan, Materials, rectangular_floor_plan  # noqa: E402

LIMIT = int(os.environ.get("LIMIT", "100"))
B = int(os.environ.get("B", "256"))
for name, rooms, shape in [("205x89", (10, 4), (19, 20)), ("195x89", (10, 4), (18, 20)), ("158x77", (9, 4), (16, 17)), ("257x80", (12, 3), (20, 24))]:
  plan = FloorPlan.from_file_input(rectangular_floor_plan(rooms, shape), Materials.sb1(), 10.0, 300.0)
  cfg = dataclasses.replace(SimConfig.sb1(), convergence_threshold=0.0, iteration_limit=LIMIT)
  env = BatchedEnvironment(plan, B, config=cfg, device=0, holiday_calendar="us", collect_info=True, num_days_in_episode=3)
  env.reset()
  acts = torch.zeros((B, 2), dtype=torch.float32, device="cuda")
  ms = []
  for t in range(5):
    si = env.make_step_in(env.current_simulation_timestamp)
    a = (acts, si, env._obs, env._reward, env._info)
    env.sim.step(*a, phases=1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    env.sim.step(*a, phases=2)
    e1.record()
    env.sim.step(*a, phases=4)
    torch.cuda.synchronize()
    ms.append(e0.elapsed_time(e1))
  li = env.sim.launch_info
  nr = li["sweep_steps"] - (4 if name == "195x89" else 0)   # (the 195 x 89 plan has a tail row: + 4 steps)
  per = np.mean(ms[2:]) * 1e3 / LIMIT
  print(f"{name}: kernel {li['kernel']}, {li['waves_per_building']} wavefronts, sweeps {float(env._info[:, 4].mean()):.0f}: {per:.2f} us per sweep = "
        f"{per / nr * 1e3:.0f} ns per step of {nr} (at 2.4 GHz: {per / nr * 2.4e3:.0f} cycles)", flush=True)
  env.close()
