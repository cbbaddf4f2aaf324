import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from sbsim_amd.environment import BatchedEnvironment
from sbsim_amd.floorplan import FloorPlan, Materials, rectangular_floor_plan
for rooms, shape in (((2, 3), (20, 30)), ((2, 3), (9, 10))):
  plan = FloorPlan.from_file_input(rectangular_floor_plan(rooms, shape), Materials.sb1(), 10.0, 300.0)
  B = 65536
  env = BatchedEnvironment(plan, B, holiday_calendar=None, collect_info=True)
  env.reset()
  acts = torch.rand((36, B, 2), device="cuda") * 2 - 1
  for t in range(12): env.step(acts[t])
  torch.cuda.synchronize(); t0 = time.perf_counter(); sw = 0.0
  for t in range(12, 36):
    env.step(acts[t]); sw += float(env.info[:, 4].mean())
  torch.cuda.synchronize()
  print(rooms, shape, env.sim.launch_info["sweep_steps"], f"{(time.perf_counter()-t0)/24*1e3:.3f} ms/step, sweeps {sw/24:.2f}")
  env.close()
