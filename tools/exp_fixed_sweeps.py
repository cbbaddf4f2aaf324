"""Dev tool: the R9 bench batch at FIXED sweep counts (convergence threshold < 0, iteration_limit = n):
time per step against n gives the per-sweep cost under full load (slope) and the fixed cost
(intercept) -- comparable between builds whose results differ (timing experiments, SBSIM_LIB=...)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sbsim_amd.environment import BatchedEnvironment, SimConfig
from bench import r9_plan

B = int(os.environ.get("B", 65536))
plan = r9_plan()
if os.environ.get("PLAN"):   # PLAN=SB2-synth / SB1-synth: bench.py's mixed classes (B=21845)
  import bench
  from sbsim_amd.floorplan import FloorPlan, Materials, rectangular_floor_plan
  name, rooms, shape = [c for c in bench.MIXED_CLASSES if c[0] == os.environ["PLAN"]][0]
  plan = FloorPlan.from_file_input(rectangular_floor_plan(rooms, shape), Materials.sb1(), 10.0, 300.0)
res = []
for lim in [int(x) for x in os.environ.get("LIMS", "1,3,5,9").split(",")]:
  env = BatchedEnvironment(plan, B, config=SimConfig(iteration_limit=lim, convergence_threshold=-1.0), holiday_calendar=None,
                           collect_info=True)
  env.reset()
  acts = torch.rand((20, B, 2), device="cuda") * 2 - 1
  for t in range(4):
    env.step(acts[t])
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for t in range(4, 16):
    env.step(acts[t])
  torch.cuda.synchronize()
  dt = (time.perf_counter() - t0) / 12
  res.append((lim, dt * 1e3))
  print(f"sweeps={lim}: {dt*1e3:.3f} ms/step (mean sweeps reported {float(env.info[:, 4].mean()):.2f})", flush=True)
  env.close()
(l0, t0_), (l1, t1_) = res[0], res[-1]
slope = (t1_ - t0_) / (l1 - l0)
print(f"{os.environ.get('SBSIM_LIB', 'product')}: per sweep {slope:.4f} ms, fixed {t0_ - slope * l0:.4f} ms")
