"""Developer aid: random floor-plan shapes through the library's own kernel choice against the LDS-grid kernel (the streaming
kernel where the plan does not fit a CU's LDS): sweep counts per building and step must be equal, zone temperatures within
1e-9 K.  Usage (GPU box): N=60 SEED=1 python tools/fuzz_planner.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from sbsim_amd import _ffi  # noqa: E402
from sbsim_amd.environment import BatchedEnvironment  # noqa: E402
from sbsim_amd.floorplan import FloorPlan, Materials, rectangular_floor_plan  # noqa: E402

N, SEED = int(os.environ.get("N", "40")), int(os.environ.get("SEED", "1"))
B, T = int(os.environ.get("B", "512")), int(os.environ.get("T", "40"))
rs = np.random.RandomState(SEED)
bad = 0
for it in range(N):
  while True:
    r0, r1 = int(rs.randint(1, 8)), int(rs.randint(1, 6))
    h, w = int(rs.randint(4, 60)), int(rs.randint(4, 45))
    rows, cols = r0 * (h + 1) + 3, r1 * (w + 1) + 3
    if 12 <= rows <= 262 and 12 <= cols <= 110 and r0 * r1 <= 60:
      break
  plan = FloorPlan.from_file_input(rectangular_floor_plan((r0, r1), (h, w)), Materials.sb1(), 10.0, 300.0)
  envs = []
  for force in (None, "SBSIM_FORCE_LDS_PATH", "SBSIM_FORCE_STREAM_PATH"):
    if force:
      os.environ[force] = "1"
    try:
      e = BatchedEnvironment(plan, B, holiday_calendar="us", collect_info=True, num_days_in_episode=2)
    except Exception as ex:  # the forced kernel does not hold this plan
      e = None
      if not force:
        raise
    finally:
      if force:
        os.environ.pop(force)
    if e is not None:
      if force and e.sim.launch_info["kernel"] == envs[0].sim.launch_info["kernel"] and e.sim.transposed == envs[0].sim.transposed:
        e.close()
        continue
      e.reset()
      envs.append(e)
    if len(envs) == 2:
      break
  gen = torch.Generator(device="cuda")
  gen.manual_seed(3 + it)
  mism, worst = 0, 0.0
  for t in range(T):
    a = torch.rand((B, 2), generator=gen, device="cuda") * 2 - 1
    outs = [e.step(a) for e in envs]
    if len(envs) < 2 or outs[0].step_type[0].item() == 0:
      continue
    mism += int((envs[0].info[:, 4] != envs[1].info[:, 4]).sum())
    worst = max(worst, float((envs[0].sim.zone_temps() - envs[1].sim.zone_temps()).abs().max()))
  li = envs[0].sim.launch_info
  other = _ffi.SWEEP_KERNELS[envs[1].sim.launch_info["kernel"]] if len(envs) == 2 else "none"
  ok = mism == 0 and worst < 1e-9   # (a plan only the streaming kernel holds has nothing to be compared with: "against none")
  bad += 0 if ok else 1
  print(f"{rows}x{cols} / {r0 * r1} zones{' T' if envs[0].sim.transposed else ''}: {_ffi.SWEEP_KERNELS[li['kernel']]}({li['waves_per_building']}, {li['sweep_steps']} steps)"
        f" against {other}: sweep-count mismatches {mism}, max |dT_zone| {worst:.2e} {'ok' if ok else 'FAILED'}", flush=True)
  for e in envs:
    e.close()
print("plans", N, "failed", bad)
