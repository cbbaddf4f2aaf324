"""BASELINE.json configs[2]: 65,536 buildings split evenly over three floor-plan classes
(R9; "SB2-synth" 8x5 rooms, 40 zones; "SB1-synth" 14x9 rooms, 126 zones -- SURVEY.md 8d), random
setpoint actions, one MI355X.  Not the driver's bench line (that is bench.py, configs[1]); prints
one JSON object with the aggregate zone-updates/s and the per-class step times."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sbsim_amd.environment import BatchedEnvironment   # noqa: E402
from sbsim_amd.floorplan import FloorPlan, Materials, rectangular_floor_plan   # noqa: E402

CLASSES = [("R9", (3, 3), (20, 30)), ("SB2-synth", (8, 5), (12, 14)), ("SB1-synth", (14, 9), (8, 7))]
B_TOTAL, K, W = int(os.environ.get("B", 65536)), int(os.environ.get("STEPS", 12)), int(os.environ.get("WARM", 12))

envs = []
for name, rooms, shape in CLASSES:
  plan = FloorPlan.from_file_input(rectangular_floor_plan(rooms, shape), Materials.sb1(), 10.0, 300.0)
  B = B_TOTAL // len(CLASSES)
  env = BatchedEnvironment(plan, B, holiday_calendar="us", collect_info=True)
  rs = np.random.RandomState(7)
  H, Wd = plan.shape
  t_init = torch.tensor(np.clip(294.0 + rs.randn(B), 285.0, 305.0), dtype=torch.float64, device="cuda")
  env.reset()
  env.sim.reset(temps=t_init[:, None].expand(B, H * Wd).contiguous())
  gen = torch.Generator(device="cuda")
  gen.manual_seed(1234)
  acts = torch.rand((W + K, B, 2), generator=gen, device="cuda") * 2.0 - 1.0
  envs.append((name, env, acts))

for t in range(W):
  for _, env, acts in envs:
    env.step(acts[t])
torch.cuda.synchronize()
ev = {name: [] for name, _, _ in envs}
t0 = time.perf_counter()
for t in range(W, W + K):
  for name, env, acts in envs:
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    env.step(acts[t])
    b.record()
    ev[name].append((a, b))
torch.cuda.synchronize()
elapsed = time.perf_counter() - t0
zone_updates = sum(env.sim.B * env.sim.Z * K for _, env, _ in envs)
print(json.dumps({
    "workload": "BASELINE.json configs[2]: 3 floor-plan classes, %d buildings each" % (B_TOTAL // 3),
    "zone_updates_per_s": zone_updates / elapsed, "env_steps_per_s": sum(e.sim.B for _, e, _ in envs) * K / elapsed,
    "ms_per_round": elapsed / K * 1e3,
    "classes": {name: {"grid": list(env.sim.plan.shape), "zones": env.sim.Z,
                       "kernel": "registers" if env.sim.launch_info["path"] == 1 else "lds",
                       "buildings_per_cu_slots": env.sim.launch_info["workgroups"] * env.sim.launch_info["waves_per_workgroup"] // max(env.sim.launch_info["waves_per_building"], 1) // 256,
                       "ms_per_step": float(np.mean([a.elapsed_time(b) for a, b in ev[name]])),
                       "mean_sweeps": float(env.info[:, 4].mean())}
                for name, env, _ in envs}}))
