"""Developer aid: the two-rows-per-lane kernel (step_two.hip) on one floor-plan class at a time --
sweep-kernel time, mean sweeps, blocks that overran (SBSIM_PHASE_TIMING=1), cycle stamps."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
from sbsim_amd import _ffi  # noqa: E402
from sbsim_amd.environment import BatchedEnvironment  # noqa: E402
from sbsim_amd.floorplan import FloorPlan, Materials, rectangular_floor_plan  # noqa: E402

dev = torch.device("cuda", 0)
B = int(os.environ.get("B", "21845"))
K = int(os.environ.get("K", "8"))
for name, rooms, shape in bench.MIXED_CLASSES[1:]:
  plan = FloorPlan.from_file_input(rectangular_floor_plan(rooms, shape), Materials.sb1(), 10.0, 300.0)
  env = BatchedEnvironment(plan, B, device=0, holiday_calendar="us", collect_info=True, num_days_in_episode=3)
  rs = np.random.RandomState(7)
  H, W = plan.shape
  t_init = torch.tensor(np.clip(294.0 + rs.randn(B), 285.0, 305.0), dtype=torch.float64, device=dev)
  env.reset()
  env.sim.reset(temps=t_init[:, None].expand(B, H * W).contiguous())
  gen = torch.Generator(device=dev)
  gen.manual_seed(1234)
  acts = torch.rand((K + 4, B, 2), generator=gen, device=dev, dtype=torch.float32) * 2.0 - 1.0
  ev, sweeps = [], []
  for t in range(K + 4):
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
    if t >= 4:
      ev.append((e0, e1))
      sweeps.append(float(env._info[:, 4].mean()))
  torch.cuda.synchronize()
  ms = np.mean([a.elapsed_time(b) for a, b in ev])
  line = f"{name}: sweep kernel {ms:.2f} ms/step, mean sweeps {np.mean(sweeps):.2f}, path {env.sim.launch_info['path']}, steps/sweep {env.sim.launch_info['sweep_steps']}"
  if os.environ.get("SBSIM_PHASE_TIMING"):
    buf = (ctypes.c_longlong * 16)()
    _ffi.check(_ffi.load().sb_debug_phase_cycles(env.sim._h, buf), "dbg")
    d = list(buf)
    line += f", blocks {d[13] / (B * (K + 4)):.2f} + single sweeps {d[14] / (B * (K + 4)):.2f} per building-step"
    # step_two_impl.h: measure-free / measuring rolling periods, blocks run again because a measurement came too late (step_band.hip: spins)
    line += f", periods measure-free {d[6] / (B * (K + 4)):.2f} measuring {d[7] / (B * (K + 4)):.2f} late measurements {d[8] / (B * (K + 4)):.4f} per building-step"
    line += f", overrun blocks {d[15]} ({d[15] / (B * (K + 4)):.3f} per building-step), stamps {[d[i] - d[0] for i in (1, 2, 3, 4, 5)]}, period {d[11] - d[10]} end+decision+next period {d[12] - d[11]}, n_sweeps {d[9]}"
  print(line)
  env.close() if hasattr(env, "close") else None
