"""Developer aid: step_stream.hip (the grid in global memory) -- sweep-kernel time and cell-sweeps/s on a
floor plan beyond one CU (299 x 401 control volumes, 126 zones) and, forced, on R9 for comparison with
k_sweep_roll.  Usage (GPU box): python tools/bench_stream.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from sbsim_amd.environment import BatchedEnvironment  # noqa: E402
from sbsim_amd.floorplan import FloorPlan, Materials, rectangular_floor_plan  # noqa: E402

dev = torch.device("cuda", 0)
K = int(os.environ.get("K", "6"))
CASES = [("299x401 / 126 zones", (14, 9), (20, 43), int(os.environ.get("B_BIG", "1024")), False),
         ("299x401 / 126 zones", (14, 9), (20, 43), 3 * int(os.environ.get("B_BIG", "1024")), False),   # a multiple of the resident workgroups (768)
         ("150x118 / 35 zones, forced", (7, 5), (20, 22), 4096, True),
         ("R9 68x98 / 9 zones, forced", (3, 3), (20, 30), 16384, True),
         ("R9 68x98 / 9 zones, k_sweep_roll", (3, 3), (20, 30), 16384, False)]
for name, rooms, shape, B, force in CASES:
  if force:
    os.environ["SBSIM_FORCE_STREAM_PATH"] = "1"
  else:
    os.environ.pop("SBSIM_FORCE_STREAM_PATH", None)
  plan = FloorPlan.from_file_input(rectangular_floor_plan(rooms, shape), Materials.sb1(), 10.0, 300.0)
  env = BatchedEnvironment(plan, B, device=0, holiday_calendar="us", collect_info=True, num_days_in_episode=3)
  rs = np.random.RandomState(7)
  H, W = plan.shape
  t_init = torch.tensor(np.clip(294.0 + rs.randn(B), 285.0, 305.0), dtype=torch.float64, device=dev)
  env.reset()
  env.sim.reset(temps=t_init[:, None].expand(B, H * W).contiguous())
  gen = torch.Generator(device=dev)
  gen.manual_seed(1234)
  acts = torch.rand((K + 3, B, 2), generator=gen, device=dev, dtype=torch.float32) * 2.0 - 1.0
  ev, sweeps = [], []
  for t in range(K + 3):
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
    if t >= 3:
      ev.append((e0, e1))
      sweeps.append(float(env._info[:, 4].mean()))
  torch.cuda.synchronize()
  ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
  li = env.sim.launch_info
  cells = H * W
  print(f"{name}: B={B}, kernel {li['kernel']} ({li['waves_per_building']} wavefronts per building, {li['workgroups']} workgroups, "
        f"{li['lds_bytes_per_workgroup']} B LDS), sweep kernel {ms:.2f} ms/step, mean sweeps {np.mean(sweeps):.2f}, "
        f"{B * cells * np.mean(sweeps) / (ms * 1e-3):.3e} cell-sweeps/s, {B * cells * np.mean(sweeps) * 28 / (ms * 1e-3) / 1e12:.2f} TB/s at 28 B per cell-sweep")
  if os.environ.get("SBSIM_PHASE_TIMING") and li["kernel"] == 6:
    import ctypes
    from sbsim_amd import _ffi
    buf = (ctypes.c_longlong * 16)()
    if _ffi.load().sb_debug_phase_cycles(env.sim._h, buf) == 0 and buf[0]:
      print(f"    step_stream_ms.hip: {buf[1] / buf[0]:.2f} passes, {buf[2] / buf[0]:.2f} sweep slots, {buf[3] / buf[0]:.2f} sweeps per building-step")
  env.close()
