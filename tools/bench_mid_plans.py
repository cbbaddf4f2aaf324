"""Developer aid: the floor plans step_band.hip runs by the library's choice -- 131..258 rows (three / four wavefronts
per building) and 67..130 rows with 81..96 columns (two) -- against the kernel such a plan ran on before
(SBSIM_NO_BAND_PATH=1: step_lds.hip, step_stream.hip or k_sweep_reg on two wavefronts): sweep-kernel time per step
and cell-sweeps/s.  Usage (GPU box): python tools/bench_mid_plans.py"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from sbsim_amd import _ffi  # noqa: E402
from sbsim_amd.environment import BatchedEnvironment  # noqa: E402
from sbsim_amd.floorplan import FloorPlan, Materials, rectangular_floor_plan  # noqa: E402

dev = torch.device("cuda", 0)
K = int(os.environ.get("K", "8"))
WARM = int(os.environ.get("WARM", "3"))   # steps before the timed ones (the sweep count per step grows over the first ~30)
B = int(os.environ.get("B", "4096"))
ONLY = os.environ.get("PLANS")   # e.g. PLANS=0,1
PLANS = [("205x89 / 40 zones", (10, 4), (19, 20)), ("195x89 / 40 zones", (10, 4), (18, 20)), ("158x77 / 36 zones", (9, 4), (16, 17)),
         ("257x80 / 36 zones", (12, 3), (20, 24)),
         # 67..130 rows, 81..96 columns: two wavefronts per building (before: k_sweep_reg on two wavefronts / the LDS-grid kernel)
         ("113x93 / 24 zones", (6, 4), (17, 21)), ("125x97 / 20 zones", (5, 4), (23, 22)), ("109x92 / 12 zones", (4, 3), (25, 28))]
for name, rooms, shape in [p for i, p in enumerate(PLANS) if ONLY is None or str(i) in ONLY.split(",")]:
  rates = {}
  for label, flag in (("step_band", None), ("before", "SBSIM_NO_BAND_PATH"))[:1 if os.environ.get("BAND_ONLY") else 2]:
    if flag:
      os.environ[flag] = "1"
    plan = FloorPlan.from_file_input(rectangular_floor_plan(rooms, shape), Materials.sb1(), 10.0, 300.0)
    env = BatchedEnvironment(plan, B, device=0, holiday_calendar="us", collect_info=True, num_days_in_episode=3)
    if flag:
      os.environ.pop(flag)
    rs = np.random.RandomState(7)
    H, W = plan.shape
    t_init = torch.tensor(np.clip(294.0 + rs.randn(B), 285.0, 305.0), dtype=torch.float64, device=dev)
    env.reset()
    env.sim.reset(temps=t_init[:, None].expand(B, H * W).contiguous())
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234)
    acts = torch.rand((K + WARM, B, 2), generator=gen, device=dev, dtype=torch.float32) * 2.0 - 1.0
    ev, sweeps = [], []
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
        sweeps.append(env._info[:, 4].double().cpu().numpy().copy())
    torch.cuda.synchronize()
    ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    li = env.sim.launch_info
    msw = float(np.mean(sweeps))
    rates[label] = (B * H * W * msw / (ms * 1e-3), np.stack(sweeps))
    print(f"{name}: {label}: kernel {li['kernel']}, {li['waves_per_building']} wavefronts per building, {li['workgroups']} workgroups, "
          f"{li['lds_bytes_per_workgroup']} B LDS; B={B}: sweep kernel {ms:.2f} ms/step, mean sweeps {msw:.2f}, "
          f"{rates[label][0]:.3e} cell-sweeps/s", flush=True)
    if os.environ.get("SBSIM_PHASE_TIMING") and li["kernel"] == 5:   # a tools/build_variant.sh ... -DSB_PHASE_STAMPS build
      buf = (ctypes.c_longlong * 16)()
      _ffi.check(_ffi.load().sb_debug_phase_cycles(env.sim._h, buf), "dbg")
      d, n = list(buf), B * (K + WARM)
      print(f"  per building-step: blocks {d[13] / n:.2f} + single sweeps {d[14] / n:.2f}, blocks run again {d[15] / n:.3f}; spins (64 cycles each) "
            f"for the wavefront above {d[6] / n:.0f}, below {d[7] / n:.0f}, for a sweep's max|delta| {d[8] / n:.0f}; cycles from a building's start: "
            f"{[d[i] - d[0] for i in (1, 2, 3, 4, 5)]} (A pass done, sweeps done, handed over, reduced), n_sweeps {d[9]}; "
            f"wavefront 0's rolling periods: {d[10] / max(d[11], 1):.0f} cycles each, {d[11] / n:.2f} per building-step", flush=True)
    env.close()
  if os.environ.get("BAND_ONLY"):
    continue
  same = bool((rates["step_band"][1] == rates["before"][1]).all())
  print(f"{name}: step_band / before = {rates['step_band'][0] / rates['before'][0]:.2f}x; sweep counts of all {B} buildings x {K} steps equal: {same}", flush=True)
