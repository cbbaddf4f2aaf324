"""Developer aid: k_sweep_roll's two regimes.  The bench workload (65,536 R9 buildings, random
actions) with the sweep count capped by `iteration_limit` = 1, 2, 3, ...: the sweep kernel's time per dispatch (HIP
events around phase 2 of sb_step), the real HBM rate it sustains (state read once + written once per launch,
`state_bytes_per_env_step`) and the roofline fraction on the algorithmic bytes.  HBM-bound at <= 2 sweeps per step,
instruction-issue-bound above.  -> profiles/r07_roll_regimes.txt"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from bench import r9_plan  # noqa: E402
from sbsim_amd.environment import BatchedEnvironment, SimConfig  # noqa: E402

B, K = int(os.environ.get("B", "65536")), int(os.environ.get("K", "12"))
plan = r9_plan()
print("iteration_limit  mean sweeps  sweep kernel ms (min / mean)  real HBM TB/s (min-time / mean)  roofline frac (mean; 8 TB/s)")
for lim in [int(x) for x in os.environ.get("LIMS", "1,2,3,4,5,6,8,100").split(",")]:
  env = BatchedEnvironment(plan, B, config=SimConfig(iteration_limit=lim), holiday_calendar=None, collect_info=True)
  env.reset()
  rs = np.random.RandomState(7)
  t_init = np.clip(294.0 + rs.randn(B), 285.0, 305.0)
  env.sim.reset(temps=torch.tensor(t_init, dtype=torch.float64, device="cuda")[:, None].expand(B, 68 * 98).contiguous())
  gen = torch.Generator(device="cuda")
  gen.manual_seed(1234)
  acts = torch.rand((K + 30, B, 2), generator=gen, device="cuda", dtype=torch.float32) * 2.0 - 1.0
  ev, sw = [], []
  for t in range(K + 30):
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
    if t >= 30:
      ev.append((e0, e1))
      sw.append(float(env._info[:, 4].mean()))
  torch.cuda.synchronize()
  ms = np.array([a.elapsed_time(b) for a, b in ev])
  li = env.sim.launch_info
  real, alg = li["state_bytes_per_env_step"] * B, li["algorithmic_bytes_per_env_step"] * B
  print(f"{lim:15d}  {np.mean(sw):11.2f}  {ms.min():8.3f} / {ms.mean():8.3f}          {real / ms.min() / 1e9:6.2f} / {real / ms.mean() / 1e9:6.2f}"
        f"                  {alg / ms.mean() / 1e9 / 8.0:6.3f}")
  env.close()
