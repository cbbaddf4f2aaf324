"""Dev tool: cost of the stochastic convection shuffle (p = 1, distance = 5, sim_config.gin:36-39) on the bench workload.

Prints ms per step with the shuffle off and on, the GPU time of the step's last phase (k_convect + k_post; HIP
events), and the ratio "at equal sweeps": the step with the shuffle over the same step without its k_convect
launch (shuffled air needs more Gauss-Seidel sweeps per step -- the reference's does too -- so the plain on / off
ratio mixes the kernel's cost with the physics)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sbsim_amd.environment import BatchedEnvironment
from sbsim_amd.host_inputs import StochasticConvectionSimulator
from bench import r9_plan

B = int(os.environ.get("B", 65536))
plan = r9_plan()
res = {}
for conv in (None, StochasticConvectionSimulator(1.0, 5, 17)):
  env = BatchedEnvironment(plan, B, holiday_calendar=None, collect_info=True, convection_simulator=conv)
  env.reset()
  rs = np.random.RandomState(7)
  t_init = np.clip(294.0 + rs.randn(B), 285.0, 305.0)
  env.sim.reset(temps=torch.tensor(t_init, dtype=torch.float64, device="cuda")[:, None].expand(B, 68 * 98).contiguous())
  acts = torch.rand((36, B, 2), device="cuda") * 2 - 1
  ev = []

  def step(t, timed):
    si = env.make_step_in(env.current_simulation_timestamp)
    a = (acts[t], si, env._obs, env._reward, env._info)
    env.sim.step(*a, phases=3)              # k_pre + the sweep kernel
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    env.sim.step(*a, phases=4)              # (k_convect +) k_post
    e1.record()
    env._prev_thermostat_ts = env._now
    env._now = env._now + env._step_interval
    if timed:
      ev.append((e0, e1))

  for t in range(12):
    step(t, False)
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  sw = 0.0
  for t in range(12, 36):
    step(t, True)
    sw += float(env.info[:, 4].mean())
  torch.cuda.synchronize()
  dt = (time.perf_counter() - t0) / 24
  post = float(np.mean([a.elapsed_time(b) for a, b in ev]))
  res["on" if conv else "off"] = (dt * 1e3, post, sw / 24)
  print(f"convection={'on' if conv else 'off'}: {dt*1e3:.3f} ms/step, last phase {post:.3f} ms, mean sweeps {sw/24:.2f}, B={B}")
  env.close()
on, off = res["on"], res["off"]
convect = on[1] - off[1]
print(f"k_convect {convect:.3f} ms; on / off {on[0] / off[0]:.2f}; at equal sweeps (the step with the shuffle over the same step without k_convect): {on[0] / (on[0] - convect):.2f}")
