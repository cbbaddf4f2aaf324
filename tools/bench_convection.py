"""Dev tool: cost of the stochastic convection shuffle (p = 1, distance = 5) on the bench workload."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sbsim_amd.environment import BatchedEnvironment
from sbsim_amd.host_inputs import StochasticConvectionSimulator
from bench import r9_plan

B = int(os.environ.get("B", 65536))
plan = r9_plan()
for conv in (None, StochasticConvectionSimulator(1.0, 5, 17)):
  env = BatchedEnvironment(plan, B, holiday_calendar=None, collect_info=True, convection_simulator=conv)
  env.reset()
  rs = np.random.RandomState(7)
  t_init = np.clip(294.0 + rs.randn(B), 285.0, 305.0)
  env.sim.reset(temps=torch.tensor(t_init, dtype=torch.float64, device="cuda")[:, None].expand(B, 68 * 98).contiguous())
  acts = torch.rand((36, B, 2), device="cuda") * 2 - 1
  for t in range(12):
    env.step(acts[t])
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  sw = 0.0
  for t in range(12, 36):
    env.step(acts[t])
    sw += float(env.info[:, 4].mean())
  torch.cuda.synchronize()
  dt = (time.perf_counter() - t0) / 24
  print(f"convection={'on' if conv else 'off'}: {dt*1e3:.3f} ms/step, mean sweeps {sw/24:.2f}, B={B}")
  env.close()
