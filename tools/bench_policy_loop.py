"""Dev tool: BASELINE.json configs[4]-style loop on one GPU -- BatchedEnvironment.step() driven by a
SAC-shaped actor (2 x 128 MLP, tanh-squashed Gaussian) evaluated on the environment's own GPU, so
that no observation leaves HBM.  Prints env-steps/s including policy inference and the host-side
step inputs (calendar, tariffs, occupancy)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sbsim_amd.environment import BatchedEnvironment
from bench import r9_plan

B = int(os.environ.get("B", 65536))
STEPS = int(os.environ.get("STEPS", 48))
env = BatchedEnvironment(r9_plan(), B, holiday_calendar=None)
ts = env.reset()
rs = np.random.RandomState(7)
t_init = np.clip(294.0 + rs.randn(B), 285.0, 305.0)
env.sim.reset(temps=torch.tensor(t_init, dtype=torch.float64, device="cuda")[:, None].expand(B, 68 * 98).contiguous())
torch.manual_seed(0)
O = env.observation_spec().shape[0]
actor = torch.nn.Sequential(torch.nn.Linear(O, 128), torch.nn.ReLU(), torch.nn.Linear(128, 128), torch.nn.ReLU(),
                            torch.nn.Linear(128, 4)).cuda()
gen = torch.Generator(device="cuda").manual_seed(1)

@torch.no_grad()
def act(obs):
  out = actor(torch.nan_to_num(obs))
  mean, log_std = out[:, :2], out[:, 2:].clamp(-5, 2)
  return torch.tanh(mean + log_std.exp() * torch.randn(mean.shape, device="cuda", generator=gen)).contiguous()

obs = ts.observation
ret = torch.zeros((B,), device="cuda")
for _ in range(12):
  ts = env.step(act(obs)); obs = ts.observation
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(STEPS):
  ts = env.step(act(obs)); obs = ts.observation
  ret += ts.reward
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / STEPS
print(f"policy loop: {dt*1e3:.3f} ms/step, {B/dt:.3e} env-steps/s, {9*B/dt:.3e} zone-updates/s, mean return/step {float(ret.mean())/STEPS:.4f}")
env.close()
