import os, sys, time, torch
sys.path.insert(0, "/root/repo")
from sbsim_amd.environment import BatchedEnvironment
from bench import r9_plan
B, steps = 65536, 40
for name, orient in (("rows (k_sweep_roll)", "rows"), ("columns (k_sweep_two)", "columns")):
  os.environ["SBSIM_ORIENTATION"] = orient
  plan = r9_plan()
  env = BatchedEnvironment(plan, B, holiday_calendar=None, collect_info=True)
  env.reset()
  gen = torch.Generator(device="cuda"); gen.manual_seed(5)
  t0 = (294.0 + torch.randn((B, 1), generator=gen, device="cuda", dtype=torch.float64)).clamp(285.0, 305.0)
  env.sim.reset(temps=t0.expand(B, 68 * 98).contiguous())
  acts = torch.rand((steps + 30, B, 2), generator=gen, device="cuda") * 2 - 1
  for t in range(30): env.step(acts[t])
  torch.cuda.synchronize()
  a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  a.record()
  for t in range(30, 30 + steps): env.step(acts[t])
  b.record(); torch.cuda.synchronize()
  print(name, "%.4f ms/step" % (a.elapsed_time(b) / steps), "sweeps %.2f" % float(env.info[:, 4].double().mean()), env.sim.launch_info["kernel"], flush=True)
  env.close()
