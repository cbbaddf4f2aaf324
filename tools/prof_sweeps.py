"""Dev tool: per-sweep and fixed cost of a step (caps the sweep count via iteration_limit)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sbsim_amd.environment import BatchedEnvironment, SimConfig
from bench import r9_plan

B = int(os.environ.get("B", 65536))
plan = r9_plan()
for lim in [int(x) for x in os.environ.get("LIMS", "1,2,4,8").split(",")]:
  env = BatchedEnvironment(plan, B, config=SimConfig(iteration_limit=lim), holiday_calendar=None, collect_info=True)
  env.reset()
  rs = np.random.RandomState(7)
  t_init = np.clip(294.0 + rs.randn(B), 285.0, 305.0)
  env.sim.reset(temps=torch.tensor(t_init, dtype=torch.float64, device="cuda")[:, None].expand(B, 68 * 98).contiguous())
  acts = torch.rand((20, B, 2), device="cuda") * 2 - 1
  for t in range(4):
    env.step(acts[t])
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  sw = 0.0
  for t in range(4, 16):
    env.step(acts[t])
    sw += float(env.info[:, 4].mean())
  torch.cuda.synchronize()
  dt = (time.perf_counter() - t0) / 12
  print(f"iter_limit={lim}: {dt*1e3:.3f} ms/step, mean sweeps {sw/12:.2f}, B={B}")
  env.close()

if os.environ.get("SBSIM_PHASE_TIMING"):
  import ctypes as C
  from sbsim_amd import _ffi
  env = BatchedEnvironment(plan, B, holiday_calendar=None, collect_info=True)
  env.reset()
  env.sim.reset(temps=torch.tensor(t_init, dtype=torch.float64, device="cuda")[:, None].expand(B, 68 * 98).contiguous())
  names = ["setup(g,tail,wait rows)", "A pass", "sweeps", "-", "store+scatter", "next-row loads", "gsum", "zone reduce"]
  for t in range(8):
    env.step(acts[t])
    buf = (C.c_longlong * 16)()
    _ffi.check(_ffi.load().sb_debug_phase_cycles(env.sim._h, buf), "dbg")
    d = [buf[i + 1] - buf[i] for i in range(8)]
    print(f"step {t}: sweeps={buf[9]} " + " ".join(f"{n}={v}" for n, v in zip(names, d)) + f" total={buf[8]-buf[0]}"
          + (f" ramp-up={buf[15]-buf[2]}" if buf[15] > buf[2] and not os.environ.get("SB_STAMP_NEXT") else ""))
    if buf[9] >= 2 and buf[11] > buf[10]:   # the second sweep's period (k_sweep_roll): 96 steps | publish row 63 | tail scan | decision
      print(f"        second sweep: steps={buf[11]-buf[10]} publish={buf[12]-buf[11]} tail={buf[13]-buf[12]} decide={buf[14]-buf[13]}"
            + (f" back edge (to the next period's top; -DSB_STAMP_NEXT builds)={buf[15]-buf[14]}" if os.environ.get("SB_STAMP_NEXT") and buf[15] > buf[14] else ""))
