"""Developer aid: would a two-wavefront-per-SIMD column split of k_sweep_roll pay on
the bench workload?  Takes the MEASURED Gauss-Seidel sweep counts of the bench rollout (65,536... here
B buildings of R9, random setpoints) and the MEASURED step costs of tools/proto/split_proto.hip, and
prices three schedules per building-step, in SIMD cycles per building:

  today      one wavefront, 96 slots, sweeps overlapped with copies: 63 + 96 n steps
  split-ideal  two wavefronts x 48 slots, 2 per SIMD, a block of exactly n sweeps known in advance:
               (63 + 48 lag + 48 n) steps per wavefront
  split-real   the same with what can be known: lane 0 starts sweep j while the verdicts of sweeps
               j-3 .. j-1 are still out (a sweep spans 159 steps, a period is 48), so a step is one
               blind block of m = n_prev sweeps (the building's previous count); n > m: the rest as
               single sweeps (159 steps each); n < m: the overrun is found when sweep n's verdict arrives
               and the step is replayed with the right count (the copy-free recovery of step_two.hip)

Usage (GPU box): python tools/split_schedule_model.py [cycles_today cycles_split]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
from sbsim_amd.environment import BatchedEnvironment  # noqa: E402

C_TODAY = float(sys.argv[1]) if len(sys.argv) > 1 else 101.0   # cycles per step, one wavefront per SIMD (k_sweep_roll, stamps)
C_SPLIT = float(sys.argv[2]) if len(sys.argv) > 2 else 78.2    # SIMD cycles per wavefront step, two per SIMD (split_proto)
B = int(os.environ.get("B", 8192))
dev = torch.device("cuda", 0)
plan = bench.r9_plan()
env = BatchedEnvironment(plan, B, device=0, holiday_calendar=None, collect_info=True, num_days_in_episode=3)
rs = np.random.RandomState(7)
H, W = plan.shape
t_init = torch.tensor(np.clip(294.0 + rs.randn(B), 285.0, 305.0), dtype=torch.float64, device=dev)
env.reset()
env.sim.reset(temps=t_init[:, None].expand(B, H * W).contiguous())
gen = torch.Generator(device=dev)
gen.manual_seed(1234)
counts = []
for t in range(140):
  a = torch.rand((B, 2), generator=gen, device=dev, dtype=torch.float32) * 2.0 - 1.0
  env.step(a)
  counts.append(env._info[:, 4].cpu().numpy().astype(int))
counts = np.stack(counts)   # [T][B]


def price(n, n_prev):
  today = (63 + 96 * n) * C_TODAY
  ramp = 63 + 48
  ideal = 2 * (ramp + 48 * n) * C_SPLIT
  m = np.maximum(n_prev, 1)
  under = 2 * ((ramp + 48 * m) + (n - m) * (ramp + 48)) * C_SPLIT          # n >= m
  over = 2 * ((48 * n + 170) + (ramp + 48 * n)) * C_SPLIT                   # n < m: found out, replayed
  real = np.where(n >= m, under, over)
  return today, ideal, real


for name, lo, hi in (("bench window (steps 5..24)", 5, 25), ("steady state (steps 100..139)", 100, 140)):
  n, p = counts[lo:hi], counts[lo - 1:hi - 1]
  vals, cnt = np.unique(n, return_counts=True)
  print(f"{name}: mean sweeps {n.mean():.2f}; histogram " + " ".join(f"{v}:{c / n.size:.3f}" for v, c in zip(vals, cnt) if c / n.size >= 0.002))
  d = n - p
  print(f"  n_t - n_(t-1): ==0 {np.mean(d == 0):.3f}  <0 {np.mean(d < 0):.3f}  >0 {np.mean(d > 0):.3f}  mean |d| {np.abs(d).mean():.2f}")
  today, ideal, real = price(n, p)
  print(f"  SIMD cycles per building-step, sweeps only: today {today.mean():.0f}  split-ideal {ideal.mean():.0f} "
        f"({today.mean() / ideal.mean():.2f}x)  split-real {real.mean():.0f} ({today.mean() / real.mean():.2f}x)")
env.close()
