"""Developer aid: a time line of ONE building-step of step_band.hip's workgroup 0 (every wavefront's sync points, period
ends, decisions), from a -DSB_PHASE_STAMPS build: SBSIM_LIB=tools/libexp_stamps.so python tools/band_timeline.py [plan index]"""
import ctypes
import os
import sys

import numpy as np
import torch

os.environ["SBSIM_PHASE_TIMING"] = "1"
os.environ["SBSIM_DEBUG_TIMELINE"] = "1"
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from sbsim_amd import _ffi  # noqa: E402
from sbsim_amd.environment import BatchedEnvironment  # noqa: E402
from sbsim_amd.floorplan import FloorPlan, Materials, rectangular_floor_plan  # noqa: E402

PLANS = [((10, 4), (19, 20)), ((10, 4), (18, 20)), ((9, 4), (16, 17)), ((12, 3), (20, 24))]
rooms, shape = PLANS[int(sys.argv[1]) if len(sys.argv) > 1 else 0]
B = int(os.environ.get("B", "1024"))
dev = torch.device("cuda", 0)
plan = FloorPlan.from_file_input(rectangular_floor_plan(rooms, shape), Materials.sb1(), 10.0, 300.0)
env = BatchedEnvironment(plan, B, device=0, holiday_calendar="us", collect_info=True, num_days_in_episode=3)
H, W = plan.shape
rs = np.random.RandomState(7)
t_init = torch.tensor(np.clip(294.0 + rs.randn(B), 285.0, 305.0), dtype=torch.float64, device=dev)
env.reset()
env.sim.reset(temps=t_init[:, None].expand(B, H * W).contiguous())
gen = torch.Generator(device=dev)
gen.manual_seed(1234)
STEPS = int(os.environ.get("STEPS", "6"))
for t in range(STEPS):
  ts = env.step(torch.rand((B, 2), generator=gen, device=dev, dtype=torch.float32) * 2.0 - 1.0)
torch.cuda.synchronize()
nw = env.sim.launch_info["waves_per_building"]
buf = (ctypes.c_longlong * (16 + 2048))()
_ffi.check(_ffi.load().sb_debug_phase_cycles(env.sim._h, buf), "dbg")
d = np.array(list(buf), dtype=np.int64)
print("sweeps of that building-step (last step):", d[9], "waves", nw)
names = {0xf01: "block start", 0xf02: "decision", 0xf03: "period steps done", 0xf04: "part published", 0xf05: "final steps done", 0xf06: "pair63", 0xf07: "pre-read", 0xf08: "parts read"}
t0 = None
rows = []
for w in range(nw):
  for v in d[16 + 512 * w: 16 + 512 * (w + 1)]:
    if v == 0:
      break
    cyc, code = int(v >> 12), int(v & 0xfff)
    rows.append((cyc, w, code))
t0 = min(r[0] for r in rows)
for w in range(nw):
  prev = None
  line = []
  for cyc, ww, code in rows:
    if ww != w:
      continue
    dt = cyc - (prev if prev is not None else t0)
    prev = cyc
    line.append(f"{names.get(code, 's' + str(code))}+{dt}")
  print(f"wavefront {w}:", " ".join(line))
