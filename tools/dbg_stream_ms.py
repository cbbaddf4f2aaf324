"""Developer aid: step_stream_ms.hip against step_stream.hip (SBSIM_STREAM_SINGLE=1) on one plan, iteration limits 1..L:
the first limit at which the grids differ tells which sweep of a pass goes wrong.   ROOMS=2,2 SHAPE=5,9 LIMS=1,2,3,4,5"""
import os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, numpy as np, torch
sys.path.insert(0, %r)
from sbsim_amd.environment import BatchedSimulator, SimConfig
from sbsim_amd.floorplan import FloorPlan, Materials, rectangular_floor_plan
from sbsim_amd import _ffi
rooms = tuple(int(x) for x in os.environ.get("ROOMS", "2,2").split(","))
shape = tuple(int(x) for x in os.environ.get("SHAPE", "5,9").split(","))
lim = int(os.environ["LIM"])
plan = FloorPlan.from_file_input(rectangular_floor_plan(rooms, shape), Materials.sb1(), 10.0, 300.0)
cfg = SimConfig.sb1(); cfg.iteration_limit = lim
B = int(os.environ.get("NB", "2"))
sim = BatchedSimulator(plan, cfg, B, 100.0)
rs = np.random.RandomState(3)
H, W = plan.shape
init = 294.0 + rs.rand(B, H * W)
sim.reset(temps=torch.tensor(init, dtype=torch.float64, device="cuda"))
si = _ffi.StepIn(); si.t_amb_now = si.t_amb_next = 285.0; si.comfort_now = si.comfort_next = 1; si.comfort_prev = -1
si.has_action = 0; si.occupancy = 1.0; si.e_price = si.e_carbon = si.g_price = si.g_carbon = 1e-8
rew = torch.zeros((B,), dtype=torch.float32, device="cuda"); info = torch.zeros((B, _ffi.SB_INFO_STRIDE), dtype=torch.float32, device="cuda")
for t in range(int(os.environ.get("STEPS", "1"))):
  sim.step(None, si, None, rew, info)
np.save(os.environ["OUT"], sim.temps().cpu().numpy())
print("INFO", sim.launch_info["kernel"], sim.launch_info["waves_per_building"], info[:, 4].cpu().numpy().tolist())
''' % ROOT
for lim in [int(x) for x in os.environ.get("LIMS", "1,2,3,4,5,6,9").split(",")]:
  out = {}
  for tag, env_extra in (("single", {}), ("ms", {"SBSIM_STREAM_MS": "1"})):
    env = dict(os.environ); env.update(env_extra); env["SBSIM_FORCE_STREAM_PATH"] = "1"; env["LIM"] = str(lim); env["OUT"] = f"/tmp/dbg_{tag}.npy"
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=300)
    line = [l for l in r.stdout.splitlines() if l.startswith("INFO")]
    out[tag] = (np.load(f"/tmp/dbg_{tag}.npy") if line else None, line[0] if line else r.stderr[-400:])
  a, b = out["single"][0], out["ms"][0]
  if a is None or b is None:
    print(lim, "FAILED", out["single"][1], out["ms"][1]); continue
  d = np.abs(a - b)
  H = a.shape[1]
  bad = np.argwhere(d[0] > 1e-9)
  print(f"limit {lim}: max |diff| {d.max():.3e}; {out['single'][1]} | {out['ms'][1]}; differing cells of building 0: {len(bad)}"
        + (f", rows {bad[:,0].min()}..{bad[:,0].max()}, cols {bad[:,1].min()}..{bad[:,1].max()}, first {bad[:6].tolist()}" if len(bad) else ""))
  if len(bad) and os.environ.get("SHOW"):
    r0, c0 = bad[0]
    np.set_printoptions(linewidth=250, precision=3, suppress=True)
    print("single:\n", a[0][r0:r0+4, c0-2:c0+8]); print("ms:\n", b[0][r0:r0+4, c0-2:c0+8])
    m = (d[0] > 1e-9).astype(int)
    for row in m[:min(H, 70)]: print("".join(".#"[v] for v in row))
