import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
from sbsim_amd.environment import BatchedEnvironment, SimConfig
from bench import MIXED_CLASSES, r9_plan
from sbsim_amd.floorplan import FloorPlan, Materials, rectangular_floor_plan
B, T = int(os.environ.get("B", 4096)), int(os.environ.get("T", 900))
plan = r9_plan()
if os.environ.get("PLAN"):   # PLAN=SB2-synth / SB1-synth: the two-rows-per-lane kernel against the LDS-grid kernel
    _, rooms, shape = next(c for c in MIXED_CLASSES if c[0] == os.environ["PLAN"])
    plan = FloorPlan.from_file_input(rectangular_floor_plan(rooms, shape), Materials.sb1(), 10.0, 300.0)
if os.environ.get("PLAN_ROOMS"):   # PLAN_ROOMS=10,4,19,20: rooms (rows, columns) of (height, width) -- e.g. step_band.hip's 205 x 89 plan
    r0, r1, h, w = (int(v) for v in os.environ["PLAN_ROOMS"].split(","))
    plan = FloorPlan.from_file_input(rectangular_floor_plan((r0, r1), (h, w)), Materials.sb1(), 10.0, 300.0)
envs = []
# FIRST=band / stream: that kernel (step_band.hip / step_stream.hip) instead of the library's choice, against the LDS-grid kernel
first = {"band": "SBSIM_BAND_PATH", "stream": "SBSIM_FORCE_STREAM_PATH"}.get(os.environ.get("FIRST", ""))
for force in (False, True):
    if force: os.environ["SBSIM_FORCE_LDS_PATH"] = "1"
    if first and not force: os.environ[first] = "1"
    e = BatchedEnvironment(plan, B, holiday_calendar="us", collect_info=True, num_days_in_episode=2)
    if first: os.environ.pop(first, None)
    e.reset(); envs.append(e)
gen = torch.Generator(device="cuda"); gen.manual_seed(3)
mism = 0; worst = 0.0
for t in range(T):
    a = torch.rand((B, 2), generator=gen, device="cuda") * 2 - 1
    outs = [e.step(a) for e in envs]
    if outs[0].step_type[0].item() == 0: continue   # reset step
    mism += int((envs[0].info[:, 4] != envs[1].info[:, 4]).sum())
    worst = max(worst, float((envs[0].sim.zone_temps() - envs[1].sim.zone_temps()).abs().max()))
print("steps", T, "buildings", B, "sweep-count mismatches", mism, "max |dT_zone| between kernels", worst, "kernels", envs[0].sim.launch_info["kernel"], envs[1].sim.launch_info["kernel"])
