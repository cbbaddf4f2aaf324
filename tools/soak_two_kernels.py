import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
from sbsim_amd.environment import BatchedEnvironment, SimConfig
from bench import r9_plan
B, T = int(os.environ.get("B", 4096)), int(os.environ.get("T", 900))
plan = r9_plan()
envs = []
for force in (False, True):
    if force: os.environ["SBSIM_FORCE_LDS_PATH"] = "1"
    e = BatchedEnvironment(plan, B, holiday_calendar="us", collect_info=True, num_days_in_episode=2)
    e.reset(); envs.append(e)
gen = torch.Generator(device="cuda"); gen.manual_seed(3)
mism = 0; worst = 0.0
for t in range(T):
    a = torch.rand((B, 2), generator=gen, device="cuda") * 2 - 1
    outs = [e.step(a) for e in envs]
    if outs[0].step_type[0].item() == 0: continue   # reset step
    mism += int((envs[0].info[:, 4] != envs[1].info[:, 4]).sum())
    worst = max(worst, float((envs[0].sim.zone_temps() - envs[1].sim.zone_temps()).abs().max()))
print("steps", T, "buildings", B, "sweep-count mismatches", mism, "max |dT_zone| between kernels", worst, "paths", envs[0].sim.launch_info["path"], envs[1].sim.launch_info["path"])
