import os, sys, time, numpy as np, torch
sys.path.insert(0, os.getcwd())
from sbsim_amd.environment import BatchedSimulator, SimConfig
from bench import r9_plan
import tests.test_gpu_parity as tg
from tests.golden_util import load
g = load("h2_sb1_r9_random.npz")
B=65536
sim = BatchedSimulator(r9_plan(), SimConfig.sb1(), B, 100.0, orientation="columns")
print(sim.launch_info)
rs=np.random.RandomState(7)
t=torch.tensor(np.clip(294+rs.randn(B),285,305),dtype=torch.float64,device="cuda")[:,None].expand(B,68*98).contiguous()
sim.reset(temps=t)
obs=torch.zeros((B,sim.O),dtype=torch.float32,device="cuda"); rew=torch.zeros((B,),dtype=torch.float32,device="cuda"); info=torch.zeros((B,8),dtype=torch.float32,device="cuda")
acts=torch.rand((40,B,2),device="cuda")*2-1
for k in range(12): sim.step(acts[k], tg._step_in(g, 100+k), obs, rew, info)
torch.cuda.synchronize(); t0=time.perf_counter()
for k in range(12,36): sim.step(acts[k], tg._step_in(g, 100+k), obs, rew, info)
torch.cuda.synchronize(); print("pair mode ms/step", (time.perf_counter()-t0)/24*1e3, "sweeps", float(info[:,4].mean()))
