"""Dev tool: A/B timing of library builds (tools/build_variant.sh) on the R9 bench batch, interleaved rounds.

    python tools/exp_ab.py product desync desync_nosr ...      (ROUNDS=3 STEPS=60; "product" = the library itself)
Each (round, variant) is its own process (SBSIM_LIB is read at import); prints ms per env step per round and
the minimum.  With -DSB_EXP_DESYNC builds the sweep counts are fixed per building, so builds whose numbers are
wrong are still comparable."""
import os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, time, torch
sys.path.insert(0, %r)
from sbsim_amd.environment import BatchedEnvironment, SimConfig
from bench import r9_plan
B, steps = int(os.environ.get("B", 65536)), int(os.environ.get("STEPS", 60))
env = BatchedEnvironment(r9_plan(), B, holiday_calendar=None, collect_info=True)
env.reset()
gen = torch.Generator(device="cuda"); gen.manual_seed(5)
t0 = (294.0 + torch.randn((B, 1), generator=gen, device="cuda", dtype=torch.float64)).clamp(285.0, 305.0)
env.sim.reset(temps=t0.expand(B, 68 * 98).contiguous())
acts = torch.rand((steps + 30, B, 2), generator=gen, device="cuda") * 2 - 1
for t in range(30): env.step(acts[t])
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record(); sw = 0.0
for t in range(30, 30 + steps): env.step(acts[t])
b.record(); torch.cuda.synchronize()
print("RESULT %%.4f %%.3f" %% (a.elapsed_time(b) / steps, float(env.info[:, 4].double().mean())))
''' % ROOT


def main():
  variants = sys.argv[1:] or ["product"]
  rounds = int(os.environ.get("ROUNDS", 3))
  res = {v: [] for v in variants}
  for r in range(rounds):
    for v in variants:
      env = dict(os.environ)
      name, _, flag = v.partition("+")          # "desync+exact": SBSIM_ROLL_EXACT=1 on that build
      if name != "product": env["SBSIM_LIB"] = os.path.join(ROOT, "tools", f"libexp_{name}.so")
      if flag == "exact": env["SBSIM_ROLL_EXACT"] = "1"
      out = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True).stdout
      line = [l for l in out.splitlines() if l.startswith("RESULT")]
      if line:
        ms, sw = line[0].split()[1:]
        res[v].append((float(ms), float(sw)))
  for v in variants:
    ms = [m for m, _ in res[v]]
    print(f"{v:28s} " + " ".join(f"{m:.4f}" for m in ms) + (f"   min {min(ms):.4f} ms/step, sweeps {res[v][-1][1]:.2f}" if ms else "   FAILED"), flush=True)


main()
