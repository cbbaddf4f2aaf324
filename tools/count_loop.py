"""Developer tool: instruction mix of the largest loop (the unrolled period) of every kernel in a gfx950
assembly listing (`hipcc ... --cuda-device-only -S file.hip -o file.s`).

    python tools/count_loop.py /tmp/step_roll.s [name-substring]
"""
import collections
import re
import sys


def classify(ins: str) -> str:
  op = ins.split()[0]
  if op.startswith(("v_fma", "v_mul_f64", "v_add_f64", "v_max_f64", "v_min_f64")): return "fp64"
  if "dpp" in ins: return "dpp"
  if "sdwa" in ins: return "sdwa"
  if op.startswith("ds_"): return "lds"
  if op.startswith("v_accvgpr"): return "acc"
  if op.startswith("s_waitcnt"): return "wait"
  if op.startswith("s_nop"): return "nop"
  if op.startswith(("global_", "buffer_", "flat_", "scratch_")): return "vmem"
  if op.startswith("s_"): return "salu"
  if op.startswith("v_"): return "valu32"
  return "other"


def main():
  txt = open(sys.argv[1]).read()
  want = sys.argv[2] if len(sys.argv) > 2 else ""
  for f in re.split(r"\n(?=_Z\w+:)", txt)[1:]:
    name = f.split(":")[0]
    if want not in name or ".amdhsa_kernel" not in f: continue
    labels, instrs = {}, []
    for l in f.split("\n"):
      m = re.match(r"^(\.LBB\d+_\d+):", l)
      if m: labels[m.group(1)] = len(instrs)
      elif l.startswith("\t") and not l.startswith("\t.") and not l.strip().startswith(";"): instrs.append(l.strip())
      if l.startswith(".Lfunc_end"): break
    best = None
    for i, ins in enumerate(instrs):
      m = re.match(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", ins)
      if m and m.group(1) in labels and labels[m.group(1)] < i:
        n = i - labels[m.group(1)]
        if best is None or n > best[0]: best = (n, labels[m.group(1)], i)
    tot = collections.Counter(classify(i) for i in instrs)
    print(name, "instructions", len(instrs), dict(tot))
    if best:
      n, a, b = best
      print("  largest loop:", n + 1, dict(collections.Counter(classify(i) for i in instrs[a:b + 1])))


main()
