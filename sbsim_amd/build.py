"""Builds the HIP shared library in-tree (``sbsim_amd/libsbsim_amd.so``) for gfx950.

hipcc cross-compiles without a GPU, so this runs in the build container; the resulting
``.so`` travels to the GPU box with the repo snapshot (it is git-ignored, not
gpurun-ignored)."""
from __future__ import annotations

import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_HERE)
SRC = os.path.join(_HERE, "csrc", "sbsim_hip.hip")
LIB = os.path.join(_HERE, "libsbsim_amd.so")
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-honor-nans", "-fPIC", "-shared"]


def hipcc() -> str:
  for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
    if cand and os.path.exists(cand):
      return cand
  raise RuntimeError("hipcc not found (expected /opt/rocm/bin/hipcc)")


def build(force: bool = False, verbose: bool = False) -> str:
  deps = [SRC, os.path.join(ROOT, "include", "sbsim_amd.h")]
  if (not force and os.path.exists(LIB)
      and all(os.path.getmtime(LIB) >= os.path.getmtime(d) for d in deps)):
    return LIB
  cmd = [hipcc(), *HIPCC_FLAGS, "-I", os.path.join(ROOT, "include"), SRC, "-o", LIB]
  if verbose:
    cmd.append("-Rpass-analysis=kernel-resource-usage")
  subprocess.run(cmd, check=True)
  return LIB


if __name__ == "__main__":
  print(build(force=True, verbose=True))
