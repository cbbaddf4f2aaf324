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
CSRC = os.path.join(_HERE, "csrc")
SOURCES = ["sbsim_hip.hip", "step_reg.hip", "step_roll.hip", "step_two.hip", "step_two_64.hip", "step_two_76.hip", "step_two_80.hip", "step_band.hip", "step_band_68.hip", "step_band_72.hip", "step_band_76.hip", "step_band_80.hip", "step_band_84.hip", "step_band_88.hip", "step_band_92.hip", "step_band_96.hip", "step_stream.hip", "step_lds.hip",    # one translation unit per step kernel
           "generators.hip",                                 # occupancy / convection generators
           "floorplan.cpp", "episode.cpp"]                   # host-only: floor-plan preprocessing, episode shards
HEADERS = [os.path.join(CSRC, "sb_device.h"), os.path.join(CSRC, "sb_host.h"), os.path.join(CSRC, "sweep_common.h"), os.path.join(CSRC, "step_two_impl.h"), os.path.join(CSRC, "step_two_cfg.h"), os.path.join(CSRC, "step_band_impl.h"), os.path.join(CSRC, "step_band_cfg.h"),
           os.path.join(ROOT, "include", "sbsim_amd.h")]
# SBSIM_BUILD_EXPERIMENTAL=1: also the sweep kernels that are exact and tested but slower than what they were meant to replace
# (step_stream_ms.hip: several sweeps per pass; step_stream.hip's k_sweep_stream_roll: overlapped sweeps) -- opt-in at run time
# (SBSIM_STREAM_MS=1 / SBSIM_STREAM_ROLL=1), not part of the default library
EXPERIMENTAL = os.environ.get("SBSIM_BUILD_EXPERIMENTAL", "") == "1"
if EXPERIMENTAL:
  SOURCES = SOURCES + ["step_stream_ms.hip"]
  HIPCC_FLAGS_EXTRA = ["-DSB_EXPERIMENTAL"]
else:
  HIPCC_FLAGS_EXTRA = []
OBJ_DIR = os.path.join(CSRC, "_obj" + ("_exp" if EXPERIMENTAL else ""))
LIB = os.path.join(_HERE, "libsbsim_amd_exp.so" if EXPERIMENTAL else "libsbsim_amd.so")   # (the experimental library is used through SBSIM_LIB=...)
# -amdgpu-atomic-optimizer-strategy=None: the optimizer turns the sweep kernels' one-lane draw (`if (lane == 0) atomicAdd(next_b, 1)`)
# into a wave reduction that reads the atomic's result AT ONCE -- an s_waitcnt vmcnt(0), i.e. a wait for the next building's rows
# (all in flight at that point) at the top of every building (step_roll.hip)
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-honor-nans", "-fPIC",
               "-mllvm", "-amdgpu-atomic-optimizer-strategy=None"]


def hipcc() -> str:
  for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
    if cand and os.path.exists(cand):
      return cand
  raise RuntimeError("hipcc not found (expected /opt/rocm/bin/hipcc)")


def _newer(target: str, deps) -> bool:
  return os.path.exists(target) and all(os.path.getmtime(target) >= os.path.getmtime(d) for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
  srcs = [os.path.join(CSRC, s) for s in SOURCES]
  if not force and _newer(LIB, srcs + HEADERS):
    return LIB
  os.makedirs(OBJ_DIR, exist_ok=True)
  cc = hipcc()
  inc = ["-I", os.path.join(ROOT, "include"), "-I", CSRC]
  procs, objs = [], []
  for src in srcs:                       # the translation units compile in parallel
    obj = os.path.join(OBJ_DIR, os.path.splitext(os.path.basename(src))[0] + ".o")
    objs.append(obj)
    if not force and _newer(obj, [src] + HEADERS):
      continue
    cmd = [cc, *HIPCC_FLAGS, *HIPCC_FLAGS_EXTRA, *os.environ.get("SBSIM_EXTRA_HIPCC_FLAGS", "").split(), *inc, "-c", src, "-o", obj]   # developer builds
    if verbose:
      cmd.append("-Rpass-analysis=kernel-resource-usage")
    procs.append((cmd, subprocess.Popen(cmd)))
  for cmd, pr in procs:
    if pr.wait() != 0:
      raise subprocess.CalledProcessError(pr.returncode, cmd)
  subprocess.run([cc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB], check=True)
  return LIB


if __name__ == "__main__":
  print(build(force=True, verbose=True))
