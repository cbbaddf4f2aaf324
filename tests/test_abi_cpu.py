"""CPU-only checks of the drop-in boundary: the shared library builds, loads and exports
every symbol include/sbsim_amd.h declares, and fails loudly (no fallback) without a GPU."""
import ctypes as C
import os
import re

import pytest

from sbsim_amd import _ffi, build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
  text = open(os.path.join(ROOT, "include", "sbsim_amd.h")).read()
  text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
  return sorted(set(re.findall(r"\b(sb_[a-z_]+)\s*\(", text)))


def test_library_builds_and_exports_every_declared_symbol():
  path = build.build()
  lib = C.CDLL(path)
  names = _declared()
  assert set(names) == set(_ffi.EXPORTS), (names, _ffi.EXPORTS)
  for n in names:
    assert hasattr(lib, n), n
  assert _ffi.load().sb_abi_version() == 1


def test_struct_layouts_match_header():
  # sizes are what the C compiler computes for the same field lists (checked via a tiny probe)
  import subprocess, tempfile, textwrap
  src = textwrap.dedent("""
      #include <stdio.h>
      #include "sbsim_amd.h"
      int main(void) { printf("%zu %zu %zu %zu %zu\\n", sizeof(sb_plan_desc), sizeof(sb_params),
                       sizeof(sb_obs_layout), sizeof(sb_step_in), sizeof(sb_launch_info)); return 0; }
  """)
  with tempfile.TemporaryDirectory() as d:
    c = os.path.join(d, "p.c")
    open(c, "w").write(src)
    exe = os.path.join(d, "p")
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), c, "-o", exe], check=True)
    sizes = [int(x) for x in subprocess.run([exe], check=True, capture_output=True, text=True).stdout.split()]
  assert sizes == [C.sizeof(_ffi.PlanDesc), C.sizeof(_ffi.Params), C.sizeof(_ffi.ObsLayout),
                   C.sizeof(_ffi.StepIn), C.sizeof(_ffi.LaunchInfo)]


def test_no_cpu_fallback_without_gpu():
  import torch
  if torch.cuda.is_available():
    pytest.skip("GPU present")
  from sbsim_amd.environment import BatchedSimulator, SimConfig
  from sbsim_amd.floorplan import FloorPlan, Materials, rectangular_floor_plan
  fp = FloorPlan.from_file_input(rectangular_floor_plan((1, 2), (6, 8)), Materials.sb1(), 10.0, 300.0)
  with pytest.raises(_ffi.SbsimError):
    BatchedSimulator(fp, SimConfig(), 4, 12.0)


def test_product_does_not_import_oracle():
  pkg = os.path.join(ROOT, "sbsim_amd")
  for dirpath, _, files in os.walk(pkg):
    for f in files:
      if f.endswith((".py", ".hip", ".h")):
        text = open(os.path.join(dirpath, f)).read()
        assert "import oracle" not in text and "from oracle" not in text, f
