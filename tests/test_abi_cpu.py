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
  import torch  # noqa: F401  (before the library: see _ffi.load)
  lib = C.CDLL(path)
  names = _declared()
  assert set(names) == set(_ffi.EXPORTS), (names, _ffi.EXPORTS)
  for n in names:
    assert hasattr(lib, n), n
  assert _ffi.load().sb_abi_version() == _ffi.SB_ABI_VERSION == 8


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


def _plan_info(fp, n_obs=46, n_buildings=1024):
  import numpy as np
  cp = fp.compile(300.0, 12.0)
  keep = [np.ascontiguousarray(x) for x in (cp.cell_class, cp.class_coef, cp.class_zone, cp.zone_off, cp.zone_cells)]
  desc = _ffi.PlanDesc(cp.H, cp.W, cp.Z, cp.n_classes, keep[0].ctypes.data_as(C.POINTER(C.c_uint8)),
                       keep[1].ctypes.data_as(_ffi._dp), keep[2].ctypes.data_as(_ffi._ip),
                       keep[3].ctypes.data_as(_ffi._ip), keep[4].ctypes.data_as(_ffi._ip))
  info = _ffi.LaunchInfo()
  rc = _ffi.load().sb_plan_info(C.byref(desc), n_obs, n_buildings, C.byref(info))
  return rc, {f[0]: getattr(info, f[0]) for f in _ffi.LaunchInfo._fields_}


def test_plan_info_picks_the_step_kernel_without_a_gpu():
  """Host-only launch planning (sb_plan_info): R9 (68x98, 66x96 inside the exterior ring) runs
  on the register path either way -- lanes = rows: one wavefront owns rows 0..63 and the two
  remaining wall rows are finished by a scan (four buildings = one workgroup per CU, sharing the steps'
  class words in LDS); lanes = columns: one
  wavefront, two columns per lane, two buildings per CU; a 129-row plan: two rows per lane + one
  tail row; a small plan uses one wavefront per building."""
  from sbsim_amd.floorplan import FloorPlan, Materials, rectangular_floor_plan
  r9 = FloorPlan.from_file_input(rectangular_floor_plan((3, 3), (20, 30)), Materials.sb1(), 10.0, 300.0)
  rc, rows = _plan_info(r9)
  assert rc == 0 and rows["path"] == 1 and rows["waves_per_building"] == 1 and rows["kernel"] == 3   # k_sweep_roll
  assert rows["waves_per_workgroup"] == 4 and rows["workgroups"] == 256 and rows["lds_bytes_per_workgroup"] <= 160 * 1024
  rc, cols = _plan_info(r9.transposed())
  assert rc == 0 and cols["path"] == 1 and cols["waves_per_building"] == 1 and cols["kernel"] == 4   # k_sweep_two
  # step_two.hip: 66 columns -> 76 slots; a rectangular plan's cells are two-coefficient cells, so the rows are
  # shifted by one (a pad row above row 0): 96 rows -> 49 lanes; most of A streams from L2: four buildings per CU
  assert cols["sweep_steps"] == 76 + 49 - 1
  assert 4 * ((cols["lds_bytes_per_workgroup"] + 1279) // 1280 * 1280) <= 160 * 1024
  assert cols["algorithmic_bytes_per_env_step"] == 53764
  sb1 = FloorPlan.from_file_input(rectangular_floor_plan((14, 9), (8, 7)), Materials.sb1(), 10.0, 300.0)
  rc, big = _plan_info(sb1)
  assert rc == 0 and big["path"] == 1 and big["waves_per_building"] == 1 and big["kernel"] == 4
  assert big["sweep_steps"] == 76 + 64 - 1 + 8         # 129 x 75 inside the ring: rows -1 .. 126 on 64 lanes + two tail rows
  assert 4 * ((big["lds_bytes_per_workgroup"] + 1279) // 1280 * 1280) <= 160 * 1024
  small = FloorPlan.from_file_input(rectangular_floor_plan((1, 2), (6, 8)), Materials.sb1(), 10.0, 300.0)
  rc, one = _plan_info(small)
  assert rc == 0 and one["path"] == 1 and one["waves_per_building"] == 1 and one["kernel"] == 1
  # a plan of <= 64 rows whose k_sweep_reg sweep (NR + rows - 1 steps) is at least as long as k_sweep_roll's 96-step
  # period runs on k_sweep_roll without tail rows (round 4): 45 x 96 inside the ring
  wide = FloorPlan.from_file_input(rectangular_floor_plan((2, 3), (20, 30)), Materials.sb1(), 10.0, 300.0)
  rc, wr = _plan_info(wide)
  assert rc == 0 and wr["kernel"] == 3 and wr["waves_per_workgroup"] == 4 and wr["sweep_steps"] == 96
  # ... and plans of <= 64 columns on its 64-slot instantiation: 47 x 48 (k_sweep_reg: 66 + 46 steps per sweep)
  narrow = FloorPlan.from_file_input(rectangular_floor_plan((4, 5), (10, 8)), Materials.sb1(), 10.0, 300.0)
  rc, nr = _plan_info(narrow, n_obs=3 * 20 + 19)
  assert rc == 0 and nr["kernel"] == 3 and nr["waves_per_workgroup"] == 4 and nr["sweep_steps"] == 64
  assert nr["lds_bytes_per_workgroup"] <= 160 * 1024   # four buildings per workgroup, one workgroup per CU
  # 65..80 columns: the 80-slot instantiation (72 of its A slots in LDS, like the 96-slot one); 65 x 78: + one tail row
  for shape, steps in (((20, 24), 80), ((30, 24), 84)):
    mid = FloorPlan.from_file_input(rectangular_floor_plan((2, 3), shape), Materials.sb1(), 10.0, 300.0)
    rc, mr = _plan_info(mid)
    assert rc == 0 and mr["kernel"] == 3 and mr["sweep_steps"] == steps and mr["lds_bytes_per_workgroup"] <= 160 * 1024


def test_plan_info_for_plans_beyond_one_cu_and_the_opt_in_kernel(monkeypatch):
  """Host-only planning of this round's kernels: a 299 x 401 plan (0.96 MB of state) goes to the streaming
  kernel on five wavefronts, its LDS (seam rows + zone sums) leaves room for three workgroups per CU (round 5: an
  8-column zone-sum scratch and 128 registers per lane); a plan
  of more than 1,024 rows in one orientation still runs in the other and is SB_ERR_TOO_LARGE when both
  are too long; SBSIM_BAND_PATH=1 puts a 107-row plan on two wavefronts (step_band.hip), two buildings per CU."""
  import numpy as np
  from sbsim_amd.floorplan import FloorPlan, Materials, rectangular_floor_plan
  big = FloorPlan.from_file_input(rectangular_floor_plan((14, 9), (20, 43)), Materials.sb1(), 10.0, 300.0)
  assert big.shape == (299, 401)
  rc, info = _plan_info(big, n_obs=397, n_buildings=4096)
  assert rc == 0 and info["path"] == 2 and info["kernel"] == 6 and info["waves_per_building"] == 5
  assert info["sweep_steps"] == 64 * 4 + 400 + 63
  assert 3 * ((info["lds_bytes_per_workgroup"] + 1279) // 1280 * 1280) <= 160 * 1024
  assert info["workgroups"] == 768 and info["algorithmic_bytes_per_env_step"] == 8 * 299 * 401 + 24 * 126 + 8 + 4 * 397 + 44
  tall = FloorPlan.from_file_input(rectangular_floor_plan((1, 1), (1119, 20)), Materials.sb1(), 10.0, 300.0)   # 1,125 x 26
  rc, t_rows = _plan_info(tall, n_obs=22)
  assert rc == -4 and b"1,024 rows" in _ffi.load().sb_last_error()      # SB_ERR_TOO_LARGE in this orientation ...
  rc, t_cols = _plan_info(tall.transposed(), n_obs=22)
  assert rc == 0 and t_cols["kernel"] == 6 and t_cols["waves_per_building"] == 1   # ... 24 rows x 1,123 columns in the other
  huge = FloorPlan.from_file_input(rectangular_floor_plan((1, 1), (514, 1094)), Materials.sb1(), 10.0, 300.0)  # 520 x 1,100
  rc, _ = _plan_info(huge, n_obs=22)
  assert rc == -4 and b"seam rows" in _ffi.load().sb_last_error()        # nine wavefronts x 1,100 columns of seam rows: > 160 KiB
  rc, _ = _plan_info(huge.transposed(), n_obs=22)
  assert rc == -4 and b"1,024 rows" in _ffi.load().sb_last_error()
  sb2 = FloorPlan.from_file_input(rectangular_floor_plan((8, 5), (12, 14)), Materials.sb1(), 10.0, 300.0)
  rc, two = _plan_info(sb2, n_obs=3 * 40 + 19)
  assert rc == 0 and two["kernel"] == 4
  monkeypatch.setenv("SBSIM_BAND_PATH", "1")
  rc, band = _plan_info(sb2, n_obs=3 * 40 + 19)
  assert rc == 0 and band["kernel"] == 5 and band["waves_per_building"] == 2 and band["sweep_steps"] == 80
  assert 2 * ((band["lds_bytes_per_workgroup"] + 1279) // 1280 * 1280) <= 160 * 1024


def test_plan_info_for_plans_of_131_to_258_rows(monkeypatch):
  """Round 4: plans beyond 128 rows (up to 258, up to 96 columns inside the exterior ring) run on step_band.hip with
  three or four wavefronts per building, one building per CU (its LDS holds A of every wavefront); SBSIM_NO_BAND_PATH=1
  puts them back on the LDS-grid / streaming kernel; a plan wider than 96 columns both ways does not qualify."""
  from sbsim_amd.floorplan import FloorPlan, Materials, rectangular_floor_plan
  mk = lambda rooms, shape: FloorPlan.from_file_input(rectangular_floor_plan(rooms, shape), Materials.sb1(), 10.0, 300.0)
  p4 = mk((10, 4), (19, 20))          # 205 x 89: 203 x 87 inside the ring
  rc, i4 = _plan_info(p4, n_obs=3 * 40 + 19, n_buildings=4096)
  assert rc == 0 and i4["path"] == 1 and i4["kernel"] == 5 and i4["waves_per_building"] == 4 == i4["waves_per_workgroup"]
  assert i4["sweep_steps"] == 88 and i4["workgroups"] == 256 and i4["lds_bytes_per_workgroup"] <= 160 * 1024   # 87 columns: 88 slots
  rc, i3 = _plan_info(mk((10, 4), (18, 20)), n_obs=3 * 40 + 19)   # 193 x 87: three wavefronts + ONE tail row
  assert rc == 0 and i3["kernel"] == 5 and i3["waves_per_building"] == 3 and i3["sweep_steps"] == 88 + 4
  rc, i96 = _plan_info(mk((10, 4), (19, 22)), n_obs=3 * 40 + 19)  # 203 x 95: the widest instantiation
  assert rc == 0 and i96["kernel"] == 5 and i96["waves_per_building"] == 4 and i96["sweep_steps"] == 96
  assert i96["lds_bytes_per_workgroup"] <= 160 * 1024
  rc, i3b = _plan_info(mk((9, 4), (16, 17)), n_obs=3 * 36 + 19)   # 156 x 75: 76 slots
  assert rc == 0 and i3b["kernel"] == 5 and i3b["waves_per_building"] == 3 and i3b["sweep_steps"] == 76
  rc, it = _plan_info(p4.transposed(), n_obs=3 * 40 + 19)         # lanes = the file's columns: 87 rows x 203 columns
  assert rc == 0 and it["kernel"] != 5
  rc, wide = _plan_info(mk((6, 6), (24, 24)), n_obs=3 * 36 + 19)  # 153 x 153 inside the ring: too wide either way
  assert rc == 0 and wide["kernel"] != 5
  rc, w2 = _plan_info(mk((6, 4), (17, 21)), n_obs=3 * 24 + 19, n_buildings=4096)   # 111 x 91: too wide for step_two.hip -> two wavefronts, 92 slots
  assert rc == 0 and w2["kernel"] == 5 and w2["waves_per_building"] == 2 and w2["sweep_steps"] == 92 and w2["workgroups"] == 512
  monkeypatch.setenv("SBSIM_NO_BAND_PATH", "1")
  rc, w2off = _plan_info(mk((6, 4), (17, 21)), n_obs=3 * 24 + 19)
  assert rc == 0 and w2off["kernel"] == 2
  rc, off = _plan_info(p4, n_obs=3 * 40 + 19)
  assert rc == 0 and off["kernel"] == 0 and off["waves_per_building"] == 1


def test_plan_checks_reject_bad_tables_without_a_gpu():
  """Host-side validation shared by sb_plan_info and sb_create: status codes, not crashes."""
  import numpy as np
  L = _ffi.load()
  info = _ffi.LaunchInfo()
  cls = np.zeros(6, np.uint8); coef = np.zeros((1, 8)); coef[0, 5] = 1.0
  cz = np.full(1, -1, np.int32); zoff = np.zeros(1, np.int32); zc = np.zeros(1, np.int32)
  def desc(**kw):
    d = dict(H=2, W=3, Z=0, n=1, cls=cls.ctypes.data_as(C.POINTER(C.c_uint8)), coef=coef.ctypes.data_as(_ffi._dp),
             cz=cz.ctypes.data_as(_ffi._ip), zoff=zoff.ctypes.data_as(_ffi._ip), zc=zc.ctypes.data_as(_ffi._ip))
    d.update(kw)
    return _ffi.PlanDesc(d["H"], d["W"], d["Z"], d["n"], d["cls"], d["coef"], d["cz"], d["zoff"], d["zc"])
  assert L.sb_plan_info(C.byref(desc()), 20, 4, C.byref(info)) == 0     # all exterior: LDS-grid kernel
  assert info.path == 0
  assert L.sb_plan_info(C.byref(desc(H=0)), 20, 4, C.byref(info)) == -1
  assert L.sb_plan_info(C.byref(desc(cls=None)), 20, 4, C.byref(info)) == -1
  assert b"null floor-plan table" in L.sb_last_error()
  bad = np.full(6, 7, np.uint8)
  assert L.sb_plan_info(C.byref(desc(cls=bad.ctypes.data_as(C.POINTER(C.c_uint8)))), 20, 4, C.byref(info)) == -1
  assert L.sb_plan_info(None, 20, 4, C.byref(info)) == -1
  assert L.sb_plan_info(C.byref(desc()), 20, 4, None) == -1


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


def test_plan_info_narrowest_instantiations(monkeypatch):
  """The planner takes the narrowest instantiation that holds a plan's columns (end of round 4): k_sweep_two on 64
  slots for 107 x 58 (SBSIM_NO_TWO_64=1: 76), step_band.hip on 68 / 72 slots for 156 x 67 / 255 x 72."""
  from sbsim_amd.floorplan import FloorPlan, Materials, rectangular_floor_plan
  mk = lambda rooms, shape: FloorPlan.from_file_input(rectangular_floor_plan(rooms, shape), Materials.sb1(), 10.0, 300.0)
  plan = mk((8, 5), (12, 10))
  rc, a = _plan_info(plan, n_obs=3 * 40 + 19)
  assert rc == 0 and a["kernel"] == 4 and a["lds_bytes_per_workgroup"] <= 160 * 1024
  monkeypatch.setenv("SBSIM_NO_TWO_64", "1")
  rc, b = _plan_info(plan, n_obs=3 * 40 + 19)
  monkeypatch.delenv("SBSIM_NO_TWO_64")
  assert rc == 0 and b["kernel"] == 4 and b["sweep_steps"] == a["sweep_steps"] + 12   # 76 - 64 slots
  rc, c = _plan_info(mk((9, 4), (16, 15)), n_obs=3 * 36 + 19)
  assert rc == 0 and c["kernel"] == 5 and c["waves_per_building"] == 3 and c["sweep_steps"] == 68
  rc, d = _plan_info(mk((12, 3), (20, 22)), n_obs=3 * 36 + 19)
  assert rc == 0 and d["kernel"] == 5 and d["waves_per_building"] == 4 and d["sweep_steps"] in (72, 76)   # 255 rows: + a tail row's 4 steps when it has one
