"""ctypes binding of the C ABI in include/sbsim_amd.h (host code stays Python; HIP does the work).

There is deliberately no fallback: if the shared library is missing or no MI355X is
visible the constructors raise.  Nothing in this package imports ``oracle``."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SBSIM_LIB") or os.path.join(_HERE, "libsbsim_amd.so")

SB_NUM_ACTIONS = 2     # the SB1 action set; sb_params.n_actions is the width of an action row
SB_ACTION_KEEP = -3.0e38   # sb_step_in.actions_native: this column leaves its field alone
SB_NUM_AUX = 7
SB_ABI_VERSION = 8   # include/sbsim_amd.h
# sb_action_kind
SB_ACT_BOILER_SUPPLY_WATER_SETPOINT, SB_ACT_AHU_SUPPLY_AIR_HEATING_SETPOINT = 0, 1
SB_ACT_AHU_SUPPLY_AIR_COOLING_SETPOINT, SB_ACT_VAV_SUPPLY_AIR_DAMPER_COMMAND = 2, 3
SB_INFO_STRIDE = 24
SB_NUM_SCALARS = 16

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)
_fp = C.POINTER(C.c_float)


class PlanDesc(C.Structure):
  _fields_ = [("H", C.c_int32), ("W", C.c_int32), ("Z", C.c_int32), ("n_classes", C.c_int32),
              ("cell_class", C.POINTER(C.c_uint8)), ("class_coef", _dp), ("class_zone", _ip),
              ("zone_off", _ip), ("zone_cells", _ip)]


PARAM_FIELDS = [
    ("dt", C.c_double), ("conv_threshold", C.c_double), ("iter_limit", C.c_int32),
    ("ahu_has_weather", C.c_int32),
    ("vav_max_air_flow", C.c_double), ("vav_max_water_flow", C.c_double),
    ("ahu_recirc", C.c_double), ("ahu_heat_sp", C.c_double), ("ahu_cool_sp", C.c_double),
    ("ahu_dp", C.c_double), ("ahu_eff", C.c_double), ("ahu_max_flow", C.c_double),
    ("blr_setpoint", C.c_double), ("blr_head", C.c_double), ("blr_pump_eff", C.c_double),
    ("blr_heating_rate", C.c_double), ("blr_cooling_rate", C.c_double),
    ("blr_conv", C.c_double), ("blr_len", C.c_double), ("blr_radius", C.c_double),
    ("blr_capacity", C.c_double), ("blr_ins_k", C.c_double), ("blr_ins_thick", C.c_double),
    ("comfort_lo", C.c_double), ("comfort_hi", C.c_double), ("eco_lo", C.c_double),
    ("eco_hi", C.c_double),
    ("max_prod", C.c_double), ("min_prod", C.c_double), ("max_elec", C.c_double),
    ("max_gas", C.c_double), ("prod_delta", C.c_double), ("prod_stiff", C.c_double),
    ("w_prod", C.c_double), ("w_cost", C.c_double), ("w_carbon", C.c_double),
    ("n_actions", C.c_int32), ("act_kind", C.POINTER(C.c_int32)), ("act_zone", C.POINTER(C.c_int32)),
    ("act_lo", C.POINTER(C.c_double)), ("act_hi", C.POINTER(C.c_double)),
]


class Params(C.Structure):
  _fields_ = PARAM_FIELDS


class ObsLayout(C.Structure):
  _fields_ = [("n_obs", C.c_int32), ("col_ahu", C.c_int32), ("col_boiler", C.c_int32),
              ("col_aux", C.c_int32), ("col_zone", _ip), ("mean", _dp), ("sigma", _dp),
              ("n_src", C.c_int32), ("n_hist", C.c_int32), ("src_dest", _ip), ("hist_col", _ip),
              ("hist_off", _ip), ("hist_bins", _dp), ("hist_normalize", C.c_int32)]


class StepIn(C.Structure):
  _fields_ = [("t_amb_now", C.c_double), ("t_amb_next", C.c_double), ("t_amb_dev", C.c_void_p),
              ("weather_lohi_dev", C.c_void_p), ("weather_f_now", C.c_double), ("weather_f_next", C.c_double),
              ("weather_times_dev", C.c_void_p), ("weather_tempf_dev", C.c_void_p), ("weather_offset_dev", C.c_void_p),
              ("weather_n", C.c_int32), ("weather_t_now", C.c_double), ("weather_t_next", C.c_double),
              ("comfort_now", C.c_int32), ("comfort_prev", C.c_int32),
              ("comfort_next", C.c_int32), ("has_action", C.c_int32), ("reject_dev", C.c_void_p), ("actions_native", C.c_int32),
              ("occupancy", C.c_double), ("occupancy_dev", C.c_void_p),
              ("occupancy_bz_dev", C.c_void_p), ("num_occupants_dev", C.c_void_p), ("occupancy_norm", C.c_double),
              ("e_price", C.c_double), ("e_carbon", C.c_double),
              ("g_price", C.c_double), ("g_carbon", C.c_double),
              ("aux", C.c_float * SB_NUM_AUX)]


class OccupancyConfig(C.Structure):
  _fields_ = [("zone_assignment", C.c_int32), ("earliest_arrival_hour", C.c_int32),
              ("latest_arrival_hour", C.c_int32), ("earliest_departure_hour", C.c_int32),
              ("latest_departure_hour", C.c_int32), ("time_step_sec", C.c_double),
              ("seed", C.c_uint64), ("first_building", C.c_int64)]


class TapBld(C.Structure):   # sb_tap_bld
  _fields_ = [("t_now", C.c_double), ("t_next", C.c_double), ("heat_sp", C.c_double), ("cool_sp", C.c_double),
              ("blr_sp", C.c_double), ("t_sa", C.c_double), ("ahu_flow", C.c_double), ("blr_flow", C.c_double),
              ("blr_return", C.c_double), ("ahu_count", C.c_int32), ("blr_count", C.c_int32),
              ("tank", C.c_double), ("tank_change", C.c_double), ("duration", C.c_double), ("rejected", C.c_int32)]


class PbTime(C.Structure):
  _fields_ = [("seconds", C.c_int64), ("nanos", C.c_int32)]


class LaunchInfo(C.Structure):
  _fields_ = [("waves_per_workgroup", C.c_int32), ("workgroups", C.c_int32),
              ("lds_bytes_per_workgroup", C.c_int32), ("sweep_steps", C.c_int32),
              ("algorithmic_bytes_per_env_step", C.c_int64),
              ("state_bytes_per_env_step", C.c_int64),
              ("path", C.c_int32), ("waves_per_building", C.c_int32),
              ("kernel", C.c_int32), ("reserved", C.c_int32)]


# sb_sweep_kernel
SWEEP_KERNELS = {0: "k_sweep_lds", 1: "k_sweep_reg", 2: "k_sweep_reg (two wavefronts)", 3: "k_sweep_roll", 4: "k_sweep_two",
                 5: "k_sweep_band", 6: "k_sweep_stream"}


EXPORTS = ("sb_abi_version", "sb_has_experimental_kernels", "sb_last_error", "sb_plan_info", "sb_create", "sb_destroy", "sb_get_launch_info",
           "sb_reset", "sb_observe", "sb_observe_occupancy", "sb_occupancy_attach", "sb_occupancy_peek", "sb_convection_attach", "sb_step", "sb_step_phases", "sb_get_temps", "sb_set_temps", "sb_get_zone_temps",
           "sb_get_scalars", "sb_get_modes", "sb_get_zone_power", "sb_debug_phase_cycles",
           "sb_floorplan_padded_shape", "sb_floorplan_preprocess", "sb_floorplan_diffusers", "sb_debug_numpy_choice", "sb_pb_reward_info", "sb_pb_reward_response",
           "sb_pb_observation_response", "sb_pb_action_response", "sb_shard_append", "sb_pb_device_info",
           "sb_pb_zone_info", "sb_pb_variable_info", "sb_record_append", "sb_tap_pre", "sb_tap_post")

_lib = None


class SbsimError(RuntimeError):
  pass


def load():
  """Loads libsbsim_amd.so; raises if it has not been built (no silent fallback)."""
  global _lib
  if _lib is not None:
    return _lib
  if not os.path.exists(LIB_PATH):
    raise SbsimError(
        f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
        "(hipcc --offload-arch=gfx950).  sbsim_amd has no CPU fallback.")
  # torch FIRST: it ships its own HIP runtime, and the device memory this library works on is torch's.  Loaded before torch,
  # the library binds the system's libamdhip64 instead -- a second runtime in the process that sees no device
  # (`sb_create: no HIP device visible` after __graft_entry__.build() and smoke() in one process)
  import torch  # noqa: F401
  L = C.CDLL(LIB_PATH)
  vp = C.c_void_p
  L.sb_abi_version.restype = C.c_int
  if L.sb_abi_version() != SB_ABI_VERSION:   # a stale library would read these structs with another layout
    raise SbsimError(f"{LIB_PATH} has ABI version {L.sb_abi_version()}, this package needs {SB_ABI_VERSION}: "
                     "rebuild it (`python -c 'import __graft_entry__ as g; g.build()'`)")
  L.sb_last_error.restype = C.c_char_p
  L.sb_create.argtypes = [C.POINTER(PlanDesc), C.POINTER(Params), C.POINTER(ObsLayout),
                          C.c_int32, C.c_int32, C.POINTER(vp)]
  L.sb_plan_info.argtypes = [C.POINTER(PlanDesc), C.c_int32, C.c_int32, C.POINTER(LaunchInfo)]
  L.sb_destroy.argtypes = [vp]
  L.sb_destroy.restype = None
  L.sb_get_launch_info.argtypes = [vp, C.POINTER(LaunchInfo)]
  L.sb_reset.argtypes = [vp, C.c_double, vp, vp]
  L.sb_observe.argtypes = [vp, C.POINTER(C.c_float), C.c_double, vp, vp, vp]
  L.sb_observe_occupancy.argtypes = [vp, C.POINTER(C.c_float), C.c_double, vp, vp, C.c_double, vp, vp]
  L.sb_occupancy_attach.argtypes = [vp, C.POINTER(OccupancyConfig)]
  L.sb_occupancy_peek.argtypes = [vp, C.c_int32, C.c_int32, vp, vp, vp]
  L.sb_convection_attach.argtypes = [vp, C.c_double, C.c_int32, C.c_uint64, C.c_int64, C.c_int32]
  L.sb_step.argtypes = [vp, vp, C.POINTER(StepIn), vp, vp, vp, vp]
  L.sb_step_phases.argtypes = [vp, vp, C.POINTER(StepIn), vp, vp, vp, vp, C.c_int32]
  for name in ("sb_get_temps", "sb_set_temps", "sb_get_zone_temps", "sb_get_scalars", "sb_get_modes",
               "sb_get_zone_power"):
    getattr(L, name).argtypes = [vp, vp, vp]
  L.sb_debug_phase_cycles.argtypes = [vp, C.POINTER(C.c_longlong)]
  L.sb_tap_pre.argtypes = [vp, C.c_int32, _dp, _ip, _dp, _fp, C.POINTER(StepIn), C.POINTER(TapBld), _dp, _dp, _ip]
  L.sb_tap_post.argtypes = [vp, C.c_int32, C.POINTER(TapBld), _dp, C.c_double, C.c_int32, C.POINTER(StepIn), _fp, _fp]
  cpp, fp_ = C.POINTER(C.c_char_p), C.POINTER(C.c_float)
  L.sb_pb_reward_info.argtypes = [PbTime, PbTime, C.c_char_p, C.c_char_p, C.c_int32, cpp, fp_, C.c_int32, cpp, fp_,
                                  C.c_int32, cpp, fp_, vp, C.c_int64]
  L.sb_pb_reward_response.argtypes = [fp_, PbTime, PbTime, vp, C.c_int64]
  L.sb_pb_observation_response.argtypes = [PbTime, C.c_int32, cpp, cpp, fp_, vp, vp, C.c_int64]
  L.sb_pb_action_response.argtypes = [PbTime, PbTime, C.c_int32, cpp, cpp, fp_, vp, vp, C.c_int64]
  for name in ("sb_pb_reward_info", "sb_pb_reward_response", "sb_pb_observation_response", "sb_pb_action_response"):
    getattr(L, name).restype = C.c_int64
  L.sb_shard_append.argtypes = [C.c_char_p, C.c_char_p, C.c_int64, vp, C.c_int64]
  ip_ = C.POINTER(C.c_int32)
  L.sb_pb_device_info.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_int32, C.c_int32, cpp, ip_,
                                  C.c_int32, cpp, ip_, vp, C.c_int64]
  L.sb_pb_zone_info.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_float, C.c_int32, cpp, C.c_int32, C.c_int32,
                                vp, C.c_int64]
  L.sb_pb_device_info.restype = L.sb_pb_zone_info.restype = C.c_int64
  L.sb_record_append.argtypes = [C.c_char_p, vp, C.c_int64, C.c_int32]
  L.sb_pb_variable_info.argtypes = [C.c_char_p, C.c_int32, PbTime, C.c_int32, PbTime, C.c_int32, fp_, vp, C.c_int64]
  L.sb_pb_variable_info.restype = C.c_int64
  L.sb_floorplan_padded_shape.argtypes = [vp, C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
  L.sb_floorplan_preprocess.argtypes = [vp, vp, C.c_int32, C.c_int32, vp, vp, vp, vp, C.POINTER(C.c_int32)]
  L.sb_floorplan_diffusers.argtypes = [vp, vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, vp]
  L.sb_debug_numpy_choice.argtypes = [C.c_uint32, C.c_int64, C.c_int64, vp]
  _lib = L
  return L


def check(rc: int, what: str) -> None:
  if rc != 0:
    msg = load().sb_last_error()
    raise SbsimError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")
