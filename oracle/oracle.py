"""ctypes front-end of the CPU parity oracle (TEST INFRASTRUCTURE).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module.  The product package ``sbsim_amd`` never does: it fails loudly when
its HIP library is missing instead of falling back to anything in here.

The numeric work is in ``sb_oracle.c`` (float64, reference operation order).  This file
only marshals numpy arrays into the C structs and restates two tiny pieces of reference
bookkeeping: the neighbour lists (``building.py:794-813`` /  ``:529-545``) and the
default thermostat/device construction state.
"""
from __future__ import annotations

import ctypes as C
import dataclasses
import os
import subprocess
from typing import List, Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libsb_oracle.so")
_lib = None

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)
_fp = C.POINTER(C.c_float)


class _Plan(C.Structure):
  _fields_ = [("H", C.c_int32), ("W", C.c_int32), ("Z", C.c_int32),
              ("dx", C.c_double), ("zh", C.c_double),
              ("k", _dp), ("rho", _dp), ("c", _dp),
              ("nbr_cnt", _ip), ("nbr_idx", _ip),
              ("zone_off", _ip), ("zone_cells", _ip), ("diffuser", _dp)]


_PARAM_FIELDS = [
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
]


class _Params(C.Structure):
  _fields_ = _PARAM_FIELDS


class _State(C.Structure):
  _fields_ = [("temp", _dp), ("input_q", _dp), ("scratch", _dp), ("mode", _ip),
              ("damper", _dp), ("valve", _dp), ("zone_air_temp", _dp),
              ("ahu_heat_sp", C.c_double), ("ahu_cool_sp", C.c_double),
              ("ahu_flow", C.c_double), ("ahu_count", C.c_int32),
              ("blr_setpoint", C.c_double), ("blr_flow", C.c_double),
              ("blr_return_temp", C.c_double), ("blr_tank_temp", C.c_double),
              ("blr_tank_change", C.c_double), ("blr_last_duration", C.c_double),
              ("blr_count", C.c_int32), ("blr_has_action_ts", C.c_int32),
              ("blr_action_ts", C.c_double), ("thermostat_has_prev", C.c_int32),
              ("thermostat_prev_comfort", C.c_int32), ("thermostat_skipped", C.c_int32)]


class _StepIn(C.Structure):
  _fields_ = [("now_ts", C.c_double), ("has_action", C.c_int32),
              ("blr_setpoint", C.c_double), ("ahu_heat_sp", C.c_double),
              ("t_amb_now", C.c_double), ("h_conv", C.c_double), ("t_amb_next", C.c_double),
              ("comfort_now", C.c_int32), ("comfort_prev", C.c_int32),
              ("comfort_next", C.c_int32), ("occupancy", _dp), ("observe", C.c_int32),
              ("e_price", C.c_double), ("e_carbon", C.c_double),
              ("g_price", C.c_double), ("g_carbon", C.c_double),
              ("reject", C.c_int32), ("has_cool_sp", C.c_int32), ("ahu_cool_sp", C.c_double),
              ("damper_cmd", _dp)]


class _StepOut(C.Structure):
  _fields_ = [("n_sweeps", C.c_int32), ("converged", C.c_int32),
              ("t_supply_air", C.c_double), ("recirc_pre", C.c_double),
              ("blower_rate", C.c_float), ("ac_rate", C.c_float),
              ("gas_rate", C.c_float), ("pump_rate", C.c_float),
              ("reward", C.c_float), ("reward_f64", C.c_double),
              ("productivity", C.c_double), ("norm_prod_regret", C.c_double),
              ("norm_energy_cost", C.c_double), ("norm_carbon", C.c_double),
              ("zone_temp_pre", _dp), ("zone_temp_post", _dp), ("q_zone", _dp),
              ("action_accepted", C.c_int32)]


def build(force: bool = False) -> str:
  """Compiles ``libsb_oracle.so`` with the committed Makefile (gcc, no other deps)."""
  srcs = [os.path.join(_HERE, f) for f in ("sb_oracle.c", "sb_oracle_batch.c", "sb_oracle.h")]
  stale = (not os.path.exists(_LIB_PATH) or
           any(os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs))
  if force or stale:
    subprocess.run(["make", "-C", _HERE, "-s", "-B" if force else "-s"], check=True)
  return _LIB_PATH


def lib():
  global _lib
  if _lib is None:
    build()
    L = C.CDLL(_LIB_PATH)
    L.sbo_pairwise_sum.restype = C.c_double
    L.sbo_pairwise_sum.argtypes = [_dp, C.c_int64]
    L.sbo_np_mean.restype = C.c_double
    L.sbo_np_mean.argtypes = [_dp, C.c_int64]
    L.sbo_zone_means.argtypes = [C.POINTER(_Plan), _dp, _dp]
    L.sbo_sweep.restype = C.c_double
    L.sbo_sweep.argtypes = [C.POINTER(_Plan), _dp, _dp, _dp, C.c_double, C.c_double, C.c_double]
    L.sbo_fd_timestep.restype = C.c_int32
    L.sbo_fd_timestep.argtypes = [C.POINTER(_Plan), _dp, _dp, _dp, C.c_double, C.c_double,
                                  C.c_double, C.c_double, C.c_int32, _ip]
    L.sbo_reset.argtypes = [C.POINTER(_Plan), C.POINTER(_Params), C.POINTER(_State),
                            C.c_double, _dp]
    L.sbo_setup_step.argtypes = [C.POINTER(_Plan), C.POINTER(_Params), C.POINTER(_State),
                                 C.c_int32, C.c_int32]
    L.sbo_step.argtypes = [C.POINTER(_Plan), C.POINTER(_Params), C.POINTER(_State),
                           C.POINTER(_StepIn), C.POINTER(_StepOut)]
    L.sbo_boiler_dissipation.restype = C.c_double
    L.sbo_boiler_dissipation.argtypes = [C.POINTER(_Params), C.c_double, C.c_double]
    d = C.c_double
    for name, args in (("sbo_dev_ahu_mixed", [d, d, d]), ("sbo_dev_ahu_supply", [d, d, d]),
                       ("sbo_dev_ahu_blower_power", [C.POINTER(_Params), d]),
                       ("sbo_dev_ahu_thermal_rate", [d, d, d]), ("sbo_dev_vav_supply_temp", [d, d, d, d]),
                       ("sbo_dev_vav_energy", [d, d, d]),
                       ("sbo_dev_boiler_gas_rate", [C.POINTER(_Params), d, d, d, d, d, d]),
                       ("sbo_dev_boiler_pump_power", [C.POINTER(_Params), d]),
                       ("sbo_dev_boiler_adjust", [d, d, d, d, d])):
      getattr(L, name).restype = d
      getattr(L, name).argtypes = args
    L.sbo_dev_thermostat.restype = C.c_int32
    L.sbo_dev_thermostat.argtypes = [C.c_int32, d, d, d, C.c_int32, C.c_int32, _dp, _dp]
    L.sbo_reward.restype = C.c_double
    L.sbo_reward.argtypes = [C.POINTER(_Params), C.c_int32, _fp, _fp, _fp, _fp, C.c_float,
                             C.c_float, C.c_float, C.c_float, C.c_double, C.c_double,
                             C.c_double, C.c_double, C.c_double, _dp]
    L.sbo_observe_boiler.argtypes = [C.POINTER(_Params), C.POINTER(_State), C.c_double]
    L.sbo_max_threads.restype = C.c_int32
    L.sbo_step_batch.argtypes = [C.POINTER(_Plan), C.POINTER(_Params), C.POINTER(_State),
                                 C.POINTER(_StepIn), C.c_int32, C.POINTER(_StepOut),
                                 C.c_int32, C.c_int32]
    _lib = L
  return _lib


def _d(a: np.ndarray):
  return a.ctypes.data_as(_dp)


def _i(a: np.ndarray):
  return a.ctypes.data_as(_ip)


@dataclasses.dataclass
class OracleParams:
  """Scalar configuration; defaults are the constructor defaults of the reference
  (boiler.py:54-67, air_handler.py:55) -- everything else must be given."""
  dt: float
  conv_threshold: float
  iter_limit: int
  vav_max_air_flow: float
  vav_max_water_flow: float
  ahu_recirc: float
  ahu_heat_sp: float
  ahu_cool_sp: float
  ahu_dp: float
  ahu_eff: float
  blr_setpoint: float
  blr_head: float
  blr_pump_eff: float
  comfort_lo: float
  comfort_hi: float
  eco_lo: float
  eco_hi: float
  ahu_max_flow: float = 8.67
  ahu_has_weather: int = 0
  blr_heating_rate: float = 0.0
  blr_cooling_rate: float = 0.0
  blr_conv: float = 5.6
  blr_len: float = 2.0
  blr_radius: float = 0.5
  blr_capacity: float = 1.5
  blr_ins_k: float = 0.067
  blr_ins_thick: float = 0.06
  max_prod: float = 300.0
  min_prod: float = 100.0
  max_elec: float = 160000.0
  max_gas: float = 400000.0
  prod_delta: float = 0.5
  prod_stiff: float = 4.3
  w_prod: float = 0.2
  w_cost: float = 0.4
  w_carbon: float = 0.4

  def to_c(self) -> _Params:
    p = _Params()
    for name, _ in _PARAM_FIELDS:
      setattr(p, name, getattr(self, name))
    return p


class OraclePlan:
  """Post-preprocessing floor plan (the arrays ``FloorPlanBasedBuilding`` ends up with)."""

  def __init__(self, k, rho, c, exterior_space, zones: Sequence[np.ndarray], diffuser,
               cv_size_cm: float, floor_height_cm: float, skip_exterior: bool = True):
    self.k = np.ascontiguousarray(k, dtype=np.float64)
    self.rho = np.ascontiguousarray(rho, dtype=np.float64)
    self.c = np.ascontiguousarray(c, dtype=np.float64)
    self.H, self.W = self.k.shape
    self.diffuser = np.ascontiguousarray(diffuser, dtype=np.float64)
    ext = np.asarray(exterior_space, dtype=bool)
    H, W = self.H, self.W
    # building.py:794-813 (floor-plan building: exterior-space cells have no neighbours
    # and are nobody's neighbour) / building.py:529-545 (rectangular: all in-bounds).
    nbr_idx = np.full((H, W, 4), -1, dtype=np.int32)
    nbr_cnt = np.zeros((H, W), dtype=np.int32)
    for x in range(H):
      for y in range(W):
        if skip_exterior and ext[x, y]:
          continue
        n = 0
        for nx, ny in ((x - 1, y), (x + 1, y), (x, y - 1), (x, y + 1)):
          if 0 <= nx < H and 0 <= ny < W and not (skip_exterior and ext[nx, ny]):
            nbr_idx[x, y, n] = nx * W + ny
            n += 1
        nbr_cnt[x, y] = n
    self.nbr_idx, self.nbr_cnt = nbr_idx, nbr_cnt
    self.zones = [np.ascontiguousarray(z, dtype=np.int32) for z in zones]
    self.Z = len(self.zones)
    self.zone_off = np.zeros(self.Z + 1, dtype=np.int32)
    self.zone_off[1:] = np.cumsum([len(z) for z in self.zones])
    self.zone_cells = (np.concatenate(self.zones) if self.zones
                       else np.zeros(0, np.int32)).astype(np.int32)
    self.dx = cv_size_cm / 100.0
    self.zh = floor_height_cm / 100.0
    self._c = _Plan(self.H, self.W, self.Z, self.dx, self.zh, _d(self.k), _d(self.rho),
                    _d(self.c), _i(self.nbr_cnt), _i(self.nbr_idx), _i(self.zone_off),
                    _i(self.zone_cells), _d(self.diffuser))

  @property
  def n_cells(self) -> int:
    return self.H * self.W

  def cptr(self):
    return C.byref(self._c)


def pairwise_sum(a: np.ndarray) -> float:
  a = np.ascontiguousarray(a, dtype=np.float64)
  return lib().sbo_pairwise_sum(_d(a), a.size)


def np_mean(a: np.ndarray) -> float:
  a = np.ascontiguousarray(a, dtype=np.float64)
  return lib().sbo_np_mean(_d(a), a.size)


def sweep(plan: OraclePlan, prev, est, q, t_amb, h, dt) -> float:
  """In-place sweep on ``est`` (simulator.py:278-316); returns max_delta."""
  assert est.dtype == np.float64 and est.flags.c_contiguous
  prev = np.ascontiguousarray(prev, dtype=np.float64)
  q = np.ascontiguousarray(q, dtype=np.float64)
  return lib().sbo_sweep(plan.cptr(), _d(prev), _d(est), _d(q), t_amb, h, dt)


def fd_timestep(plan: OraclePlan, temp, q, t_amb, h, dt, thr, iter_limit):
  """simulator.py:318-371 on a copy; returns (new_temp, n_sweeps, converged)."""
  t = np.array(temp, dtype=np.float64, order="C")
  scratch = np.empty_like(t)
  q = np.ascontiguousarray(q, dtype=np.float64)
  n = C.c_int32(0)
  conv = lib().sbo_fd_timestep(plan.cptr(), _d(t), _d(scratch), _d(q), t_amb, h, dt, thr,
                               iter_limit, C.byref(n))
  return t, n.value, bool(conv)


def zone_means(plan: OraclePlan, temp) -> np.ndarray:
  t = np.ascontiguousarray(temp, dtype=np.float64)
  out = np.zeros(plan.Z)
  lib().sbo_zone_means(plan.cptr(), _d(t), _d(out))
  return out


def reward(params: OracleParams, zone_temp, heat_sp, cool_sp, occ, blower, ac, gas, pump, dt,
           e_price, e_carbon, g_price, g_carbon):
  zt = np.ascontiguousarray(zone_temp, dtype=np.float32)
  hs = np.ascontiguousarray(np.broadcast_to(heat_sp, zt.shape), dtype=np.float32)
  cs = np.ascontiguousarray(np.broadcast_to(cool_sp, zt.shape), dtype=np.float32)
  oc = np.ascontiguousarray(np.broadcast_to(occ, zt.shape), dtype=np.float32)
  diag = np.zeros(4)
  pc = params.to_c()
  r = lib().sbo_reward(C.byref(pc), zt.size, zt.ctypes.data_as(_fp), hs.ctypes.data_as(_fp),
                       cs.ctypes.data_as(_fp), oc.ctypes.data_as(_fp),
                       np.float32(blower), np.float32(ac), np.float32(gas), np.float32(pump),
                       dt, e_price, e_carbon, g_price, g_carbon, _d(diag))
  return r, diag


def device(name: str, *args, params: Optional[OracleParams] = None) -> float:
  """One of the device formulas sbo_step is made of (sb_oracle.h: sbo_dev_*)."""
  fn = getattr(lib(), "sbo_dev_" + name)
  if params is not None:
    pc = params.to_c()
    return fn(C.byref(pc), *args)
  return fn(*args)


def thermostat(mode: int, tz: float, window: Tuple[float, float], comfort_now: bool, comfort_prev: int):
  """thermostat.py:114-148 + vav.py:229-243 -> (mode, damper, reheat valve); comfort_prev -1 = none."""
  damper, valve = np.zeros(1), np.zeros(1)
  mode = lib().sbo_dev_thermostat(mode, tz, window[0], window[1], int(comfort_now), int(comfort_prev),
                                  _d(damper), _d(valve))
  return mode, float(damper[0]), float(valve[0])


def boiler_dissipation(params: OracleParams, water_temp: float, outside_temp: float) -> float:
  pc = params.to_c()
  return lib().sbo_boiler_dissipation(C.byref(pc), water_temp, outside_temp)


class OracleBuilding:
  """One building: the oracle's counterpart of Simulator + SimulatorBuilding state."""

  def __init__(self, plan: OraclePlan, params: OracleParams, initial_temp: float,
               reset_temps: Optional[np.ndarray] = None):
    self.plan, self.params = plan, params
    self.initial_temp = float(initial_temp)
    self.reset_temps = (None if reset_temps is None
                        else np.ascontiguousarray(reset_temps, dtype=np.float64))
    N, Z = plan.n_cells, plan.Z
    self.temp = np.zeros(N)
    self.input_q = np.zeros(N)
    self.scratch = np.zeros(N)
    self.mode = np.zeros(Z, dtype=np.int32)
    self.damper = np.zeros(Z)
    self.valve = np.zeros(Z)
    self.zone_air_temp = np.zeros(Z)
    self._pc = params.to_c()
    self._s = _State()
    s = self._s
    s.temp, s.input_q, s.scratch = _d(self.temp), _d(self.input_q), _d(self.scratch)
    s.mode, s.damper, s.valve = _i(self.mode), _d(self.damper), _d(self.valve)
    s.zone_air_temp = _d(self.zone_air_temp)
    # thermostat.py:66-69, smart_device.py:71-72: construction state, survives reset()
    s.thermostat_has_prev = 0
    s.blr_has_action_ts = 0
    s.blr_action_ts = 0.0
    self.reset()

  def reset(self) -> None:
    rt = None if self.reset_temps is None else _d(self.reset_temps)
    lib().sbo_reset(self.plan.cptr(), C.byref(self._pc), C.byref(self._s),
                    self.initial_temp, rt)

  def observe_boiler(self, obs_ts: float) -> None:
    """One read of boiler.supply_water_temperature_sensor at ``obs_ts`` (boiler.py:146-217), e.g.
    Environment.reset()'s first observation (environment.py:1165-1176)."""
    lib().sbo_observe_boiler(C.byref(self._pc), C.byref(self._s), float(obs_ts))

  @property
  def state(self) -> _State:
    return self._s

  def grid(self) -> np.ndarray:
    return self.temp.reshape(self.plan.H, self.plan.W)

  def step(self, *, now_ts: float, t_amb_now: float, h_conv: float, t_amb_next: float,
           comfort_now: bool, comfort_prev: bool, comfort_next: bool, occupancy,
           e_price: float, e_carbon: float, g_price: float, g_carbon: float,
           action: Optional[Sequence[float]] = None, observe: bool = True, reject: bool = False,
           cool_sp: Optional[float] = None, damper_cmd: Optional[Sequence[float]] = None) -> dict:
    """One Environment._step (H2 ordering).  ``action`` = (boiler supply-water setpoint,
    AHU heating setpoint) in native units as the proto would carry them (fp32)."""
    Z = self.plan.Z
    occ = np.ascontiguousarray(np.broadcast_to(np.asarray(occupancy, dtype=np.float64), (Z,)))
    tz_pre, tz_post, qz = np.zeros(Z), np.zeros(Z), np.zeros(Z)
    si = _StepIn()
    si.now_ts = now_ts
    si.has_action = 0 if action is None else 1
    if action is not None:
      si.blr_setpoint = float(np.float32(action[0]))
      si.ahu_heat_sp = float(np.float32(action[1]))
    si.t_amb_now, si.h_conv, si.t_amb_next = t_amb_now, h_conv, t_amb_next
    si.comfort_now, si.comfort_prev, si.comfort_next = int(comfort_now), int(comfort_prev), int(comfort_next)
    si.occupancy = _d(occ)
    si.observe = int(observe)
    si.reject = int(reject)   # rejection_simulator_building.py:52-60
    if cool_sp is not None:
      si.has_cool_sp, si.ahu_cool_sp = 1, float(np.float32(cool_sp))
    if damper_cmd is not None:
      dc = np.ascontiguousarray(damper_cmd, dtype=np.float64)
      si.damper_cmd = _d(dc)
    si.e_price, si.e_carbon, si.g_price, si.g_carbon = e_price, e_carbon, g_price, g_carbon
    so = _StepOut()
    so.zone_temp_pre, so.zone_temp_post, so.q_zone = _d(tz_pre), _d(tz_post), _d(qz)
    lib().sbo_step(self.plan.cptr(), C.byref(self._pc), C.byref(self._s), C.byref(si),
                   C.byref(so))
    s = self._s
    return dict(
        n_sweeps=so.n_sweeps, converged=bool(so.converged), action_accepted=bool(so.action_accepted),
        t_supply_air=so.t_supply_air,
        recirc_pre=so.recirc_pre, zone_temp_pre=tz_pre, zone_temp_post=tz_post, q_zone=qz,
        blower_rate=so.blower_rate, ac_rate=so.ac_rate, gas_rate=so.gas_rate,
        pump_rate=so.pump_rate, reward=so.reward, reward_f64=so.reward_f64,
        productivity=so.productivity, norm_prod_regret=so.norm_prod_regret,
        norm_energy_cost=so.norm_energy_cost, norm_carbon=so.norm_carbon,
        ahu_flow=s.ahu_flow, ahu_count=s.ahu_count, blr_flow=s.blr_flow,
        blr_count=s.blr_count, blr_return_temp=s.blr_return_temp,
        blr_tank_temp=s.blr_tank_temp, blr_setpoint=s.blr_setpoint,
        ahu_heat_sp=s.ahu_heat_sp, ahu_cool_sp=s.ahu_cool_sp,
        damper=self.damper.copy(), valve=self.valve.copy(), mode=self.mode.copy(),
        zone_air_temp=self.zone_air_temp.copy())


class OracleBatch:
  """``nb`` independent buildings stepped with OpenMP (CPU baseline / in-run parity)."""

  def __init__(self, plan: OraclePlan, params: OracleParams, initial_temps: np.ndarray):
    self.plan, self.params = plan, params
    init = np.ascontiguousarray(initial_temps, dtype=np.float64)
    self.nb = init.shape[0]
    self.buildings: List[OracleBuilding] = []
    for b in range(self.nb):
      ob = OracleBuilding(plan, params, 0.0, reset_temps=init[b].reshape(-1))
      self.buildings.append(ob)
    self._states = (_State * self.nb)()
    self._outs = (_StepOut * self.nb)()
    self._pc = params.to_c()
    self.max_threads = lib().sbo_max_threads()

  def _sync_in(self):
    for b, ob in enumerate(self.buildings):
      C.memmove(C.byref(self._states[b]), C.byref(ob._s), C.sizeof(_State))

  def _sync_out(self):
    for b, ob in enumerate(self.buildings):
      C.memmove(C.byref(ob._s), C.byref(self._states[b]), C.sizeof(_State))

  def step(self, step_ins: Sequence[_StepIn], n_threads: int = 1):
    self._sync_in()
    per = 1 if len(step_ins) > 1 else 0
    arr = (_StepIn * len(step_ins))(*step_ins)
    lib().sbo_step_batch(self.plan.cptr(), C.byref(self._pc), self._states, arr, per,
                         self._outs, self.nb, n_threads)
    self._sync_out()
    return self._outs


def make_step_in(**kw) -> _StepIn:
  """Builds a raw step-input struct (keeps ``occupancy`` array alive on the struct)."""
  si = _StepIn()
  occ = np.ascontiguousarray(kw.pop("occupancy"), dtype=np.float64)
  si._occ_keepalive = occ
  si.occupancy = _d(occ)
  action = kw.pop("action", None)
  si.has_action = 0 if action is None else 1
  if action is not None:
    si.blr_setpoint = float(np.float32(action[0]))
    si.ahu_heat_sp = float(np.float32(action[1]))
  for key, val in kw.items():
    setattr(si, key, int(val) if isinstance(val, (bool, np.bool_)) else val)
  return si
