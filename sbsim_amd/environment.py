"""Vectorised drop-in for sbsim's RL environment: ``reset()`` / ``step(action)`` over B buildings.

Mirrors ``smart_control/environment/environment.py`` ``Environment`` (a TF-Agents
``PyEnvironment``) for the simulator-backed configuration of
``configs/resources/sb1/sim_config.gin``, with a leading batch axis:

    reference                                   here
    ---------                                   ----
    Environment._reset()  (:1165-1212)          BatchedEnvironment.reset()
    Environment._step(a)  (:1228-1370)          BatchedEnvironment.step(a)      a: [B, A] in [-1, 1]
    action_spec()/observation_spec() (:1215-1221)   same names; shapes without the batch axis
    steps_per_episode (:522), current_simulation_timestamp (:816), discount_factor

Every step is one call through the C ABI (include/sbsim_amd.h: sb_step = three HIP kernel
launches on the current torch stream); observations, rewards and actions stay in HBM as torch
tensors.  Calendar,
weather, occupancy and tariffs are resolved on the host once per step (all buildings share
the simulator clock) by ``sbsim_amd.host_inputs``.  Episode bookkeeping follows
``environment.py:427-435,1311-1368``: an episode is N transitions plus one terminal step;
a ``step`` after termination resets.  Action rejection (``RejectionSimulatorBuilding``)
and metrics writing are out of scope.
"""
from __future__ import annotations

import collections
import copy
import ctypes as C
import dataclasses
import datetime as dt
import math
import os
from typing import Dict, List, Mapping, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _ffi, host_inputs
from .floorplan import CompiledPlan, FloorPlan, Materials

TimeStep = collections.namedtuple("TimeStep", ["step_type", "reward", "discount", "observation"])
STEP_FIRST, STEP_MID, STEP_LAST = 0, 1, 2   # tf_agents.trajectories.time_step.StepType


@dataclasses.dataclass(frozen=True)
class ArraySpec:
  shape: Tuple[int, ...]
  dtype: np.dtype
  name: str
  minimum: Optional[float] = None
  maximum: Optional[float] = None


# air_handler.py:66-95, boiler.py:69-79, vav.py:54-64: observable fields in sorted order.
_AHU_FIELDS = ("cooling_request_count", "differential_pressure_setpoint",
               "discharge_fan_speed_percentage_command", "outside_air_flowrate_sensor",
               "outside_air_temperature_sensor", "supply_air_cooling_temperature_setpoint",
               "supply_air_flowrate_sensor", "supply_air_heating_temperature_setpoint",
               "supply_fan_speed_percentage_command")
_BOILER_FIELDS = ("heating_request_count", "supply_water_setpoint",
                  "supply_water_temperature_sensor")
_VAV_FIELDS = ("supply_air_damper_percentage_command", "supply_air_flowrate_setpoint",
               "zone_air_temperature_sensor")
_AUX_FIELDS = ("hod_cos_000", "hod_sin_000", "dow_cos_000", "dow_sin_000",
               "comfort_mode_now", "comfort_mode_soon", "num_occupants")
ACTION_NAMES = ("supply_water_setpoint", "supply_air_heating_temperature_setpoint")
# settable fields of the simulated devices (boiler.py:81-85, air_handler.py:97-104, vav.py:65-69) -> sb_action_kind
ACTION_KINDS = {
    "supply_water_setpoint": _ffi.SB_ACT_BOILER_SUPPLY_WATER_SETPOINT,
    "supply_air_heating_temperature_setpoint": _ffi.SB_ACT_AHU_SUPPLY_AIR_HEATING_SETPOINT,
    "supply_air_cooling_temperature_setpoint": _ffi.SB_ACT_AHU_SUPPLY_AIR_COOLING_SETPOINT,
    "supply_air_damper_percentage_command": _ffi.SB_ACT_VAV_SUPPLY_AIR_DAMPER_COMMAND,
}
ACTION_REJECTION_REWARD = float("-inf")   # environment.py:52
# sim_config.gin:164: the SB1 configuration's clock starts at a tz-aware UTC time, which the schedule
# converts to US/Pacific.  BatchedEnvironment's default start is the NAIVE 2023-07-06 07:00 of the
# reference's test scenarios and of BASELINE.json configs[0..1]; a naive time stamp is taken as UTC
# wall clock whatever `time_zone` says (setpoint_schedule.py:100-106).
SB1_START_TIMESTAMP = dt.datetime(2023, 7, 6, 7, 0, 0, tzinfo=dt.timezone.utc)


@dataclasses.dataclass
class SimConfig:
  """Scalar configuration: the constructor arguments of the reference's Simulator, Hvac
  devices, SetpointSchedule and SetpointEnergyCarbonRegretFunction.  ``sb1()`` gives
  ``configs/resources/sb1/sim_config.gin:23-250``."""
  time_step_sec: float = 300.0
  convergence_threshold: float = 0.1
  iteration_limit: int = 100
  cv_size_cm: float = 10.0
  floor_height_cm: float = 300.0
  initial_temp: float = 294.0
  materials: Materials = dataclasses.field(default_factory=Materials.sb1)
  # schedule
  morning_start_hour: int = 6
  evening_start_hour: int = 19
  comfort_temp_window: Tuple[float, float] = (294.0, 297.0)
  eco_temp_window: Tuple[float, float] = (289.0, 298.0)
  schedule_holidays: Tuple[int, ...] = ()
  time_zone: str = "US/Pacific"
  # devices
  vav_max_air_flow_rate: float = 0.035
  vav_reheat_max_water_flow_rate: float = 0.03
  ahu_recirculation: float = 0.3
  ahu_heating_air_temp_setpoint: float = 285.0
  ahu_cooling_air_temp_setpoint: float = 298.0
  ahu_fan_differential_pressure: float = 10000.0
  ahu_fan_efficiency: float = 0.9
  ahu_max_air_flow_rate: float = 8.67
  ahu_has_weather_sensor: bool = True
  boiler_reheat_water_setpoint: float = 360.0
  boiler_water_pump_differential_head: float = 6.0
  boiler_water_pump_efficiency: float = 0.98
  boiler_heating_rate: float = 0.5
  boiler_cooling_rate: float = 0.1
  boiler_convection_coefficient: float = 5.6
  boiler_tank_length: float = 2.0
  boiler_tank_radius: float = 0.5
  boiler_water_capacity: float = 1.5
  boiler_insulation_conductivity: float = 0.067
  boiler_insulation_thickness: float = 0.06
  # reward
  max_productivity_personhour_usd: float = 300.0
  min_productivity_personhour_usd: float = 100.0
  max_electricity_rate: float = 160000.0
  max_natural_gas_rate: float = 400000.0
  productivity_midpoint_delta: float = 0.5
  productivity_decay_stiffness: float = 4.3
  productivity_weight: float = 0.2
  energy_cost_weight: float = 0.4
  carbon_emission_weight: float = 0.4
  # action normalisation (bounded_action_normalizer.py; sim_config.gin:229-237)
  action_ranges: Tuple[Tuple[float, float], ...] = ((310.0, 355.0), (285.0, 300.0))
  # the action vector (environment.py:278-307,591-653: one column per (device, setpoint) of the
  # ActionConfig that has a normalizer): setpoint names, and the zone index for a VAV field.
  # Default: the SB1 set of sim_config.gin:239-242.  `action_ranges[i]` is column i's native range.
  action_names: Tuple[str, ...] = ACTION_NAMES
  action_zones: Tuple[int, ...] = ()

  @staticmethod
  def sb1() -> "SimConfig":
    return SimConfig()

  def schedule(self) -> host_inputs.SetpointSchedule:
    return host_inputs.SetpointSchedule(
        self.morning_start_hour, self.evening_start_hour, self.comfort_temp_window,
        self.eco_temp_window, set(self.schedule_holidays), self.time_zone)

  def to_params(self) -> _ffi.Params:
    p = _ffi.Params()
    p.dt, p.conv_threshold, p.iter_limit = self.time_step_sec, self.convergence_threshold, self.iteration_limit
    p.ahu_has_weather = int(self.ahu_has_weather_sensor)
    p.vav_max_air_flow, p.vav_max_water_flow = self.vav_max_air_flow_rate, self.vav_reheat_max_water_flow_rate
    p.ahu_recirc, p.ahu_heat_sp, p.ahu_cool_sp = (self.ahu_recirculation, self.ahu_heating_air_temp_setpoint,
                                                  self.ahu_cooling_air_temp_setpoint)
    p.ahu_dp, p.ahu_eff, p.ahu_max_flow = (self.ahu_fan_differential_pressure, self.ahu_fan_efficiency,
                                           self.ahu_max_air_flow_rate)
    p.blr_setpoint, p.blr_head, p.blr_pump_eff = (self.boiler_reheat_water_setpoint,
                                                  self.boiler_water_pump_differential_head,
                                                  self.boiler_water_pump_efficiency)
    p.blr_heating_rate, p.blr_cooling_rate = self.boiler_heating_rate, self.boiler_cooling_rate
    p.blr_conv, p.blr_len, p.blr_radius = (self.boiler_convection_coefficient, self.boiler_tank_length,
                                           self.boiler_tank_radius)
    p.blr_capacity, p.blr_ins_k, p.blr_ins_thick = (self.boiler_water_capacity,
                                                    self.boiler_insulation_conductivity,
                                                    self.boiler_insulation_thickness)
    p.comfort_lo, p.comfort_hi = self.comfort_temp_window
    p.eco_lo, p.eco_hi = self.eco_temp_window
    p.max_prod, p.min_prod = self.max_productivity_personhour_usd, self.min_productivity_personhour_usd
    p.max_elec, p.max_gas = self.max_electricity_rate, self.max_natural_gas_rate
    p.prod_delta, p.prod_stiff = self.productivity_midpoint_delta, self.productivity_decay_stiffness
    p.w_prod, p.w_cost, p.w_carbon = self.productivity_weight, self.energy_cost_weight, self.carbon_emission_weight
    n = len(self.action_names)
    if n != len(self.action_ranges) or n < 1:
      raise ValueError("one native range per action, at least one action")
    for name in self.action_names:
      if name not in ACTION_KINDS:   # simulator_building.py:236-251: REJECTED_NOT_ENABLED_OR_AVAILABLE at run time
        raise ValueError(f"{name!r} is not a settable field of the simulated devices")
    p.n_actions = n
    # host arrays the struct points at (sb_create copies them; they live as long as the Params object)
    kinds = (C.c_int32 * n)(*[ACTION_KINDS[name] for name in self.action_names])
    zones = (C.c_int32 * n)(*[self.action_zones[i] if i < len(self.action_zones) else 0 for i in range(n)])
    los = (C.c_double * n)(*[float(lo) for lo, _ in self.action_ranges])
    his = (C.c_double * n)(*[float(hi) for _, hi in self.action_ranges])
    p.act_kind, p.act_zone = C.cast(kinds, C.POINTER(C.c_int32)), C.cast(zones, C.POINTER(C.c_int32))
    p.act_lo, p.act_hi = C.cast(los, C.POINTER(C.c_double)), C.cast(his, C.POINTER(C.c_double))
    p._keep_alive = (kinds, zones, los, his)
    return p


# Observation normalisation constants of sim_config.gin:252-590 that apply to simulated
# devices: field name -> (sample_mean, sample_variance).  Unknown fields get (0, 1)
# (observation_normalizer.py:57-66).
SB1_OBSERVATION_NORMALIZATION: Dict[str, Tuple[float, float]] = {
    "differential_pressure_setpoint": (83810.269540, 14889040.603647),
    "outside_air_flowrate_sensor": (3.701930, 20.300565),
    "outside_air_temperature_sensor": (291.244931, 12.904175),
    "supply_air_flowrate_sensor": (177.520026, 50499.153481),
    "supply_fan_speed_percentage_command": (26.543748, 575.094979),
    "supply_water_setpoint": (310.0, 2500.0),
    "supply_water_temperature_sensor": (321.520315, 658.413066),
    "zone_air_temperature_sensor": (190.0, 408.113303),
}


def observation_field_names(zone_names: Sequence[str], ahu_has_weather: bool,
                            ahu_id: str = "air_handler_id", boiler_id: str = "boiler_id"):
  """environment.py:543-553,783-813: (device_id, field) sorted, then auxiliary features.
  Returns (names, col_ahu, col_boiler, col_zone[Z], col_aux)."""
  devices = [(ahu_id, "ahu", -1), (boiler_id, "blr", -1)]
  devices += [(f"vav_{z}", "vav", i) for i, z in enumerate(zone_names)]
  names, col_ahu, col_blr, col_zone = [], -1, -1, [0] * len(zone_names)
  for dev_id, kind, zi in sorted(devices, key=lambda d: d[0]):
    if kind == "ahu":
      col_ahu = len(names)
      fields = [f for f in _AHU_FIELDS if ahu_has_weather or f != "outside_air_temperature_sensor"]
    elif kind == "blr":
      col_blr = len(names)
      fields = _BOILER_FIELDS
    else:
      col_zone[zi] = len(names)
      fields = _VAV_FIELDS
    names += [f"{dev_id}/{f}" for f in fields]
  col_aux = len(names)
  names += list(_AUX_FIELDS)
  return names, col_ahu, col_blr, col_zone, col_aux


def histogram_reducer_layout(source_names: Sequence[str],
                             histogram_parameters: Sequence[Tuple[str, Sequence[float]]]):
  """Output layout of the observation HistogramReducer (utils/histogram_reducer.py:204-471) in
  the field order of ``Environment._get_observation_spec_histogram_reducer``
  (environment.py:731-777): devices sorted by id, fields sorted; a reduced measurement
  contributes one column per bin (``<measurement>_h_<bin:.2f>``) where it is first met, every
  other field passes through.  ``source_names`` are the device fields ("device/field") in that
  sorted order.  Returns (names, src_dest, hist_col, hist_off, hist_bins)."""
  bins = {}
  for name, values in histogram_parameters:
    b = [float(v) for v in values]
    if len(b) < 1 or any(b[i] >= b[i + 1] for i in range(len(b) - 1)):
      raise ValueError(f"histogram bins of {name} must be ascending")
    bins[name] = b
  names: list = []
  src_dest = [0] * len(source_names)
  feature_index, hist_col, hist_off, hist_bins = {}, [], [0], []
  for i, full in enumerate(source_names):
    _, meas = full.split("/", 1)
    if meas in bins:
      if meas not in feature_index:
        feature_index[meas] = len(hist_col)
        hist_col.append(len(names))
        hist_bins += bins[meas]
        hist_off.append(len(hist_bins))
        names += [f"{meas}_h_{v:.2f}" for v in bins[meas]]
      src_dest[i] = -(feature_index[meas] + 1)
    else:
      src_dest[i] = len(names)
      names.append(full)
  return names, src_dest, hist_col, hist_off, hist_bins


class BatchedSimulator:
  """Thin owner of the C-ABI handle: B building instances of one floor plan on one GPU.

  Counterpart of ``SimulatorBuilding`` + ``Simulator`` + reward function for the batched
  path (simulator_building.py:44-315); tensors in, tensors out."""

  def __init__(self, plan: FloorPlan, config: SimConfig, n_buildings: int, h_conv: float,
               device: int = 0,
               observation_normalization: Optional[Mapping[str, Tuple[float, float]]] = None,
               zone_names: Optional[Sequence[str]] = None, orientation: str = "auto",
               histogram_parameters: Optional[Sequence[Tuple[str, Sequence[float]]]] = None,
               normalize_reduce: bool = False):
    self._lib = _ffi.load()
    if not torch.cuda.is_available():
      raise _ffi.SbsimError("sbsim_amd needs a HIP device (MI355X); there is no CPU path")
    self.plan, self.config, self.B, self.device = plan, config, int(n_buildings), int(device)
    self.n_actions = len(config.action_names)
    H0, W0 = plan.shape
    if orientation not in ("auto", "rows", "columns"):
      raise ValueError("orientation must be 'auto', 'rows' or 'columns'")
    self.Z, self.H, self.W = plan.n_zones, H0, W0
    zone_names = list(zone_names or plan.zone_names or [f"room_{i + 1}" for i in range(self.Z)])
    self.zone_names = zone_names
    (source_names, col_ahu, col_blr, col_zone, n_src) = observation_field_names(
        zone_names, config.ahu_has_weather_sensor)
    aux_names, source_names = source_names[n_src:], source_names[:n_src]
    # optional HistogramReducer (environment.py:731-777,1032-1071): device-side binning
    if histogram_parameters:
      out_names, src_dest, hist_col, hist_off, hist_bins = histogram_reducer_layout(
          source_names, histogram_parameters)
    else:
      out_names, src_dest, hist_col, hist_off, hist_bins = list(source_names), None, [], [0], []
    col_aux = len(out_names)
    self.field_names = out_names + list(aux_names)
    self.source_names = source_names
    self.O = len(self.field_names)

    def describe(p: FloorPlan):
      cp = p.compile(config.time_step_sec, h_conv)
      keep = dict(cls=np.ascontiguousarray(cp.cell_class), coef=np.ascontiguousarray(cp.class_coef),
                  czone=np.ascontiguousarray(cp.class_zone), zoff=np.ascontiguousarray(cp.zone_off),
                  zcells=np.ascontiguousarray(cp.zone_cells))
      desc = _ffi.PlanDesc(cp.H, cp.W, cp.Z, cp.n_classes,
                           keep["cls"].ctypes.data_as(C.POINTER(C.c_uint8)),
                           keep["coef"].ctypes.data_as(_ffi._dp), keep["czone"].ctypes.data_as(_ffi._ip),
                           keep["zoff"].ctypes.data_as(_ffi._ip), keep["zcells"].ctypes.data_as(_ffi._ip))
      info = _ffi.LaunchInfo()
      rc = self._lib.sb_plan_info(C.byref(desc), self.O, self.B, C.byref(info))
      return cp, keep, desc, (rc, info)

    # device-side orientation (internal to the library; reset()/temps() convert, so callers
    # always see the reference's [H, W] layout): the library reports, per orientation, which
    # step kernel it would use, how many wavefront steps one sweep takes and on how many wavefronts
    if orientation == "auto" and os.environ.get("SBSIM_ORIENTATION") in ("rows", "columns"):   # developer knob
      orientation = os.environ["SBSIM_ORIENTATION"]
    cands = {"rows": [False], "columns": [True], "auto": [False, True]}[orientation]
    best = None
    for tr in cands:
      cand = describe(plan.transposed() if tr else plan)
      rc, info = cand[3]
      # SIMD steps per building-sweep: a kernel that spreads a building over two wavefronts holds half as many
      pref = {1: 0, 0: 1, 2: 2}.get(info.path, 3) if rc == 0 else 9   # registers, then LDS, then the streaming kernel
      key = (rc != 0, pref, info.sweep_steps * max(info.waves_per_building, 1) if rc == 0 else 0)
      if best is None or key < best[0]:
        best = (key, tr, cand)
    self.transposed = best[1]
    self.compiled, keep, pd_, _ = best[2]
    norm = dict(observation_normalization or {})
    mean = np.zeros(max(self.O, n_src))     # by source index
    sigma = np.ones(max(self.O, n_src))
    for i, name in enumerate(source_names):
      mu, var = norm.get(name.split("/", 1)[1], (0.0, 1.0))
      # ContinuousVariableInfo fields are proto floats (smart_control_normalization.proto)
      mu, var = float(np.float32(mu)), float(np.float32(var))
      mean[i] = mu
      sigma[i] = math.sqrt(var) if var > 0.0 else 0.0
    self._keep = dict(keep, colz=np.asarray(col_zone, dtype=np.int32), mean=mean, sigma=sigma)
    self._keep.update(src_dest=np.asarray(src_dest if src_dest is not None else [0], dtype=np.int32),
                      hist_col=np.asarray(hist_col or [0], dtype=np.int32),
                      hist_off=np.asarray(hist_off, dtype=np.int32),
                      hist_bins=np.asarray(hist_bins or [0.0], dtype=np.float64))
    kp = self._keep
    ol = _ffi.ObsLayout(self.O, col_ahu, col_blr, col_aux, kp["colz"].ctypes.data_as(_ffi._ip),
                        mean.ctypes.data_as(_ffi._dp), sigma.ctypes.data_as(_ffi._dp),
                        n_src if src_dest is not None else 0, len(hist_col),
                        kp["src_dest"].ctypes.data_as(_ffi._ip), kp["hist_col"].ctypes.data_as(_ffi._ip),
                        kp["hist_off"].ctypes.data_as(_ffi._ip), kp["hist_bins"].ctypes.data_as(_ffi._dp),
                        1 if normalize_reduce else 0)
    params = config.to_params()
    h = C.c_void_p()
    with torch.cuda.device(self.device):
      _ffi.check(self._lib.sb_create(C.byref(pd_), C.byref(params), C.byref(ol), self.B, self.device,
                                     C.byref(h)), "sb_create")
    self._h = h
    self.tdev = torch.device("cuda", self.device)
    info = _ffi.LaunchInfo()
    _ffi.check(self._lib.sb_get_launch_info(self._h, C.byref(info)), "sb_get_launch_info")
    self.launch_info = {f[0]: getattr(info, f[0]) for f in _ffi.LaunchInfo._fields_}
    self.sweep_events = None   # a list: step() appends (start, end) HIP events of every sweep-kernel launch (bench.py)

  def close(self) -> None:
    if getattr(self, "_h", None):
      self._lib.sb_destroy(self._h)
      self._h = None

  def __del__(self):
    try:
      self.close()
    except Exception:  # pylint: disable=broad-except
      pass

  def _stream(self) -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream(self.tdev).cuda_stream)

  # ---- Simulator.reset() ----
  def reset(self, initial_temp: Optional[float] = None, temps: Optional[torch.Tensor] = None) -> None:
    ptr = None
    if temps is not None:
      if temps.dtype != torch.float64 or tuple(temps.shape) != (self.B, self.H * self.W) or not temps.is_contiguous():
        raise ValueError("temps must be a contiguous float64 [B, H*W] tensor")
      if self.transposed:
        temps = temps.view(self.B, self.H, self.W).transpose(1, 2).contiguous()
      ptr = C.c_void_p(temps.data_ptr())
    t0 = self.config.initial_temp if initial_temp is None else float(initial_temp)
    _ffi.check(self._lib.sb_reset(self._h, t0, ptr, self._stream()), "sb_reset")

  def occupancy_attach(self, zone_assignment: int, hours: Sequence[int], time_step_sec: float, seed: int,
                       first_building: int = 0) -> None:
    """sb_occupancy_attach: one RandomizedArrivalDepartureOccupancy per building on the device."""
    cfg = _ffi.OccupancyConfig(int(zone_assignment), int(hours[0]), int(hours[1]), int(hours[2]), int(hours[3]),
                               float(time_step_sec), int(seed) & 0xFFFFFFFFFFFFFFFF, int(first_building))
    _ffi.check(self._lib.sb_occupancy_attach(self._h, C.byref(cfg)), "sb_occupancy_attach")

  def convection_attach(self, p: float, distance: int, seed: int, first_building: int = 0) -> None:
    """sb_convection_attach: StochasticConvectionSimulator(p, distance, seed) after every FD update
    (stochastic_convection_simulator.py:62-145), counter-based draws per building."""
    _ffi.check(self._lib.sb_convection_attach(self._h, float(p), int(distance), int(seed) & 0xFFFFFFFFFFFFFFFF,
                                              int(first_building), int(self.transposed)), "sb_convection_attach")

  def occupancy_peek(self, local_hour: int, is_work_day: bool, count: Optional[torch.Tensor] = None,
                     total: Optional[torch.Tensor] = None) -> None:
    """sb_occupancy_peek: advances every occupant once; count [B, Z] / total [B] float32."""
    for t, shape in ((count, (self.B, self.Z)), (total, (self.B,))):
      if t is not None and (t.dtype != torch.float32 or tuple(t.shape) != shape or not t.is_contiguous()):
        raise ValueError(f"occupancy output must be a contiguous float32 {shape} tensor")
    _ffi.check(self._lib.sb_occupancy_peek(
        self._h, int(local_hour), int(bool(is_work_day)), C.c_void_p(count.data_ptr()) if count is not None else None,
        C.c_void_p(total.data_ptr()) if total is not None else None, self._stream()), "sb_occupancy_peek")

  def observe(self, aux: Sequence[float], t_amb, out: Optional[torch.Tensor] = None,
              num_occupants: Optional[torch.Tensor] = None, occupancy_norm: float = 0.0) -> torch.Tensor:
    """t_amb: one ambient temperature, or a float64 [B] device tensor (per-building weather).
    num_occupants: optional float32 [B] device tensor (per-building occupancy feature)."""
    out = out if out is not None else torch.empty((self.B, self.O), dtype=torch.float32, device=self.tdev)
    a = (C.c_float * _ffi.SB_NUM_AUX)(*[float(x) for x in aux])
    per_b = None
    if isinstance(t_amb, torch.Tensor):
      if t_amb.dtype != torch.float64 or tuple(t_amb.shape) != (self.B,):
        raise ValueError("per-building t_amb must be a float64 [B] tensor")
      per_b, t_amb = C.c_void_p(t_amb.contiguous().data_ptr()), 0.0
    _ffi.check(self._lib.sb_observe_occupancy(
        self._h, a, float(t_amb), per_b, C.c_void_p(num_occupants.data_ptr()) if num_occupants is not None else None,
        float(occupancy_norm), C.c_void_p(out.data_ptr()), self._stream()), "sb_observe")
    return out

  def step(self, actions: Optional[torch.Tensor], step_in: _ffi.StepIn, obs: torch.Tensor,
           reward: torch.Tensor, info: Optional[torch.Tensor] = None, phases: int = 7) -> None:
    """One Environment._step for every building (sb_step).  `phases` selects a subset of its
    three launches (1 = device algebra before the sweep, 2 = sweep kernel, 4 = reward /
    observation) -- a measurement aid; a step is complete once all three ran in order."""
    if actions is not None:
      if actions.dtype != torch.float32 or tuple(actions.shape) != (self.B, self.n_actions):
        raise ValueError(f"actions must be float32 [{self.B}, {self.n_actions}]")
      if not actions.is_contiguous():
        actions = actions.contiguous()
    args = (self._h, C.c_void_p(actions.data_ptr()) if actions is not None else None, C.byref(step_in),
            C.c_void_p(obs.data_ptr()) if obs is not None else None, C.c_void_p(reward.data_ptr()),
            C.c_void_p(info.data_ptr()) if info is not None else None, self._stream())
    if self.sweep_events is not None and phases == 7:
      # measurement aid (bench.py): the step's three launches one by one, HIP events around the sweep kernel on its stream
      e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
      _ffi.check(self._lib.sb_step_phases(*args, 1), "sb_step")
      e0.record()
      _ffi.check(self._lib.sb_step_phases(*args, 2), "sb_step")
      e1.record()
      _ffi.check(self._lib.sb_step_phases(*args, 4), "sb_step")
      self.sweep_events.append((e0, e1))
      return
    _ffi.check(self._lib.sb_step_phases(*args, int(phases)), "sb_step")

  # ---- parity taps ----
  def _get(self, fn, shape, dtype):
    out = torch.empty(shape, dtype=dtype, device=self.tdev)
    _ffi.check(fn(self._h, C.c_void_p(out.data_ptr()), self._stream()), fn.__name__)
    return out

  def temps(self) -> torch.Tensor:
    if self.transposed:
      t = self._get(self._lib.sb_get_temps, (self.B, self.W, self.H), torch.float64)
      return t.transpose(1, 2).contiguous()
    return self._get(self._lib.sb_get_temps, (self.B, self.H, self.W), torch.float64)

  def set_temps(self, temps: torch.Tensor) -> None:
    """sb_set_temps: building.temp <- temps ([B, H, W] float64 on the device) between two steps; every
    device state stays (what Building.apply_convection does to the reference's array, building.py:891-893)."""
    if temps.dtype != torch.float64 or tuple(temps.shape) != (self.B, self.H, self.W):
      raise ValueError(f"temps must be float64 [{self.B}, {self.H}, {self.W}]")
    t = temps.transpose(1, 2).contiguous() if self.transposed else temps.contiguous()
    _ffi.check(self._lib.sb_set_temps(self._h, C.c_void_p(t.data_ptr()), self._stream()), "sb_set_temps")

  def zone_temps(self) -> torch.Tensor:
    return self._get(self._lib.sb_get_zone_temps, (self.B, self.Z), torch.float64)

  def scalars(self) -> torch.Tensor:
    return self._get(self._lib.sb_get_scalars, (self.B, _ffi.SB_NUM_SCALARS), torch.float64)

  def modes(self) -> torch.Tensor:
    return self._get(self._lib.sb_get_modes, (self.B, self.Z), torch.int32)

  def zone_power(self) -> torch.Tensor:
    return self._get(self._lib.sb_get_zone_power, (self.B, self.Z), torch.float64)


class BatchedEnvironment:
  """B lock-stepped sbsim environments behind ``reset()`` / ``step(action)``.

  ``start_timestamp``: naive by default (see SB1_START_TIMESTAMP for the SB1 configuration's own).
  The TimeStep tensors returned by ``reset`` / ``step`` are the environment's own device buffers and
  are overwritten by the next call: clone what must outlive a step (replay buffers, n-step returns)."""

  def __init__(self, plan: FloorPlan, n_buildings: int, config: Optional[SimConfig] = None,
               weather=None, occupancy=None, start_timestamp=dt.datetime(2023, 7, 6, 7, 0, 0),
               num_days_in_episode: float = 3, discount_factor: float = 1.0, device: int = 0,
               observation_normalization: Optional[Mapping[str, Tuple[float, float]]] = None,
               occupancy_normalization_constant: float = 0.0, holiday_calendar="us",
               electricity_energy_cost=None, natural_gas_energy_cost=None, collect_info: bool = False,
               observation_histogram_parameters: Optional[Sequence[Tuple[str, Sequence[float]]]] = None,
               normalize_reduce: bool = False, convection_simulator=None):
    if discount_factor <= 0 or discount_factor > 1:
      raise ValueError("Discount factor must be in (0,1]")   # environment.py:454-455
    self.config = config or SimConfig.sb1()
    self.weather = weather or host_inputs.WeatherController(273.0, 283.0, convection_coefficient=100.0)
    self.occupancy = occupancy or host_inputs.StepFunctionOccupancy(
        dt.timedelta(hours=9), dt.timedelta(hours=17), 10.0, 0.1, holiday_calendar=holiday_calendar)
    self.schedule = self.config.schedule()
    self.electricity = electricity_energy_cost or host_inputs.ElectricityEnergyCost(
        holiday_calendar=holiday_calendar)
    self.gas = natural_gas_energy_cost or host_inputs.NaturalGasEnergyCost()
    self.discount_factor = float(discount_factor)
    self._start_timestamp = host_inputs.as_datetime(start_timestamp)
    self._occ_norm = float(occupancy_normalization_constant)
    h_conv = self.weather.get_air_convection_coefficient(self._start_timestamp)
    self.sim = BatchedSimulator(plan, self.config, n_buildings, h_conv, device,
                                observation_normalization,
                                histogram_parameters=observation_histogram_parameters,
                                normalize_reduce=normalize_reduce)
    self.batch_size = self.sim.B
    self._weather_lohi = self._weather_replay = None
    if isinstance(self.weather, host_inputs.BatchedReplayWeather):
      if self.weather.offsets_sec.shape[0] != self.batch_size:
        raise ValueError("BatchedReplayWeather needs one offset per building")
      td = lambda a: torch.tensor(a, dtype=torch.float64, device=self.sim.tdev).contiguous()
      self._weather_replay = (td(self.weather.times), td(self.weather.temps_f), td(self.weather.offsets_sec))
    if isinstance(self.weather, host_inputs.BatchedSinusoidWeather):
      if self.weather.low.shape[0] != self.batch_size:
        raise ValueError("BatchedSinusoidWeather needs one (low, high) pair per building")
      self._weather_lohi = torch.tensor(np.stack([self.weather.low, self.weather.high], axis=1),
                                        dtype=torch.float64, device=self.sim.tdev).contiguous()
    if convection_simulator is not None:   # simulator_flexible_floor_plan.py:71,156
      c = convection_simulator
      self.sim.convection_attach(c.p, c.distance, c.seed or 0, getattr(c, "first_building", 0))
    self._occ_count = self._occ_total = None
    if isinstance(self.occupancy, host_inputs.BatchedRandomizedArrivalDepartureOccupancy):
      o = self.occupancy   # per-building occupants live on the device
      self.sim.occupancy_attach(o.zone_assignment, o.hours, o.time_step_sec, o.seed or 0, o.first_building)
      self._occ_count = torch.zeros((self.batch_size, self.sim.Z), dtype=torch.float32, device=self.sim.tdev)
      self._occ_total = torch.zeros((self.batch_size,), dtype=torch.float32, device=self.sim.tdev)
    self._step_interval = dt.timedelta(seconds=self.config.time_step_sec)
    # environment.py:427-435
    self._num_timesteps_in_episode = int(dt.timedelta(days=num_days_in_episode) / self._step_interval)
    self._action_spec = ArraySpec((self.sim.n_actions,), np.dtype(np.float32), "action", -1.0, 1.0)
    self._observation_spec = ArraySpec((self.sim.O,), np.dtype(np.float32), "observation")
    self.field_names = self.sim.field_names
    dev = self.sim.tdev
    self._obs = torch.zeros((self.batch_size, self.sim.O), dtype=torch.float32, device=dev)
    self._reward = torch.zeros((self.batch_size,), dtype=torch.float32, device=dev)
    self._info = (torch.zeros((self.batch_size, _ffi.SB_INFO_STRIDE), dtype=torch.float32, device=dev)
                  if collect_info else None)
    self._discount = torch.full((self.batch_size,), self.discount_factor, dtype=torch.float32, device=dev)
    self._zero = torch.zeros((self.batch_size,), dtype=torch.float32, device=dev)
    self._episode_ended = False
    self._step_count = 0
    self._episode_count = 0
    self._prev_thermostat_ts = None   # Thermostat._previous_timestamp survives reset()
    self._now = self._start_timestamp
    self._needs_reset = True

  # ---- specs / properties (environment.py:522-540,1215-1221) ----
  def action_spec(self) -> ArraySpec:
    return self._action_spec

  def observation_spec(self) -> ArraySpec:
    return self._observation_spec

  @property
  def batched(self) -> bool:
    return True

  @property
  def steps_per_episode(self) -> int:
    return self._num_timesteps_in_episode

  @property
  def current_simulation_timestamp(self) -> dt.datetime:
    return self._now

  @property
  def info(self) -> Optional[torch.Tensor]:
    return self._info

  # ---- host-side step inputs ----
  def _aux(self, ts: dt.datetime):
    """environment.py:916-956 auxiliary features at the observation time."""
    hod = host_inputs.get_radian_time(ts, hour_of_day=True)
    dow = host_inputs.get_radian_time(ts, hour_of_day=False)
    n_occ = 0.0   # simulator_building.py:305-315
    if self._occ_total is not None:   # per building, on the device: overrides aux[6]
      t5 = ts - dt.timedelta(minutes=5)
      self.sim.occupancy_peek(self.occupancy.local(t5).hour, self.occupancy.is_work_day(t5), None, self._occ_total)
    elif isinstance(self.occupancy, host_inputs.StepFunctionOccupancy):
      v = self.occupancy.average_zone_occupancy("", ts - dt.timedelta(minutes=5), ts)   # stateless, the same for
      for _ in range(self.sim.Z):                                                        # every zone: one query,
        n_occ += v                                                                       # the reference's sum
    else:
      for z in range(self.sim.Z):
        n_occ += self.occupancy.average_zone_occupancy(self.sim.zone_names[z], ts - dt.timedelta(minutes=5), ts)
    n_occ = int(n_occ)
    return [np.float32(np.cos(hod)), np.float32(np.sin(hod)), np.float32(np.cos(dow)),
            np.float32(np.sin(dow)), np.float32(self.schedule.is_comfort_mode(ts)),
            np.float32(self.schedule.is_comfort_mode(ts + dt.timedelta(minutes=60))),
            np.float32((n_occ - self._occ_norm) / (self._occ_norm + 1))]

  def make_step_in(self, ts: dt.datetime, has_action: bool = True) -> _ffi.StepIn:
    nxt = ts + self._step_interval
    si = _ffi.StepIn()
    if self._weather_replay is not None:   # per-building replay weather, interpolated on the device
      tt, tf, off = self._weather_replay
      si.weather_times_dev, si.weather_tempf_dev, si.weather_offset_dev = tt.data_ptr(), tf.data_ptr(), off.data_ptr()
      si.weather_n = int(tt.shape[0])
      si.weather_t_now, si.weather_t_next = self.weather.query_time(ts), self.weather.query_time(nxt)
    elif isinstance(self.weather, host_inputs.BatchedSinusoidWeather):   # per-building weather, on the device
      si.weather_lohi_dev = self._weather_lohi.data_ptr()
      si.weather_f_now, si.weather_f_next = self.weather.factor(ts), self.weather.factor(nxt)
    else:
      si.t_amb_now = self.weather.get_current_temp(ts)
      si.t_amb_next = self.weather.get_current_temp(nxt)
    si.comfort_now = int(self.schedule.is_comfort_mode(ts))
    si.comfort_prev = (-1 if self._prev_thermostat_ts is None
                       else int(self.schedule.is_comfort_mode(self._prev_thermostat_ts)))
    si.comfort_next = int(self.schedule.is_comfort_mode(nxt))
    si.has_action = int(has_action)
    # the reference asks the occupancy model in this order (environment.py:1310-1330):
    # _get_observation (num_occupants), then _get_reward (reward_info) -- it matters for the
    # randomized model, whose every query advances the occupants
    for i, v in enumerate(self._aux(nxt)):
      si.aux[i] = float(v)
    if self._occ_count is not None:
      self.sim.occupancy_peek(self.occupancy.local(nxt).hour, self.occupancy.is_work_day(nxt), self._occ_count, None)
      si.occupancy_bz_dev = self._occ_count.data_ptr()
      si.num_occupants_dev = self._occ_total.data_ptr()
      si.occupancy_norm = self._occ_norm
    elif isinstance(self.occupancy, host_inputs.RandomizedArrivalDepartureOccupancy):
      occ = [self.occupancy.average_zone_occupancy(self.sim.zone_names[z], nxt, nxt + self._step_interval)
             for z in range(self.sim.Z)]   # one shared instance: per-zone values, same for every building
      self._occ_zone = torch.tensor(occ, dtype=torch.float64, device=self.sim.tdev)
      si.occupancy_dev = self._occ_zone.data_ptr()
    else:
      si.occupancy = self.occupancy.average_zone_occupancy("", nxt, nxt + self._step_interval)
    start_utc = host_inputs.reward_start_time_utc(nxt)
    si.e_price, si.e_carbon = self.electricity.rates(start_utc)
    si.g_price, si.g_carbon = self.gas.rates(start_utc)
    return si

  # ---- PyEnvironment API ----
  def reset(self) -> TimeStep:
    """environment.py:1165-1212."""
    self.sim.reset()
    self._now = self._start_timestamp
    self._episode_ended = False
    self._episode_count += 1
    self._step_count = 0
    self._needs_reset = False
    if isinstance(self.weather, (host_inputs.BatchedSinusoidWeather, host_inputs.BatchedReplayWeather)):
      t_amb = torch.tensor(self.weather.temps(self._now), dtype=torch.float64, device=self.sim.tdev)
    else:
      t_amb = self.weather.get_current_temp(self._now)
    self.sim.observe(self._aux(self._now), t_amb, self._obs, self._occ_total, self._occ_norm)
    first = torch.full((self.batch_size,), STEP_FIRST, dtype=torch.int32, device=self.sim.tdev)
    return TimeStep(first, self._zero, torch.ones_like(self._discount), self._obs)

  def step(self, action: torch.Tensor, rejected: Optional[torch.Tensor] = None) -> TimeStep:
    """environment.py:1228-1370.  ``rejected``: optional bool / uint8 [B] on the device -- buildings
    whose BaseBuilding.request_action raised (environment.py:1266-1309; RejectionSimulatorBuilding):
    they skip setup_step_sim and the actions, step and observe like the others, and return the
    reward ``ACTION_REJECTION_REWARD`` (-inf).

    The returned TimeStep's tensors are the environment's own buffers: the next ``step`` / ``reset``
    overwrites them (clone what you keep across steps)."""
    if self._needs_reset or self._episode_ended:
      return self.reset()
    si = self.make_step_in(self._now)
    if rejected is not None:
      if tuple(rejected.shape) != (self.batch_size,) or rejected.device != self.sim.tdev:
        raise ValueError(f"rejected must be a [{self.batch_size}] tensor on {self.sim.tdev}")
      self._rejected = rejected.to(torch.uint8).contiguous()
      si.reject_dev = self._rejected.data_ptr()
    elif getattr(self, "_rejected", None) is not None:
      # once buildings may skip thermostat updates, "the previous update" stays per building on the
      # device (scal[18]): the host's comfort_prev assumes every building updated at the last step
      if getattr(self, "_no_reject", None) is None:
        self._no_reject = torch.zeros((self.batch_size,), dtype=torch.uint8, device=self.sim.tdev)
      si.reject_dev = self._no_reject.data_ptr()
    self.sim.step(action, si, self._obs, self._reward, self._info)
    self._prev_thermostat_ts = self._now
    self._now = self._now + self._step_interval
    self._episode_ended = self._step_count >= self._num_timesteps_in_episode   # :1366-1368
    dev = self.sim.tdev
    if self._episode_ended:
      last = torch.full((self.batch_size,), STEP_LAST, dtype=torch.int32, device=dev)
      return TimeStep(last, self._reward, self._zero, self._obs)
    self._step_count += 1
    mid = torch.full((self.batch_size,), STEP_MID, dtype=torch.int32, device=dev)
    return TimeStep(mid, self._reward, self._discount, self._obs)

  def close(self) -> None:
    self.sim.close()


class MixedBatchedEnvironment:
  """One vectorised environment over SEVERAL floor-plan classes (BASELINE.json configs[2]: a batch mixed over
  floor plans with different zone counts): the same ``reset()`` / ``step(action)`` -> TimeStep contract as
  ``BatchedEnvironment`` over the concatenated batch.

      env = MixedBatchedEnvironment([(plan_a, 21845), (plan_b, 21845), (plan_c, 21845)], holiday_calendar="us")
      ts = env.reset()
      ts = env.step(actions)            # actions [B_total, n_actions] in HBM, buildings in class order

  Inside: one ``BatchedEnvironment`` (one library handle, the sweep kernel the planner picks for that plan)
  per class, each on its own HIP stream, so that the classes' launches overlap on the chip; ``step`` makes
  every class stream wait for the caller's stream (the actions) and the caller's stream wait for every class.
  The classes share the simulator clock and the episode length, so they step and end together.

  Observations: a class's observation row has ``3 * zones + 19`` fields, so rows differ in width.  The
  mixed observation is ``[B_total, max width]``; a building's row holds its class's fields first (the layout
  of ``BatchedEnvironment.field_names`` of its class, ``self.envs[k].field_names``) and zeros after
  ``self.observation_widths[k]``; ``self.class_of_building`` ([B_total] int32 in HBM) and ``self.slices``
  say which class a building belongs to.  Every keyword argument goes to each class's ``BatchedEnvironment``;
  a seeded per-building generator among them (``convection_simulator``, randomized ``occupancy``) gets its
  ``first_building`` moved to the class's first GLOBAL building on this rank, so every building of the mixed batch
  draws from its own stream whatever the sharding.

  Several GPUs (``rank``, ``world``): the counts in ``classes`` are the GLOBAL ones and every class
  is block-partitioned over the ranks on its own (``sbsim_amd.distributed.class_shard_ranges``), so each rank
  holds the same class mix -- the classes need very different numbers of sweeps per step, a rank with only the
  large class would set the pace.  ``self.class_totals`` / ``self.class_ranges`` say which global buildings of
  each class this rank holds; ``distributed.gather_returns_by_class`` puts per-building results back into
  global (class-major) order."""

  def __init__(self, classes: Sequence[Tuple[FloorPlan, int]], device: int = 0, rank: int = 0, world: int = 1,
               **env_kwargs):
    if not classes:
      raise ValueError("MixedBatchedEnvironment needs at least one (floor plan, number of buildings) class")
    from . import distributed as _sd
    self.device = int(device)
    self.tdev = torch.device("cuda", self.device)
    self.envs: List[BatchedEnvironment] = []
    self.streams: List[torch.cuda.Stream] = []
    self.slices: List[Tuple[int, int]] = []
    self.rank, self.world = int(rank), int(world)
    self.class_totals = [int(n) for _, n in classes]
    self.class_ranges = _sd.class_shard_ranges(self.class_totals, self.rank, self.world)
    # the same test on every rank (a rank that raised alone would leave the others in their first collective)
    if any(n < self.world for n in self.class_totals):
      raise ValueError(f"every class needs at least one building per rank: totals {self.class_totals}, {world} ranks")
    classes = [(plan, hi - lo_) for (plan, _), (lo_, hi) in zip(classes, self.class_ranges)]
    lo = 0
    for k, (plan, n) in enumerate(classes):
      # per-building generators (convection shuffle, randomized occupancy) draw from streams keyed by the GLOBAL
      # building index: class k's building i is building sum(totals[:k]) + i of the whole mixed batch, on whichever
      # rank it lives -- so the draws do not depend on the sharding, and no two buildings share a stream
      first = sum(self.class_totals[:k]) + self.class_ranges[k][0]
      kw = dict(env_kwargs)
      for key in ("convection_simulator", "occupancy"):
        gen = kw.get(key)
        if gen is not None and hasattr(gen, "first_building"):
          gen = copy.copy(gen)
          gen.first_building = int(gen.first_building) + first
          kw[key] = gen
      stream = torch.cuda.Stream(device=self.tdev)
      with torch.cuda.stream(stream):
        env = BatchedEnvironment(plan, int(n), device=self.device, **kw)
      self.envs.append(env)
      self.streams.append(stream)
      self.slices.append((lo, lo + env.batch_size))
      lo += env.batch_size
    torch.cuda.synchronize(self.tdev)
    self.batch_size = lo
    specs = {tuple(e.action_spec().shape) for e in self.envs}
    if len(specs) != 1 or len({e.steps_per_episode for e in self.envs}) != 1:
      raise ValueError("the classes of a MixedBatchedEnvironment share the action set and the episode length")
    self.observation_widths = [e.sim.O for e in self.envs]
    O = max(self.observation_widths)
    self._action_spec = self.envs[0].action_spec()
    self._observation_spec = ArraySpec((O,), np.dtype(np.float32), "observation")
    self._obs = torch.zeros((self.batch_size, O), dtype=torch.float32, device=self.tdev)
    self._reward = torch.zeros((self.batch_size,), dtype=torch.float32, device=self.tdev)
    self._discount = torch.zeros((self.batch_size,), dtype=torch.float32, device=self.tdev)
    self._step_type = torch.zeros((self.batch_size,), dtype=torch.int32, device=self.tdev)
    cls = torch.zeros((self.batch_size,), dtype=torch.int32, device=self.tdev)
    for k, (a, b) in enumerate(self.slices):
      cls[a:b] = k
    self.class_of_building = cls
    self.profile = False          # True: every step records (start, end) HIP events per class on its stream
    self.class_events: List[List[Tuple[torch.cuda.Event, torch.cuda.Event]]] = [[] for _ in self.envs]

  def action_spec(self) -> ArraySpec:
    return self._action_spec

  def observation_spec(self) -> ArraySpec:
    return self._observation_spec

  @property
  def batched(self) -> bool:
    return True

  @property
  def steps_per_episode(self) -> int:
    return self.envs[0].steps_per_episode

  @property
  def current_simulation_timestamp(self) -> dt.datetime:
    return self.envs[0].current_simulation_timestamp

  def _each(self, fn) -> TimeStep:
    """Runs fn(class index, env) -> TimeStep on every class's stream between two joins with the caller's stream."""
    cur = torch.cuda.current_stream(self.tdev)
    ready = torch.cuda.Event()
    ready.record(cur)
    for k, (env, stream, (a, b)) in enumerate(zip(self.envs, self.streams, self.slices)):
      stream.wait_event(ready)
      with torch.cuda.stream(stream):
        if self.profile:
          e0 = torch.cuda.Event(enable_timing=True)
          e0.record(stream)
        ts = fn(k, env)
        if self.profile:
          e1 = torch.cuda.Event(enable_timing=True)
          e1.record(stream)
          self.class_events[k].append((e0, e1))
        self._obs[a:b, : self.observation_widths[k]].copy_(ts.observation)
        self._reward[a:b].copy_(ts.reward)
        self._discount[a:b].copy_(ts.discount)
        self._step_type[a:b].copy_(ts.step_type)
        done = torch.cuda.Event()
        done.record(stream)
      cur.wait_event(done)
    return TimeStep(self._step_type, self._reward, self._discount, self._obs)

  def reset(self) -> TimeStep:
    return self._each(lambda k, env: env.reset())

  def step(self, action: torch.Tensor, rejected: Optional[torch.Tensor] = None) -> TimeStep:
    """``BatchedEnvironment.step`` for every class (actions / rejected flags of its slice of the batch).  The
    TimeStep's tensors are this environment's own buffers: the next ``step`` / ``reset`` overwrites them."""
    if tuple(action.shape) != (self.batch_size,) + tuple(self._action_spec.shape):
      raise ValueError(f"action must be [{self.batch_size}, {self._action_spec.shape[0]}]")
    action = action.contiguous()
    return self._each(lambda k, env: env.step(action[self.slices[k][0]:self.slices[k][1]],
                                              None if rejected is None else rejected[self.slices[k][0]:self.slices[k][1]]))

  def close(self) -> None:
    for env in self.envs:
      env.close()


class GymVectorEnv:
  """gymnasium.vector-style view of a ``BatchedEnvironment`` (no gymnasium import needed):

      obs, info = venv.reset()
      obs, reward, terminated, truncated, info = venv.step(actions)      # all [B, ...] tensors in HBM

  Every sub-environment shares the simulator clock, so they end together: the last step of an
  episode (``environment.py:1366-1368``: its time is up) returns ``truncated = True`` for all of
  them (``terminated`` with ``time_limit_is_truncation=False``), and -- gymnasium's "next-step" autoreset mode -- the following ``step`` ignores its
  actions and returns the first observation of the next episode with zero reward."""

  def __init__(self, env: BatchedEnvironment, time_limit_is_truncation: bool = True):
    self.env = env
    self.time_limit_is_truncation = bool(time_limit_is_truncation)
    self.num_envs = env.batch_size
    self.single_action_shape = tuple(env.action_spec().shape)
    self.single_observation_shape = tuple(env.observation_spec().shape)
    self.action_low, self.action_high = -1.0, 1.0

  def reset(self, seed=None, options=None):
    del seed, options   # the simulator is deterministic; inputs come from the host controllers
    ts = self.env.reset()
    return ts.observation, {"step_type": ts.step_type}

  def step(self, actions: torch.Tensor):
    ts = self.env.step(actions)
    # an episode ends only because its time is up: gymnasium's TRUNCATION (a learner keeps bootstrapping);
    # time_limit_is_truncation=False reports it as a termination, like the reference's discount 0
    last = ts.step_type == STEP_LAST
    terminated, truncated = (torch.zeros_like(last), last) if self.time_limit_is_truncation else (last, torch.zeros_like(last))
    return ts.observation, ts.reward, terminated, truncated, {"step_type": ts.step_type, "discount": ts.discount}

  def close(self) -> None:
    self.env.close()
