"""``HipSimulatorBuilding``: the HIP step behind the reference's building interface.

The reference's ``Environment`` talks to a ``BaseBuilding`` (models/base_building.py:27-95); its
simulated one is ``SimulatorBuilding`` (simulator/simulator_building.py:151-315):

    request_action(ActionRequest) -> ActionResponse      setup_step_sim + set_action per setpoint
    wait_time()                                           execute_step_sim: one FD time step
    request_observations(ObservationRequest) -> ObservationResponse
    reward_info -> RewardInfo                             what the reward function consumes
    reset(), devices, zones, current_timestamp, time_step_sec, is_comfort_mode(ts), num_occupants

This class offers exactly that on a handle of the C ABI (include/sbsim_amd.h), one building (or
``n_replicas`` identical ones, building ``building`` being the one that answers).  The messages
are plain dataclasses with the protos' field names (proto/smart_control_building.proto:107-190,
proto/smart_control_reward.proto:49-117); ``float`` proto fields hold float32-rounded values, as
the reference's protos do.  A real ``smart_control_building_pb2`` request can be passed as well:
only attributes are read.

Semantics carried over:
  * ``request_action`` answers per setpoint: ACCEPTED, REJECTED_INVALID_DEVICE (unknown device),
    REJECTED_NOT_ENABLED_OR_AVAILABLE (not a settable field / outside the VAV damper's [0, 1],
    simulator_building.py:224-260).  Fields the handle was not configured to drive
    (``SimConfig.action_names``) are REJECTED_NOT_ENABLED_OR_AVAILABLE too.
  * ``wait_time`` without a preceding ``request_action`` is what ``Environment`` does when the
    building rejected the request (RuntimeError, environment.py:1266-1309): no thermostat update,
    no set_action, the step goes on.
  * one observation of the boiler per step advances its tank lag (boiler.py:146-217), as in
    ``Environment._step``.
"""
from __future__ import annotations

import dataclasses
import datetime as dt
import enum
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from sbsim_amd import _ffi, host_inputs
from sbsim_amd.environment import ACTION_KINDS, BatchedEnvironment, SimConfig
from sbsim_amd.floorplan import FloorPlan


class ActionResponseType(enum.IntEnum):   # smart_control_building.proto:164-184
  UNDEFINED = 0
  ACCEPTED = 1
  PENDING = 2
  TIMED_OUT = 3
  REJECTED_INVALID_SETTING = 4
  REJECTED_NOT_ENABLED_OR_AVAILABLE = 5
  REJECTED_OVERRIDE = 6
  REJECTED_INVALID_DEVICE = 7
  REJECTED_DEVICE_OFFLINE = 8
  UNKNOWN = 9
  OTHER = 10


class DeviceType(enum.IntEnum):           # smart_control_building.proto:57-82 (the simulated ones)
  UNDEFINED = 0
  VAV = 4
  AHU = 6
  BLR = 7


def _f32(x) -> float:
  return float(np.float32(x))


@dataclasses.dataclass
class SingleActionRequest:
  device_id: str = ""
  setpoint_name: str = ""
  continuous_value: float = 0.0


@dataclasses.dataclass
class ActionRequest:
  timestamp: Optional[dt.datetime] = None
  single_action_requests: List[SingleActionRequest] = dataclasses.field(default_factory=list)


@dataclasses.dataclass
class SingleActionResponse:
  request: SingleActionRequest
  response_type: ActionResponseType = ActionResponseType.UNDEFINED
  additional_info: str = ""


@dataclasses.dataclass
class ActionResponse:
  timestamp: Optional[dt.datetime] = None
  request: Optional[ActionRequest] = None
  single_action_responses: List[SingleActionResponse] = dataclasses.field(default_factory=list)


@dataclasses.dataclass
class SingleObservationRequest:
  device_id: str = ""
  measurement_name: str = ""


@dataclasses.dataclass
class ObservationRequest:
  timestamp: Optional[dt.datetime] = None
  single_observation_requests: List[SingleObservationRequest] = dataclasses.field(default_factory=list)


@dataclasses.dataclass
class SingleObservationResponse:
  timestamp: Optional[dt.datetime] = None
  single_observation_request: Optional[SingleObservationRequest] = None
  observation_valid: bool = False
  continuous_value: float = 0.0


@dataclasses.dataclass
class ObservationResponse:
  timestamp: Optional[dt.datetime] = None
  request: Optional[ObservationRequest] = None
  single_observation_responses: List[SingleObservationResponse] = dataclasses.field(default_factory=list)


@dataclasses.dataclass
class ZoneRewardInfo:
  heating_setpoint_temperature: float = 0.0
  cooling_setpoint_temperature: float = 0.0
  zone_air_temperature: float = 0.0
  air_flow_rate_setpoint: float = 0.0
  air_flow_rate: float = 0.0
  average_occupancy: float = 0.0


@dataclasses.dataclass
class AirHandlerRewardInfo:
  blower_electrical_energy_rate: float = 0.0
  air_conditioning_electrical_energy_rate: float = 0.0


@dataclasses.dataclass
class BoilerRewardInfo:
  natural_gas_heating_energy_rate: float = 0.0
  pump_electrical_energy_rate: float = 0.0


@dataclasses.dataclass
class RewardInfo:
  start_timestamp: Optional[dt.datetime] = None
  end_timestamp: Optional[dt.datetime] = None
  agent_id: str = ""
  scenario_id: str = ""
  zone_reward_infos: Dict[str, ZoneRewardInfo] = dataclasses.field(default_factory=dict)
  air_handler_reward_infos: Dict[str, AirHandlerRewardInfo] = dataclasses.field(default_factory=dict)
  boiler_reward_infos: Dict[str, BoilerRewardInfo] = dataclasses.field(default_factory=dict)


@dataclasses.dataclass
class DeviceInfo:
  device_id: str
  device_type: DeviceType
  zone_id: str = ""
  namespace: str = ""
  code: str = ""
  observable_fields: Dict[str, str] = dataclasses.field(default_factory=dict)   # field -> "VALUE_CONTINUOUS"
  action_fields: Dict[str, str] = dataclasses.field(default_factory=dict)


@dataclasses.dataclass
class ZoneInfo:
  zone_id: str
  building_id: str = ""
  zone_description: str = ""
  area: float = 0.0
  devices: List[str] = dataclasses.field(default_factory=list)
  zone_type: int = 1   # ROOM


_SETTABLE = {   # boiler.py:81-85, air_handler.py:97-104, vav.py:65-69
    DeviceType.BLR: ("supply_water_setpoint",),
    DeviceType.AHU: ("supply_air_heating_temperature_setpoint", "supply_air_cooling_temperature_setpoint"),
    DeviceType.VAV: ("supply_air_damper_percentage_command",),
}


class HipSimulatorBuilding:
  """simulator_building.py:39-315 on the HIP library (see the module docstring)."""

  def __init__(self, plan: FloorPlan, config: Optional[SimConfig] = None, n_replicas: int = 1, building: int = 0,
               device: int = 0, weather=None, occupancy=None, start_timestamp=dt.datetime(2023, 7, 6, 7, 0, 0),
               holiday_calendar="us", agent_id: str = "", scenario_id: str = "",
               air_handler_id: str = "air_handler_id", boiler_id: str = "boiler_id",
               convection_simulator=None, reproducible_convection: bool = False):
    """convection_simulator: host_inputs.StochasticConvectionSimulator(p, distance, seed) -- the shuffle after every
    finite-difference update (simulator_flexible_floor_plan.py:71, 156).  By default it runs on the device
    (statistically the reference's process); reproducible_convection=True (one building only) runs the
    reference's seeded shuffle on the host instead, draw for draw (host_convection.py): the temperature array
    is then bit-identical to the reference's for the same seed."""
    self.config = config or SimConfig.sb1()
    self._host_convection = None
    if convection_simulator is not None and reproducible_convection:
      if n_replicas != 1:
        raise ValueError("reproducible_convection replays ONE random stream: n_replicas must be 1")
      from .host_convection import SeededHostConvection
      c = convection_simulator
      self._host_convection = SeededHostConvection(c.p, c.distance, c.seed)
      self._rooms = [[(int(x), int(y)) for x, y in zip(*np.unravel_index(cells, plan.shape))] for cells in plan.zone_cell_lists()]
      convection_simulator = None
    # identity normalisation: request_observations returns native values (Environment normalises them itself)
    self.env = BatchedEnvironment(plan, n_replicas, config=self.config, weather=weather, occupancy=occupancy,
                                  start_timestamp=start_timestamp, device=device, observation_normalization=None,
                                  holiday_calendar=holiday_calendar, collect_info=True,
                                  convection_simulator=convection_simulator)
    if self.env._occ_count is not None:
      raise ValueError("HipSimulatorBuilding takes a host-side occupancy model (one building)")
    sim = self.env.sim
    self._b = int(building)
    self.agent_id, self.scenario_id = agent_id, scenario_id
    self._ahu_id, self._boiler_id = air_handler_id, boiler_id
    self._columns = {n: i for i, n in enumerate(sim.source_names)}       # "device/field" -> observation column
    zone_names = list(sim.zone_names)
    self._vav_ids = [f"vav_{z}" for z in zone_names]
    self._zone_ids = [f"zone_id_{z}" for z in zone_names]
    fields = lambda dev: {n.split("/", 1)[1]: "VALUE_CONTINUOUS" for n in sim.source_names if n.split("/", 1)[0] == dev}
    act = lambda kind: {f: "VALUE_CONTINUOUS" for f in _SETTABLE[kind]}
    self._devices = [DeviceInfo(self._ahu_id, DeviceType.AHU, observable_fields=fields(self._ahu_id), action_fields=act(DeviceType.AHU)),
                     DeviceInfo(self._boiler_id, DeviceType.BLR, observable_fields=fields(self._boiler_id), action_fields=act(DeviceType.BLR))]
    self._devices += [DeviceInfo(v, DeviceType.VAV, zone_id=z, observable_fields=fields(v), action_fields=act(DeviceType.VAV))
                      for v, z in zip(self._vav_ids, self._zone_ids)]
    self._device_map = {d.device_id: d for d in self._devices}
    self._zones = [ZoneInfo(z, devices=[v], zone_description=n) for z, v, n in zip(self._zone_ids, self._vav_ids, zone_names)]
    # action columns of the handle: (device_id, setpoint) -> column
    self._action_col = {}
    for i, name in enumerate(self.config.action_names):
      kind = ACTION_KINDS[name]
      if kind == _ffi.SB_ACT_BOILER_SUPPLY_WATER_SETPOINT:
        dev = self._boiler_id
      elif kind == _ffi.SB_ACT_VAV_SUPPLY_AIR_DAMPER_COMMAND:
        dev = self._vav_ids[self.config.action_zones[i] if i < len(self.config.action_zones) else 0]
      else:
        dev = self._ahu_id
      self._action_col[(dev, name)] = i
    dv = sim.tdev
    self._actions = torch.full((sim.B, sim.n_actions), _ffi.SB_ACTION_KEEP, dtype=torch.float32, device=dv)
    self._reject = torch.zeros((sim.B,), dtype=torch.uint8, device=dv)
    self._requested = False
    self._all_accepted = True
    self._info_row = None
    self._occupancy_at_reward = 0.0
    self.last_reward = 0.0
    self.reset()

  # ---- BaseBuilding ----
  def reset(self) -> None:
    env = self.env
    env.reset()                              # Simulator.reset + the first observation (environment.py:1165-1176)
    self._requested = False
    self._actions.fill_(_ffi.SB_ACTION_KEEP)
    self._info_row = None

  @property
  def devices(self) -> Sequence[DeviceInfo]:
    return self._devices

  @property
  def zones(self) -> Sequence[ZoneInfo]:
    return self._zones

  @property
  def time_step_sec(self) -> float:
    return self.config.time_step_sec

  @property
  def current_timestamp(self) -> dt.datetime:
    return self.env.current_simulation_timestamp

  def is_comfort_mode(self, current_time) -> bool:
    return bool(self.env.schedule.is_comfort_mode(host_inputs.as_datetime(current_time)))

  @property
  def num_occupants(self) -> int:   # simulator_building.py:303-315
    ts = self.current_timestamp
    n = 0.0
    for z in self._zone_ids:
      n += self.env.occupancy.average_zone_occupancy(z, ts - dt.timedelta(minutes=5), ts)
    return int(n)

  def request_action(self, action_request) -> ActionResponse:
    """simulator_building.py:204-263: the thermostats update now, then every setpoint is applied."""
    self._requested = True
    self._actions.fill_(_ffi.SB_ACTION_KEEP)
    resp = ActionResponse(timestamp=self.current_timestamp, request=action_request)
    row = [_ffi.SB_ACTION_KEEP] * self.env.sim.n_actions
    for single in action_request.single_action_requests:
      r = SingleActionResponse(request=single, response_type=ActionResponseType.ACCEPTED)
      dev = self._device_map.get(single.device_id)
      if dev is None:
        r.response_type = ActionResponseType.REJECTED_INVALID_DEVICE
      elif single.setpoint_name not in dev.action_fields:
        r.response_type = ActionResponseType.REJECTED_NOT_ENABLED_OR_AVAILABLE
        r.additional_info = f"{single.setpoint_name} is not an action field of {single.device_id}"
      elif (single.device_id, single.setpoint_name) not in self._action_col:
        r.response_type = ActionResponseType.REJECTED_NOT_ENABLED_OR_AVAILABLE
        r.additional_info = "the handle was not configured to drive this field (SimConfig.action_names)"
      else:
        v = _f32(single.continuous_value)
        # vav.py:125-129: the setter raises for `value < 0 or value > 1` (a NaN passes, as in the
        # reference).  The value goes to the device either way: k_pre makes the same test, leaves the
        # damper alone and returns the reward -inf (environment.py:1301-1302) for this building.
        if dev.device_type == DeviceType.VAV and (v < 0.0 or v > 1.0):
          r.response_type = ActionResponseType.REJECTED_NOT_ENABLED_OR_AVAILABLE
          r.additional_info = "damper_setting must be in [0 ,1]"
          # a FINITE out-of-range value: anything <= SB_ACTION_KEEP (-inf, -3.4e38) would read as "field not
          # mentioned" on the device, which would then not reject the step
          v = -1.0 if v < 0.0 else 2.0
        row[self._action_col[(single.device_id, single.setpoint_name)]] = v
      resp.single_action_responses.append(r)
    self._actions[:] = torch.tensor(row, dtype=torch.float32, device=self._actions.device)
    self._all_accepted = all(r.response_type == ActionResponseType.ACCEPTED for r in resp.single_action_responses)
    return resp

  def wait_time(self) -> None:
    """simulator_building.py:265-268 execute_step_sim (+ the observation / reward_info state of the new time)."""
    env = self.env
    si = env.make_step_in(env._now, has_action=True)
    si.actions_native = 1
    self._reject.fill_(0 if self._requested else 1)      # no request_action since the last step: the request was rejected
    si.reject_dev = self._reject.data_ptr()
    self._occupancy_at_reward = float(si.occupancy)
    env.sim.step(self._actions, si, env._obs, env._reward, env._info)
    if self._host_convection is not None:     # Building.apply_convection (building.py:891-893), the reference's own draws
      grid = env.sim.temps()
      host = grid[0].cpu().numpy()
      self._host_convection.apply(self._rooms, host)
      grid[0] = torch.from_numpy(host).to(grid.device)
      env.sim.set_temps(grid)
    env._prev_thermostat_ts = env._now
    env._now = env._now + env._step_interval
    self._requested = False
    self._info_row = env._info[self._b].cpu().numpy().astype(np.float64)
    self.last_reward = float(self._info_row[7])          # what compute_reward returns (before Environment's -inf)

  def request_observations(self, observation_request) -> ObservationResponse:
    """simulator_building.py:151-202: every (device, measurement) of the request."""
    ts = self.current_timestamp
    row = self.env._obs[self._b].cpu().numpy()
    resp = ObservationResponse(timestamp=ts, request=observation_request)
    for single in observation_request.single_observation_requests:
      col = self._columns.get(f"{single.device_id}/{single.measurement_name}")
      resp.single_observation_responses.append(SingleObservationResponse(
          timestamp=ts, single_observation_request=single, observation_valid=col is not None,
          continuous_value=float(row[col]) if col is not None else 0.0))
    return resp

  def observation_request_for_all_fields(self) -> ObservationRequest:
    return ObservationRequest(timestamp=self.current_timestamp, single_observation_requests=[
        SingleObservationRequest(*n.split("/", 1)) for n in self.env.sim.source_names])

  @property
  def reward_info(self) -> RewardInfo:
    """simulator.py:548-576 at the current time (after ``wait_time``)."""
    if self._info_row is None:
      raise RuntimeError("reward_info needs a completed step (wait_time)")
    env, cfg, i = self.env, self.config, self._info_row
    now = self.current_timestamp
    hsp, csp = env.schedule.get_temperature_window(now)
    zt = env.sim.zone_temps()[self._b].cpu().numpy()
    flow = float(env.sim.scalars()[self._b, 2])
    ri = RewardInfo(start_timestamp=now, end_timestamp=now + dt.timedelta(seconds=cfg.time_step_sec),
                    agent_id=self.agent_id, scenario_id=self.scenario_id)
    for z, zone_id in enumerate(self._zone_ids):
      occ = env.occupancy.average_zone_occupancy(zone_id, now, now + dt.timedelta(seconds=cfg.time_step_sec))
      ri.zone_reward_infos[zone_id] = ZoneRewardInfo(_f32(hsp), _f32(csp), _f32(zt[z]), _f32(cfg.vav_max_air_flow_rate),
                                                     _f32(flow), _f32(occ))
    ri.air_handler_reward_infos[self._ahu_id] = AirHandlerRewardInfo(float(i[0]), float(i[1]))
    ri.boiler_reward_infos[self._boiler_id] = BoilerRewardInfo(float(i[2]), float(i[3]))
    return ri

  def close(self) -> None:
    self.env.close()
