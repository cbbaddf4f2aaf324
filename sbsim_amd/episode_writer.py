"""Episode shards in the reference's on-disk format.

``EpisodeWriter`` mirrors ``ProtoWriter`` (utils/controller_writer.py:35-131): the same four
``write_*`` methods and the same files -- ``<prefix>_YYYY.MM.DD.HH`` holding
``<4-byte little-endian size><serialized proto>`` records -- so the reference's ``ProtoReader``
and plotting tools can read runs of this library.  The messages are not protobuf objects but
plain values (this package has no protobuf dependency); ``libsbsim_amd.so`` encodes them
(``sb_pb_*``, csrc/episode.cpp).  ``log_building_step`` pulls one building's step out of a
``BatchedEnvironment`` the way ``Environment._step`` logs it (environment.py:1268-1330)."""
from __future__ import annotations

import ctypes as C
import datetime as dt
from typing import Mapping, Optional, Sequence, Tuple

import numpy as np

from sbsim_amd import _ffi, host_inputs

OBSERVATION_RESPONSE_FILE_PREFIX = "observation_response"   # utils/constants.py:51-56
ACTION_RESPONSE_FILE_PREFIX = "action_response"
REWARD_INFO_PREFIX = "reward_info"
REWARD_RESPONSE_PREFIX = "reward_response"
ACCEPTED = 1   # SingleActionResponse.ActionResponseType


def _pb_time(ts) -> Tuple[_ffi.PbTime, int]:
  """conversion_utils.pandas_to_proto_timestamp: seconds + nanos of the instant; naive = UTC."""
  ts = host_inputs.as_datetime(ts)
  if ts.tzinfo is None:
    ts = ts.replace(tzinfo=dt.timezone.utc)
  delta = ts - dt.datetime(1970, 1, 1, tzinfo=dt.timezone.utc)
  seconds = delta.days * 86400 + delta.seconds
  return _ffi.PbTime(seconds, delta.microseconds * 1000), seconds


def _strs(items: Sequence[str]):
  arr = (C.c_char_p * max(len(items), 1))()
  for i, s in enumerate(items):
    arr[i] = s.encode()
  return arr


def _floats(a) -> np.ndarray:
  return np.ascontiguousarray(a, dtype=np.float32)


def _fptr(a: np.ndarray):
  return a.ctypes.data_as(C.POINTER(C.c_float))


class EpisodeWriter:

  def __init__(self, output_dir: str):
    import os
    self._dir = str(output_dir)
    os.makedirs(self._dir, exist_ok=True)   # controller_writer.py:50
    self._lib = _ffi.load()

  # ---- encoders (serialized bytes) ----
  def _run(self, fn, *args) -> bytes:
    need = fn(*args, None, 0)
    if need < 0:
      raise _ffi.SbsimError(f"{fn.__name__} failed ({need})")
    buf = (C.c_uint8 * max(need, 1))()
    fn(*args, C.cast(buf, C.c_void_p), need)
    return bytes(buf[:need])

  def encode_reward_info(self, start, end, zone_ids: Sequence[str], zone_vals, ahu_ids: Sequence[str], ahu_vals,
                         boiler_ids: Sequence[str], boiler_vals, agent_id: str = "", scenario_id: str = "") -> bytes:
    zv = _floats(zone_vals).reshape(len(zone_ids), 6)
    av = _floats(ahu_vals).reshape(len(ahu_ids), 2)
    bv = _floats(boiler_vals).reshape(len(boiler_ids), 2)
    return self._run(self._lib.sb_pb_reward_info, _pb_time(start)[0], _pb_time(end)[0], agent_id.encode(),
                     scenario_id.encode(), len(zone_ids), _strs(zone_ids), _fptr(zv), len(ahu_ids), _strs(ahu_ids),
                     _fptr(av), len(boiler_ids), _strs(boiler_ids), _fptr(bv))

  def encode_reward_response(self, values17, start, end) -> bytes:
    v = _floats(values17).reshape(17)
    return self._run(self._lib.sb_pb_reward_response, _fptr(v), _pb_time(start)[0], _pb_time(end)[0])

  def encode_observation_response(self, timestamp, device_ids: Sequence[str], measurement_names: Sequence[str],
                                  values, valid=None) -> bytes:
    v = _floats(values).reshape(len(device_ids))
    ok = np.ascontiguousarray(np.ones(len(device_ids)) if valid is None else valid, dtype=np.uint8)
    return self._run(self._lib.sb_pb_observation_response, _pb_time(timestamp)[0], len(device_ids), _strs(device_ids),
                     _strs(measurement_names), _fptr(v), C.c_void_p(ok.ctypes.data))

  def encode_action_response(self, timestamp, device_ids: Sequence[str], setpoint_names: Sequence[str], values,
                             response_types=None, request_timestamp=None) -> bytes:
    v = _floats(values).reshape(len(device_ids))
    rt = np.ascontiguousarray(np.full(len(device_ids), ACCEPTED) if response_types is None else response_types,
                              dtype=np.int32)
    req_ts = timestamp if request_timestamp is None else request_timestamp
    return self._run(self._lib.sb_pb_action_response, _pb_time(timestamp)[0], _pb_time(req_ts)[0], len(device_ids),
                     _strs(device_ids), _strs(setpoint_names), _fptr(v), C.c_void_p(rt.ctypes.data))

  # ---- ProtoWriter's interface ----
  def _append(self, prefix: str, timestamp, msg: bytes) -> None:
    ts = host_inputs.as_datetime(timestamp)
    # _get_serial: timestamp.strftime('%Y.%m.%d.%H') -- the timestamp's own wall clock
    wall = ts.replace(tzinfo=dt.timezone.utc)
    seconds = int((wall - dt.datetime(1970, 1, 1, tzinfo=dt.timezone.utc)).total_seconds())
    _ffi.check(self._lib.sb_shard_append(self._dir.encode(), prefix.encode(), seconds, msg, len(msg)), "sb_shard_append")

  def write_observation_response(self, observation_response: bytes, timestamp) -> None:
    self._append(OBSERVATION_RESPONSE_FILE_PREFIX, timestamp, observation_response)

  def write_action_response(self, action_response: bytes, timestamp) -> None:
    self._append(ACTION_RESPONSE_FILE_PREFIX, timestamp, action_response)

  def write_reward_info(self, reward_info: bytes, timestamp) -> None:
    self._append(REWARD_INFO_PREFIX, timestamp, reward_info)

  def write_reward_response(self, reward_response: bytes, timestamp) -> None:
    self._append(REWARD_RESPONSE_PREFIX, timestamp, reward_response)

  def write_building_image(self, base64_img: bytes, timestamp) -> None:
    """utils/controller_writer.py:65-71: one CSV row [timestamp.timestamp(), the image] appended to
    building_images.csv (utils/constants.py:54) with the csv module's default dialect, as the reference does."""
    import csv
    import os
    ts = host_inputs.as_datetime(timestamp)
    seconds = ts.timestamp() if ts.tzinfo else ts.replace(tzinfo=dt.timezone.utc).timestamp()  # pd.Timestamp.timestamp(): naive = UTC
    with open(os.path.join(self._dir, BUILDING_IMAGE_CSV_FILE), "a") as fh:
      csv.writer(fh).writerow([seconds, base64_img])


# DeviceInfo.DeviceType / ValueType, ZoneInfo.ZoneType (proto/smart_control_building.proto)
DEVICE_TYPES = dict(UNDEFINED=0, FAN=1, PMP=2, FCU=3, VAV=4, DH=5, AHU=6, BLR=7, OTHER=23)
VALUE_CONTINUOUS, ZONE_TYPE_ROOM = 1, 1
DEVICE_INFO_PREFIX, ZONE_INFO_PREFIX = "device_info", "zone_info"   # utils/constants.py:57-58
BUILDING_IMAGE_CSV_FILE = "building_images.csv"                       # utils/constants.py:54


def _ints(a):
  arr = np.ascontiguousarray(a, dtype=np.int32)
  return arr, arr.ctypes.data_as(C.POINTER(C.c_int32))


def encode_device_info(device_id: str, namespace: str, code: str, zone_id: str, device_type: int,
                       observable_fields: Mapping[str, int], action_fields: Mapping[str, int]) -> bytes:
  lib = _ffi.load()
  on, an = list(observable_fields), list(action_fields)
  ot, otp = _ints([observable_fields[k] for k in on] or [0])
  at, atp = _ints([action_fields[k] for k in an] or [0])
  args = (device_id.encode(), namespace.encode(), code.encode(), zone_id.encode(), int(device_type), len(on),
          _strs(on), otp, len(an), _strs(an), atp)
  need = lib.sb_pb_device_info(*args, None, 0)
  if need < 0:
    raise _ffi.SbsimError(f"sb_pb_device_info failed ({need})")
  buf = (C.c_uint8 * max(need, 1))()
  lib.sb_pb_device_info(*args, C.cast(buf, C.c_void_p), need)
  return bytes(buf[:need])


def encode_zone_info(zone_id: str, building_id: str, zone_description: str, area: float, devices: Sequence[str],
                     zone_type: int = ZONE_TYPE_ROOM, floor: int = 0) -> bytes:
  lib = _ffi.load()
  args = (zone_id.encode(), building_id.encode(), zone_description.encode(), float(area), len(devices),
          _strs(list(devices)), int(zone_type), int(floor))
  need = lib.sb_pb_zone_info(*args, None, 0)
  if need < 0:
    raise _ffi.SbsimError(f"sb_pb_zone_info failed ({need})")
  buf = (C.c_uint8 * max(need, 1))()
  lib.sb_pb_zone_info(*args, C.cast(buf, C.c_void_p), need)
  return bytes(buf[:need])


NORMALIZATION_FILENAME = "normalization_info"   # utils/constants.py:50


def encode_variable_info(variable_id: str, sample_size: int, variance: float, mean: float, median: float = 0.0,
                         maximum: float = 0.0, minimum: float = 0.0, sample_start=None, sample_end=None) -> bytes:
  """ContinuousVariableInfo (write_normalization_info, controller_writer.py:134-147)."""
  lib = _ffi.load()
  stats = _floats([variance, mean, median, maximum, minimum])
  t0 = _pb_time(sample_start)[0] if sample_start is not None else _ffi.PbTime(0, 0)
  t1 = _pb_time(sample_end)[0] if sample_end is not None else _ffi.PbTime(0, 0)
  args = (variable_id.encode(), int(sample_start is not None), t0, int(sample_end is not None), t1, int(sample_size),
          _fptr(stats))
  need = lib.sb_pb_variable_info(*args, None, 0)
  if need < 0:
    raise _ffi.SbsimError(f"sb_pb_variable_info failed ({need})")
  buf = (C.c_uint8 * max(need, 1))()
  lib.sb_pb_variable_info(*args, C.cast(buf, C.c_void_p), need)
  return bytes(buf[:need])


def write_records(path: str, records: Sequence[bytes]) -> None:
  """write_device_infos / write_zone_infos (controller_writer.py:149-171): the file starts over."""
  lib = _ffi.load()
  for i, r in enumerate(records):
    _ffi.check(lib.sb_record_append(path.encode(), r, len(r), int(i == 0)), "sb_record_append")


def read_shard(path: str):
  """The records of one shard file (utils/controller_reader.py: 4-byte size, message)."""
  out = []
  with open(path, "rb") as fh:
    data = fh.read()
  i = 0
  while i < len(data):
    n = int.from_bytes(data[i:i + 4], "little")
    out.append(data[i + 4:i + 4 + n])
    i += 4 + n
  return out


class BuildingLogger:
  """Logs chosen buildings of a ``BatchedEnvironment`` the way ``Environment`` logs its single
  building (environment.py:891-894 observation, :1083-1090 reward info + response, :1290-1293
  action response), one ``EpisodeWriter`` directory per building:

      logger = BuildingLogger(env, "/tmp/run", buildings=[0, 17])
      ts = env.reset(); logger.log_reset()
      t0 = env.current_simulation_timestamp
      ts = env.step(actions); logger.log_step(actions, t0)

  The environment must have been built with ``collect_info=True`` and without a histogram reducer
  (observations are logged per device field; normalised fields are mapped back through the
  normaliser's mean and sigma)."""

  def __init__(self, env, output_dir: str, buildings: Sequence[int], zone_ids: Optional[Sequence[str]] = None,
               agent_id: str = "", scenario_id: str = ""):
    import os
    if env.info is None:
      raise ValueError("BuildingLogger needs BatchedEnvironment(collect_info=True)")
    sim = env.sim
    if len(sim.source_names) + _ffi.SB_NUM_AUX != sim.O:
      raise ValueError("BuildingLogger needs per-field observations (no histogram reducer)")
    self.env, self.buildings = env, [int(b) for b in buildings]
    self.writers = {b: EpisodeWriter(os.path.join(output_dir, "building_%06d" % b)) for b in self.buildings}
    self.devices = [n.split("/", 1)[0] for n in sim.source_names]
    self.measurements = [n.split("/", 1)[1] for n in sim.source_names]
    self.zone_ids = list(zone_ids or sim.zone_names)
    self.agent_id, self.scenario_id = agent_id, scenario_id
    self.ahu_id = next(d for d in self.devices if "air_handler" in d)
    self.boiler_id = next(d for d in self.devices if "boiler" in d)
    n = len(sim.source_names)
    self._mean, self._sigma = sim._keep["mean"][:n], sim._keep["sigma"][:n]

  def _raw_observations(self, obs_rows: np.ndarray) -> np.ndarray:
    return obs_rows[:, :len(self.devices)].astype(np.float64) * self._sigma + self._mean

  def _log_observation(self, ts) -> None:
    rows = self.env._obs[self.buildings].cpu().numpy()
    raw = self._raw_observations(rows)
    for k, b in enumerate(self.buildings):
      w = self.writers[b]
      w.write_observation_response(w.encode_observation_response(ts, self.devices, self.measurements, raw[k]), ts)

  def log_reset(self) -> None:
    self._log_observation(self.env.current_simulation_timestamp)

  def log_step(self, actions, timestamp_before) -> None:
    """After ``env.step(actions)``; ``timestamp_before``: the simulator time the step started at."""
    env, cfg = self.env, self.env.config
    now = env.current_simulation_timestamp
    end = now + dt.timedelta(seconds=cfg.time_step_sec)
    act = actions[self.buildings].cpu().numpy().astype(np.float64)
    (w_lo, w_hi), (a_lo, a_hi) = cfg.action_ranges
    native = np.stack([np.float32((act[:, 0] + 1.0) / 2.0 * (w_hi - w_lo) + w_lo),      # bounded_action_normalizer.py:93-98
                       np.float32((act[:, 1] + 1.0) / 2.0 * (a_hi - a_lo) + a_lo)], axis=1)
    info = env.info[self.buildings].cpu().numpy()
    zt = env.sim.zone_temps()[self.buildings].cpu().numpy()
    flow = env.sim.scalars()[self.buildings, 2].cpu().numpy()
    hsp, csp = env.schedule.get_temperature_window(now)
    occ = (env._occ_count[self.buildings].cpu().numpy() if env._occ_count is not None
           else np.full((len(self.buildings), env.sim.Z), info[:, 17:18] / env.sim.Z))
    self._log_observation(now)
    for k, b in enumerate(self.buildings):
      w = self.writers[b]
      w.write_action_response(w.encode_action_response(
          timestamp_before, [self.boiler_id, self.ahu_id],
          ["supply_water_setpoint", "supply_air_heating_temperature_setpoint"], native[k]), timestamp_before)
      zone_vals = np.stack([np.full(env.sim.Z, hsp), np.full(env.sim.Z, csp), zt[k],
                            np.full(env.sim.Z, cfg.vav_max_air_flow_rate), np.full(env.sim.Z, flow[k]), occ[k]], axis=1)
      w.write_reward_info(w.encode_reward_info(now, end, self.zone_ids, zone_vals, [self.ahu_id], info[k, 0:2],
                                               [self.boiler_id], info[k, 2:4], self.agent_id, self.scenario_id), now)
      resp = np.concatenate([info[k, 7:8], info[k, 8:24]])
      w.write_reward_response(w.encode_reward_response(resp, now, end), now)
