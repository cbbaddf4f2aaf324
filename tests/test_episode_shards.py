"""Episode shards: the hand-written proto3 encoders and the shard files
against bytes produced by the reference's own message classes and its own ProtoWriter
(tests/golden/episode_shards.npz, generator oracle/gen_golden_shards.py)."""
import datetime as dt
import os

import numpy as np

from sbsim_amd.episode_writer import EpisodeWriter, read_shard
from tests.golden_util import load

UTC = dt.timezone.utc


def _times(g):
  return [dt.datetime.fromtimestamp(int(s), tz=UTC) + dt.timedelta(microseconds=int(ns) // 1000)
          for s, ns in zip(g["times_s"], g["times_ns"])]


def test_encoders_reproduce_the_reference_serialization(tmp_path):
  g = load("episode_shards.npz")
  w = EpisodeWriter(str(tmp_path))
  times = _times(g)
  zone_ids = [str(z) for z in g["zone_ids"]]
  for k, t in enumerate(times):
    end = t + dt.timedelta(seconds=300)
    got = w.encode_reward_info(t, end, zone_ids, g["zone_vals"][k], ["air_handler_id"], g["ahu_vals"][k],
                               ["boiler_id"], g["blr_vals"][k], agent_id="agent" if k else "", scenario_id="s%d" % k)
    assert got == g[f"reward_info_{k}"].tobytes(), ("reward_info", k)
    got = w.encode_reward_response(g["resp_vals"][k], t, end)
    assert got == g[f"reward_response_{k}"].tobytes(), ("reward_response", k)
    w.write_reward_response(got, t)
    got = w.encode_observation_response(t, [str(s) for s in g["obs_dev"]], [str(s) for s in g["obs_meas"]],
                                        g["obs_vals"][k], g["obs_valid"][k])
    assert got == g[f"observation_response_{k}"].tobytes(), ("observation_response", k)
    w.write_observation_response(got, t)
    got = w.encode_action_response(t, [str(s) for s in g["act_dev"]], [str(s) for s in g["act_names"]],
                                   g["act_vals"][k], g["act_types"][k])
    assert got == g[f"action_response_{k}"].tobytes(), ("action_response", k)
    w.write_action_response(got, t)
  # the files: same names (hour of the timestamp), same bytes as the reference's ProtoWriter wrote
  names = sorted(os.listdir(tmp_path))
  assert names == [str(n) for n in g["shard_names"] if str(n) not in ("zone_info", "normalization_info")]   # (their own tests below)
  for n in names:
    assert open(tmp_path / n, "rb").read() == g["shard_" + n].tobytes(), n
  assert len(read_shard(str(tmp_path / "observation_response_2023.07.06.08"))) == 2


def test_wire_format_corner_cases(tmp_path):
  w = EpisodeWriter(str(tmp_path / "made" / "on" / "demand"))     # ProtoWriter.__init__: os.makedirs
  w.write_reward_info(b"\x01\x02", dt.datetime(2024, 2, 29, 23, 59, tzinfo=UTC))
  assert read_shard(str(tmp_path / "made" / "on" / "demand" / "reward_info_2024.02.29.23")) == [b"\x01\x02"]
  t = dt.datetime(1970, 1, 1, tzinfo=UTC)
  # all defaults: only the two (empty) timestamp submessages remain
  assert w.encode_reward_response(np.zeros(17), t, t) == bytes([0x92, 0x01, 0x00, 0x9a, 0x01, 0x00])
  # an invalid observation carries no value; a valid zero is written (oneof)
  a = w.encode_observation_response(t, ["d"], ["m"], [0.0], [0])
  b = w.encode_observation_response(t, ["d"], ["m"], [0.0], [1])
  assert len(b) == len(a) + 2 + 5 and b.endswith(bytes([0x18, 0x01, 0x25, 0, 0, 0, 0]))
  # pre-epoch instants: negative seconds are ten-byte varints
  early = dt.datetime(1969, 12, 31, 23, 59, 59, tzinfo=UTC)
  m = w.encode_reward_response(np.zeros(17), early, t)
  assert m[:3] == bytes([0x92, 0x01, 0x0b]) and m[3] == 0x08 and m[4:14] == bytes([0xff] * 9 + [0x01])


# ------------------------------------------------------------------------------------- GPU
def _decode(msg: bytes):
  """Generic proto wire reader: {field: [values]}; length-delimited values stay bytes."""
  import struct
  out, i = {}, 0
  while i < len(msg):
    key, shift = 0, 0
    while True:
      b = msg[i]; i += 1
      key |= (b & 0x7f) << shift; shift += 7
      if b < 0x80:
        break
    field, wire = key >> 3, key & 7
    if wire == 0:
      v, shift = 0, 0
      while True:
        b = msg[i]; i += 1
        v |= (b & 0x7f) << shift; shift += 7
        if b < 0x80:
          break
    elif wire == 5:
      v = struct.unpack("<f", msg[i:i + 4])[0]; i += 4
    elif wire == 2:
      n, shift = 0, 0
      while True:
        b = msg[i]; i += 1
        n |= (b & 0x7f) << shift; shift += 7
        if b < 0x80:
          break
      v = msg[i:i + n]; i += n
    else:
      raise AssertionError(("wire type", wire))
    out.setdefault(field, []).append(v)
  return out


import pytest  # noqa: E402


@pytest.mark.gpu
def test_building_logger_writes_what_the_environment_computed(tmp_path):
  import torch
  if not torch.cuda.is_available():
    pytest.skip("needs an MI355X")
  from sbsim_amd.environment import BatchedEnvironment, SimConfig
  from sbsim_amd.episode_writer import BuildingLogger
  from tests.test_gpu_parity import _plan
  p = load("plan_r9_sb1.npz")
  B, T = 5, 14      # 07:00 + 14 x 5 min: the shards roll over to the 08 hour
  env = BatchedEnvironment(_plan(p), B, config=SimConfig.sb1(), holiday_calendar=None, collect_info=True,
                           observation_normalization={"zone_air_temperature_sensor": (294.0, 4.0)})
  logger = BuildingLogger(env, str(tmp_path), buildings=[1, 3], scenario_id="unit")
  env.reset()
  logger.log_reset()
  rs = np.random.RandomState(1)
  rewards, zts, flows = [], [], []
  for t in range(T):
    a = torch.tensor(rs.uniform(-1, 1, size=(B, 2)).astype(np.float32), device="cuda")
    t0 = env.current_simulation_timestamp
    ts = env.step(a)
    logger.log_step(a, t0)
    rewards.append(ts.reward.cpu().numpy().copy()); zts.append(env.sim.zone_temps().cpu().numpy().copy())
    flows.append(env.sim.scalars()[:, 2].cpu().numpy().copy())
  env.close()
  d = tmp_path / "building_000003"
  assert sorted(os.listdir(d)) == [f"{pfx}_2023.07.06.{h}" for pfx in ("action_response", "observation_response",
                                   "reward_info", "reward_response") for h in ("07", "08")]
  resp = read_shard(str(d / "reward_response_2023.07.06.07")) + read_shard(str(d / "reward_response_2023.07.06.08"))
  infos = read_shard(str(d / "reward_info_2023.07.06.07")) + read_shard(str(d / "reward_info_2023.07.06.08"))
  obs = read_shard(str(d / "observation_response_2023.07.06.07")) + read_shard(str(d / "observation_response_2023.07.06.08"))
  acts = read_shard(str(d / "action_response_2023.07.06.07")) + read_shard(str(d / "action_response_2023.07.06.08"))
  assert (len(resp), len(infos), len(obs), len(acts)) == (T, T, T + 1, T)
  for t in range(T):
    r = _decode(resp[t])
    assert abs(r[1][0] - rewards[t][3]) < 1e-6 and r[12] == [1.0] and 13 not in r and 6 not in r
    start = _decode(r[18][0])[1][0]
    assert start == int(dt.datetime(2023, 7, 6, 7, 5, tzinfo=UTC).timestamp()) + 300 * t
    ri = _decode(infos[t])
    zones = {}
    for entry in ri[5]:
      e = _decode(entry)
      zones[e[1][0].decode()] = _decode(e[2][0])
    assert sorted(zones) == sorted(f"room_{i + 1}" for i in range(9)) and ri[4] == [b"unit"]
    for i in range(9):
      z = zones[f"room_{i + 1}"]
      assert abs(z[3][0] - zts[t][3, i]) < 1e-4 and abs(z[5][0] - flows[t][3]) < 1e-6 * max(1.0, flows[t][3])
      assert z[1][0] in (285.0, 294.0) and z[2][0] in (297.0, 303.0)      # eco / comfort window
  for k, want in ((0, 0.0), (1, 294.0)):   # at reset the VAVs have not read their zone yet (reference: 0.0)
    singles = [_decode(s) for s in _decode(obs[k])[3]]
    assert len(singles) == 27 + 9 + 3
    zone_sensor = [s for s in singles if _decode(s[2][0])[2][0] == b"zone_air_temperature_sensor"]
    assert len(zone_sensor) == 9 and all(abs(s[4][0] - want) < 1e-3 for s in zone_sensor)   # de-normalised
  a0 = _decode(acts[0])
  assert [_decode(_decode(s)[1][0])[2][0] for s in a0[3]] == [b"supply_water_setpoint", b"supply_air_heating_temperature_setpoint"]
  assert all(_decode(s)[2] == [1] for s in a0[3])          # ACCEPTED


def test_device_and_zone_info_records(tmp_path):
  """DeviceInfo / ZoneInfo (write_device_infos / write_zone_infos, controller_writer.py:149-171)."""
  from sbsim_amd import episode_writer as ew
  g = load("episode_shards.npz")
  vav = ew.encode_device_info("vav_room_1", "sim", "VAV-1", "room_1", ew.DEVICE_TYPES["VAV"],
                              {"supply_air_flowrate_setpoint": 1, "zone_air_temperature_sensor": 1,
                               "supply_air_damper_percentage_command": 1},
                              {"supply_air_damper_percentage_command": 1})
  assert vav == g["device_info_0"].tobytes()
  blr = ew.encode_device_info("boiler_id", "", "", "", ew.DEVICE_TYPES["BLR"], {}, {"supply_water_setpoint": 1})
  assert blr == g["device_info_1"].tobytes()
  z0 = ew.encode_zone_info("room_1", "US-SIM-001", "Simulated zone", 62.5, ["vav_room_1", "sensor_1"], floor=2)
  z1 = ew.encode_zone_info("room_2", "", "", 0.0, ["vav_room_2"], zone_type=0)
  assert z0 == g["zone_info_0"].tobytes() and z1 == g["zone_info_1"].tobytes()
  path = str(tmp_path / ew.ZONE_INFO_PREFIX)
  ew.write_records(path, [z1])            # an existing file is replaced, not extended
  ew.write_records(path, [z0, z1])
  assert open(path, "rb").read() == g["shard_zone_info"].tobytes()


def test_normalization_info_records(tmp_path):
  """ContinuousVariableInfo (write_normalization_info, controller_writer.py:134-147)."""
  from sbsim_amd import episode_writer as ew
  g = load("episode_shards.npz")
  times = _times(g)
  v0 = ew.encode_variable_info("zone_air_temperature_sensor", 1000, 4.0, 294.0, 293.5, 305.25, 285.0,
                               sample_start=times[0], sample_end=times[2])
  v1 = ew.encode_variable_info("supply_air_flowrate_sensor", 0, 0.25, 0.5)
  assert v0 == g["variable_info_0"].tobytes() and v1 == g["variable_info_1"].tobytes()
  path = str(tmp_path / ew.NORMALIZATION_FILENAME)
  ew.write_records(path, [v0, v1])
  assert open(path, "rb").read() == g["shard_normalization_info"].tobytes()


def test_building_image_rows_are_the_reference_writers(tmp_path):
  """utils/controller_writer.py:65-71 `write_building_image`: rows of building_images.csv.  The expected text is
  what the reference's own ProtoWriter wrote for these two calls (a naive timestamp counts as UTC, the image's
  bytes go through csv as their repr), run in the build container through oracle/refshim."""
  w = EpisodeWriter(str(tmp_path))
  w.write_building_image(b"aGVsbG8=", dt.datetime(2023, 7, 6, 7, 5, 0))
  w.write_building_image(b"d29ybGQ=", dt.datetime(2023, 7, 6, 8, 5, 0, tzinfo=UTC))
  with open(os.path.join(str(tmp_path), "building_images.csv")) as fh:
    assert fh.read() == "1688627100.0,b'aGVsbG8='\n1688630700.0,b'd29ybGQ='\n"
