"""Golden bytes for the episode-shard writer.

TEST INFRASTRUCTURE -- runs only in the build container: the reference's own message classes
(built from its .proto files by oracle/refshim, real protobuf runtime) serialize a fixed set of
RewardInfo / RewardResponse / ObservationResponse / ActionResponse messages, and the reference's
own ProtoWriter (utils/controller_writer.py) writes them to hourly shards.  Output:
tests/golden/episode_shards.npz -- the inputs as plain arrays, each message's
SerializeToString(deterministic=True) bytes, and the shard files ProtoWriter produced for the
message types whose serialization does not depend on map iteration order.

    python -m oracle.gen_golden_shards
"""
from __future__ import annotations

import os
import tempfile

import numpy as np

from oracle import refshim

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def main() -> None:
  refshim.install()
  rpb = refshim.ref("proto.smart_control_reward_pb2")
  bpb = refshim.ref("proto.smart_control_building_pb2")
  cw = refshim.ref("utils.controller_writer")
  conv = refshim.ref("utils.conversion_utils")
  import pandas as pd

  rs = np.random.RandomState(12)
  out = {}
  times = [pd.Timestamp("2023-07-06 07:55:00", tz="UTC"), pd.Timestamp("2023-07-06 08:00:00", tz="UTC"),
           pd.Timestamp("2023-07-06 08:05:00.250", tz="UTC")]
  out["times_s"] = np.array([int(t.timestamp()) for t in times], dtype=np.int64)
  out["times_ns"] = np.array([int(round((t.timestamp() % 1) * 1e9)) for t in times], dtype=np.int64)
  zone_ids = ["zone_id_(%d, %d)" % (i // 3, i % 3) for i in range(9)][::-1]   # not in key order
  zone_vals = rs.uniform(0.0, 300.0, size=(3, 9, 6)).astype(np.float32)
  zone_vals[0, 2, 5] = 0.0          # proto3 default: omitted on the wire
  zone_vals[1, 4, 3] = -0.0         # negative zero: written
  ahu_vals = rs.uniform(-5e4, 5e4, size=(3, 1, 2)).astype(np.float32)
  blr_vals = rs.uniform(0.0, 9e4, size=(3, 1, 2)).astype(np.float32)
  resp_vals = rs.uniform(-2.0, 2.0, size=(3, 17)).astype(np.float32)
  resp_vals[2, 6] = 0.0
  out.update(zone_ids=np.array(zone_ids), zone_vals=zone_vals, ahu_vals=ahu_vals, blr_vals=blr_vals,
             resp_vals=resp_vals)
  dev_ids = ["vav_room_%d" % (i + 1) for i in range(4)] + ["air_handler_id", "boiler_id"]
  meas = ["zone_air_temperature_sensor"] * 4 + ["supply_air_flowrate_sensor", "supply_water_setpoint"]
  obs_vals = rs.uniform(0.0, 350.0, size=(3, 6)).astype(np.float32)
  obs_vals[1, 4] = 0.0              # oneof member: written although zero
  obs_valid = np.ones((3, 6), dtype=np.uint8)
  obs_valid[2, 1] = 0
  act_dev = ["boiler_id", "air_handler_id"]
  act_names = ["supply_water_setpoint", "supply_air_heating_temperature_setpoint"]
  act_vals = rs.uniform(285.0, 355.0, size=(3, 2)).astype(np.float32)
  act_types = np.array([[1, 1], [1, 5], [7, 1]], dtype=np.int32)
  out.update(obs_dev=np.array(dev_ids), obs_meas=np.array(meas), obs_vals=obs_vals, obs_valid=obs_valid,
             act_dev=np.array(act_dev), act_names=np.array(act_names), act_vals=act_vals, act_types=act_types)

  tmp = tempfile.mkdtemp()
  writer = cw.ProtoWriter(tmp)
  for k, t in enumerate(times):
    end = t + pd.Timedelta(300, unit="s")
    info = rpb.RewardInfo(start_timestamp=conv.pandas_to_proto_timestamp(t),
                          end_timestamp=conv.pandas_to_proto_timestamp(end),
                          agent_id="agent" if k else "", scenario_id="s%d" % k)
    for zi, z in enumerate(zone_ids):
      v = zone_vals[k, zi]
      info.zone_reward_infos[z].CopyFrom(rpb.RewardInfo.ZoneRewardInfo(
          heating_setpoint_temperature=v[0], cooling_setpoint_temperature=v[1], zone_air_temperature=v[2],
          air_flow_rate_setpoint=v[3], air_flow_rate=v[4], average_occupancy=v[5]))
    info.air_handler_reward_infos["air_handler_id"].CopyFrom(rpb.RewardInfo.AirHandlerRewardInfo(
        blower_electrical_energy_rate=ahu_vals[k, 0, 0], air_conditioning_electrical_energy_rate=ahu_vals[k, 0, 1]))
    info.boiler_reward_infos["boiler_id"].CopyFrom(rpb.RewardInfo.BoilerRewardInfo(
        natural_gas_heating_energy_rate=blr_vals[k, 0, 0], pump_electrical_energy_rate=blr_vals[k, 0, 1]))
    out[f"reward_info_{k}"] = np.frombuffer(info.SerializeToString(deterministic=True), dtype=np.uint8)

    names = [f.name for f in rpb.RewardResponse.DESCRIPTOR.fields if f.number <= 17]
    resp = rpb.RewardResponse(**{n: float(resp_vals[k, i]) for i, n in enumerate(names)})
    resp.start_timestamp.CopyFrom(conv.pandas_to_proto_timestamp(t))
    resp.end_timestamp.CopyFrom(conv.pandas_to_proto_timestamp(end))
    out[f"reward_response_{k}"] = np.frombuffer(resp.SerializeToString(deterministic=True), dtype=np.uint8)
    writer.write_reward_response(resp, t)

    req = bpb.ObservationRequest()
    for d, m in zip(dev_ids, meas):
      req.single_observation_requests.append(bpb.SingleObservationRequest(device_id=d, measurement_name=m))
    obs = bpb.ObservationResponse()
    obs.request.CopyFrom(req)
    obs.timestamp.CopyFrom(conv.pandas_to_proto_timestamp(t))
    for i, sreq in enumerate(req.single_observation_requests):   # simulator_building.py:151-202
      r = bpb.SingleObservationResponse()
      r.single_observation_request.CopyFrom(sreq)
      r.timestamp.CopyFrom(conv.pandas_to_proto_timestamp(t))
      r.observation_valid = bool(obs_valid[k, i])
      if obs_valid[k, i]:
        r.continuous_value = float(obs_vals[k, i])
      obs.single_observation_responses.append(r)
    out[f"observation_response_{k}"] = np.frombuffer(obs.SerializeToString(deterministic=True), dtype=np.uint8)
    writer.write_observation_response(obs, t)

    areq = bpb.ActionRequest()
    areq.timestamp.CopyFrom(conv.pandas_to_proto_timestamp(t))
    for i in range(2):
      areq.single_action_requests.append(bpb.SingleActionRequest(
          device_id=act_dev[i], setpoint_name=act_names[i], continuous_value=float(act_vals[k, i])))
    aresp = bpb.ActionResponse()
    aresp.request.CopyFrom(areq)
    aresp.timestamp.CopyFrom(conv.pandas_to_proto_timestamp(t))
    for i, sreq in enumerate(areq.single_action_requests):       # simulator_building.py:204-263
      r = bpb.SingleActionResponse()
      r.request.CopyFrom(sreq)
      r.response_type = int(act_types[k, i])
      aresp.single_action_responses.append(r)
    out[f"action_response_{k}"] = np.frombuffer(aresp.SerializeToString(deterministic=True), dtype=np.uint8)
    writer.write_action_response(aresp, t)

  # device / zone infos
  dev = bpb.DeviceInfo(device_id="vav_room_1", namespace="sim", code="VAV-1", zone_id="room_1",
                       device_type=bpb.DeviceInfo.DeviceType.VAV)
  for n in ("zone_air_temperature_sensor", "supply_air_damper_percentage_command", "supply_air_flowrate_setpoint"):
    dev.observable_fields[n] = bpb.DeviceInfo.ValueType.VALUE_CONTINUOUS
  dev.action_fields["supply_air_damper_percentage_command"] = bpb.DeviceInfo.ValueType.VALUE_CONTINUOUS
  blr = bpb.DeviceInfo(device_id="boiler_id", device_type=bpb.DeviceInfo.DeviceType.BLR)
  blr.action_fields["supply_water_setpoint"] = bpb.DeviceInfo.ValueType.VALUE_CONTINUOUS
  zone = bpb.ZoneInfo(zone_id="room_1", building_id="US-SIM-001", zone_description="Simulated zone", area=62.5,
                      devices=["vav_room_1", "sensor_1"], zone_type=bpb.ZoneInfo.ZoneType.ROOM, floor=2)
  zone0 = bpb.ZoneInfo(zone_id="room_2", devices=["vav_room_2"])
  out["device_info_0"] = np.frombuffer(dev.SerializeToString(deterministic=True), dtype=np.uint8)
  out["device_info_1"] = np.frombuffer(blr.SerializeToString(deterministic=True), dtype=np.uint8)
  out["zone_info_0"] = np.frombuffer(zone.SerializeToString(deterministic=True), dtype=np.uint8)
  out["zone_info_1"] = np.frombuffer(zone0.SerializeToString(deterministic=True), dtype=np.uint8)
  writer.write_zone_infos([zone, zone0])

  npb = refshim.ref("proto.smart_control_normalization_pb2")
  v0 = npb.ContinuousVariableInfo(id="zone_air_temperature_sensor", sample_size=1000, sample_variance=4.0,
                                  sample_mean=294.0, sample_median=293.5, sample_maximum=305.25, sample_minimum=285.0)
  v0.sample_start.CopyFrom(conv.pandas_to_proto_timestamp(times[0]))
  v0.sample_end.CopyFrom(conv.pandas_to_proto_timestamp(times[2]))
  v1 = npb.ContinuousVariableInfo(id="supply_air_flowrate_sensor", sample_variance=0.25, sample_mean=0.5)
  out["variable_info_0"] = np.frombuffer(v0.SerializeToString(deterministic=True), dtype=np.uint8)
  out["variable_info_1"] = np.frombuffer(v1.SerializeToString(deterministic=True), dtype=np.uint8)
  writer.write_normalization_info({"a": v0, "b": v1})

  files = sorted(os.listdir(tmp))
  out["shard_names"] = np.array(files)
  for f in files:
    out["shard_" + f] = np.frombuffer(open(os.path.join(tmp, f), "rb").read(), dtype=np.uint8)
  np.savez_compressed(os.path.join(GOLD, "episode_shards.npz"), **out)
  print(files, {k: v.shape for k, v in out.items() if k.startswith(("reward_info", "shard_"))})


if __name__ == "__main__":
  main()
