// episode.cpp -- ProtoWriter-compatible episode shards, host-only.
//
// The reference logs four message types per step, each as a 4-byte little-endian size followed by
// the serialized proto, appended to hourly files "<prefix>_YYYY.MM.DD.HH"
// (utils/controller_writer.py:53-131, prefixes utils/constants.py:51-56); ProtoReader
// (utils/controller_reader.py) and the plotting tools read those.  This file writes the same
// files without protoc or the protobuf library: the four messages are encoded by hand from
// plain arrays (proto3 wire format, fields in number order, map entries in key order = what
// SerializeToString(deterministic=True) emits).  Message layouts:
// proto/smart_control_reward.proto:49-185 (RewardInfo, RewardResponse),
// proto/smart_control_building.proto (ObservationResponse, ActionResponse and their parts).
#include "sbsim_amd.h"

#include <algorithm>
#include <cerrno>
#include <cstdio>
#include <cstring>
#include <ctime>
#include <string>
#include <vector>

#include <sys/stat.h>

namespace {

using Buf = std::string;

void varint(Buf &b, uint64_t v) {
  while (v >= 0x80) { b.push_back((char)(v | 0x80)); v >>= 7; }
  b.push_back((char)v);
}
void tag(Buf &b, int field, int wire) { varint(b, (uint64_t)field << 3 | (uint64_t)wire); }
// proto3 scalar fields outside a oneof are omitted at their default value
void f_float(Buf &b, int field, float v, bool always = false) {
  uint32_t u;
  std::memcpy(&u, &v, 4);
  if (!always && u == 0) return; // +0.0 only: -0.0 has a bit set and is written
  tag(b, field, 5);
  for (int i = 0; i < 4; ++i) b.push_back((char)(u >> (8 * i)));
}
void f_varint(Buf &b, int field, int64_t v, bool always = false) {
  if (!always && v == 0) return;
  tag(b, field, 0);
  varint(b, (uint64_t)v); // negative int32 / int64: ten bytes, as protobuf does
}
void f_bytes(Buf &b, int field, const char *s, size_t n, bool always = false) {
  if (!always && n == 0) return;
  tag(b, field, 2);
  varint(b, n);
  b.append(s, n);
}
void f_str(Buf &b, int field, const char *s, bool always = false) { f_bytes(b, field, s ? s : "", s ? std::strlen(s) : 0, always); }
void f_msg(Buf &b, int field, const Buf &m) { f_bytes(b, field, m.data(), m.size(), true); } // a set message is written even when empty

Buf timestamp(sb_pb_time t) { // google.protobuf.Timestamp
  Buf m;
  f_varint(m, 1, t.seconds);
  f_varint(m, 2, t.nanos);
  return m;
}

// map<string, Msg>: one entry message {1: key, 2: value} per key, keys in byte order
void f_map(Buf &b, int field, int n, const char *const *keys, const std::vector<Buf> &vals) {
  std::vector<int> order(n);
  for (int i = 0; i < n; ++i) order[i] = i;
  std::sort(order.begin(), order.end(), [&](int a, int c) { return std::strcmp(keys[a], keys[c]) < 0; });
  for (int i : order) {
    Buf e;
    f_str(e, 1, keys[i], true);
    f_msg(e, 2, vals[i]);
    f_msg(b, field, e);
  }
}

int64_t emit(const Buf &m, uint8_t *out, int64_t cap) {
  if (out && (int64_t)m.size() <= cap) std::memcpy(out, m.data(), m.size());
  return (int64_t)m.size(); // the size needed; nothing is written when it exceeds cap
}

} // namespace

extern "C" {

int64_t sb_pb_reward_info(sb_pb_time start, sb_pb_time end, const char *agent_id, const char *scenario_id,
                          int32_t n_zones, const char *const *zone_ids, const float *zone_vals,
                          int32_t n_ahu, const char *const *ahu_ids, const float *ahu_vals,
                          int32_t n_blr, const char *const *blr_ids, const float *blr_vals,
                          uint8_t *out, int64_t cap) {
  if (n_zones < 0 || n_ahu < 0 || n_blr < 0 || (n_zones && (!zone_ids || !zone_vals)) ||
      (n_ahu && (!ahu_ids || !ahu_vals)) || (n_blr && (!blr_ids || !blr_vals)))
    return SB_ERR_INVALID;
  Buf m;
  f_msg(m, 1, timestamp(start));
  f_msg(m, 2, timestamp(end));
  f_str(m, 3, agent_id);
  f_str(m, 4, scenario_id);
  auto group = [&](int field, int n, const char *const *ids, const float *vals, int width) {
    std::vector<Buf> v(n);
    for (int i = 0; i < n; ++i)
      for (int k = 0; k < width; ++k) f_float(v[i], k + 1, vals[i * width + k]);
    f_map(m, field, n, ids, v);
  };
  group(5, n_zones, zone_ids, zone_vals, 6); // ZoneRewardInfo: heating sp, cooling sp, zone T, flow sp, flow, occupancy
  group(6, n_ahu, ahu_ids, ahu_vals, 2);     // AirHandlerRewardInfo: blower W, air-conditioning W
  group(7, n_blr, blr_ids, blr_vals, 2);     // BoilerRewardInfo: gas W, pump W
  return emit(m, out, cap);
}

int64_t sb_pb_reward_response(const float vals[17], sb_pb_time start, sb_pb_time end, uint8_t *out, int64_t cap) {
  if (!vals) return SB_ERR_INVALID;
  Buf m;
  for (int k = 0; k < 17; ++k) f_float(m, k + 1, vals[k]); // fields 1..17 in proto order
  f_msg(m, 18, timestamp(start));
  f_msg(m, 19, timestamp(end));
  return emit(m, out, cap);
}

int64_t sb_pb_observation_response(sb_pb_time ts, int32_t n, const char *const *device_ids,
                                   const char *const *measurement_names, const float *values,
                                   const uint8_t *valid, uint8_t *out, int64_t cap) {
  if (n < 0 || (n && (!device_ids || !measurement_names || !values || !valid))) return SB_ERR_INVALID;
  Buf m, req;
  const Buf t = timestamp(ts);
  f_msg(m, 1, t);
  std::vector<Buf> single(n);
  for (int i = 0; i < n; ++i) { // SingleObservationRequest {device_id, measurement_name}
    f_str(single[i], 1, device_ids[i]);
    f_str(single[i], 2, measurement_names[i]);
    f_msg(req, 2, single[i]);
  }
  f_msg(m, 2, req); // the ObservationRequest as the environment built it: no timestamp
  for (int i = 0; i < n; ++i) {
    Buf r;
    f_msg(r, 1, t);
    f_msg(r, 2, single[i]);
    f_varint(r, 3, valid[i] ? 1 : 0);
    if (valid[i]) f_float(r, 4, values[i], true); // oneof member: written even when 0
    f_msg(m, 3, r);
  }
  return emit(m, out, cap);
}

int64_t sb_pb_action_response(sb_pb_time ts, sb_pb_time request_ts, int32_t n, const char *const *device_ids,
                              const char *const *setpoint_names, const float *values,
                              const int32_t *response_types, uint8_t *out, int64_t cap) {
  if (n < 0 || (n && (!device_ids || !setpoint_names || !values || !response_types))) return SB_ERR_INVALID;
  Buf m, req;
  f_msg(m, 1, timestamp(ts));
  f_msg(req, 1, timestamp(request_ts)); // environment.py:834-850: the ActionRequest carries the step's time
  std::vector<Buf> single(n);
  for (int i = 0; i < n; ++i) { // SingleActionRequest {device_id, setpoint_name, oneof continuous_value}
    f_str(single[i], 1, device_ids[i]);
    f_str(single[i], 2, setpoint_names[i]);
    f_float(single[i], 3, values[i], true);
    f_msg(req, 2, single[i]);
  }
  f_msg(m, 2, req);
  for (int i = 0; i < n; ++i) {
    Buf r;
    f_msg(r, 1, single[i]);
    f_varint(r, 2, response_types[i]);
    f_msg(m, 3, r);
  }
  return emit(m, out, cap);
}

// DeviceInfo / ZoneInfo (proto/smart_control_building.proto): what ProtoWriter.write_device_infos /
// write_zone_infos put, record by record, into the files "device_info" / "zone_info".
int64_t sb_pb_device_info(const char *device_id, const char *name_space, const char *code, const char *zone_id,
                          int32_t device_type, int32_t n_observable, const char *const *observable_names,
                          const int32_t *observable_types, int32_t n_action, const char *const *action_names,
                          const int32_t *action_types, uint8_t *out, int64_t cap) {
  if (n_observable < 0 || n_action < 0 || (n_observable && (!observable_names || !observable_types)) ||
      (n_action && (!action_names || !action_types)))
    return SB_ERR_INVALID;
  Buf m;
  f_str(m, 1, device_id);
  f_str(m, 2, name_space);
  f_str(m, 3, code);
  f_str(m, 4, zone_id);
  f_varint(m, 5, device_type);
  auto fields = [&](int field, int n, const char *const *names, const int32_t *types) { // map<string, ValueType>
    std::vector<int> order(n);
    for (int i = 0; i < n; ++i) order[i] = i;
    std::sort(order.begin(), order.end(), [&](int a, int c) { return std::strcmp(names[a], names[c]) < 0; });
    for (int i : order) {
      Buf e;
      f_str(e, 1, names[i], true);
      f_varint(e, 2, types[i], true);
      f_msg(m, field, e);
    }
  };
  fields(6, n_observable, observable_names, observable_types);
  fields(7, n_action, action_names, action_types);
  return emit(m, out, cap);
}

int64_t sb_pb_zone_info(const char *zone_id, const char *building_id, const char *zone_description, float area,
                        int32_t n_devices, const char *const *devices, int32_t zone_type, int32_t floor,
                        uint8_t *out, int64_t cap) {
  if (n_devices < 0 || (n_devices && !devices)) return SB_ERR_INVALID;
  Buf m;
  f_str(m, 1, zone_id);
  f_str(m, 2, building_id);
  f_str(m, 3, zone_description);
  f_float(m, 4, area);
  for (int i = 0; i < n_devices; ++i) f_str(m, 5, devices[i], true); // repeated: every element is written
  f_varint(m, 6, zone_type);
  f_varint(m, 7, floor);
  return emit(m, out, cap);
}

// ContinuousVariableInfo (proto/smart_control_normalization.proto): the records of
// ProtoWriter.write_normalization_info (file "normalization_info").  stats: variance, mean,
// median, maximum, minimum.  A timestamp with has_* == 0 is left unset.
int64_t sb_pb_variable_info(const char *id, int32_t has_start, sb_pb_time start, int32_t has_end, sb_pb_time end,
                            int32_t sample_size, const float stats[5], uint8_t *out, int64_t cap) {
  if (!stats) return SB_ERR_INVALID;
  Buf m;
  f_str(m, 1, id);
  if (has_start) f_msg(m, 2, timestamp(start));
  if (has_end) f_msg(m, 3, timestamp(end));
  f_varint(m, 4, sample_size);
  for (int k = 0; k < 5; ++k) f_float(m, 5 + k, stats[k]);
  return emit(m, out, cap);
}

// One length-prefixed record appended to (or, with truncate, starting) the file `path`.
int sb_record_append(const char *path, const uint8_t *msg, int64_t n, int32_t truncate) {
  if (!path || (!msg && n) || n < 0 || n > 0x7fffffffLL) return SB_ERR_INVALID;
  FILE *f = std::fopen(path, truncate ? "wb" : "ab");
  if (!f) return SB_ERR_INVALID;
  const uint32_t size = (uint32_t)n;
  unsigned char le[4] = {(unsigned char)size, (unsigned char)(size >> 8), (unsigned char)(size >> 16), (unsigned char)(size >> 24)};
  const bool ok = std::fwrite(le, 1, 4, f) == 4 && (n == 0 || std::fwrite(msg, 1, (size_t)n, f) == (size_t)n);
  return (std::fclose(f) == 0 && ok) ? SB_OK : SB_ERR_INVALID;
}

// ProtoWriter._write_msg_to_disk (controller_writer.py:118-131): append size (4 bytes, little
// endian) + message to <dir>/<prefix>_YYYY.MM.DD.HH of the timestamp's hour (its own clock:
// the caller passes the seconds of the timestamp as strftime would see it).
int sb_shard_append(const char *dir, const char *prefix, int64_t unix_seconds, const uint8_t *msg, int64_t n) {
  if (!dir || !prefix || (!msg && n) || n < 0 || n > 0x7fffffffLL) return SB_ERR_INVALID;
  if (mkdir(dir, 0777) != 0 && errno != EEXIST) return SB_ERR_INVALID;
  const time_t t = (time_t)unix_seconds;
  struct tm tmv;
  if (!gmtime_r(&t, &tmv)) return SB_ERR_INVALID;
  char serial[32];
  std::strftime(serial, sizeof serial, "%Y.%m.%d.%H", &tmv);
  const std::string path = std::string(dir) + "/" + prefix + "_" + serial;
  FILE *f = std::fopen(path.c_str(), "ab");
  if (!f) return SB_ERR_INVALID;
  const uint32_t size = (uint32_t)n;
  unsigned char le[4] = {(unsigned char)size, (unsigned char)(size >> 8), (unsigned char)(size >> 16), (unsigned char)(size >> 24)};
  const bool ok = std::fwrite(le, 1, 4, f) == 4 && (n == 0 || std::fwrite(msg, 1, (size_t)n, f) == (size_t)n);
  return (std::fclose(f) == 0 && ok) ? SB_OK : SB_ERR_INVALID;
}

} // extern "C"
