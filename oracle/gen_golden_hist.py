"""Golden vectors for the observation HistogramReducer.

TEST INFRASTRUCTURE -- runs only in the build container: it imports the REFERENCE
(`/root/reference`, through oracle/refshim) and runs its own `HistogramReducer.reduce` exactly
the way `Environment._normalized_observation_response_to_observation_map_histogram_reducer`
does (environment.py:1032-1071), on observation responses rebuilt from the committed H2
golden rollout.  Output: tests/golden/hist_reducer.npz + hist_reducer.json (inputs, bins,
field order of `_get_observation_spec_histogram_reducer`, expected vectors).

    python -m oracle.gen_golden_hist
"""
from __future__ import annotations

import json
import os

import numpy as np

from oracle import refshim

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")

# sim_config.gin:586-590 (the SB1 bins), applied to R9's devices; a second case with bins that
# straddle NORMALISED values and normalize_reduce=True
CASES = {
    "sb1_bins": dict(
        params=[("zone_air_temperature_sensor", [285., 286., 287., 288, 289., 290., 291., 292., 293., 294., 295.,
                                                  296., 297., 298., 299., 300., 301, 302, 303]),
                ("supply_air_damper_percentage_command", [0.0, 0.2, 0.4, 0.6, 0.8, 1.0]),
                ("supply_air_flowrate_setpoint", [0., 0.05, .1, .2, .3, .4, .5, .7, .9])],
        normalize_reduce=False, normalized=False),
    "normalised": dict(
        params=[("zone_air_temperature_sensor", [-2.0, -1.0, -0.5, -0.25, 0.0, 0.25, 0.5, 1.0, 2.0]),
                ("supply_air_damper_percentage_command", [-2.0, -1.0, 1.0, 2.0])],
        normalize_reduce=True, normalized=True),
}


def main() -> None:
  refshim.install()
  pb = refshim.ref("proto.smart_control_building_pb2")
  hr = refshim.ref("utils.histogram_reducer")
  rbu = refshim.ref("utils.regression_building_utils")
  reader_lib = refshim.ref("utils.reader_lib")
  conv = refshim.ref("utils.conversion_utils")
  import pandas as pd

  g = np.load(os.path.join(GOLD, "h2_sb1_r9_random.npz"), allow_pickle=True)
  names = [str(n) for n in g["obs_names"]]
  steps = [0, 40, 100, 101, 150, 200, 287]
  start = pd.Timestamp(str(g["start_timestamp"]))

  class Reader(reader_lib.BaseReader):  # only what HistogramReducer.__init__ asks for
    def __init__(self, responses): self._r = responses
    def read_observation_responses(self, start_time, end_time): return self._r
    def read_action_responses(self, start_time, end_time): raise NotImplementedError()
    def read_reward_infos(self, start_time, end_time): raise NotImplementedError()
    def read_reward_responses(self, start_time, end_time): raise NotImplementedError()
    def read_normalization_info(self): raise NotImplementedError()
    def read_zone_infos(self): raise NotImplementedError()
    def read_device_infos(self): raise NotImplementedError()

  def response(values, t):
    r = pb.ObservationResponse()
    ts = start + pd.Timedelta(300 * (t + 1), unit="s")
    r.timestamp.CopyFrom(conv.pandas_to_proto_timestamp(ts))
    for n, v in zip(names, values):
      dev, meas = n.split("/", 1)
      s = r.single_observation_responses.add()
      s.single_observation_request.device_id = dev
      s.single_observation_request.measurement_name = meas
      s.observation_valid = True
      s.continuous_value = float(v)      # proto float: fp32
    return r

  out, meta = {}, {"steps": steps, "source_names": names, "cases": {}}
  for case, cfg in CASES.items():
    vals = g["obs"][steps].astype(np.float64)
    if cfg["normalized"]:   # a fixed affine normalisation per field, rounded to fp32 like the proto field
      mu = np.array([294.0 if "temperature" in n else 0.5 for n in names])
      sg = np.array([3.0 if "temperature" in n else 0.25 for n in names])   # variances 9 and 1/16: exact in fp32
      vals = ((vals - mu) / sg).astype(np.float32).astype(np.float64)
      meta["cases"][case] = {"mean": mu.tolist(), "sigma": sg.tolist()}
    else:
      meta["cases"][case] = {}
    responses = [response(vals[i], t) for i, t in enumerate(steps)]
    reducer = hr.HistogramReducer(
        histogram_parameters_tuples=[(n, np.array(b, dtype=np.float64)) for n, b in cfg["params"]],
        reader=Reader(responses), normalize_reduce=cfg["normalize_reduce"])
    rows, keys = [], None
    for r in responses:   # environment.py:1050-1071
      seq = rbu.get_observation_sequence([r], rbu.get_feature_tuples(r), "UTC", 1, 1)
      rs = reducer.reduce(seq).reduced_sequence
      m = rs.iloc[0].to_dict()
      m = {"_".join(k): m[k] for k in m if isinstance(k, tuple)}
      # the sequence's own hod/dow columns are overwritten by Environment._get_observation (:917-941)
      m = {k: v for k, v in m.items() if not k.startswith(("hod_", "dow_"))}
      keys = keys or list(m.keys())
      rows.append([np.float32(m[k]) for k in keys])
    # field order of Environment._get_observation_spec_histogram_reducer (environment.py:731-777)
    devices = sorted({n.split("/", 1)[0] for n in names})
    order, seen = [], set()
    for dev in devices:
      for meas in sorted(n.split("/", 1)[1] for n in names if n.split("/", 1)[0] == dev):
        if meas in dict(cfg["params"]):
          if meas not in seen:
            seen.add(meas)
            order += [f"{meas}_h_{v:.2f}" for v in dict(cfg["params"])[meas]]
        else:
          order.append(f"{dev}_{meas}")
    assert set(order) == set(keys), (set(order) ^ set(keys))
    table = np.array(rows, dtype=np.float32)
    out[f"{case}_inputs"] = vals.astype(np.float32)
    out[f"{case}_expected"] = table[:, [keys.index(k) for k in order]]
    meta["cases"][case].update(dict(
        params=[[n, [float(x) for x in b]] for n, b in cfg["params"]],
        normalize_reduce=cfg["normalize_reduce"], field_order=order))
    print(case, "->", table.shape, "fields", len(order))
  np.savez_compressed(os.path.join(GOLD, "hist_reducer.npz"), **out)
  with open(os.path.join(GOLD, "hist_reducer.json"), "w") as fh:
    json.dump(meta, fh, indent=1)


if __name__ == "__main__":
  main()
