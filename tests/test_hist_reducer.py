"""Observation HistogramReducer: oracle and host layout against the
reference's own outputs (tests/golden/hist_reducer.*, made by oracle/gen_golden_hist.py), and
the device-side reducer against the same vectors on the GPU."""
import json
import os

import numpy as np
import pytest

from tests.golden_util import GOLDEN as GOLDEN_DIR, load
from oracle import hist_oracle                                   # checker only
from sbsim_amd.environment import histogram_reducer_layout, observation_field_names


def _meta():
  with open(os.path.join(GOLDEN_DIR, "hist_reducer.json")) as fh:
    return json.load(fh)


@pytest.mark.parametrize("case", ["sb1_bins", "normalised"])
def test_oracle_reproduces_the_reference_reducer(case):
  g, m = load("hist_reducer.npz"), _meta()
  cfg = m["cases"][case]
  params = [(n, b) for n, b in cfg["params"]]
  assert hist_oracle.field_order(m["source_names"], params) == cfg["field_order"]
  for i in range(len(m["steps"])):
    out = hist_oracle.reduce(m["source_names"], g[f"{case}_inputs"][i], params, cfg["normalize_reduce"])
    got = np.array([out[k] for k in cfg["field_order"]], np.float32)
    assert np.array_equal(got, g[f"{case}_expected"][i]), (case, i)   # bit-exact, counts and pass-throughs


def test_clipped_histogram_edges():
  """histogram_reducer.py:136-146 (and its docstring :221-231): values below the first bin and
  at or above the last one are clipped into the end bins; a value on an edge opens its bin."""
  bins = [285.0, 286.0, 287.0, 288.0]
  h = hist_oracle.clipped_histogram(np.array([100.0, 285.0, 285.999, 286.0, 287.5, 288.0, 999.0]), bins)
  assert h.tolist() == [3.0, 1.0, 1.0, 2.0]
  assert hist_oracle.clipped_histogram(np.array([]), bins).tolist() == [0.0] * 4


@pytest.mark.parametrize("case", ["sb1_bins", "normalised"])
def test_host_layout_matches_the_reference_field_order(case):
  m = _meta()
  cfg = m["cases"][case]
  zone_names = [f"room_{i + 1}" for i in range(9)]
  names, _, _, _, n_src = observation_field_names(zone_names, True)
  assert names[:n_src] == m["source_names"]
  out, src_dest, hist_col, hist_off, hist_bins = histogram_reducer_layout(names[:n_src], cfg["params"])
  # our pass-through ids keep the "device/field" spelling of the rest of the package
  assert [n.replace("/", "_") for n in out] == cfg["field_order"]
  for name, b in cfg["params"]:       # histograms are numbered in the order they are first met
    members = [i for i, s in enumerate(names[:n_src]) if s.endswith("/" + name)]
    k = -src_dest[members[0]] - 1
    assert all(src_dest[i] == -(k + 1) for i in members) and len(members) == 9
    assert hist_bins[hist_off[k]:hist_off[k + 1]] == [float(v) for v in b]
    assert out[hist_col[k]] == f"{name}_h_{b[0]:.2f}"
  passed = [d for d in src_dest if d >= 0]
  assert len(set(passed)) == len(passed) and len(passed) + len(hist_bins) == len(out)
  with pytest.raises(ValueError):
    histogram_reducer_layout(names[:n_src], [("zone_air_temperature_sensor", [2.0, 1.0])])


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["sb1_bins", "normalised"])
def test_device_reducer_against_reference_golden(case):
  """The 288-step reference rollout on the GPU with the reducer enabled: the observation rows at
  the golden steps equal the reference's HistogramReducer output (counts exactly, pass-through
  values to fp32 round-off)."""
  import torch
  if not torch.cuda.is_available():
    pytest.skip("needs a GPU")
  from sbsim_amd import _ffi
  from sbsim_amd.environment import BatchedSimulator, SimConfig
  from tests.test_gpu_parity import _plan, _step_in
  g, gh, m = load("h2_sb1_r9_random.npz"), load("hist_reducer.npz"), _meta()
  cfg = m["cases"][case]
  norm = None
  if "mean" in cfg:   # the affine normalisation the golden inputs were made with (mean, variance)
    norm = {n.split("/", 1)[1]: (mu, sg * sg) for n, mu, sg in zip(m["source_names"], cfg["mean"], cfg["sigma"])}
  B = 3
  sim = BatchedSimulator(_plan(load("plan_r9_sb1.npz")), SimConfig.sb1(), B, float(g["h_conv"]),
                         observation_normalization=norm, zone_names=[str(z) for z in g["zone_names"]],
                         histogram_parameters=[(n, b) for n, b in cfg["params"]],
                         normalize_reduce=cfg["normalize_reduce"])
  n_dev = len(cfg["field_order"])
  assert sim.O == n_dev + _ffi.SB_NUM_AUX
  assert [n.replace("/", "_") for n in sim.field_names[:n_dev]] == cfg["field_order"]
  sim.reset()
  obs = torch.zeros((B, sim.O), dtype=torch.float32, device="cuda")
  rew = torch.zeros((B,), dtype=torch.float32, device="cuda")
  is_count = np.array(["_h_" in n for n in cfg["field_order"]])
  checked = 0
  for t in range(max(m["steps"]) + 1):
    act = torch.tensor(np.tile(g["actions_norm"][t], (B, 1)), dtype=torch.float32, device="cuda")
    sim.step(act, _step_in(g, t), obs, rew, None)
    if t in m["steps"]:
      want = gh[f"{case}_expected"][m["steps"].index(t)]
      got = obs.cpu().numpy()[:, :n_dev]
      assert np.array_equal(got[:, is_count], np.tile(want[is_count], (B, 1))), (case, t)
      assert np.allclose(got[:, ~is_count], want[~is_count], rtol=2e-6, atol=2e-6), (case, t)
      if not cfg["normalize_reduce"]:
        assert got[:, is_count].sum() == B * 9 * len(cfg["params"])
      checked += 1
  assert checked == len(m["steps"])
