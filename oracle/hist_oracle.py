"""CPU restatement of the observation HistogramReducer (TEST INFRASTRUCTURE: checker only).

Follows the reference (all under /root/reference/smart_control):
  utils/histogram_reducer.py:136-146   get_clipped_histogram: clip to [min(bins), max(bins)], then
                                       np.histogram with the edges bins + [max(bins)]
  utils/histogram_reducer.py:412-436   one histogram per reduced feature over all devices that
                                       report it; normalize_reduce divides by the count
  environment/environment.py:731-777   field order: devices sorted by id, fields sorted; a reduced
                                       feature contributes all its bins where it is first met
  environment/environment.py:1032-1071 the reducer runs on the NORMALISED observation response
Pinned against tests/golden/hist_reducer.* (outputs of the reference's own class)."""
from __future__ import annotations

from typing import Dict, List, Sequence, Tuple

import numpy as np


def field_order(source_names: Sequence[str], params: Sequence[Tuple[str, Sequence[float]]]) -> List[str]:
  """source_names: 'device/measurement' in sorted (device, measurement) order."""
  bins = {n: list(b) for n, b in params}
  order, seen = [], set()
  for name in source_names:
    dev, meas = name.split("/", 1)
    if meas in bins:
      if meas not in seen:
        seen.add(meas)
        order += [f"{meas}_h_{v:.2f}" for v in bins[meas]]
    else:
      order.append(f"{dev}_{meas}")
  return order


def clipped_histogram(measurements: np.ndarray, bins: Sequence[float]) -> np.ndarray:
  b = np.asarray(bins, dtype=np.float64)
  m = np.clip(np.asarray(measurements, dtype=np.float64), b.min(), b.max())
  idx = np.searchsorted(b[1:], m, side="right")          # bin i: b[i] <= v < b[i+1]; last: v == b[-1]
  return np.bincount(np.minimum(idx, len(b) - 1), minlength=len(b)).astype(np.float32)


def reduce(source_names: Sequence[str], values: np.ndarray, params: Sequence[Tuple[str, Sequence[float]]],
           normalize_reduce: bool) -> Dict[str, np.float32]:
  """values: the (normalised) fp32 measurements in source order -> {field id: float32}."""
  out: Dict[str, np.float32] = {}
  bins = {n: list(b) for n, b in params}
  v32 = np.asarray(values, dtype=np.float32)
  for name, v in zip(source_names, v32):
    dev, meas = name.split("/", 1)
    if meas not in bins:
      out[f"{dev}_{meas}"] = np.float32(v)
  for meas, b in bins.items():
    cols = [i for i, n in enumerate(source_names) if n.split("/", 1)[1] == meas]
    if not cols:
      continue
    h = clipped_histogram(v32[cols].astype(np.float64), b)
    if normalize_reduce:
      h = (h.astype(np.float64) / float(np.sum(h))).astype(np.float32)
    for val, c in zip(b, h):
      out[f"{meas}_h_{val:.2f}"] = np.float32(c)
  return out
