"""The reference's SEEDED stochastic convection, call by call (VERDICT r3 missing #2).

TEST INFRASTRUCTURE -- runs only in the build container: imports the REFERENCE (`/root/reference`, through
oracle/refshim) and runs its own `StochasticConvectionSimulator(p, distance, seed).apply_convection`
(stochastic_convection_simulator.py:43-145) three times in a row on a 14 x 26 grid with rooms of 9, 60 and 100 cells
whose temperatures are the cells' own indices.  Recorded per case: the array after every call.  The product's host
path (sbsim_amd/host_convection.py) must reproduce them exactly (tests/test_convection.py).
Output: tests/golden/convection_seeded.npz.

    python -m oracle.gen_golden_convection_seeded
"""
from __future__ import annotations

import os

import numpy as np

from oracle import refshim

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
H, W = 14, 26
ROOMS = [(1, 1, 3, 3), (5, 1, 6, 10), (1, 13, 10, 10)]          # (x0, y0, height, width)
CASES = [(1.0, 5, 7), (0.5, 3, 11), (1.0, -1, 5), (0.4, -1, 9), (1.0, 30, 3)]   # (p, distance, seed)
CALLS = 3


def main() -> None:
  refshim.install()
  scs = refshim.ref("simulator.stochastic_convection_simulator")
  label = -np.ones((H, W), dtype=np.int64)
  for r, (x0, y0, h, w) in enumerate(ROOMS):
    label[x0:x0 + h, y0:y0 + w] = r
  # the room dict as building_utils.py:406-414 builds it: one raster pass, keys in order of first appearance
  room_dict = {}
  for x in range(H):
    for y in range(W):
      key = "interior_wall" if label[x, y] < 0 else f"room_{label[x, y] + 1}"
      room_dict.setdefault(key, []).append((x, y))
  ids = np.arange(H * W, dtype=np.float64).reshape(H, W)
  after = np.zeros((len(CASES), CALLS, H, W), dtype=np.int32)
  for ci, (p, dist, seed) in enumerate(CASES):
    sim = scs.StochasticConvectionSimulator(p=p, distance=dist, seed=seed)
    temp = ids.copy()
    for k in range(CALLS):
      sim.apply_convection(room_dict, temp)
      assert np.array_equal(np.sort(temp.reshape(-1)), ids.reshape(-1))
      after[ci, k] = temp.astype(np.int32)
  np.savez_compressed(os.path.join(ROOT, "tests", "golden", "convection_seeded.npz"), H=np.array(H), W=np.array(W),
                      rooms=np.array(ROOMS, dtype=np.int32), cases=np.array(CASES, dtype=np.float64), after=after)
  print("wrote convection_seeded.npz", after.shape)


if __name__ == "__main__":
  main()
