"""Displacement statistics of the reference's stochastic convection.

TEST INFRASTRUCTURE -- runs only in the build container: imports the REFERENCE
(`/root/reference`, through oracle/refshim) and runs its own
`StochasticConvectionSimulator.apply_convection` (stochastic_convection_simulator.py:62-145) on a
16 x 20 grid with one 14 x 18 room whose temperatures are the cells' own indices, many times.
Recorded per (p, distance): the probability that a value starting at least 4 cells away from the
walls ends at offset (dx, dy), dx, dy in -8..8; the fraction of ALL room values that do not move;
the mean squared displacement of all room values.  A second layout holds rooms of 9, 60 and 100 cells under
the wide windows (distance = -1 with p < 1; distance 30): per room, the fraction of values that stay and the
mean squared displacement.  Output: tests/golden/convection_stats.npz.

    python -m oracle.gen_golden_convection
"""
from __future__ import annotations

import os

import numpy as np

from oracle import refshim

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
H, W = 16, 20
# (1.0, -1): the whole-room shuffle (:78-99); (0.5, -1): distance -1 with p < 1 falls into _shuffle_max_dist
# with a 1000-cell window (:108-109: every cell within sqrt(1000) = 31.6 cells); (1.0, 20): a window of 69
# offsets (squared distance <= 20).  New cases go at the END: a case's seed is 1234 + its index.
CASES = [(1.0, 5), (0.5, 5), (1.0, 2), (1.0, -1), (0.5, -1), (1.0, 20)]
TRIALS = 400
# a second layout: rooms of 9, 60 and 100 cells (x0, y0, height, width) in a 14 x 26 grid, wide windows only
SMALL_H, SMALL_W = 14, 26
SMALL_ROOMS = [(1, 1, 3, 3), (5, 1, 6, 10), (1, 13, 10, 10)]
SMALL_CASES = [(0.5, -1), (1.0, 30)]
SMALL_TRIALS = 800


def main() -> None:
  refshim.install()
  scs = refshim.ref("simulator.stochastic_convection_simulator")
  room = [(x, y) for x in range(1, H - 1) for y in range(1, W - 1)]
  wall = [(x, y) for x in range(H) for y in range(W) if (x, y) not in set(room)]
  room_dict = {"exterior_space": [], "interior_wall": wall, "room_1": room}
  out = {"H": np.array(H), "W": np.array(W), "trials": np.array(TRIALS),
         "cases": np.array(CASES, dtype=np.float64)}
  ids = np.arange(H * W, dtype=np.float64).reshape(H, W)
  for ci, (p, dist) in enumerate(CASES):
    sim = scs.StochasticConvectionSimulator(p=p, distance=dist, seed=1234 + ci)
    hist = np.zeros((17, 17))
    n_inner = 0
    fixed = 0
    msd = 0.0
    for _ in range(TRIALS):
      temp = ids.copy()
      sim.apply_convection(room_dict, temp)
      assert np.array_equal(np.sort(temp.reshape(-1)), ids.reshape(-1))
      for x in range(1, H - 1):
        for y in range(1, W - 1):
          src = int(temp[x, y])
          sx, sy = divmod(src, W)
          dx, dy = x - sx, y - sy
          fixed += int(dx == 0 and dy == 0)
          msd += dx * dx + dy * dy
          if 5 <= sx <= H - 6 and 5 <= sy <= W - 6:
            if abs(dx) <= 8 and abs(dy) <= 8:   # the whole-room shuffle moves values further: that mass stays outside
              hist[dx + 8, dy + 8] += 1
            n_inner += 1
    out[f"hist_{ci}"] = hist / n_inner
    out[f"fixed_{ci}"] = np.array(fixed / (TRIALS * len(room)))
    out[f"msd_{ci}"] = np.array(msd / (TRIALS * len(room)))
    print(p, dist, "fixed", out[f"fixed_{ci}"], "msd", out[f"msd_{ci}"], "inner samples", n_inner)
  # Rooms SMALLER than a wide window: the reference always draws from the room's candidate list
  # (:108-131), so a 9-cell room mixes as thoroughly as a 100-cell one.  Per room: the fraction of values
  # that do not move and the mean squared displacement.
  room_dict = {"exterior_space": [], "interior_wall": []}
  for ri, (x0, y0, h, w) in enumerate(SMALL_ROOMS):
    room_dict[f"room_{ri + 1}"] = [(x, y) for x in range(x0, x0 + h) for y in range(y0, y0 + w)]
  ids = np.arange(SMALL_H * SMALL_W, dtype=np.float64).reshape(SMALL_H, SMALL_W)
  out["small_shape"] = np.array([SMALL_H, SMALL_W])
  out["small_rooms"] = np.array(SMALL_ROOMS)
  out["small_cases"] = np.array(SMALL_CASES, dtype=np.float64)
  out["small_trials"] = np.array(SMALL_TRIALS)
  for ci, (p, dist) in enumerate(SMALL_CASES):
    sim = scs.StochasticConvectionSimulator(p=p, distance=dist, seed=4321 + ci)
    fixed = np.zeros(len(SMALL_ROOMS))
    msd = np.zeros(len(SMALL_ROOMS))
    for _ in range(SMALL_TRIALS):
      temp = ids.copy()
      sim.apply_convection(room_dict, temp)
      for ri, (x0, y0, h, w) in enumerate(SMALL_ROOMS):
        blk = temp[x0:x0 + h, y0:y0 + w].astype(np.int64)
        sx, sy = np.divmod(blk, SMALL_W)
        x, y = np.meshgrid(np.arange(x0, x0 + h), np.arange(y0, y0 + w), indexing="ij")
        assert ((sx >= x0) & (sx < x0 + h) & (sy >= y0) & (sy < y0 + w)).all()   # values stay in their room
        fixed[ri] += ((sx == x) & (sy == y)).sum()
        msd[ri] += ((sx - x) ** 2 + (sy - y) ** 2).sum()
    n = np.array([h * w for _, _, h, w in SMALL_ROOMS], dtype=np.float64) * SMALL_TRIALS
    out[f"small_fixed_{ci}"] = fixed / n
    out[f"small_msd_{ci}"] = msd / n
    print("small rooms", p, dist, "fixed", out[f"small_fixed_{ci}"], "msd", out[f"small_msd_{ci}"])
  np.savez_compressed(os.path.join(GOLD, "convection_stats.npz"), **out)


if __name__ == "__main__":
  main()
