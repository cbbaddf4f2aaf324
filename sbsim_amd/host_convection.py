"""The reference's stochastic convection on the HOST, draw for draw -- the opt-in bit-reproducible mode of
the single-building adapter (``HipSimulatorBuilding(convection_simulator=..., reproducible_convection=True)``).

The reference seeds Python's Mersenne Twister when a seed is given (stochastic_convection_simulator.py:59-60)
and then shuffles the temperature array in place after every finite-difference update
(simulator_flexible_floor_plan.py:156, building.py:891-893).  A batched device kernel cannot replay ONE global
random stream (the batch path, ``sb_convection_attach``, keeps the process and draws per building from a
counter-based generator: statistically equal, tests/test_convection.py), but a single building can: this module
consumes a ``random.Random(seed)`` -- the same generator ``random.seed(seed)`` seeds -- in the reference's order:
per room in room-dict order and per control volume in raster order one uniform draw against ``p``; for the cells
that move, one ``choice`` among the room's cells inside the window (candidates in raster order of the window,
squared distance <= the distance parameter itself, :126-133; distance -1 with p < 1: a window of 1000, :108-109);
then one ``shuffle`` of the room's swap list and the swaps in that order (:136-145).  p = 1 with distance -1 is the
whole-room permutation (:78-99).  Given the same seed and the same sequence of calls the temperatures are
bit-identical to the reference's (tests/golden/convection_seeded.npz holds the reference's own permutations).
"""
from __future__ import annotations

import random
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

Cell = Tuple[int, int]


class SeededHostConvection:
  """apply(rooms, temp): one convection pass over ``temp`` ([H, W], in place).  ``rooms``: the rooms in the
  reference's room-dict order (room_1, room_2, ...: building_utils.py:406-414), each a list of (x, y) in raster
  order -- ``FloorPlan.zone_cell_lists()``."""

  def __init__(self, p: float, distance: int, seed: Optional[int]):
    self.p, self.distance = float(p), int(distance)
    self._rng = random.Random(seed) if seed is not None else random.Random()
    self._windows: Dict[int, List[List[int]]] = {}   # per room: for every cell the indices of its candidates

  def _candidates(self, room_index: int, cells: Sequence[Cell], reach: int) -> List[List[int]]:
    tab = self._windows.get(room_index)
    if tab is None:
      index_of = {c: i for i, c in enumerate(cells)}
      tab = []
      for (x, y) in cells:
        found = []
        for dx in range(-reach, reach):          # the window is half-open at its far side (:123-124)
          for dy in range(-reach, reach):
            j = index_of.get((x + dx, y + dy))
            if j is not None and dx * dx + dy * dy <= reach:
              found.append(j)
        tab.append(found)
      self._windows[room_index] = tab
    return tab

  def apply(self, rooms: Sequence[Sequence[Cell]], temp: np.ndarray) -> None:
    if self.p == 0 or self.distance == 0:
      return
    rng = self._rng
    for r, cells in enumerate(rooms):
      cells = [(int(x), int(y)) for x, y in cells]
      xs = np.fromiter((c[0] for c in cells), dtype=np.intp, count=len(cells))
      ys = np.fromiter((c[1] for c in cells), dtype=np.intp, count=len(cells))
      if self.distance == -1 and self.p == 1:
        order = list(range(len(cells)))
        rng.shuffle(order)                        # the value of cell i goes to cell order[i]
        vals = temp[xs, ys].copy()
        temp[xs[order], ys[order]] = vals
        continue
      reach = 1000 if self.distance == -1 else self.distance
      window = self._candidates(r, cells, reach)
      swaps = []
      for i in range(len(cells)):
        if rng.uniform(0, 1) > self.p:
          continue
        swaps.append((i, rng.choice(window[i])))
      rng.shuffle(swaps)
      for i, j in swaps:
        a, b = cells[i], cells[j]
        temp[a], temp[b] = temp[b], temp[a]
