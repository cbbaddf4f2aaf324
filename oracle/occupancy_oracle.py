"""CPU restatement of the device occupancy generator (TEST INFRASTRUCTURE: only tests/ may
import this).

Checker for `sb_occupancy_peek` (sbsim_amd/csrc/sbsim_hip.hip: k_occupancy): the per-occupant
state machine is `ZoneOccupant.peek` of the reference
(simulator/randomized_arrival_departure_occupancy.py:138-160, event probability :100-112); the
Bernoulli trials come from Philox4x32-10 (Salmon et al., SC'11) with key = seed and counter =
(global building index lo, hi, zone, query * 8 + occupant // 4), occupant i taking word i % 4,
u = (x >> 8) / 2**24.  The reference's own stream (np.random.RandomState) cannot be reproduced
by a counter-based generator; what IS pinned against the reference are the statistics
(tests/golden/occupancy_randomized.npz: mean occupancy fraction per step over 400 seeds)."""
from __future__ import annotations

import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0: int, k1: int):
  """Counter words as uint64 arrays holding 32-bit values; returns the four output words."""
  c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint64) & MASK for c in (c0, c1, c2, c3))
  for _ in range(10):
    p0, p1 = M0 * c0, M1 * c2
    n0 = (p1 >> np.uint64(32)) ^ c1 ^ np.uint64(k0)
    n2 = (p0 >> np.uint64(32)) ^ c3 ^ np.uint64(k1)
    c1, c3, c0, c2 = p1 & MASK, p0 & MASK, n0, n2
    k0, k1 = (k0 + W0) & 0xFFFFFFFF, (k1 + W1) & 0xFFFFFFFF
  return c0, c1, c2, c3


class DeviceOccupancyOracle:

  def __init__(self, n_buildings: int, n_zones: int, zone_assignment: int, earliest_arrival_hour: int,
               latest_arrival_hour: int, earliest_departure_hour: int, latest_departure_hour: int,
               time_step_sec: float, seed: int, first_building: int = 0):
    self.B, self.Z, self.n = n_buildings, n_zones, zone_assignment
    self.hours = (earliest_arrival_hour, latest_arrival_hour, earliest_departure_hour, latest_departure_hour)
    self.p_arr = 1.0 / ((latest_arrival_hour - earliest_arrival_hour) * 3600.0 / time_step_sec / 2.0)
    self.p_dep = 1.0 / ((latest_departure_hour - earliest_departure_hour) * 3600.0 / time_step_sec / 2.0)
    self.seed, self.first = int(seed), int(first_building)
    self.at_work = np.zeros((n_buildings, n_zones, zone_assignment), dtype=bool)
    self.query = 0

  def peek(self, local_hour: int, is_work_day: bool) -> np.ndarray:
    """One average_zone_occupancy per zone and building; returns counts [B, Z]."""
    q = self.query
    self.query += 1
    if not is_work_day:
      self.at_work[:] = False
      return np.zeros((self.B, self.Z))
    e_arr, l_arr, e_dep, _ = self.hours
    arr_open = not (local_hour < e_arr or local_hour > l_arr)
    dep_open = not (local_hour < e_dep)
    if arr_open or dep_open:
      gb = (self.first + np.arange(self.B, dtype=np.uint64))[:, None, None]
      z = np.arange(self.Z, dtype=np.uint64)[None, :, None]
      blk = (np.arange(self.n, dtype=np.uint64) // np.uint64(4))[None, None, :]
      ctr3 = (np.uint64(q * 8) + blk) & MASK
      shape = (self.B, self.Z, self.n)
      out = philox4x32_10(np.broadcast_to(gb & MASK, shape), np.broadcast_to(gb >> np.uint64(32), shape),
                          np.broadcast_to(z, shape), np.broadcast_to(ctr3, shape),
                          self.seed & 0xFFFFFFFF, (self.seed >> 32) & 0xFFFFFFFF)
      word = np.arange(self.n) % 4
      x = np.choose(np.broadcast_to(word[None, None, :], shape), out)
      u = (x >> np.uint64(8)).astype(np.float64) / 16777216.0
      arrive = ~self.at_work & arr_open & (u < self.p_arr)
      depart = self.at_work & dep_open & (u < self.p_dep)
      self.at_work = (self.at_work | arrive) & ~depart
    return self.at_work.sum(axis=2).astype(np.float64)
