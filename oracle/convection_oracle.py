"""CPU restatement of the device convection shuffle (TEST INFRASTRUCTURE: only tests/ may import
this).

Checker for `k_convect` (sbsim_amd/csrc/generators.hip).  The random PROCESS is
`StochasticConvectionSimulator._shuffle_max_dist` of the reference
(simulator/stochastic_convection_simulator.py:101-145): every cell of a room starts a swap with
probability p, its partner is uniform over the room's cells inside the offset window
(:125-131: dx, dy in [-distance, distance), dx^2 + dy^2 <= distance), the swaps run one after
the other in uniformly random order (:137-145).  Here the swaps are really applied one after
the other (the device follows every value through the sequence instead); the draws are the
device's: a counter-based mixer -- MurmurHash3's 32-bit finaliser -- over (stream, cell, word number):
stream = (s0, s1) = conv_stream(seed, global building, call number), word k of the cell with grid index g0 =
fmix32((s0 ^ g0 * 0x9E3779B1) + s1 + k * 0x6C8E9CF5); word 0 -> inclusion (u = (x >> 8) / 2**24, included
unless u > p; not formed when p >= 1), word 1 -> order (swaps by increasing ((x >> 12) << 11 | the cell's rank in the
room)) AND the partner (((x & 0xfff) * count) >> 12 among the valid offsets in (dx, dy) raster order: one mixer round
for both since round 5); wide windows: words 2, 3, .. -> candidates until one is accepted.  The reference draws from Python's
global `random`, which no counter-based generator reproduces: what is pinned against the reference is the
displacement statistics (tests/golden/convection_stats.npz)."""
from __future__ import annotations

import numpy as np

from oracle.occupancy_oracle import philox4x32_10

M32 = 0xFFFFFFFF


def fmix32(h: int) -> int:
  """MurmurHash3's 32-bit finaliser."""
  h &= M32
  h ^= h >> 16; h = (h * 0x85EBCA6B) & M32
  h ^= h >> 13; h = (h * 0xC2B2AE35) & M32
  h ^= h >> 16
  return h


def conv_salt(seed: int, call: int):
  """What the two stream words take from (seed, call number): the same for every building of a call."""
  h = fmix32((seed & M32) ^ 0x9E3779B9)
  h = fmix32(h ^ ((seed >> 32) & M32))
  h = fmix32(h ^ (call & M32))
  g = fmix32((call & M32) ^ 0x7F4A7C15)
  g = fmix32((g + ((seed >> 32) & M32)) & M32)
  g = fmix32((g + (seed & M32)) & M32)
  return (h, g)


def conv_stream(seed: int, gb: int, call: int):
  """(seed, global building, call) -> the building's stream (s0, s1), 64 bits -- with one word 65,536 buildings had 0.5
  colliding pairs per call.  One round per word over the building's number modulo 2**32 and a (seed, call) salt: each
  word is a bijection of the building's number (generators.hip)."""
  a, b = conv_salt(seed, call)
  gb &= M32
  return (fmix32(gb ^ a), fmix32((gb * 0x9E3779B1 + b) & M32))


def conv_word(stream, g0: int, k: int) -> int:
  key = ((stream[0] ^ ((g0 * 0x9E3779B1) & M32)) + stream[1]) & M32
  return fmix32((key + k * 0x6C8E9CF5) & M32)


def offsets(distance: int):
  return [(dx, dy) for dx in range(-distance, distance) for dy in range(-distance, distance)
          if dx * dx + dy * dy <= distance]


CONV_MAX_TRIES = 1 << 14   # partner candidates of a wide-window draw (words 2 .. CONV_MAX_TRIES - 1)


def wide_partner(stream: int, g0: int, distance: int, accept, ranked=None, W: int = 0):
  """The device's partner choice for windows of more than 64 offsets (distance >= 20; distance = -1
  with p < 1 is the reference's 1000: stochastic_convection_simulator.py:108-109): rejection sampling,
  uniform over the reference's candidate list (:122-131; the window [-d, d) does not cut the disc:
  floor(sqrt(d)) < d).  A room with fewer cells than the window's box (2R + 1)^2, R = floor(sqrt(distance)):
  a cell of the room, uniform by its rank in raster order (`ranked`: the room's grid indices, sorted), kept
  when dx^2 + dy^2 <= distance -> its grid index.  Otherwise: (dx, dy) uniform in [-R, R]^2, kept when inside
  the disc and in the room (`accept(dx, dy)` -> the partner's index in the room, or -1).  The cell itself is a
  candidate, so a draw is accepted sooner or later (after CONV_MAX_TRIES words the cell stays: None)."""
  R = int(np.floor(np.sqrt(distance)))
  by_rank = ranked is not None and len(ranked) < (2 * R + 1) ** 2
  x0, y0 = divmod(g0, W) if by_rank else (0, 0)
  for k in range(2, CONV_MAX_TRIES):
    x = conv_word(stream, g0, k)
    if by_rank:
      g = int(ranked[(x * len(ranked)) >> 32])
      xx, yy = divmod(g, W)
      if (xx - x0) ** 2 + (yy - y0) ** 2 <= distance:
        return ("cell", g)
      continue
    dx = (((x & 0xFFFF) * (2 * R + 1)) >> 16) - R
    dy = (((x >> 16) * (2 * R + 1)) >> 16) - R
    if dx * dx + dy * dy > distance:
      continue
    j = accept(dx, dy)
    if j >= 0:
      return ("index", j)
  return None


def whole_room_permutation(n: int, gb: int, call: int, z: int, seed: int) -> np.ndarray:
  """dest[r] of the device's whole-room shuffle (k_convect_all, generators.hip): a keyed bijection on
  the ranks 0..n-1 of a room's cells in raster order -- three rounds of (odd multiply, add,
  xor-shift) on k = ceil(log2 n) bits, cycle-walked into [0, n).  Keys: two Philox blocks with
  counters (building lo, hi, call, 0x80000000 | 2 z [+ 1])."""
  M = 0xFFFFFFFF
  one = lambda v: np.array([v], dtype=np.uint64)
  c = philox4x32_10(one(gb & M), one(gb >> 32), one(call), one(0x80000000 | (2 * z)), seed & M, (seed >> 32) & M)
  d = philox4x32_10(one(gb & M), one(gb >> 32), one(call), one(0x80000000 | (2 * z + 1)), seed & M, (seed >> 32) & M)
  k = max(1, int(n - 1).bit_length())
  mask = (1 << k) - 1
  s0, s1 = max(1, k // 2), max(1, (k + 1) // 3)
  m = [int(c[0][0]) | 1, int(c[1][0]) | 1, int(c[2][0]) | 1]
  a = [int(c[3][0]), int(d[0][0]), int(d[1][0])]
  dest = np.zeros(n, dtype=np.int64)
  for r in range(n):
    x = r
    while True:
      x = ((x * m[0] + a[0]) & M) & mask; x ^= x >> s0
      x = ((x * m[1] + a[1]) & M) & mask; x ^= x >> s1
      x = ((x * m[2] + a[2]) & M) & mask; x ^= x >> s0
      if x < n:
        break
    dest[r] = x
  return dest


class ConvectionOracle:

  def __init__(self, zone_cell_lists, H: int, W: int, p: float, distance: int, seed: int, first_building: int = 0):
    self.zones = [np.asarray(c, dtype=np.int64) for c in zone_cell_lists]
    self.H, self.W, self.p, self.seed, self.first = H, W, float(p), int(seed), int(first_building)
    self.whole_room = distance == -1 and float(p) == 1.0   # stochastic_convection_simulator.py:78-99
    if distance == -1 and not self.whole_room:
      distance = 1000                                       # :108-109
    self.distance = distance
    self.off = [] if self.whole_room else (offsets(distance) if distance <= 19 else None)   # None: more than 64 offsets, the wide window
    self.room = np.full(H * W, -1, dtype=np.int64)
    self.local = np.full(H * W, -1, dtype=np.int64)
    for z, cells in enumerate(self.zones):
      self.room[cells] = z
      self.local[cells] = np.arange(len(cells))
    self.ranked = [np.sort(c) for c in self.zones]          # a room's cells by rank = raster order of the caller's grid
    self.call = 0

  def swaps(self, b: int, z: int, call: int):
    """The room's swap sequence [(cell index, partner index)] in application order."""
    cells = self.zones[z]
    gb = self.first + b
    n = len(cells)
    stream = conv_stream(self.seed, gb, call)
    rank = np.empty(n, dtype=np.int64)
    rank[np.argsort(cells, kind="stable")] = np.arange(n)     # the cell's rank in raster order
    seq = []
    for i in range(n):
      g0 = int(cells[i])
      if self.p < 1.0:       # (u < 1: with p >= 1 every cell is included and the device does not form word 0)
        u = (conv_word(stream, g0, 0) >> 8) / 16777216.0
        if u > self.p:
          continue
      x, y = divmod(g0, self.W)

      def accept(dx, dy, x=x, y=y):
        xx, yy = x + dx, y + dy
        if 0 <= xx < self.H and 0 <= yy < self.W and self.room[xx * self.W + yy] == z:
          return int(self.local[xx * self.W + yy])
        return -1
      if self.off is None:   # more than 64 offsets: rejection sampling (the device's wide path)
        hit = wide_partner(stream, g0, self.distance, accept, ranked=self.ranked[z], W=self.W)
        other = i if hit is None else (int(self.local[hit[1]]) if hit[0] == "cell" else hit[1])
      else:                  # uniform over the valid offsets, in (dx, dy) raster order ((0, 0) always is one)
        cand = [j for j in (accept(dx, dy) for dx, dy in self.off) if j >= 0]
        other = cand[((conv_word(stream, g0, 1) & 0xfff) * len(cand)) >> 12]   # word 1's low 12 bits (its top 20: the time stamp)
      if other != i:
        seq.append(((((conv_word(stream, g0, 1) >> 12) << 11) | int(rank[i])) + 1, i, other))
    seq.sort()
    return [(i, o) for _, i, o in seq]

  def apply(self, grids: np.ndarray) -> None:
    """grids [B, H, W] float64, shuffled in place (one call of apply_convection per building)."""
    B = grids.shape[0]
    flat = grids.reshape(B, -1)
    for b in range(B):
      for z, cells in enumerate(self.zones):
        v = flat[b, cells].copy()
        if self.whole_room:
          order = np.argsort(cells, kind="stable")          # rank = position in raster order
          dest = whole_room_permutation(len(cells), self.first + b, self.call, z, self.seed)
          out = v.copy()
          out[order[dest]] = v[order]                       # the value of rank r goes to rank dest[r]
          flat[b, cells] = out
          continue
        for i, o in self.swaps(b, z, self.call):
          v[i], v[o] = v[o], v[i]
        flat[b, cells] = v
    self.call += 1
