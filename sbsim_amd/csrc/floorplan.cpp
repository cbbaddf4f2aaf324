// floorplan.cpp -- floor-plan preprocessing on the host (no device code): what the reference does
// with OpenCV + NumPy before a FloorPlanBasedBuilding exists.
//
//   guarantee_air_padding_in_frame            simulator/building_utils.py:137-208
//   exterior space / exterior-wall shell      simulator/building_utils.py:222-330
//   enlarge_exterior_walls                    simulator/building.py:183-229 (cv2.distanceTransform
//                                             DIST_L2 / mask 3, "<= 2": chamfer weights a = 0.955,
//                                             b = 1.3693 select the 13 offsets below)
//   rooms = connected components              simulator/building_utils.py:406-414
//                                             (cv2.connectedComponents, 4-connectivity; labels in
//                                             raster order of each component's first cell)
#include "sbsim_amd.h"

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace {

constexpr int kInterior = 0, kWall = 1, kExterior = 2; // simulator/constants.py:24-36

struct Grid {
  int H, W;
  std::vector<int8_t> v;
  int8_t at(int x, int y) const { return v[(size_t)x * W + y]; }
};

// A wall on the frame gets one line of exterior space in front of it, edge by edge in the
// reference's order (top, left, bottom, right): later edges see the lines added before.
Grid pad_frame(const int8_t *src, int H, int W) {
  Grid g{H, W, std::vector<int8_t>(src, src + (size_t)H * W)};
  auto row_has_wall = [&](int x) { for (int y = 0; y < g.W; ++y) if (g.at(x, y) == kWall) return true; return false; };
  auto col_has_wall = [&](int y) { for (int x = 0; x < g.H; ++x) if (g.at(x, y) == kWall) return true; return false; };
  auto add_row = [&](bool top) {
    std::vector<int8_t> n((size_t)(g.H + 1) * g.W, (int8_t)kExterior);
    std::memcpy(n.data() + (top ? g.W : 0), g.v.data(), (size_t)g.H * g.W);
    g.v.swap(n); ++g.H;
  };
  auto add_col = [&](bool left) {
    std::vector<int8_t> n((size_t)g.H * (g.W + 1), (int8_t)kExterior);
    for (int x = 0; x < g.H; ++x) std::memcpy(n.data() + (size_t)x * (g.W + 1) + (left ? 1 : 0), g.v.data() + (size_t)x * g.W, (size_t)g.W);
    g.v.swap(n); ++g.W;
  };
  if (row_has_wall(0)) add_row(true);
  if (col_has_wall(0)) add_col(true);
  if (row_has_wall(g.H - 1)) add_row(false);
  if (col_has_wall(g.W - 1)) add_col(false);
  return g;
}

int find(std::vector<int> &parent, int i) {
  while (parent[i] != i) { parent[i] = parent[parent[i]]; i = parent[i]; }
  return i;
}

} // namespace

extern "C" int sb_floorplan_padded_shape(const int8_t *floor_plan, int32_t H, int32_t W, int32_t *H_out, int32_t *W_out) {
  if (!floor_plan || !H_out || !W_out || H < 2 || W < 2) return SB_ERR_INVALID;
  const Grid g = pad_frame(floor_plan, H, W);
  *H_out = g.H; *W_out = g.W;
  return SB_OK;
}

extern "C" int sb_floorplan_preprocess(const int8_t *floor_plan, const int8_t *zone_map, int32_t H, int32_t W,
                                       uint8_t *exterior_space, uint8_t *wall_kind, uint8_t *interior_wall,
                                       int16_t *zone_label, int32_t *n_rooms) {
  if (!floor_plan || !exterior_space || !wall_kind || !interior_wall || !zone_label || !n_rooms || H < 2 || W < 2)
    return SB_ERR_INVALID;
  const Grid fp = pad_frame(floor_plan, H, W);
  const Grid zm = zone_map ? pad_frame(zone_map, H, W) : fp;
  if (zm.H != fp.H || zm.W != fp.W) return SB_ERR_INVALID; // the two maps must pad alike
  const int Hp = fp.H, Wp = fp.W;
  const size_t N = (size_t)Hp * Wp;
  auto in = [&](int x, int y) { return x >= 0 && x < Hp && y >= 0 && y < Wp; };
  std::vector<uint8_t> shell(N, 0), near(N, 0);
  for (int x = 0; x < Hp; ++x)
    for (int y = 0; y < Wp; ++y) {
      const size_t i = (size_t)x * Wp + y;
      exterior_space[i] = fp.at(x, y) == kExterior;
      if (exterior_space[i]) continue;
      static const int d4[4][2] = {{-1, 0}, {1, 0}, {0, -1}, {0, 1}};
      for (auto &d : d4) // exterior-wall shell: touches exterior space by an edge
        if (in(x + d[0], y + d[1]) && fp.at(x + d[0], y + d[1]) == kExterior) shell[i] = 1;
    }
  static const int d13[13][2] = {{0, 0}, {-1, 0}, {1, 0}, {0, -1}, {0, 1}, {-2, 0}, {2, 0}, {0, -2}, {0, 2},
                                 {-1, -1}, {-1, 1}, {1, -1}, {1, 1}};
  for (int x = 0; x < Hp; ++x)
    for (int y = 0; y < Wp; ++y)
      if (shell[(size_t)x * Wp + y])
        for (auto &d : d13)
          if (in(x + d[0], y + d[1])) near[(size_t)(x + d[0]) * Wp + (y + d[1])] = 1;
  for (size_t i = 0; i < N; ++i) {
    const bool iw = fp.v[i] == kWall && !shell[i];       // _label_interior_walls
    const bool ew = (int)near[i] + (int)iw + (int)shell[i] >= 2; // building.py:183-229
    interior_wall[i] = iw;                               // un-shrunk: the diffuser placement tests this one
    wall_kind[i] = ew ? 2 : (iw ? 1 : 0);
  }
  // rooms: 4-connected components of the zone map's interior space, two-pass union-find
  std::vector<int> prov(N, -1), parent;
  for (int x = 0; x < Hp; ++x)
    for (int y = 0; y < Wp; ++y) {
      const size_t i = (size_t)x * Wp + y;
      if (zm.v[i] != kInterior) continue;
      const int up = x > 0 ? prov[i - Wp] : -1, left = y > 0 ? prov[i - 1] : -1;
      if (up < 0 && left < 0) { prov[i] = (int)parent.size(); parent.push_back(prov[i]); }
      else if (up >= 0 && left >= 0) {
        const int a = find(parent, up), b = find(parent, left);
        prov[i] = a < b ? a : b;
        parent[a < b ? b : a] = prov[i];
      } else prov[i] = find(parent, up >= 0 ? up : left);
    }
  std::vector<int> final_label(parent.size(), -1);
  int rooms = 0;
  for (size_t i = 0; i < N; ++i) {
    zone_label[i] = -1;
    if (prov[i] < 0) continue;
    const int root = find(parent, prov[i]);
    if (final_label[root] < 0) final_label[root] = rooms++; // raster order of the first cell
    if (rooms > 32767) return SB_ERR_UNSUPPORTED;
    zone_label[i] = (int16_t)final_label[root];
  }
  *n_rooms = rooms;
  return SB_OK;
}

// ---------------------------------------------------------------- thermal diffusers
// simulator/thermal_diffuser_utils.py:34-262 + building.py:301-345 (_assign_thermal_diffusers): per room, diffusers
// on an evenly spaced grid of its bounding box when the room is "rectangular enough", else a seeded random choice
// of its control volumes; those that sit in an interior wall are dropped; each of the n that remain gets 1 / n.
namespace {

// numpy.random.default_rng(seed).choice(N, n, replace=False) for a 32-bit seed, bit for bit (NumPy >= 1.17:
// SeedSequence -> PCG64 (XSL-RR 128/64) -> Lemire's bounded 32-bit integers from buffered halves -> Floyd's
// sampling + a final shuffle, or a tail shuffle of arange(N) for large dense draws).  The reference's
// irregular-room branch is exactly that call with seed 23 (thermal_diffuser_utils.py:105-131).
class NumpyDefaultRng {
 public:
  explicit NumpyDefaultRng(uint32_t seed) {
    // SeedSequence(seed): a pool of four 32-bit words mixed from the entropy [seed]
    uint32_t pool[4], hc = 0x43b0d7e5u;
    auto hashmix = [&](uint32_t v) { v ^= hc; hc *= 0x931e8875u; v *= hc; v ^= v >> 16; return v; };
    auto mix = [](uint32_t x, uint32_t y) { uint32_t r = 0xca01f9ddu * x - 0x4973f715u * y; r ^= r >> 16; return r; };
    for (int i = 0; i < 4; ++i) pool[i] = hashmix(i == 0 ? seed : 0u);
    for (int s = 0; s < 4; ++s)
      for (int d = 0; d < 4; ++d)
        if (s != d) pool[d] = mix(pool[d], hashmix(pool[s]));
    // generate_state(4, uint64): eight words, little-endian pairs
    uint32_t w[8], hb = 0x8b51f9ddu;
    for (int i = 0; i < 8; ++i) {
      uint32_t v = pool[i % 4];
      v ^= hb; hb *= 0x58f38dedu; v *= hb; v ^= v >> 16;
      w[i] = v;
    }
    auto u64 = [&](int k) { return (uint64_t)w[2 * k] | ((uint64_t)w[2 * k + 1] << 32); };
    const u128 initstate = ((u128)u64(0) << 64) | u64(1), initseq = ((u128)u64(2) << 64) | u64(3);
    state_ = 0; inc_ = (initseq << 1) | 1u;
    step(); state_ += initstate; step();
  }
  uint64_t bounded(uint64_t rng) { // random_bounded_uint64(off = 0, rng, mask = 0, use_masked = false): uniform on [0, rng]
    if (rng == 0) return 0;
    if (rng <= 0xffffffffull) {
      if (rng == 0xffffffffull) return next32();
      const uint32_t excl = (uint32_t)rng + 1u;
      uint64_t m = (uint64_t)next32() * excl;
      uint32_t left = (uint32_t)m;
      if (left < excl) {
        const uint32_t threshold = (0xffffffffu - (uint32_t)rng) % excl;
        while (left < threshold) { m = (uint64_t)next32() * excl; left = (uint32_t)m; }
      }
      return m >> 32;
    }
    if (rng == ~0ull) return next64();
    const uint64_t excl = rng + 1;
    u128 m = (u128)next64() * excl;
    uint64_t left = (uint64_t)m;
    if (left < excl) {
      const uint64_t threshold = (~0ull - rng) % excl;
      while (left < threshold) { m = (u128)next64() * excl; left = (uint64_t)m; }
    }
    return (uint64_t)(m >> 64);
  }
  void shuffle(std::vector<int64_t> &v, int64_t n, int64_t first) { // _shuffle_int
    for (int64_t i = n - 1; i >= first; --i) std::swap(v[(size_t)i], v[(size_t)bounded((uint64_t)i)]);
  }
  std::vector<int64_t> choice_without_replacement(int64_t pop, int64_t size) {
    std::vector<int64_t> idx;
    if (pop > 10000 && size > pop / 50) { // tail shuffle of arange(pop)
      std::vector<int64_t> all((size_t)pop);
      for (int64_t i = 0; i < pop; ++i) all[(size_t)i] = i;
      shuffle(all, pop, std::max<int64_t>(pop - size, 1));
      idx.assign(all.begin() + (pop - size), all.end());
      return idx;
    }
    idx.resize((size_t)size);
    uint64_t mask = (uint64_t)(1.2 * (double)size); // Floyd's algorithm over a hash set of the next power of two
    mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16; mask |= mask >> 32;
    std::vector<uint64_t> set((size_t)mask + 1, ~0ull);
    for (int64_t j = pop - size; j < pop; ++j) {
      const uint64_t val = bounded((uint64_t)j);
      uint64_t loc = val & mask;
      while (set[loc] != ~0ull && set[loc] != val) loc = (loc + 1) & mask;
      if (set[loc] == ~0ull) {
        set[loc] = val;
        idx[(size_t)(j - pop + size)] = (int64_t)val;
      } else { // val was drawn before: j itself goes in
        loc = (uint64_t)j & mask;
        while (set[loc] != ~0ull) loc = (loc + 1) & mask;
        set[loc] = (uint64_t)j;
        idx[(size_t)(j - pop + size)] = j;
      }
    }
    shuffle(idx, size, 1); // choice(..., shuffle=True), the default
    return idx;
  }

 private:
  typedef unsigned __int128 u128;
  u128 state_, inc_;
  bool has32_ = false;
  uint32_t half_ = 0;
  void step() { state_ = state_ * ((((u128)2549297995355413924ull) << 64) | 4865540595714422341ull) + inc_; }
  uint64_t next64() {
    step();
    const uint64_t hi = (uint64_t)(state_ >> 64), lo = (uint64_t)state_, x = hi ^ lo;
    const unsigned rot = (unsigned)(hi >> 58);
    return (x >> rot) | (x << ((64 - rot) & 63));
  }
  uint32_t next32() {
    if (has32_) { has32_ = false; return half_; }
    const uint64_t v = next64();
    has32_ = true; half_ = (uint32_t)(v >> 32);
    return (uint32_t)v;
  }
};

// thermal_diffuser_utils.py:34-64: np.arange(start, end, len / (n + 1))[1:], rounded up
std::vector<int> evenly_spaced(int start, int end, int spacing) {
  const int len = end - start;
  if (len == 0) return {start};
  const double n = std::max(1.0, std::nearbyint((double)len / (double)spacing)); // np.round: half to even
  const double step = (double)len / (n + 1.0);
  const double first = (double)start, delta = (first + step) - first;              // np.arange fills first + i * delta
  const long count = (long)std::ceil(((double)end - first) / step);
  std::vector<int> out;
  for (long i = 1; i < count; ++i) out.push_back((int)std::ceil(first + (double)i * delta));
  return out;
}

} // namespace

int sb_floorplan_diffusers(const int16_t *zone_label, const uint8_t *interior_wall, int32_t H, int32_t W, int32_t n_rooms,
                           int32_t spacing, int32_t buffer_from_walls, double *diffusers) {
  if (!zone_label || !interior_wall || !diffusers || H < 1 || W < 1 || n_rooms < 0 || spacing < 1 || buffer_from_walls < 0)
    return SB_ERR_INVALID;
  const size_t N = (size_t)H * W;
  for (size_t i = 0; i < N; ++i) diffusers[i] = 0.0;
  std::vector<std::vector<int>> cells((size_t)n_rooms); // flat indices in raster order
  for (size_t i = 0; i < N; ++i)
    if (zone_label[i] >= 0 && zone_label[i] < n_rooms) cells[(size_t)zone_label[i]].push_back((int)i);
  for (const auto &room : cells) {
    if (room.empty()) continue;
    int sx = H, ex = -1, sy = W, ey = -1;
    for (int i : room) {
      sx = std::min(sx, i / W); ex = std::max(ex, i / W);
      sy = std::min(sy, i % W); ey = std::max(ey, i % W);
    }
    std::vector<int> picked;
    const double rect = (double)std::max(ex - sx, 1) * (double)std::max(ey - sy, 1);
    if ((double)room.size() / rect > 0.1) { // _rectangularity_test(threshold = 0.1)
      if (ex - sx > 2 * buffer_from_walls) { sx += buffer_from_walls; ex -= buffer_from_walls; } // only x is buffered (:160-164)
      const std::vector<int> px = evenly_spaced(sx, ex, spacing), py = evenly_spaced(sy, ey, spacing);
      auto in = [](const std::vector<int> &v, int k) { for (int e : v) if (e == k) return true; return false; };
      for (int i : room)
        if (in(px, i / W) && in(py, i % W)) picked.push_back(i);
    } else {
      const int64_t n = (int64_t)std::max(1.0, std::nearbyint((double)room.size() / ((double)spacing * (double)spacing)));
      NumpyDefaultRng rng(23u);
      for (int64_t k : rng.choice_without_replacement((int64_t)room.size(), std::min<int64_t>(n, (int64_t)room.size())))
        picked.push_back(room[(size_t)k]);
    }
    std::vector<int> kept;
    for (int i : picked)
      if (!interior_wall[i]) kept.push_back(i);
    for (int i : kept) diffusers[i] = 1.0 / (double)kept.size();
  }
  return SB_OK;
}

// Developer / test entry: numpy.random.default_rng(seed).choice(pop, size, replace=False) -- the indices.
int sb_debug_numpy_choice(uint32_t seed, int64_t pop, int64_t size, int64_t *out) {
  if (!out || pop < 1 || size < 1 || size > pop) return SB_ERR_INVALID;
  NumpyDefaultRng rng(seed);
  const std::vector<int64_t> idx = rng.choice_without_replacement(pop, size);
  for (size_t i = 0; i < idx.size(); ++i) out[i] = idx[i];
  return SB_OK;
}
