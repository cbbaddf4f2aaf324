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
