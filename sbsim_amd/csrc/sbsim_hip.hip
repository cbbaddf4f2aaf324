// sbsim_hip.hip -- C ABI (include/sbsim_amd.h) of the MI355X batched building-thermal step:
// handle creation (floor-plan tables, launch geometry, choice of step kernel), reset /
// observe / parity-tap kernels.  The step kernels live in step_reg.hip (temperature grid in
// registers; floor plans whose trimmed grid has <= 128 rows and <= 96 columns) and
// step_lds.hip (grid in LDS; any shape that fits a CU's LDS); both share sb_device.h.  The
// optional input generators (occupancy, convection) live in generators.hip; sb_host.h holds
// what the two host files share (the handle, device buffers, error reporting).
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off (fma only where written).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "sb_host.h"

// step_stream.hip's overlapped-sweeps kernel (Dev::stream_ms == 2; declared here: sb_device.h is part of the traffic profile's source hash)
namespace sb {
int launch_sweep_stream_roll(const Dev &d, double *abuf, double *ebuf, int waves, hipStream_t stream);
int prepare_sweep_stream_roll(const Dev &d, int waves);
int sweep_stream_roll_xchg_extra_doubles();
} // namespace sb

// The experimental sweep kernels (step_stream_ms.hip, step_stream.hip's k_sweep_stream_roll: exact, tested, slower than what
// they were meant to replace) are in the library only when it is built with SBSIM_BUILD_EXPERIMENTAL=1 (-DSB_EXPERIMENTAL,
// sbsim_amd/build.py); the default build answers for step_stream_ms.hip's entry points here and never plans them.
#ifdef SB_EXPERIMENTAL
constexpr bool kExperimental = true;
#else
constexpr bool kExperimental = false;
namespace sb {
int launch_sweep_stream_ms(const Dev &, double *, double *, int, hipStream_t) { return (int)hipErrorNotSupported; }
int prepare_sweep_stream_ms(const Dev &, int) { return (int)hipErrorNotSupported; }
int sweep_stream_ms_sweeps() { return 1; }
int sweep_stream_ms_seam_doubles(int, int) { return 0; }
int sweep_stream_ms_xchg_doubles(int) { return 0; }
} // namespace sb
#endif

using namespace sb;

namespace {

constexpr int kGuardHost = 16; // must match kGuard in step_lds.hip

// ---------------------------------------------------------------- kernels
// Simulator.reset(): grid (either state layout), zone means, device scalars.
// temps_only (sb_set_temps): the grid inside the building and the zone means follow `temps`; every device state stays.
__global__ void k_reset(Dev a, double initial_temp, const double *temps, int first, int steps_since_reset, int temps_only) {
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int nwaves = gridDim.x * (blockDim.x >> 6);
  for (int b = wave; b < a.B; b += nwaves) {
    double s = 0.0, rlo = INFINITY, rhi = -INFINITY;
    if (a.reg) {
      double *T = a.temp + (size_t)b * a.state_doubles;
      const double *S0 = a.scal + (size_t)b * kNScal;
      const bool ring_uniform = S0[16] == S0[17]; // (temps_only) the ring holds one value: the ambient temperature of the last step
      for (int g = lane; g < a.N; g += 64) {
        double v = temps ? temps[(size_t)b * a.N + g] : initial_temp;
        const int cs = a.cell_state[g];
        if (cs >= 0) {
          T[cs] = v;
        } else if (!temps_only) {
          a.ring[(size_t)b * a.n_ring + (-cs - 1)] = v;
          rlo = fmin(rlo, v);
          rhi = fmax(rhi, v);
        } else { // sb_set_temps leaves exterior space alone: the grid mean counts what the ring holds, not what `temps` says
          v = ring_uniform ? S0[16] : a.ring[(size_t)b * a.n_ring + (-cs - 1)];
        }
        s += v;
      }
      rlo = wave_min(rlo);
      rhi = wave_max(rhi);
    } else {
      double *T = a.temp + (size_t)b * a.Np;
      for (int i = lane; i < a.Np; i += 64) {
        const int g = i - kPad;
        double v = (g >= 0 && g < a.N) ? (temps ? temps[(size_t)b * a.N + g] : initial_temp) : 0.0;
        T[i] = v;
        s += v;
      }
    }
    s = wave_sum(s);
    for (int z = 0; z < a.Z; ++z) {
      double zs = 0.0;
      const int b0 = a.zone_off[z], b1 = a.zone_off[z + 1];
      for (int i = b0 + lane; i < b1; i += 64) {
        int li = a.zone_cells_l[i];
        int g = a.pitch == a.W ? li : (li / a.pitch) * a.W + (li % a.pitch);
        zs += temps ? temps[(size_t)b * a.N + g] : initial_temp;
      }
      zs = wave_sum(zs);
      if (lane == 0) {
        size_t o = (size_t)b * a.Z + z;
        a.zmean[o] = zs / (double)(b1 - b0);
        if (temps_only) continue;
        a.zair[o] = 0.0;   // vav.py:93-99
        a.damper[o] = 0.1;
        a.mode[o] &= kModeMask; // ... the reheat valve closes; the thermostat keeps its mode
        a.qz[o] = 0.0;     // building.py:791 input_q <- 0
      }
    }
    // sb_set_temps: the recirculation temperature of the next step is the mean of the NEW grid (simulator.py:423-426 takes
    // building.temp.mean() every step; S[11] caches it between steps)
    if (lane == 0 && temps_only) a.scal[(size_t)b * kNScal + 11] = s / (double)a.N;
    if (lane == 0 && !temps_only) {
      double *S = a.scal + (size_t)b * kNScal;
      S[0] = a.p.ahu_heat_sp; S[1] = a.p.ahu_cool_sp; S[2] = 0.0; S[3] = 0.0; // air_handler.py:131-139
      S[4] = a.p.blr_setpoint; S[5] = 0.0; S[6] = 0.0; S[7] = 0.0;           // boiler.py:110-121
      S[8] = a.p.blr_setpoint; S[9] = 0.0; S[10] = 0.0;
      S[11] = s / (double)a.N;
      S[12] = S[13] = S[14] = S[15] = 0.0;
      S[16] = a.reg && a.n_ring > 0 ? rlo : 0.0; // extremes of the exterior-space ring (register path)
      S[17] = a.reg && a.n_ring > 0 ? rhi : 0.0;
      // Thermostat._previous_timestamp and SmartDevice._action_timestamp are construction state:
      // Simulator.reset() leaves them alone (thermostat.py:66-69, smart_device.py:71-72)
      // ... but the clock rewinds under them: the boiler's age of its last action goes negative
      if (first) { S[18] = -1.0; S[19] = kAgeNone; }
      else if (!age_is_none(S[19])) S[19] -= (double)steps_since_reset;
    }
  }
}

__global__ void k_observe(Dev a, float *obs, float aux0, float aux1, float aux2, float aux3,
                          float aux4, float aux5, float aux6, double t_amb, const double *t_amb_b,
                          const float *num_occupants, double occ_norm) {
  const float aux[SB_NUM_AUX] = {aux0, aux1, aux2, aux3, aux4, aux5, aux6};
  for (int b = blockIdx.x * blockDim.x + threadIdx.x; b < a.B; b += gridDim.x * blockDim.x) {
    double *S = a.scal + (size_t)b * kNScal;
    observe_boiler(a, S);
    write_obs(a, b, obs, aux, t_amb_b ? t_amb_b[b] : t_amb, S, num_occupants, occ_norm);
  }
}

// Thermostat modes without the valve bit.
__global__ void k_copy_modes(Dev a, int32_t *out) {
  const size_t n = (size_t)a.B * a.Z;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    out[i] = a.mode[i] & kModeMask;
}

// building.temp in the caller's row-major layout, whichever state layout the handle uses.
__global__ void k_copy_temps(Dev a, double *out) {
  size_t n = (size_t)a.B * a.N;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x) {
    const size_t b = i / a.N;
    const int g = (int)(i - b * a.N);
    if (a.reg) {
      const int cs = a.cell_state[g];
      if (cs >= 0) {
        out[i] = a.temp[b * a.state_doubles + cs];
      } else { // exterior space: T_ambient of the last step, or what sb_reset stored
        const double *S = a.scal + b * kNScal;
        out[i] = S[16] == S[17] ? S[16] : a.ring[b * a.n_ring + (-cs - 1)];
      }
    } else {
      out[i] = a.temp[b * a.Np + kPad + g];
    }
  }
}

__global__ void k_copy_scalars(Dev a, double *out) {
  size_t n = (size_t)a.B * kNScalOut;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x)
    out[i] = a.scal[(i / kNScalOut) * kNScal + (i % kNScalOut)];
}

// Per-building algebra before / after the sweep kernel (sb_device.h).  k_pre: one 16-lane row of a wavefront per building,
// lanes = zones -- four buildings per wavefront, sixteen per workgroup; the building's demand sums in the reference's zone
// order through DPP row broadcasts.  k_post: one thread per building (the row shape was measured slower: sb_device.h).
// only >= 0: that building alone (the known-answer taps).
constexpr int kRowsPerBlock = 16; // buildings per workgroup of 256 threads
__global__ void __launch_bounds__(16 * kRowsPerBlock) k_pre(Dev a, StepArgs s, int only) {
  if (blockIdx.x == 0 && threadIdx.x == 0) { // the sweep kernel's draw counter; mode 3: the redo list's counters
    *a.next_b = 0;
    if (a.redo_ctr) a.redo_ctr[0] = a.redo_ctr[1] = 0;
  }
  const int i = threadIdx.x & 15, r = threadIdx.x >> 4;
  if (only >= 0) {
    if (blockIdx.x == 0 && r == 0) pre_building(a, s, only, i);
    return;
  }
  for (int b = blockIdx.x * kRowsPerBlock + r; b < a.B; b += gridDim.x * kRowsPerBlock) pre_building(a, s, b, i);
}

__global__ void __launch_bounds__(64) k_post(Dev a, StepArgs s, int only) { // one thread per building: see sb_device.h
  if (only >= 0) {
    if (blockIdx.x == 0 && threadIdx.x == 0) post_building(a, s, only);
    return;
  }
  for (int b = blockIdx.x * blockDim.x + threadIdx.x; b < a.B; b += gridDim.x * blockDim.x) post_building(a, s, b);
}

// ---------------------------------------------------------------- register-path planning
// Host-only: decides whether k_step_reg can own this floor plan and builds its tables.
struct RegPlan {
  bool ok = false;
  std::string why;
  int NR = 0, P = 0, RS = 0, Ws = 0, r0 = 0, c0 = 0, n_ring = 0, T = 0, state_doubles = 0, ts = 32;
  std::vector<uint8_t> tcls, tcset;
  std::vector<double> csetab; // mode 3: distinct (bU, bD, bL, bR), the pad set (all zero) last
  std::vector<double> tmul;   // mode 3: the tail scan's static multipliers (sweep_common.h, tail_pass_static)
  int lw[4] = {0, 0, 0, 0}, l0[2] = {0, 0}, rowbase[2] = {0, 0}, nch[2] = {0, 0};
  int lag = 0, nslots = 0, steps = 0;
  int r_seam = 0, r_A = 0, r_xchg = 0, lds_bytes = 0, wg_per_cu = 0, AS = 0;
  int r_cmap = 0, wave_doubles = 0, waves_per_wg = 1; // mode 3: four buildings per workgroup share the class words
  int ZRS = 0; // modes 1..3: row stride of the zone-sum scratch
  int stream_ms = 0; // mode 6: several sweeps per pass (step_stream_ms.hip)
  std::vector<unsigned long long> cmapS, amapS, zmapS;
  std::vector<int> cell_state;
  // mode 4 (plan_two)
  int two_sym = 0, two_level = 0, tail_set_base = 0, tail_pad_set = 0;
  std::vector<int> zs_off; // [Z + 2] the compact zone-sum scratch: slots of zone z are zs_off[z] .. zs_off[z + 1] - 1
};

constexpr int kRegSeamPad = 8, kLdsCap = 160 * 1024;
constexpr int kLdsGranule = 1280; // LDS is allocated to workgroups in blocks of this many bytes (gfx950)
constexpr int kRegSlots[] = {32, 66, 96};

// wave 1 runs this many 8-step chunks behind wave 0: every seam value crosses at least one
// workgroup barrier between its write and its (2-steps-early) read, in both directions
int seam_lag(int lw0) { return (lw0 + 8) / 8 + 1; }

bool env_flag(const char *name) {
  const char *e = getenv(name);
  return e && e[0] == '1';
}

// Mode 4 (step_two.hip): one wavefront, two rows per lane, up to 128 + 2 rows and 80 columns
// inside the exterior ring.  (x0, y0): the trim box's corner; zone_of: zone of every cell or -1.
// Three variants (the kernel's template parameters SYM, DENSE):
//   * general: four coefficients per cell, lane l owns rows 2l, 2l + 1, A's slots 72 / 74 in LDS: two
//     buildings per CU;
//   * sym: every cell of the wavefront rows is two coefficients (bV, bH) with T' = A + bV (U + D) + bH (L + R)
//     -- true of interior control volumes (simulator.py:225-237) and of a rectangular building's edge / corner
//     volumes (:130-142, :176-195) once a missing neighbour reads as zero: a pad column follows the last
//     column in the circular slot order (NR > width), the rows are shifted by one so that lane 0's upper
//     cell is a pad row above row 0 (lane l owns rows 2l - 1, 2l), pad rows follow the last row;
//   * dense = sym with only 46 / 56 of A's slots in LDS (the rest in registers): three buildings per CU.
// The zone-sum scratch that aliases A is compact: one slot per (zone, lane) pair that owns a cell of the
// zone (SB1-synth: 1.1 K slots instead of 127 x 65).
bool plan_two(const sb_plan_desc *plan, int Hs, int Ws, int x0, int y0, const std::vector<int> &zone_of, RegPlan &r) {
  const int W = plan->W, Z = plan->Z, ncls = plan->n_classes, N = plan->H * plan->W;
  auto coef = [&](int c, int j) { return plan->class_coef[c * 8 + j]; };
  int NR = 0;
  for (int s : {64, 76, 80}) // the narrowest instantiation that holds the width (SBSIM_NO_TWO_64=1: 76 / 80 alone, the tree before the end of round 4)
    if (!NR && s >= Ws && sweep_two_supported(s) && !(s == 64 && env_flag("SBSIM_NO_TWO_64"))) NR = s;
  if (!NR || Hs > 128 + 2) return false;
  int ts = 32;
  while (ts < ncls + 1) ts *= 2;
  if (ts > 256) return false;
  const int pad = ncls;
  auto cell_class = [&](int R, int col) { // trimmed coordinates
    return (R >= 0 && R < Hs && col >= 0 && col < Ws) ? (int)plan->cell_class[(x0 + R) * W + (y0 + col)] : pad;
  };
  // ---- can every wavefront cell do with two coefficients?  (rows shifted by one: 127 wavefront rows)
  bool sym = NR > Ws && Hs <= 127 + 2 && !env_flag("SBSIM_TWO_GENERAL");
  const int Hw_sym = std::min(Hs, 127);
  for (int R = 0; sym && R < Hw_sym; ++R)
    for (int col = 0; sym && col < Ws; ++col) {
      const int c = cell_class(R, col);
      const double bU = coef(c, 0), bD = coef(c, 1), bL = coef(c, 2), bR = coef(c, 3);
      // a missing neighbour reads as zero only outside the grid (pad row / pad column); below the last
      // wavefront row only when no tail row follows
      const bool v_ok = bU == bD || (bU == 0.0 && R == 0) || (bD == 0.0 && R == Hs - 1);
      const bool h_ok = bL == bR || (bL == 0.0 && col == 0) || (bR == 0.0 && col == Ws - 1);
      sym = v_ok && h_ok;
    }
  const int rowoff = sym ? 1 : 0;
  const int Hw = std::min(Hs, 128 - rowoff), T = Hs - Hw, nl = (Hw + rowoff + 1) / 2; // wavefront rows, tail rows, lanes that own rows
  for (int x = x0 + Hw; x < x0 + Hs; ++x)
    for (int y = y0; y < y0 + Ws; ++y)
      if (zone_of[x * W + y] >= 0) return false; // the tail scan adds no zone sums
  // ---- coefficient sets.  general: (bU, bD, bL, bR) for every cell.  sym: (bV, bH) for the wavefront's
  // cells [kSets][2], then (bU, bD, bL, bR) for the tail cells [kSets / 2][4]; the pad sets last.
  const int kSets = sweep_two_set_table();
  std::vector<int> set_of(ncls + 1, 0), tset_of(ncls + 1, 0);
  std::vector<double> wtab, ttab; // the wavefront's table, the tail cells' table
  auto intern = [](std::vector<double> &tab, int width, const double *v) {
    for (size_t k = 0; k < tab.size() / width; ++k) {
      bool same = true;
      for (int j = 0; j < width; ++j) same = same && tab[width * k + j] == v[j];
      if (same) return (int)k;
    }
    for (int j = 0; j < width; ++j) tab.push_back(v[j]);
    return (int)(tab.size() / width) - 1;
  };
  std::vector<char> in_wave(ncls + 1, 0), in_tail(ncls + 1, 0);
  for (int R = 0; R < Hs; ++R)
    for (int col = 0; col < Ws; ++col) (R < Hw ? in_wave : in_tail)[cell_class(R, col)] = 1;
  for (int c = 0; c < ncls; ++c) {
    const double four[4] = {coef(c, 0), coef(c, 1), coef(c, 2), coef(c, 3)};
    if (sym) {
      if (in_wave[c]) {
        const double two[2] = {four[0] != 0.0 ? four[0] : four[1], four[2] != 0.0 ? four[2] : four[3]};
        set_of[c] = intern(wtab, 2, two);
      }
      if (in_tail[c]) tset_of[c] = intern(ttab, 4, four);
    } else {
      set_of[c] = tset_of[c] = intern(wtab, 4, four);
    }
  }
  const double zeros[4] = {0.0, 0.0, 0.0, 0.0};
  const int wwidth = sym ? 2 : 4;
  wtab.insert(wtab.end(), zeros, zeros + wwidth); // the pad set: no neighbour counts
  set_of[pad] = (int)wtab.size() / wwidth - 1;
  if (sym) {
    ttab.insert(ttab.end(), zeros, zeros + 4);
    tset_of[pad] = (int)ttab.size() / 4 - 1;
    if ((int)wtab.size() / 2 > kSets || (int)ttab.size() / 4 > kSets / 2) return false;
    wtab.resize((size_t)2 * kSets, 0.0);
    r.csetab = wtab;
    r.csetab.insert(r.csetab.end(), ttab.begin(), ttab.end());
  } else {
    tset_of[pad] = set_of[pad];
    if ((int)wtab.size() / 4 > kSets) return false;
    r.csetab = wtab;
  }
  r.two_sym = sym ? 1 : 0;
  r.tail_set_base = sym ? 2 * kSets * 8 : 0;
  r.tail_pad_set = r.tail_set_base + tset_of[pad] * 32;

  // ---- the lanes' cells: register J = 2 * slot + k of lane l is row 2l + k - rowoff, column (slot - l) mod NR
  const int NE = 2 * NR;
  auto row_of = [&](int lane, int k) { return 2 * lane + k - rowoff; };
  auto reg_cell = [&](int lane, int J, int &R, int &col) {
    const int j = J / 2;
    col = ((j - lane) % NR + NR) % NR;
    R = row_of(lane, J & 1);
    return J < NE && R >= 0 && R < Hw && col < Ws;
  };
  // ---- the zone-sum scratch: one slot per (zone, lane) that owns a cell of the zone; zone Z (every cell
  // outside a zone, the tail rows) has a slot for every lane
  std::vector<std::vector<int>> zl(Z + 1);
  {
    std::vector<char> seen((size_t)(Z + 1) * 64, 0);
    for (int lane = 0; lane < 64; ++lane) {
      seen[(size_t)Z * 64 + lane] = 1;
      for (int J = 0; J < NE; ++J) {
        int R, col;
        if (reg_cell(lane, J, R, col) && zone_of[(x0 + R) * W + (y0 + col)] >= 0) seen[(size_t)zone_of[(x0 + R) * W + (y0 + col)] * 64 + lane] = 1;
      }
    }
    for (int z = 0; z <= Z; ++z)
      for (int lane = 0; lane < 64; ++lane)
        if (seen[(size_t)z * 64 + lane]) zl[z].push_back(lane);
  }
  r.zs_off.assign(Z + 2, 0);
  for (int z = 0; z <= Z; ++z) r.zs_off[z + 1] = r.zs_off[z] + (int)zl[z].size();
  const int zs_slots = r.zs_off[Z + 1];
  if (zs_slots > 65535) return false;
  auto zslot = [&](int z, int lane) {
    const auto it = std::lower_bound(zl[z].begin(), zl[z].end(), lane);
    return r.zs_off[z] + (int)(it - zl[z].begin());
  };

  // ---- LDS: [sets 4 kSets][ap g: 2 (ncls + 1), rounded to 8][first tail row][A rows: nl (+ 1 for the idle lanes)]
  // As many buildings per CU as the kernel has a level for (A's other slots stream from L2: step_two.hip).
  auto lds_bytes_for = [&](int level, int &AS, int &r_seam, int &r_A) {
    AS = sweep_two_a_stride(NR, level);
    int off = 4 * kSets + ((2 * (ncls + 1) + 7) & ~7);
    r_seam = off; off += sweep_two_seam_doubles(NR);
    r_A = off; off += std::max((nl < 64 ? nl + 1 : 64) * AS, zs_slots);
    return off * 8;
  };
  auto per_cu = [&](int bytes) { return std::min(4, kLdsCap / ((bytes + kLdsGranule - 1) / kLdsGranule * kLdsGranule)); };
  int AS = 0, r_seam = 0, r_A = 0, bytes = 0, level = 0;
  int max_level = sym ? sweep_two_levels() - 1 : 0;
  if (const char *e = getenv("SBSIM_TWO_MAX_LEVEL")) max_level = std::min(max_level, std::max(0, atoi(e)));
  for (int lv = 0; lv <= max_level; ++lv) { // the lowest level that holds the most buildings
    int as_, rs_, ra_;
    const int b_ = lds_bytes_for(lv, as_, rs_, ra_);
    if (lv == 0 || per_cu(b_) > per_cu(bytes)) { level = lv; bytes = b_; AS = as_; r_seam = rs_; r_A = ra_; }
  }
  if (const char *padb = getenv("SBSIM_DEBUG_LDS_PAD")) bytes += atoi(padb);
  r.two_level = level;
  r.r_seam = r_seam; r.r_A = r_A; r.r_xchg = 0; r.AS = AS;
  r.lds_bytes = bytes;
  r.wg_per_cu = per_cu(bytes);
  if (r.wg_per_cu < 1) { r.csetab.clear(); return false; }

  r.NR = NR; r.P = 4; r.RS = 64; r.Ws = Ws; r.r0 = x0; r.c0 = y0; r.n_ring = N - Hs * Ws;
  r.T = T; r.ts = ts;
  r.state_doubles = NE * 64 + T * NR;
  r.lw[0] = nl; r.l0[0] = 0; r.rowbase[0] = 0; r.nch[0] = 0;
  r.lag = 0; r.nslots = 0;
  r.steps = NR + nl - 1 + 4 * T;
  const int NW = NR / 4, NWD = (NE + 3) / 4; // step_two.hip: class words of four steps, NR / 4 per lane
  r.cmapS.assign((size_t)NW * 64, 0);
  r.amapS.assign((size_t)NWD * 64, 0);
  r.zmapS.assign((size_t)NWD * 64, 0);
  r.tcls.assign((size_t)std::max(T, 1) * NR, (uint8_t)pad);
  r.tcset.assign((size_t)std::max(T, 1) * NR, (uint8_t)(8 * tset_of[pad]));
  for (int t = 0; t < T; ++t)
    for (int c = 0; c < NR; ++c) {
      r.tcls[(size_t)t * NR + c] = (uint8_t)cell_class(Hw + t, c);
      r.tcset[(size_t)t * NR + c] = (uint8_t)(8 * tset_of[cell_class(Hw + t, c)]);
    }
  for (int lane = 0; lane < 64; ++lane) {
    auto row_class = [&](int k, int col) { // upper / lower cell of the lane at a column of the wavefront rows
      const int R = row_of(lane, k);
      return (R >= 0 && R < Hw && col >= 0 && col < NR) ? cell_class(R, col) : pad;
    };
    for (int wd = 0; wd < NW; ++wd) { // a byte per cell and step: the coefficient sets of the lane's (upper, lower) cell
      unsigned long long word = 0;
      for (int k = 0; k < 4; ++k) { // at step s (of a ramp-up or of any period) the lane works on column (s - lane) mod NR
        const int col = ((4 * wd + k - lane) % NR + NR) % NR;
        word |= (unsigned long long)set_of[row_class(0, col)] << (16 * k);
        word |= (unsigned long long)set_of[row_class(1, col)] << (16 * k + 8);
      }
      r.cmapS[(size_t)wd * 64 + lane] = word;
    }
    for (int g = 0; g < NWD; ++g) { // register J = 2 * slot + k
      unsigned long long aword = 0, zword = 0;
      for (int k = 0; k < 4; ++k) {
        const int J = 4 * g + k;
        int R, col;
        const bool cell = reg_cell(lane, J, R, col);
        aword |= (unsigned long long)((cell ? cell_class(R, col) : pad) * 16) << (16 * k);
        int z = Z; // the dump zone
        if (cell && zone_of[(x0 + R) * W + (y0 + col)] >= 0) z = zone_of[(x0 + R) * W + (y0 + col)];
        zword |= (unsigned long long)zslot(z, lane) << (16 * k);
      }
      r.amapS[(size_t)g * 64 + lane] = aword;
      r.zmapS[(size_t)g * 64 + lane] = zword;
    }
  }
  r.cell_state.assign(N, 0);
  int ring = 0;
  for (int x = 0; x < plan->H; ++x)
    for (int y = 0; y < W; ++y) {
      const int R = x - x0, col = y - y0;
      if (R < 0 || R >= Hs || col < 0 || col >= Ws) { r.cell_state[x * W + y] = -(++ring); continue; }
      if (R >= Hw) { r.cell_state[x * W + y] = NE * 64 + (R - Hw) * NR + col; continue; }
      const int lane = (R + rowoff) >> 1;
      r.cell_state[x * W + y] = (2 * ((col + lane) % NR) + ((R + rowoff) & 1)) * 64 + lane;
    }
  r.ok = true;
  return true;
}

// Mode 5 (step_band.hip): two to four wavefronts per building, one row per lane (rows 64 w .. 64 w + 63) + at
// most two tail rows, up to 96 columns inside the exterior ring; sweeps overlapped in predicted blocks.
bool plan_band(const sb_plan_desc *plan, int Hs, int Ws, int x0, int y0, const std::vector<int> &zone_of, RegPlan &r) {
  const int W = plan->W, Z = plan->Z, ncls = plan->n_classes, N = plan->H * plan->W;
  auto coef = [&](int c, int j) { return plan->class_coef[c * 8 + j]; };
  int NR = 0;
  for (int s : {68, 72, 76, 80, 84, 88, 92, 96}) // the narrowest instantiation that holds the width: a period is NR steps
    if (!NR && s >= Ws && sweep_band_supported(s)) NR = s;
  if (!NR || Hs <= 64) return false;
  // wavefronts: the fewest whose rows + two tail rows hold the plan; a zone cell in a tail row needs one more
  int NWV = 0, T = 0;
  for (int w = 2; w <= sweep_band_max_waves() && !NWV; ++w) {
    if (Hs > 64 * w + 2) continue;
    const int t = std::max(0, Hs - 64 * w);
    bool zone_free = true; // the tail scan adds no zone sums
    for (int x = x0 + Hs - t; x < x0 + Hs; ++x)
      for (int y = y0; y < y0 + Ws; ++y) zone_free = zone_free && zone_of[x * W + y] < 0;
    if (zone_free) { NWV = w; T = t; }
  }
  if (!NWV) return false;
  const int Hw = Hs - T, RS = 64 * NWV;
  int ts = 32;
  while (ts < ncls + 1) ts *= 2;
  if (ts > 256) return false;
  const int pad = ncls;
  std::vector<int> set_of(ncls + 1, 0);
  r.csetab.clear();
  for (int c = 0; c < ncls; ++c) {
    int found = -1;
    for (size_t k = 0; k < r.csetab.size() / 4 && found < 0; ++k)
      if (r.csetab[4 * k] == coef(c, 0) && r.csetab[4 * k + 1] == coef(c, 1) && r.csetab[4 * k + 2] == coef(c, 2) &&
          r.csetab[4 * k + 3] == coef(c, 3)) found = (int)k;
    if (found < 0) {
      found = (int)r.csetab.size() / 4;
      for (int j = 0; j < 4; ++j) r.csetab.push_back(coef(c, j));
    }
    set_of[c] = found;
  }
  set_of[pad] = (int)r.csetab.size() / 4; // the pad set: no neighbour counts
  for (int j = 0; j < 4; ++j) r.csetab.push_back(0.0);
  if ((int)r.csetab.size() / 4 > sweep_band_set_table()) { r.csetab.clear(); return false; }
  auto cell_class = [&](int R, int col) { // trimmed coordinates
    return (R >= 0 && R < Hs && col >= 0 && col < Ws) ? (int)plan->cell_class[(x0 + R) * W + (y0 + col)] : pad;
  };
  auto zone_at = [&](int R, int col) { // the zone of a wavefront-row cell; Z: none (the dump zone)
    if (R >= Hw || col >= Ws) return Z;
    const int z = zone_of[(x0 + R) * W + (y0 + col)];
    return z >= 0 ? z : Z;
  };
  // ---- the zone-sum scratch: one slot per (zone, row) whose row holds a cell of the zone; zone Z (every
  // cell outside a zone, the tail rows) has a slot for every lane of every wavefront
  std::vector<std::vector<int>> zl(Z + 1);
  for (int R = 0; R < RS; ++R) {
    std::vector<char> seen(Z + 1, 0);
    seen[Z] = 1;
    for (int col = 0; col < Ws && R < Hw; ++col) seen[zone_at(R, col)] = 1;
    for (int z = 0; z <= Z; ++z)
      if (seen[z]) zl[z].push_back(R);
  }
  r.zs_off.assign(Z + 2, 0);
  for (int z = 0; z <= Z; ++z) r.zs_off[z + 1] = r.zs_off[z] + (int)zl[z].size();
  const int zs_slots = r.zs_off[Z + 1];
  if (zs_slots > 65535) { r.csetab.clear(); return false; }
  auto zslot = [&](int z, int R) {
    const auto it = std::lower_bound(zl[z].begin(), zl[z].end(), R);
    return r.zs_off[z] + (int)(it - zl[z].begin());
  };

  const int AS = sweep_band_lds_slots(NR);
  int off = 4 * sweep_band_set_table() + 2 * ts;
  r.r_seam = off; off += sweep_band_seam_doubles(NR, NWV);
  r.r_xchg = off; off += sweep_band_sync_doubles(NWV);
  off = (off + 1) & ~1;
  r.r_A = off; off += std::max(RS * AS, zs_slots); // the zone-sum scratch aliases A
  r.AS = AS;
  r.lds_bytes = off * 8;
  if (const char *padb = getenv("SBSIM_DEBUG_LDS_PAD")) r.lds_bytes += atoi(padb);
  r.wg_per_cu = std::min(4 / NWV, kLdsCap / ((r.lds_bytes + kLdsGranule - 1) / kLdsGranule * kLdsGranule)); // one wavefront per SIMD
  if (r.wg_per_cu < 1) { r.csetab.clear(); r.zs_off.clear(); return false; }

  r.NR = NR; r.P = 5; r.RS = RS; r.Ws = Ws; r.r0 = x0; r.c0 = y0; r.n_ring = N - Hs * Ws;
  r.T = T; r.ts = ts;
  r.state_doubles = NR * RS + T * NR;
  for (int w = 0; w < 4; ++w) r.lw[w] = w < NWV ? std::max(0, std::min(64, Hw - 64 * w)) : 0;
  r.l0[0] = r.l0[1] = 0; r.rowbase[0] = 0; r.rowbase[1] = 64;
  r.nch[0] = r.nch[1] = 0; r.lag = 0; r.nslots = 0;
  r.steps = NR + 4 * T;
  const int NW = NR + 63, NWD = NR / 4;
  std::vector<uint32_t> cw((size_t)NWV * NW * 64, 0);
  r.amapS.assign((size_t)NWV * NWD * 64, 0);
  r.zmapS.assign((size_t)NWV * NWD * 64, 0);
  r.tcls.assign((size_t)std::max(T, 1) * NR, (uint8_t)pad);
  r.tcset.assign((size_t)std::max(T, 1) * NR, (uint8_t)(8 * set_of[pad]));
  for (int t = 0; t < T; ++t)
    for (int c = 0; c < NR; ++c) {
      r.tcls[(size_t)t * NR + c] = (uint8_t)cell_class(Hw + t, c);
      r.tcset[(size_t)t * NR + c] = (uint8_t)(8 * set_of[cell_class(Hw + t, c)]);
    }
  for (int w = 0; w < NWV; ++w)
    for (int lane = 0; lane < 64; ++lane) {
      const int R = 64 * w + lane;
      const bool valid = R < Hw;
      for (int st = 0; st < NW; ++st) { // one word per step: the set offset of the lane's cell
        int col = st - lane;
        if (col >= NR) col -= NR; // rolling periods: the lane is in its next sweep
        const int c = (valid && col >= 0) ? cell_class(R, col) : pad;
        cw[((size_t)w * NW + st) * 64 + lane] = (uint32_t)(set_of[c] * 32);
      }
      for (int g = 0; g < NWD; ++g) {
        unsigned long long aword = 0, zword = 0;
        for (int k = 0; k < 4; ++k) {
          const int j = 4 * g + k, col = ((j - lane) % NR + NR) % NR;
          const bool cell = valid && col < Ws;
          aword |= (unsigned long long)((cell ? cell_class(R, col) : pad) * 16) << (16 * k);
          zword |= (unsigned long long)zslot(cell ? zone_at(R, col) : Z, R) << (16 * k);
        }
        r.amapS[((size_t)w * NWD + g) * 64 + lane] = aword;
        r.zmapS[((size_t)w * NWD + g) * 64 + lane] = zword;
      }
    }
  r.cmapS.assign((cw.size() + 1) / 2, 0);
  std::memcpy(r.cmapS.data(), cw.data(), cw.size() * sizeof(uint32_t));
  r.cell_state.assign(N, 0);
  int ring = 0;
  for (int x = 0; x < plan->H; ++x)
    for (int y = 0; y < W; ++y) {
      const int R = x - x0, col = y - y0;
      if (R < 0 || R >= Hs || col < 0 || col >= Ws) { r.cell_state[x * W + y] = -(++ring); continue; }
      if (R >= Hw) { r.cell_state[x * W + y] = NR * RS + (R - Hw) * NR + col; continue; }
      const int slot = (col + (R & 63)) % NR;
      r.cell_state[x * W + y] = (slot / 2) * 2 * RS + R * 2 + (slot & 1); // state layout [NR / 2][64 W][2]
    }
  r.ok = true;
  return true;
}

// Mode 6 (step_stream.hip): the grid stays in global memory; W = ceil(rows / 64) wavefronts per building
// (W <= 16), any width whose seam rows fit in LDS.  For floor plans no other kernel holds.
bool plan_stream(const sb_plan_desc *plan, RegPlan &r, std::string &why) {
  const int H = plan->H, W = plan->W, Z = plan->Z, ncls = plan->n_classes, N = H * W;
  auto coef = [&](int c, int j) { return plan->class_coef[c * 8 + j]; };
  auto cls_at = [&](int x, int y) { return (int)plan->cell_class[x * W + y]; };
  std::vector<char> ambient(ncls, 0); // T' = T_ambient whatever the neighbours are (simulator.py:256-258)
  for (int c = 0; c < ncls; ++c)
    ambient[c] = coef(c, 0) == 0 && coef(c, 1) == 0 && coef(c, 2) == 0 && coef(c, 3) == 0 &&
                 coef(c, 4) == 0 && coef(c, 5) == 1.0 && coef(c, 6) == 0;
  int x0 = H, x1 = -1, y0 = W, y1 = -1;
  for (int x = 0; x < H; ++x)
    for (int y = 0; y < W; ++y)
      if (!ambient[cls_at(x, y)]) {
        x0 = std::min(x0, x); x1 = std::max(x1, x);
        y0 = std::min(y0, y); y1 = std::max(y1, y);
      }
  if (x1 < 0) { why = "no cell inside the building"; return false; }
  const int Hs = x1 - x0 + 1, Ws = y1 - y0 + 1;
  for (int y = y0; y <= y1; ++y)
    if (coef(cls_at(x0, y), 0) != 0 || coef(cls_at(x1, y), 1) != 0) { why = "coupling across the trim box"; return false; }
  for (int x = x0; x <= x1; ++x)
    if (coef(cls_at(x, y0), 2) != 0 || coef(cls_at(x, y1), 3) != 0) { why = "coupling across the trim box"; return false; }
  const int NWV = (Hs + 63) / 64;
  if (NWV > 16) { why = "more than 1,024 rows inside the building (try the other orientation)"; return false; }
  const int NS = std::max(72, (Ws + 1) & ~1), RS = 64 * NWV;
  int ts = 32;
  while (ts < ncls + 1) ts *= 2;
  if (ts > 256) { why = "more than 255 cell classes"; return false; }
  const int pad = ncls;
  std::vector<int> set_of(ncls + 1, 0);
  r.csetab.clear();
  for (int c = 0; c < ncls; ++c) {
    int found = -1;
    for (size_t k = 0; k < r.csetab.size() / 4 && found < 0; ++k)
      if (r.csetab[4 * k] == coef(c, 0) && r.csetab[4 * k + 1] == coef(c, 1) && r.csetab[4 * k + 2] == coef(c, 2) &&
          r.csetab[4 * k + 3] == coef(c, 3)) found = (int)k;
    if (found < 0) {
      found = (int)r.csetab.size() / 4;
      for (int j = 0; j < 4; ++j) r.csetab.push_back(coef(c, j));
    }
    set_of[c] = found;
  }
  set_of[pad] = (int)r.csetab.size() / 4;
  for (int j = 0; j < 4; ++j) r.csetab.push_back(0.0);
  if ((int)r.csetab.size() / 4 > sweep_stream_set_table()) { why = "more than 31 distinct coefficient sets"; r.csetab.clear(); return false; }
  if (Z > 65534) { why = "too many zones"; r.csetab.clear(); return false; }
  const int ZC = sweep_stream_zone_columns();
  int off = 4 * sweep_stream_set_table() + 2 * ts;
  // step_stream_ms.hip (several sweeps per pass over the grid: a seam row per sweep) -- OPT-IN, SBSIM_STREAM_MS=1: correct
  // (bit-identical iterates, tests/test_gpu_parity.py) but not faster yet (LABNOTES.md 5.4c: 256 registers leave two
  // wavefronts per SIMD, and a step of four sweep slots takes 3.3x a step of step_stream.hip's one).  Needs its LDS to fit
  // and the last wavefront to own the spare rows its bands need (the band of sweep j sits j rows further up).
  {
    const int ms_off = off + sweep_stream_ms_seam_doubles(NS, NWV) + sweep_stream_ms_xchg_doubles(NWV) + (Z + 1) * ZC;
    r.stream_ms = kExperimental && env_flag("SBSIM_STREAM_MS") && ms_off * 8 <= kLdsCap && 64 * NWV - Hs >= sweep_stream_ms_sweeps() - 1 &&
                  NWV <= 8; // (its 256 registers: two wavefronts per SIMD, workgroups of at most 8)
  }
  r.r_seam = off; off += r.stream_ms == 1 ? sweep_stream_ms_seam_doubles(NS, NWV) : 2 * NWV * (NS + 8);
  // overlapped sweeps (step_stream.hip's k_sweep_stream_roll, round 5) -- OPT-IN, SBSIM_STREAM_ROLL=1: stream_ms == 2.  Exact
  // (tests/test_gpu_parity.py), 9 % faster on R9 forced here, 9 % SLOWER on the 299 x 401 plan it was written for: with three
  // workgroups per CU another workgroup already fills a wavefront's idle steps, and the sweep started in vain is traffic
  if (kExperimental && !r.stream_ms && env_flag("SBSIM_STREAM_ROLL")) r.stream_ms = 2;
  r.r_xchg = off; off += r.stream_ms == 1 ? sweep_stream_ms_xchg_doubles(NWV) : 32 + 64 * NWV + (r.stream_ms == 2 ? sweep_stream_roll_xchg_extra_doubles() : 0); // progress, max|delta| parts, the publish scratch
  r.r_A = off; off += (Z + 1) * ZC;
  r.lds_bytes = off * 8;
  if (r.lds_bytes > kLdsCap) { why = "the seam rows (rows / 64 x columns x 16 bytes) and the zone sums do not fit in 160 KiB of LDS"; r.csetab.clear(); return false; }
  // workgroups per CU: LDS, threads (2,048 per CU) and registers (<= 128 per lane: four wavefronts per SIMD)
  r.wg_per_cu = std::max(1, std::min({kLdsCap / ((r.lds_bytes + kLdsGranule - 1) / kLdsGranule * kLdsGranule), 32 / NWV, (r.stream_ms == 1 ? 8 : 16) / NWV}));
  r.NR = NS; r.P = 6; r.RS = RS; r.Ws = Ws; r.r0 = x0; r.c0 = y0; r.n_ring = N - Hs * Ws;
  r.T = 0; r.ts = ts; r.AS = 0;
  r.state_doubles = NS * RS;
  r.lw[0] = std::min(Hs, 64); r.lw[1] = NWV; // lw[1]: wavefronts per building
  r.steps = 64 * (NWV - 1) + NS + 63;
  auto cell_class = [&](int R, int col) {
    return (R >= 0 && R < Hs && col >= 0 && col < Ws) ? cls_at(x0 + R, y0 + col) : pad;
  };
  std::vector<int> zone_of(N, -1);
  for (int z = 0; z < Z; ++z)
    for (int i = plan->zone_off[z]; i < plan->zone_off[z + 1]; ++i) zone_of[plan->zone_cells[i]] = z;
  const int NW = NS + 63;
  std::vector<uint32_t> cw((size_t)NWV * NW * 64, 0);
  std::vector<uint16_t> zm((size_t)NWV * NS * 64, (uint16_t)Z);
  for (int w = 0; w < NWV; ++w)
    for (int lane = 0; lane < 64; ++lane) {
      const int R = 64 * w + lane;
      for (int t = 0; t < NW; ++t) {
        const int col = t - lane;
        const int c = (R < Hs && col >= 0 && col < NS) ? cell_class(R, col) : pad;
        cw[((size_t)w * NW + t) * 64 + lane] = (uint32_t)(set_of[c] * 32) | ((uint32_t)(c * 16) << 16);
      }
      for (int s = 0; s < NS; ++s) {
        const int col = ((s - lane) % NS + NS) % NS;
        if (R < Hs && col < Ws && zone_of[(x0 + R) * W + (y0 + col)] >= 0)
          zm[((size_t)w * NS + s) * 64 + lane] = (uint16_t)zone_of[(x0 + R) * W + (y0 + col)];
      }
    }
  r.cmapS.assign((cw.size() + 1) / 2, 0);
  std::memcpy(r.cmapS.data(), cw.data(), cw.size() * sizeof(uint32_t));
  r.zmapS.assign((zm.size() + 3) / 4, 0);
  std::memcpy(r.zmapS.data(), zm.data(), zm.size() * sizeof(uint16_t));
  r.amapS.assign(1, 0);
  r.tcls.assign(1, (uint8_t)0);
  r.tcset.assign(1, (uint8_t)0);
  r.cell_state.assign(N, 0);
  int ring = 0;
  for (int x = 0; x < H; ++x)
    for (int y = 0; y < W; ++y) {
      const int R = x - x0, col = y - y0;
      if (R < 0 || R >= Hs || col < 0 || col >= Ws) { r.cell_state[x * W + y] = -(++ring); continue; }
      r.cell_state[x * W + y] = ((col + (R & 63)) % NS) * RS + R; // state layout [NS][RS]
    }
  r.ok = true;
  return true;
}

// lds_per_cu: buildings per CU the LDS-grid kernel would hold (0: the plan does not fit it).
void plan_reg(const sb_plan_desc *plan, int lds_per_cu, RegPlan &r) {
  const int H = plan->H, W = plan->W, Z = plan->Z, ncls = plan->n_classes, N = H * W;
  auto coef = [&](int c, int j) { return plan->class_coef[c * 8 + j]; };
  auto cls_at = [&](int x, int y) { return (int)plan->cell_class[x * W + y]; };
  std::vector<char> ambient(ncls, 0); // T' = T_ambient whatever the neighbours are (simulator.py:256-258)
  for (int c = 0; c < ncls; ++c)
    ambient[c] = coef(c, 0) == 0 && coef(c, 1) == 0 && coef(c, 2) == 0 && coef(c, 3) == 0 &&
                 coef(c, 4) == 0 && coef(c, 5) == 1.0 && coef(c, 6) == 0;
  int x0 = H, x1 = -1, y0 = W, y1 = -1;
  for (int x = 0; x < H; ++x)
    for (int y = 0; y < W; ++y)
      if (!ambient[cls_at(x, y)]) {
        x0 = std::min(x0, x); x1 = std::max(x1, x);
        y0 = std::min(y0, y); y1 = std::max(y1, y);
      }
  if (x1 < 0) { r.why = "no cell inside the building"; return; }
  const int Hs = x1 - x0 + 1, Ws = y1 - y0 + 1;
  // nothing may couple to a cell outside the trim box
  for (int y = y0; y <= y1; ++y)
    if (coef(cls_at(x0, y), 0) != 0 || coef(cls_at(x1, y), 1) != 0) { r.why = "coupling across the trim box"; return; }
  for (int x = x0; x <= x1; ++x)
    if (coef(cls_at(x, y0), 2) != 0 || coef(cls_at(x, y1), 3) != 0) { r.why = "coupling across the trim box"; return; }
  std::vector<int> zone_of(N, -1);
  for (int z = 0; z < Z; ++z)
    for (int i = plan->zone_off[z]; i < plan->zone_off[z + 1]; ++i) zone_of[plan->zone_cells[i]] = z;
  // mode 5: two wavefronts, one row per lane, sweeps overlapped in blocks (67..130 rows, <= 80 columns).
  // Measured level with mode 4 (profiles/r04_band_vs_two_rows.txt): all four SIMDs run, but four
  // wavefronts share the CU's LDS pipe, every block starts with wavefront 1's 64-step lag and the tail
  // scan sits on wavefront 1's critical path.  Kept, tested, behind SBSIM_BAND_PATH=1.
  if (Hs > 64 + 2 && env_flag("SBSIM_BAND_PATH") && plan_band(plan, Hs, Ws, x0, y0, zone_of, r)) return;
  // mode 4: one wavefront, two rows per lane (67..130 rows, <= 80 columns)
  if (Hs > 64 + 2 && !env_flag("SBSIM_NO_TWO_ROW_PATH") && plan_two(plan, Hs, Ws, x0, y0, zone_of, r)) return;
  // mode 5 again, for what mode 4 does not hold: beyond 128 rows (up to 258) three or four wavefronts share a
  // building; 67..130 rows with 81..96 columns two (measured 1.2-2.6x the two-wavefront k_sweep_reg / the LDS-grid
  // kernel on 109 x 92, 113 x 93 and 125 x 97: tools/bench_mid_plans.py)
  if (Hs > 64 + 2 && !env_flag("SBSIM_NO_BAND_PATH") && plan_band(plan, Hs, Ws, x0, y0, zone_of, r)) return;
  auto pick_slots = [&](int mode) { // narrowest instantiation that holds the width and the class count
    if (mode == 3) {
      for (int s : {64, 72, 80, 88, 96}) // SBSIM_NO_ROLL_64=1: the 96-slot instantiation alone (the tree before round 4's last additions)
        if (s >= Ws && sweep_roll_supported(s) && ncls + 1 <= 32 && !(s < 96 && env_flag("SBSIM_NO_ROLL_64"))) return s;
      return 0;
    }
    for (int s : kRegSlots)
      if (s >= Ws && sweep_reg_supported(s, mode) && ncls + 1 <= sweep_reg_table_stride(s, mode)) return s;
    return 0;
  };
  // mode 1: one wavefront; mode 3: one wavefront + one or two tail rows finished by a scan
  // (no zone cell may sit in a tail row); mode 2: two wavefronts
  int P = Hs <= 64 ? 1 : 0, NR = P ? pick_slots(1) : 0;
  // ... but not on k_sweep_reg<NR,1> (round 1's kernel: a sweep of NR + rows - 1 steps, no overlap) when k_sweep_roll's
  // period is no longer than that: the roll kernel without tail rows, the lanes beyond the plan own pad rows
  // (measured, 65,536 buildings: 47 x 98 / 6 zones 6.8 -> 1.9 ms per step, 62 x 97 4.8 -> 2.5, 49 x 50 8.9 -> 7.6;
  // 25 x 38 and 17 x 25 stay: 3.3 against 4.4, 2.6 against 4.9)
  // (only when the roll kernel's own limits hold -- its zone-sum scratch takes <= 31 zones; its LDS layout always fits: checked
  // HERE, before the switch, so that a plan it cannot take stays on k_sweep_reg<NR,1> instead of falling through to the
  // LDS-grid kernel)
  if (P == 1 && NR && !env_flag("SBSIM_NO_ROLL_SMALL") && pick_slots(3) && Z <= 31 && NR + Hs - 1 >= pick_slots(3)) { P = 3; NR = pick_slots(3); }
  if (!P && Hs <= 64 + 2 && pick_slots(3)) {
    bool zone_free = true;
    for (int x = x0 + 64; x <= x1; ++x)
      for (int y = y0; y <= y1; ++y) zone_free = zone_free && zone_of[x * W + y] < 0;
    if (zone_free) { P = 3; NR = pick_slots(3); }
  }
  if (!P && Hs <= 128) {
    P = 2; NR = pick_slots(2);
    // a two-wavefront variant that runs one wavefront per SIMD holds two buildings per CU and
    // pays the seam lag: measured no faster than the LDS-grid kernel at two buildings per CU
    if (NR && sweep_reg_waves_per_simd(NR, 2) == 1 && lds_per_cu >= 2) {
      r.why = "the LDS-grid kernel holds as many buildings per CU";
      return;
    }
  }
  if (!P) { r.why = "more than 128 rows inside the building"; return; }
  if (!NR) { r.why = "no kernel variant for this width and class count"; return; }
  const int TS = P == 3 ? 32 : sweep_reg_table_stride(NR, P), cscale = 256 / TS; // class byte = class * cscale
  const int nl_slots = P == 3 ? sweep_roll_lds_slots(NR) : sweep_reg_lds_slots(NR, P);
  r.ts = TS;
  const int RS = P == 3 ? 64 : Hs;
  // the zone-sum scratch aliases A.  Modes 1, 2: [Z + 1][ZRS = rows | 1] (odd stride: the zone reduce reads 16 zone rows at once),
  // 16-bit byte offsets.  Mode 3 (step_roll.hip): [64 rows][ZRS = (Z + 1) | 1], a slot's offset inside its lane's row is zone * 8:
  // one byte (Z <= 31: every zone has a cell class of its own and the class table holds 32)
  const int ZRS = P == 3 ? ((Z + 1) | 1) : (RS | 1);
  if (P == 3 ? (Z > 31 || ZRS > nl_slots) : ((size_t)(Z + 1) * ZRS > (size_t)RS * nl_slots || (size_t)(Z + 1) * ZRS * 8 > 65535)) {
    r.why = "too many zones for the zone-sum scratch"; // it aliases A
    return;
  }
  r.ZRS = ZRS;
  r.NR = NR; r.P = P; r.RS = RS; r.Ws = Ws; r.r0 = x0; r.c0 = y0; r.n_ring = N - Hs * Ws;
  r.T = P == 3 ? std::max(0, Hs - 64) : 0;
  r.state_doubles = NR * RS + r.T * NR;
  const int maxch = (NR + 63 + 7) / 8;
  if (P == 3) {
    r.lw[0] = 64; r.l0[0] = 0; r.rowbase[0] = 0;
    r.nch[0] = (NR + 63 + 7) / 8;
    r.lag = 0; r.nslots = r.nch[0];
    r.steps = NR + 4 * r.T; // overlapped sweeps: NR steps per sweep; a tail row's scan costs about 4 wavefront steps
  } else if (P == 1) {
    r.lw[0] = Hs; r.l0[0] = 0; r.rowbase[0] = 0;
    r.nch[0] = (NR + Hs - 1 + 7) / 8;
    r.lag = 0; r.nslots = r.nch[0];
    r.steps = sweep_reg_overlaps_sweeps(NR, 1) ? NR : NR + Hs - 1;
  } else {
    int best = -1, best_slots = 1 << 30;
    for (int a0 = Hs - 64; a0 <= 64; ++a0) {
      if (a0 < 1 || Hs - a0 < 1) continue;
      const int n0 = (NR + a0 - 1 + 7) / 8, n1 = (NR + (Hs - a0) - 1 + 7) / 8;
      const int slots = std::max(n0, seam_lag(a0) + n1);
      if (slots < best_slots || (slots == best_slots && std::abs(2 * a0 - Hs) < std::abs(2 * best - Hs))) {
        best = a0; best_slots = slots;
      }
    }
    r.lw[0] = best; r.lw[1] = Hs - best;
    r.l0[0] = 64 - best; r.l0[1] = 0;       // wave 0's rows sit in its top lanes: seam row = lane 63
    r.rowbase[0] = 0; r.rowbase[1] = best;
    r.nch[0] = (NR + r.lw[0] - 1 + 7) / 8; r.nch[1] = (NR + r.lw[1] - 1 + 7) / 8;
    r.lag = seam_lag(best); r.nslots = best_slots;
    r.steps = 8 * (r.lag + r.nch[1]); // critical path: wave 1 starts `lag` chunks late
  }
  // LDS layout (doubles).  A's rows are read lane-per-row at a common column: an odd row
  // stride spreads the 64 lanes over all banks (stride 96 doubles would put them all on one)
  // -- unless the extra column costs a resident building.
  auto layout = [&](int AS) {
    r.AS = AS;
    if (P == 3) { // step_roll.hip: [coefficient sets | class words] shared, then per wavefront [ap g | first tail row | A]
      r.waves_per_wg = sweep_roll_waves();
      r.r_cmap = 4 * TS;
      r.r_seam = 2 * TS;
      r.r_A = r.r_seam + ((sweep_roll_seam_doubles(NR, r.T) + 1) & ~1); // 16-byte aligned rows
      r.wave_doubles = r.r_A + RS * nl_slots; // rows of AS slots, then [RS][nl_slots - AS]
      r.lds_bytes = (r.r_cmap + (NR / 8) * 64 + sweep_roll_tail_mul_doubles(NR, r.T) + r.waves_per_wg * r.wave_doubles) * 8;
      if (const char *pad = getenv("SBSIM_DEBUG_LDS_PAD")) r.lds_bytes += atoi(pad);
      r.wg_per_cu = r.lds_bytes <= kLdsCap ? 1 : 0;
      return;
    }
    int off = 5 * TS + TS;
    r.r_seam = off;
    off += P == 2 ? 2 * (NR + 2 * kRegSeamPad) : 0;
    r.r_A = off; off += RS * AS;
    r.r_xchg = off; off += 8;
    r.lds_bytes = off * 8;
    if (const char *pad = getenv("SBSIM_DEBUG_LDS_PAD")) r.lds_bytes += atoi(pad); // developer knob: fewer buildings per CU
    // workgroups per CU: LDS, and the registers (4 SIMDs x wavefronts per SIMD / wavefronts per building)
    const int by_regs = 4 * sweep_reg_waves_per_simd(NR, P) / (P == 2 ? 2 : 1);
    r.wg_per_cu = std::min(by_regs, kLdsCap / ((r.lds_bytes + kLdsGranule - 1) / kLdsGranule * kLdsGranule));
  };
  const int nl = nl_slots; // slots of A in LDS (the kernel keeps the rest in registers)
  if (P == 3) {
    layout(sweep_roll_a_stride(NR)); // 2 mod 4 doubles: 16-byte aligned rows, conflict-free ds_read_b128
  } else {
    layout(nl);
    const int plain = r.wg_per_cu;
    layout(nl | 1);
    if (r.wg_per_cu < plain) layout(nl);
  }
  if (r.wg_per_cu < 1) { r.why = "one building does not fit in LDS"; return; }

  const int pad = ncls;
  // mode 3: the sweep looks its four neighbour coefficients up by coefficient SET (classes that
  // differ only in ap / g share one): fewer distinct LDS addresses per wavefront read
  std::vector<int> set_of(ncls + 1, 0);
  if (P == 3) {
    for (int c = 0; c < ncls; ++c) {
      int found = -1;
      for (size_t k = 0; k < r.csetab.size() / 4 && found < 0; ++k)
        if (r.csetab[4 * k] == coef(c, 0) && r.csetab[4 * k + 1] == coef(c, 1) && r.csetab[4 * k + 2] == coef(c, 2) &&
            r.csetab[4 * k + 3] == coef(c, 3)) found = (int)k;
      if (found < 0) {
        found = (int)r.csetab.size() / 4;
        for (int j = 0; j < 4; ++j) r.csetab.push_back(coef(c, j));
      }
      set_of[c] = found;
    }
    set_of[pad] = (int)r.csetab.size() / 4; // the pad set: no neighbour counts
    for (int j = 0; j < 4; ++j) r.csetab.push_back(0.0);
    if (r.csetab.size() / 4 > 32) { r.why = "more than 32 distinct coefficient sets"; r.ok = false; return; }
  }
  auto cell_class = [&](int R, int col) { // trimmed coordinates
    return (R >= 0 && R < Hs && col >= 0 && col < Ws) ? cls_at(x0 + R, y0 + col) : pad;
  };
  const int aslots = (NR + 7) / 8, zslots = P == 3 ? NR / 8 : (NR + 3) / 4;
  const int nw = P == 2 ? 2 : 1;
  r.cmapS.assign(P == 3 ? (size_t)(NR / 8) * 64 : (size_t)nw * (maxch + 3) * 64, 0);
  r.amapS.assign((size_t)nw * aslots * 64, 0);
  r.zmapS.assign((size_t)nw * zslots * 64, 0);
  r.tcls.assign((size_t)std::max(r.T, 1) * NR, (uint8_t)(8 * pad));
  r.tcset.assign((size_t)std::max(r.T, 1) * NR, (uint8_t)(8 * set_of[pad]));
  for (int t = 0; t < r.T; ++t)
    for (int c = 0; c < NR; ++c) {
      r.tcls[(size_t)t * NR + c] = (uint8_t)(8 * cell_class(64 + t, c));
      r.tcset[(size_t)t * NR + c] = (uint8_t)(8 * set_of[cell_class(64 + t, c)]);
    }
  if (P == 3 && r.T > 0) {
    // The tail scan (sweep_common.h): along a tail row x_c = bL_c x_{c-1} + q_c; lane l >= L0 composes its two
    // columns, six DPP levels scan the lanes.  The multiplicative half of every level does not depend on the
    // temperatures: it is run here, in affine_scan's own order (bit-identical), and the kernel reads what each
    // level multiplies by: `a` entering levels 1..5, 0 where the level has no source lane for the lane.
    const int n = NR / 2, L0 = 64 - n;
    r.tmul.assign((size_t)sweep_roll_tail_mul_doubles(NR, r.T), 0.0);
    auto src_lane = [](int level, int lane) { // the DPP source of affine_scan's levels: row_shr 1 / 2 / 4 / 8, row_bcast:15 (rows 1, 3), row_bcast:31 (rows 2, 3)
      const int row = lane >> 4, i = lane & 15;
      switch (level) {
        case 0: return i >= 1 ? lane - 1 : -1;
        case 1: return i >= 2 ? lane - 2 : -1;
        case 2: return i >= 4 ? lane - 4 : -1;
        case 3: return i >= 8 ? lane - 8 : -1;
        case 4: return (row & 1) ? (row << 4) - 1 : -1;
        default: return row >= 2 ? 31 : -1;
      }
    };
    for (int t = 0; t < r.T; ++t) {
      auto bL = [&](int c) { const int cl = cell_class(64 + t, c); return cl == pad ? 0.0 : coef(cl, 2); };
      double a[64], an[64];
      for (int lane = 0; lane < 64; ++lane) a[lane] = lane >= L0 ? bL(2 * (lane - L0) + 1) * bL(2 * (lane - L0)) : 0.0;
      for (int d = 0; d < 6; ++d) {
        for (int lane = 0; lane < 64; ++lane) {
          const int src = src_lane(d, lane), lp = lane - L0;
          const double A = src >= 0 ? a[lane] : 0.0;
          if (lp >= 0 && (d == 1 || d == 2)) r.tmul[(size_t)4 * n * t + 2 * lp + (d - 1)] = A;
          if (lp >= 0 && (d == 3 || d == 4)) r.tmul[(size_t)4 * n * t + 2 * n + 2 * lp + (d - 3)] = A;
          if (lp >= 0 && d == 5) r.tmul[(size_t)4 * n * r.T + 2 * lp + t] = A;
          an[lane] = a[lane] * (src >= 0 ? a[src] : 1.0);
        }
        std::memcpy(a, an, sizeof(a));
      }
    }
  }
  for (int w = 0; w < nw; ++w)
    for (int lane = 0; lane < 64; ++lane) {
      const int lp = lane - r.l0[w];
      const bool valid = lp >= 0 && lp < r.lw[w];
      const int R = r.rowbase[w] + lp;
      if (P == 3) {
        // step_roll.hip: one byte per step = the cell's coefficient set (its LDS byte offset is set * 32),
        // eight steps per word; at step s (of the ramp-up or of any period) the lane works on column
        // (s - lane) mod NR
        for (int ch = 0; ch < NR / 8; ++ch) {
          unsigned long long word = 0;
          for (int k = 0; k < 8; ++k) {
            const int col = ((8 * ch + k - lp) % NR + NR) % NR;
            const int c = valid ? cell_class(R, col) : pad;
            word |= (unsigned long long)set_of[c] << (8 * k);
          }
          r.cmapS[(size_t)ch * 64 + lane] = word;
        }
      } else
      for (int ch = 0; ch < maxch + 3; ++ch) {
        unsigned long long word = 0;
        for (int k = 0; k < 8; ++k) {
          int col = 8 * ch + k - lp;
          const int c = valid ? cell_class(R, col) : pad;
          word |= (unsigned long long)(c * cscale) << (8 * k); // stride 32: the byte offset into a table column
        }
        r.cmapS[((size_t)w * (maxch + 3) + ch) * 64 + lane] = word;
      }
      for (int g = 0; g < aslots; ++g) {
        unsigned long long word = 0;
        for (int k = 0; k < 8; ++k) {
          const int j = 8 * g + k;
          const int col = ((j - lp) % NR + NR) % NR;
          const int c = (valid && j < NR) ? cell_class(R, col) : pad;
          word |= (unsigned long long)(c * cscale) << (8 * k);
        }
        r.amapS[((size_t)w * aslots + g) * 64 + lane] = word;
      }
      if (P == 3) { // step_roll.hip: a byte per slot = zone * 8 (the byte offset inside the lane's row of the scratch), eight slots per word
        for (int g = 0; g < zslots; ++g) {
          unsigned long long word = 0;
          for (int k = 0; k < 8; ++k) {
            const int j = 8 * g + k;
            const int col = ((j - lp) % NR + NR) % NR;
            int z = Z; // dump column
            if (valid && R < Hs && col < Ws) { // (below 64 rows the lanes beyond the plan own pad rows)
              const int zz = zone_of[(x0 + R) * W + (y0 + col)];
              if (zz >= 0) z = zz;
            }
            word |= (unsigned long long)(z * 8) << (8 * k);
          }
          r.zmapS[(size_t)g * 64 + lane] = word;
        }
      } else
      for (int g = 0; g < zslots; ++g) {
        unsigned long long word = 0;
        for (int k = 0; k < 4; ++k) {
          const int j = 4 * g + k;
          const int col = ((j - lp) % NR + NR) % NR;
          int z = Z; // dump row
          if (valid && R < Hs && j < NR && col < Ws) { // (mode 3 below 64 rows: the lanes beyond the plan own pad rows)
            const int zz = zone_of[(x0 + R) * W + (y0 + col)];
            if (zz >= 0) z = zz;
          }
          const unsigned offb = (unsigned)((z * ZRS + (valid ? R : 0)) * 8);
          word |= (unsigned long long)offb << (16 * k);
        }
        r.zmapS[((size_t)w * zslots + g) * 64 + lane] = word;
      }
    }
  r.cell_state.assign(N, 0);
  int ring = 0;
  for (int x = 0; x < H; ++x)
    for (int y = 0; y < W; ++y) {
      if (x < x0 || x > x1 || y < y0 || y > y1) { r.cell_state[x * W + y] = -(++ring); continue; }
      const int R = x - x0, col = y - y0;
      if (P == 3 && R >= 64) { r.cell_state[x * W + y] = NR * 64 + (R - 64) * NR + col; continue; }
      const int w = (P == 2 && R >= r.lw[0]) ? 1 : 0;
      const int lp = R - r.rowbase[w];
      const int slot = (col + lp) % NR;
      if (P == 3) r.cell_state[x * W + y] = (slot / 2) * 128 + R * 2 + (slot & 1); // step_roll.hip: [NR / 2][64][2]
      else r.cell_state[x * W + y] = slot * RS + R;                                   // state layout [NR][RS]
    }
  r.ok = true;
}

struct LdsPlan { // launch geometry of the LDS-grid kernel
  int pitch, NL, S, nsteps, nbands, fast, ts, off_agtab, off_zscr, off_zmode, lds_wave_doubles;
  unsigned S_magic;
  size_t shared_bytes, wave_bytes;
  bool fits;
};

LdsPlan plan_lds(const sb_plan_desc *plan) {
  LdsPlan q{};
  const int H = plan->H, W = plan->W, Z = plan->Z;
  q.pitch = (W + 1) & ~1; // even pitch: skewed ds_read_b64 is bank-conflict free
  q.NL = H * q.pitch;
  q.nbands = (H + 63) / 64;
  // band stride: >= kChunk idle positions between a lane's rows, and with several bands the
  // seam row must be written >= 17 steps before lane 0 prefetches it (S - 63 > 2*kChunk)
  q.S = std::max(W + kChunk, q.nbands > 1 ? 64 + 2 * kChunk : 64);
  q.S_magic = (unsigned)((1ull << 32) / (unsigned)q.S) + 1u;
  const int rows_last = H - (q.nbands - 1) * 64;
  q.nsteps = (q.nbands - 1) * q.S + rows_last + W - 1;
  q.fast = (q.nbands == 1) || (q.S - 63 > 2 * kChunk);
  int off = ((q.NL + 1) & ~1) + 2 * kGuardHost;
  q.ts = plan->n_classes <= 32 ? 32 : (plan->n_classes <= 128 ? 128 : 256);
  q.off_agtab = off; off += q.ts;
  q.off_zscr = off; off += (3 * Z + 1) & ~1;
  q.off_zmode = off; off += ((Z + 1) / 2 + 1) & ~1;
  q.lds_wave_doubles = off;
  q.shared_bytes = (size_t)5 * q.ts * 8 + (size_t)((Z + 2) >> 1) * 8;
  q.wave_bytes = (size_t)off * 8;
  q.fits = q.shared_bytes + q.wave_bytes <= (size_t)kLdsCap && q.NL < 65535;
  return q;
}

int lds_buildings_per_cu(const LdsPlan &q) {
  return q.fits ? (int)(((size_t)kLdsCap - q.shared_bytes) / q.wave_bytes) : 0;
}

int check_plan(const sb_plan_desc *plan) {
  if (!plan) return fail(SB_ERR_INVALID, "null floor plan");
  if (plan->H < 1 || plan->W < 1 || plan->Z < 0 || plan->n_classes < 1 || plan->n_classes > 255)
    return fail(SB_ERR_INVALID, "sb_create: bad floor-plan dimensions");
  if (!plan->cell_class || !plan->class_coef || !plan->class_zone || !plan->zone_off ||
      (plan->zone_off[plan->Z] > 0 && !plan->zone_cells))
    return fail(SB_ERR_INVALID, "sb_create: null floor-plan table");
  for (int z = 0; z < plan->Z; ++z)
    if (plan->zone_off[z + 1] < plan->zone_off[z] || plan->zone_off[0] != 0)
      return fail(SB_ERR_INVALID, "sb_create: zone offsets must start at 0 and be non-decreasing");
  const int N = plan->H * plan->W;
  for (int i = 0; i < plan->zone_off[plan->Z]; ++i)
    if (plan->zone_cells[i] < 0 || plan->zone_cells[i] >= N)
      return fail(SB_ERR_INVALID, "sb_create: zone cell out of range");
  for (int i = 0; i < N; ++i)
    if (plan->cell_class[i] >= plan->n_classes) return fail(SB_ERR_INVALID, "sb_create: cell class out of range");
  for (int c = 0; c < plan->n_classes; ++c)
    if (plan->class_zone[c] >= plan->Z) return fail(SB_ERR_INVALID, "sb_create: class zone out of range");
  return SB_OK;
}

void fill_launch_info(const sb_plan_desc *plan, const RegPlan &r, const LdsPlan &q, int n_obs, int cus,
                      int n_buildings, sb_launch_info *out, int n_actions = SB_NUM_ACTIONS) {
  const int64_t N = (int64_t)plan->H * plan->W, Z = plan->Z;
  out->algorithmic_bytes_per_env_step = 8ll * N + 24ll * Z + 4ll * n_actions + 4ll * n_obs + 44;
  const int64_t rest = (8 * 4 + 4 * 2) * Z + 16ll * kNScalOut + 4ll * n_actions + 4ll * n_obs + 4;
  if (r.ok && r.P == 6) { // step_stream.hip: the grid in global memory
    out->path = 2;
    out->waves_per_building = out->waves_per_workgroup = r.lw[1];
    out->kernel = r.P;
    out->workgroups = std::max(1, std::min(n_buildings, cus * r.wg_per_cu));
    out->lds_bytes_per_workgroup = r.lds_bytes;
    out->sweep_steps = r.steps;
    out->state_bytes_per_env_step = 16ll * r.state_doubles + rest;
  } else if (r.ok) {
    out->path = 1;
    out->waves_per_building = r.P == 5 ? r.RS / 64 : r.P == 2 ? 2 : 1;
    out->kernel = r.P; // sb_sweep_kernel: modes 1..5 are SB_KERNEL_REG .. SB_KERNEL_BAND
    out->waves_per_workgroup = r.P == 5 ? r.RS / 64 : r.P == 2 ? 2 : r.waves_per_wg; // mode 3: four buildings per workgroup
    out->workgroups = std::max(1, std::min((n_buildings + r.waves_per_wg - 1) / r.waves_per_wg, cus * r.wg_per_cu));
    out->lds_bytes_per_workgroup = r.lds_bytes;
    out->sweep_steps = r.steps;
    out->state_bytes_per_env_step = 16ll * r.state_doubles + rest;
  } else {
    int wpw = (int)(((size_t)kLdsCap - q.shared_bytes) / q.wave_bytes);
    wpw = std::max(1, std::min(wpw, 16));
    int wg_per_cu = 1;
    if (wpw > 4) { wg_per_cu = std::min(8, wpw / 4); wpw = 4; } // several smaller workgroups once 4 waves fit
    out->path = 0;
    out->waves_per_building = 1;
    out->kernel = SB_KERNEL_LDS;
    out->waves_per_workgroup = wpw;
    out->workgroups = std::max(1, std::min((n_buildings + wpw - 1) / wpw, cus * wg_per_cu));
    out->lds_bytes_per_workgroup = (int32_t)(q.shared_bytes + q.wave_bytes * wpw);
    out->sweep_steps = q.nsteps;
    out->state_bytes_per_env_step = 16ll * N + rest;
  }
}

// Which sweep kernel owns this floor plan: a register kernel (plan_reg), else the LDS-grid kernel, else --
// plans that fit neither -- the streaming kernel.  SBSIM_FORCE_STREAM_PATH=1: the streaming kernel (tests).
int choose_kernel(const sb_plan_desc *plan, const LdsPlan &q, RegPlan &r) {
  std::string why;
  if (env_flag("SBSIM_FORCE_STREAM_PATH")) {
    if (!plan_stream(plan, r, why)) return fail(SB_ERR_TOO_LARGE, "the streaming sweep kernel cannot hold this floor plan: " + why);
    return SB_OK;
  }
  if (!env_flag("SBSIM_FORCE_LDS_PATH")) plan_reg(plan, lds_buildings_per_cu(q), r);
  if (r.ok || q.fits) return SB_OK;
  r = RegPlan();
  if (!plan_stream(plan, r, why))
    return fail(SB_ERR_TOO_LARGE, "no sweep kernel holds this floor plan (the streaming kernel takes up to 1,024 rows in one "
                                  "orientation and rows / 64 x columns <= ~9,000): " + why);
  return SB_OK;
}

} // namespace

extern "C" {

int sb_abi_version(void) { return SB_ABI_VERSION; }
int sb_has_experimental_kernels(void) { return kExperimental ? 1 : 0; }
const char *sb_last_error(void) { return sb::host::g_err.c_str(); }

int sb_plan_info(const sb_plan_desc *plan, int32_t n_obs, int32_t n_buildings, sb_launch_info *out) {
  if (!out) return fail(SB_ERR_INVALID, "sb_plan_info: null argument");
  int rc = check_plan(plan);
  if (rc != SB_OK) return rc;
  RegPlan r;
  const LdsPlan q = plan_lds(plan);
  rc = choose_kernel(plan, q, r);
  if (rc != SB_OK) return rc;
  fill_launch_info(plan, r, q, n_obs, 256, std::max(n_buildings, 1), out);
  return SB_OK;
}

int sb_create(const sb_plan_desc *plan, const sb_params *params, const sb_obs_layout *obs,
              int32_t n_buildings, int32_t device, sb_handle **out) {
  if (!plan || !params || !obs || !out) return fail(SB_ERR_INVALID, "sb_create: null argument");
  int rc = check_plan(plan);
  if (rc != SB_OK) return rc;
  if (n_buildings < 1) return fail(SB_ERR_INVALID, "sb_create: n_buildings must be >= 1");
  if (params->dt <= 0 || params->iter_limit < 1)
    return fail(SB_ERR_INVALID, "sb_create: time_step_sec and iteration_limit must be positive");
  if (params->ahu_cool_sp <= params->ahu_heat_sp) // air_handler.py:60-64
    return fail(SB_ERR_INVALID, "cooling_air_temp_setpoint must greater than heating_air_temp_setpoint");
  if (params->n_actions < 1 || params->n_actions > 3 + plan->Z)
    return fail(SB_ERR_INVALID, "sb_create: n_actions must be in 1 .. 3 + zones (one column per settable field)");
  if (!params->act_kind || !params->act_zone || !params->act_lo || !params->act_hi)
    return fail(SB_ERR_INVALID, "sb_create: null action table (sb_params.act_kind / act_zone / act_lo / act_hi)");
  std::vector<int> zone_act((size_t)std::max(plan->Z, 1), -1);
  int n_damper_actions = 0;
  {
    int seen[3] = {0, 0, 0};
    for (int i = 0; i < params->n_actions; ++i) {
      const int k = params->act_kind[i];
      if (k < SB_ACT_BOILER_SUPPLY_WATER_SETPOINT || k > SB_ACT_VAV_SUPPLY_AIR_DAMPER_COMMAND)
        return fail(SB_ERR_INVALID, "sb_create: unknown action kind");
      if (k == SB_ACT_VAV_SUPPLY_AIR_DAMPER_COMMAND) {
        const int z = params->act_zone[i];
        if (z < 0 || z >= plan->Z) return fail(SB_ERR_INVALID, "sb_create: VAV action for a zone the floor plan does not have");
        // environment.py:591-653 keys its action normalisers by (device, setpoint): a field has one column
        if (zone_act[z] >= 0) return fail(SB_ERR_INVALID, "sb_create: two action columns drive the same VAV damper");
        zone_act[z] = i;
        ++n_damper_actions;
      } else if (seen[k]++) {
        return fail(SB_ERR_INVALID, "sb_create: two action columns drive the same setpoint");
      }
    }
  }
  if (obs->n_obs < 1) return fail(SB_ERR_INVALID, "sb_create: empty observation layout");
  if (!obs->mean || !obs->sigma || (plan->Z > 0 && !obs->col_zone))
    return fail(SB_ERR_INVALID, "sb_create: null observation-layout table");
  { // every device field is written unguarded by write_obs: its index must be inside the row / the source table
    const int n_fields = obs->n_src ? obs->n_src : obs->n_obs, n_ahu = params->ahu_has_weather ? 9 : 8;
    bool ok = obs->col_ahu >= 0 && obs->col_ahu + n_ahu <= n_fields && obs->col_boiler >= 0 &&
              obs->col_boiler + 3 <= n_fields && obs->col_aux >= 0 && obs->col_aux + SB_NUM_AUX <= obs->n_obs;
    for (int z = 0; z < plan->Z; ++z) ok = ok && obs->col_zone[z] >= 0 && obs->col_zone[z] + 3 <= n_fields;
    if (!ok) return fail(SB_ERR_INVALID, "sb_create: observation columns outside the observation row");
  }
  for (int z = 0; z < plan->Z; ++z) // building.py:579-587: the mean over a zone's air CVs
    if (plan->zone_off[z + 1] == plan->zone_off[z]) return fail(SB_ERR_INVALID, "sb_create: a zone without cells");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
    return fail(SB_ERR_NO_DEVICE, "sb_create: no HIP device visible (this library has no CPU path)");
  if (device < 0 || device >= ndev) return fail(SB_ERR_INVALID, "sb_create: bad device ordinal");
  SB_ON_DEVICE(device);
  hipDeviceProp_t prop;
  SB_HIP(hipGetDeviceProperties(&prop, device));

  RegPlan r;
  const LdsPlan q = plan_lds(plan);
  rc = choose_kernel(plan, q, r);
  if (rc != SB_OK) return rc;

  auto h = new sb_handle();
  h->device = device;
  h->cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  Dev &d = h->d;
  d.B = n_buildings; d.H = plan->H; d.W = plan->W; d.Z = plan->Z;
  d.N = plan->H * plan->W;
  d.Np = ((d.N + 1) & ~1) + 2 * kPad;
  d.ncls = plan->n_classes;
  d.p = *params;
  d.p.act_kind = d.p.act_zone = nullptr; d.p.act_lo = d.p.act_hi = nullptr; // (host pointers: the device reads Dev::act_*)
  d.reg = r.ok ? 1 : 0;
  fill_launch_info(plan, r, q, obs->n_obs, h->cus, n_buildings, &h->info, params->n_actions);

#define SB_TRY(x) do { rc = (x); if (rc != SB_OK) { delete h; return rc; } } while (0)
  { // one extra row: the register path's pad class (T' = Tprev: ap = 1, everything else 0)
    std::vector<double> ct((size_t)(d.ncls + 1) * 8, 0.0);
    std::memcpy(ct.data(), plan->class_coef, (size_t)d.ncls * 8 * sizeof(double));
    ct[(size_t)d.ncls * 8 + 4] = 1.0;
    SB_TRY(upload(h->ctab, ct.data(), ct.size()));
  }
  SB_TRY(upload(h->czone, plan->class_zone, (size_t)d.ncls));
  SB_TRY(upload(h->zone_off, plan->zone_off, (size_t)d.Z + 1));
  h->h_zone_off.assign(plan->zone_off, plan->zone_off + d.Z + 1);
  h->h_zone_cells.assign(plan->zone_cells, plan->zone_cells + plan->zone_off[plan->Z]);
  h->h_state_index.resize((size_t)d.N);
  for (int g = 0; g < d.N; ++g) h->h_state_index[g] = d.reg ? r.cell_state[g] : kPad + g;
  if (d.reg) {
    d.pitch = d.W; d.NL = d.N; d.ts = r.ts;
    d.NR = r.NR; d.P = r.P; d.RS = r.RS; d.Ws = r.Ws; d.n_ring = r.n_ring; d.n_ring_f64 = (double)r.n_ring;
    d.T = r.T; d.state_doubles = r.state_doubles; d.AS = r.AS; d.ZRS = r.ZRS ? r.ZRS : (r.RS | 1);
    for (int w = 0; w < 2; ++w) { d.l0[w] = r.l0[w]; d.rowbase[w] = r.rowbase[w]; d.nch[w] = r.nch[w]; }
    for (int w = 0; w < 4; ++w) d.lw[w] = r.lw[w];
    d.lag = r.lag; d.nslots = r.nslots; d.nsteps = r.steps;
    d.lds_reg_bytes = r.lds_bytes; d.wg_per_cu = r.wg_per_cu;
    d.r_seam = r.r_seam; d.r_A = r.r_A; d.r_xchg = r.r_xchg;
    if (r.P == 3) { d.r_cmap = r.r_cmap; d.lds_wave_doubles = r.wave_doubles; }
    d.pred_haste = 1.0f; d.pred_slack = 1.0f; // measured (tools/bench_two_rows.py); developer knobs: speed only,
    if (const char *e = getenv("SBSIM_DEBUG_PRED_HASTE")) d.pred_haste = (float)atof(e); // never the result
    if (const char *e = getenv("SBSIM_DEBUG_PRED_SLACK")) d.pred_slack = (float)atof(e);
    // measure-free rolling periods (step_two_impl.h, Hist): only when max|delta| provably never grows from one sweep to
    // the next -- every neighbour coefficient >= 0 and every class's four summing to <= 1 (then ||(I - L)^-1 U||_inf <= 1)
    d.two_skip = 0; d.skip_kappa = 0.9f;
    if (r.P == 4 && !env_flag("SBSIM_TWO_NO_SKIP")) {
      bool mono = true;
      for (int c = 0; c < plan->n_classes; ++c) {
        double sum = 0.0;
        for (int j = 0; j < 4; ++j) { mono = mono && plan->class_coef[c * 8 + j] >= 0.0; sum += plan->class_coef[c * 8 + j]; }
        mono = mono && sum <= 1.0;
      }
      d.two_skip = mono ? 1 : 0;
    }
    if (const char *e = getenv("SBSIM_DEBUG_SKIP_KAPPA")) d.skip_kappa = (float)atof(e); // speed only
    d.pred_first = r.P == 5 ? 2 : 3; // step_two.hip: periods a first block rolls unseen (+ 1); step_band.hip: its first block aims at the previous step's count - (pred_first - 2) (measured: 2 beats 3 and 4 by 2-4 %; 1: it never rolls unseen)
    if (const char *e = getenv("SBSIM_DEBUG_PRED_FIRST")) d.pred_first = std::max(1, atoi(e));
    SB_TRY(upload(h->zone_cells_l, plan->zone_cells, (size_t)plan->zone_off[plan->Z]));
    SB_TRY(upload(h->cmapS, r.cmapS.data(), r.cmapS.size()));
    SB_TRY(upload(h->amapS, r.amapS.data(), r.amapS.size()));
    SB_TRY(upload(h->zmapS, r.zmapS.data(), r.zmapS.size()));
    SB_TRY(upload(h->cell_state, r.cell_state.data(), r.cell_state.size()));
    SB_TRY(alloc_zero(h->ring, (size_t)d.B * std::max(r.n_ring, 1)));
    SB_TRY(upload(h->tcls, r.tcls.data(), r.tcls.size()));
    SB_TRY(upload(h->tcset, r.tcset.data(), r.tcset.size()));
    SB_TRY(upload(h->csetab, r.csetab.data(), r.csetab.size()));
    SB_TRY(upload(h->tmul, r.tmul.data(), r.tmul.size()));
    d.tmulS = h->tmul.p; d.tmul_doubles = (int)r.tmul.size();
    d.tcset = h->tcset.p; d.csetab = h->csetab.p; d.ncset = (int)r.csetab.size() / 4;
    d.csetab_doubles = (int)r.csetab.size();
    d.two_sym = r.two_sym; d.two_level = r.two_level; d.tail_set_base = r.tail_set_base;
    d.tail_pad_set = r.P == 4 ? r.tail_pad_set : (d.ncset - 1) * 32;
    d.zs_off = nullptr;
    if (!r.zs_off.empty()) {
      SB_TRY(upload(h->zs_off, r.zs_off.data(), r.zs_off.size()));
      d.zs_off = h->zs_off.p;
    }
    SB_TRY(alloc_zero(h->temp, (size_t)d.B * d.state_doubles));
    if (d.P == 6) SB_TRY(alloc_zero(h->abuf, (size_t)h->info.workgroups * d.state_doubles)); // A = ap*Tprev + g, one grid per resident workgroup
    d.stream_ms = d.P == 6 ? r.stream_ms : 0;
    if (d.stream_ms) SB_TRY(alloc_zero(h->ebuf, (size_t)h->info.workgroups * d.state_doubles)); // the pass's other grid
    d.two_abuf = nullptr;
    if (d.P == 4) { // step_two.hip: the slots of A that stream from L2, one strip per resident workgroup
      SB_TRY(alloc_zero(h->abuf, (size_t)h->info.workgroups * (size_t)std::max(1, d.NR - sweep_two_lds_slots(d.NR, d.two_level)) * 128));
      d.two_abuf = h->abuf.p;
    }
    d.tcls = h->tcls.p;
    d.cmapS = h->cmapS.p; d.amapS = h->amapS.p; d.zmapS = h->zmapS.p; d.cell_state = h->cell_state.p;
    d.ring = h->ring.p;
  } else {
    d.pitch = q.pitch; d.NL = q.NL; d.S = q.S; d.S_magic = q.S_magic; d.nsteps = q.nsteps;
    d.nbands = q.nbands; d.fast = q.fast; d.ts = q.ts;
    if (env_flag("SBSIM_FORCE_GENERIC_SWEEP")) d.fast = 0;
    d.off_agtab = q.off_agtab; d.off_zscr = q.off_zscr; d.off_zmode = q.off_zmode;
    d.lds_wave_doubles = q.lds_wave_doubles;
    std::vector<int> zl((size_t)plan->zone_off[plan->Z]);
    for (size_t i = 0; i < zl.size(); ++i) {
      const int g = plan->zone_cells[i];
      zl[i] = (g / d.W) * d.pitch + g % d.W;
    }
    {
      std::vector<uint8_t> padded((size_t)d.N + 2 * kPad + 8, 0);
      std::memcpy(padded.data() + kPad, plan->cell_class, (size_t)d.N);
      SB_TRY(upload(h->cls, padded.data(), padded.size()));
    }
    SB_TRY(upload(h->zone_cells_l, zl.data(), zl.size()));
    { // zone-sum blocks: u16 LDS indices, zones padded to 512 with NL (zero guard cell)
      std::vector<uint16_t> idx;
      std::vector<int> bz;
      for (int z = 0; z < d.Z; ++z) {
        const int c0 = plan->zone_off[z], c1 = plan->zone_off[z + 1];
        std::vector<uint16_t> cells;
        for (int i = c0; i < c1; ++i) cells.push_back((uint16_t)zl[i]);
        while (cells.size() % 512) cells.push_back((uint16_t)d.NL);
        // within a block, element t of lane l is cell t*64+l: one gather instruction touches
        // 64 consecutive cells of the (sorted) zone list -> consecutive LDS addresses, no
        // bank conflicts
        for (size_t b0 = 0; b0 < cells.size(); b0 += 512) {
          for (int l = 0; l < 64; ++l)
            for (int t = 0; t < 8; ++t) idx.push_back(cells[b0 + (size_t)t * 64 + l]);
          bz.push_back(z);
        }
      }
      if (idx.empty()) { idx.assign(512, (uint16_t)d.NL); bz.push_back(0); }
      d.n_zblocks = (int)bz.size();
      SB_TRY(upload(h->zl16, (const uint4 *)idx.data(), idx.size() / 8));
      SB_TRY(upload(h->zblk_zone, bz.data(), bz.size()));
    }
    { // fast-sweep schedule (see stage_g): where every lane is during every 8-step chunk
      const int nch = (d.nsteps + kChunk - 1) / kChunk;
      const int nct = ((nch + 1) & ~1) + 6; // the pipeline prefetches up to 5 chunks past the end
      std::vector<int4> sc((size_t)nct * 64);
      std::vector<unsigned long long> mk((size_t)nct * kChunk, 0ull);
      for (int ch = 0; ch < nct; ++ch)
        for (int lane = 0; lane < 64; ++lane) {
          const int v0 = ch * kChunk - lane, vl = v0 + kChunk - 1;
          const int band = vl >= 0 ? vl / d.S : 0;
          const int y0 = v0 - band * d.S;
          const int rr = band * 64 + lane;
          const bool row_ok = vl >= 0 && rr < d.H;
          const int ra = std::min(rr, d.H - 1);
          const int y0a = std::min(std::max(vl >= 0 ? y0 : v0, -kChunk), d.W);
          int4 e;
          e.x = ra * d.pitch + y0a;
          e.y = ra * d.W + y0a;
          e.z = std::min(ra + 1, d.H - 1) * d.pitch + y0a;
          e.w = std::max(ra - 1, 0) * d.pitch + y0a;
          sc[(size_t)ch * 64 + lane] = e;
          for (int k = 0; k < kChunk; ++k)
            if (row_ok && y0 + k >= 0 && y0 + k < d.W) mk[(size_t)ch * kChunk + k] |= 1ull << lane;
        }
      SB_TRY(upload(h->sched, sc.data(), sc.size()));
      SB_TRY(upload(h->smask, mk.data(), mk.size()));
    }
    SB_TRY(alloc_zero(h->temp, (size_t)d.B * d.Np));
    d.cls = h->cls.p; d.zl16 = h->zl16.p; d.zblk_zone = h->zblk_zone.p; d.sched = h->sched.p;
    d.smask = h->smask.p;
  }
  SB_TRY(upload(h->act_kind, params->act_kind, (size_t)params->n_actions));
  SB_TRY(upload(h->act_zone, params->act_zone, (size_t)params->n_actions));
  SB_TRY(upload(h->act_lo, params->act_lo, (size_t)params->n_actions));
  SB_TRY(upload(h->act_hi, params->act_hi, (size_t)params->n_actions));
  SB_TRY(upload(h->zone_act, zone_act.data(), zone_act.size()));
  d.act_kind = h->act_kind.p; d.act_zone = h->act_zone.p; d.act_lo = h->act_lo.p; d.act_hi = h->act_hi.p;
  d.zone_act = h->zone_act.p; d.n_damper_actions = n_damper_actions;
  SB_TRY(upload(h->col_zone, obs->col_zone, (size_t)d.Z));
  const size_t n_norm = (size_t)std::max(obs->n_obs, obs->n_src);
  SB_TRY(upload(h->obs_mean, obs->mean, n_norm));
  SB_TRY(upload(h->obs_sigma, obs->sigma, n_norm));
  d.n_src = obs->n_src; d.n_hist = obs->n_src ? obs->n_hist : 0; d.hist_normalize = obs->hist_normalize;
  if (obs->n_src) { // optional HistogramReducer
    const int n_dev_fields = 3 * d.Z + (params->ahu_has_weather ? 9 : 8) + 3;
    if (obs->n_src != n_dev_fields || !obs->src_dest || obs->n_hist < 0 ||
        (obs->n_hist > 0 && (!obs->hist_col || !obs->hist_off || !obs->hist_bins))) {
      delete h;
      return fail(SB_ERR_INVALID, "sb_create: inconsistent histogram-reducer layout");
    }
    for (int i = 0; i < obs->n_src; ++i)
      if (obs->src_dest[i] >= obs->n_obs || obs->src_dest[i] < -obs->n_hist) {
        delete h;
        return fail(SB_ERR_INVALID, "sb_create: histogram-reducer destination out of range");
      }
    for (int k = 0; k < obs->n_hist; ++k) {
      const int n = obs->hist_off[k + 1] - obs->hist_off[k];
      if (n < 1 || obs->hist_col[k] < 0 || obs->hist_col[k] + n > obs->n_obs) {
        delete h;
        return fail(SB_ERR_INVALID, "sb_create: histogram columns out of range");
      }
    }
    SB_TRY(upload(h->src_dest, obs->src_dest, (size_t)obs->n_src));
    SB_TRY(upload(h->hist_col, obs->hist_col, (size_t)obs->n_hist));
    SB_TRY(upload(h->hist_off, obs->hist_off, (size_t)obs->n_hist + 1));
    SB_TRY(upload(h->hist_bins, obs->hist_bins, (size_t)(obs->n_hist ? obs->hist_off[obs->n_hist] : 0)));
    d.src_dest = h->src_dest.p; d.hist_col = h->hist_col.p; d.hist_off = h->hist_off.p;
    d.hist_bins = h->hist_bins.p;
  }
  SB_TRY(alloc_zero(h->zmean, (size_t)d.B * d.Z));
  SB_TRY(alloc_zero(h->zair, (size_t)d.B * d.Z));
  SB_TRY(alloc_zero(h->damper, (size_t)d.B * d.Z));
  SB_TRY(alloc_zero(h->qz, (size_t)d.B * d.Z));
  SB_TRY(alloc_zero(h->mode, (size_t)d.B * d.Z)); // thermostat.py:66-69: Mode.OFF
  SB_TRY(alloc_zero(h->scal, (size_t)d.B * kNScal));
  SB_TRY(alloc_zero(h->bld, (size_t)d.B));
  SB_TRY(alloc_zero(h->gtabg, (size_t)d.B * d.ts));
  SB_TRY(alloc_zero(h->zsum, (size_t)d.B * d.Z));
  SB_TRY(alloc_zero(h->gsum, (size_t)d.B));
  SB_TRY(alloc_zero(h->nsw, (size_t)d.B));
  SB_TRY(alloc_zero(h->next_b, 1));
  d.redo_ctr = nullptr; d.redo_list = nullptr; d.redo_scratch = nullptr; d.redo_mode = 0; d.roll_exact = 0; d.dbg_redo_mod = 0;
  if (d.reg && d.P == 3) { // step_roll.hip: the list of buildings the fast kernel leaves to the exact one
    SB_TRY(alloc_zero(h->redo_ctr, 2));
    SB_TRY(alloc_zero(h->redo_list, (size_t)d.B));
    SB_TRY(alloc_zero(h->redo_scratch, (size_t)h->info.workgroups * h->info.waves_per_workgroup * d.state_doubles));
    d.redo_ctr = h->redo_ctr.p; d.redo_list = h->redo_list.p; d.redo_scratch = h->redo_scratch.p;
    if (const char *e = getenv("SBSIM_ROLL_EXACT")) d.roll_exact = atoi(e) != 0;
    if (const char *e = getenv("SBSIM_DEBUG_FORCE_REDO")) d.dbg_redo_mod = std::max(0, atoi(e));
  }
#undef SB_TRY
  d.next_b = h->next_b.p;
  // buildings handed out statically before the draw counter: register path = workgroups, LDS-grid path and mode 3 = wavefronts
  d.sweep_wgs = (d.reg && d.P != 3) ? h->info.workgroups : h->info.workgroups * h->info.waves_per_workgroup;
  d.bld = h->bld.p; d.gtabg = h->gtabg.p; d.zsum = h->zsum.p; d.gsum = h->gsum.p; d.nsw = h->nsw.p;
  d.ctab = h->ctab.p; d.czone = h->czone.p; d.zone_off = h->zone_off.p;
  d.zone_cells_l = h->zone_cells_l.p; d.temp = h->temp.p; d.zmean = h->zmean.p;
  d.zair = h->zair.p; d.damper = h->damper.p; d.qz = h->qz.p; d.mode = h->mode.p;
  d.scal = h->scal.p;
  d.O = obs->n_obs; d.col_ahu = obs->col_ahu; d.col_blr = obs->col_boiler; d.col_aux = obs->col_aux;
  d.col_zone = h->col_zone.p; d.obs_mean = h->obs_mean.p; d.obs_sigma = h->obs_sigma.p;
  d.dbg = nullptr; d.dbg_timeline = 0;
  if (getenv("SBSIM_PHASE_TIMING")) { // developer aid: cycle stamps of wave 0's first building
    d.dbg_timeline = getenv("SBSIM_DEBUG_TIMELINE") ? 1 : 0; // + [2048]: step_band.hip's time line of one building-step (-DSB_PHASE_STAMPS builds)
    if (alloc_zero(h->dbg, 16 + (d.dbg_timeline ? 2048 : 0)) == SB_OK) d.dbg = h->dbg.p;
  }

  const int e = d.reg ? (d.P == 6 ? (d.stream_ms == 1 ? prepare_sweep_stream_ms(d, h->info.waves_per_workgroup) : d.stream_ms == 2 ? prepare_sweep_stream_roll(d, h->info.waves_per_workgroup) : prepare_sweep_stream(d, h->info.waves_per_workgroup)) : d.P == 5 ? prepare_sweep_band(d) : d.P == 4 ? prepare_sweep_two(d) : d.P == 3 ? prepare_sweep_roll(d) : prepare_sweep_reg(d))
                      : prepare_sweep_lds((size_t)h->info.lds_bytes_per_workgroup);
  if (e != (int)hipSuccess) {
    delete h;
    return fail(SB_ERR_HIP, std::string("hipFuncSetAttribute(MaxDynamicSharedMemorySize): ") +
                                hipGetErrorString((hipError_t)e));
  }
  rc = sb_reset(h, 0.0, nullptr, nullptr); // on the NULL stream ...
  if (rc == SB_OK && hipStreamSynchronize(nullptr) != hipSuccess) // ... and complete before a caller's own stream touches the state
    rc = fail(SB_ERR_HIP, "sb_create: the initial reset failed");
  if (rc != SB_OK) { delete h; return rc; }
  *out = h;
  return SB_OK;
}

void sb_destroy(sb_handle *h) {
  if (!h) return;
  DeviceGuard guard(h->device);
  delete h;
}

int sb_get_launch_info(const sb_handle *h, sb_launch_info *out) {
  if (!h || !out) return fail(SB_ERR_INVALID, "sb_get_launch_info: null argument");
  *out = h->info;
  return SB_OK;
}

int sb_reset(sb_handle *h, double initial_temp, const double *temps_dev, void *stream) {
  if (!h) return fail(SB_ERR_INVALID, "sb_reset: null handle");
  SB_ON_DEVICE(h->device);
  const int wpb = 4;
  const int blocks = std::min((h->d.B + wpb - 1) / wpb, 4096);
  hipLaunchKernelGGL(k_reset, dim3(blocks), dim3(64 * wpb), 0, (hipStream_t)stream, h->d, initial_temp,
                     temps_dev, h->was_reset ? 0 : 1, h->steps_since_reset, 0);
  h->was_reset = true;
  h->steps_since_reset = 0;
  h->counters_zeroed_on = (void *)(uintptr_t)1; // (whatever came before a reset is no promise about the draw counters)
  SB_HIP(hipGetLastError());
  return SB_OK;
}

int sb_set_temps(sb_handle *h, const double *temps_dev, void *stream) {
  if (!h || !temps_dev) return fail(SB_ERR_INVALID, "sb_set_temps: null argument");
  if (!h->was_reset) return fail(SB_ERR_INVALID, "sb_set_temps: sb_reset first");
  SB_ON_DEVICE(h->device);
  const int wpb = 4;
  const int blocks = std::min((h->d.B + wpb - 1) / wpb, 4096);
  hipLaunchKernelGGL(k_reset, dim3(blocks), dim3(64 * wpb), 0, (hipStream_t)stream, h->d, 0.0, temps_dev, 0, 0, 1);
  SB_HIP(hipGetLastError());
  return SB_OK;
}

int sb_observe_occupancy(sb_handle *h, const float aux[SB_NUM_AUX], double t_amb, const double *t_amb_dev,
                         const float *num_occupants_dev, double occupancy_norm, float *obs_dev, void *stream) {
  if (!h || !aux || !obs_dev) return fail(SB_ERR_INVALID, "sb_observe: null argument");
  SB_ON_DEVICE(h->device);
  const int blocks = std::max(1, std::min((h->d.B + 63) / 64, 4096));
  hipLaunchKernelGGL(k_observe, dim3(blocks), dim3(64), 0, (hipStream_t)stream, h->d, obs_dev, aux[0],
                     aux[1], aux[2], aux[3], aux[4], aux[5], aux[6], t_amb, t_amb_dev, num_occupants_dev,
                     occupancy_norm);
  SB_HIP(hipGetLastError());
  return SB_OK;
}

int sb_observe(sb_handle *h, const float aux[SB_NUM_AUX], double t_amb, const double *t_amb_dev,
               float *obs_dev, void *stream) {
  return sb_observe_occupancy(h, aux, t_amb, t_amb_dev, nullptr, 0.0, obs_dev, stream);
}

int sb_step_phases(sb_handle *h, const float *actions_dev, const sb_step_in *in, float *obs_dev,
                   float *reward_dev, float *info_dev, void *stream, int32_t phases) {
  if (!h || !in || !reward_dev) return fail(SB_ERR_INVALID, "sb_step: null argument");
  if (in->has_action && !actions_dev) return fail(SB_ERR_INVALID, "sb_step: has_action set but actions_dev is NULL");
  SB_ON_DEVICE(h->device);
  StepArgs s;
  s.actions = actions_dev; s.obs = obs_dev; s.reward = reward_dev; s.info = info_dev; s.in = *in;
  const Dev &d = h->d;
  const int blocks = std::max(1, std::min((d.B + kRowsPerBlock - 1) / kRowsPerBlock, h->cus * 32)); // a 16-lane row per building
  if (phases & SB_PHASE_PRE) {
    hipLaunchKernelGGL(k_pre, dim3(blocks), dim3(16 * kRowsPerBlock), 0, (hipStream_t)stream, d, s, -1);
    SB_HIP(hipGetLastError());
    h->counters_zeroed_on = stream; // k_pre zeroes the sweep kernel's draw counters
  }
  if (phases & SB_PHASE_SWEEP) {
    // without a k_pre in front of it ON THIS STREAM since the last sweep launch (phase by phase: bench.py brackets the
    // sweep with events, so k_pre came with the previous call), the counters are zeroed here -- two more dispatches
    if (h->counters_zeroed_on != stream) {
      SB_HIP(hipMemsetAsync(d.next_b, 0, sizeof(int), (hipStream_t)stream));
      if (d.redo_ctr) SB_HIP(hipMemsetAsync(d.redo_ctr, 0, 2 * sizeof(int), (hipStream_t)stream));
    }
    h->counters_zeroed_on = (void *)(uintptr_t)1; // (no stream)
    const int e = d.reg ? (d.P == 6   ? (d.stream_ms == 1 ? launch_sweep_stream_ms(d, h->abuf.p, h->ebuf.p, h->info.waves_per_workgroup, (hipStream_t)stream)
                                   : d.stream_ms == 2 ? launch_sweep_stream_roll(d, h->abuf.p, h->ebuf.p, h->info.waves_per_workgroup, (hipStream_t)stream)
                                                     : launch_sweep_stream(d, h->abuf.p, h->info.waves_per_workgroup, (hipStream_t)stream))
                       : d.P == 5 ? launch_sweep_band(d, (hipStream_t)stream)
                       : d.P == 4 ? launch_sweep_two(d, (hipStream_t)stream)
                       : d.P == 3 ? launch_sweep_roll(d, (hipStream_t)stream)
                                  : launch_sweep_reg(d, h->cus, (hipStream_t)stream))
                        : launch_sweep_lds(d, h->info.workgroups, h->info.waves_per_workgroup,
                                           (size_t)h->info.lds_bytes_per_workgroup, (hipStream_t)stream);
    if (e != (int)hipSuccess)
      return fail(SB_ERR_HIP, std::string("sweep kernel launch: ") + hipGetErrorString((hipError_t)e));
  }
  if (phases & SB_PHASE_POST) {
    if (h->conv_attached) { // simulator_flexible_floor_plan.py:156: after the FD update; the zone sums of
      const int rc = sb_launch_convection(h, (hipStream_t)stream); // k_post do not change (values move inside rooms)
      if (rc != SB_OK) return rc;
    }
    hipLaunchKernelGGL(k_post, dim3(std::max(1, std::min((d.B + 63) / 64, h->cus * 16))), dim3(64), 0, (hipStream_t)stream, d, s, -1);
    ++h->steps_since_reset;
    SB_HIP(hipGetLastError());
  }
  return SB_OK;
}

int sb_step(sb_handle *h, const float *actions_dev, const sb_step_in *in, float *obs_dev,
            float *reward_dev, float *info_dev, void *stream) {
  return sb_step_phases(h, actions_dev, in, obs_dev, reward_dev, info_dev, stream,
                        SB_PHASE_PRE | SB_PHASE_SWEEP | SB_PHASE_POST);
}

int sb_get_temps(sb_handle *h, double *out_dev, void *stream) {
  if (!h || !out_dev) return fail(SB_ERR_INVALID, "sb_get_temps: null argument");
  SB_ON_DEVICE(h->device);
  hipLaunchKernelGGL(k_copy_temps, dim3(2048), dim3(256), 0, (hipStream_t)stream, h->d, out_dev);
  SB_HIP(hipGetLastError());
  return SB_OK;
}

int sb_get_scalars(sb_handle *h, double *out_dev, void *stream) {
  if (!h || !out_dev) return fail(SB_ERR_INVALID, "sb_get_scalars: null argument");
  SB_ON_DEVICE(h->device);
  hipLaunchKernelGGL(k_copy_scalars, dim3(256), dim3(256), 0, (hipStream_t)stream, h->d, out_dev);
  SB_HIP(hipGetLastError());
  return SB_OK;
}

#define SB_COPY_OUT(name, src, count, type)                                                       \
  int name(sb_handle *h, type *out_dev, void *stream) {                                           \
    if (!h || !out_dev) return fail(SB_ERR_INVALID, #name ": null argument");                     \
    SB_ON_DEVICE(h->device);                                                              \
    SB_HIP(hipMemcpyAsync(out_dev, src, (size_t)(count) * sizeof(type), hipMemcpyDeviceToDevice,  \
                          (hipStream_t)stream));                                                  \
    return SB_OK;                                                                                 \
  }
SB_COPY_OUT(sb_get_zone_temps, h->d.zmean, (size_t)h->d.B *h->d.Z, double)
int sb_get_modes(sb_handle *h, int32_t *out_dev, void *stream) {
  if (!h || !out_dev) return fail(SB_ERR_INVALID, "sb_get_modes: null argument");
  SB_ON_DEVICE(h->device);
  hipLaunchKernelGGL(k_copy_modes, dim3(256), dim3(256), 0, (hipStream_t)stream, h->d, out_dev);
  SB_HIP(hipGetLastError());
  return SB_OK;
}
SB_COPY_OUT(sb_get_zone_power, h->d.qz, (size_t)h->d.B *h->d.Z, double)

/* ---- known-answer taps: k_pre / k_post on prescribed state of one building ---- */
int sb_tap_pre(sb_handle *h, int32_t building, const double *zone_temps, const int32_t *modes,
               const double *scalars, const float *actions, const sb_step_in *in, sb_tap_bld *bld,
               double *q_zone, double *damper, int32_t *modes_out) {
  if (!h || !zone_temps || !in) return fail(SB_ERR_INVALID, "sb_tap_pre: null argument");
  const Dev &d = h->d;
  if (building < 0 || building >= d.B) return fail(SB_ERR_INVALID, "sb_tap_pre: building out of range");
  SB_ON_DEVICE(h->device);
  h->counters_zeroed_on = (void *)(uintptr_t)1; // the tap's k_pre runs on one building: the next sweep launch zeroes its counters itself
  SB_HIP(hipDeviceSynchronize());
  const size_t zb = (size_t)building * d.Z;
  SB_HIP(hipMemcpy(d.zmean + zb, zone_temps, sizeof(double) * d.Z, hipMemcpyHostToDevice));
  if (modes) SB_HIP(hipMemcpy(d.mode + zb, modes, sizeof(int32_t) * d.Z, hipMemcpyHostToDevice));
  if (scalars) SB_HIP(hipMemcpy(d.scal + (size_t)building * kNScal, scalars, sizeof(double) * kNScalOut, hipMemcpyHostToDevice));
  DevBuf<float> act;
  if (actions) {
    std::vector<float> all((size_t)d.B * d.p.n_actions, 0.0f);
    for (int i = 0; i < d.p.n_actions; ++i) all[(size_t)building * d.p.n_actions + i] = actions[i];
    int rc = upload(act, all.data(), all.size());
    if (rc != SB_OK) return rc;
  } else if (in->has_action) {
    return fail(SB_ERR_INVALID, "sb_tap_pre: has_action without actions");
  }
  StepArgs s{};
  s.actions = act.p; s.in = *in;
  hipLaunchKernelGGL(k_pre, dim3(1), dim3(16 * kRowsPerBlock), 0, nullptr, d, s, building); // this building alone
  SB_HIP(hipGetLastError());
  SB_HIP(hipDeviceSynchronize());
  if (bld) {
    Bld v;
    SB_HIP(hipMemcpy(&v, d.bld + building, sizeof(Bld), hipMemcpyDeviceToHost));
    *bld = sb_tap_bld{v.t_now, v.t_next, v.heat_sp, v.cool_sp, v.blr_sp, v.t_sa, v.ahu_flow, v.blr_flow, v.blr_return,
                      v.ahu_count, v.blr_count, v.tank, v.tank_change, v.duration, v.rejected};
  }
  if (q_zone) SB_HIP(hipMemcpy(q_zone, d.qz + zb, sizeof(double) * d.Z, hipMemcpyDeviceToHost));
  if (damper) SB_HIP(hipMemcpy(damper, d.damper + zb, sizeof(double) * d.Z, hipMemcpyDeviceToHost));
  if (modes_out) {
    SB_HIP(hipMemcpy(modes_out, d.mode + zb, sizeof(int32_t) * d.Z, hipMemcpyDeviceToHost));
    for (int z = 0; z < d.Z; ++z) modes_out[z] &= kModeMask;
  }
  return SB_OK;
}

int sb_tap_post(sb_handle *h, int32_t building, const sb_tap_bld *bld, const double *zone_temps,
                double grid_mean, int32_t n_sweeps, const sb_step_in *in, float *reward, float *info) {
  if (!h || !bld || !zone_temps || !in || !reward) return fail(SB_ERR_INVALID, "sb_tap_post: null argument");
  const Dev &d = h->d;
  if (building < 0 || building >= d.B) return fail(SB_ERR_INVALID, "sb_tap_post: building out of range");
  SB_ON_DEVICE(h->device);
  SB_HIP(hipDeviceSynchronize());
  Bld v;
  SB_HIP(hipMemcpy(&v, d.bld + building, sizeof(Bld), hipMemcpyDeviceToHost)); // keeps the fields the tap does not expose
  v.t_now = bld->t_now; v.t_next = bld->t_next; v.heat_sp = bld->heat_sp; v.cool_sp = bld->cool_sp;
  v.blr_sp = bld->blr_sp; v.t_sa = bld->t_sa; v.ahu_flow = bld->ahu_flow; v.blr_flow = bld->blr_flow;
  v.blr_return = bld->blr_return; v.ahu_count = bld->ahu_count; v.blr_count = bld->blr_count;
  v.tank = bld->tank; v.tank_change = bld->tank_change; v.duration = bld->duration; v.rejected = bld->rejected;
  SB_HIP(hipMemcpy(d.bld + building, &v, sizeof(Bld), hipMemcpyHostToDevice));
  std::vector<double> zs(d.Z);
  for (int z = 0; z < d.Z; ++z) zs[z] = zone_temps[z] * (double)(h->h_zone_off[z + 1] - h->h_zone_off[z]);
  SB_HIP(hipMemcpy(d.zsum + (size_t)building * d.Z, zs.data(), sizeof(double) * d.Z, hipMemcpyHostToDevice));
  const double gs = grid_mean * (double)d.N;
  const int nsw = n_sweeps | (1 << 16);
  SB_HIP(hipMemcpy(d.gsum + building, &gs, sizeof(double), hipMemcpyHostToDevice));
  SB_HIP(hipMemcpy(d.nsw + building, &nsw, sizeof(int), hipMemcpyHostToDevice));
  DevBuf<float> rew, inf;
  int rc = alloc_zero(rew, (size_t)d.B);
  if (rc == SB_OK) rc = alloc_zero(inf, (size_t)d.B * SB_INFO_STRIDE);
  if (rc != SB_OK) return rc;
  StepArgs s{};
  s.reward = rew.p; s.info = inf.p; s.in = *in;
  hipLaunchKernelGGL(k_post, dim3(1), dim3(64), 0, nullptr, d, s, building); // this building alone
  SB_HIP(hipGetLastError());
  SB_HIP(hipDeviceSynchronize());
  SB_HIP(hipMemcpy(reward, rew.p + building, sizeof(float), hipMemcpyDeviceToHost));
  if (info) SB_HIP(hipMemcpy(info, inf.p + (size_t)building * SB_INFO_STRIDE, sizeof(float) * SB_INFO_STRIDE, hipMemcpyDeviceToHost));
  return SB_OK;
}

/* Developer aid (not part of the parity/bench path): 16 int64 cycle stamps, host pointer (16 + 2048 when the handle
   was created under SBSIM_DEBUG_TIMELINE=1). */
int sb_debug_phase_cycles(sb_handle *h, long long *out_host) {
  if (!h || !out_host) return fail(SB_ERR_INVALID, "sb_debug_phase_cycles: null argument");
  if (!h->d.dbg) return fail(SB_ERR_INVALID, "sb_debug_phase_cycles: set SBSIM_PHASE_TIMING=1 before sb_create");
  SB_ON_DEVICE(h->device);
  SB_HIP(hipDeviceSynchronize());
  SB_HIP(hipMemcpy(out_host, h->d.dbg, (16 + (h->d.dbg_timeline ? 2048 : 0)) * sizeof(long long), hipMemcpyDeviceToHost));
  return SB_OK;
}

} // extern "C"
