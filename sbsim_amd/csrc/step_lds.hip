// step_lds.hip -- the sweep kernel whose temperature grid lives in LDS (any floor-plan shape).
//
// One 64-lane wavefront owns one building instance for the whole step: the float64 grid
// lives in LDS, the Gauss-Seidel sweep walks it in a skewed (anti-diagonal) order that
// reproduces the reference's row-major in-place update exactly.
// step_reg.hip holds the faster register-resident variant for floor plans that fit it; this
// kernel is the general path (and the cross-check of the other one in the GPU tests).
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off (fma only where written).
#include "sb_device.h"

namespace sb {
namespace {

__device__ __forceinline__ int lds_index(const Dev &a, int gidx) {
  if (a.pitch == a.W) return gidx;
  int x = gidx / a.W;
  return x * a.pitch + (gidx - x * a.W);
}

// ---------------------------------------------------------------- the sweep
// Lane l owns rows l, l+64, ... ("bands"); its cells form one sequence of positions
// v = band*S + y and at step d it updates position v = d - l.  Cell (x,y) is therefore
// updated after (x-1,y) and (x,y-1) and before (x+1,y) and (x,y+1): the reference's
// row-major in-place order (simulator.py:302-314).  Every update is
//     T' = ap*Tprev + g + bU*U + bD*D + bL*L + bR*R
// (sbsim_amd/floorplan.py folds the corner / edge / interior / exterior formulas into
// per-class coefficients; missing neighbours have b = 0, so any finite value may stand in).
//
// LDS tables, structure-of-arrays so that every lookup is a ds_read_b64: tab[j*ts + c] for
// j = bU,bD,bL,bR,ap (shared by the workgroup) and gtab[c] (per wave: g depends on the
// building's ambient temperature and VAV power).  Measured on gfx950: gathering table rows
// with ds_read_b128 costs ~45 cycles per instruction here, ds_read_b64 ~6.

__device__ __forceinline__ int clampi(int x, int lo, int hi) { return min(max(x, lo), hi); }

// lane l <- lane l-1 (lane 0 keeps its own value) / lane l <- lane l+1 (lane 63 keeps).
__device__ __forceinline__ double wave_shr1(double x) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_update_dpp(lo, lo, 0x138, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(hi, hi, 0x138, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_shl1(double x) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_update_dpp(lo, lo, 0x130, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(hi, hi, 0x130, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}

// -- generic sweep: every neighbour through LDS; legal for any shape (fallback) ----------
struct Chunk {
  double P[kChunk];
  int li[kChunk]; // LDS index, -1 when the lane is idle at that step
  int c[kChunk];  // class id
};

__device__ __forceinline__ void load_chunk(const Dev &a, const double *__restrict__ Pg, int d0,
                                           int lane, Chunk &ch) {
#pragma unroll
  for (int k = 0; k < kChunk; ++k) {
    int v = d0 + k - lane;
    int band = (int)__umulhi((unsigned)max(v, 0), a.S_magic);
    int y = v - band * a.S;
    int r = band * 64 + lane;
    bool act = (v >= 0) && (y < a.W) && (r < a.H);
    int g = act ? r * a.W + y : 0;
    ch.li[k] = act ? (r * a.pitch + y) : -1;
    ch.P[k] = Pg[g];
    ch.c[k] = (int)a.cls[kPad + g];
  }
}

__device__ double sweep_generic(const Dev &a, double *E, const double *gtab, const double *tab,
                                const double *__restrict__ Pg, int lane) {
  double dmax = 0.0;
  const int pitch = a.pitch, last = a.NL - 1;
  Chunk cur, nxt;
  load_chunk(a, Pg, 0, lane, nxt);
  for (int d0 = 0; d0 < a.nsteps; d0 += kChunk) {
    cur = nxt;
    if (d0 + kChunk < a.nsteps) load_chunk(a, Pg, d0 + kChunk, lane, nxt);
#pragma unroll
    for (int k = 0; k < kChunk; ++k) {
      const int li = cur.li[k];
      if (li >= 0) {
        const double *bt = tab + cur.c[k];
        const int ts = a.ts;
        const double U = E[max(li - pitch, 0)], D = E[min(li + pitch, last)];
        const double L = E[max(li - 1, 0)], R = E[min(li + 1, last)];
        const double old = E[li];
        double nv = fma(bt[4 * ts], cur.P[k], gtab[cur.c[k]]);
        nv = fma(bt[ts], D, nv);
        nv = fma(bt[3 * ts], R, nv);
        nv = fma(bt[2 * ts], L, nv);
        nv = fma(bt[0], U, nv);
        dmax = fmax(dmax, fabs(nv - old));
        E[li] = nv;
      }
      __builtin_amdgcn_wave_barrier(); // keep step d's LDS store before step d+1's loads
    }
  }
  return wave_max(dmax);
}

// -- fast sweep ------------------------------------------------------------------------
// A single wavefront issues about one instruction every 5 cycles and only 3 buildings fit
// in a CU's LDS, so the sweep is instruction-issue bound: the loop below is written to
// need as few instructions per control volume as possible.
//   * L is the lane's previous result (register); U is the previous result of lane l-1,
//     one DPP wave_shr:1 move per 32-bit half.  Lane 0 has no source lane and keeps the
//     DPP `old` operand, which is pre-loaded with the band-seam value from LDS (MULTI).
//   * R (old value one column ahead in the lane's own row -- next step it is the cell's own
//     old value) and D (old value one row down) are plain ds_read_b64 with immediate
//     offsets from per-chunk base addresses; guard doubles around E keep every address
//     legal without clamping.
//   * Coefficients, ap*Tprev+g, R, D and the seam value are gathered one 8-step chunk
//     ahead into the slot that was just consumed; Tprev and the class bytes two chunks
//     ahead (three-stage software pipeline G -> L -> C): nothing in the loop waits on memory.
//   * Idle lanes run the same arithmetic on whatever (finite) values their addresses hold;
//     only the LDS store and the max-delta update are masked (per-slot lane masks in SGPRs).
// Requires S >= W + 8 (one row segment per chunk) and, when H > 64, S - 63 > 16 (the seam
// row is written at least 17 steps before lane 0 prefetches it).
constexpr int kGuard = 16; // finite guard doubles before and after E in LDS

struct StageG {
  double P[kChunk];
  unsigned long long cw; // 8 class bytes
  int li0;               // LDS index of position 0 of the chunk (may point before the row)
  int liD0, liU0;        // same position one row down / one row up (row clamped to the grid)
  int mlo, mhi;          // lane k holds the 64-bit lane mask of slot k ("is a real cell")
};
struct StageL {
  double A[kChunk], bU[kChunk], bD[kChunk], bL[kChunk], bR[kChunk], Rn[kChunk], Dn[kChunk], Ee[kChunk];
};

typedef double dbl2u __attribute__((ext_vector_type(2), aligned(8)));

// Stage S: where the lane is during chunk `ch` -- tables built once per floor plan
// (sb_create), so the loop spends no instructions on index arithmetic.  Loaded one chunk
// before stage G needs it (its addresses feed G's loads).
struct StageS {
  int4 sc;               // {li0, g0, liD0, liU0}
  unsigned long long mk; // slot k's lane mask, held by lane k
};
__device__ __forceinline__ void stage_s(const Dev &a, int ch, int lane, StageS &st) {
  st.sc = a.sched[ch * 64 + lane];
  st.mk = a.smask[ch * kChunk + (lane & (kChunk - 1))];
}

template <bool FIRST>
__device__ __forceinline__ void stage_g(const Dev &a, const double *E, const double *__restrict__ Pg,
                                        const StageS &st, StageG &g) {
  g.li0 = st.sc.x;
  g.liD0 = st.sc.z;
  g.liU0 = st.sc.w;
  g.mlo = (int)(unsigned)st.mk;
  g.mhi = (int)(unsigned)(st.mk >> 32);
  const int g0 = st.sc.y; // in [-8, N]: the grid is padded by kPad on both sides
  __builtin_memcpy(&g.cw, a.cls + kPad + g0, 8);
  if (FIRST) { // first sweep: the estimate still equals the previous temperatures
#pragma unroll
    for (int k = 0; k < kChunk; ++k) g.P[k] = E[g.li0 + k];
  } else {
    const dbl2u *src = (const dbl2u *)(Pg + g0);
#pragma unroll
    for (int k = 0; k < kChunk / 2; ++k) {
      dbl2u v = src[k];
      g.P[2 * k] = v.x;
      g.P[2 * k + 1] = v.y;
    }
  }
}

// slot k's 64-bit lane mask (held by lane k) -> a lane predicate without any VALU compare
template <int K>
__device__ __forceinline__ bool slot_active(int mlo, int mhi) {
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane(mlo, K);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane(mhi, K);
  return __builtin_amdgcn_inverse_ballot_w64(((unsigned long long)hi << 32) | lo);
}

// Gathers slot K of the NEXT chunk (described by g) into l.  R of a position is the old
// value of the lane's NEXT position; for the chunk's last slot that position belongs to the
// chunk after (base index li0_after), which may sit in another band row.
template <bool MULTI, int TS, int K>
__device__ __forceinline__ void load_slot(const double *E, const double *gtab, const double *tab,
                                          const StageG &g, int li0_after, StageL &l) {
  const int c = (int)((g.cw >> (8 * K)) & 0xffull);
  const double *bt = tab + c;
  l.bU[K] = bt[0]; l.bD[K] = bt[TS]; l.bL[K] = bt[2 * TS]; l.bR[K] = bt[3 * TS];
  l.A[K] = fma(bt[4 * TS], g.P[K], gtab[c]);
  l.Rn[K] = K == kChunk - 1 ? E[li0_after] : E[g.li0 + K + 1];
  l.Dn[K] = E[g.liD0 + K];
  if (MULTI) l.Ee[K] = E[g.liU0 + K];
}

// lane l <- lane l-1's x; lane 0 (no source lane) keeps `seam` (MULTI) or reads 0.
template <bool MULTI>
__device__ __forceinline__ double shr1_seam(double x, double seam) {
  int lo, hi;
  if (MULTI) { // the DPP `old` operand is the seam value: no extra instruction
    lo = __builtin_amdgcn_update_dpp(__double2loint(seam), __double2loint(x), 0x138, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(__double2hiint(seam), __double2hiint(x), 0x138, 0xf, 0xf, false);
  } else {     // bound_ctrl: out-of-range source reads 0, `old` is a don't-care (no v_mov to set it up)
    lo = __builtin_amdgcn_update_dpp(__double2loint(x), __double2loint(x), 0x138, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(__double2hiint(x), __double2hiint(x), 0x138, 0xf, 0xf, true);
  }
  return __hiloint2double(hi, lo);
}

// One Gauss-Seidel update of the lane's current cell (slot K of the current chunk).
template <bool MULTI, int K>
__device__ __forceinline__ void update_slot(double *E, const StageL &l, int li0, bool act,
                                            double &nv, double &oldv, double &dmax) {
  const double U = shr1_seam<MULTI>(nv, MULTI ? l.Ee[K] : 0.0);
  double t = fma(l.bD[K], l.Dn[K], l.A[K]);
  t = fma(l.bR[K], l.Rn[K], t);
  t = fma(l.bL[K], nv, t);
  const double nvn = fma(l.bU[K], U, t);
  if (act) {
    E[li0 + K] = nvn;
    dmax = fmax(dmax, fabs(nvn - oldv));
  }
  oldv = l.Rn[K];
  nv = nvn;
}

// Computes the chunk held in l (li0, masks) while refilling each freed slot with chunk `nx`.
template <bool MULTI, int TS>
__device__ __forceinline__ void run_chunk(double *E, const double *gtab, const double *tab,
                                          StageL &l, int li0, int mlo, int mhi, const StageG &nx,
                                          int li0_after, double &nv, double &oldv, double &dmax) {
#define SB_SLOT(K)                                                                        \
  update_slot<MULTI, K>(E, l, li0, slot_active<K>(mlo, mhi), nv, oldv, dmax);              \
  load_slot<MULTI, TS, K>(E, gtab, tab, nx, li0_after, l);
  SB_SLOT(0) SB_SLOT(1) SB_SLOT(2) SB_SLOT(3) SB_SLOT(4) SB_SLOT(5) SB_SLOT(6) SB_SLOT(7)
#undef SB_SLOT
}

template <bool FIRST, bool MULTI, int TS>
__device__ double sweep_fast(const Dev &a, double *E, const double *gtab, const double *tab,
                             const double *__restrict__ Pg, int lane) {
  double dmax = 0.0, nv = 0.0;
  const int nch = (a.nsteps + kChunk - 1) / kChunk;
  StageS s0, s1;
  StageG g0, g1;
  StageL l;
  stage_s(a, 0, lane, s0);
  stage_s(a, 1, lane, s1);
  stage_g<FIRST>(a, E, Pg, s0, g0);
  stage_g<FIRST>(a, E, Pg, s1, g1);
  stage_s(a, 2, lane, s0);
  stage_s(a, 3, lane, s1);
#define SB_LOAD0(K) load_slot<MULTI, TS, K>(E, gtab, tab, g0, g1.li0, l);
  SB_LOAD0(0) SB_LOAD0(1) SB_LOAD0(2) SB_LOAD0(3) SB_LOAD0(4) SB_LOAD0(5) SB_LOAD0(6) SB_LOAD0(7)
#undef SB_LOAD0
  double oldv = E[g0.li0];
  int li0 = g0.li0, mlo = g0.mlo, mhi = g0.mhi;
  for (int ch = 0; ch < nch; ch += 2) {
    stage_g<FIRST>(a, E, Pg, s0, g0);   // chunk ch+2
    stage_s(a, ch + 4, lane, s0);
    run_chunk<MULTI, TS>(E, gtab, tab, l, li0, mlo, mhi, g1, g0.li0, nv, oldv, dmax);
    li0 = g1.li0; mlo = g1.mlo; mhi = g1.mhi;
    stage_g<FIRST>(a, E, Pg, s1, g1);   // chunk ch+3
    stage_s(a, ch + 5, lane, s1);
    run_chunk<MULTI, TS>(E, gtab, tab, l, li0, mlo, mhi, g0, g1.li0, nv, oldv, dmax);
    li0 = g0.li0; mlo = g0.mlo; mhi = g0.mhi;
  }
  return wave_max(dmax);
}

template <int TS>
__device__ __forceinline__ double sweep_ts(const Dev &a, double *E, const double *gtab,
                                           const double *tab, const double *__restrict__ Pg,
                                           int lane, bool first) {
  if (a.nbands > 1)
    return first ? sweep_fast<true, true, TS>(a, E, gtab, tab, Pg, lane)
                 : sweep_fast<false, true, TS>(a, E, gtab, tab, Pg, lane);
  return first ? sweep_fast<true, false, TS>(a, E, gtab, tab, Pg, lane)
               : sweep_fast<false, false, TS>(a, E, gtab, tab, Pg, lane);
}

__device__ __forceinline__ double sweep(const Dev &a, double *E, const double *gtab,
                                        const double *tab, const double *__restrict__ Pg,
                                        int lane, bool first) {
  if (!a.fast) return sweep_generic(a, E, gtab, tab, Pg, lane);
  if (a.ts == 32) return sweep_ts<32>(a, E, gtab, tab, Pg, lane, first);
  if (a.ts == 128) return sweep_ts<128>(a, E, gtab, tab, Pg, lane, first);
  return sweep_ts<256>(a, E, gtab, tab, Pg, lane, first);
}

extern __shared__ __attribute__((aligned(16))) double lds[];

__global__ void __launch_bounds__(256) k_sweep_lds(Dev a) {
  const int lane = threadIdx.x & 63;
  const int wib = threadIdx.x >> 6;
  const int wpb = blockDim.x >> 6;
  const int wave = blockIdx.x * wpb + wib;
  const int nwaves = gridDim.x * wpb;
  // LDS: [tab: 5*ts | zone_off] then per wave [guard | E: NL | guard | gtab: ts | zscr: 3*Z | zmode]
  double *tab = lds;
  int *zoffL = (int *)(lds + 5 * a.ts); // [Z+1] zone_off, shared
  double *mine = lds + 5 * a.ts + ((a.Z + 2) >> 1) + (size_t)wib * a.lds_wave_doubles;
  double *E = mine + kGuard; // [guard | E: NL | guard]: the guards stay zero (finite) forever
  for (int i = lane; i < kGuard; i += 64) {
    mine[i] = 0.0;
    mine[kGuard + a.NL + i] = 0.0;
  }
  double *gtab = mine + a.off_agtab;
  double *zscr = mine + a.off_zscr; // [Z..2Z): post-update zone sums
  for (int i = threadIdx.x; i < 5 * a.ts; i += blockDim.x) {
    const int j = i / a.ts, c = i - j * a.ts;
    tab[i] = c < a.ncls ? a.ctab[c * 8 + j] : 0.0; // columns 0..4 of class_coef: bU bD bL bR ap
  }
  for (int i = threadIdx.x; i <= a.Z; i += blockDim.x) zoffL[i] = a.zone_off[i];
  __syncthreads();

  const sb_params &p = a.p;
#define SB_STAMP(i) do { if (a.dbg && b == 0 && lane == 0) a.dbg[i] = (long long)__builtin_readcyclecounter(); } while (0)
  // after its first building a wavefront draws the next one from a device counter (zeroed
  // before every launch): buildings need different numbers of sweeps
  for (int b = wave; b < a.B;) {
    SB_STAMP(0);
    double *T = a.temp + (size_t)b * a.Np + kPad; // T[-kPad..-1] and T[N..N+kPad) are zero
    for (int c = lane; c < a.ts; c += 64) gtab[c] = a.gtabg[(size_t)b * a.ts + c];
    SB_STAMP(1);
    // grid -> LDS.  Even W: one flat copy by LDS-DMA (global_load_lds_dwordx4: 1 KiB per
    // wave instruction, no VGPR round trip, every piece in flight at once).
    if (a.pitch == a.W) {
      for (int i = 0; i < a.N; i += 128) {
        if (i + 2 * lane < a.N)
          __builtin_amdgcn_global_load_lds(
              (const __attribute__((address_space(1))) void *)(T + i + 2 * lane),
              (__attribute__((address_space(3))) void *)(E + i), 16, 0, 0);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      for (int i = lane; i < a.N; i += 64) E[lds_index(a, i)] = T[i];
      for (int x = lane; x < a.H; x += 64)
        for (int y = a.W; y < a.pitch; ++y) E[x * a.pitch + y] = 0.0;
    }
    __builtin_amdgcn_wave_barrier();

    SB_STAMP(2);
    int n_sweeps = 0, converged = 0;
    for (int it = 0; it < p.iter_limit; ++it) { // simulator.py:348-368
      const double md = sweep(a, E, gtab, tab, T, lane, it == 0);
      ++n_sweeps;
      if (md <= p.conv_threshold) { converged = 1; break; }
    }

    SB_STAMP(4);
    // ---- grid back to HBM, new means (feed reward_info now and the next step) ----
    double gsum = 0.0;
    if (a.pitch == a.W) {
      const double2 *src = (const double2 *)E;
      double2 *dst = (double2 *)T;
#pragma unroll 8
      for (int i = lane; i < a.N / 2; i += 64) {
        double2 w = src[i];
        dst[i] = w;
        gsum += w.x + w.y;
      }
    } else {
      for (int i = lane; i < a.N; i += 64) {
        double w = E[lds_index(a, i)];
        T[i] = w;
        gsum += w;
      }
    }
    gsum = wave_sum(gsum);

    SB_STAMP(5);
    // zone sums: u16 index blocks (512 cells each).  Index loads run one batch ahead of the
    // gathers; blocks are walked in zone order (block -> zone from the zone sizes, no lookups)
    for (int z = lane; z < a.Z; z += 64) zscr[a.Z + z] = 0.0;
    __builtin_amdgcn_wave_barrier();
    {
      constexpr int kZB = 6;
      uint4 w[kZB], wn[kZB];
#pragma unroll
      for (int j = 0; j < kZB; ++j) wn[j] = a.zl16[(size_t)min(j, a.n_zblocks - 1) * 64 + lane];
      int zcur = 0, left = a.Z > 0 ? (zoffL[1] - zoffL[0] + 511) >> 9 : 0;
      double zacc = 0.0;
      for (int b0 = 0; b0 < a.n_zblocks; b0 += kZB) {
#pragma unroll
        for (int j = 0; j < kZB; ++j) {
          w[j] = wn[j];
          wn[j] = a.zl16[(size_t)min(b0 + kZB + j, a.n_zblocks - 1) * 64 + lane];
        }
        double part[kZB];
#pragma unroll
        for (int j = 0; j < kZB; ++j) {
          const unsigned q[4] = {w[j].x, w[j].y, w[j].z, w[j].w};
          double acc = 0.0;
#pragma unroll
          for (int t = 0; t < 4; ++t) acc += E[q[t] & 0xffffu] + E[q[t] >> 16];
          part[j] = acc;
        }
#pragma unroll
        for (int j = 0; j < kZB; ++j) part[j] = wave_sum(part[j]);
#pragma unroll
        for (int j = 0; j < kZB; ++j) {
          if (b0 + j < a.n_zblocks) { // uniform control flow; block order == zone order
            while (left == 0) {       // zones without cells
              if (lane == 0) zscr[a.Z + zcur] = zacc;
              zacc = 0.0; ++zcur;
              left = (zoffL[zcur + 1] - zoffL[zcur] + 511) >> 9;
            }
            zacc += part[j];
            if (--left == 0) {
              if (lane == 0) zscr[a.Z + zcur] = zacc;
              zacc = 0.0; ++zcur;
              left = zcur < a.Z ? (zoffL[zcur + 1] - zoffL[zcur] + 511) >> 9 : 1 << 30;
            }
          }
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
    for (int z = lane; z < a.Z; z += 64) a.zsum[(size_t)b * a.Z + z] = zscr[a.Z + z];
    if (lane == 0) {
      a.gsum[b] = gsum;
      a.nsw[b] = n_sweeps | (converged << 16);
    }
    __builtin_amdgcn_wave_barrier();
    SB_STAMP(8);
    if (a.dbg && b == 0 && lane == 0) a.dbg[9] = n_sweeps;
    int nb = 0;
    if (lane == 0) nb = a.sweep_wgs + atomicAdd(a.next_b, 1);
    b = __builtin_amdgcn_readfirstlane(nb);
  }
#undef SB_STAMP
}

} // namespace

int prepare_sweep_lds(size_t lds_bytes) {
  return (int)hipFuncSetAttribute((const void *)k_sweep_lds, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)lds_bytes);
}

int launch_sweep_lds(const Dev &d, int workgroups, int waves_per_wg, size_t lds_bytes, hipStream_t stream) {
  hipLaunchKernelGGL(k_sweep_lds, dim3(workgroups), dim3(64 * waves_per_wg), lds_bytes, stream, d);
  return (int)hipGetLastError();
}

} // namespace sb
