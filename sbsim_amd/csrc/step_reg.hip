// step_reg.hip -- the sweep kernel whose temperature grid lives in registers.
//
// Layout.  The floor plan is trimmed to the bounding box of its non-exterior cells (the
// exterior-space ring is the constant T_ambient and is handled analytically).  Row R of the
// trimmed grid belongs to one lane; the lane keeps the row in NR registers ("slots"), column
// c in slot (c + l) mod NR where l is the lane's index inside its wavefront.  At sweep step d
// every lane works on slot d mod NR -- the SAME register index in all lanes -- which for lane l
// is column d - l: the anti-diagonal wavefront that reproduces the reference's row-major
// in-place Gauss-Seidel order (simulator.py:302-314).  Neighbours:
//     L = slot d-1 of the lane (updated one step ago)      R = slot d+1 (old)
//     U = slot d-1 of lane l-1 (DPP wave_shr:1, new)        D = slot d+1 of lane l+1 (DPP wave_shl:1, old)
// so the sweep touches no memory for the grid at all.  The sweep is fully unrolled (slot
// indices and lane masks are compile-time constants); per step it reads four coefficients
// (two ds_read2_b64), A = ap*Tprev + g (computed once per step into LDS) and runs 4 DPP
// moves + 4 fp64 FMAs in the association order of step_lds.hip (bit-identical iterates).
//
// One or two rows beyond the 64th (R9 has 66) are finished after the wavefront's pass by a
// scan along the row (mode 3, tail_pass below).  In that mode consecutive sweeps overlap: lanes
// start sweep k+1 while the others finish sweep k ("overlapped sweeps" below), so a sweep costs
// NR steps instead of NR + 63.  Otherwise floor plans with more than 64
// rows use two wavefronts per building (mode 2): wave 0 owns the
// upper rows (in its TOP lanes, so that its seam row is lane 63), wave 1 the lower rows
// (seam row = lane 0).  The two seam rows are exchanged through LDS once per 8-step chunk;
// wave 0 counts its finished chunks in LDS and wave 1 starts chunk i only once wave 0 has
// finished chunk i + lag (sb_create: seam_lag), which orders every cross-wave read-after-
// write (wave 0's new seam row) and write-after-read (wave 1's old seam row) -- a workgroup
// barrier per chunk costs ~600 cycles on gfx950, a satisfied flag check nothing.  The DPP
// `old` operand carries the seam value into the edge lane.
//
// HBM state of a building: [NR][RS] float64, slot-major (one coalesced 8*RS-byte row per
// register), pad cells 0.
#include "sb_device.h"

namespace sb {
namespace {

// Kernel modes (template parameter P): 1 = one wavefront per building, <= 64 rows;
// 2 = two wavefronts per building (<= 128 rows); 3 = one wavefront owns rows 0..63 and the
// last one or two rows ("tail") are finished after the wavefront's pass by a parallel scan.
constexpr int kPair = 2, kTail = 3;

// Slots of A = ap*Tprev + g kept in LDS; the remaining NR - lds_slots live in registers
// (AGPRs).  96-slot plans on one wavefront (R9 in mode kTail): 71 of 96 slots in LDS make a
// building fit a quarter of a CU's LDS, so all four SIMDs own a building instead of three.
// The 96-slot two-wavefront variant (up to 128 x 96 cells) keeps 89 slots in LDS: two buildings
// per CU (it runs one wavefront per SIMD: 192 registers of grid + the rest do not fit twice).
constexpr int lds_slots(int NR, int P) {
  return (NR == 96 && P != kPair) ? 71 : ((NR == 96 && P == kPair) ? 89 : NR);
}
constexpr int waves_per_simd(int NR, int P) { return (P == kPair && NR <= 66) ? 2 : 1; }
#ifndef SB_LOOK
#define SB_LOOK 2
#endif
constexpr int kLook = SB_LOOK; // steps between the LDS reads of a step and its arithmetic
#ifndef SB_ROLL
#define SB_ROLL 1
#endif
constexpr int kWin = 63; // overlapped sweeps: steps of a period in which the lanes are in two different sweeps
// Which instantiations overlap consecutive sweeps (see "overlapped sweeps" below): the period (NR
// steps) must hold the window and the read-ahead.  Only the tail-row mode, whose 64 lanes all own
// a row: with 43 rows on one wavefront (mode 1) the 63 heavier window steps cost what the shorter
// sweep saves (measured 2.78 vs 2.71 ms per step for 65,536 buildings of 45 x 96 cells).
#ifndef SB_ROLL_MODE1
#define SB_ROLL_MODE1 0
#endif
constexpr bool rolls(int NR, int P) {
  return SB_ROLL != 0 && (P == 3 || (SB_ROLL_MODE1 != 0 && P == 1)) && NR > kWin + kLook;
}
// Coefficient-table stride: classes + the pad class <= stride.  The class maps hold
// class * (256 / stride) in a byte; times stride / 32 that is the class's byte offset into a
// table column (stride 32: the byte IS the offset, one SDWA add per step).
constexpr int table_stride(int NR, int P) { return (NR == 96 && P == 2) ? 64 : 32; }
constexpr int kSeamPad = 8;

// Hides a value from loop-invariant code motion: without it the compiler precomputes every
// table address of the unrolled loops once per kernel and spills them to scratch.
// (An integer offset is hidden, not the pointer: the pointer keeps its address space.)
__device__ __forceinline__ int opaque(int v) {
  asm volatile("" : "+v"(v));
  return v;
}
__device__ __forceinline__ int opaque_s(int v) { // the same for a wave-uniform value (stays in an SGPR)
  asm volatile("" : "+s"(v));
  return v;
}

// Lanes 0..J as a lane predicate without a VALU compare: the mask is built by one SALU
// instruction where it is used (a compare per step costs a VALU slot plus wait states before
// the select; masks computed once per kernel would be ~130 SGPR pairs, i.e. spilled).
template <int J>
__device__ __forceinline__ bool lanes_upto(int) {
  unsigned long long m;
  asm volatile("s_bfm_b64 %0, %1, 0" : "=s"(m) : "n"(J + 1));
  return __builtin_amdgcn_inverse_ballot_w64(m);
}

struct Co { double bU, bD, bL, bR, A, smU, smD; };
struct Pipe {
  Co co[kLook + 1];
  unsigned long long cw[3]; // class bytes of three consecutive chunks
};

// lane l <- lane l-1 (CTRL 0x138, wave_shr:1) / lane l+1 (0x130, wave_shl:1).  SEAM: a lane
// without source keeps `old` (bound_ctrl = 0; the DPP destination is pre-loaded with the
// seam value, so `old` must be a register nobody else needs); otherwise it reads 0.
template <int CTRL, bool SEAM>
__device__ __forceinline__ double wave_shift1(double x, double old) {
  int lo, hi;
  if (SEAM) {
    lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(x), CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(x), CTRL, 0xf, 0xf, false);
  } else {
    lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, 0xf, 0xf, true);
  }
  return __hiloint2double(hi, lo);
}

// LDS reads of step PD (issued kLook steps early): coefficients by class, A, seam value.
template <int NR, int P, int PD, int NAR>
__device__ __forceinline__ void prefetch(Pipe &p, const double *tab, const double *Arow,
                                         const double (&Areg)[NAR], const double *seam_in,
                                         const double *seam_in2) {
  const unsigned long long cw = p.cw[(PD / 8) % 3];
  constexpr int kTS = table_stride(NR, P);
  const int c8 = (int)((cw >> (8 * (PD % 8))) & 0xffull) * (kTS / 32); // the class's byte offset into a table column
  const double *bt = (const double *)((const char *)tab + c8);
  Co &o = p.co[PD % (kLook + 1)];
  o.bU = bt[0]; o.bD = bt[kTS]; o.bL = bt[2 * kTS]; o.bR = bt[3 * kTS];
  constexpr int NL = lds_slots(NR, P);
  if constexpr ((PD % NR) < NL) o.A = Arow[PD % NR];
  else o.A = Areg[(PD % NR) - NL];
  if (P == kPair) { // two reads of the same value: each DPP consumes its `old` register
    o.smU = seam_in[PD];
    o.smD = seam_in2[PD];
  }
  if (P == kTail) o.smD = seam_in[PD]; // old value of the first tail row under lane 63
}

// One Gauss-Seidel update of every lane's current cell.  lp = lane's row index inside its
// wavefront (0x80000000 for lanes without a row): the lane holds a real column at step D
// iff 0 <= D - lp < NR.
template <int NR, int P, int D>
__device__ __forceinline__ void update(double (&e)[NR], const Co &o, int lp, unsigned long long rowmask,
                                       double &dmax) {
  constexpr int r = D % NR, rm = (D + NR - 1) % NR, rp = (D + 1) % NR;
  const double U = wave_shift1<0x138, P == kPair>(e[rm], o.smU);
  const double Dn = wave_shift1<0x130, P == kPair || P == kTail>(e[rp], o.smD);
  double t = fma(o.bD, Dn, o.A);
  t = fma(o.bR, e[rp], t);
  t = fma(o.bL, e[rm], t);
  const double nv = fma(o.bU, U, t);
  constexpr bool need_hi = D < 63;   // lp <= D can fail
  constexpr bool need_lo = D >= NR;  // lp > D - NR can fail
  bool act;
  if (need_hi && need_lo) act = (unsigned)(D - lp) < (unsigned)NR;
  else if (need_hi) act = P == kTail ? lanes_upto<(D < 63 ? D : 0)>(lp) : (unsigned)lp <= (unsigned)D; // kTail: lp == lane
  else if (need_lo) act = lp > D - NR;
  else act = __builtin_amdgcn_inverse_ballot_w64(rowmask);
  // mode kTail: all 64 lanes own a row, so the middle steps need no select at all
  // all 64 lanes own a row (mode kTail), or the schedule overlaps sweeps (lanes without a row work
  // on a copy nobody stores; the caller drops their max |delta|): the middle steps need no select
  const double sel = ((P == kTail || rolls(NR, P)) && !need_hi && !need_lo) ? nv : (act ? nv : e[r]);
  dmax = fmax(dmax, fabs(sel - e[r]));
  e[r] = sel;
}

struct SweepCtx {
  const double *tab, *Arow, *seam_in, *seam_in2; // seam_in2 == seam_in, but the compiler cannot tell
  double *seam_out;
  const unsigned long long *cmap;
  const unsigned long long *cmapu; // the same table, wave-uniform base (index with lane + 64 * chunk)
  unsigned long long rowmask;
  int lane, lp, nch, edge_off;
  bool edge;
  // P == 2: wave 0 counts its finished chunks in *prog; wave 1 starts chunk i only when wave
  // 0 has finished chunk i + lag (capped at wave 0's last chunk)
  volatile int *prog;
  int role, lag, nch0, prog_base;
};

template <int NR, int P, int CI, int K, int NAR>
__device__ __forceinline__ void chunk_steps(double (&e)[NR], const double (&Areg)[NAR], Pipe &p,
                                            const SweepCtx &x, double &dmax) {
  if constexpr (K < 8 && 8 * CI + K < NR + 63) {
    constexpr int D = 8 * CI + K;
    prefetch<NR, P, D + kLook>(p, x.tab, x.Arow, Areg, x.seam_in, x.seam_in2);
    update<NR, P, D>(e, p.co[D % (kLook + 1)], x.lp, x.rowmask, dmax);
    __builtin_amdgcn_sched_barrier(0);
    chunk_steps<NR, P, CI, K + 1>(e, Areg, p, x, dmax);
  }
}

template <int NR, int P, int CI, int K>
__device__ __forceinline__ void publish(const double (&e)[NR], double *seam_out) {
  if constexpr (K < 8 && 8 * CI + K < NR + 63) {
    seam_out[8 * CI + K] = e[(8 * CI + K) % NR];
    publish<NR, P, CI, K + 1>(e, seam_out);
  }
}

// Wave 1 waits until wave 0's progress counter reaches `target`.  `seen` caches the last
// value read: wave 0 is usually several chunks ahead, so most chunks need no LDS read at all.
__device__ __forceinline__ void wait_progress(volatile int *prog, int target, int &seen) {
  while (seen < target) {
    asm volatile("" ::: "memory");
    seen = __builtin_amdgcn_readfirstlane(*prog);
    if (seen < target) __builtin_amdgcn_s_sleep(1);
  }
  asm volatile("" ::: "memory");
}

template <int NR, int P, int CI, int NAR>
__device__ __forceinline__ void chunks(double (&e)[NR], const double (&Areg)[NAR], Pipe &p, const SweepCtx &x,
                                       double &dmax, int &seen) {
  constexpr int kMaxCh = (NR + 63 + 7) / 8;
  if constexpr (CI < kMaxCh) {
    if (CI < x.nch) { // wave-uniform
      if (P == kPair && x.role == 1) wait_progress(x.prog, x.prog_base + min(CI + x.lag, x.nch0), seen);
      p.cw[(CI + 2) % 3] = x.cmap[opaque(0) + (CI + 2) * 64];
      chunk_steps<NR, P, CI, 0>(e, Areg, p, x, dmax);
      if (P == kPair) {
        // publish the edge row's new values of this chunk (columns 8*CI-edge_off .. +7)
        const int c0 = 8 * CI - x.edge_off;
        if (c0 + 7 >= 0 && c0 < NR) {
          if (x.edge) publish<NR, P, CI, 0>(e, x.seam_out);
        }
      }
      if (P == kPair) {
        // LDS executes a wavefront's instructions in order: whoever sees the counter sees the
        // seam values written before it, and every LDS read this wave issued so far is done
        asm volatile("" ::: "memory");
        if (x.role == 0 && x.lane == 0) *x.prog = x.prog_base + CI + 1;
        asm volatile("" ::: "memory");
      }
      chunks<NR, P, CI + 1>(e, Areg, p, x, dmax, seen);
    }
  }
}

template <int NR, int P, int NAR>
__device__ __forceinline__ double sweep_reg(double (&e)[NR], const double (&Areg)[NAR], const SweepCtx &xin) {
  double dmax = 0.0;
  int seen = 0;
  // the per-step lane predicates are one v_cmp each; hoisted out of the sweep loop they would
  // be ~150 SGPR pairs spilled to VGPR lanes (2 v_readlane + wait states per step)
  SweepCtx x = xin;
  asm volatile("" : "+v"(x.lp));
  if (P == kPair && x.role == 1) wait_progress(x.prog, x.prog_base + min(x.lag, x.nch0), seen);
  Pipe p;
  p.cw[0] = x.cmap[opaque(0)];
  p.cw[1] = x.cmap[opaque(0) + 64];
  p.cw[2] = 0;
  prefetch<NR, P, 0>(p, x.tab, x.Arow, Areg, x.seam_in, x.seam_in2);
  prefetch<NR, P, 1>(p, x.tab, x.Arow, Areg, x.seam_in, x.seam_in2);
  __builtin_amdgcn_sched_barrier(0);
  chunks<NR, P, 0>(e, Areg, p, x, dmax, seen);
  if (P == kTail && x.edge) { // row 63's new values for the tail scan: lane 63's registers, column c in slot c + 63
#pragma unroll
    for (int c = 0; c < NR; ++c) x.seam_out[c + 63] = e[(c + 63) % NR];
  }
  return dmax;
}

// ---------------------------------------------------------------- overlapped sweeps (mode kTail)
// With all 64 lanes owning a row, the slot of a step is the same register in every lane no
// matter which sweep the lane is in, as long as consecutive sweeps start exactly NR steps
// apart: lane l then works on column (s - l) mod NR at global step s.  So lanes 0..j START
// sweep k+1 during steps j = 0..62 of a period while lanes j+1..63 FINISH sweep k -- no lane
// idles on the ramps, a sweep costs NR steps instead of NR + 63.  The price: whether sweep k
// was the last one (simulator.py:360) is known only after step 62 (plus the tail pass), so the
// start of sweep k+1 is speculative; every step of the window first copies the value it
// overwrites (bk[j]), and the last period restores lanes <= j from the copies.  Max |delta|
// goes to the accumulator of the lane's own sweep (dcur: sweep k, dnext: sweep k+1).

template <int NR, int P, int J>
__device__ __forceinline__ void update_mixed(double (&e)[NR], double (&bk)[kWin], const Co &o, int lp,
                                             double &dcur, double &dnext) {
  constexpr int r = J, rm = (J + NR - 1) % NR, rp = J + 1;
  const double U = wave_shift1<0x138, false>(e[rm], 0.0);
  const double Dn = wave_shift1<0x130, P == kTail>(e[rp], o.smD);
  double t = fma(o.bD, Dn, o.A);
  t = fma(o.bR, e[rp], t);
  t = fma(o.bL, e[rm], t);
  const double nv = fma(o.bU, U, t);
  const double d = nv - e[r];
  bk[J] = e[r];
  e[r] = nv;
  // |d| into one accumulator, (almost) zero into the other: only the high word is switched,
  // the low word alone is a subnormal < 5e-314
  const bool nw = lanes_upto<J>(lp);
  const int hi = __double2hiint(d), lo = __double2loint(d);
  dnext = fmax(dnext, fabs(__hiloint2double(nw ? hi : 0, lo)));
  dcur = fmax(dcur, fabs(__hiloint2double(nw ? 0 : hi, lo)));
  // here, not after the window: e[J] and bk[J] both survive the window, so the compiler would
  // sink all of this behind it and keep 63 lane masks alive (spilled SGPRs)
  asm volatile("" : "+v"(dcur), "+v"(dnext));
}

// Steps D0 <= D < D1 of the overlapped schedule: D < 63 ramp-up of the first sweep (lanes > D
// idle), 63 <= D < NR all lanes in one sweep, NR <= D < NR + 63 the mixed window.
template <int NR, int P, int D, int D1, int NAR>
__device__ __forceinline__ void roll_steps(double (&e)[NR], double (&bk)[kWin], const double (&Areg)[NAR],
                                           Pipe &p, const SweepCtx &x, double &dcur, double &dnext) {
  if constexpr (D < D1) {
    // uniform base + lane offset + immediate: one global_load, no 64-bit address arithmetic
    if constexpr (D % 8 == 0) p.cw[(D / 8 + 2) % 3] = x.cmapu[x.lane + (D / 8 + 2) * 64];
    if constexpr (D + kLook < D1) prefetch<NR, P, D + kLook>(p, x.tab, x.Arow, Areg, x.seam_in, x.seam_in2);
    if constexpr (D < NR) update<NR, P, D>(e, p.co[D % (kLook + 1)], x.lp, x.rowmask, dcur);
    else update_mixed<NR, P, D - NR>(e, bk, p.co[D % (kLook + 1)], x.lp, dcur, dnext);
    __builtin_amdgcn_sched_barrier(0);
    roll_steps<NR, P, D + 1, D1>(e, bk, Areg, p, x, dcur, dnext);
  }
}

template <int NR, int J>
__device__ __forceinline__ void roll_back(double (&e)[NR], const double (&bk)[kWin], int lp) {
  if constexpr (J < kWin) {
    e[J] = lanes_upto<J>(lp) ? bk[J] : e[J];
    roll_back<NR, J + 1>(e, bk, lp);
  }
}

// ---------------------------------------------------------------- tail rows (mode kTail)
// Rows 64.. of the trimmed grid (at most two) are finished after the wavefront's pass, lanes
// = columns.  Along a row the Gauss-Seidel update is the first-order recurrence
//     x_c = bL_c * x_{c-1} + q_c,    q_c = A + bD*D_old + bR*R_old + bU*U_new,
// which an inclusive scan over the affine maps f_c(x) = bL_c x + q_c evaluates in
// log2(64) DPP steps (F <- F o F_shifted; lanes without a source compose with the identity).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ void scan_step(double &a, double &q) {
  const double one = 1.0;
  const int alo = __builtin_amdgcn_update_dpp(__double2loint(one), __double2loint(a), CTRL, ROW_MASK, 0xf, false);
  const int ahi = __builtin_amdgcn_update_dpp(__double2hiint(one), __double2hiint(a), CTRL, ROW_MASK, 0xf, false);
  const int qlo = __builtin_amdgcn_update_dpp(0, __double2loint(q), CTRL, ROW_MASK, 0xf, false);
  const int qhi = __builtin_amdgcn_update_dpp(0, __double2hiint(q), CTRL, ROW_MASK, 0xf, false);
  const double as = __hiloint2double(ahi, alo), qs = __hiloint2double(qhi, qlo);
  q = fma(a, qs, q); // (a, q) o (as, qs) = (a*as, a*qs + q)
  a = a * as;
}
__device__ __forceinline__ void affine_scan(double &a, double &q) {
  scan_step<0x111, 0xf>(a, q); // row_shr:1,2,4,8: inclusive scan inside each row of 16 lanes
  scan_step<0x112, 0xf>(a, q);
  scan_step<0x114, 0xf>(a, q);
  scan_step<0x118, 0xf>(a, q);
  scan_step<0x142, 0xa>(a, q); // row_bcast:15 -> rows 1 and 3
  scan_step<0x143, 0xc>(a, q); // row_bcast:31 -> rows 2 and 3
}

constexpr int kTailMax = 2;
// A lane owns two neighbouring columns of a tail row: NR <= 128 columns are one 64-lane scan.
__device__ __forceinline__ int tail_col(int lane, int k) { return 2 * lane + k; }

// One Gauss-Seidel pass over the tail rows; returns the lane's max |delta|.
// tE: [T][NR+2] current values (column c at [1 + c]); r63: new values of row 63 by column.
// The lane first composes the maps of its two columns (x_{2l+1} = bL1*(bL0*x + q0) + q1),
// the scan runs over the 64 composed maps, and the even column follows from its left
// neighbour's result.
template <int NR>
__device__ __forceinline__ double tail_pass(int T, int lane, const double *tab, double *tE, const double *r63,
                                            const double (&At)[kTailMax][2], unsigned tclsw) {
  constexpr int kRow = NR + 2, kTS = table_stride(NR, kTail);
  static_assert(kTS == 32, "the tail-row class bytes are byte offsets");
  static_assert(NR % 2 == 0 && NR <= 128, "two columns per lane");
  const bool active = 2 * lane < NR;
  const int c0 = active ? 2 * lane : NR - 2;
  double dmax = 0.0;
#pragma unroll
  for (int t = 0; t < kTailMax; ++t) {
    if (t < T) {
      double *row = tE + t * kRow + 1;
      const double *bt0 = (const double *)((const char *)tab + (int)((tclsw >> (8 * (t * 2))) & 0xffu));
      const double *bt1 = (const double *)((const char *)tab + (int)((tclsw >> (8 * (t * 2 + 1))) & 0xffu));
      const double bU0 = bt0[0], bD0 = bt0[kTS], bL0 = bt0[2 * kTS], bR0 = bt0[3 * kTS];
      const double bU1 = bt1[0], bD1 = bt1[kTS], bL1 = bt1[2 * kTS], bR1 = bt1[3 * kTS];
      const double old0 = row[c0], old1 = row[c0 + 1], R1 = row[c0 + 2];
      const double U0 = t == 0 ? r63[c0] : row[c0 - kRow], U1 = t == 0 ? r63[c0 + 1] : row[c0 + 1 - kRow];
      const double D0 = t + 1 < T ? row[c0 + kRow] : 0.0, D1 = t + 1 < T ? row[c0 + 1 + kRow] : 0.0;
      const double q0 = fma(bU0, U0, fma(bR0, old1, fma(bD0, D0, At[t][0]))); // right neighbour: not yet updated
      const double q1 = fma(bU1, U1, fma(bR1, R1, fma(bD1, D1, At[t][1])));
      double a = bL1 * bL0, Q = fma(bL1, q0, q1);
      affine_scan(a, Q); // Q: the odd column (the row starts from bL = 0: no carry-in)
      const double xl = wave_shift1<0x138, false>(Q, 0.0); // column 2l - 1
      const double x0 = fma(bL0, xl, q0);
      __builtin_amdgcn_wave_barrier(); // every read of the row's old values is done
      if (active) {
        dmax = fmax(dmax, fmax(fabs(x0 - old0), fabs(Q - old1)));
        row[c0] = x0;
        row[c0 + 1] = Q;
      }
      __builtin_amdgcn_wave_barrier(); // the next row reads this one
    }
  }
  return dmax;
}

extern __shared__ __attribute__((aligned(16))) double lds[];

template <int NR, int P>
__global__ void __launch_bounds__(P == kPair ? 128 : 64)
    __attribute__((amdgpu_waves_per_eu(waves_per_simd(NR, P), waves_per_simd(NR, P)))) k_sweep_reg(Dev a) {
  const int lane = threadIdx.x & 63;
  const int w = P == kPair ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : 0;
  constexpr int kASlots = (NR + 7) / 8, kZSlots = (NR + 3) / 4, kMaxCh = (NR + 63 + 7) / 8;
  constexpr int kTS = table_stride(NR, P);

  double *tab = lds;                       // [5][kTS]: bU bD bL bR ap
  double *gtab = lds + 5 * kTS;            // [kTS]
  double *seamD = lds + a.r_seam;          // [pad | NR | pad] old values of wave 1's first row
  double *seamU = seamD + NR + 2 * kSeamPad; // new values of wave 0's last row
  // mode kTail uses the same region as [pad | r63: NR | pad][tE: T x (NR + 2)]
  double *r63 = seamD + kSeamPad;          // new values of row 63, by column
  double *tE = seamD + NR + 2 * kSeamPad;  // tail rows, column c at [t*(NR+2) + 1 + c]
  double *A = lds + a.r_A;                 // [RS][AS >= NR]; after the sweeps: zone sums [Z+1][RS]
  double *xchg = lds + a.r_xchg;           // [0..3] max delta (2 sweeps x 2 waves), [4..5] grid sums, [6] progress counter
  // every byte of LDS starts finite: seam / A reads next to the arrays' ends are multiplied by 0
  for (int i = threadIdx.x; i < a.lds_reg_bytes / 8; i += blockDim.x) lds[i] = 0.0;
  __syncthreads();
  for (int i = threadIdx.x; i < 5 * kTS; i += blockDim.x) {
    const int j = i / kTS, c = i - j * kTS;
    tab[i] = c <= a.ncls ? a.ctab[c * 8 + j] : 0.0; // row `ncls` is the pad class: T' = Tprev (ap = 1)
  }
  __syncthreads();

  const sb_params &p = a.p;
  const int lw = a.lw[w], l0 = a.l0[w], rowbase = a.rowbase[w];
  const int lp = lane - l0;
  const bool rowvalid = lp >= 0 && lp < lw;
  const int R = rowbase + (rowvalid ? lp : 0);
  const int RS = P == kTail ? 64 : a.RS; // mode kTail: a compile-time stride -> immediate offsets
  SweepCtx x;
  x.tab = tab;
  x.Arow = A + (size_t)R * a.AS; // odd row stride: the 64 lanes of a ds_read_b64 cover all 32 banks
  constexpr int kNL = lds_slots(NR, P), kNAR = NR - kNL > 0 ? NR - kNL : 1;
  x.cmap = a.cmapS + (size_t)w * (kMaxCh + 3) * 64 + lane;
  x.cmapu = a.cmapS + (size_t)w * (kMaxCh + 3) * 64;
  x.rowmask = __builtin_amdgcn_ballot_w64(rowvalid);
  x.lane = lane; x.lp = rowvalid ? lp : (int)0x80000000; x.nch = a.nch[w];
  x.edge_off = (P == kPair && w == 0) ? lw - 1 : (P == kTail ? 63 : 0);
  x.edge = (P == kPair && lane == (w == 0 ? 63 : 0)) || (P == kTail && lane == 63);
  x.seam_in = (P == kTail ? tE + 1 : (w == 0 ? seamD : seamU) + kSeamPad) - x.edge_off;
  x.seam_in2 = x.seam_in + opaque(0);
  x.seam_out = (P == kTail ? r63 : (w == 0 ? seamU : seamD) + kSeamPad) - x.edge_off;
  unsigned tclsw = 0; // mode kTail: classes of the lane's tail cells, byte [t*2 + block]
  if (P == kTail)
    for (int t = 0; t < a.T; ++t)
      for (int k = 0; k < 2; ++k)
        tclsw |= (unsigned)a.tcls[t * NR + min(tail_col(lane, k), NR - 1)] << (8 * (t * 2 + k));
  x.prog = (volatile int *)(xchg + 6);
  x.role = w; x.lag = a.lag; x.nch0 = a.nch[0]; x.prog_base = 0;
  const unsigned long long *amap = a.amapS + (size_t)w * kASlots * 64 + lane;
  const unsigned long long *zmap = a.zmapS + (size_t)w * kZSlots * 64 + lane;

// developer aid: cycle stamps of workgroup 0's 11th building (steady state, not the cold start)
#define SB_STAMP(i) do { if (a.dbg && blockIdx.x == 0 && iter == 10 && w == 0 && lane == 0) a.dbg[i] = (long long)__builtin_readcyclecounter(); } while (0)
  // The lane's class bytes by slot (A pass) do not depend on the building: loaded once, they
  // live in AGPRs between their uses (the allocator spills what the sweep does not touch)
  // instead of costing a global round trip per building.  (The two-wavefront mode has no
  // registers to spare and reloads them.)
  constexpr bool kReloadAmap = P == kPair || rolls(NR, P); // no registers to spare
  unsigned long long amapw[kASlots];
  if (!kReloadAmap) {
#pragma unroll
    for (int g = 0; g < kASlots; ++g) amapw[g] = amap[g * 64];
  }

  // The lane's row of the NEXT building is loaded while this building's zone sums are reduced
  // (its registers are free once the row is stored), so the loop never waits on HBM latency.
  double e[NR];
#define SB_LOAD_ROW(bb)                                                                         \
  do {                                                                                          \
    const double *tp_ = a.temp + (size_t)(bb) * a.state_doubles; /* wave-uniform: SGPR base + lane offset */ \
    _Pragma("unroll") for (int j = 0; j < NR; ++j) { /* pad lanes mirror a real row: never updated, never stored */ \
      e[j] = tp_[R];                                                                            \
      tp_ += P == kTail ? RS : opaque_s(RS);                                                    \
    }                                                                                           \
  } while (0)
  // ... and so are the building's small inputs: its g table entry, tail rows, ambient
  // temperature and the extremes of its exterior ring.
  double nx_g = 0.0, nx_tail[kTailMax][2] = {{0.0, 0.0}, {0.0, 0.0}}, nx_tnow = 0.0, nx_lo = 0.0, nx_hi = 0.0;
#define SB_LOAD_AUX(bb)                                                                         \
  do {                                                                                          \
    nx_tnow = a.bld[(bb)].t_now;                                                                \
    nx_lo = a.scal[(size_t)(bb) * kNScal + 16];                                                 \
    nx_hi = a.scal[(size_t)(bb) * kNScal + 17];                                                 \
    if (w == 0) nx_g = a.gtabg[(size_t)(bb) * kTS + (lane & (kTS - 1))];                        \
    if (P == kTail) {                                                                           \
      const double *tt_ = a.temp + (size_t)(bb) * a.state_doubles + NR * 64;                    \
      _Pragma("unroll") for (int t = 0; t < kTailMax; ++t)                                      \
        _Pragma("unroll") for (int k = 0; k < 2; ++k)                                           \
          if (t < a.T) nx_tail[t][k] = tt_[t * NR + min(tail_col(lane, k), NR - 1)];            \
    }                                                                                           \
  } while (0)
  if ((int)blockIdx.x < a.B) {
    SB_LOAD_ROW(blockIdx.x);
    SB_LOAD_AUX(blockIdx.x);
  }
  // Buildings need different numbers of sweeps, so a static split would leave the last
  // workgroups running alone: after its first building a workgroup draws the next one from a
  // device counter (zeroed before every launch).  In the two-wavefront mode wave 0 draws
  // and hands the index to wave 1 through LDS at the building's first barrier.
  int *draw_slot = (int *)(xchg + 7);
  int iter = 0;
  for (int b = blockIdx.x, bn = 0; b < a.B; b = bn, ++iter) {
    if (w == 0) { // early: the epilogue prefetches building bn
      int nb = 0;
      if (lane == 0) nb = a.sweep_wgs + atomicAdd(a.next_b, 1);
      bn = __builtin_amdgcn_readfirstlane(nb);
      if (P == kPair && lane == 0) *draw_slot = bn;
    }
    SB_STAMP(0);
    if (kReloadAmap && P != kPair) { // issued here, used by the A pass: the setup hides the latency
      const int o = opaque(0);
#pragma unroll
      for (int g = 0; g < kASlots; ++g) amapw[g] = amap[o + g * 64];
    }
    double *T = a.temp + (size_t)b * a.state_doubles + R;
    double *Ttail = a.temp + (size_t)b * a.state_doubles + NR * 64; // mode kTail: [T][NR]
    __builtin_amdgcn_sched_barrier(0);
    const double t_now = nx_tnow;
    // exterior-space cells outside the trim box all become t_now in the first sweep
    // (simulator.py:256-258); their largest |delta| follows from their extreme values
    const double ring_d = a.n_ring > 0 ? fmax(fabs(t_now - nx_lo), fabs(t_now - nx_hi)) : 0.0;
    if (w == 0) {
      if (P == kPair && lane == 0) *x.prog = 0;
      if (lane < kTS) gtab[lane] = nx_g;
      if (P == kPair) // old values of wave 1's first row (its lane 0: column c sits in slot c)
        for (int c = lane; c < NR; c += 64)
          seamD[kSeamPad + c] = a.temp[(size_t)b * a.state_doubles + (size_t)c * RS + a.rowbase[1]];
      if (P == kTail) {
#pragma unroll
        for (int t = 0; t < kTailMax; ++t)
#pragma unroll
          for (int k = 0; k < 2; ++k)
            if (t < a.T && tail_col(lane, k) < NR) tE[t * (NR + 2) + 1 + tail_col(lane, k)] = nx_tail[t][k];
      }
    }
    if (P == kPair) __syncthreads(); else __builtin_amdgcn_wave_barrier();
    if (P == kPair && w == 1) bn = __builtin_amdgcn_readfirstlane(*draw_slot);
    __builtin_amdgcn_sched_barrier(0);
    SB_STAMP(1);
    double At[kTailMax][2] = {{0.0, 0.0}, {0.0, 0.0}}; // mode kTail: A of the lane's tail cells
    if (P == kTail) {
#pragma unroll
      for (int t = 0; t < kTailMax; ++t)
#pragma unroll
        for (int k = 0; k < 2; ++k)
          if (t < a.T) {
            const int cls8 = (int)((tclsw >> (8 * (t * 2 + k))) & 0xffu);
            const double tp = tE[t * (NR + 2) + 1 + min(tail_col(lane, k), NR - 1)];
            At[t][k] = fma(*(const double *)((const char *)(tab + 4 * kTS) + cls8), tp,
                             *(const double *)((const char *)gtab + cls8));
          }
    }

    // A = ap*Tprev + g for every cell of the lane's row (E = Tprev before the first sweep)
    double Areg[kNAR];
    Areg[0] = 0.0;
    if (P != kTail) { // lanes without a row keep finite values (0 * NaN would reach a real row)
#pragma unroll
      for (int k = 0; k < kNAR; ++k) Areg[k] = 0.0;
    }
    if (rowvalid) {
      if (kReloadAmap && P == kPair) {
        const int o = opaque(0);
#pragma unroll
        for (int g = 0; g < kASlots; ++g) amapw[g] = amap[o + g * 64];
      }
      const unsigned long long(&cw)[kASlots] = amapw;
      double *Aw = A + (size_t)R * a.AS;
      // groups of 8, software-pipelined: the table reads of group g+1 are issued before the
      // A values of group g are written (the compiler cannot prove that A and the tables do
      // not alias and would serialise every read behind the previous write)
      constexpr int kGroups = (NR + 7) / 8;
      double ap[2][8], gg[2][8];
      auto fetch = [&](int g, double (&pa)[8], double (&pg)[8]) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int j = min(8 * g + k, NR - 1);
          const int c8 = (int)((cw[j >> 3] >> (8 * (j & 7))) & 0xffull) * (kTS / 32);
          pa[k] = *(const double *)((const char *)(tab + 4 * kTS) + c8);
          pg[k] = *(const double *)((const char *)gtab + c8);
        }
      };
      fetch(0, ap[0], gg[0]);
#pragma unroll
      for (int g = 0; g < kGroups; ++g) {
        if (g + 1 < kGroups) fetch(g + 1, ap[(g + 1) & 1], gg[(g + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < 8; ++k)
          if (8 * g + k < NR) {
            const double av = fma(ap[g & 1][k], e[8 * g + k], gg[g & 1][k]);
            if (8 * g + k < kNL) Aw[8 * g + k] = av;
            else Areg[8 * g + k - kNL] = av;
          }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_sched_barrier(0);
    SB_STAMP(2);

    int n_sweeps = 0, converged = 0;
    if constexpr (rolls(NR, P)) {
      SweepCtx xr = x;
      asm volatile("" : "+v"(xr.lp)); // one v_cmp per step instead of hoisted, spilled masks
      Pipe pp;
      double dcur = 0.0, dnext = 0.0, bk[kWin];
      pp.cw[0] = xr.cmapu[xr.lane];
      pp.cw[1] = xr.cmapu[xr.lane + 64];
      pp.cw[2] = 0;
      prefetch<NR, P, 0>(pp, xr.tab, xr.Arow, Areg, xr.seam_in, xr.seam_in2);
      prefetch<NR, P, 1>(pp, xr.tab, xr.Arow, Areg, xr.seam_in, xr.seam_in2);
      __builtin_amdgcn_sched_barrier(0);
      roll_steps<NR, P, 0, kWin>(e, bk, Areg, pp, xr, dcur, dnext);
      // class bytes of steps 56..79: loaded before the tail pass that precedes their use
      auto restart_classes = [&]() {
        pp.cw[(kWin / 8) % 3] = xr.cmapu[xr.lane + (kWin / 8) * 64];
        pp.cw[(kWin / 8 + 1) % 3] = xr.cmapu[xr.lane + (kWin / 8 + 1) * 64];
        pp.cw[(kWin / 8 + 2) % 3] = xr.cmapu[xr.lane + (kWin / 8 + 2) * 64];
      };
      restart_classes();
#pragma nounroll
      for (;;) { // simulator.py:348-368
        asm volatile("" : "+v"(xr.lp));
        // the LDS reads of steps 63.. follow the tail pass of the previous sweep (row 63 reads
        // the first tail row), so the read-ahead pipeline restarts here
        prefetch<NR, P, kWin>(pp, xr.tab, xr.Arow, Areg, xr.seam_in, xr.seam_in2);
        prefetch<NR, P, kWin + 1>(pp, xr.tab, xr.Arow, Areg, xr.seam_in, xr.seam_in2);
        __builtin_amdgcn_sched_barrier(0);
        roll_steps<NR, P, kWin, NR + kWin>(e, bk, Areg, pp, xr, dcur, dnext);
        if (xr.edge) { // row 63 is still in sweep k: its new values for the tail scan
#pragma unroll
          for (int c = 0; c < NR; ++c) xr.seam_out[c + 63] = e[(c + 63) % NR];
        }
        restart_classes();
        __builtin_amdgcn_wave_barrier();
        double dm = rowvalid ? dcur : 0.0; // lanes without a row ran on a copy
        if constexpr (P == kTail) dm = fmax(dm, tail_pass<NR>(a.T, lane, tab, tE, r63, At, tclsw));
        double md = wave_max(dm);
        if (n_sweeps == 0) md = fmax(md, ring_d);
        ++n_sweeps;
        converged = md <= p.conv_threshold;
        if (converged || n_sweeps >= p.iter_limit) {
          asm volatile("" : "+v"(xr.lp));
          roll_back<NR, 0>(e, bk, xr.lp); // undo the started sweep
          break;
        }
        dcur = dnext;
        dnext = 0.0;
      }
    } else
#pragma nounroll
    for (int it = 0; it < p.iter_limit; ++it) { // simulator.py:348-368
      x.prog_base = it * 32; // the counter only grows within a building's step
      double dm = sweep_reg<NR, P>(e, Areg, x);
      if (P == kTail) {
        __builtin_amdgcn_wave_barrier(); // row 63's last values are in LDS before the scan reads them
        dm = fmax(dm, tail_pass<NR>(a.T, lane, tab, tE, r63, At, tclsw));
      }
      double md = wave_max(dm);
      if (P == kPair) {
        double *xd = xchg + 2 * (it & 1); // double-buffered: wave 0 may finish its next sweep early
        if (lane == 0) xd[w] = md;
        __syncthreads();
        md = fmax(xd[0], xd[1]);
      }
      if (it == 0) md = fmax(md, ring_d);
      ++n_sweeps;
      if (md <= p.conv_threshold) { converged = 1; break; }
    }
    __builtin_amdgcn_sched_barrier(0);
    SB_STAMP(3);
    SB_STAMP(4);

    // grid back to HBM.  Zone sums (A is dead now): every lane adds its cells into its own
    // column of zs[zone][row]; row Z collects every cell outside a zone, so that the sum of all
    // rows is the grid sum.
    double *zs = A;
    const int ZRS = a.ZRS;
    if (rowvalid) {
      unsigned long long zwv[kZSlots]; // zone-sum offsets: loaded while the row is stored
      {
        const int o = opaque(0);
#pragma unroll
        for (int g = 0; g < kZSlots; ++g) zwv[g] = zmap[o + g * 64];
      }
      __builtin_amdgcn_sched_barrier(0);
      // one 8-byte store per register straight from e[]: a wider store would have to be
      // assembled in a temporary that the next store overwrites (measured 6x slower)
      double *tp = a.temp + (size_t)b * a.state_doubles;
      constexpr bool kFuse = P == kTail; // store, zone add and the next building's load slot by slot
      if (!kFuse) {
#pragma unroll
        for (int j = 0; j < NR; ++j) {
          tp[R] = e[j];
          tp += P == kTail ? RS : opaque_s(RS);
        }
      }
      for (int z = 0; z <= a.Z; ++z) zs[(size_t)z * ZRS + R] = 0.0;
      __builtin_amdgcn_wave_barrier();
      if (P == kTail) // the tail rows hold no zone cells (sb_create checks): all into row Z
        for (int t = 0; t < a.T; ++t)
          for (int c = lane; c < NR; c += 64) {
            const double tv = tE[t * (NR + 2) + 1 + c];
            Ttail[t * NR + c] = tv;
            __hip_atomic_fetch_add(zs + (size_t)a.Z * ZRS + lane, tv, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_WORKGROUP);
          }
      const double *np_ = a.temp + (size_t)(bn < a.B ? bn : b) * a.state_doubles;
#pragma unroll
      for (int j = 0; j < NR; ++j) {
        const unsigned off = (unsigned)((zwv[j >> 2] >> (16 * (j & 3))) & 0xffffull);
        if (kFuse) { tp[R] = e[j]; tp += RS; }
        __hip_atomic_fetch_add((double *)((char *)zs + off), e[j], __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_WORKGROUP);
        if (kFuse) { e[j] = np_[R]; np_ += RS; }
        if ((j & 7) == 7) __builtin_amdgcn_sched_barrier(0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    SB_STAMP(5);
    if (bn < a.B) {
      if (P != kTail) SB_LOAD_ROW(bn);
      SB_LOAD_AUX(bn);
    }
    __builtin_amdgcn_sched_barrier(0);
    SB_STAMP(6);
    if (P == kPair) __syncthreads(); else __builtin_amdgcn_wave_barrier();
    SB_STAMP(7);

    if (w == 0) { // hand the zone sums, the grid sum and the sweep count to k_post
      // 16 zones x 4 row groups per pass: lane (zone = lane & 15, group = lane >> 4) adds every
      // fourth row of its zone, two xor-shuffles combine the groups
      double gacc = 0.0;
      for (int zb = 0; zb <= a.Z; zb += 16) {
        const int zz = zb + (lane & 15), g = lane >> 4;
        double v = 0.0;
        if (zz <= a.Z)
          for (int r = g; r < RS; r += 4) v += zs[(size_t)zz * ZRS + r];
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        if (lane < 16 && zz < a.Z) a.zsum[(size_t)b * a.Z + zz] = v;
        if (lane < 16 && zz <= a.Z) gacc += v;
      }
      const double gsum = wave_sum(gacc);
      if (lane == 0) {
        a.gsum[b] = gsum + (double)a.n_ring * t_now;
        a.nsw[b] = n_sweeps | (converged << 16);
      }
      SB_STAMP(8);
      if (a.dbg && blockIdx.x == 0 && iter == 10 && lane == 0) a.dbg[9] = n_sweeps;
    }
    // no barrier here: wave 1 touches no LDS of the next building before the barrier that
    // follows wave 0's g-table load
  }
#undef SB_STAMP
#undef SB_LOAD_ROW
#undef SB_LOAD_AUX
}

struct Variant {
  int NR, P;
  const void *fn;
  void (*launch)(const Dev &, int, hipStream_t);
};

template <int NR, int P>
void launch_variant(const Dev &d, int workgroups, hipStream_t stream) {
  hipLaunchKernelGGL((k_sweep_reg<NR, P>), dim3(workgroups), dim3(P == kPair ? 128 : 64),
                     (size_t)d.lds_reg_bytes, stream, d);
}

#define SB_VARIANT(NR, P) {NR, P, (const void *)k_sweep_reg<NR, P>, launch_variant<NR, P>}
const Variant kVariants[] = {SB_VARIANT(32, 1), SB_VARIANT(66, 1), SB_VARIANT(66, 2), SB_VARIANT(96, 1),
                            SB_VARIANT(96, 2)}; // mode 3 (tail rows, overlapped sweeps): step_roll.hip
#undef SB_VARIANT

const Variant *find_variant(int NR, int P) {
  for (const Variant &v : kVariants)
    if (v.NR == NR && v.P == P) return &v;
  return nullptr;
}

} // namespace

bool sweep_reg_supported(int NR, int P) { return find_variant(NR, P) != nullptr; }
int sweep_reg_table_stride(int NR, int P) { return table_stride(NR, P); }
int sweep_reg_lds_slots(int NR, int P) { return lds_slots(NR, P); }
int sweep_reg_waves_per_simd(int NR, int P) { return waves_per_simd(NR, P); }
bool sweep_reg_overlaps_sweeps(int NR, int P) { return rolls(NR, P); }

int prepare_sweep_reg(const Dev &d) {
  const Variant *v = find_variant(d.NR, d.P);
  if (!v) return (int)hipErrorInvalidValue;
  return (int)hipFuncSetAttribute(v->fn, hipFuncAttributeMaxDynamicSharedMemorySize, d.lds_reg_bytes);
}

int launch_sweep_reg(const Dev &d, int cus, hipStream_t stream) {
  const Variant *v = find_variant(d.NR, d.P);
  if (!v) return (int)hipErrorInvalidValue;
  (void)cus;
  v->launch(d, d.sweep_wgs, stream); // == min(B, CUs * wg_per_cu)
  return (int)hipGetLastError();
}

} // namespace sb
