// sweep_common.h -- what the register-resident sweep kernels share (step_roll.hip, step_two.hip):
// lane predicates and DPP moves, and the tail rows -- the rows below the wavefront's (at most two),
// finished by an affine scan with lanes = columns.  simulator.py:278-371.
#pragma once

#include "sb_device.h"

namespace sb {
namespace sweep {

constexpr int kTailMax = 2;

__device__ __forceinline__ int opaque(int v) {
  asm volatile("" : "+v"(v));
  return v;
}

template <int J>
__device__ __forceinline__ bool lanes_upto() { // lanes 0..J as a lane predicate: one SALU instruction
  unsigned long long m;
  asm volatile("s_bfm_b64 %0, %1, 0" : "=s"(m) : "n"(J + 1));
  return __builtin_amdgcn_inverse_ballot_w64(m);
}

// lane l <- lane l-1 (CTRL 0x138, wave_shr:1) / lane l+1 (0x130, wave_shl:1) / rotate (0x13c,
// wave_ror:1).  SEAM: the lane without a source keeps `old`; otherwise it reads 0.
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

typedef double d2 __attribute__((ext_vector_type(2))); // two doubles = one ds_read_b128
typedef const d2 __attribute__((address_space(3))) *lds_d2;

// ---------------------------------------------------------------- tail rows
// Along a row the Gauss-Seidel update is x_c = bL_c * x_{c-1} + q_c: an inclusive scan over the
// affine maps f_c(x) = bL_c x + q_c in log2(64) DPP steps.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ void scan_step(double &a, double &q) {
  const double one = 1.0;
  const int alo = __builtin_amdgcn_update_dpp(__double2loint(one), __double2loint(a), CTRL, ROW_MASK, 0xf, false);
  const int ahi = __builtin_amdgcn_update_dpp(__double2hiint(one), __double2hiint(a), CTRL, ROW_MASK, 0xf, false);
  const int qlo = __builtin_amdgcn_update_dpp(0, __double2loint(q), CTRL, ROW_MASK, 0xf, false);
  const int qhi = __builtin_amdgcn_update_dpp(0, __double2hiint(q), CTRL, ROW_MASK, 0xf, false);
  const double as = __hiloint2double(ahi, alo), qs = __hiloint2double(qhi, qlo);
  q = fma(a, qs, q);
  a = a * as;
}
__device__ __forceinline__ void affine_scan(double &a, double &q) {
  scan_step<0x111, 0xf>(a, q);
  scan_step<0x112, 0xf>(a, q);
  scan_step<0x114, 0xf>(a, q);
  scan_step<0x118, 0xf>(a, q);
  scan_step<0x142, 0xa>(a, q);
  scan_step<0x143, 0xc>(a, q);
}

// A lane owns two neighbouring columns of a tail row, in the TOP lanes (where the reversed shift
// registers deliver the last wavefront row): lane l owns columns 2 (l - L0), 2 (l - L0) + 1, L0 = 64 - NR / 2.
template <int NR>
__device__ __forceinline__ int tail_col(int lane, int k) { return 2 * (lane - (64 - NR / 2)) + k; }

// One Gauss-Seidel pass over the tail rows; returns the lane's max |delta|.  tv: the lane's cells;
// U0 / U1: the row above; the first tail row also goes to LDS (lane 63's lower neighbours during
// the next sweep).  tset[t]: LDS offsets of the two cells' coefficient sets (low / high half).
template <int NR>
__device__ __forceinline__ double tail_pass(int T, bool active, double *tE0c, double U0, double U1,
                                            double (&tv)[kTailMax][2], const int (&tset)[kTailMax],
                                            const double (&At)[kTailMax][2]) {
  static_assert(NR % 2 == 0 && NR <= 128, "two columns per lane");
  double dmax = 0.0;
#pragma unroll
  for (int t = 0; t < kTailMax; ++t) {
    if (t < T) {
      const lds_d2 s0 = (lds_d2)(unsigned)(tset[t] & 0xffff), s1 = (lds_d2)((unsigned)tset[t] >> 16);
      const d2 ud0 = s0[0], lr0 = s0[1], ud1 = s1[0], lr1 = s1[1];
      const double old0 = tv[t][0], old1 = tv[t][1];
      const double R1 = wave_shift1<0x130, false>(old0, 0.0); // the next lane's first column (not yet updated)
      const double D0 = t + 1 < T ? tv[t + 1 < kTailMax ? t + 1 : t][0] : 0.0, D1 = t + 1 < T ? tv[t + 1 < kTailMax ? t + 1 : t][1] : 0.0;
      const double q0 = fma(ud0.x, U0, fma(lr0.y, old1, fma(ud0.y, D0, At[t][0])));
      const double q1 = fma(ud1.x, U1, fma(lr1.y, R1, fma(ud1.y, D1, At[t][1])));
      double a = lr1.x * lr0.x, Q = fma(lr1.x, q0, q1);
      affine_scan(a, Q);
      const double xl = wave_shift1<0x138, false>(Q, 0.0);
      const double x0 = fma(lr0.x, xl, q0);
      if (active) {
        dmax = fmax(dmax, fmax(fabs(x0 - old0), fabs(Q - old1)));
        tv[t][0] = x0;
        tv[t][1] = Q;
        if (t == 0) *(d2 *)tE0c = d2{x0, Q};
      }
      U0 = x0;
      U1 = Q;
    }
  }
  return dmax;
}

// ---------------------------------------------------------------- tail rows, static scan (step_roll.hip)
// The multiplicative halves of the scan's composed maps -- the `a` of affine_scan at every level -- are
// products of bL along the row: static per floor plan.  The planner runs that half on the host (the same
// multiplications in the same order: bit-identical to affine_scan) and ships, per tail row and lane, the
// value of `a` ENTERING levels 1..5 (level 0's is lr1.x * lr0.x, formed here from the coefficients the pass
// reads anyway); 0 where a level has no source lane for the lane.  A level is then two DPP moves and one
// FMA (affine_scan: four moves, four more that set up their `old` operands, an FMA and a multiplication).
// A lane without a source reads 0 (bound_ctrl) or, in the rows a broadcast level's row_mask leaves out,
// keeps the previous level's shifted value -- finite, times 0.
// LDS layout of the table (shared by the workgroup), n = NR / 2 lanes own tail columns, l' = lane - (64 - n):
//   [t < T][d2 (A1, A2)][n] [d2 (A3, A4)][n]   then   [d2 (A5 of row 0, A5 of row 1)][n]
constexpr int tail_mul_doubles(int NR, int T) { return T > 0 ? (4 * T + 2) * (NR / 2) : 0; }

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ void scan_level(double &q, double A, double &qs) {
  const int lo = __builtin_amdgcn_update_dpp(__double2loint(qs), __double2loint(q), CTRL, ROW_MASK, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(__double2hiint(qs), __double2hiint(q), CTRL, ROW_MASK, 0xf, true);
  qs = __hiloint2double(hi, lo);
  q = fma(A, qs, q);
}

// tm: the lane's entry of the table's first array (table + 2 l').  Lanes that own no tail column (pad coefficient
// set, A = 0, cells 0) compute zeros; their tE0c points at the zero guards in front of the row.
// (Tried: the second row's LDS reads issued before the first row's scan, so that their latency hides behind it --
// 26 more registers live across the scan, and k_sweep_roll<96> has none to spare: scratch.)
template <int NR>
__device__ __forceinline__ double tail_pass_static(int T, const double *tm, double *tE0c, double U0, double U1,
                                                   double (&tv)[kTailMax][2], const int (&tset)[kTailMax],
                                                   const double (&At)[kTailMax][2]) {
  static_assert(NR % 2 == 0 && NR <= 128, "two columns per lane");
  constexpr int n = NR / 2;
  double dmax = 0.0;
#pragma unroll
  for (int t = 0; t < kTailMax; ++t) {
    if (t < T) {
      const lds_d2 s0 = (lds_d2)(unsigned)(tset[t] & 0xffff), s1 = (lds_d2)((unsigned)tset[t] >> 16);
      const d2 ud0 = s0[0], lr0 = s0[1], ud1 = s1[0], lr1 = s1[1];
      const d2 A12 = *(const d2 *)(tm + 4 * n * t), A34 = *(const d2 *)(tm + 4 * n * t + 2 * n);
      const double old0 = tv[t][0], old1 = tv[t][1];
      const double R1 = wave_shift1<0x130, false>(old0, 0.0); // the next lane's first column (not yet updated)
      double q0 = At[t][0], q1 = At[t][1];
      if (t + 1 < T) { // the row below (not yet updated); association order as in tail_pass
        q0 = fma(ud0.y, tv[t + 1 < kTailMax ? t + 1 : t][0], q0);
        q1 = fma(ud1.y, tv[t + 1 < kTailMax ? t + 1 : t][1], q1);
      }
      q0 = fma(ud0.x, U0, fma(lr0.y, old1, q0));
      q1 = fma(ud1.x, U1, fma(lr1.y, R1, q1));
      double Q = fma(lr1.x, q0, q1), qs = 0.0;
      scan_level<0x111, 0xf>(Q, lr1.x * lr0.x, qs);
      scan_level<0x112, 0xf>(Q, A12.x, qs);
      scan_level<0x114, 0xf>(Q, A12.y, qs);
      scan_level<0x118, 0xf>(Q, A34.x, qs);
      scan_level<0x142, 0xa>(Q, A34.y, qs);
      scan_level<0x143, 0xc>(Q, tm[4 * n * T + t], qs);
      const double xl = wave_shift1<0x138, false>(Q, 0.0);
      const double x0 = fma(lr0.x, xl, q0);
      dmax = fmax(dmax, fmax(fabs(x0 - old0), fabs(Q - old1)));
      tv[t][0] = x0;
      tv[t][1] = Q;
      if (t == 0) *(d2 *)tE0c = d2{x0, Q};
      U0 = x0;
      U1 = Q;
    }
  }
  return dmax;
}

} // namespace sweep
} // namespace sb
