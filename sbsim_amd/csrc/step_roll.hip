// step_roll.hip -- the sweep kernel for floor plans of up to 66 rows inside their exterior ring
// (R9: 66 x 96; below 64 rows the other lanes own pad rows; NR = 64, 72, 80, 88 or 96 slots by the plan's
// width): one wavefront per building, the grid in registers, rows 64.. finished by a scan,
// consecutive Gauss-Seidel sweeps overlapped ("rolling" sweeps).  simulator.py:278-371.
//
// Layout and schedule are those of step_reg.hip (lane l owns row l, column c in register slot
// (c + l) mod NR, every lane works on slot s mod NR at step s, neighbours by DPP) with all 64
// lanes owning a row, so that lanes 0..j START sweep k+1 during steps j = 0..62 of a period
// while lanes j+1..63 FINISH sweep k: a sweep costs NR steps.  Whether sweep k was the last one
// (simulator.py:360) is known only after step 62 and the tail scan, so the first steps of sweep
// k+1 are speculative: every window step keeps a copy of the value it overwrites and the last
// period restores lanes <= j from the copies.
//
// What this file is built around (tools/ubench_issue.hip, one wavefront per SIMD on gfx950):
// every instruction of a lone wavefront -- VALU of any width, DPP, SALU, LDS, s_waitcnt, s_nop,
// dependent or not -- costs one ~4.35-cycle issue slot, and the four wavefronts of a CU share one
// LDS pipe (ds_read_b64 2 cycles, ds_read_b128 4, ds_read2_b64 8, ds_write_b64 6).  So the
// step is written for the fewest instructions and few LDS cycles:
//   * coefficients as array-of-structs [set][bU bD | bL bR]: two ds_read_b128 (8 LDS cycles;
//     the struct-of-arrays table took two ds_read2_b64 = 16, which alone made the step LDS-bound);
//     cells with equal coefficients share a set (R9: 9 sets for 20 classes): fewer bank conflicts
//   * steps go in pairs: A = ap*Tprev + g of two neighbouring slots and the two seam values under
//     lane 63 are one ds_read_b128 each (A rows are stored rotated by one slot so that the pair of
//     an odd step is 16-byte aligned; row stride 70 doubles: conflict-free), one s_waitcnt per pair
//   * four buildings = four wavefronts share a workgroup, and with it ONE table of the steps' coefficient
//     sets in LDS (a byte per step and lane, 6 KB): a ds_read_b64 per eight steps.  (Read from global
//     memory, a word per four steps, the class words cost 11 % of the step: a load is three issue slots,
//     and behind another wavefront's hand-over in the CU's one vector-memory queue an L1 hit takes
//     longer than the twelve steps the reads ran ahead -- tools/exp_fixed_sweeps.py, LABNOTES.md 5.2)
//   * window steps route |delta| to the accumulator of the lane's own sweep by its SIGN: a lane
//     mask that one DPP shift per step maintains is OR-ed into the high word, one running max
//     (sweep k) and one running min (sweep k+1) -- no lane compare, no select
//     (the float64 instantiation; the library's kernel carries the high word of |delta| in a TRAVELLING maximum: struct Acc)
// Plain step: 14.5 issue slots (was 16), window step 19.5 (was 22), 12 LDS cycles (was 20).
//
// Around the sweeps (round 5; the order of a building's memory operations decides what its wavefront waits for):
//   * nothing at the top of a building reads a memory result: the draw of the next building (an atomic) is READ after the
//     sweeps, and the compiler's atomic optimizer -- which turns the one-lane atomic into a wave reduction that reads the
//     result at once, i.e. waits for every load in flight = the next building's rows -- is off (sbsim_amd/build.py);
//   * the next building's small inputs and class bytes are asked for BEFORE its rows (memory operations return in order);
//   * A = ap*Tprev + g has no pass of its own: a slot's A is formed during the ramp-up of the first sweep, a few slots ahead
//     of the step that reads it (struct APass), so the ramp-up starts on the first rows while the last are in flight;
//   * the tail rows' scan multiplies by STATIC factors (the multiplicative halves of its composed maps are products of bL:
//     the planner runs that half once, sweep_common.h tail_pass_static): a level is two DPP moves and one FMA;
//   * the stopping decision of the library's kernel is taken on high words throughout (rows, tail cells, ring): one 32-bit
//     wave maximum; what 32 bits cannot decide goes to the float64 instantiation through the redo list, as before;
//   * zone sums: a row-major scratch [lane][zone] with ONE BYTE per slot (hand_over);
//   * no kernel-lifetime per-lane pointers or converted constants (each one the register allocator parks in scratch costs a
//     scratch reload = a wait for every load in flight): per-lane addresses are formed where they are used (opaque()).
// Policy loop (two sweeps per step): 3.73e8 -> 4.45e8 zone-updates/s; the driver's window: frac 0.199 -> 0.208.
#include "sweep_common.h"

namespace sb {
namespace {

using namespace sweep;

#ifndef SB_ONE_WAIT
#define SB_ONE_WAIT 1
#endif
constexpr int kWin = 63;      // steps of a period in which the lanes are in two different sweeps
constexpr int kTS = 32;       // entries of the per-class tables (ap, g) and of the coefficient-set table
constexpr int kSeamPad = 8;

// Slots of A = ap*Tprev + g kept in LDS (the rest: registers).  72 of 96: a building needs 38.2 KB of
// LDS, so four buildings -- one per SIMD -- and the shared tables (7 KB) fill a CU's 160 KB.  Rows of
// 70 slots (the stride is 2 mod 4 doubles: rows are 16-byte aligned and 16 lanes' ds_read_b128 cover
// all banks) and a second array [64][2] with slots 70, 71 (a stride of 72 would be four-way conflicted,
// 74 does not fit).  NR = 72, 80, 88: the same 70 + 2 slots in LDS (NR - 72 in registers); NR = 64: all of them,
// rows of 66.
constexpr int lds_slots(int NR) { return NR >= 72 ? 72 : ((NR / 2) % 2 ? NR : NR + 2); }
constexpr int a_stride(int NR) { return NR >= 72 ? 70 : ((NR / 2) % 2 ? NR : NR + 2); }
constexpr int kWaves = 4;     // wavefronts = buildings per workgroup
constexpr int tail_row(int NR) { return NR + 4; } // tail rows in LDS: column c at [2 + c], zero guards around

struct PairBuf { // LDS values of two consecutive steps
  d2 ud0, lr0, ud1, lr1; // (bU, bD), (bL, bR)
  d2 A;                  // A of the two slots
  d2 sm;                 // old values of the first tail row under lane 63
};

struct Acc {
  // EXACT: float64 maxima, routed by sign in the window steps
  double cur; // max |delta| of the sweep the lanes are finishing
  double neg; // -(max |delta|) of the sweep the lanes have started (window steps)
  int sg;     // 0x80000000 in the lanes that have started the next sweep
  // !EXACT: the TRAVELLING maximum.  Column c of a sweep is worked on by lane l at step c + l (+ sweep * NR):
  // an accumulator that moves one lane down per step stays in ONE column of ONE sweep, enters lane 0 as 0
  // and leaves lane 63 as that column's max |delta| -- no lane ever mixes two sweeps, so the window steps
  // need no routing.  It carries the HIGH WORD of |delta| (as a float: the order of positive doubles is the
  // order of their high words is the order of those words read as floats), so moving and accumulating is
  // one v_max_f32_dpp; lane 63 collects what leaves in `fin` (its sweep ends exactly at the period's end).
  // 20 mantissa bits decide every sweep whose max |delta| is not within 6e-8 K of the threshold; the rest
  // (high words equal) are handed to the EXACT instantiation (k_sweep_roll<NR, true>, the redo list).
  int tr, fin;
  // Row 63's new values on their way to the tail scan (lanes = columns there): lane 63's result of
  // the previous step reaches lane 0 as its "upper neighbour" (wave_ror; row 0's bU is 0), from
  // where two shift registers -- even / odd columns -- carry it one lane further per insertion.
  double sre, sro;
};


template <int NR>
struct Ctx {
  const double *Arow;  // the lane's row of A, rotated: slot j at position (j + 1) mod NR; positions < a_stride here,
  const double *Aext;  // positions a_stride .. lds_slots - 1 here
  const double *seam;  // first tail row by step: the value under lane 63 at step s is seam[s]
  const unsigned long long *cw; // the lane's column of the class words in LDS: word k at [64 * k]
  unsigned long long w, wn;     // class word (8 steps) in use / the next one, in flight
#ifdef SB_EXP_NOCOEF
  d2 kc;
#endif
};

// Class words hold one byte per step: the step's coefficient set (its LDS byte offset is set * 32; the
// table sits at LDS address 0), eight steps per 64-bit word, NR / 8 words per lane in LDS.  The set of a
// step depends on (lane, step mod NR) only: word k covers the steps = 8k .. 8k+7 (mod NR) of the ramp-up
// and of every period; it is read one word (eight steps) ahead.
template <int NR, int S>
__device__ __forceinline__ lds_d2 step_set(Ctx<NR> &x) {
  constexpr int j = S % NR;
  static_assert(NR % 8 == 0, "whole class words per period");
  if constexpr (j % 8 == 0) {
    x.w = x.wn;
    x.wn = x.cw[((j / 8 + 1) % (NR / 8)) * 64];
  }
  const unsigned h = j % 8 < 4 ? (unsigned)x.w : (unsigned)(x.w >> 32);
  unsigned off; // (byte j % 4 of h) << 5 in one instruction
  if constexpr (j % 4 == 0) asm("v_lshlrev_b32_sdwa %0, 5, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(off) : "v"(h));
  else if constexpr (j % 4 == 1) asm("v_lshlrev_b32_sdwa %0, 5, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(off) : "v"(h));
  else if constexpr (j % 4 == 2) asm("v_lshlrev_b32_sdwa %0, 5, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(off) : "v"(h));
  else asm("v_lshlrev_b32_sdwa %0, 5, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(off) : "v"(h));
  return (lds_d2)off;
}

// LDS reads of the pair (S, S + 1), first half: coefficients of step S, both A values, both
// seam values.  Position of slot j in the lane's A row: (j + 1) mod NR -- even for odd S.
template <int NR, int S, bool SEAM, int NAR>
__device__ __forceinline__ void load_first(PairBuf &p, Ctx<NR> &x, const double (&Areg)[NAR]) {
#ifdef SB_EXP_NOCOEF // timing experiment: coefficients from registers, no table reads
  p.ud0 = p.lr0 = x.kc;
  const lds_d2 ct = nullptr;
#else
  const lds_d2 ct = step_set<NR, S>(x);
  p.ud0 = ct[0];
#endif
#if defined(SB_EXP_ONECOEF)
  p.lr0 = p.ud0;
#elif !defined(SB_EXP_NOCOEF)
  p.lr0 = ct[1];
#endif
  constexpr int q = (S + 1) % NR, NL = lds_slots(NR), AS = a_stride(NR);
  static_assert(q % 2 == 0 && AS % 2 == 0, "pairs start at odd steps");
  if constexpr (q < AS) p.A = *(const d2 *)(x.Arow + q);
  else if constexpr (q < NL) p.A = *(const d2 *)(x.Aext + (q - AS));
  else p.A = d2{Areg[q - NL], Areg[q + 1 - NL]};
  if constexpr (SEAM && S >= kWin) p.sm = *(const d2 *)(x.seam + S);
  else p.sm = d2{0.0, 0.0};
}
template <int NR, int S>
__device__ __forceinline__ void load_second(PairBuf &p, Ctx<NR> &x) {
#ifdef SB_EXP_NOCOEF
  p.ud1 = p.lr1 = x.kc;
#else
  const lds_d2 ct = step_set<NR, S>(x);
  p.ud1 = ct[0];
#ifdef SB_EXP_ONECOEF // timing experiment: one coefficient read per step
  p.lr1 = p.ud1;
#else
  p.lr1 = ct[1];
#endif
#endif
}

// One Gauss-Seidel update of every lane's current cell at step S of the overlapped schedule:
//   S < 63            ramp-up of a building's first sweep: lanes > S have not started
//   63 <= S < NR      all 64 lanes are in the same sweep
//   NR <= S < NR + 63 window: lanes <= S - NR are in the next sweep
// Association order of the four products as in step_reg.hip / step_lds.hip.
// |d|'s high word joins the travelling maximum (which moves one lane down); lane 63 collects what arrived
// with the PREVIOUS step first (the hazard recogniser puts a wait state between an inline-asm result and the
// next instruction if that reads it -- gfx950's dst_sel forwarding rule, assumed of every asm -- so the two
// instructions must not be producer and consumer back to back; the period's end collects the last step's)
__device__ __forceinline__ void track(Acc &acc, double d) {
#ifdef SB_EXP_NOTRACK // timing experiment: no max |delta|
  return;
#endif
  asm("v_max_f32 %0, %0, %1" : "+v"(acc.fin) : "v"(acc.tr));
  asm("v_max_f32_dpp %0, %0, |%1| wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(acc.tr) : "v"(__double2hiint(d)));
}

template <int NR, int S, bool EXACT>
__device__ __forceinline__ void step(double (&e)[NR], double (&bk)[kWin], d2 ud, d2 lr, double A,
                                     double sm, Acc &acc) {
  constexpr int r = S % NR, rm = (S + NR - 1) % NR, rp = (S + 1) % NR;
  const double Dn = wave_shift1<0x130, true>(e[rp], sm);
  // the chain starts in a register of its own (early clobber): accumulated in place in A's
  // register, the new value would have to be copied out of the ds_read_b128 tuple every step
  double t;
  asm("v_fma_f64 %0, %1, %2, %3" : "=&v"(t) : "v"(ud.y), "v"(Dn), "v"(A));
  t = fma(lr.y, e[rp], t);
  const double U = wave_shift1<0x13c, false>(e[rm], 0.0); // wave_ror:1: lane 0 sees row 63's latest value (times bU = 0)
  t = fma(lr.x, e[rm], t);
  const double nv = fma(ud.x, U, t);
#ifndef SB_EXP_NOSR // timing experiment: row 63 is not published
  if constexpr (S > kWin) { // lane 0's U is column S - 64 of row 63: into the shift register of its parity
    double &sr = (S - kWin - 1) % 2 == 0 ? acc.sre : acc.sro;
    sr = wave_shift1<0x138, true>(sr, U);
  }
#endif
  if constexpr (S < kWin) {
    const double sel = lanes_upto<S>() ? nv : e[r];
    if constexpr (EXACT) acc.cur = fmax(acc.cur, fabs(sel - e[r]));
    else track(acc, sel - e[r]);
    e[r] = sel;
  } else if constexpr (S < NR) {
    if constexpr (EXACT) acc.cur = fmax(acc.cur, fabs(nv - e[r]));
    else track(acc, nv - e[r]);
    e[r] = nv;
  } else if constexpr (!EXACT) {
    constexpr int J = S - NR;
    track(acc, nv - e[r]);
#ifndef SB_EXP_NOCOPY // timing experiment: the window keeps no copies
    bk[J] = e[r];
#endif
    e[r] = nv;
  } else {
    constexpr int J = S - NR;
    const double d = nv - e[r];
    bk[J] = e[r];
    e[r] = nv;
    // +|d| in the lanes still in sweep k, -|d| in the lanes already in sweep k+1
    const double sd = __hiloint2double((__double2hiint(d) & 0x7fffffff) | acc.sg, __double2loint(d));
    acc.cur = fmax(acc.cur, sd);
    acc.neg = fmin(acc.neg, sd);
    if constexpr (J + 1 < kWin) // lane J + 1 starts its next sweep at the next step (lane 0 keeps its bit)
      acc.sg = __builtin_amdgcn_update_dpp(acc.sg, acc.sg, 0x138, 0xf, 0xf, false);
    // here, not after the window: e[J] and bk[J] both survive the window
    asm volatile("" : "+v"(acc.cur), "+v"(acc.neg));
  }
}

// A = ap * Tprev + g, slot by slot (slot j goes to position (j + 1) mod NR of the lane's rotated row: LDS, or a register
// for the positions beyond lds_slots).  Per cell: one add that turns the class byte (class * 8: the byte offset into ap[]
// and into g[]) into the table address, one ds_read2_b64, one FMA into a register of its own (accumulated in place in the
// ds_read2 tuple, the value would have to be copied out for the write), half a ds_write2_b64.
// There is no pass of its own for this: a slot's A is formed during the RAMP-UP of the building's first sweep (a few slots
// before the ramp, three per pair of steps) -- slot j is first read two pairs before step j, and until step j every lane's
// slot j still holds Tprev.  The slots come from HBM in this order (the previous building's hand-over loaded them), so the
// ramp-up starts on the first ones while the last are still in flight: a pass over all 96 before the first step waited for
// the LAST load (~2 k cycles per building, tools/prof_sweeps.py).
constexpr int kA0 = 8; // slots whose A is formed before step 0
template <int NR, int NAR>
struct APass {
  const unsigned long long (&cw)[(NR + 7) / 8]; // class bytes by slot, eight per word
  unsigned tbase;                               // LDS address of ap[kTS] | g[kTS]
  double *Aw, *Ax;                              // positions < a_stride / the rest of the lane's rotated row
  double (&Areg)[NAR];
  template <int J>
  __device__ __forceinline__ d2 fetch() const {
    static_assert(kTS * 8 == 256, "a class byte (class * 8) is the byte offset into ap[] and into g[]");
    const unsigned h = (J & 7) < 4 ? (unsigned)cw[J >> 3] : (unsigned)(cw[J >> 3] >> 32);
    unsigned ad; // table base + byte (J & 3) of h: one instruction
    if constexpr ((J & 3) == 0) asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(ad) : "s"(tbase), "v"(h));
    else if constexpr ((J & 3) == 1) asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(ad) : "s"(tbase), "v"(h));
    else if constexpr ((J & 3) == 2) asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(ad) : "s"(tbase), "v"(h));
    else asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(ad) : "s"(tbase), "v"(h));
    const __attribute__((address_space(3))) double *tp = (const __attribute__((address_space(3))) double *)ad;
    return d2{tp[0], tp[kTS]}; // (ap, g): ds_read2_b64 offset1:32
  }
  template <int J>
  __device__ __forceinline__ void put(const d2 &pg, const double &ej) const {
    constexpr int q = (J + 1) % NR, NL = lds_slots(NR), AS = a_stride(NR);
    double av;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(av) : "v"(pg.x), "v"(ej), "v"(pg.y));
    if constexpr (q < AS) Aw[q] = av;
    else if constexpr (q < NL) Ax[q - AS] = av;
    else Areg[q - NL] = av;
  }
};

// Pairs S, S + 2, .. < S1 (S odd).  While pair S runs, the LDS reads of pair S + 2 * kDepth are
// issued in two halves; in the main loop (WRAP) the last kDepth pairs read ahead for the next
// period's first pairs (whose seam values the caller reads after the tail scan).
#ifndef SB_DEPTH
#define SB_DEPTH 1
#endif
constexpr int kDepth = SB_DEPTH;          // pairs between the LDS reads of a step and its arithmetic
constexpr int kBufs = kDepth + 1;         // NR / 2 pairs per period: kBufs must divide that (checked per instantiation in k_sweep_roll)
constexpr int pair_buf(int S) { return ((S - 1) / 2) % kBufs; }

template <int NR, int S, int S1, bool WRAP, bool EXACT, bool APASS, int NAR>
__device__ __forceinline__ void roll_pairs(double (&e)[NR], double (&bk)[kWin], double (&Areg)[NAR],
                                           PairBuf (&pb)[kBufs], Ctx<NR> &x, Acc &acc, const APass<NR, NAR> &ap) {
  if constexpr (S < S1) {
    constexpr bool wrapped = WRAP && S + 2 * kDepth >= S1;
    constexpr int N = wrapped ? S + 2 * kDepth - NR : S + 2 * kDepth;
    PairBuf &cur = pb[pair_buf(S)], &nxt = pb[pair_buf(N)];
#if SB_ONE_WAIT // one s_waitcnt per pair: every LDS value of the pair is asked for here, before the next pair's reads are issued
    asm volatile("" ::"v"(cur.ud0.x), "v"(cur.A.x), "v"(cur.sm.x), "v"(cur.ud1.x), "v"(cur.lr1.x));
    __builtin_amdgcn_sched_barrier(0);
#endif
    load_first<NR, N, !wrapped>(nxt, x, Areg);
    // the ramp-up forms A of three more slots per pair (APass): the table reads here, the values after the pair's steps
    constexpr int j0 = kA0 + 3 * ((S - 1) / 2);
    static_assert(!APASS || (S % 2 == 1 && j0 >= S + 2 * kDepth + 2), "a slot's A is written before the pair that reads it ahead");
    d2 pa0 = d2{0.0, 0.0}, pa1 = d2{0.0, 0.0}, pa2 = d2{0.0, 0.0};
    if constexpr (APASS && j0 < NR) pa0 = ap.template fetch<j0 < NR ? j0 : 0>();
    if constexpr (APASS && j0 + 1 < NR) pa1 = ap.template fetch<j0 + 1 < NR ? j0 + 1 : 0>();
    if constexpr (APASS && j0 + 2 < NR) pa2 = ap.template fetch<j0 + 2 < NR ? j0 + 2 : 0>();
    __builtin_amdgcn_sched_barrier(0);
    step<NR, S, EXACT>(e, bk, cur.ud0, cur.lr0, cur.A.x, cur.sm.x, acc);
    __builtin_amdgcn_sched_barrier(0);
    load_second<NR, N + 1>(nxt, x);
    __builtin_amdgcn_sched_barrier(0);
    step<NR, S + 1, EXACT>(e, bk, cur.ud1, cur.lr1, cur.A.y, cur.sm.y, acc);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (APASS && j0 < NR) ap.template put<j0 < NR ? j0 : 0>(pa0, e[j0 < NR ? j0 : 0]);
    if constexpr (APASS && j0 + 1 < NR) ap.template put<j0 + 1 < NR ? j0 + 1 : 0>(pa1, e[j0 + 1 < NR ? j0 + 1 : 0]);
    if constexpr (APASS && j0 + 2 < NR) ap.template put<j0 + 2 < NR ? j0 + 2 : 0>(pa2, e[j0 + 2 < NR ? j0 + 2 : 0]);
    if constexpr (APASS) __builtin_amdgcn_sched_barrier(0);
    roll_pairs<NR, S + 2, S1, WRAP, EXACT, APASS>(e, bk, Areg, pb, x, acc, ap);
  }
}

// The end of a building's step, slot by slot: undo the started sweep (lanes <= J restore slot J
// from the window's copies), store the slot, add it to its zone sum (LDS), load the same slot of
// the next building.  The loop runs at the pace of the memory pipe; the restores cost nothing there.
// Zone sums: the scratch (it aliases A) is [row = lane][ZRS] -- a lane's sums of the zones its row crosses, zone-minor, ZRS odd
// (a half-wavefront's ds_add_f64 to one zone covers all banks) -- so a slot's destination is the lane's row + zone * 8: ONE BYTE
// per slot (zone <= 31 on this kernel: at most 32 cell classes), eight slots per 64-bit word, NR / 8 words per lane, all read
// before the loop.  (Until round 4: [zone][row] with 16-bit offsets, NR / 4 words -- 48 registers, a third of them read inside the
// loop behind the next building's row loads.  Memory operations return in order, so a word read inside the loop waits for the
// rows issued before it, i.e. for HBM.)
// HBM state layout: [NR / 2][64 lanes][2] -- a lane's two neighbouring slots are 16 bytes, so the
// loop moves a building with NR / 2 stores and NR / 2 loads of 16 bytes per lane (a wavefront has
// at most 63 memory operations in flight: the count of operations, not their size, paces the loop).
template <int NR, int J>
__device__ __forceinline__ void hand_over(double (&e)[NR], const double (&bk)[kWin], const unsigned long long (&zw)[NR / 8],
                                          double *tp, const double *np_, double *zrow) {
  static_assert(NR % 8 == 0, "whole zone words");
  if constexpr (J < NR) {
    if constexpr (J < kWin) e[J] = lanes_upto<J>() ? bk[J] : e[J];
    if constexpr (J + 1 < kWin) e[J + 1] = lanes_upto<J + 1>() ? bk[J + 1] : e[J + 1];
    const unsigned w = (J & 7) < 4 ? (unsigned)zw[J / 8] : (unsigned)(zw[J / 8] >> 32);
    const unsigned off0 = (w >> (8 * (J & 3))) & 0xffu, off1 = (w >> (8 * ((J + 1) & 3))) & 0xffu;
#ifndef SB_EXP_NOMEM // timing experiment: the hand-over without its HBM traffic
    __builtin_nontemporal_store(d2{e[J], e[J + 1]}, (d2 *)(tp + J * 64)); // the state streams: read once, written once per launch
#endif
    __hip_atomic_fetch_add((double *)((char *)zrow + off0), e[J], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_fetch_add((double *)((char *)zrow + off1), e[J + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#ifndef SB_EXP_NOMEM
    const d2 nv = __builtin_nontemporal_load((const d2 *)(np_ + J * 64));
    e[J] = nv.x;
    e[J + 1] = nv.y;
#endif
    if constexpr ((J & 7) == 6) __builtin_amdgcn_sched_barrier(0);
    hand_over<NR, J + 2>(e, bk, zw, tp, np_, zrow);
  }
}

extern __shared__ __attribute__((aligned(16))) double lds[];

// EXACT = false: the library's kernel (travelling max |delta| on high words; a sweep whose decision those 32
// bits cannot make puts its building on the redo list and leaves its state alone).  EXACT = true: float64
// maxima; launched after the fast kernel on the redo list (a.redo_mode: building numbers come from the list),
// or as the only kernel (SBSIM_ROLL_EXACT=1: the cross-check).
template <int NR, bool EXACT>
__global__ void __launch_bounds__(64 * kWaves) __attribute__((amdgpu_waves_per_eu(1, 1))) k_sweep_roll(Dev a) {
  const int nB = a.redo_mode ? __builtin_amdgcn_readfirstlane(a.redo_ctr[0]) : a.B; // buildings of this launch
  if (nB == 0) return;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6); // one wavefront per SIMD, each with its own buildings
  const int gw = (int)blockIdx.x * kWaves + wave;
  constexpr int kASlots = (NR + 7) / 8, kZSlots = NR / 8, kCW = NR / 8;
  static_assert((NR / 2) % kBufs == 0, "pair buffers rotate through a whole period: pair_buf() must not alias across the wrap");
  constexpr int kNL = lds_slots(NR), kAS = a_stride(NR), kNAR = NR - kNL > 0 ? NR - kNL : 2;

  // shared by the workgroup's wavefronts (read-only after this block):
  double *tabc = lds;                      // [kTS][4]: bU bD bL bR per coefficient set
  unsigned long long *ctab8 = (unsigned long long *)(lds + a.r_cmap); // [kCW][64]: the steps' coefficient sets, a byte each
  // the wavefront's own region:
  double *tmul = lds + a.r_cmap + kCW * 64; // the tail scan's static multipliers (sweep_common.h, tail_pass_static)
  double *wl = tmul + a.tmul_doubles + (size_t)wave * a.lds_wave_doubles;
  double *tapg = wl;                       // [2][kTS]: ap by class, then g by class (g of this building); a class byte of the A pass (class * 8) IS the
                                           // byte offset into either: one add (SDWA byte select) forms the address, one ds_read2_b64 fetches both
  double *tE0 = wl + a.r_seam + 2;         // the first tail row by column ([2 guards | NR | 2 guards]): lane 63's lower neighbours
  double *A = wl + a.r_A;                  // [64][kAS]; after the sweeps: zone sums [64 rows][ZRS]
  // every byte of LDS starts finite: reads next to the arrays' ends are multiplied by 0
  for (int i = threadIdx.x; i < a.lds_reg_bytes / 8; i += blockDim.x) lds[i] = 0.0;
  __syncthreads();
  for (int i = threadIdx.x; i < 4 * kTS; i += blockDim.x) tabc[i] = i < 4 * a.ncset ? a.csetab[i] : 0.0;
  for (int i = threadIdx.x; i < kCW * 64; i += blockDim.x) ctab8[i] = a.cmapS[i];
  for (int i = threadIdx.x; i < a.tmul_doubles; i += blockDim.x) tmul[i] = a.tmulS[i];
  if (lane < kTS) tapg[lane] = lane <= a.ncls ? a.ctab[lane * 8 + 4] : 0.0; // row `ncls` is the pad class
  __syncthreads(); // the only barriers: from here on the wavefronts go their own ways

  const sb_params &p = a.p;
  const int R = lane;
  constexpr int RS = 64;
  if ((unsigned)(size_t)(__attribute__((address_space(3))) double *)tabc != 0u) __builtin_trap(); // class words hold LDS addresses
  Ctx<NR> x;
  x.Arow = A + (size_t)R * kAS;
  x.Aext = A + 64 * kAS + (size_t)R * (kNL - kAS);
  x.seam = tE0 - kWin; // lane 63 works on column s - 63 at step s
  x.cw = ctab8 + lane;
  x.w = x.wn = 0;
#ifdef SB_EXP_NOCOEF
  x.kc = d2{0.2 + 1e-9 * opaque(lane), 0.2};
#endif
  // the lane's tail cells (static per floor plan): table offsets of their coefficient sets
  // (two 16-bit halves) and of their classes (two bytes)
  const bool tactive = tail_col<NR>(lane, 0) >= 0; // the lane owns two tail columns
  const int tc0 = tactive ? tail_col<NR>(lane, 0) : 0;
  // where the lane's cells of the first tail row go: column tc0; idle lanes write their zeros to the zero guards.
  // (an offset, not a pointer: per-lane pointers that live as long as the kernel end up in scratch)
  const int tcw = tactive ? tc0 : -2;
  int tset[kTailMax], tcls8[kTailMax];
#pragma unroll
  for (int t = 0; t < kTailMax; ++t) {
    tset[t] = (a.ncset - 1) * 32 * 0x10001; // the pad set (the table's last): no neighbour counts
    tcls8[t] = (8 * a.ncls) | ((8 * a.ncls) << 8);
    if (t < a.T && tactive) {
      const int c0 = tc0, c1 = tc0 + 1;
      tset[t] = ((int)a.tcset[t * NR + c0] << 2) | ((int)a.tcset[t * NR + c1] << 18); // set * 8 -> set * 32
      tcls8[t] = (int)a.tcls[t * NR + c0] | ((int)a.tcls[t * NR + c1] << 8);
    }
  }
  const unsigned long long *amap = a.amapS + lane;
  const unsigned long long *zmap = a.zmapS + lane;

// developer aid: cycle stamps of workgroup 0's 11th building (steady state, not the cold start)
// (compiled in with -DSB_PHASE_STAMPS only -- tools/prof_sweeps.py on a tools/build_variant.sh build: the tests'
// uniform branches were 1 % of the product kernel's instructions)
#ifdef SB_PHASE_STAMPS
#define SB_STAMP(i) do { if (a.dbg && gw == 0 && iter == 10 && lane == 0) a.dbg[i] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define SB_STAMP(i) do { } while (0)
#endif

  // The lane's row of the NEXT building is loaded while this building's row is stored, slot by
  // slot, so the loop never waits on HBM latency; so are the building's small inputs.
  double e[NR];
  double nx_g = 0.0, nx_tail[kTailMax][2] = {{0.0, 0.0}, {0.0, 0.0}}, nx_tnow = 0.0, nx_lo = 0.0, nx_hi = 0.0;
#define SB_LOAD_AUX(bb)                                                                         \
  do {                                                                                          \
    nx_tnow = a.bld[(bb)].t_now;                                                                \
    nx_lo = a.scal[(size_t)(bb) * kNScal + 16];                                                 \
    nx_hi = a.scal[(size_t)(bb) * kNScal + 17];                                                 \
    nx_g = a.gtabg[(size_t)(bb) * kTS + (opaque(lane) & (kTS - 1))]; /* (formed here: hoisted out of the building loop the per-lane address lives in scratch, and a scratch reload waits for every load in flight) */ \
    const double *tt_ = a.temp + (size_t)(bb) * a.state_doubles + NR * 64;                      \
    _Pragma("unroll") for (int t = 0; t < kTailMax; ++t)                                        \
      _Pragma("unroll") for (int k = 0; k < 2; ++k)                                             \
        if (t < a.T) nx_tail[t][k] = tt_[t * NR + tc0 + k];                                     \
  } while (0)
  // launch slot -> building: the redo launch walks the list the fast kernel wrote
  auto building = [&](int s) { return a.redo_mode ? __builtin_amdgcn_readfirstlane(a.redo_list[s]) : s; };
  const int b_first = gw < nB ? building(gw) : 0;
  if (gw < nB) {
    const double *tp_ = a.temp + (size_t)b_first * a.state_doubles;
#pragma unroll
    for (int j = 0; j < NR; j += 2) { // state layout [NR / 2][64][2]
      const d2 v = __builtin_nontemporal_load((const d2 *)(tp_ + j * 64 + 2 * R));
      e[j] = v.x;
      e[j + 1] = v.y;
    }
    SB_LOAD_AUX(b_first);
  }
  // class bytes by slot (static per lane): read from L2 for the NEXT building right after a building's hand-over (the zone
  // reduce and the set-up hide the latency), used by APass during the ramp-up
  unsigned long long amapw[kASlots];
#define SB_LOAD_AMAP(g0, g1)                                                                    \
  do {                                                                                          \
    const int o_ = opaque(0);                                                                   \
    _Pragma("unroll") for (int g = (g0); g < (g1); ++g) amapw[g] = amap[o_ + g * 64];           \
  } while (0)
  constexpr int kAHalf = kASlots / 2; // words asked for before the next building's rows; the others after them
  SB_LOAD_AMAP(0, kASlots);
  // Buildings need different numbers of sweeps: after its first building a workgroup draws the
  // next one from a device counter (zeroed before every launch).
  int iter = 0;
  for (int s = gw, sn = 0, b = b_first, bn = 0; s < nB; s = sn, b = bn, ++iter) {
    // the draw of the NEXT building: issued here, read before the hand-over (an atomic's round trip
    // to L2 is 1-2 us: the sweeps hide it)
    // (nothing may USE the result here -- not even an addition: the wait for it is a wait for every load in flight, i.e. for
    // the next building's rows; and the compiler's atomic optimizer, which turns the one-lane atomic into a wave reduction
    // that reads the result at once, is off for this library: sbsim_amd/build.py)
    int nb = 0;
    if (lane == 0) nb = atomicAdd(a.next_b, 1); // (the redo launch: its own counter and wavefront count)
    SB_STAMP(0);
    x.wn = x.cw[0]; // the first class word of the ramp-up
    __builtin_amdgcn_sched_barrier(0);
    const double t_now = nx_tnow;
    // exterior-space cells outside the trim box all become t_now in the first sweep
    // (simulator.py:256-258); their largest |delta| follows from their extreme values
    const double ring_d = a.n_ring > 0 ? fmax(fabs(t_now - nx_lo), fabs(t_now - nx_hi)) : 0.0;
    if (lane < kTS) tapg[kTS + opaque(lane)] = nx_g; // (the address formed here: see SB_LOAD_AUX)
    double tv[kTailMax][2]; // the lane's tail cells: current values
#pragma unroll
    for (int t = 0; t < kTailMax; ++t)
#pragma unroll
      for (int k = 0; k < 2; ++k) tv[t][k] = tactive ? nx_tail[t][k] : 0.0; // (idle lanes compute zeros in the tail pass)
    *(d2 *)(tE0 + opaque(tcw)) = d2{tv[0][0], tv[0][1]};
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_sched_barrier(0);
    SB_STAMP(1);
    double At[kTailMax][2] = {{0.0, 0.0}, {0.0, 0.0}}; // A of the lane's tail cells
    if constexpr (!EXACT) {
      // unconditional (a row beyond T and the lanes that own no tail column hold the pad class: ap = 1, g = 0, on cells that
      // are 0): four table reads in flight, then four FMAs -- a branch per cell is four LDS round trips one after the other
      const double *pg[kTailMax][2];
#pragma unroll
      for (int t = 0; t < kTailMax; ++t)
#pragma unroll
        for (int k = 0; k < 2; ++k) pg[t][k] = (const double *)((const char *)tapg + ((opaque(tcls8[t]) >> (8 * k)) & 0xff)); // (opaque: the addresses are formed here, not kept for the kernel's lifetime)
      double tap[kTailMax][2], tg[kTailMax][2];
#pragma unroll
      for (int t = 0; t < kTailMax; ++t)
#pragma unroll
        for (int k = 0; k < 2; ++k) { tap[t][k] = pg[t][k][0]; tg[t][k] = pg[t][k][kTS]; }
#pragma unroll
      for (int t = 0; t < kTailMax; ++t)
#pragma unroll
        for (int k = 0; k < 2; ++k) At[t][k] = fma(tap[t][k], tv[t][k], tg[t][k]);
    } else { // (the float64 instantiation has no registers for the four reads in flight: scratch)
#pragma unroll
      for (int t = 0; t < kTailMax; ++t)
#pragma unroll
        for (int k = 0; k < 2; ++k)
          if (t < a.T && tactive) {
            const int c8 = (opaque(tcls8[t]) >> (8 * k)) & 0xff;
            const double *pg = (const double *)((const char *)tapg + c8);
            At[t][k] = fma(pg[0], tv[t][k], pg[kTS]);
          }
    }

    // A = ap*Tprev + g (E = Tprev before the first sweep): the first kA0 slots here, the others during the ramp-up (APass)
    double Areg[kNAR];
    Areg[0] = 0.0;
    const APass<NR, kNAR> ap{amapw, (unsigned)(size_t)(__attribute__((address_space(3))) double *)tapg, A + (size_t)R * kAS,
                             A + 64 * kAS + (size_t)R * (kNL - kAS), Areg};
    {
      static_assert(kA0 == 8 && kA0 + 3 * ((kWin - 1) / 2) >= NR, "the ramp-up's pairs reach the last slot");
      const d2 g0 = ap.template fetch<0>(), g1 = ap.template fetch<1>(), g2 = ap.template fetch<2>(), g3 = ap.template fetch<3>(),
               g4 = ap.template fetch<4>(), g5 = ap.template fetch<5>(), g6 = ap.template fetch<6>(), g7 = ap.template fetch<7>();
      __builtin_amdgcn_sched_barrier(0);
      ap.template put<0>(g0, e[0]); ap.template put<1>(g1, e[1]); ap.template put<2>(g2, e[2]); ap.template put<3>(g3, e[3]);
      ap.template put<4>(g4, e[4]); ap.template put<5>(g5, e[5]); ap.template put<6>(g6, e[6]); ap.template put<7>(g7, e[7]);
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_sched_barrier(0);
    SB_STAMP(2);

    int n_sweeps = 0, converged = 0, redo = 0;
    double bk[kWin]; // the window's copies: what the started sweep overwrote
    {
      PairBuf pb[kBufs];
      Acc acc;
      acc.cur = 0.0; acc.neg = 0.0; acc.sg = lane == 0 ? (int)0x80000000 : 0;
      acc.tr = acc.fin = 0;
      acc.sre = acc.sro = 0.0;
      // step 0 (lane 0, column 0) on its own: pairs start at odd steps
      {
        const lds_d2 ct = step_set<NR, 0>(x);
        const d2 ud = ct[0], lr = ct[1];
        load_first<NR, 1, true>(pb[pair_buf(1)], x, Areg);
        load_second<NR, 2>(pb[pair_buf(1)], x);
        if constexpr (kDepth > 1) {
          load_first<NR, 3, true>(pb[pair_buf(3)], x, Areg);
          load_second<NR, 4>(pb[pair_buf(3)], x);
        }
        static_assert(kDepth <= 2, "initial fill of the pair buffers");
        step<NR, 0, EXACT>(e, bk, ud, lr, x.Arow[1], 0.0, acc);
      }
      __builtin_amdgcn_sched_barrier(0);
      roll_pairs<NR, 1, kWin, false, EXACT, true>(e, bk, Areg, pb, x, acc, ap); // ramp-up (+ A of the slots from kA0 on); reads ahead for the first pairs of the period
#ifndef SB_STAMP_NEXT
      SB_STAMP(15);
#endif
#pragma nounroll
      for (;;) { // simulator.py:348-368
        __builtin_amdgcn_sched_barrier(0);
#ifdef SB_PHASE_STAMPS
#ifndef SB_STAMP_SWEEP
#define SB_STAMP_SWEEP 1 // which sweep of the building the period's stamps are taken in (0: the first one, right after the ramp-up)
#endif
// (sweep_idx: the sweep's number at the period's top -- n_sweeps is incremented before the decision's stamp, which until round 6
// was therefore taken one sweep early: "decide" came out negative)
#define SB_STAMP2(i) do { if (a.dbg && gw == 0 && iter == 10 && sweep_idx == SB_STAMP_SWEEP && lane == 0) a.dbg[i] = (long long)__builtin_readcyclecounter(); } while (0)
        const int sweep_idx = n_sweeps;
#ifdef SB_STAMP_NEXT // [15] = the NEXT period's top instead of the ramp-up's end: [15] - [14] is the loop's back edge
        if (a.dbg && gw == 0 && iter == 10 && sweep_idx == SB_STAMP_SWEEP + 1 && lane == 0) a.dbg[15] = (long long)__builtin_readcyclecounter();
#endif
#else
#define SB_STAMP2(i) do { } while (0)
#endif
        SB_STAMP2(10);
        roll_pairs<NR, kWin, NR + kWin, true, EXACT, false>(e, bk, Areg, pb, x, acc, ap);
        SB_STAMP2(11);
        // row 63's last column (lane 63's result of the period's last step) enters its shift register;
        // then both are reversed: lane l holds columns 2 (l - L0), 2 (l - L0) + 1, the tail scan's layout
        double U0, U1;
        {
          constexpr int last = (NR + kWin - 1) % NR;
          const double ul = wave_shift1<0x13c, false>(e[last], 0.0);
          double &sr = (NR - 1) % 2 == 0 ? acc.sre : acc.sro;
          sr = wave_shift1<0x138, true>(sr, ul);
          const int rev = (63 - lane) * 4;
          U0 = __hiloint2double(__builtin_amdgcn_ds_bpermute(rev, __double2hiint(acc.sre)),
                                __builtin_amdgcn_ds_bpermute(rev, __double2loint(acc.sre)));
          U1 = __hiloint2double(__builtin_amdgcn_ds_bpermute(rev, __double2hiint(acc.sro)),
                                __builtin_amdgcn_ds_bpermute(rev, __double2loint(acc.sro)));
        }
        SB_STAMP2(12);
        const double dt_ = tail_pass_static<NR>(a.T, tmul + opaque(tc0), tE0 + opaque(tcw), U0, U1, tv, tset, At); // tmul + 2 l'

        SB_STAMP2(13);
        if constexpr (EXACT) {
          double md = wave_max(fmax(acc.cur, dt_));
          if (n_sweeps == 0) md = fmax(md, ring_d);
          ++n_sweeps;
          converged = md <= p.conv_threshold;
        } else {
          // the wavefront's rows: lane 63 has collected every column's maximum (high words); the tail cells' and the
          // ring's join them as high words too (non-negative doubles order like their high words read as integers)
          asm("v_max_f32 %0, %0, %1" : "+v"(acc.fin) : "v"(acc.tr)); // the period's last step
          int m_hi = max(__builtin_amdgcn_readlane(acc.fin, 63), wave_max_i32(__double2hiint(dt_)));
          if (n_sweeps == 0) m_hi = max(m_hi, __double2hiint(ring_d));
          const int thr_hi = __double2hiint(p.conv_threshold);
          ++n_sweeps;
          acc.fin = 0;
          acc.tr = lanes_upto<62>() ? acc.tr : 0; // lane 63's was this sweep's last column: the next step's collect must not see it again
          const bool undecided = m_hi == thr_hi || (a.dbg_redo_mod > 0 && b % a.dbg_redo_mod == 0); // (a test hook)
          if (m_hi <= thr_hi && undecided) { redo = 1; break; } // 32 bits cannot tell: the exact kernel takes the building
          converged = m_hi < thr_hi;
        }
        SB_STAMP2(14);
#ifdef SB_EXP_DESYNC // timing experiments: sweep counts 1..9 by building number (mean 5), whatever the numbers are
        if (n_sweeps >= 1 + (int)(((unsigned)b * 2654435761u >> 13) % 9u)) break;
#else
        if (converged || n_sweeps >= p.iter_limit) break; // the started sweep is undone while the row is stored
#endif
        // after the tail scan: the first tail row's new values under lane 63, for the pairs already read ahead
        pb[pair_buf(kWin)].sm = *(const d2 *)(x.seam + kWin);
        if constexpr (kDepth > 1) pb[pair_buf(kWin + 2)].sm = *(const d2 *)(x.seam + kWin + 2);
        acc.cur = -acc.neg;
        acc.neg = 0.0;
        acc.sg = lane == 0 ? (int)0x80000000 : 0;
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    SB_STAMP(3);
    sn = a.sweep_wgs + __builtin_amdgcn_readfirstlane(nb);
    bn = sn < nB ? building(sn) : 0;
    // the next building's small inputs and the class bytes: asked for BEFORE its rows (the hand-over below), because
    // memory operations return in order -- behind the rows, the set-up's first use of any of them waited for the last row
    // (the class bytes of the first half of the slots only: the ramp-up reaches the other half ~3 k cycles in, when the
    // rows have long arrived, and every value asked for here lives through the hand-over -- k_sweep_roll<96> has 28
    // registers for that, not 40)
    if (sn < nB) SB_LOAD_AUX(bn);
    SB_LOAD_AMAP(0, kAHalf);
    __builtin_amdgcn_sched_barrier(0);
    SB_STAMP(4);

    // grid back to HBM.  Zone sums (A is dead now): every lane adds its cells into its own row of zs[row][zone]; column Z
    // collects every cell outside a zone, so that the sum of all entries is the grid sum.
    double *zs = A;
    const int ZRS = a.ZRS;
    {
      unsigned long long zw[kZSlots]; // zone bytes, eight slots per word
      const unsigned long long *zm = zmap + opaque(0);
#pragma unroll
      for (int g = 0; g < kZSlots; ++g) zw[g] = zm[g * 64];
      __builtin_amdgcn_sched_barrier(0);
      // a building on its way to the redo list keeps its state: its rows go to the wavefront's scratch
      double *tp = redo ? a.redo_scratch + (size_t)gw * a.state_doubles : a.temp + (size_t)b * a.state_doubles;
      double *Ttail = tp + NR * 64; // [T][NR]
      double *zrow = zs + (size_t)opaque(R) * ZRS; // the lane's row of the scratch
      for (int z = 0; z <= a.Z; ++z) zrow[z] = 0.0;
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int t = 0; t < kTailMax; ++t) // the tail rows hold no zone cells (sb_create checks): all into column Z
        if (t < a.T && tactive) {
          *(d2 *)(Ttail + t * NR + tc0) = d2{tv[t][0], tv[t][1]};
          zrow[a.Z] += tv[t][0] + tv[t][1];
        }
      const double *np_ = a.temp + (size_t)(sn < nB ? bn : b) * a.state_doubles;
      static_assert(RS == 64, "hand_over: slot stride");
      hand_over<NR, 0>(e, bk, zw, tp + 2 * R, np_ + 2 * R, zrow);
    }
    __builtin_amdgcn_sched_barrier(0);
    SB_STAMP(5);
    SB_LOAD_AMAP(kAHalf, kASlots);
    __builtin_amdgcn_sched_barrier(0);
    SB_STAMP(6);
    __builtin_amdgcn_wave_barrier();
    SB_STAMP(7);

    { // hand the zone sums, the grid sum and the sweep count to k_post
      // 16 zones x 4 row groups per pass: lane (zone = lane >> 2, group = lane & 3) adds every fourth row of its zone; the
      // four groups of a zone sit in one quad of lanes, so two DPP quad permutes combine them (no LDS round trip)
      double gacc = 0.0;
      for (int zb = 0; zb <= a.Z; zb += 16) {
        const int zz = zb + (lane >> 2), g = lane & 3;
        const double *zr = zs + (size_t)g * ZRS + (zz <= a.Z ? zz : a.Z);
        double part[RS / 4];
#pragma unroll
        for (int k = 0; k < RS / 4; ++k) part[k] = zr[(size_t)4 * k * ZRS]; // sixteen independent reads, then one wait
        double v = 0.0;
#pragma unroll
        for (int k = 0; k < RS / 4; ++k) v += part[k];
        if (zz > a.Z) v = 0.0;
        v += dpp_or_zero<0xB1, 0xf>(v); // quad_perm [1, 0, 3, 2]: the neighbour in the pair
        v += dpp_or_zero<0x4E, 0xf>(v); // quad_perm [2, 3, 0, 1]: the other pair of the quad
        if (g == 0 && zz < a.Z && !redo) a.zsum[(size_t)b * a.Z + zz] = v;
        if (g == 0 && zz <= a.Z) gacc += v;
      }
      const double gsum = wave_sum(gacc);
      if (lane == 0 && !redo) {
        a.gsum[b] = gsum + a.n_ring_f64 * t_now;
        a.nsw[b] = n_sweeps | (converged << 16);
      }
      if (lane == 0 && redo) a.redo_list[atomicAdd(a.redo_ctr, 1)] = b;
      SB_STAMP(8);
#ifdef SB_PHASE_STAMPS
      if (a.dbg && gw == 0 && iter == 10 && lane == 0) a.dbg[9] = n_sweeps;
#endif
    }
  }
#undef SB_STAMP
#undef SB_LOAD_AUX
#undef SB_LOAD_AMAP
}

} // namespace

bool sweep_roll_supported(int NR) { return NR == 64 || NR == 72 || NR == 80 || NR == 88 || NR == 96; }
int sweep_roll_lds_slots(int NR) { return lds_slots(NR); }
int sweep_roll_a_stride(int NR) { return a_stride(NR); }
int sweep_roll_seam_doubles(int NR, int T) { (void)T; return tail_row(NR); } // the first tail row, by column
int sweep_roll_waves() { return kWaves; }
int sweep_roll_tail_mul_doubles(int NR, int T) { return tail_mul_doubles(NR, T); } // the tail scan's static multipliers (shared by the workgroup)

int sweep_roll_redo_workgroups() { return 8; } // the exact kernel's launch on the redo list (a handful of buildings per step at most)

namespace {

template <int NR>
int prepare(const Dev &d) {
  const int e = (int)hipFuncSetAttribute((const void *)k_sweep_roll<NR, false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                         d.lds_reg_bytes);
  if (e != (int)hipSuccess) return e;
  return (int)hipFuncSetAttribute((const void *)k_sweep_roll<NR, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  d.lds_reg_bytes);
}

// The fast kernel, then the exact one on whatever the fast one could not decide (it returns at once when the
// list is empty).  d.roll_exact (SBSIM_ROLL_EXACT=1): the exact kernel alone, on every building.
template <int NR>
int launch(const Dev &d, hipStream_t stream) {
  const dim3 grid((d.sweep_wgs + kWaves - 1) / kWaves), block(64 * kWaves);
  if (d.roll_exact) {
    hipLaunchKernelGGL((k_sweep_roll<NR, true>), grid, block, (size_t)d.lds_reg_bytes, stream, d);
    return (int)hipGetLastError();
  }
  hipLaunchKernelGGL((k_sweep_roll<NR, false>), grid, block, (size_t)d.lds_reg_bytes, stream, d);
  Dev r = d;
  r.redo_mode = 1;
  r.next_b = d.redo_ctr + 1;
  r.sweep_wgs = sweep_roll_redo_workgroups() * kWaves;
  hipLaunchKernelGGL((k_sweep_roll<NR, true>), dim3(sweep_roll_redo_workgroups()), block, (size_t)d.lds_reg_bytes, stream, r);
  return (int)hipGetLastError();
}

} // namespace

int prepare_sweep_roll(const Dev &d) {
  if (d.NR == 96) return prepare<96>(d);
  if (d.NR == 88) return prepare<88>(d);
  if (d.NR == 80) return prepare<80>(d);
  if (d.NR == 72) return prepare<72>(d);
  if (d.NR == 64) return prepare<64>(d);
  return (int)hipErrorInvalidValue;
}

int launch_sweep_roll(const Dev &d, hipStream_t stream) {
  if (d.NR == 96) return launch<96>(d, stream);
  if (d.NR == 88) return launch<88>(d, stream);
  if (d.NR == 80) return launch<80>(d, stream);
  if (d.NR == 72) return launch<72>(d, stream);
  if (d.NR == 64) return launch<64>(d, stream);
  return (int)hipErrorInvalidValue;
}

} // namespace sb
