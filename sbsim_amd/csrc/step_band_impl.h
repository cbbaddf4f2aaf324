#pragma once
// step_band_impl.h -- (included by step_band_NN.hip: one translation unit per slot count, compiled in parallel)
// the sweep kernel for floor plans of 67..258 rows and <= 96 columns inside their
// exterior ring: W = 2..4 wavefronts per building (one workgroup), the grid in their registers, one
// row per lane -- wavefront w owns rows 64 w .. 64 w + 63, rows 64 W .. (at most two) are finished
// by the affine scan of sweep_common.h.  simulator.py:278-371; any H x W is legal in building.py:609-764.
//
// step_two.hip holds a plan of up to 130 rows in ONE wavefront (two rows per lane); beyond that -- and
// beyond 80 columns -- a building's rows are split over the SIMDs of a CU.  A step is the one-row step
// of step_roll.hip.  Layout and schedule per wavefront are those of step_reg.hip / step_roll.hip: lane
// l owns one row, column c in register slot (c + l) mod NR, every lane works on slot s mod NR at step
// s, neighbours by DPP; consecutive sweeps are overlapped in predicted BLOCKS as in step_two.hip
// (ramp-up, rolling periods of NR steps, final period; a block that ran past the step's last sweep is
// run again from the stored grid with the right count).
//
// The wavefronts of a building:
//   * Wavefront w runs >= 64 steps behind wavefront w - 1 (row 64 w's upper neighbours are row
//     64 w - 1's new values) and <= NR + 62 steps behind (row 64 w - 1's lower neighbours are row
//     64 w's values of the previous sweep).  Both bounds are kept by a progress counter per wavefront
//     in LDS, checked every kGrp steps (a middle wavefront checks both of its neighbours').
//   * The seam values travel through LDS: every step each wavefront writes ONE ds_write_b64 whose
//     per-lane base address sends lane 63's and lane 0's results to the wavefront's two seam rows and
//     every other lane's to a scratch strip; the step's immediate offset 8 * (s mod NR) does the
//     indexing.  Readers use two uniform ds_read_b64 per step whose results enter as the DPP `old`
//     operand of lane 0 (upper neighbour) and lane 63 (lower neighbour).  All wavefronts execute the
//     SAME code (their roles differ in four base addresses), so the unrolled periods are in the
//     instruction cache once.
//   * max|delta| of a sweep is the maximum over the wavefronts (and the tail rows): each publishes its
//     part per sweep; the decisions -- roll on, end the block, run it again -- are functions of the
//     published parts only, evaluated by all wavefronts alike.  The last wavefront's rows finish a sweep
//     W - 1 periods after wavefront 0's: a decision after sweep G reads wavefront k's parts up to sweep
//     G - k - 1 (its own up to G; a period of slack, so that no decision waits), predicts every
//     wavefront's last sweep from the decay of ITS part and rolls on while the latest of them is beyond
//     the next sweep.  Whether a sweep was the step's last is known for certain W periods later; a block
//     that has run past it by then is run again.  A step's first block aims at one sweep less than the
//     building's previous step took until its own parts say more.
// The iterates and the sweep count are always those of the plain schedule (tests: oracle twins,
// step_lds.hip on the same batch); the prediction only decides the speed.
#include <type_traits>

#include "step_band_cfg.h"
#include "sweep_common.h"

namespace sb {
namespace {

using namespace sweep;
using namespace band;

#ifndef SB_BAND_EXP
#define SB_BAND_EXP 0
#endif
constexpr int kWA = 11;    // class words (one step each) are read this many steps ahead (an L2 hit is ~800 cycles away)
constexpr int kZA = 8;     // zone-offset words (four slots each) read ahead in the hand-over
static_assert(kWA % 8 != 0 && (63 + kWA) % 8 != 0, "step_set: the start words leave the offset inside a chunk");

// Slots of the lane's grid row homed in VGPRs (the rest: AGPRs).  Up to 80 slots the row, the pair
// buffers and the class-word ring fit the 256 registers VALU instructions can address.
constexpr int row_vgpr_slots(int NR) { return NR <= 80 ? NR : 72; }
// Pairs of steps whose LDS reads are in flight ahead of the arithmetic (SB_BAND_PD).  Two were no faster than one
// (tools/band_period.py: 10.4 against 10.1 us per 96-step period; lgkmcnt counts to 15 and a pair has nine LDS
// instructions, so the second pair's reads wait for the first's anyway) and cost 172 B of scratch at 96 slots.
#ifndef SB_BAND_SLEEP
#define SB_BAND_SLEEP 1 // s_sleep argument of the polls (x 64 cycles)
#endif
#ifndef SB_BAND_PD
#define SB_BAND_PD 1
#endif
constexpr int kPD = SB_BAND_PD, kPR = kPD + 1; // ... and the ring of pair buffers

typedef const double __attribute__((address_space(3))) *lds_d;
typedef double __attribute__((address_space(3))) *lds_dw;
typedef volatile int __attribute__((address_space(3))) *lds_vi;
typedef volatile double __attribute__((address_space(3))) *lds_vd;

struct PairBuf { // LDS values of two consecutive steps (the first one odd)
  d2 ud0, lr0, ud1, lr1; // (bU, bD), (bL, bR)
  d2 A, rU, rD;          // A of the two slots; upper neighbours of lane 0, lower neighbours of lane 63
};

struct Acc {
  double cur; // max |delta| of the sweep the lanes are finishing
  double neg; // -(max |delta|) of the sweep the lanes have started (rolling periods)
  int sg;     // 0x80000000 in the lanes that have started the next sweep
};

struct Ctx {
  unsigned arow;   // LDS byte address of the lane's row of A
  unsigned ubase;  // LDS byte address: upper neighbours of lane 0 by (step + 63) mod NR
  unsigned dbase;  // LDS byte address: lower neighbours of lane 63 by step (column + 63)
  unsigned pub;    // LDS byte address the lane publishes to (seam row / scratch strip), by step mod NR
  const char *cmap;
  unsigned voff;
  unsigned w[kWA + 1];
};

// The neighbouring wavefronts' progress counters and how far each must be (relative to this
// wavefront's step count) before a group of steps may run.
struct Sync {
  lds_vi mine, up, dn;
  int off_up, off_dn; // + 64 / - NR - 62; a wavefront without that neighbour: far below zero
  long long *dbg;     // developer builds (-DSB_PHASE_STAMPS): spin counters
  long long *tl;      // ... and (SBSIM_DEBUG_TIMELINE=1) a time line of one building-step of workgroup 0: [512] per wavefront, (cycle << 12) | step or code
  mutable int tl_i;
};
#ifdef SB_PHASE_STAMPS
__device__ __forceinline__ void tl_mark(const Sync &sy, int code) {
  if (sy.tl && (threadIdx.x & 63) == 0 && sy.tl_i < 512) sy.tl[sy.tl_i++] = ((long long)__builtin_readcyclecounter() << 12) | (long long)(code & 0xfff);
}
#else
__device__ __forceinline__ void tl_mark(const Sync &, int) {}
#endif

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

// Class words: one 32-bit word per step = the LDS byte offset (set * 32) of the cell's coefficient
// set, read from global memory (L2 hits) kWA steps ahead; [wavefront][NR + 63 steps][64 lanes].
__device__ __forceinline__ unsigned class_word(const Ctx &x) { return *(const unsigned *)(x.cmap + x.voff); }
// The first kWA words of a rolling period (steps 63 ..) are the same every time: read once per kernel and kept
// in registers where the read-ahead cannot run on across the period's end (words_run_on; a block's first words
// are read again).
// Word S + kWA is read at step S through base + 32-bit offset + immediate: the offset register moves
// once per 8 steps.
// Values that are touched once per period (or per building) are homed in AGPRs by hand: the grid
// row, the pair buffers and the class-word ring fill the 256 VGPRs that VALU instructions can address,
// and left to the register allocator it is the grid that travels through AGPRs in every step.
__device__ __forceinline__ int to_agpr(int v) {
  int a;
  asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(a) : "v"(v));
  return a;
}
__device__ __forceinline__ int from_agpr(int a) {
  int v;
  asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v) : "a"(a));
  return v;
}
template <int NR>
constexpr bool words_run_on() { return NR % (kWA + 1) == 0 && NR % 8 == 0; }
template <bool KEEP>
struct StartWords {
  int period[KEEP ? kWA : 1]; // AGPRs
};
template <bool KEEP>
__device__ __forceinline__ void load_start_words(StartWords<KEEP> &sw, const Ctx &x, int lane) {
  if constexpr (KEEP) {
#pragma unroll
    for (int k = 0; k < kWA; ++k) sw.period[k] = to_agpr((int)*(const unsigned *)(x.cmap + (unsigned)(lane * 4 + (63 + k) * 256)));
  }
}
__device__ __forceinline__ void first_words(Ctx &x, int lane) { // words 0 .. kWA-1 of a block (once per block: from memory)
  x.voff = (unsigned)opaque(lane * 4 + (kWA / 8) * 2048);
  const unsigned o = (unsigned)opaque(lane * 4);
#pragma unroll
  for (int k = 0; k < kWA; ++k) x.w[k] = *(const unsigned *)(x.cmap + o + (unsigned)(k * 256));
}
template <int NR, bool KEEP>
__device__ __forceinline__ void period_words(Ctx &x, const StartWords<KEEP> &sw, int lane) { // a rolling period starts at step 63
  if constexpr (KEEP) {
    x.voff = (unsigned)opaque(lane * 4 + ((63 + kWA) / 8) * 2048);
#pragma unroll
    for (int k = 0; k < kWA; ++k) x.w[(63 + k) % (kWA + 1)] = (unsigned)from_agpr(sw.period[k]);
  } else {
    static_assert(KEEP || words_run_on<NR>(), "the read-ahead has run on into the next period (step_set)");
  }
}
// The slots of A that do not fit in LDS.
template <int N>
struct ARegs {
  int a[2 * N]; // AGPRs
  template <int K>
  __device__ __forceinline__ void set(double v) {
    a[2 * K] = to_agpr(__double2loint(v));
    a[2 * K + 1] = to_agpr(__double2hiint(v));
  }
  template <int K>
  __device__ __forceinline__ double get() const {
    return __hiloint2double(from_agpr(a[2 * K + 1]), from_agpr(a[2 * K]));
  }
};
// The lane's grid row: slot j = column (j - lane) mod NR.  The first NV slots live in VGPRs, the rest
// in AGPRs (every version of a slot in the same one: the old value is a tied input).  A step reads the
// slot one ahead of its own and writes its own: at most four v_accvgpr moves.
template <int NR, int NV>
struct Row {
  double v[NV];
  int a[2 * (NR - NV) + 2];
  __device__ __forceinline__ void init() {
#pragma unroll
    for (int k = 0; k < 2 * (NR - NV); ++k) asm volatile("v_accvgpr_write_b32 %0, 0" : "=a"(a[k]));
#pragma unroll
    for (int k = 0; k < NV; ++k) v[k] = 0.0;
  }
  template <int J>
  __device__ __forceinline__ double get() const {
    static_assert(J >= 0 && J < NR, "slot");
    if constexpr (J < NV) return v[J];
    else {
      int lo, hi;
      asm("v_accvgpr_read_b32 %0, %1" : "=v"(lo) : "a"(a[2 * (J - NV)]));
      asm("v_accvgpr_read_b32 %0, %1" : "=v"(hi) : "a"(a[2 * (J - NV) + 1]));
      return __hiloint2double(hi, lo);
    }
  }
  template <int J>
  __device__ __forceinline__ void set(double x) {
    static_assert(J >= 0 && J < NR, "slot");
    if constexpr (J < NV) v[J] = x;
    else {
      asm("v_accvgpr_write_b32 %0, %2" : "=a"(a[2 * (J - NV)]) : "0"(a[2 * (J - NV)]), "v"(__double2loint(x)));
      asm("v_accvgpr_write_b32 %0, %2" : "=a"(a[2 * (J - NV) + 1]) : "0"(a[2 * (J - NV) + 1]), "v"(__double2hiint(x)));
    }
  }
};
// The two slots a step has in hand: the previous slot's new value (the left-hand neighbour) and the
// current slot's old value (the slot one ahead is read by the step itself).
struct Win {
  double p, c;
};

// The class words of a rolling period repeat with period NR from step 63 on (step s + NR works on the same
// column as step s, one sweep later).  Where NR is a multiple of the ring's kWA + 1 entries the read-ahead simply
// runs on across a period's end -- word s + kWA - NR into the slot of word s + kWA -- and the next period starts
// with its first words in place; otherwise they come from the StartWords.
template <int NR, int S, bool ROLL>
__device__ __forceinline__ lds_d2 step_set(Ctx &x) { // the coefficient set of step S; reads word S + kWA
  if constexpr (S + kWA < NR + 63) {
    if constexpr ((S + kWA) % 8 == 0 && S > 0 && S != 63) { // (the start words leave the offset at its chunk)
      x.voff += 2048u;
      asm volatile("" : "+v"(x.voff)); // a running offset: nothing for the compiler to hoist
    }
    x.w[(S + kWA) % (kWA + 1)] = *(const unsigned *)(x.cmap + x.voff + (unsigned)(((S + kWA) % 8) * 256));
  } else if constexpr (ROLL && words_run_on<NR>()) {
    constexpr int T = S + kWA - NR; // >= 63: the next period's step
    if constexpr (T == 63) {
      x.voff -= (unsigned)(((NR + 62) / 8 - 63 / 8) * 2048);
      asm volatile("" : "+v"(x.voff));
    } else if constexpr (T % 8 == 0) {
      x.voff += 2048u;
      asm volatile("" : "+v"(x.voff));
    }
    x.w[(S + kWA) % (kWA + 1)] = *(const unsigned *)(x.cmap + x.voff + (unsigned)((T % 8) * 256));
  }
  return (lds_d2)x.w[S % (kWA + 1)];
}

// LDS reads of the pair (S, S + 1), S odd.
template <int NR, int S, bool ROLL, int NAR>
__device__ __forceinline__ void load_pair(PairBuf &p, Ctx &x, const ARegs<NAR> &Areg) {
  static_assert(S % 2 == 1, "pairs start at odd steps");
  const lds_d2 s0 = step_set<NR, S, ROLL>(x), s1 = step_set<NR, S + 1, ROLL>(x);
#if SB_BAND_EXP >= 2
  p.rU = d2{0.0, 0.0};
  p.rD = d2{0.0, 0.0};
#else
  p.rU = *(lds_d2)(x.ubase + 8u * ((S + 63) % NR));
  p.rD = *(lds_d2)(x.dbase + 8u * S);
#endif
  p.ud0 = s0[0];
  p.lr0 = s0[1];
  p.ud1 = s1[0];
  p.lr1 = s1[1];
  __builtin_amdgcn_sched_barrier(0);
  constexpr int q = (S + 1) % NR, NL = lds_slots(NR); // position of slot S mod NR in the lane's A row: even
  static_assert(q % 2 == 0, "A pairs are 16-byte aligned");
  if constexpr (q < NL) p.A = *(lds_d2)(x.arow + 8u * q);
  else p.A = d2{Areg.template get<q - NL>(), Areg.template get<q + 1 - NL>()};
}

// One Gauss-Seidel update of every lane's current cell at step S of a block:
//   S < 63             ramp-up of the block's first sweep: lanes > S have not started
//   63 <= S < NR       all 64 lanes are in the same sweep
//   NR <= S < NR + 63  ROLL: lanes <= S - NR are in the next sweep; else they have finished (masked)
// Association order of the four products as in step_reg.hip / step_lds.hip / step_roll.hip.
template <int NR, int NV, int S, bool ROLL>
__device__ __forceinline__ void step(Row<NR, NV> &e, Win &w, d2 ud, d2 lr, double A, double rU, double rD, Acc &acc, const Ctx &x) {
  constexpr int r = S % NR, rp = (S + 1) % NR;
  const double nx = e.template get<rp>(); // the old value one column ahead
  const double Dn = wave_shift1<0x130, true>(nx, rD);
  double t;
  asm("v_fma_f64 %0, %1, %2, %3" : "=&v"(t) : "v"(ud.y), "v"(Dn), "v"(A));
  t = fma(lr.y, nx, t);
  const double U = wave_shift1<0x138, true>(w.p, rU);
  t = fma(lr.x, w.p, t);
  const double nv = fma(ud.x, U, t);
  double sel = nv;
  if constexpr (ROLL && S >= NR) {
    constexpr int J = S - NR;
    const double d = nv - w.c;
    // +|d| in the lanes still in sweep k, -|d| in the lanes already in sweep k+1
    const double sd = __hiloint2double((__double2hiint(d) & 0x7fffffff) | acc.sg, __double2loint(d));
    acc.cur = fmax(acc.cur, sd);
    acc.neg = fmin(acc.neg, sd);
    asm volatile("" : "+v"(acc.neg));
    if constexpr (J + 1 < 63) // lane J + 1 starts its next sweep at the next step (lane 0 keeps its bit)
      acc.sg = __builtin_amdgcn_update_dpp(acc.sg, acc.sg, 0x138, 0xf, 0xf, false);
  } else {
    if constexpr (S < 63) sel = lanes_upto<S>() ? nv : w.c;
    else if constexpr (S >= NR) sel = lanes_upto<S - NR>() ? w.c : nv;
    acc.cur = fmax(acc.cur, fabs(sel - w.c));
  }
  asm volatile("" : "+v"(acc.cur)); // here, not after the sweep (the maxima would keep every delta of the sweep alive)
  e.template set<r>(sel);
#if SB_BAND_EXP != 1 // (timing experiments: 1 = no publish, 2 = no seam reads, 3 = neither)
#if SB_BAND_EXP != 3
  *(lds_dw)(x.pub + 8u * r) = sel; // lane 63 / lane 0: the seam rows; every other lane: its scratch strip
#endif
#endif
  w.p = sel;
  w.c = nx;
}

// A wait that cannot hang the GPU: the neighbours are a few hundred steps away at most.
__device__ __forceinline__ void wait_for(lds_vi ctr, int need, long long *dbg = nullptr) {
  int spins = 0;
  while (__builtin_amdgcn_readfirstlane(*ctr) < need) {
    __builtin_amdgcn_s_sleep(SB_BAND_SLEEP);
    if (++spins > (1 << 24)) __builtin_trap();
  }
#ifdef SB_PHASE_STAMPS // developer aid: how long do the wavefronts wait for each other?
  if (dbg && spins && (threadIdx.x & 63) == 0) atomicAdd((unsigned long long *)dbg, (unsigned long long)spins);
#endif
}
// The neighbours' progress (steps completed in this block), before this wavefront runs the steps up to
// `upto` (it reads one step ahead): the wavefront above must be 64 steps further (row 64 w - 1's new
// values must exist), the one below at most NR + 62 steps back (row 64 w + 64's values of the previous
// sweep must exist, and it must have read the upper neighbours this wavefront is about to overwrite).
__device__ __forceinline__ void sync_steps(const Sync &sy, int done, int upto) {
  tl_mark(sy, done & 0x7ff);
  *sy.mine = done;
  if (upto + sy.off_up > 0) wait_for(sy.up, upto + sy.off_up, sy.dbg ? sy.dbg + 6 : nullptr);
  if (upto + sy.off_dn > 0) wait_for(sy.dn, upto + sy.off_dn, sy.dbg ? sy.dbg + 7 : nullptr);
  asm volatile("" ::: "memory");
}

// Pairs S, S + 2, .. < S1 (S odd) of a block whose period started at local step tb (the block's step
// count is tb + S); the LDS reads of the next pair are issued before the arithmetic of a pair.
template <int NR, int NV, int S, int S1, bool ROLL, int NAR>
__device__ __forceinline__ void run_pairs(Row<NR, NV> &e, Win &w, const ARegs<NAR> &Areg, PairBuf (&pb)[kPR], Ctx &x, Acc &acc,
                                          const Sync &sy, int tb, int last_step) {
  if constexpr (S < S1) {
    if constexpr (!ROLL && S >= NR && (S - NR) % 4 == 1)
      if (S > last_step) return; // uniform: only lanes without rows are left
    if constexpr (S % kGrp == (S < 63 ? 1 : 63 % kGrp)) { // the group's last pair reads ahead for the pair after it
      constexpr int n = (S1 - S < kGrp ? S1 - S : kGrp) + 2 * kPD; // the group's steps + the pairs read ahead
#if SB_BAND_EXP != 5 // (timing experiment 5: no progress checks -- the wavefronts run free, the seam values are garbage)
      sync_steps(sy, tb + S, tb + S + n);
#endif
    }
    PairBuf &cur = pb[((S - 1) / 2) % kPR], &nxt = pb[((S - 1) / 2 + kPD) % kPR];
    if constexpr (S + 2 * kPD < NR + 63) load_pair<NR, S + 2 * kPD, ROLL>(nxt, x, Areg);
    __builtin_amdgcn_sched_barrier(0);
    step<NR, NV, S, ROLL>(e, w, cur.ud0, cur.lr0, cur.A.x, cur.rU.x, cur.rD.x, acc, x);
    __builtin_amdgcn_sched_barrier(0);
    step<NR, NV, S + 1, ROLL>(e, w, cur.ud1, cur.lr1, cur.A.y, cur.rU.y, cur.rD.y, acc, x);
    __builtin_amdgcn_sched_barrier(0);
    run_pairs<NR, NV, S + 2, S1, ROLL>(e, w, Areg, pb, x, acc, sy, tb, last_step);
  }
}

// Sweeps to go until max|delta| reaches the threshold, from the decay over the last two sweeps
// (d1 -> d0), `haste` x as fast in the exponent; a large number when it does not decay.
__device__ __forceinline__ float sweeps_to_go(float d1, float d0, float thr, float haste) {
  if (!(d0 > thr)) return 0.0f;
  if (!(d0 < d1)) return 1e9f;
  return __log2f(thr / d0) / (haste * __log2f(d0 / d1));
}

// A = ap*Tprev + g for the lane's cells (e = Tprev before the first sweep).  aw: class offsets into
// the (ap, g) table, four slots per word.
template <int NR, int NV, int NAR>
__device__ __forceinline__ void a_pass(const Row<NR, NV> &e, ARegs<NAR> &Areg, double *Aw, const char *tapg,
                                       const unsigned long long *amap) {
  constexpr int NL = lds_slots(NR), NWD = NR / 4;
  static_assert(NR % 4 == 0, "a_pass: four slots per word");
  constexpr int kAA = 6; // words read ahead
  unsigned long long aw[kAA + 1];
#pragma unroll
  for (int k = 0; k < kAA; ++k) aw[k] = amap[k * 64];
  static_for<0, NWD>([&](auto gc) {
    constexpr int W0 = decltype(gc)::value, j0 = 4 * W0;
    if constexpr (W0 + kAA < NWD) aw[(W0 + kAA) % (kAA + 1)] = amap[(W0 + kAA) * 64];
    const unsigned long long w0 = aw[W0 % (kAA + 1)];
    d2 pg[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) pg[k] = *(const d2 *)(tapg + (unsigned)((w0 >> (16 * k)) & 0xffffull));
    __builtin_amdgcn_sched_barrier(0);
    static_for<0, 4>([&](auto kc) {
      constexpr int k = decltype(kc)::value, j = j0 + k;
      const double av = fma(pg[k].x, e.template get<j>(), pg[k].y);
      constexpr int q = (j + 1) % NR; // slot j's position in the lane's row
      if constexpr (q < NL) Aw[q] = av;
      else Areg.template set<q - NL>(av);
    });
    __builtin_amdgcn_sched_barrier(0);
  });
}

// The lane's row <-> the building's state in HBM, [NR / 2][64 W rows][2]: two slots per access.
// base: the building's state (uniform); off: 16 * the lane's row; rsb: bytes between two slot pairs (32 * rows).
// The offset is a running one the compiler cannot see through (NR / 2 loop-invariant addresses would live
// in scratch).
template <int NR, int NV>
__device__ __forceinline__ void load_row(Row<NR, NV> &e, const double *base, unsigned off, unsigned rsb) {
  static_for<0, NR / 2>([&](auto jc) {
    constexpr int j = 2 * decltype(jc)::value;
    asm volatile("" : "+v"(off));
    const d2 v = *(const d2 *)((const char *)base + off);
    off += rsb;
    e.template set<j>(v.x);
    e.template set<j + 1>(v.y);
  });
}
template <int NR, int NV>
__device__ __forceinline__ void store_row(const Row<NR, NV> &e, double *base, unsigned off, unsigned rsb) {
  static_for<0, NR / 2>([&](auto jc) {
    constexpr int j = 2 * decltype(jc)::value;
    asm volatile("" : "+v"(off));
    *(d2 *)((char *)base + off) = d2{e.template get<j>(), e.template get<j + 1>()};
    off += rsb;
  });
}

// The end of a building's step, two slots at a time: store them, add them to their zone sums (LDS: the
// lane's own slot of the zone, plan_band), load the same slots of the next building.  A VGPR slot is its
// load's destination; an AGPR slot's value arrives in a VGPR first, so those loads run kHA pairs ahead.
constexpr int kHA = 3;
template <int NR, int NV, int J>
__device__ __forceinline__ void hand_over(Row<NR, NV> &e, unsigned long long (&zw)[kZA + 1], const unsigned long long *zmap,
                                          double *tb, const double *nb, unsigned rsb, unsigned &off, d2 (&pre)[kHA + 1], double *zs) {
  static_assert(NV % 2 == 0 && (NR == NV || NR - NV >= 2 * kHA), "slot pairs; the read-ahead fits the AGPR slots");
  if constexpr (J < NR) {
    if constexpr (J % 4 == 0 && J / 4 + kZA < NR / 4) zw[(J / 4 + kZA) % (kZA + 1)] = zmap[(J / 4 + kZA) * 64];
    const unsigned long long w = zw[(J / 4) % (kZA + 1)];
    const unsigned i0 = (unsigned)((w >> (16 * (J & 3))) & 0xffffull), i1 = (unsigned)((w >> (16 * ((J + 1) & 3))) & 0xffffull);
    const double v0 = e.template get<J>(), v1 = e.template get<J + 1>();
    asm volatile("" : "+v"(off));
    if constexpr (J == NV) {
#pragma unroll
      for (int k = 0; k < kHA; ++k) pre[k] = *(const d2 *)((const char *)nb + (off + (unsigned)k * rsb));
    }
    if constexpr (J >= NV && J + 2 * kHA < NR) pre[((J - NV) / 2 + kHA) % (kHA + 1)] = *(const d2 *)((const char *)nb + (off + (unsigned)kHA * rsb));
    *(d2 *)((char *)tb + off) = d2{v0, v1};
    __hip_atomic_fetch_add(zs + i0, v0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_fetch_add(zs + i1, v1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    d2 nv;
    if constexpr (J < NV) nv = *(const d2 *)((const char *)nb + off);
    else nv = pre[((J - NV) / 2) % (kHA + 1)];
    off += rsb;
    e.template set<J>(nv.x);
    e.template set<J + 1>(nv.y);
    if constexpr ((J & 7) == 6) __builtin_amdgcn_sched_barrier(0);
    hand_over<NR, NV, J + 2>(e, zw, zmap, tb, nb, rsb, off, pre, zs);
  }
}

extern __shared__ __attribute__((aligned(16))) double lds[];

// LDS (doubles): [tabc 4 kSets][tapg 2 ts] | r_seam: [zero][up W][dn W][tE0] (seam_region each)
// [scratch W x (64 + NR + 8)] | r_xchg: progress [W][64] ints, max|delta| records [W][kHist], misc | r_A: A [W][64][AS]
// (after the sweeps: zone sums)
template <int NR>
__global__ void __launch_bounds__(64 * kWMax) __attribute__((amdgpu_waves_per_eu(1, 1))) k_sweep_band(Dev a) {
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); // wavefront-uniform: SGPRs
  const int W = (int)(blockDim.x >> 6);
  constexpr int kNL = lds_slots(NR), kAS = kNL, kNAR = NR - kNL > 0 ? NR - kNL : 1, kRG = seam_region(NR), NV = row_vgpr_slots(NR);
  static_assert(NR % 4 == 0 && NR >= 68, "slots");

  double *tabc = lds;                    // [kSets][4]: bU bD bL bR per coefficient set
  double *tapg = lds + 4 * kSets;        // [ts][2]: (ap, g) per class; g of this building
  double *seam = lds + a.r_seam;
  double *S_zero = seam + 64, *S_up = seam + kRG + 64, *S_dn = seam + (1 + W) * kRG + 64; // S_up[w], S_dn[w]: + w kRG
  double *S_tl = S_up + (W - 1) * kRG;   // the last wavefront row's new values: the tail scan's upper neighbours
  double *tE0 = seam + (1 + 2 * W) * kRG + 64; // the first tail row by column (zeros without tail rows)
  double *scratch = seam + (2 + 2 * W) * kRG;  // [W][64 + NR + 8]
  int *sync = (int *)(lds + a.r_xchg);   // [W][64] progress counters
  double *mrec = lds + a.r_xchg + 32 * W; // [W][kHist] records {max|delta| part, sweep number}: 16 bytes each
  int *misc = (int *)(mrec + 2 * W * kHist); // [0]: the next building
  double *A = lds + a.r_A;               // [W][64][kAS]; after the sweeps: zone sums (a.zs_off)
  for (int i = threadIdx.x; i < a.lds_reg_bytes / 8; i += blockDim.x) lds[i] = 0.0; // every byte starts finite
  __syncthreads();
  for (int i = threadIdx.x; i < 4 * kSets; i += blockDim.x) tabc[i] = i < 4 * a.ncset ? a.csetab[i] : 0.0;
  for (int c = threadIdx.x; c < a.ts; c += blockDim.x) tapg[2 * c] = c <= a.ncls ? a.ctab[c * 8 + 4] : 0.0; // row `ncls`: the pad class
  __syncthreads();

  const sb_params &p = a.p;
  if ((unsigned)(size_t)(__attribute__((address_space(3))) double *)tabc != 0u) __builtin_trap(); // class words hold LDS addresses
  auto lds_addr = [](const void *q) { return (unsigned)(size_t)(const __attribute__((address_space(3))) char *)q; };
  Ctx x;
  x.arow = lds_addr(A + ((size_t)wv * 64 + lane) * kAS);
  x.cmap = (const char *)a.cmapS + (size_t)wv * (NR + 63) * 256;
  x.voff = 0;
  const bool tails = a.T > 0, first_w = wv == 0, last_w = wv == W - 1;
  // upper neighbours of lane 0: the wavefront above's last row (new values); wavefront 0: nothing counts (zeros).
  // Lower neighbours of lane 63: the wavefront below's first row; the last wavefront: the first tail row (or nothing)
  x.ubase = lds_addr(first_w ? S_zero : S_up + (wv - 1) * kRG);
  x.dbase = lds_addr(!last_w ? S_dn + (wv + 1) * kRG : (tails ? tE0 : S_zero)) - 8u * 63u;
  {
    double *strip = scratch + (size_t)wv * (64 + NR + 8) + lane;
    double *target = strip;
    if (lane == 63) target = S_up + wv * kRG;
    if (lane == 0) target = S_dn + wv * kRG; // (wavefront 0's has no reader)
    x.pub = lds_addr(target);
  }
  Sync sy;
  sy.mine = (lds_vi)(lds_addr(sync + wv * 64 + lane));
  sy.up = (lds_vi)(lds_addr(sync + (first_w ? 0 : wv - 1) * 64 + lane));
  sy.dn = (lds_vi)(lds_addr(sync + (last_w ? wv : wv + 1) * 64 + lane));
  sy.off_up = first_w ? -(1 << 29) : 64;
  sy.off_dn = last_w ? -(1 << 29) : -NR - 62;
  sy.dbg = a.dbg;
  sy.tl = nullptr;
  sy.tl_i = 0;
  const int rows_mine = a.lw[wv];        // lanes that own rows
  const int last_step = NR + rows_mine - 2;
  // the lane's tail cells (the last wavefront; static per floor plan)
  const bool tactive = last_w && tails && tail_col<NR>(lane, 0) >= 0;
  const int tc0 = tactive ? tail_col<NR>(lane, 0) : 0;
  int tset[kTailMax];
#pragma unroll
  for (int t = 0; t < kTailMax; ++t) {
    tset[t] = (a.ncset - 1) * 32 * 0x10001; // the pad set (the table's last)
    if (t < a.T && tactive) tset[t] = ((int)a.tcset[t * NR + tc0] << 2) | ((int)a.tcset[t * NR + tc0 + 1] << 18);
  }
  StartWords<!words_run_on<NR>()> sw;
  load_start_words(sw, x, lane);
  // (a.amapS / a.zmapS + the wavefront's part + lane: formed where they are used, from an opaque lane number -- as
  // kernel-lifetime 64-bit values they live in scratch)
  const int map_w = wv * (NR / 4) * 64;
  const int R = wv * 64 + lane;  // the lane's row of the state [NR / 2][64 W][2]
  const unsigned rsb = 16u * (unsigned)a.RS; // bytes between two slot pairs of the state (a.RS = 64 W rows)

#ifdef SB_PHASE_STAMPS
#define SB_STAMP(i) do { if (a.dbg && blockIdx.x == 0 && iter == 3 && threadIdx.x == 0) a.dbg[i] = (long long)__builtin_readcyclecounter(); } while (0)
#define SB_COUNT(i) do { if (a.dbg && threadIdx.x == 0) atomicAdd((unsigned long long *)a.dbg + (i), 1ull); } while (0)
#else
#define SB_STAMP(i) do { } while (0)
#define SB_COUNT(i) do { } while (0)
#endif

  Row<NR, NV> e;
  e.init();
  Win w;
  double nx_tnow = 0.0, nx_lo = 0.0, nx_hi = 0.0;
  double tv[kTailMax][2] = {{0.0, 0.0}, {0.0, 0.0}};
#define SB_LOAD_AUX(bb)                                                                          \
  do {                                                                                           \
    nx_tnow = a.bld[(bb)].t_now;                                                                 \
    nx_lo = a.scal[(size_t)(bb) * kNScal + 16];                                                  \
    nx_hi = a.scal[(size_t)(bb) * kNScal + 17];                                                  \
    for (int c = threadIdx.x; c < a.ts; c += blockDim.x) tapg[2 * c + 1] = a.gtabg[(size_t)(bb) * a.ts + c]; \
    const double *tt_ = a.temp + (size_t)(bb) * a.state_doubles + (size_t)NR * a.RS;             \
    _Pragma("unroll") for (int t = 0; t < kTailMax; ++t)                                         \
      _Pragma("unroll") for (int k = 0; k < 2; ++k)                                              \
        if (t < a.T && tactive) tv[t][k] = tt_[t * NR + opaque(tc0) + k];                        \
  } while (0)
  if ((int)blockIdx.x < a.B) {
    load_row(e, a.temp + (size_t)blockIdx.x * a.state_doubles, 16u * (unsigned)R, rsb);
    SB_LOAD_AUX(blockIdx.x);
  }
  int iter = 0;
  for (int b = blockIdx.x, bn = 0; b < a.B; b = bn, ++iter) {
    if (threadIdx.x == 0) misc[0] = a.sweep_wgs + atomicAdd(a.next_b, 1);
    SB_STAMP(0);
#ifdef SB_PHASE_STAMPS
    sy.tl = (a.dbg && a.dbg_timeline && blockIdx.x == 0 && iter == 2) ? a.dbg + 16 + 512 * wv : nullptr;
#endif
    first_words(x, lane);
    double *Ttail = a.temp + (size_t)b * a.state_doubles + (size_t)NR * a.RS; // [T][NR]
    double *tp = a.temp + (size_t)b * a.state_doubles; // uniform; the lane's row: + 16 R bytes
    const double t_now = nx_tnow;
    // exterior-space cells outside the trim box all become t_now in the first sweep
    // (simulator.py:256-258); their largest |delta| follows from their extreme values
    const double ring_d = a.n_ring > 0 ? fmax(fabs(t_now - nx_lo), fabs(t_now - nx_hi)) : 0.0;
    if (tactive) *(d2 *)(tE0 + opaque(tc0)) = d2{tv[0][0], tv[0][1]};
    __syncthreads(); // the (ap, g) table and the tail row are in LDS; the previous building's zone sums are read
    bn = __builtin_amdgcn_readfirstlane(*(volatile int *)misc);
    SB_STAMP(1);
    double At[kTailMax][2] = {{0.0, 0.0}, {0.0, 0.0}}; // A of the lane's tail cells
#pragma unroll
    for (int t = 0; t < kTailMax; ++t)
#pragma unroll
      for (int k = 0; k < 2; ++k)
        if (t < a.T && tactive) {
          const d2 pg = *(const d2 *)((const char *)tapg + 16 * (int)a.tcls[t * NR + opaque(tc0) + k]);
          At[t][k] = fma(pg.x, tv[t][k], pg.y);
        }
    ARegs<kNAR> Areg;
    a_pass<NR, NV>(e, Areg, A + ((size_t)wv * 64 + lane) * kAS, (const char *)tapg, a.amapS + map_w + opaque(lane));
    __builtin_amdgcn_sched_barrier(0);
    SB_STAMP(2);

    int n_sweeps = 0, converged = 0;
    {
      PairBuf pb[kPR];
      Acc acc;
      const float thr = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, (float)p.conv_threshold))); // uniform: an SGPR
      const int prev_sweeps = __builtin_amdgcn_readfirstlane(a.nsw[b] & 0xffff); // of this building's previous step
      // max |delta| of sweep G of this step = the maximum of the wavefronts' parts (G >= 1).  A part is
      // published as a 16-byte record {value, G}: one ds_read_b128 per part tells whether it is there.
      auto publish_part = [&](int G, double part) {
        const double m = wave_max(part);
        if (lane == 0) {
          const unsigned rec = lds_addr(mrec + 2 * (wv * kHist + (G % kHist)));
          *(lds_vd)rec = m;
          *(lds_vi)(rec + 8u) = G;
        }
      };
      typedef const volatile d2 __attribute__((address_space(3))) *lds_vd2;
      // wavefront k's part of sweep G (waits for it)
      auto wait_part = [&](int k, int G) -> double {
        const unsigned rec = lds_addr(mrec + 2 * (k * kHist + (G % kHist)));
        int spins = 0;
        double v;
        for (;;) {
          const d2 pk = *(lds_vd2)rec;
          if (__builtin_amdgcn_readfirstlane(__double2loint(pk.y)) == G) {
            v = pk.x;
            break;
          }
          __builtin_amdgcn_s_sleep(SB_BAND_SLEEP);
          if (++spins > (1 << 24)) __builtin_trap();
        }
#ifdef SB_PHASE_STAMPS
        if (a.dbg && spins && lane == 0) atomicAdd((unsigned long long *)a.dbg + 8, (unsigned long long)spins);
#endif
        asm volatile("" ::: "memory");
        return v;
      };
      auto sweep_md = [&](int G) -> double { // max |delta| of sweep G: every wavefront's part
        double m = 0.0;
        for (int k = 0; k < W; ++k) m = fmax(m, wait_part(k, G));
        return m;
      };
      // A decision's reads, all in flight at once (under load an LDS round trip is ~500 cycles; one after the other
      // a dozen of them stalled every wavefront for a third of a period -- tools/band_timeline.py).  part k of
      // sweep g, g <= 0: none.  A record that is not there yet (rare: the lags below leave a period of slack) is waited for.
      struct Parts { double v[3 * kWMax]; };
      auto read_parts = [&](const int (&gs)[3 * kWMax], Parts &out) {
        d2 rec[3 * kWMax];
#pragma unroll
        for (int i = 0; i < 3 * kWMax; ++i) {
          const int k = i / 3;
          rec[i] = d2{0.0, 0.0};
          if (k < W && gs[i] > 0) rec[i] = *(lds_vd2)lds_addr(mrec + 2 * (k * kHist + (gs[i] % kHist)));
        }
#pragma unroll
        for (int i = 0; i < 3 * kWMax; ++i) {
          const int k = i / 3;
          out.v[i] = 0.0;
          if (k < W && gs[i] > 0)
            out.v[i] = __builtin_amdgcn_readfirstlane(__double2loint(rec[i].y)) == gs[i] ? rec[i].x : wait_part(k, gs[i]);
        }
      };
      // The step's last sweep as the published parts predict it: sweep n is the last one when EVERY wavefront's
      // part is below the threshold, so n = max over the wavefronts of where each one's decay gets there.  When
      // wavefront v stands at its decision point after sweep G, wavefront k's parts are there without waiting
      // up to sweep G (k = 0) / G - k - 1 (k > 0: one period of slack on top of the k periods its rows lag; at
      // G - k the chain of hand-overs would stall wavefront 0 in every period -- measured, tools/band_timeline.py).
      // Every wavefront evaluates the same parts, so all decide alike.  have: wavefronts with two parts;
      // md_c: max |delta| of sweep G - W, complete in every wavefront (0 when there is no such sweep in this block).
      auto predicted_last = [&](int G, int n0_, int &have, double &md_c) -> float {
        int gs[3 * kWMax];
#pragma unroll
        for (int k = 0; k < kWMax; ++k) {
          const int g = k == 0 ? G : G - k - 1;
          gs[3 * k] = g >= 2 ? g : 0;
          gs[3 * k + 1] = g >= 2 ? g - 1 : 0;
          gs[3 * k + 2] = G - W > n0_ ? G - W : 0;
        }
        Parts pt;
        read_parts(gs, pt);
        float np = 0.0f;
        have = 0;
        md_c = G - W > n0_ ? 0.0 : 1e300;
#pragma unroll
        for (int k = 0; k < kWMax; ++k)
          if (k < W) {
            if (gs[3 * k] > 0) {
              np = fmaxf(np, (float)gs[3 * k] + sweeps_to_go((float)pt.v[3 * k + 1], (float)pt.v[3 * k], thr, a.pred_haste));
              ++have;
            }
            if (gs[3 * k + 2] > 0) md_c = fmax(md_c, pt.v[3 * k + 2]);
          }
        return np;
      };
      // the end of this wavefront's sweep G: the tail rows (the last wavefront), its part of max |delta|
      auto sweep_end = [&](int G) {
        double dm = acc.cur;
        if (tails && last_w) { // the last wavefront row's new values by column c sit at S_tl[(c + 63) mod NR]
          const int c0 = opaque(tc0); // (addresses formed here: as kernel-lifetime values they live in scratch)
          const double U0 = *(lds_d)(lds_addr(S_tl) + 8u * (unsigned)((c0 + 63) % NR)), U1 = *(lds_d)(lds_addr(S_tl) + 8u * (unsigned)((c0 + 64) % NR));
          dm = fmax(dm, tail_pass<NR>(a.T, tactive, tE0 + c0, U0, U1, tv, tset, At));
        }
        if (G == 1 && first_w) dm = fmax(dm, ring_d);
        publish_part(G, dm);
      };
      float d1 = 0.0f, d0 = 0.0f; // max |delta| of the step's last two sweeps as of the last block's end
#pragma nounroll
      for (;;) { // blocks.  simulator.py:348-368
        const int n0 = n_sweeps;
        // does this block roll at all?  By the decay so far, or -- a step's first block -- by the previous step's
        // count: the first block aims at ending with as many sweeps as last time until its own parts say more (a
        // hint; a step that converges sooner is found out and run again, one that needs more goes on in a second
        // block, which starts from complete knowledge of the decay)
        const int first_m = prev_sweeps - (a.pred_first - 2); // (a.pred_first = 2: as many as last time; developer knob SBSIM_DEBUG_PRED_FIRST)
        int roll0 = n0 >= 2 ? (int)(sweeps_to_go(d1, d0, thr, a.pred_haste) > a.pred_slack)
                            : (int)(n0 == 0 && first_m >= 2 && a.pred_first > 1);
        roll0 = __builtin_amdgcn_readfirstlane(roll0) && n0 + 2 <= p.iter_limit;
        SB_COUNT(roll0 ? 13 : 14); // developer aid: blocks / single sweeps
        if (roll0 && n0 > 0) { // the grid as of n0 sweeps, should the block overrun (before any sweep: Tprev is still there)
          store_row(e, tp, 16u * (unsigned)opaque(R), rsb);
#pragma unroll
          for (int t = 0; t < kTailMax; ++t)
            if (t < a.T && tactive) *(d2 *)(Ttail + t * NR + opaque(tc0)) = d2{tv[t][0], tv[t][1]};
        }
        int m = 0; // > 0: the block is being run again and ends with its m-th sweep
        int q = 0; // rolling periods completed
#pragma nounroll
        for (;;) { // at most twice
          // block start: every wavefront is here; progress 0, no published parts of this block yet;
          // the first rows' current values for the first sweep of the wavefronts above them
          __syncthreads();
          *sy.mine = 0;
          if (!first_w && lane == 0) {
            double *dn_mine = S_dn + wv * kRG;
            static_for<0, NR>([&](auto cc) { dn_mine[decltype(cc)::value] = e.template get<decltype(cc)::value>(); });
          }
          if (n0 == 0 && m == 0 && (int)threadIdx.x < W * kHist) *(lds_vi)(lds_addr(mrec + 2 * threadIdx.x) + 8u) = 0; // no part of this step is published (a step's parts stay: later blocks predict from them)
          __syncthreads();
          __builtin_amdgcn_sched_barrier(0);
          acc.cur = 0.0;
          acc.neg = 0.0;
          acc.sg = lane == 0 ? (int)0x80000000 : 0;
          n_sweeps = n0;
          q = 0;
          w.p = e.template get<NR - 1>();
          w.c = e.template get<0>();
          tl_mark(sy, 0xf01); // block start
          sync_steps(sy, 0, 1 + 2 * kPD); // the wavefront above's first new values must exist
          { // step 0 (lane 0, column 0) on its own: pairs start at odd steps
            const lds_d2 st = step_set<NR, 0, false>(x);
            const d2 ud = st[0], lr = st[1];
            const double A0 = *(lds_d)(x.arow + 8u), rU0 = *(lds_d)(x.ubase + 8u * 63u), rD0 = *(lds_d)(x.dbase);
            static_for<0, kPD>([&](auto kc) { load_pair<NR, 1 + 2 * decltype(kc)::value, false>(pb[decltype(kc)::value % kPR], x, Areg); });
            step<NR, NV, 0, false>(e, w, ud, lr, A0, rU0, rD0, acc, x);
          }
          __builtin_amdgcn_sched_barrier(0);
          run_pairs<NR, NV, 1, 63, false>(e, w, Areg, pb, x, acc, sy, 0, last_step); // ramp-up; reads ahead for the pair (63, 64)
          bool overrun = false;
#pragma nounroll
          for (;;) {
            // decision point: q rolling periods done = sweeps n0 + 1 .. n0 + q complete in THIS wavefront.
            // May period q + 1 roll, i.e. is sweep n0 + q + 1 certainly not the step's last?
            const int G = n0 + q; // the last sweep whose part this wavefront has published
            int go;
            if (m > 0) go = q + 1 < m;
            else if (!roll0 || G + 2 > p.iter_limit) go = 0; // the final period must fit under the limit
            else {
              // sweep G - W is complete in every wavefront without waiting: did it converge (and this block run past it)?
              int have;
              double md_c;
              const float n_pred = predicted_last(G, n0, have, md_c);
              if (md_c <= p.conv_threshold) { overrun = true; m = G - W - n0; break; }
              const float need = (float)G + a.pred_slack; // sweep G + 1 is not the last one: the next period may roll
              // (an overrun costs the block twice, a block that ends too early one more ramp: when in doubt, end it)
              if (have == W) go = n_pred > need;
              else if (have > 0 && n_pred > need + 1.0f) go = 1; // the wavefronts that have news are far from done
              else if (have > 0 && !(n_pred > need)) go = 0;     // ... are about to be done: the others will not be far behind
              else if (n0 >= 2) go = sweeps_to_go(d1, d0, thr, a.pred_haste) > a.pred_slack + (float)q; // the last block's decay
              else go = q + 2 <= first_m; // a first block without news of its own rolls up to its target
            }
            tl_mark(sy, 0xf02); // decision made
            if (!__builtin_amdgcn_readfirstlane(go)) break;
            asm volatile("" : "+v"(x.arow), "+v"(x.ubase), "+v"(x.dbase), "+v"(x.pub)); // not loop invariants: nothing to hoist (and spill)
            __builtin_amdgcn_sched_barrier(0);
#ifdef SB_PHASE_STAMPS // developer aid: wavefront 0's rolling periods, summed over all buildings ([10] cycles, [11] periods)
            const long long t_p0 = (long long)__builtin_readcyclecounter();
#endif
            run_pairs<NR, NV, 63, NR + 63, true>(e, w, Areg, pb, x, acc, sy, q * NR, last_step);
            __builtin_amdgcn_sched_barrier(0);
            *sy.mine = q * NR + NR + 63; // every step of the period is done
#ifdef SB_PHASE_STAMPS
            if (a.dbg && threadIdx.x == 0) {
              atomicAdd((unsigned long long *)a.dbg + 10, (unsigned long long)((long long)__builtin_readcyclecounter() - t_p0));
              atomicAdd((unsigned long long *)a.dbg + 11, 1ull);
            }
#endif
            ++q;
            ++n_sweeps;
            period_words<NR>(x, sw, lane);
            tl_mark(sy, 0xf03); // period's steps done
            sweep_end(n0 + q);
            tl_mark(sy, 0xf04); // part published
            acc.cur = -acc.neg;
            acc.neg = 0.0;
            acc.sg = lane == 0 ? (int)0x80000000 : 0;
            // (after the tail scan: lane 63's lower neighbours are new)
            static_for<0, kPD>([&](auto kc) { load_pair<NR, 63 + 2 * decltype(kc)::value, false>(pb[(31 + decltype(kc)::value) % kPR], x, Areg); });
          }
          if (!overrun) {
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("" : "+v"(x.arow), "+v"(x.ubase), "+v"(x.dbase), "+v"(x.pub));
            run_pairs<NR, NV, 63, NR + 63, false>(e, w, Areg, pb, x, acc, sy, q * NR, last_step); // the block's last sweep
            __builtin_amdgcn_sched_barrier(0);
            *sy.mine = 1 << 30; // the other wavefronts need nothing more from this one
            ++n_sweeps;
            first_words(x, lane);
            tl_mark(sy, 0xf05); // final period's steps done
            sweep_end(n_sweeps);
            // the block's last sweeps, complete: did one before the last converge already?  (A decision
            // checks the sweep W before its own: the last W, here.)
            const int Gl = n_sweeps;
            if (m == 0)
              for (int j = Gl - (W + 1) > n0 ? Gl - (W + 1) : n0 + 1; j < Gl; ++j)
                if (sweep_md(j) <= p.conv_threshold) { overrun = true; m = j - n0; break; }
            if (!overrun) {
              const double mdl = sweep_md(Gl);
              const double mdp = Gl - 1 > n0 ? sweep_md(Gl - 1) : (double)d0;
              d1 = Gl - 1 >= 1 ? (float)mdp : 0.0f;
              d0 = (float)mdl;
              converged = mdl <= p.conv_threshold;
              break;
            }
          } else {
            *sy.mine = 1 << 30;
          }
          // back to the stored grid; this time the block ends with sweep n0 + m
          SB_COUNT(15);
          __syncthreads(); // every wavefront has left the block
          load_row(e, tp, 16u * (unsigned)opaque(R), rsb);
#pragma unroll
          for (int t = 0; t < kTailMax; ++t)
#pragma unroll
            for (int k = 0; k < 2; ++k)
              if (t < a.T && tactive) tv[t][k] = Ttail[t * NR + opaque(tc0) + k];
          if (tactive) *(d2 *)(tE0 + opaque(tc0)) = d2{tv[0][0], tv[0][1]};
          first_words(x, lane);
        }
        if (converged || n_sweeps >= p.iter_limit) break;
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    SB_STAMP(3);

    // grid back to HBM.  Zone sums (A is dead in every wavefront after the barrier): every lane adds its cells
    // into its own slot of the zone (a.zs_off: the slots of zone z; only the (wavefront, lane) pairs that own a
    // cell of the zone have one); zone Z collects every cell outside a zone, so that the sum of all slots is
    // the grid sum.
    double *zs = A;
    const int zs_n = a.zs_off[a.Z + 1], zs_dump = a.zs_off[a.Z];
    __syncthreads();
    for (int i = threadIdx.x; i < zs_n; i += blockDim.x) zs[i] = 0.0;
    __syncthreads();
    {
      unsigned long long zw[kZA + 1];
      const unsigned long long *zm = a.zmapS + map_w + opaque(lane);
#pragma unroll
      for (int g = 0; g < kZA; ++g) zw[g] = zm[g * 64];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < kTailMax; ++t) // the tail rows hold no zone cells (sb_create checks): all into zone Z
        if (t < a.T && tactive) {
          *(d2 *)(Ttail + t * NR + opaque(tc0)) = d2{tv[t][0], tv[t][1]};
          __hip_atomic_fetch_add(zs + zs_dump + R, tv[t][0] + tv[t][1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
      const double *np_ = a.temp + (size_t)(bn < a.B ? bn : b) * a.state_doubles;
      unsigned ho = 16u * (unsigned)opaque(R);
      d2 pre[kHA + 1];
#pragma unroll
      for (int k = 0; k <= kHA; ++k) pre[k] = d2{0.0, 0.0};
      hand_over<NR, NV, 0>(e, zw, zm, tp, np_, rsb, ho, pre, zs);
    }
    __builtin_amdgcn_sched_barrier(0);
    SB_STAMP(4);
    __syncthreads(); // the zone sums are complete; the (ap, g) table is free
    if (bn < a.B) SB_LOAD_AUX(bn);
    __builtin_amdgcn_sched_barrier(0);

    if (first_w) { // hand the zone sums, the grid sum and the sweep count to k_post: a lane per zone adds the
      // zone's slots in slot order (deterministic), four at a time
      double gacc = 0.0;
      for (int zb = 0; zb < a.Z; zb += 64) {
        const int zz = zb + lane;
        double v = 0.0;
        if (zz < a.Z) {
          const int s0 = a.zs_off[zz], s1 = a.zs_off[zz + 1];
          double v1 = 0.0, v2 = 0.0, v3 = 0.0;
          int i = s0;
          for (; i + 4 <= s1; i += 4) { v += zs[i]; v1 += zs[i + 1]; v2 += zs[i + 2]; v3 += zs[i + 3]; }
          for (; i < s1; ++i) v += zs[i];
          v = (v + v1) + (v2 + v3);
          a.zsum[(size_t)b * a.Z + zz] = v;
          gacc += v;
        }
      }
      // zone Z (the cells outside every zone: a slot per row of every wavefront) feeds the grid sum only: all lanes
      // share it (one lane alone: 64 W dependent rounds -- as long as all the zones together)
      for (int i = zs_dump + lane; i < zs_n; i += 64) gacc += zs[i];
      const double gsum = wave_sum(gacc);
      if (lane == 0) {
        a.gsum[b] = gsum + (double)a.n_ring * t_now;
        a.nsw[b] = n_sweeps | (converged << 16);
      }
      SB_STAMP(5);
#ifdef SB_PHASE_STAMPS
      if (a.dbg && blockIdx.x == 0 && iter == 3 && lane == 0) a.dbg[9] = n_sweeps;
#endif
    }
    // the next building's first barrier separates this reduce from the next A pass
  }
#undef SB_STAMP
#undef SB_COUNT
#undef SB_LOAD_AUX
}

template <int NR>
int launch(const Dev &d, hipStream_t stream, bool prepare) {
  if (prepare)
    return (int)hipFuncSetAttribute((const void *)k_sweep_band<NR>, hipFuncAttributeMaxDynamicSharedMemorySize, d.lds_reg_bytes);
  hipLaunchKernelGGL((k_sweep_band<NR>), dim3(d.sweep_wgs), dim3(d.RS), (size_t)d.lds_reg_bytes, stream, d); // d.RS = 64 W threads
  return (int)hipGetLastError();
}

} // namespace
} // namespace sb
