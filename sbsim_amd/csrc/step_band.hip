// step_band.hip -- the sweep kernel for floor plans of 67..130 rows and <= 80 columns inside their
// exterior ring: TWO wavefronts per building (one workgroup), the grid in their registers, one row
// per lane -- wavefront 0 owns rows 0..63, wavefront 1 rows 64..127, rows 128.. (at most two) are
// finished by the affine scan of sweep_common.h.  simulator.py:278-371.
//
// step_two.hip holds such a plan in ONE wavefront (two rows per lane): its A = ap*Tprev + g fills
// the LDS share of two buildings, so two of a CU's four SIMDs idle.  Here a building's rows are split
// over two SIMDs; a CU still holds two buildings, all four SIMDs run, and a step is the one-row step
// of step_roll.hip (no AGPR traffic, no second cell).  Layout and schedule per wavefront are those of
// step_reg.hip / step_roll.hip: lane l owns one row, column c in register slot (c + l) mod NR, every
// lane works on slot s mod NR at step s, neighbours by DPP; consecutive sweeps are overlapped in
// predicted BLOCKS as in step_two.hip (ramp-up, rolling periods of NR steps, final period; a block
// that ran past the step's last sweep is run again from the stored grid with the right count).
//
// The two wavefronts of a building:
//   * Wavefront 1 runs >= 64 steps behind wavefront 0 (row 64's upper neighbours are row 63's new
//     values) and <= NR + 62 steps behind (row 63's lower neighbours are row 64's values of the
//     previous sweep).  Both bounds are kept by two progress counters in LDS, checked every kGrp steps.
//   * The seam values travel through LDS: every step each wavefront writes ONE ds_write_b64 whose
//     per-lane base address sends lane 63's result (wavefront 0) / lane 0's and lane 63's results
//     (wavefront 1) to the seam rows and every other lane's to a scratch strip; the step's immediate
//     offset 8 * (s mod NR) does the indexing.  Readers use two uniform ds_read_b64 per step whose
//     results enter as the DPP `old` operand of lane 0 (upper neighbour) and lane 63 (lower neighbour).
//     Both wavefronts execute the SAME code (their roles differ in four base addresses), so the
//     unrolled periods are in the instruction cache once.
//   * max|delta| of a sweep is the maximum over both wavefronts (and the tail rows): each publishes its
//     part per sweep; the decisions -- roll on, end the block, run it again -- are functions of the
//     published sequence only, evaluated by both wavefronts alike.  Wavefront 0 reaches a decision
//     point about one period before wavefront 1's part of the last sweep exists, so far from
//     convergence the decision uses the sweeps before it (may_roll extrapolates one sweep further);
//     only near convergence does wavefront 0 wait for the latest value.
// The iterates and the sweep count are always those of the plain schedule (tests: oracle twins,
// step_lds.hip on the same batch); the prediction only decides the speed.
#include <type_traits>

#include "sweep_common.h"

namespace sb {
namespace {

using namespace sweep;

#ifndef SB_BAND_EXP
#define SB_BAND_EXP 0
#endif
constexpr int kSets = 32;  // entries of the coefficient-set table (at LDS address 0)
constexpr int kWA = 11;    // class words (one step each) are read this many steps ahead (an L2 hit is ~800 cycles away)
constexpr int kZA = 8;     // zone-offset words (four slots each) read ahead in the hand-over
constexpr int kGrp = 8;    // steps between two checks of the other wavefront's progress
constexpr int kHist = 8;   // ring of published max|delta| parts, by sweep number
static_assert(kWA % 8 != 0 && (63 + kWA) % 8 != 0, "step_set: the start words leave the offset inside a chunk");

// Slots of A kept in LDS (the rest: registers) = A's row stride: even (the steps go in pairs, one
// ds_read_b128 per pair) and 2 mod 4 doubles (16-byte aligned rows, 16 lanes' ds_read_b128 cover all
// banks).  66: two wavefronts' A (67.6 KB) + tables + seam rows stay under 80 KB -- two buildings per
// CU -- and the zone-sum scratch of 127 zone rows x 65 (66.0 KB) fits inside A.  A row holds slot j at
// position (j + 1) mod NR: the pair of an odd step is 16-byte aligned.
constexpr int lds_slots(int NR) { return 66; }
constexpr int seam_region(int NR) { return NR + 72; }                    // doubles per seam row: 64 finite ones in front (steps < 63)

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
// The first kWA words of a block (steps 0 ..) and of a rolling period (steps 63 ..) are the same every
// time: read once per kernel, kept in registers (a period would otherwise start with an L2 round trip).
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
struct StartWords {
  int first[kWA], period[kWA]; // AGPRs
};
__device__ __forceinline__ void load_start_words(StartWords &sw, const Ctx &x, int lane) {
#pragma unroll
  for (int k = 0; k < kWA; ++k) {
    sw.first[k] = to_agpr((int)*(const unsigned *)(x.cmap + (unsigned)(lane * 4 + k * 256)));
    sw.period[k] = to_agpr((int)*(const unsigned *)(x.cmap + (unsigned)(lane * 4 + (63 + k) * 256)));
  }
}
__device__ __forceinline__ void first_words(Ctx &x, const StartWords &sw, int lane) { // words 0 .. kWA-1 of a sweep
  x.voff = (unsigned)opaque(lane * 4 + (kWA / 8) * 2048);
#pragma unroll
  for (int k = 0; k < kWA; ++k) x.w[k] = (unsigned)from_agpr(sw.first[k]);
}
__device__ __forceinline__ void period_words(Ctx &x, const StartWords &sw, int lane) { // a rolling period starts at step 63
  x.voff = (unsigned)opaque(lane * 4 + ((63 + kWA) / 8) * 2048);
#pragma unroll
  for (int k = 0; k < kWA; ++k) x.w[(63 + k) % (kWA + 1)] = (unsigned)from_agpr(sw.period[k]);
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
template <int NR, int S>
__device__ __forceinline__ lds_d2 step_set(Ctx &x) { // the coefficient set of step S; reads word S + kWA
  if constexpr (S + kWA < NR + 63) {
    if constexpr ((S + kWA) % 8 == 0 && S > 0 && S != 63) { // (the start words leave the offset at its chunk)
      x.voff += 2048u;
      asm volatile("" : "+v"(x.voff)); // a running offset: nothing for the compiler to hoist
    }
    x.w[(S + kWA) % (kWA + 1)] = *(const unsigned *)(x.cmap + x.voff + (unsigned)(((S + kWA) % 8) * 256));
  }
  return (lds_d2)x.w[S % (kWA + 1)];
}

// LDS reads of the pair (S, S + 1), S odd.
template <int NR, int S, int NAR>
__device__ __forceinline__ void load_pair(PairBuf &p, Ctx &x, const ARegs<NAR> &Areg) {
  static_assert(S % 2 == 1, "pairs start at odd steps");
  const lds_d2 s0 = step_set<NR, S>(x), s1 = step_set<NR, S + 1>(x);
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
template <int NR, int S, bool ROLL>
__device__ __forceinline__ void step(double (&e)[NR], d2 ud, d2 lr, double A, double rU, double rD, Acc &acc, const Ctx &x) {
  constexpr int r = S % NR, rm = (S + NR - 1) % NR, rp = (S + 1) % NR;
  const double Dn = wave_shift1<0x130, true>(e[rp], rD);
  double t;
  asm("v_fma_f64 %0, %1, %2, %3" : "=&v"(t) : "v"(ud.y), "v"(Dn), "v"(A));
  t = fma(lr.y, e[rp], t);
  const double U = wave_shift1<0x138, true>(e[rm], rU);
  t = fma(lr.x, e[rm], t);
  const double nv = fma(ud.x, U, t);
  double sel = nv;
  if constexpr (ROLL && S >= NR) {
    constexpr int J = S - NR;
    const double d = nv - e[r];
    // +|d| in the lanes still in sweep k, -|d| in the lanes already in sweep k+1
    const double sd = __hiloint2double((__double2hiint(d) & 0x7fffffff) | acc.sg, __double2loint(d));
    acc.cur = fmax(acc.cur, sd);
    acc.neg = fmin(acc.neg, sd);
    asm volatile("" : "+v"(acc.neg));
    if constexpr (J + 1 < 63) // lane J + 1 starts its next sweep at the next step (lane 0 keeps its bit)
      acc.sg = __builtin_amdgcn_update_dpp(acc.sg, acc.sg, 0x138, 0xf, 0xf, false);
  } else {
    if constexpr (S < 63) sel = lanes_upto<S>() ? nv : e[r];
    else if constexpr (S >= NR) sel = lanes_upto<S - NR>() ? e[r] : nv;
    acc.cur = fmax(acc.cur, fabs(sel - e[r]));
  }
  asm volatile("" : "+v"(acc.cur)); // here, not after the sweep (the maxima would keep every delta of the sweep alive)
  e[r] = sel;
#if SB_BAND_EXP != 1 // (timing experiments: 1 = no publish, 2 = no seam reads, 3 = neither)
#if SB_BAND_EXP != 3
  *(lds_dw)(x.pub + 8u * r) = sel; // lane 63 / lane 0: the seam rows; every other lane: its scratch strip
#endif
#endif
}

// The other wavefront's progress (steps completed in this block), before this wavefront runs steps
// t .. t + n - 1 (and reads one step ahead).  need_off: + 64 (wavefront 1: row 63's new values must
// exist) or - NR - 62 (wavefront 0: row 64's values of the previous sweep must exist, and wavefront 1
// must have read the upper neighbours this wavefront is about to overwrite).
__device__ __forceinline__ void sync_steps(lds_vi mine, lds_vi theirs, int done, int need, long long *dbg = nullptr) {
  *mine = done;
  if (need > 0) {
#ifdef SB_BAND_COUNT_SPINS // developer aid: how long do the wavefronts wait for each other?
    int spins = 0;
    while (__builtin_amdgcn_readfirstlane(*theirs) < need) { __builtin_amdgcn_s_sleep(1); ++spins; }
    if (dbg && spins && (threadIdx.x & 63) == 0) atomicAdd((unsigned long long *)dbg + (threadIdx.x >> 6 ? 7 : 6), (unsigned long long)spins);
#else
    (void)dbg;
    while (__builtin_amdgcn_readfirstlane(*theirs) < need) __builtin_amdgcn_s_sleep(1);
#endif
  }
  asm volatile("" ::: "memory");
}

// Pairs S, S + 2, .. < S1 (S odd) of a block whose period started at local step tb (the block's step
// count is tb + S); the LDS reads of the next pair are issued before the arithmetic of a pair.
template <int NR, int S, int S1, bool ROLL, int NAR>
__device__ __forceinline__ void run_pairs(double (&e)[NR], const ARegs<NAR> &Areg, PairBuf (&pb)[2], Ctx &x, Acc &acc,
                                          lds_vi mine, lds_vi theirs, int tb, int need_off, int last_step, long long *dbg) {
  if constexpr (S < S1) {
    if constexpr (!ROLL && S >= NR && (S - NR) % 4 == 1)
      if (S > last_step) return; // uniform: only lanes without rows are left
    if constexpr (S % kGrp == (S < 63 ? 1 : 63 % kGrp)) { // the group's last pair reads ahead for the pair after it
      constexpr int n = (S1 - S < kGrp ? S1 - S : kGrp) + 2;
      sync_steps(mine, theirs, tb + S, tb + S + n + need_off, dbg);
    }
    PairBuf &cur = pb[((S - 1) / 2) & 1], &nxt = pb[((S + 1) / 2) & 1];
    if constexpr (S + 2 < NR + 63) load_pair<NR, S + 2>(nxt, x, Areg);
    __builtin_amdgcn_sched_barrier(0);
    step<NR, S, ROLL>(e, cur.ud0, cur.lr0, cur.A.x, cur.rU.x, cur.rD.x, acc, x);
    __builtin_amdgcn_sched_barrier(0);
    step<NR, S + 1, ROLL>(e, cur.ud1, cur.lr1, cur.A.y, cur.rU.y, cur.rD.y, acc, x);
    __builtin_amdgcn_sched_barrier(0);
    run_pairs<NR, S + 2, S1, ROLL>(e, Areg, pb, x, acc, mine, theirs, tb, need_off, last_step, dbg);
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
template <int NR, int NAR>
__device__ __forceinline__ void a_pass(const double (&e)[NR], ARegs<NAR> &Areg, double *Aw, const char *tapg,
                                       const unsigned long long *amap) {
  constexpr int NL = lds_slots(NR), NWD = NR / 4;
  static_assert(NR % 4 == 0, "a_pass: four slots per word");
  amap += opaque(0);
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
      const double av = fma(pg[k].x, e[j], pg[k].y);
      constexpr int q = (j + 1) % NR; // slot j's position in the lane's row
      if constexpr (q < NL) Aw[q] = av;
      else Areg.template set<q - NL>(av);
    });
    __builtin_amdgcn_sched_barrier(0);
  });
}

// The end of a building's step, two slots at a time: store them, add them to their zone sums (LDS),
// load the same slots of the next building.  HBM state layout [NR / 2][128 rows][2].
template <int NR, int J>
__device__ __forceinline__ void hand_over(double (&e)[NR], unsigned long long (&zw)[kZA + 1], const unsigned long long *zmap,
                                          double *tp, const double *np_, double *zs) {
  if constexpr (J < NR) {
    if constexpr (J % 4 == 0 && J / 4 + kZA < NR / 4) zw[(J / 4 + kZA) % (kZA + 1)] = zmap[(J / 4 + kZA) * 64];
    const unsigned long long w = zw[(J / 4) % (kZA + 1)];
    const unsigned i0 = (unsigned)((w >> (16 * (J & 3))) & 0xffffull), i1 = (unsigned)((w >> (16 * ((J + 1) & 3))) & 0xffffull);
    *(d2 *)(tp + J * 128) = d2{e[J], e[J + 1]};
    __hip_atomic_fetch_add(zs + i0, e[J], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_fetch_add(zs + i1, e[J + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    const d2 nv = *(const d2 *)(np_ + J * 128);
    e[J] = nv.x;
    e[J + 1] = nv.y;
    if constexpr ((J & 7) == 6) __builtin_amdgcn_sched_barrier(0);
    hand_over<NR, J + 2>(e, zw, zmap, tp, np_, zs);
  }
}

extern __shared__ __attribute__((aligned(16))) double lds[];

// LDS (doubles): [tabc 4 kSets][tapg 2 ts] | r_seam: [zero][up][dn][tl][tE0] (seam_region each)
// [scratch 2 x (64 + NR + 8)] | r_xchg: sync words | r_A: A [2][64][AS] (after the sweeps: zone sums)
template <int NR>
__global__ void __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(1, 1))) k_sweep_band(Dev a) {
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); // wavefront-uniform: SGPRs
  constexpr int kNL = lds_slots(NR), kAS = kNL, kNAR = NR - kNL > 0 ? NR - kNL : 1, kRG = seam_region(NR);
  static_assert(NR % 4 == 0 && NR >= 68, "slots");

  double *tabc = lds;                    // [kSets][4]: bU bD bL bR per coefficient set
  double *tapg = lds + 4 * kSets;        // [ts][2]: (ap, g) per class; g of this building
  double *seam = lds + a.r_seam;
  double *S_zero = seam + 64, *S_up = seam + kRG + 64, *S_dn = seam + 2 * kRG + 64, *S_tl = seam + 3 * kRG + 64;
  double *tE0 = seam + 4 * kRG + 64;     // the first tail row by column (zeros without tail rows)
  double *scratch = seam + 5 * kRG;      // [2][64 + NR + 8]
  int *sync = (int *)(lds + a.r_xchg);   // [2][64] progress counters
  double *mrec = lds + a.r_xchg + 64;    // [2][kHist] records {max|delta| part, sweep number}: 16 bytes each
  int *misc = sync + 2 * (64 + 4 * kHist); // [0]: the next building
  double *A = lds + a.r_A;               // [2][64][kAS]; after the sweeps: zone sums [Z + 1][65]
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
  const bool tails = a.T > 0;
  // wavefront 0: upper neighbours of lane 0 do not count (zeros), lower neighbours of lane 63 = row 64;
  // wavefront 1: upper neighbours of lane 0 = row 63's new values, lower neighbours of lane 63 = the
  // first tail row (or nothing)
  x.ubase = lds_addr(wv == 0 ? S_zero : S_up);
  x.dbase = lds_addr(wv == 0 ? S_dn : (tails ? tE0 : S_zero)) - 8u * 63u;
  {
    double *strip = scratch + (size_t)wv * (64 + NR + 8) + lane;
    double *target = strip;
    if (wv == 0 && lane == 63) target = S_up;
    if (wv == 1 && lane == 0) target = S_dn;
    if (wv == 1 && lane == 63) target = S_tl;
    x.pub = lds_addr(target);
  }
  lds_vi prog_mine = (lds_vi)(lds_addr(sync + wv * 64 + lane)), prog_theirs = (lds_vi)(lds_addr(sync + (wv ^ 1) * 64 + lane));
  const int need_off = wv == 0 ? -NR - 62 : 64;
  const int rows_mine = a.lw[wv];        // lanes that own rows
  const int last_step = NR + rows_mine - 2;
  // the lane's tail cells (wavefront 1; static per floor plan)
  const bool tactive = wv == 1 && tails && tail_col<NR>(lane, 0) >= 0;
  const int tc0 = tactive ? tail_col<NR>(lane, 0) : 0;
  int tset[kTailMax];
#pragma unroll
  for (int t = 0; t < kTailMax; ++t) {
    tset[t] = (a.ncset - 1) * 32 * 0x10001; // the pad set (the table's last)
    if (t < a.T && tactive) tset[t] = ((int)a.tcset[t * NR + tc0] << 2) | ((int)a.tcset[t * NR + tc0 + 1] << 18);
  }
  StartWords sw;
  load_start_words(sw, x, lane);
  const unsigned long long *amap = a.amapS + (size_t)wv * (NR / 4) * 64 + lane;
  const unsigned long long *zmap = a.zmapS + (size_t)wv * (NR / 4) * 64 + lane;
  const int R = wv * 64 + lane; // the lane's row of the state [NR / 2][128][2]

#define SB_STAMP(i) do { if (a.dbg && blockIdx.x == 0 && iter == 3 && threadIdx.x == 0) a.dbg[i] = (long long)__builtin_readcyclecounter(); } while (0)

  double e[NR];
  double nx_tnow = 0.0, nx_lo = 0.0, nx_hi = 0.0;
  double tv[kTailMax][2] = {{0.0, 0.0}, {0.0, 0.0}};
#define SB_LOAD_AUX(bb)                                                                          \
  do {                                                                                           \
    nx_tnow = a.bld[(bb)].t_now;                                                                 \
    nx_lo = a.scal[(size_t)(bb) * kNScal + 16];                                                  \
    nx_hi = a.scal[(size_t)(bb) * kNScal + 17];                                                  \
    for (int c = threadIdx.x; c < a.ts; c += 128) tapg[2 * c + 1] = a.gtabg[(size_t)(bb) * a.ts + c]; \
    const double *tt_ = a.temp + (size_t)(bb) * a.state_doubles + NR * 128;                      \
    _Pragma("unroll") for (int t = 0; t < kTailMax; ++t)                                         \
      _Pragma("unroll") for (int k = 0; k < 2; ++k)                                              \
        if (t < a.T && tactive) tv[t][k] = tt_[t * NR + tc0 + k];                                \
  } while (0)
  if ((int)blockIdx.x < a.B) {
    const double *tp_ = a.temp + (size_t)blockIdx.x * a.state_doubles + 2 * R;
#pragma unroll
    for (int j = 0; j < NR; j += 2) {
      const d2 v = *(const d2 *)(tp_ + j * 128);
      e[j] = v.x;
      e[j + 1] = v.y;
    }
    SB_LOAD_AUX(blockIdx.x);
  }
  int iter = 0;
  for (int b = blockIdx.x, bn = 0; b < a.B; b = bn, ++iter) {
    if (threadIdx.x == 0) misc[0] = a.sweep_wgs + atomicAdd(a.next_b, 1);
    SB_STAMP(0);
    first_words(x, sw, lane);
    double *Ttail = a.temp + (size_t)b * a.state_doubles + NR * 128; // [T][NR]
    double *tp = a.temp + (size_t)b * a.state_doubles + 2 * R;
    const double t_now = nx_tnow;
    // exterior-space cells outside the trim box all become t_now in the first sweep
    // (simulator.py:256-258); their largest |delta| follows from their extreme values
    const double ring_d = a.n_ring > 0 ? fmax(fabs(t_now - nx_lo), fabs(t_now - nx_hi)) : 0.0;
    if (tactive) *(d2 *)(tE0 + tc0) = d2{tv[0][0], tv[0][1]};
    __syncthreads(); // the (ap, g) table and the tail row are in LDS; the previous building's zone sums are read
    bn = __builtin_amdgcn_readfirstlane(*(volatile int *)misc);
    SB_STAMP(1);
    double At[kTailMax][2] = {{0.0, 0.0}, {0.0, 0.0}}; // A of the lane's tail cells
#pragma unroll
    for (int t = 0; t < kTailMax; ++t)
#pragma unroll
      for (int k = 0; k < 2; ++k)
        if (t < a.T && tactive) {
          const d2 pg = *(const d2 *)((const char *)tapg + 16 * (int)a.tcls[t * NR + tc0 + k]);
          At[t][k] = fma(pg.x, tv[t][k], pg.y);
        }
    ARegs<kNAR> Areg;
    a_pass<NR>(e, Areg, A + ((size_t)wv * 64 + lane) * kAS, (const char *)tapg, amap);
    __builtin_amdgcn_sched_barrier(0);
    SB_STAMP(2);

    int n_sweeps = 0, converged = 0;
    {
      PairBuf pb[2];
      Acc acc;
      const float thr = (float)p.conv_threshold;
      const int prev_sweeps = __builtin_amdgcn_readfirstlane(a.nsw[b] & 0xffff); // of this building's previous step
      // max |delta| of sweep G of this step = the maximum of both wavefronts' parts (G >= 1).  A part is
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
      // the last three values read, by sweep number: a decision needs one new one (LDS round trip)
      int c_g = -10;
      double c1 = -1.0, c2 = -1.0, c3 = -1.0; // md(c_g), md(c_g - 1), md(c_g - 2); < 0: not read yet
      auto sweep_md = [&](int G) -> double {
        if (G == c_g && c1 >= 0.0) return c1;
        if (G == c_g - 1 && c2 >= 0.0) return c2;
        if (G == c_g - 2 && c3 >= 0.0) return c3;
        const unsigned r0 = lds_addr(mrec + 2 * (G % kHist)), r1 = lds_addr(mrec + 2 * (kHist + G % kHist));
        double m;
        for (;;) { // waits for both parts
          const d2 p0 = *(lds_vd2)r0, p1 = *(lds_vd2)r1;
          if (__builtin_amdgcn_readfirstlane(__double2loint(p0.y)) == G && __builtin_amdgcn_readfirstlane(__double2loint(p1.y)) == G) {
            m = fmax(p0.x, p1.x);
            break;
          }
#ifdef SB_BAND_COUNT_SPINS
          if (a.dbg && lane == 0) atomicAdd((unsigned long long *)a.dbg + 8, 1ull);
#endif
          __builtin_amdgcn_s_sleep(1);
        }
        asm volatile("" ::: "memory");
        if (G == c_g + 1) { c3 = c2; c2 = c1; c1 = m; c_g = G; }
        else if (G > c_g + 1) { c_g = G; c1 = m; c2 = c3 = -1.0; }
        else if (G == c_g) c1 = m;
        else if (G == c_g - 1) c2 = m;
        else if (G == c_g - 2) c3 = m;
        return m;
      };
      // the end of this wavefront's sweep G: the tail rows (wavefront 1), its part of max |delta|
      auto sweep_end = [&](int G) {
        double dm = acc.cur;
        if (tails && wv == 1) { // row 127's new values by column c sit at S_tl[(c + 63) mod NR]
          const int c0 = tactive ? tc0 : 0;
          const double U0 = *(lds_d)(lds_addr(S_tl) + 8u * (unsigned)((c0 + 63) % NR)), U1 = *(lds_d)(lds_addr(S_tl) + 8u * (unsigned)((c0 + 64) % NR));
          dm = fmax(dm, tail_pass<NR>(a.T, tactive, tE0 + tc0, U0, U1, tv, tset, At));
        }
        if (G == 1 && wv == 0) dm = fmax(dm, ring_d);
        publish_part(G, dm);
      };
      float d1 = 0.0f, d0 = 0.0f; // max |delta| of the step's last two sweeps as of the last block's end
#pragma nounroll
      for (;;) { // blocks.  simulator.py:348-368
        const int n0 = n_sweeps;
        // does this block roll at all?  As in step_two.hip: by the decay so far, or -- a step's first
        // block -- by the previous step's count (a hint; a step that converges sooner is found out)
        int roll0 = n0 >= 2 ? (int)(sweeps_to_go(d1, d0, thr, a.pred_haste) > a.pred_slack)
                            : (int)(n0 == 0 && prev_sweeps >= 6 && a.pred_first > 1);
        roll0 = __builtin_amdgcn_readfirstlane(roll0) && n0 + 2 <= p.iter_limit;
        if (a.dbg && threadIdx.x == 0) atomicAdd((unsigned long long *)a.dbg + (roll0 ? 13 : 14), 1ull); // developer aid: blocks / single sweeps
        if (roll0 && n0 > 0) { // the grid as of n0 sweeps, should the block overrun (before any sweep: Tprev is still there)
#pragma unroll
          for (int j = 0; j < NR; j += 2) *(d2 *)(tp + j * 128) = d2{e[j], e[j + 1]};
#pragma unroll
          for (int t = 0; t < kTailMax; ++t)
            if (t < a.T && tactive) *(d2 *)(Ttail + t * NR + tc0) = d2{tv[t][0], tv[t][1]};
        }
        int m = 0; // > 0: the block is being run again and ends with its m-th sweep
        int q = 0; // rolling periods completed
#pragma nounroll
        for (;;) { // at most twice
          // block start: both wavefronts are here; progress 0, no published parts of this block yet;
          // row 64's current values for wavefront 0's first sweep
          __syncthreads();
          *prog_mine = 0;
          if (wv == 1 && lane == 0) {
#pragma unroll
            for (int c = 0; c < NR; ++c) S_dn[c] = e[c];
          }
          if (threadIdx.x < 2 * kHist) *(lds_vi)(lds_addr(mrec + 2 * threadIdx.x) + 8u) = 0; // no part of this block is published
          c_g = -10; // (the same sweeps of a block that is run again have the same values: this is tidiness)
          __syncthreads();
          __builtin_amdgcn_sched_barrier(0);
#define SB_STAMP2(i) do { if (a.dbg && blockIdx.x == 0 && iter == 3 && q == 4 && n0 == 0 && threadIdx.x == 0) a.dbg[i] = (long long)__builtin_readcyclecounter(); } while (0)
          acc.cur = 0.0;
          acc.neg = 0.0;
          acc.sg = lane == 0 ? (int)0x80000000 : 0;
          n_sweeps = n0;
          q = 0;
          sync_steps(prog_mine, prog_theirs, 0, 3 + need_off); // wavefront 1: row 63's first new values must exist
          { // step 0 (lane 0, column 0) on its own: pairs start at odd steps
            const lds_d2 st = step_set<NR, 0>(x);
            const d2 ud = st[0], lr = st[1];
            const double A0 = *(lds_d)(x.arow + 8u), rU0 = *(lds_d)(x.ubase + 8u * 63u), rD0 = *(lds_d)(x.dbase);
            load_pair<NR, 1>(pb[0], x, Areg);
            step<NR, 0, false>(e, ud, lr, A0, rU0, rD0, acc, x);
          }
          __builtin_amdgcn_sched_barrier(0);
          run_pairs<NR, 1, 63, false>(e, Areg, pb, x, acc, prog_mine, prog_theirs, 0, need_off, last_step, a.dbg); // ramp-up; reads ahead for the pair (63, 64)
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
              // published history: sweeps <= G - 2 are complete in both wavefronts without waiting (the
              // other one is at most ~1 period away); G - 1 may still be on its way in wavefront 1
              float h3 = d1, h2 = d0; // sweeps G - 3, G - 2 (before this block: the last block's)
              if (G - 2 > n0) {
                const double md2 = sweep_md(G - 2);
                if (md2 <= p.conv_threshold) { overrun = true; m = G - 2 - n0; break; }
                h2 = (float)md2;
                h3 = G - 3 > n0 ? (float)sweep_md(G - 3) : d0;
              } else if (G - 2 == n0) {
                h2 = d0; h3 = d1;
              }
              const bool far = G - 2 >= 2 && G - 2 >= n0 && sweeps_to_go(h3, h2, thr, a.pred_haste) > a.pred_slack + 3.0f;
              if (far) go = 1;
              else if (G - 1 > n0) { // near convergence: wait for the latest complete sweep
                const double md1 = sweep_md(G - 1);
                if (md1 <= p.conv_threshold) { overrun = true; m = G - 1 - n0; break; }
                const float h1 = (float)md1;
                go = G - 1 >= 2 ? (int)(sweeps_to_go(h2, h1, thr, a.pred_haste) > a.pred_slack + 1.0f) : (int)(q + 1 < a.pred_first);
              } else { // nothing of this block is complete yet (q <= 1): sweep n0 is the last one known
                go = n0 >= 2 ? (int)(sweeps_to_go(d1, d0, thr, a.pred_haste) > a.pred_slack + (float)q) : (int)(q + 1 < a.pred_first);
              }
            }
            if (!__builtin_amdgcn_readfirstlane(go)) break;
            asm volatile("" : "+v"(x.arow), "+v"(x.ubase), "+v"(x.dbase), "+v"(x.pub)); // not loop invariants: nothing to hoist (and spill)
            __builtin_amdgcn_sched_barrier(0);
            SB_STAMP2(10); // the fifth rolling period of wavefront 0: its steps, then its end and the next decision
            run_pairs<NR, 63, NR + 63, true>(e, Areg, pb, x, acc, prog_mine, prog_theirs, q * NR, need_off, last_step, a.dbg);
            __builtin_amdgcn_sched_barrier(0);
            *prog_mine = q * NR + NR + 63; // every step of the period is done
            SB_STAMP2(11);
            if (a.dbg && blockIdx.x == 0 && iter == 3 && q == 5 && n0 == 0 && threadIdx.x == 0) a.dbg[12] = (long long)__builtin_readcyclecounter();
            ++q;
            ++n_sweeps;
            period_words(x, sw, lane);
            sweep_end(n0 + q);
            acc.cur = -acc.neg;
            acc.neg = 0.0;
            acc.sg = lane == 0 ? (int)0x80000000 : 0;
            load_pair<NR, 63>(pb[1], x, Areg); // after the tail scan: lane 63's lower neighbours are new
          }
          if (!overrun) {
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("" : "+v"(x.arow), "+v"(x.ubase), "+v"(x.dbase), "+v"(x.pub));
            run_pairs<NR, 63, NR + 63, false>(e, Areg, pb, x, acc, prog_mine, prog_theirs, q * NR, need_off, last_step, a.dbg); // the block's last sweep
            __builtin_amdgcn_sched_barrier(0);
            *prog_mine = 1 << 30; // the other wavefront needs nothing more from this one
            ++n_sweeps;
            first_words(x, sw, lane);
            sweep_end(n_sweeps);
            // the block's last two sweeps, complete: did the one before the last converge already?
            const int Gl = n_sweeps;
            if (m == 0) // a decision looks at the sweeps two (or one) before its own: the last three, here
              for (int j = Gl - 3 > n0 ? Gl - 3 : n0 + 1; j < Gl; ++j)
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
            *prog_mine = 1 << 30;
          }
          // back to the stored grid; this time the block ends with sweep n0 + m
          if (a.dbg && threadIdx.x == 0) atomicAdd((unsigned long long *)a.dbg + 15, 1ull);
          __syncthreads(); // both wavefronts have left the block
#pragma unroll
          for (int j = 0; j < NR; j += 2) {
            const d2 v = *(const d2 *)(tp + j * 128);
            e[j] = v.x;
            e[j + 1] = v.y;
          }
#pragma unroll
          for (int t = 0; t < kTailMax; ++t)
#pragma unroll
            for (int k = 0; k < 2; ++k)
              if (t < a.T && tactive) tv[t][k] = Ttail[t * NR + tc0 + k];
          if (tactive) *(d2 *)(tE0 + tc0) = d2{tv[0][0], tv[0][1]};
          first_words(x, sw, lane);
        }
        if (converged || n_sweeps >= p.iter_limit) break;
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    SB_STAMP(3);

    // grid back to HBM.  Zone sums (A is dead in both wavefronts after the barrier): every lane adds
    // its cells into its own column of zs[zone]; row Z collects every cell outside a zone.
    double *zs = A;
    constexpr int ZRS = 65;
    __syncthreads();
    for (int i = threadIdx.x; i < (a.Z + 1) * ZRS; i += 128) zs[i] = 0.0;
    __syncthreads();
    {
      unsigned long long zw[kZA + 1];
      const unsigned long long *zm = zmap + opaque(0);
#pragma unroll
      for (int g = 0; g < kZA; ++g) zw[g] = zm[g * 64];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < kTailMax; ++t) // the tail rows hold no zone cells (sb_create checks): all into row Z
        if (t < a.T && tactive) {
          *(d2 *)(Ttail + t * NR + tc0) = d2{tv[t][0], tv[t][1]};
          __hip_atomic_fetch_add(zs + a.Z * ZRS + lane, tv[t][0] + tv[t][1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
      const double *np_ = a.temp + (size_t)(bn < a.B ? bn : b) * a.state_doubles + 2 * R;
      hand_over<NR, 0>(e, zw, zm, tp, np_, zs);
    }
    __builtin_amdgcn_sched_barrier(0);
    SB_STAMP(4);
    __syncthreads(); // the zone sums are complete; the (ap, g) table is free
    if (bn < a.B) SB_LOAD_AUX(bn);
    __builtin_amdgcn_sched_barrier(0);

    if (wv == 0) { // hand the zone sums, the grid sum and the sweep count to k_post
      // 16 zones x 4 column groups per pass: lane (zone = lane & 15, group = lane >> 4) adds every
      // fourth column of its zone, two xor-shuffles combine the groups
      double gacc = 0.0;
      for (int zb = 0; zb <= a.Z; zb += 16) {
        const int zz = zb + (lane & 15), g = lane >> 4;
        const double *zr = zs + (size_t)(zz <= a.Z ? zz : a.Z) * ZRS + g;
        double part[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) part[k] = zr[4 * k]; // sixteen independent reads, then one wait
        double v = 0.0;
#pragma unroll
        for (int k = 0; k < 16; ++k) v += part[k];
        if (zz > a.Z) v = 0.0;
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
      SB_STAMP(5);
      if (a.dbg && blockIdx.x == 0 && iter == 3 && lane == 0) a.dbg[9] = n_sweeps;
    }
    // the next building's first barrier separates this reduce from the next A pass
  }
#undef SB_STAMP
#undef SB_STAMP2
#undef SB_LOAD_AUX
}

template <int NR>
int launch(const Dev &d, hipStream_t stream, bool prepare) {
  if (prepare)
    return (int)hipFuncSetAttribute((const void *)k_sweep_band<NR>, hipFuncAttributeMaxDynamicSharedMemorySize, d.lds_reg_bytes);
  hipLaunchKernelGGL((k_sweep_band<NR>), dim3(d.sweep_wgs), dim3(128), (size_t)d.lds_reg_bytes, stream, d);
  return (int)hipGetLastError();
}

int dispatch(const Dev &d, hipStream_t stream, bool prepare) {
  if (d.NR == 76) return launch<76>(d, stream, prepare);
  if (d.NR == 80) return launch<80>(d, stream, prepare);
  return (int)hipErrorInvalidValue;
}

} // namespace

bool sweep_band_supported(int NR) { return NR == 76 || NR == 80; }
int sweep_band_lds_slots(int NR) { return lds_slots(NR); }
int sweep_band_seam_doubles(int NR) { return 5 * seam_region(NR) + 2 * (64 + NR + 8); }
int sweep_band_sync_doubles() { return 64 + kHist + 8 + 2 * kHist + 8; }
int sweep_band_set_table() { return kSets; }
int prepare_sweep_band(const Dev &d) { return dispatch(d, nullptr, true); }
int launch_sweep_band(const Dev &d, hipStream_t stream) { return dispatch(d, stream, false); }

} // namespace sb
