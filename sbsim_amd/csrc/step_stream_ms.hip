// step_stream_ms.hip -- the streaming sweep kernel (step_stream.hip: floor plans no CU holds, the grid in global
// memory) with SEVERAL Gauss-Seidel sweeps per pass over the grid.  simulator.py:278-371.
//
// step_stream.hip streams the grid through the wavefronts once per sweep: 28 B of traffic per cell and sweep, and a
// fill and a drain of the wavefront pipeline (64 steps per wavefront) per sweep.  Here a pass carries up to S = 4
// consecutive sweeps: the grid is read once and written once per PASS.
//
// Parallelogram tiling.  Wavefront w owns a band of 64 rows as before, but the band moves UP by one row per sweep:
// for sweep j of the pass, lane l works on row 64 w + l - j, at step t on column t - l - j.  Then everything sweep j
// needs from sweep j - 1 (the "old" values of the reference's in-place update) is one or two steps old and sits in
// this lane or in lane l - 1:
//     right  (row, col + 1) of sweep j - 1: lane l - 1's result of step t - 1
//     below  (row + 1, col) of sweep j - 1: this lane's result of step t - 1
//     old    (row, col)     of sweep j - 1: lane l - 1's result of step t - 2
//     above  (row - 1, col) of sweep j    : lane l - 1's result of step t - 1     left: this lane's
// -- registers and DPP moves, no memory.  The only values that cross a wavefront boundary go DOWN (lane 63 of
// wavefront w to lane 0 of wavefront w + 1: row 64 w + 63 - j of sweeps j and j - 1), so the one-sided progress
// protocol of step_stream.hip (wavefront w + 1 runs >= 64 steps behind wavefront w) carries over unchanged, with one
// seam row per sweep in LDS.  A = ap*Tprev + g and the class words of sweep j are the ones sweep 0 read two j steps
// earlier, j rows further up: they are read again (L1 / L2 hits: the HBM traffic stays one pass).
//
// The stop (simulator.py:360: after the first sweep whose max |delta| <= threshold) is decided per pass: every sweep
// of the pass keeps its own maximum.  A pass reads one grid and writes another (the building's state and a scratch
// grid per resident workgroup take turns), so when sweep j < S' - 1 of a pass of S' sweeps turns out to be the last
// one, the pass is run again from the same input with j + 1 sweeps -- the iterates and the sweep count are always
// those of the plain schedule; how many sweeps a pass carries (the decay of max |delta| extrapolated to the
// threshold, like step_two.hip / step_band.hip) only decides the speed.
#include "sb_device.h"

namespace sb {
namespace {

constexpr int kSets = 32;   // entries of the coefficient-set table (at LDS address 0)
constexpr int kPF = 8;      // steps between sweep 0's global loads and their use
constexpr int kQF = 4;      // ... the later sweeps' (A, class word): L1 / L2 hits
constexpr int kZC = 9;      // columns of the zone-sum scratch per zone (8 lane columns + 1: odd stride) -- step_stream.hip's (the planner asks it)
#ifndef SB_STREAM_S
#define SB_STREAM_S 4
#endif
constexpr int kS = SB_STREAM_S; // sweeps per pass, at most

typedef double d2 __attribute__((ext_vector_type(2)));
typedef const d2 __attribute__((address_space(3))) *lds_d2;
typedef volatile int __attribute__((address_space(3))) *lds_vi;

// (the compiler must not fold `word & 0xffff` into the load that fills the read-ahead ring: it then waits for the load at once)
__device__ __forceinline__ unsigned used_now(unsigned v) {
  asm volatile("" : "+v"(v));
  return v;
}

template <int CTRL>
__device__ __forceinline__ double dpp_seam(double x, double old) { // lanes without a source keep `old`
  const int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(x), CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(x), CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}

extern __shared__ __attribute__((aligned(16))) double lds[];

// LDS (doubles): [tabc 4 kSets][tapg 2 ts] | r_seam: up [kS][W][NS + 8], dn [W][NS + 8] | r_xchg: progress [16] ints,
// max|delta| parts [kS][16], misc [8], dummy [W][64] | r_A: zone sums [Z + 1][kZC]
template <int WMAX>
__global__ void __launch_bounds__(64 * WMAX) k_sweep_stream_ms(Dev a, double *Abuf, double *Ebuf) {
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int W = (int)(blockDim.x >> 6), NS = a.NR, RS = a.RS, NSP = NS + 8;
  double *tabc = lds;
  double *tapg = lds + 4 * kSets;
  double *up = lds + a.r_seam;                   // up[j][w][c]: row 64 w + 63 - j's value of sweep j at column c
  double *dn = up + (size_t)kS * W * NSP;        // dn[w][c]: row 64 w's value in the pass's input grid
  int *prog = (int *)(lds + a.r_xchg);           // [W] steps completed in this pass
  double *mpart = lds + a.r_xchg + 8;            // [kS][16] max |delta| of the wavefront's rows, per sweep of the pass
  int *misc = (int *)(lds + a.r_xchg + 8 + kS * 16); // [0]: the next building
  double *dummy = lds + a.r_xchg + 16 + kS * 16; // [W][64]: where the lanes that publish nothing write
  double *zs = lds + a.r_A;                      // [Z + 1][kZC]
  for (int i = threadIdx.x; i < a.lds_reg_bytes / 8; i += blockDim.x) lds[i] = 0.0;
  __syncthreads();
  for (int i = threadIdx.x; i < 4 * kSets; i += blockDim.x) tabc[i] = i < 4 * a.ncset ? a.csetab[i] : 0.0;
  for (int c = threadIdx.x; c < a.ts; c += blockDim.x) tapg[2 * c] = c <= a.ncls ? a.ctab[c * 8 + 4] : 0.0; // row `ncls`: the pad class
  __syncthreads();
  if ((unsigned)(size_t)(__attribute__((address_space(3))) double *)tabc != 0u) __builtin_trap(); // class words hold LDS addresses

  const sb_params &p = a.p;
  const int row = wv * 64 + lane;
  const int NW = NS + 63;                                          // steps of a wavefront per sweep (class words per wavefront)
  const unsigned *cmap0 = (const unsigned *)a.cmapS;               // [W][NW][64]: set offset | class * 16 << 16
  const unsigned short *zmap = (const unsigned short *)a.zmapS + (size_t)wv * NS * 64 + lane; // [W][NS][64]: zone (Z: none)
  lds_vi prog_mine = (lds_vi)(unsigned)(size_t)(__attribute__((address_space(3))) int *)(prog + wv);
  lds_vi prog_prev = (lds_vi)(unsigned)(size_t)(__attribute__((address_space(3))) int *)(prog + (wv > 0 ? wv - 1 : 0));
  const bool has_prev = wv > 0, has_next = wv + 1 < W;
  const double *dn_next = dn + (size_t)(has_next ? wv + 1 : wv) * NSP; // lane 63's lower neighbours of sweep 0
  double *dn_mine = dn + (size_t)wv * NSP;
  double *const pub_dummy = dummy + (size_t)wv * 64 + lane;

  // sweep j of a pass: the lane's row is `row - j` (rows above the grid: the lane idles), its class words are the ones
  // of that row's own (wavefront, lane): step t of sweep j = that lane's step t - 2 j (+ 64 when the row belongs to
  // the wavefront above)
  int rowj[kS], rowc[kS], o64[kS];
  const unsigned *cmj[kS]; // the row's own class words: word of ITS step u at [u * 64]; the lane's step t of sweep j is u = t - 2 j + o64
#pragma unroll
  for (int j = 0; j < kS; ++j) {
    rowj[j] = row - j;
    rowc[j] = rowj[j] < 0 ? 0 : rowj[j]; // (a lane above the grid idles: its loads go to row 0, its results nowhere)
    o64[j] = lane < j ? 64 : 0;          // the word of column c sits at step c + (row & 63) of the row's owner; the lane reaches column c at step c + lane + j
    cmj[j] = cmap0 + ((size_t)(rowc[j] >> 6) * NW) * 64 + (rowc[j] & 63);
  }
  // every global load of the step loop is UNCONDITIONAL (clamped indices): a load under a branch is waited for at the
  // branch's end, i.e. at once -- the whole memory latency in every step
  auto cw_at = [&](int j, int t) { int u = t - 2 * j + o64[j]; u = u < 0 ? 0 : (u > NW - 1 ? NW - 1 : u); return cmj[j][(size_t)u * 64]; };

  for (int b = blockIdx.x, bn = 0; b < a.B; b = bn) {
    if (threadIdx.x == 0) misc[0] = a.sweep_wgs + atomicAdd(a.next_b, 1);
    double *T = a.temp + (size_t)b * a.state_doubles;                  // [NS][RS]: the building's state
    double *Sc = Ebuf + (size_t)blockIdx.x * a.state_doubles;           // the scratch grid of this workgroup
    double *Ab = Abuf + (size_t)blockIdx.x * a.state_doubles;           // A = ap*Tprev + g, one grid per resident workgroup
    const double t_now = a.bld[b].t_now;
    const double ring_lo = a.scal[(size_t)b * kNScal + 16], ring_hi = a.scal[(size_t)b * kNScal + 17];
    const int prev_sweeps = a.nsw[b] & 0xffff;                          // of this building's previous step: the first pass's guess
    // exterior-space cells outside the trim box all become t_now in the first sweep
    // (simulator.py:256-258); their largest |delta| follows from their extreme values
    const double ring_d = a.n_ring > 0 ? fmax(fabs(t_now - ring_lo), fabs(t_now - ring_hi)) : 0.0;
    for (int c = threadIdx.x; c < a.ts; c += blockDim.x) tapg[2 * c + 1] = a.gtabg[(size_t)b * a.ts + c];
    for (int i = threadIdx.x; i < (a.Z + 1) * kZC; i += blockDim.x) zs[i] = 0.0;
    __syncthreads();
    bn = __builtin_amdgcn_readfirstlane(*(volatile int *)misc);
    // A = ap*Tprev + g for every cell of the lane's row, slot by slot (the class of slot s: the word of
    // the step at which the lane works on it, s - lane mod NS + lane)
    {
      const unsigned *cm = cmap0 + (size_t)wv * NW * 64 + lane;
      for (int s = 0; s < NS; ++s) {
        int t = s - lane;           // the lane's column at slot s
        if (t < 0) t += NS;
        const unsigned cw = cm[(size_t)(t + lane) * 64]; // step t + lane: column t
        const d2 pg = *(const d2 *)((const char *)tapg + (cw >> 16));
        Ab[(size_t)s * RS + row] = fma(pg.x, T[(size_t)s * RS + row], pg.y);
      }
    }
    int done = 0, converged = 0, Sp = min(min(max(prev_sweeps, 1), kS), p.iter_limit), redo = 0;
    int n_pass = 0, n_slots = 0; // developer counters (SBSIM_PHASE_TIMING=1): passes and sweep slots run for this building-step
    double md_last = 0.0, md_before = 0.0; // max |delta| of the last two sweeps done (the decay that predicts the next pass's length)
    const double *Ein = T;
    double *Eout = Sc;
    for (;;) { // passes
      // ---- pass prologue: progress 0, dn[w] <- row 64 w of the input grid (column c of row r sits in slot (c + r) mod NS)
      if (lane == 0) *prog_mine = 0;
      for (int s = lane; s < NS; s += 64) dn_mine[s] = Ein[(size_t)s * RS + 64 * wv]; // (lane 0's row: column s sits in slot s)
      __syncthreads();
      const int NWp = NW + Sp - 1; // steps of the pass
      ++n_pass; n_slots += Sp;
      double acc[kS], r[kS], pr[kS]; // per sweep: max |delta|, the lane's latest result, its result one step earlier
#pragma unroll
      for (int j = 0; j < kS; ++j) acc[j] = r[j] = pr[j] = 0.0;
      // sweep 0's streams: eR(t) = Ein[(t + 1) mod NS] (the right-hand neighbour at step t = the cell's own old value at
      // step t + 1), A(t), cw(t); kPF steps ahead, in a ring of registers.  The state layout puts column c of row r in
      // slot (c + (r & 63)) mod NS: sweep 0 at step t (column t - lane) works on slot t mod NS in every lane; sweep j
      // (row - j, column t - lane - j) on slot (t - 2 j) mod NS -- 64 further for the lanes whose row belongs to the
      // wavefront above (lane < j).
      double ring_e[kPF], ring_a[kPF];
      unsigned ring_c[kPF];
      const unsigned *cm0 = cmap0 + (size_t)wv * NW * 64 + lane;
      auto slot_of = [&](int t) { // t mod NS for t in [-NS, 3 NS)
        t = t < 0 ? t + NS : t;
        t = t >= NS ? t - NS : t;
        return t >= NS ? t - NS : t;
      };
#pragma unroll
      for (int k = 0; k < kPF; ++k) {
        ring_e[k] = Ein[(size_t)slot_of(k + 1) * RS + row];
        ring_a[k] = Ab[(size_t)slot_of(k) * RS + row];
        ring_c[k] = cm0[(size_t)min(k, NW - 1) * 64];
      }
      // the later sweeps' streams: (A, cw) of row - j at step t: slot (t + 64 w - 2 j) mod NS, kQF steps ahead
      double qa[kS][kQF];
      unsigned qc[kS][kQF];
#pragma unroll
      for (int j = 1; j < kS; ++j)
#pragma unroll
        for (int k = 0; k < kQF; ++k) {
          qa[j][k] = Ab[(size_t)slot_of(k - 2 * j + o64[j]) * RS + rowc[j]];
          qc[j][k] = cw_at(j, k);
        }
      double old0 = Ein[(size_t)slot_of(0) * RS + row]; // the lane's own old value at step 0
      for (int t0 = 0; t0 < NWp; t0 += kPF) {
        // the wavefront above must have published its last row for the columns lane 0 reaches here
        if (has_prev) {
          const int need = min(t0 + kPF + 63, NWp); // what lane 0 reads at step t was published at step t + 63 of the wavefront above
          int spins = 0;
          while (__builtin_amdgcn_readfirstlane(*prog_prev) < need) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1 << 26)) __builtin_trap(); // a protocol error shows up as a fault, not as a hung GPU
          }
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
#pragma unroll
        for (int k = 0; k < kPF; ++k) {
          const int t = t0 + k;
          // ---------------- sweeps S' - 1 .. 1 (before sweep 0 and each other's updates: they read last step's results)
#pragma unroll
          for (int j = kS - 1; j >= 1; --j) {
            { // (every sweep slot runs in every pass -- one beyond S' works on values nobody reads: a branch here would turn
              // the ring's loads into copies at the branch's end, i.e. into waits)
              const int kq = k % kQF;
              const double Av = qa[j][kq];
              const unsigned cw = used_now(qc[j][kq]);
              { // refill with step t + kQF
                const int tn = t + kQF;
                qa[j][kq] = Ab[(size_t)slot_of(tn - 2 * j + o64[j]) * RS + rowc[j]];
                qc[j][kq] = cw_at(j, tn);
              }
              const int col = t - lane - j;
              const bool act = col >= 0 && col < NS && rowj[j] >= 0;
              const lds_d2 st = (lds_d2)(cw & 0xffffu);
              const d2 ud = st[0], lr = st[1];
              // lane 0's neighbours live in the wavefront above: its lane 63's rows of sweeps j and j - 1
              int c0 = t - j; // lane 0's column (clamped: outside the row, and in the first wavefront, lane 0 idles)
              c0 = c0 < 0 ? 0 : (c0 > NS - 1 ? NS - 1 : c0);
              const double *upj = up + ((size_t)j * W + (wv > 0 ? wv - 1 : 0)) * NSP, *upm = up + ((size_t)(j - 1) * W + (wv > 0 ? wv - 1 : 0)) * NSP;
              const double sU = upj[c0], sR = upm[c0 + 1], sO = upm[c0];
              const double R = dpp_seam<0x138>(r[j - 1], sR);   // lane l - 1's sweep j - 1 result of the last step
              const double O = dpp_seam<0x138>(pr[j - 1], sO);  // ... of the step before: the cell's own old value
              const double U = dpp_seam<0x138>(r[j], sU);       // lane l - 1's sweep j result of the last step
              const double D = r[j - 1], L = r[j];
              double tt = fma(ud.y, D, Av);
              tt = fma(lr.y, R, tt);
              tt = fma(lr.x, L, tt);
              const double res = fma(ud.x, U, tt);
              acc[j] = fmax(acc[j], act ? fabs(res - O) : 0.0); // (selects, no branches: straight-line code keeps the loads in flight)
              pr[j] = r[j];
              r[j] = act ? res : r[j];
              // lane 63 publishes its row for the wavefront below (column col); the others write to a scratch of their own
              double *dst = (lane == 63 && act) ? up + ((size_t)j * W + wv) * NSP + col : pub_dummy;
              *dst = r[j];
            }
          }
          // ---------------- sweep 0: from the input grid
          {
            const double eR = ring_e[k], Av = ring_a[k];
            const unsigned cw = used_now(ring_c[k]);
            { // refill the ring entry with step t + kPF
              const int tn = t + kPF;
              ring_e[k] = Ein[(size_t)slot_of(tn + 1) * RS + row];
              ring_a[k] = Ab[(size_t)slot_of(tn) * RS + row];
              ring_c[k] = cm0[(size_t)min(tn, NW - 1) * 64];
            }
            const lds_d2 st = (lds_d2)(cw & 0xffffu);
            const d2 ud = st[0], lr = st[1];
            const int c0 = t, c63 = t - 63;   // columns of lane 0 / lane 63 at this step
            const double rU = has_prev && c0 < NS ? up[(size_t)(wv > 0 ? wv - 1 : 0) * NSP + c0] : 0.0;
            const double rD = has_next && c63 >= 0 && c63 < NS ? dn_next[c63] : 0.0;
            const double Dn = dpp_seam<0x130>(eR, rD);    // lane l + 1's right-hand value is this lane's lower neighbour
            const double U = dpp_seam<0x138>(r[0], rU);   // lane l - 1's previous result
            double tt = fma(ud.y, Dn, Av);
            tt = fma(lr.y, eR, tt);
            tt = fma(lr.x, r[0], tt);
            const double res = fma(ud.x, U, tt);
            const int col = t - lane;
            const bool act = col >= 0 && col < NS;
            acc[0] = fmax(acc[0], act ? fabs(res - old0) : 0.0);
            pr[0] = r[0];
            r[0] = act ? res : r[0];
            double *dst = (lane == 63 && act) ? up + (size_t)wv * NSP + col : pub_dummy;
            *dst = r[0];
            old0 = eR;
          }
          // ---------------- the pass's last sweep goes to the output grid: row - (S' - 1), column t - lane - (S' - 1).
          // A lane that has not reached its row yet stores what the slot will get later anyway; one that has left it
          // (or idles above the grid) must not touch it.
          {
            const int jl = Sp - 1;
            double v = r[0];
            int rl = row, ol = 0;
#pragma unroll
            for (int j = 1; j < kS; ++j) {
              v = jl == j ? r[j] : v;
              rl = jl == j ? rowj[j] : rl;
              ol = jl == j ? o64[j] : ol;
            }
            const int col = t - lane - jl;
            // (a predicated GLOBAL store: a select between a global and an LDS address would make it a flat store, and the
            // compiler waits for every load in flight around a flat access -- 4,000 cycles per step, measured)
            if (col < NS && rl >= 0) Eout[(size_t)slot_of(t - 2 * jl + ol) * RS + rl] = v;
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) *prog_mine = min(t0 + kPF, NWp); // steps completed
      }
      // ---- max |delta| of every sweep of the pass over the building
#pragma unroll
      for (int j = 0; j < kS; ++j) {
        const double m = wave_max(acc[j]);
        if (lane == 0) mpart[j * 16 + wv] = m;
      }
      __syncthreads();
      double md[kS];
#pragma unroll
      for (int j = 0; j < kS; ++j) {
        md[j] = 0.0;
        for (int w = 0; w < W; ++w) md[j] = fmax(md[j], mpart[j * 16 + w]);
      }
      if (done == 0) md[0] = fmax(md[0], ring_d);
      __syncthreads(); // (mpart is written again by the next pass)
      if (redo) { // the pass was run again with exactly the sweeps that count: its last sweep is the step's last
        done += Sp;
        break;
      }
      int jc = -1; // the first sweep of the pass that ends the step (simulator.py:360-368)
#pragma unroll
      for (int j = kS - 1; j >= 0; --j)
        if (j < Sp && (md[j] <= p.conv_threshold || done + j + 1 >= p.iter_limit)) jc = j;
      if (jc >= 0) converged = md[jc] <= p.conv_threshold;
      if (jc == Sp - 1) { done += Sp; break; }
      if (jc >= 0) { Sp = jc + 1; redo = 1; continue; } // same input, same output grid, fewer sweeps
      // no sweep of the pass ended the step: the output becomes the input; the next pass's length from the decay
      done += Sp;
#pragma unroll
      for (int j = 0; j < kS; ++j)
        if (j < Sp) { md_before = md_last; md_last = md[j]; }
      {
        const double *tmp = Eout;
        Eout = (double *)Ein;
        Ein = tmp;
      }
      int guess = 1;
      if (md_before > md_last && md_last > p.conv_threshold) { // sweeps to go at the last decay rate (the real decay only slows down)
        const double need = log(p.conv_threshold / md_last) / log(md_last / md_before);
        guess = need > (double)kS ? kS : (int)need;
      }
      Sp = min(min(max(guess, 1), kS), p.iter_limit - done);
    }
    // the step's grid is the last pass's output: back into the building's state if it sits in the scratch grid
    __syncthreads();
    if (Eout != T)
      for (int s = 0; s < NS; ++s) T[(size_t)s * RS + row] = Eout[(size_t)s * RS + row];
    // zone sums and the grid sum of the lane's row (row Z of the scratch: cells outside every zone)
    for (int s = 0; s < NS; ++s) {
      const double v = Eout[(size_t)s * RS + row];
      const int z = (int)zmap[(size_t)s * 64];
      __hip_atomic_fetch_add(zs + z * kZC + (lane & (kZC - 2)), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    __syncthreads();
    if (wv == 0) {
      double gacc = 0.0;
      for (int zb = 0; zb <= a.Z; zb += 64) {
        const int zz = zb + lane;
        double v = 0.0;
        if (zz <= a.Z)
          for (int k = 0; k < kZC - 1; ++k) v += zs[zz * kZC + k];
        if (zz < a.Z) a.zsum[(size_t)b * a.Z + zz] = v;
        gacc += v;
      }
      const double gsum = wave_sum(gacc);
      if (lane == 0) {
        a.gsum[b] = gsum + (double)a.n_ring * t_now;
        a.nsw[b] = done | (converged << 16);
        if (a.dbg) {
          atomicAdd((unsigned long long *)a.dbg + 0, 1ull);
          atomicAdd((unsigned long long *)a.dbg + 1, (unsigned long long)n_pass);
          atomicAdd((unsigned long long *)a.dbg + 2, (unsigned long long)n_slots);
          atomicAdd((unsigned long long *)a.dbg + 3, (unsigned long long)done);
        }
      }
    }
    __syncthreads(); // the zone sums are read; the tables may change
  }
}

template <int WMAX>
int go(const Dev &d, double *abuf, double *ebuf, int waves, hipStream_t stream, bool prepare) {
  if (prepare)
    return (int)hipFuncSetAttribute((const void *)k_sweep_stream_ms<WMAX>, hipFuncAttributeMaxDynamicSharedMemorySize, d.lds_reg_bytes);
  hipLaunchKernelGGL((k_sweep_stream_ms<WMAX>), dim3(d.sweep_wgs), dim3(64 * waves), (size_t)d.lds_reg_bytes, stream, d, abuf, ebuf);
  return (int)hipGetLastError();
}
int dispatch(const Dev &d, double *abuf, double *ebuf, int waves, hipStream_t stream, bool prepare) {
  if (waves <= 2) return go<2>(d, abuf, ebuf, waves, stream, prepare);
  if (waves <= 4) return go<4>(d, abuf, ebuf, waves, stream, prepare);
  if (waves <= 8) return go<8>(d, abuf, ebuf, waves, stream, prepare);
  return go<16>(d, abuf, ebuf, waves, stream, prepare);
}

} // namespace

static_assert(kZC == 9, "sweep_stream_zone_columns() of step_stream.hip");
int sweep_stream_ms_sweeps() { return kS; }
// LDS doubles of the seam rows (kS sweeps + the input grid's) and of the exchange area, W wavefronts, NS slots
int sweep_stream_ms_seam_doubles(int NS, int W) { return (kS + 1) * W * (NS + 8); }
int sweep_stream_ms_xchg_doubles(int W) { return 16 + kS * 16 + 64 * W; }

int prepare_sweep_stream_ms(const Dev &d, int waves) { return dispatch(d, nullptr, nullptr, waves, nullptr, true); }
int launch_sweep_stream_ms(const Dev &d, double *abuf, double *ebuf, int waves, hipStream_t stream) {
  return dispatch(d, abuf, ebuf, waves, stream, false);
}

} // namespace sb
