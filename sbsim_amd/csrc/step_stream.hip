// step_stream.hip -- the sweep kernel for floor plans that do not fit one CU (more than 130 rows or 96
// columns inside the exterior ring, or a grid beyond the LDS of a CU): the temperature grid STAYS in
// global memory (HBM / L2) and streams through the wavefronts once per Gauss-Seidel sweep.
// simulator.py:278-371; any H x W is legal in building.py:609-764.
//
// One workgroup of W = ceil(rows / 64) wavefronts (W <= 16: 1,024 rows) per building; lane l of
// wavefront w owns row 64 w + l and walks it in the anti-diagonal order of the register kernels: at
// local step t it updates column t - l, wavefront w running >= 64 steps behind wavefront w - 1.  That
// is the reference's row-major in-place order.  The state is stored
// slot-major -- column c of row r at [slot (c + r) mod NS][r], the layout of step_reg.hip's HBM state
// -- so a wavefront's access at step t is ONE coalesced 512-byte row of slot t mod NS.  Per cell and
// sweep a lane loads its right-hand neighbour (8 B; it is the cell's own old value one step later),
// A = ap*Tprev + g (8 B, written by a pass before the first sweep) and a class word (4 B, static),
// and stores the new value (8 B); the left-hand neighbour is the lane's previous result, the upper /
// lower neighbours come by DPP from the neighbouring lanes.  Everything a lane reads from global
// memory was written by its own wavefront; the rows at a seam between two wavefronts (row 64 w - 1's
// new values, row 64 w's old ones) travel through LDS under a progress counter per wavefront.
// Bound: HBM / L2 bandwidth (28 B per cell and sweep) -- in practice memory LATENCY times occupancy: a wavefront has kPF
// steps of loads in flight, so what counts is wavefronts per CU (round 5: 128 registers per lane -- 132-144 B of them spilled
// -- for four wavefronts per SIMD and a zone-sum scratch of 8 instead of 16 columns so that three workgroups of a
// 299 x 401 plan share a CU instead of two: 1.0e11 -> 1.35e11 cell-sweeps/s at 3,072 buildings).
#include "sb_device.h"

namespace sb {
namespace {

constexpr int kSets = 32;   // entries of the coefficient-set table (at LDS address 0)
// (developer knobs of tools/build_variant.sh: prefetch depth, zone-sum columns, wavefronts per SIMD asked of the compiler)
#ifndef SB_STREAM_PF
#define SB_STREAM_PF 8
#endif
#ifndef SB_STREAM_ZC
#define SB_STREAM_ZC 9
#endif
#ifndef SB_STREAM_WPE
#define SB_STREAM_WPE 4
#endif
constexpr int kSpinMax = 1 << 24; // a wait that long is a protocol error: trap instead of hanging the GPU (overlapped sweeps)
constexpr int kPF = SB_STREAM_PF; // steps between a global load and its use
constexpr int kZC = SB_STREAM_ZC; // columns of the zone-sum scratch per zone (16 lane columns + 1: odd stride)

typedef double d2 __attribute__((ext_vector_type(2)));
typedef const d2 __attribute__((address_space(3))) *lds_d2;
typedef volatile int __attribute__((address_space(3))) *lds_vi;

template <int CTRL>
__device__ __forceinline__ double dpp_seam(double x, double old) { // lanes without a source keep `old`
  const int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(x), CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(x), CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}

extern __shared__ __attribute__((aligned(16))) double lds[];

// LDS (doubles): [tabc 4 kSets][tapg 2 ts] | r_seam: up [W][NS + 8], dn [W][NS + 8] | r_xchg: progress
// [16] ints, max|delta| parts [16] | r_A: zone sums [Z + 1][kZC]
template <int WMAX> // wavefronts per workgroup <= WMAX: the register budget follows the launch bound
__global__ void __launch_bounds__(64 * WMAX)
__attribute__((amdgpu_waves_per_eu(WMAX == 2 ? 3 : SB_STREAM_WPE)))
k_sweep_stream(Dev a, double *Abuf) {
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int W = (int)(blockDim.x >> 6), NS = a.NR, RS = a.RS, NSP = NS + 8;
  double *tabc = lds;
  double *tapg = lds + 4 * kSets;
  double *up = lds + a.r_seam;              // up[w][c]: row 64 w + 63's new value at column c (this sweep)
  double *dn = up + (size_t)W * NSP;        // dn[w][c]: row 64 w's latest value at column c
  int *prog = (int *)(lds + a.r_xchg);      // [W] steps completed in this sweep
  double *mpart = lds + a.r_xchg + 8;       // [W] max |delta| of the wavefront's rows
  int *misc = (int *)(lds + a.r_xchg + 24); // [0]: the next building
  double *dummy = lds + a.r_xchg + 32;      // [W][64]: where the lanes that publish nothing write
  double *zs = lds + a.r_A;                 // [Z + 1][kZC]
  for (int i = threadIdx.x; i < a.lds_reg_bytes / 8; i += blockDim.x) lds[i] = 0.0;
  __syncthreads();
  for (int i = threadIdx.x; i < 4 * kSets; i += blockDim.x) tabc[i] = i < 4 * a.ncset ? a.csetab[i] : 0.0;
  for (int c = threadIdx.x; c < a.ts; c += blockDim.x) tapg[2 * c] = c <= a.ncls ? a.ctab[c * 8 + 4] : 0.0; // row `ncls`: the pad class
  __syncthreads();
  if ((unsigned)(size_t)(__attribute__((address_space(3))) double *)tabc != 0u) __builtin_trap(); // class words hold LDS addresses

  const sb_params &p = a.p;
  const int row = wv * 64 + lane;
  const int NW = NS + 63;                                          // steps of a wavefront per sweep
  // global addresses as a UNIFORM base (scalar registers; the slot / step part is scalar arithmetic) + a 32-bit per-lane
  // byte offset: the loads and stores take the `saddr + voffset` form, no 64-bit address pair per stream in vector registers
  const unsigned *cmap_u = (const unsigned *)a.cmapS + (size_t)wv * NW * 64; // [W][NW][64]: set offset | class * 16 << 16
  const unsigned short *zmap_u = (const unsigned short *)a.zmapS + (size_t)wv * NS * 64; // [W][NS][64]: zone (Z: none)
  const unsigned lane4 = (unsigned)lane * 4u, lane2 = (unsigned)lane * 2u, row8 = (unsigned)row * 8u;
  auto cmap_at = [&](int step) { return *(const unsigned *)((const char *)(cmap_u + (size_t)step * 64) + lane4); };
  lds_vi prog_mine = (lds_vi)(unsigned)(size_t)(__attribute__((address_space(3))) int *)(prog + wv);
  lds_vi prog_prev = (lds_vi)(unsigned)(size_t)(__attribute__((address_space(3))) int *)(prog + (wv > 0 ? wv - 1 : 0));
  const double *up_prev = up + (size_t)(wv > 0 ? wv - 1 : 0) * NSP;  // lane 0's upper neighbours
  const double *dn_next = dn + (size_t)(wv + 1 < W ? wv + 1 : wv) * NSP; // lane 63's lower neighbours
  const bool has_prev = wv > 0, has_next = wv + 1 < W;
  double *up_mine = up + (size_t)wv * NSP, *dn_mine = dn + (size_t)wv * NSP;
  // every step a lane writes its result to LDS: lane 63 to up[w][column], lane 0 to dn[w][column], the
  // others (and a lane outside its row) to a scratch double of their own -- one ds_write, no branch
  double *const pub_dummy = dummy + (size_t)wv * 64 + lane;
  double *const pub_seam = lane == 63 ? up_mine - 63 : (lane == 0 ? dn_mine : nullptr); // + step = + column + lane

  for (int b = blockIdx.x, bn = 0; b < a.B; b = bn) {
    if (threadIdx.x == 0) misc[0] = a.sweep_wgs + atomicAdd(a.next_b, 1);
    double *Eu = a.temp + (size_t)b * a.state_doubles;          // [NS][RS]; this lane: + row
    double *Au = Abuf + (size_t)blockIdx.x * a.state_doubles;   // one A grid per resident workgroup
    auto E_at = [&](int slot) -> double & { return *(double *)((char *)(Eu + (size_t)slot * RS) + row8); };
    auto A_at = [&](int slot) -> double & { return *(double *)((char *)(Au + (size_t)slot * RS) + row8); };
    const double t_now = a.bld[b].t_now;
    const double ring_lo = a.scal[(size_t)b * kNScal + 16], ring_hi = a.scal[(size_t)b * kNScal + 17];
    // exterior-space cells outside the trim box all become t_now in the first sweep
    // (simulator.py:256-258); their largest |delta| follows from their extreme values
    const double ring_d = a.n_ring > 0 ? fmax(fabs(t_now - ring_lo), fabs(t_now - ring_hi)) : 0.0;
    for (int c = threadIdx.x; c < a.ts; c += blockDim.x) tapg[2 * c + 1] = a.gtabg[(size_t)b * a.ts + c];
    for (int i = threadIdx.x; i < (a.Z + 1) * kZC; i += blockDim.x) zs[i] = 0.0;
    __syncthreads();
    bn = __builtin_amdgcn_readfirstlane(*(volatile int *)misc);
    // A = ap*Tprev + g for every cell of the lane's row, slot by slot (the class of slot s: the word of
    // the step at which the lane works on it, s - lane mod NS + lane); row 64 w's values into dn[w]
    for (int s = 0; s < NS; ++s) {
      int t = s - lane;           // the lane's column at slot s
      if (t < 0) t += NS;
      const unsigned cw = cmap_u[(size_t)(t + lane) * 64 + lane]; // step t + lane: column t
      const d2 pg = *(const d2 *)((const char *)tapg + (cw >> 16));
      const double v = E_at(s);
      A_at(s) = fma(pg.x, v, pg.y);
      if (lane == 0) dn_mine[s] = v; // lane 0: column s sits in slot s
    }
    int n_sweeps = 0, converged = 0;
    for (;;) { // simulator.py:348-368
      if (lane == 0) *prog_mine = 0;
      __syncthreads();
      double acc = 0.0;
      // streams: eR(t) = E[(t + 1) mod NS] (the right-hand neighbour at step t = the cell's own old value
      // at step t + 1), A(t) = Ab[t mod NS], cw(t); kPF steps ahead, in a ring of registers
      double ring_e[kPF], ring_a[kPF];
      unsigned ring_c[kPF];
      int sl = 0; // slot of step t + kPF ... maintained incrementally: sl = (t + kPF) mod NS
#pragma unroll
      for (int k = 0; k < kPF; ++k) { // steps 0 .. kPF-1
        const int s0 = k % NS, s1 = (k + 1) % NS;
        ring_e[k] = E_at(s1);
        ring_a[k] = A_at(s0);
        ring_c[k] = cmap_at(k);
      }
      sl = kPF % NS;
      double old = E_at(0);   // the lane's own old value at step 0 (slot 0)
      double nv = 0.0;        // the lane's previous result (left-hand neighbour)
      for (int t0 = 0; t0 < NW; t0 += kPF) {
        // the wavefront above must have published row 63's new values for the columns lane 0 reaches here
        if (has_prev) {
          const int need = min(t0 + kPF, NS) + 63; // column c is published at step c + 63: c + 64 steps completed
          while (__builtin_amdgcn_readfirstlane(*prog_prev) < need) __builtin_amdgcn_s_sleep(1);
          // acquire: the seam reads below must not be hoisted above the spin (the hardware keeps a wavefront's LDS
          // operations in order; this fence is for the compiler -- the seam values are plain doubles, the
          // progress word a volatile int, and type-based alias analysis would let one pass the other)
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
#pragma unroll
        for (int k = 0; k < kPF; ++k) {
          const int t = t0 + k;
          const int s = t < NS ? t : t - NS; // t mod NS (t < NS + 63 <= 2 NS)
          const double eR = ring_e[k], Av = ring_a[k];
          const unsigned cw = ring_c[k];
          { // refill the ring entry with step t + kPF
            const int tn = t + kPF;
            int s1 = sl + 1;
            if (s1 >= NS) s1 -= NS;
            ring_e[k] = E_at(s1);
            ring_a[k] = A_at(sl);
            ring_c[k] = cmap_at(min(tn, NW - 1));
            sl = s1;
          }
          const lds_d2 st = (lds_d2)(cw & 0xffffu);
          const d2 ud = st[0], lr = st[1];
          const int c0 = t, c63 = t - 63;   // columns of lane 0 / lane 63 at this step
          const double rU = has_prev && c0 < NS ? up_prev[c0] : 0.0;
          const double rD = has_next && c63 >= 0 ? dn_next[c63] : 0.0;
          const double Dn = dpp_seam<0x130>(eR, rD);    // lane l + 1's right-hand value is this lane's lower neighbour
          const double U = dpp_seam<0x138>(nv, rU);     // lane l - 1's previous result
          double tt = fma(ud.y, Dn, Av);
          tt = fma(lr.y, eR, tt);
          tt = fma(lr.x, nv, tt);
          const double res = fma(ud.x, U, tt);
          const int col = t - lane;
          const bool act = col >= 0 && col < NS;
          // a lane outside its row stores the slot's unchanged content back (`old` is that content)
          const double out = act ? res : old;
          acc = fmax(acc, fabs(out - old));
          E_at(s) = out;
          *((act && pub_seam) ? pub_seam + t : pub_dummy) = out;
          nv = act ? res : nv;
          old = eR;
        }
        // release: every seam value of these steps is written before the progress word says so (for the compiler;
        // the hardware keeps a wavefront's LDS operations in order)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) *prog_mine = min(t0 + kPF, NW); // steps completed
      }
      // max |delta| over the building
      double m = wave_max(acc);
      if (lane == 0) mpart[wv] = m;
      __syncthreads();
      double md = 0.0;
      for (int w = 0; w < W; ++w) md = fmax(md, mpart[w]);
      if (n_sweeps == 0) md = fmax(md, ring_d);
      ++n_sweeps;
      converged = md <= p.conv_threshold;
      if (converged || n_sweeps >= p.iter_limit) break;
    }
    // zone sums and the grid sum of the lane's row (row Z of the scratch: cells outside every zone)
    __syncthreads();
    for (int s = 0; s < NS; ++s) {
      const double v = E_at(s);
      const int z = (int)*(const unsigned short *)((const char *)(zmap_u + (size_t)s * 64) + lane2);
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
        a.nsw[b] = n_sweeps | (converged << 16);
      }
    }
    __syncthreads(); // the zone sums are read; the tables may change
  }
}

// ---------------------------------------------------------------- overlapped sweeps (round 5)
// EXPERIMENTAL (exact, tested, slower on the plan it was written for: LABNOTES.md 5.4b): compiled with -DSB_EXPERIMENTAL only
// (SBSIM_BUILD_EXPERIMENTAL=1, sbsim_amd/build.py); the default library carries stubs.
#ifdef SB_EXPERIMENTAL
// k_sweep_stream's wavefront w starts a sweep 64 steps after wavefront w - 1 and the workgroup meets at a barrier after every
// sweep: of the NW + 64 (W - 1) steps a sweep takes, a wavefront works NW (299 x 401: 464 of 720) -- and an idle wavefront has
// no loads in flight, which is what this kernel's speed is made of.  Here wavefront w goes on into sweep j + 1 as soon as it
// has finished its rows of sweep j.  Two grids in global memory take turns (sweep j reads G[j & 1] and writes G[(j + 1) & 1];
// G[0] is the building's state, G[1] a scratch grid per resident workgroup), so a sweep started in vain -- sweep j was the
// last one, which is known only when the LAST wavefront has finished it -- overwrites sweep j's INPUT, not its result:
// nothing to undo (simulator.py:348-368: the iterates and the sweep count are those of the plain schedule).  A wavefront
// reads from global memory only what it wrote itself; what crosses a seam travels through LDS as before, now under two
// conditions: wavefront w runs >= 64 steps behind wavefront w - 1 (row 64 w - 1's new values of THIS sweep) and at most
// about a sweep ahead of wavefront w + 1 (row 64 (w + 1)'s new values of the PREVIOUS sweep, and wavefront w + 1 must have
// read up[w][c] of the previous sweep before it is overwritten).  The last wavefront decides: after its part of sweep j it
// combines the wavefronts' max |delta| of sweep j (they are all ahead) and publishes the verdict; a wavefront starts sweep
// j + 2 only when sweep j's verdict is out, and leaves a sweep at the next chunk boundary once a verdict says stop.
// LDS: as above + r_xchg: [8 + 16 + W .. ): misc ints [2] sweeps decided, [3] sweeps of the step (0: undecided), [4] converged;
// the max |delta| parts of odd sweeps behind the dummy area.
template <int WMAX>
__global__ void __launch_bounds__(64 * WMAX)
__attribute__((amdgpu_waves_per_eu(WMAX == 2 ? 3 : SB_STREAM_WPE)))
k_sweep_stream_roll(Dev a, double *Abuf, double *Ebuf) {
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int W = (int)(blockDim.x >> 6), NS = a.NR, RS = a.RS, NSP = NS + 8;
  double *tabc = lds;
  double *tapg = lds + 4 * kSets;
  double *up = lds + a.r_seam;
  double *dn = up + (size_t)W * NSP;
  int *prog = (int *)(lds + a.r_xchg);      // [W] steps completed since the building's first sweep began: sweep * NW + steps
  double *mpart0 = lds + a.r_xchg + 8;      // [W] max |delta| of the wavefront's rows, even sweeps
  int *misc = (int *)(lds + a.r_xchg + 24); // [0]: the next building, [2..4]: the verdicts (above)
  double *dummy = lds + a.r_xchg + 32;
  double *mpart1 = dummy + (size_t)W * 64;  // ... odd sweeps
  double *zs = lds + a.r_A;
  for (int i = threadIdx.x; i < a.lds_reg_bytes / 8; i += blockDim.x) lds[i] = 0.0;
  __syncthreads();
  for (int i = threadIdx.x; i < 4 * kSets; i += blockDim.x) tabc[i] = i < 4 * a.ncset ? a.csetab[i] : 0.0;
  for (int c = threadIdx.x; c < a.ts; c += blockDim.x) tapg[2 * c] = c <= a.ncls ? a.ctab[c * 8 + 4] : 0.0;
  __syncthreads();
  if ((unsigned)(size_t)(__attribute__((address_space(3))) double *)tabc != 0u) __builtin_trap();

  const sb_params &p = a.p;
  const int row = wv * 64 + lane;
  const int NW = NS + 63;
  const unsigned *cmap_u = (const unsigned *)a.cmapS + (size_t)wv * NW * 64;
  const unsigned short *zmap_u = (const unsigned short *)a.zmapS + (size_t)wv * NS * 64;
  const unsigned lane4 = (unsigned)lane * 4u, lane2 = (unsigned)lane * 2u, row8 = (unsigned)row * 8u;
  auto cmap_at = [&](int step) { return *(const unsigned *)((const char *)(cmap_u + (size_t)step * 64) + lane4); };
  lds_vi prog_mine = (lds_vi)(unsigned)(size_t)(__attribute__((address_space(3))) int *)(prog + wv);
  lds_vi prog_prev = (lds_vi)(unsigned)(size_t)(__attribute__((address_space(3))) int *)(prog + (wv > 0 ? wv - 1 : 0));
  lds_vi prog_next = (lds_vi)(unsigned)(size_t)(__attribute__((address_space(3))) int *)(prog + (wv + 1 < W ? wv + 1 : wv));
  lds_vi v_decided = (lds_vi)(unsigned)(size_t)(__attribute__((address_space(3))) int *)(misc + 2);
  lds_vi v_total = (lds_vi)(unsigned)(size_t)(__attribute__((address_space(3))) int *)(misc + 3);
  const double *up_prev = up + (size_t)(wv > 0 ? wv - 1 : 0) * NSP;
  const double *dn_next = dn + (size_t)(wv + 1 < W ? wv + 1 : wv) * NSP;
  const bool has_prev = wv > 0, has_next = wv + 1 < W;
  double *up_mine = up + (size_t)wv * NSP, *dn_mine = dn + (size_t)wv * NSP;
  double *const pub_dummy = dummy + (size_t)wv * 64 + lane;
  double *const pub_seam = lane == 63 ? up_mine - 63 : (lane == 0 ? dn_mine : nullptr);

  for (int b = blockIdx.x, bn = 0; b < a.B; b = bn) {
    if (threadIdx.x == 0) { misc[0] = a.sweep_wgs + atomicAdd(a.next_b, 1); misc[2] = 0; misc[3] = 0; misc[4] = 0; }
    if (lane == 0) *prog_mine = 0;
    double *G0 = a.temp + (size_t)b * a.state_doubles;          // the building's state
    double *G1 = Ebuf + (size_t)blockIdx.x * a.state_doubles;   // the other grid of this workgroup
    double *Au = Abuf + (size_t)blockIdx.x * a.state_doubles;
    auto A_at = [&](int slot) -> double & { return *(double *)((char *)(Au + (size_t)slot * RS) + row8); };
    const double t_now = a.bld[b].t_now;
    const double ring_lo = a.scal[(size_t)b * kNScal + 16], ring_hi = a.scal[(size_t)b * kNScal + 17];
    const double ring_d = a.n_ring > 0 ? fmax(fabs(t_now - ring_lo), fabs(t_now - ring_hi)) : 0.0;
    for (int c = threadIdx.x; c < a.ts; c += blockDim.x) tapg[2 * c + 1] = a.gtabg[(size_t)b * a.ts + c];
    for (int i = threadIdx.x; i < (a.Z + 1) * kZC; i += blockDim.x) zs[i] = 0.0;
    __syncthreads();
    bn = __builtin_amdgcn_readfirstlane(*(volatile int *)misc);
    for (int s = 0; s < NS; ++s) { // A = ap*Tprev + g; row 64 w's values into dn[w]
      int t = s - lane;
      if (t < 0) t += NS;
      const unsigned cw = cmap_u[(size_t)(t + lane) * 64 + lane];
      const d2 pg = *(const d2 *)((const char *)tapg + (cw >> 16));
      const double v = *(const double *)((const char *)(G0 + (size_t)s * RS) + row8);
      A_at(s) = fma(pg.x, v, pg.y);
      if (lane == 0) dn_mine[s] = v;
    }
    __syncthreads(); // every dn row is in place before a neighbour's first sweep reads it
    bool stop = false;
    for (int j = 0; !stop; ++j) { // the sweeps of this wavefront; `stop`: a verdict ended the step
      if (j >= 2) { // sweep j - 2's verdict
        for (int spin = 0; __builtin_amdgcn_readfirstlane(*v_decided) < j - 1 && __builtin_amdgcn_readfirstlane(*v_total) == 0; ++spin) { if (spin > kSpinMax) __builtin_trap(); __builtin_amdgcn_s_sleep(1); }
      }
      if (__builtin_amdgcn_readfirstlane(*v_total) != 0) break;
      double *Gi = (j & 1) ? G1 : G0, *Go = (j & 1) ? G0 : G1;
      auto I_at = [&](int slot) -> const double & { return *(const double *)((const char *)(Gi + (size_t)slot * RS) + row8); };
      auto O_at = [&](int slot) -> double & { return *(double *)((char *)(Go + (size_t)slot * RS) + row8); };
      const int base = j * NW;
      double acc = 0.0;
      double ring_e[kPF], ring_a[kPF];
      unsigned ring_c[kPF];
      int sl = 0;
#pragma unroll
      for (int k = 0; k < kPF; ++k) {
        const int s0 = k % NS, s1 = (k + 1) % NS;
        ring_e[k] = I_at(s1);
        ring_a[k] = A_at(s0);
        ring_c[k] = cmap_at(k);
      }
      sl = kPF % NS;
      double old = I_at(0);
      double nv = 0.0;
      for (int t0 = 0; t0 < NW; t0 += kPF) {
        if (has_prev) { // row 64 w - 1's new values of this sweep for the columns lane 0 reaches here
          const int need = base + min(t0 + kPF, NS) + 63;
          for (int spin = 0; __builtin_amdgcn_readfirstlane(*prog_prev) < need && __builtin_amdgcn_readfirstlane(*v_total) == 0; ++spin) { if (spin > kSpinMax) __builtin_trap(); __builtin_amdgcn_s_sleep(1); }
        }
        if (has_next && j > 0) { // row 64 (w + 1)'s new values of the previous sweep for the columns lane 63 reaches here;
          // the same progress says that wavefront w + 1 has read the up[w][c] lane 63 is about to overwrite
          const int need = base - NW + min(max(t0 + kPF - 63, 0), NS);
          for (int spin = 0; __builtin_amdgcn_readfirstlane(*prog_next) < need && __builtin_amdgcn_readfirstlane(*v_total) == 0; ++spin) { if (spin > kSpinMax) __builtin_trap(); __builtin_amdgcn_s_sleep(1); }
        }
        if (__builtin_amdgcn_readfirstlane(*v_total) != 0) { stop = true; break; } // (never the last wavefront: it speaks the verdicts)
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#pragma unroll
        for (int k = 0; k < kPF; ++k) {
          const int t = t0 + k;
          const int s = t < NS ? t : t - NS;
          const double eR = ring_e[k], Av = ring_a[k];
          const unsigned cw = ring_c[k];
          {
            const int tn = t + kPF;
            int s1 = sl + 1;
            if (s1 >= NS) s1 -= NS;
            ring_e[k] = I_at(s1);
            ring_a[k] = A_at(sl);
            ring_c[k] = cmap_at(min(tn, NW - 1));
            sl = s1;
          }
          const lds_d2 st = (lds_d2)(cw & 0xffffu);
          const d2 ud = st[0], lr = st[1];
          const int c0 = t, c63 = t - 63;
          const double rU = has_prev && c0 < NS ? up_prev[c0] : 0.0;
          const double rD = has_next && c63 >= 0 ? dn_next[c63] : 0.0;
          const double Dn = dpp_seam<0x130>(eR, rD);
          const double U = dpp_seam<0x138>(nv, rU);
          double tt = fma(ud.y, Dn, Av);
          tt = fma(lr.y, eR, tt);
          tt = fma(lr.x, nv, tt);
          const double res = fma(ud.x, U, tt);
          const int col = t - lane;
          const bool act = col >= 0 && col < NS;
          const double out = act ? res : old;
          acc = fmax(acc, fabs(out - old));
          if (act) O_at(s) = out; // (a lane outside its row must NOT store: `old` is the INPUT grid's content, and a slot visited
          // again after its column was updated would get the previous sweep's value back)
          *((act && pub_seam) ? pub_seam + t : pub_dummy) = out;
          nv = act ? res : nv;
          old = eR;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (t0 + kPF < NW && lane == 0) *prog_mine = base + t0 + kPF;
      }
      if (stop) break;
      // the wavefront's max |delta| of sweep j, THEN its progress (the last wavefront reads the parts when its own
      // last wait has seen the wavefront above finish; by induction every wavefront above has)
      const double m = wave_max(acc);
      double *mp = (j & 1) ? mpart1 : mpart0;
      if (lane == 0) mp[wv] = m;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      if (lane == 0) *prog_mine = base + NW;
      if (!has_next) { // the verdict on sweep j (simulator.py:360-368)
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        double md = 0.0;
        for (int w = 0; w < W; ++w) md = fmax(md, ((volatile double *)mp)[w]);
        if (j == 0) md = fmax(md, ring_d);
        const int conv = md <= p.conv_threshold;
        if (conv || j + 1 >= p.iter_limit) {
          if (lane == 0) { misc[4] = conv; __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); *v_total = j + 1; }
          stop = true;
        } else if (lane == 0) *v_decided = j + 1;
      }
    }
    __syncthreads();
    const int n_sweeps = misc[3], converged = misc[4];
    const double *F = (n_sweeps & 1) ? G1 : G0; // the last sweep's output
    for (int s = 0; s < NS; ++s) { // zone sums and the grid sum of the lane's row; the state goes home when it ended in the other grid
      const double v = *(const double *)((const char *)(F + (size_t)s * RS) + row8);
      if (n_sweeps & 1) *(double *)((char *)(G0 + (size_t)s * RS) + row8) = v;
      const int z = (int)*(const unsigned short *)((const char *)(zmap_u + (size_t)s * 64) + lane2);
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
        a.nsw[b] = n_sweeps | (converged << 16);
      }
    }
    __syncthreads();
  }
}

template <int WMAX>
int go_roll(const Dev &d, double *abuf, double *ebuf, int waves, hipStream_t stream, bool prepare) {
  if (prepare)
    return (int)hipFuncSetAttribute((const void *)k_sweep_stream_roll<WMAX>, hipFuncAttributeMaxDynamicSharedMemorySize, d.lds_reg_bytes);
  hipLaunchKernelGGL((k_sweep_stream_roll<WMAX>), dim3(d.sweep_wgs), dim3(64 * waves), (size_t)d.lds_reg_bytes, stream, d, abuf, ebuf);
  return (int)hipGetLastError();
}
int dispatch_roll(const Dev &d, double *abuf, double *ebuf, int waves, hipStream_t stream, bool prepare) {
  if (waves <= 2) return go_roll<2>(d, abuf, ebuf, waves, stream, prepare);
  if (waves <= 4) return go_roll<4>(d, abuf, ebuf, waves, stream, prepare);
  if (waves <= 8) return go_roll<8>(d, abuf, ebuf, waves, stream, prepare);
  return go_roll<16>(d, abuf, ebuf, waves, stream, prepare);
}

#endif // SB_EXPERIMENTAL

template <int WMAX>
int go(const Dev &d, double *abuf, int waves, hipStream_t stream, bool prepare) {
  if (prepare)
    return (int)hipFuncSetAttribute((const void *)k_sweep_stream<WMAX>, hipFuncAttributeMaxDynamicSharedMemorySize, d.lds_reg_bytes);
  hipLaunchKernelGGL((k_sweep_stream<WMAX>), dim3(d.sweep_wgs), dim3(64 * waves), (size_t)d.lds_reg_bytes, stream, d, abuf);
  return (int)hipGetLastError();
}
int dispatch(const Dev &d, double *abuf, int waves, hipStream_t stream, bool prepare) {
  if (waves <= 2) return go<2>(d, abuf, waves, stream, prepare);
  if (waves <= 4) return go<4>(d, abuf, waves, stream, prepare);
  if (waves <= 8) return go<8>(d, abuf, waves, stream, prepare);
  return go<16>(d, abuf, waves, stream, prepare);
}

} // namespace

int sweep_stream_set_table() { return kSets; }
int sweep_stream_zone_columns() { return kZC; }

#ifdef SB_EXPERIMENTAL
int sweep_stream_roll_xchg_extra_doubles() { return 16; } // the odd sweeps' max |delta| parts behind the publish scratch
int prepare_sweep_stream_roll(const Dev &d, int waves) { return dispatch_roll(d, nullptr, nullptr, waves, nullptr, true); }
int launch_sweep_stream_roll(const Dev &d, double *abuf, double *ebuf, int waves, hipStream_t stream) { return dispatch_roll(d, abuf, ebuf, waves, stream, false); }
#else // the planner never asks for it (sbsim_hip.hip: kExperimental)
int sweep_stream_roll_xchg_extra_doubles() { return 0; }
int prepare_sweep_stream_roll(const Dev &, int) { return (int)hipErrorNotSupported; }
int launch_sweep_stream_roll(const Dev &, double *, double *, int, hipStream_t) { return (int)hipErrorNotSupported; }
#endif
int prepare_sweep_stream(const Dev &d, int waves) { return dispatch(d, nullptr, waves, nullptr, true); }
int launch_sweep_stream(const Dev &d, double *abuf, int waves, hipStream_t stream) { return dispatch(d, abuf, waves, stream, false); }

} // namespace sb
