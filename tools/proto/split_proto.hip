// split_proto.hip -- what would a two-wavefront-per-SIMD column split of the R9 sweep cost per step?
// ("next round" item 1).  Stand-alone throughput model of the sweep's inner step with the
// real instruction mix (class word from L2 -> two ds_read_b128 coefficient gathers -> two 64-bit DPP
// neighbours -> four fp64 FMAs -> |delta| -> running max), fully unrolled over a period of NS steps
// like step_roll.hip, in four configurations:
//   NS = 96, A in LDS,  one wavefront per SIMD   -- today's k_sweep_roll layout (reference point)
//   NS = 48, A in regs, two wavefronts per SIMD  -- the column split: half a row per lane, <= 256 registers,
//            the seam column exchanged through LDS every step (one ds_write_b64 + one ds_read_b64 +
//            two lane-masked selects for the cells next to the seam)
//   NS = 48 without the seam exchange, and NS = 48 with one wavefront per SIMD (what the second
//            resident wavefront buys by itself)
// The values computed are not a Gauss-Seidel sweep (no lag protocol between the two wavefronts: this
// measures issue / LDS-pipe throughput, not correctness).  Output: cycles per wavefront step and per
// 64 cells of SIMD time.  Build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/split_proto tools/proto/split_proto.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef double d2 __attribute__((ext_vector_type(2)));
typedef const d2 __attribute__((address_space(3))) *lds_d2;

template <int CTRL>
__device__ __forceinline__ double dpp_seam(double x, double old) { // lanes without a source keep `old`
  const int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(x), CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(x), CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ __forceinline__ double dpp_zero(double x) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
template <int S, int NS>
__device__ __forceinline__ bool lanes_at_seam() { // lanes l with l == S (mod NS): a 64-bit literal in SGPRs
  constexpr unsigned long long m = (1ull << (S % NS)) | ((S % NS) + NS < 64 ? 1ull << ((S % NS) + NS) : 0ull);
  unsigned lo = (unsigned)m, hi = (unsigned)(m >> 32);
  asm volatile("s_mov_b32 %0, %1" : "=s"(lo) : "n"((unsigned)m));
  if (hi) asm volatile("s_mov_b32 %0, %1" : "=s"(hi) : "n"((unsigned)(m >> 32)));
  return __builtin_amdgcn_inverse_ballot_w64(((unsigned long long)hi << 32) | lo);
}

struct Ctx {
  const char *cmap;      // class words: four 16-bit LDS offsets (coefficient set * 32) per 64-bit word
  unsigned voff;         // running byte offset of the lane's next word
  unsigned long long w, wn, wnn;
  const double *Arow;    // ALDS: the lane's row of A in LDS
  double *xw;            // SEAM: where this wavefront publishes its seam value (lane-strided ring)
  const double *xr;      // SEAM: the partner's ring
};

template <int NS, int S>
__device__ __forceinline__ lds_d2 step_set(Ctx &x) {
  if constexpr (S % 4 == 0) {
    x.w = x.wn;
    x.wn = x.wnn;
    if constexpr (S + 4 >= NS) x.voff -= (unsigned)(NS / 4 - 1) * 512u;
    else x.voff += 512u;
    asm volatile("" : "+v"(x.voff));
    x.wnn = *(const unsigned long long *)(x.cmap + x.voff);
  }
  const unsigned h = S % 4 < 2 ? (unsigned)x.w : (unsigned)(x.w >> 32);
  return (lds_d2)(S % 2 ? h >> 16 : h & 0xffffu);
}

template <int NS, int S, bool ALDS, bool SEAM, int NA>
__device__ __forceinline__ void step(double (&e)[NS], const double (&Areg)[NA], Ctx &x, double &acc, d2 &Apair) {
  constexpr int r = S % NS, rm = (S + NS - 1) % NS, rp = (S + 1) % NS;
  const lds_d2 ct = step_set<NS, S>(x);
  const d2 ud = ct[0], lr = ct[1];
  double A;
  if constexpr (ALDS) {
    if constexpr (S % 2 == 0) Apair = *(const d2 *)(x.Arow + r);
    A = S % 2 == 0 ? Apair.x : Apair.y;
  } else {
    A = Areg[r];
  }
  double Lv = e[rm], Rv = e[rp];
  if constexpr (SEAM) { // the partner's seam value of `lag` steps ago; this wavefront's for the partner
    const double xin = x.xr[(S % 16) * 64];
    x.xw[(S % 16) * 64] = e[rm];
    const bool at = lanes_at_seam<S, NS>();
    Lv = at ? xin : Lv;
    Rv = at ? Rv : Rv; // the right-hand wavefront needs only L; the left-hand one only R: one select each
  }
  const double Dn = dpp_seam<0x130>(Rv, A); // wave_shl:1 (lane 63 keeps `old`)
  double t;
  asm("v_fma_f64 %0, %1, %2, %3" : "=&v"(t) : "v"(ud.y), "v"(Dn), "v"(A));
  t = fma(lr.y, Rv, t);
  const double U = dpp_zero<0x13c>(Lv); // wave_ror:1
  t = fma(lr.x, Lv, t);
  const double nv = fma(ud.x, U, t);
  acc = fmax(acc, fabs(nv - e[r]));
  e[r] = nv;
  asm volatile("" : "+v"(acc));
}

template <int NS, int S, bool ALDS, bool SEAM, int NA>
__device__ __forceinline__ void period(double (&e)[NS], const double (&Areg)[NA], Ctx &x, double &acc, d2 &Apair) {
  if constexpr (S < NS) {
    step<NS, S, ALDS, SEAM>(e, Areg, x, acc, Apair);
    __builtin_amdgcn_sched_barrier(0);
    period<NS, S + 1, ALDS, SEAM>(e, Areg, x, acc, Apair);
  }
}

extern __shared__ __attribute__((aligned(16))) double lds[];

// One wavefront per 64 threads; blockDim.x / 64 wavefronts share the exchange rings.
template <int NS, bool ALDS, bool SEAM, int WPE>
__global__ void __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(WPE, WPE)))
k_steps(const double *init, const double *csets, const unsigned long long *cmap, int periods, double *out, long long *cyc) {
  const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
  double *tab = lds;                       // [32][4] coefficient sets at LDS address 0
  double *ring = lds + 128;                // [2 wavefronts][16 steps][64 lanes]
  double *Al = lds + 128 + 2 * 16 * 64;    // ALDS: [wavefront][64][NS + 2]
  for (int i = threadIdx.x; i < 128; i += blockDim.x) tab[i] = csets[i];
  for (int i = threadIdx.x; i < 2 * 16 * 64; i += blockDim.x) ring[i] = 0.25;
  __syncthreads();
  constexpr int NA = ALDS ? 2 : NS;
  double e[NS], Areg[NA];
#pragma unroll
  for (int j = 0; j < NS; ++j) e[j] = init[j * 64 + lane];
#pragma unroll
  for (int j = 0; j < NA; ++j) Areg[j] = 1e-3 * init[j * 64 + lane];
  Ctx x;
  x.cmap = (const char *)cmap;
  x.voff = (unsigned)lane * 8u;
  x.wn = *(const unsigned long long *)(x.cmap + x.voff);
  x.voff += 512u;
  x.wnn = *(const unsigned long long *)(x.cmap + x.voff);
  x.w = 0;
  x.Arow = Al + ((size_t)wib * 64 + lane) * (NS + 2);
  if (ALDS)
    for (int j = 0; j < NS; ++j) ((double *)x.Arow)[j] = 1e-3 * e[j];
  x.xw = ring + wib * 16 * 64 + lane;
  x.xr = ring + (wib ^ (blockDim.x > 64 ? 1 : 0)) * 16 * 64 + lane;
  __syncthreads();
  double acc = 0.0;
  d2 Apair = {0.0, 0.0};
  const long long t0 = __builtin_readcyclecounter();
#pragma nounroll
  for (int p = 0; p < periods; ++p) {
    __builtin_amdgcn_sched_barrier(0);
    period<NS, 0, ALDS, SEAM>(e, Areg, x, acc, Apair);
  }
  const long long t1 = __builtin_readcyclecounter();
  double s = acc;
#pragma unroll
  for (int j = 0; j < NS; ++j) s += e[j];
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (lane == 0) cyc[blockIdx.x * (blockDim.x >> 6) + wib] = t1 - t0;
}

template <int NS, bool ALDS, bool SEAM, int WPE>
static void run(const char *name, int waves_per_wg, int wgs_per_cu, const double *init, const double *csets,
                const unsigned long long *cmap, double *out, long long *cyc) {
  const int cus = 256, periods = 400;
  // the LDS request sets how many workgroups share a CU
  size_t lds_bytes = (size_t)(160 * 1024 / wgs_per_cu) - 1024;
  const size_t need = (128 + 2 * 16 * 64 + (ALDS ? (size_t)waves_per_wg * 64 * (NS + 2) : 0)) * 8;
  if (need > lds_bytes) { printf("%-34s skipped: needs %zu B of LDS\n", name, need); return; }
  auto kern = k_steps<NS, ALDS, SEAM, WPE>;
  hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
  const int wgs = cus * wgs_per_cu;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(wgs), dim3(64 * waves_per_wg), lds_bytes, 0, init, csets, cmap, periods, out, cyc);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  const int nw = wgs * waves_per_wg;
  std::vector<long long> c(nw);
  hipMemcpy(c.data(), cyc, nw * sizeof(long long), hipMemcpyDeviceToHost);
  double mean = 0;
  for (long long v : c) mean += (double)v;
  mean /= nw;
  const double per_step = mean / ((double)periods * NS);
  const int waves_per_simd = waves_per_wg * wgs_per_cu / 4;
  printf("%-34s %d wavefront(s)/SIMD  %7.1f cycles per wavefront step  %7.1f SIMD cycles per 64 cells  (%.3f ms, %s)\n", name,
         waves_per_simd, per_step, per_step / waves_per_simd, best, hipGetErrorString(hipGetLastError()));
}

int main() {
  double *init, *csets, *out;
  unsigned long long *cmap;
  long long *cyc;
  hipMalloc(&init, 96 * 64 * 8);
  hipMalloc(&csets, 128 * 8);
  hipMalloc(&cmap, 64 * 64 * 8);
  hipMalloc(&out, (size_t)256 * 8 * 128 * 8);
  hipMalloc(&cyc, (size_t)256 * 8 * 2 * 8);
  std::vector<double> h(96 * 64);
  for (size_t i = 0; i < h.size(); ++i) h[i] = 290.0 + (double)(i % 977) * 0.01;
  hipMemcpy(init, h.data(), h.size() * 8, hipMemcpyHostToDevice);
  std::vector<double> cs(128, 0.0);
  for (int s = 0; s < 9; ++s)
    for (int k = 0; k < 4; ++k) cs[4 * s + k] = 0.2499 - 0.001 * s;
  hipMemcpy(csets, cs.data(), cs.size() * 8, hipMemcpyHostToDevice);
  std::vector<unsigned long long> cm(64 * 64);
  for (size_t i = 0; i < cm.size(); ++i) { // mostly set 0 (air), now and then a wall set: R9's mix
    unsigned long long w = 0;
    for (int k = 0; k < 4; ++k) w |= (unsigned long long)((((i * 7 + k * 3) % 11 == 0) ? (1 + (i + k) % 8) : 0) * 32) << (16 * k);
    cm[i] = w;
  }
  hipMemcpy(cmap, cm.data(), cm.size() * 8, hipMemcpyHostToDevice);
  printf("# tools/proto/split_proto.hip -- cycles per sweep step, every CU loaded the same way (256 CUs)\n");
  run<96, true, false, 1>("NS=96 A in LDS (today)", 1, 4, init, csets, cmap, out, cyc);
  run<96, false, false, 1>("NS=96 A in registers", 1, 4, init, csets, cmap, out, cyc);
  run<48, false, false, 1>("NS=48 no seam", 1, 4, init, csets, cmap, out, cyc);
  run<48, false, false, 2>("NS=48 no seam", 1, 8, init, csets, cmap, out, cyc);
  run<48, false, true, 1>("NS=48 seam through LDS", 2, 2, init, csets, cmap, out, cyc);
  run<48, false, true, 2>("NS=48 seam through LDS", 2, 4, init, csets, cmap, out, cyc);
  return 0;
}
