// Micro-benchmark 4 (developer tool): do LDS / VMEM / VALU instructions of ONE wave overlap?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
extern __shared__ __attribute__((aligned(16))) double lds[];
#define LD128(off) asm volatile("ds_read_b128 %0, %1 offset:" #off : "=v"(r128) : "v"(addr));
#define FMA2 asm volatile("v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3" : "+v"(x0), "+v"(x1) : "v"(c0), "v"(c1));
#define FMA4 FMA2 FMA2
#define GLD(off) asm volatile("global_load_dwordx4 %0, %1, off offset:" #off : "=v"(g128) : "v"(gaddr));
#define REP16(M) M(0) M(16) M(32) M(48) M(64) M(80) M(96) M(112) M(128) M(144) M(160) M(176) M(192) M(208) M(224) M(240)
#define MIX1(off) LD128(off) FMA4
#define MIX2(off) LD128(off) GLD(off)
#define MIX3(off) LD128(off) GLD(off) FMA4
#define F4(off) FMA4
template <int MODE>
__global__ void k(long long *out, double *sink, const double *tab, int iters) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 20000; i += blockDim.x) lds[i] = 1.0;
  __syncthreads();
  unsigned addr = ((unsigned)(size_t)(lds + 512 + w * 6400) & 0xffffffffu) + lane * 784;
  const double *gaddr = tab + (lane & 7) * 4;
  double x0 = 1.0 + lane, x1 = 2.0, c0 = 0.999, c1 = 1e-3; double2 r128, g128;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) { REP16(LD128) }
    else if (MODE == 1) { REP16(F4) }
    else if (MODE == 2) { REP16(MIX1) }
    else if (MODE == 3) { REP16(GLD) }
    else if (MODE == 4) { REP16(MIX2) }
    else if (MODE == 5) { REP16(MIX3) }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  }
  long long t1 = __builtin_readcyclecounter();
  if (lane == 0) out[blockIdx.x * (blockDim.x >> 6) + w] = t1 - t0;
  sink[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + r128.x + g128.x;
}
template <int MODE>
void run(const char *name, int threads) {
  long long *d; double *s, *tab; const int blocks = 256, iters = 2000;
  hipMalloc(&d, 8 * 4096); hipMalloc(&s, 8 * blocks * threads); hipMalloc(&tab, 4096); hipMemset(tab, 0, 4096);
  hipFuncSetAttribute((const void *)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 20000 * 8, 0, d, s, tab, iters);
  hipDeviceSynchronize();
  std::vector<long long> h(blocks * (threads / 64));
  hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
  double avg = 0; for (auto v : h) avg += v; avg /= h.size();
  printf("%-52s waves/CU=%d: %.1f cycles per group\n", name, threads / 64, avg / (iters * 16.0));
  hipFree(d); hipFree(s); hipFree(tab);
}
int main() {
  for (int threads : {64, 192}) {
    run<0>("1 ds_read_b128", threads);
    run<1>("4 v_fma_f64 (2 indep chains)", threads);
    run<2>("1 ds_read_b128 + 4 v_fma_f64", threads);
    run<3>("1 global_load_dwordx4 (L1-resident table)", threads);
    run<4>("1 ds_read_b128 + 1 global_load_dwordx4", threads);
    run<5>("1 ds_read_b128 + 1 global_load_dwordx4 + 4 fma", threads);
  }
  return 0;
}
