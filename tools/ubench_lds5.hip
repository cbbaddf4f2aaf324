// Micro-benchmark 5 (developer tool): cost of an LDS read issued with few active lanes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
extern __shared__ __attribute__((aligned(16))) double lds[];
#define LD128(off) asm volatile("ds_read_b128 %0, %1 offset:" #off : "=v"(r128) : "v"(addr));
#define REP16(M) M(0) M(16) M(32) M(48) M(64) M(80) M(96) M(112) M(128) M(144) M(160) M(176) M(192) M(208) M(224) M(240)
template <int MODE>
__global__ void k(long long *out, double *sink, int iters) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 20000; i += blockDim.x) lds[i] = 1.0;
  __syncthreads();
  unsigned addr = ((unsigned)(size_t)(lds + 512 + w * 6400) & 0xffffffffu) + (lane & 7) * 32;
  double2 r128; r128.x = 0; r128.y = 0;
  long long t0 = __builtin_readcyclecounter();
  const bool on = MODE == 0 ? true : (MODE == 1 ? lane < 2 : (MODE == 2 ? lane == 37 : false));
  for (int it = 0; it < iters; ++it) {
    if (on) { REP16(LD128) }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  long long t1 = __builtin_readcyclecounter();
  if (lane == 0) out[blockIdx.x * (blockDim.x >> 6) + w] = t1 - t0;
  sink[blockIdx.x * blockDim.x + threadIdx.x] = r128.x;
}
template <int MODE>
void run(const char *name, int threads) {
  long long *d; double *s; const int blocks = 256, iters = 2000;
  hipMalloc(&d, 8 * 4096); hipMalloc(&s, 8 * blocks * threads);
  hipFuncSetAttribute((const void *)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 20000 * 8, 0, d, s, iters);
  hipDeviceSynchronize();
  std::vector<long long> h(blocks * (threads / 64));
  hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
  double avg = 0; for (auto v : h) avg += v; avg /= h.size();
  printf("%-40s waves/CU=%d: %.1f cycles per instruction\n", name, threads / 64, avg / (iters * 16.0));
  hipFree(d); hipFree(s);
}
int main() {
  for (int threads : {64, 192}) {
    run<0>("ds_read_b128, 64 lanes active", threads);
    run<1>("ds_read_b128, 2 lanes active", threads);
    run<2>("ds_read_b128, 1 lane active", threads);
    run<3>("skipped by s_cbranch_execz (0 lanes)", threads);
  }
  return 0;
}
