// Micro-benchmark 3 (developer tool): raw LDS instruction throughput per wave, inline asm:
// 16 loads issued back-to-back, one s_waitcnt, repeated.  cycles per instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
extern __shared__ __attribute__((aligned(16))) double lds[];

#define LD64(off)  asm volatile("ds_read_b64 %0, %1 offset:" #off : "=v"(r64) : "v"(addr));
#define LD128(off) asm volatile("ds_read_b128 %0, %1 offset:" #off : "=v"(r128) : "v"(addr));
#define ST64(off)  asm volatile("ds_write_b64 %0, %1 offset:" #off :: "v"(addr), "v"(r64));
#define REP16(M) M(0) M(16) M(32) M(48) M(64) M(80) M(96) M(112) M(128) M(144) M(160) M(176) M(192) M(208) M(224) M(240)

template <int MODE>
__global__ void k(long long *out, double *sink, int iters) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 20000; i += blockDim.x) lds[i] = 1.0;
  __syncthreads();
  unsigned base = (unsigned)(size_t)(lds + 512 + w * 6400) & 0xffffffffu;
  unsigned addr;
  if (MODE == 0 || MODE == 3 || MODE == 5) addr = base + lane * 784;       // distinct, conflict-free stride
  else if (MODE == 1 || MODE == 4) addr = base;                             // all lanes same address
  else addr = base + (lane & 3) * 32;                                       // 4 distinct rows
  double r64 = 1.0; double2 r128;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0 || MODE == 1 || MODE == 2) { REP16(LD64) }
    else if (MODE == 3 || MODE == 4 || MODE == 6) { REP16(LD128) }
    else { REP16(ST64) }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  long long t1 = __builtin_readcyclecounter();
  if (lane == 0) out[blockIdx.x * (blockDim.x >> 6) + w] = t1 - t0;
  sink[blockIdx.x * blockDim.x + threadIdx.x] = r64 + r128.x;
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
  printf("%-44s waves/CU=%d: %.1f cycles per instruction\n", name, threads / 64, avg / (iters * 16.0));
  hipFree(d); hipFree(s);
}
int main() {
  for (int threads : {64, 192}) {
    run<0>("ds_read_b64 distinct (stride 784B)", threads);
    run<1>("ds_read_b64 same address", threads);
    run<2>("ds_read_b64 4 rows", threads);
    run<3>("ds_read_b128 distinct (stride 784B)", threads);
    run<4>("ds_read_b128 same address", threads);
    run<6>("ds_read_b128 4 rows", threads);
    run<5>("ds_write_b64 distinct (stride 784B)", threads);
  }
  return 0;
}
