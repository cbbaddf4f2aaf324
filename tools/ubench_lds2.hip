// Micro-benchmark 2 (developer tool): per-wave issue cost of candidate access patterns.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
extern __shared__ __attribute__((aligned(16))) double lds[];

template <int MODE>
__global__ void k(long long *out, double *sink, const double *tab, const unsigned char *cls, int iters) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  double *E = lds + 1024 + w * 6400;
  for (int i = lane; i < 6400; i += 64) E[i] = 1.0 + i * 1e-9;
  for (int i = threadIdx.x; i < 1024; i += blockDim.x) lds[i] = 0.25;
  __syncthreads();
  double acc0 = 0, acc1 = 0, acc2 = 0, acc3 = 0;
  const int base = lane * 98;          // odd pitch 99: lane stride 98 doubles
  const int c = cls[lane];             // a "class" per lane (0..21)
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    const int o = (it * 8) % 88;
    if (MODE == 0) { // 4 aligned ds_read_b128 per 8 steps (own row, 16B aligned, stride 98)
#pragma unroll
      for (int k = 0; k < 4; ++k) { const double2 v = *(const double2 *)(E + base + o + 2 * k); acc0 += v.x; acc1 += v.y; }
    } else if (MODE == 1) { // 8 ds_read_b64 per 8 steps (baseline)
#pragma unroll
      for (int k = 0; k < 8; ++k) acc0 += E[base + o + k];
    } else if (MODE == 2) { // 4 ds_write_b128 per 8 steps
#pragma unroll
      for (int k = 0; k < 4; ++k) { double2 v; v.x = acc0 + k; v.y = acc1 + k; *(double2 *)(E + base + o + 2 * k) = v; }
    } else if (MODE == 3) { // 8 x 3 global_load_dwordx4 from a 1 KiB table (L1 resident), by class
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const double2 *row = (const double2 *)(tab + ((c + k + it) % 22) * 6);
        const double2 a = row[0], b = row[1], d = row[2];
        acc0 += a.x + b.x + d.x; acc1 += a.y + b.y + d.y;
      }
    } else if (MODE == 4) { // 8 x 3 ds_read_b128 table (current design)
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const double2 *row = (const double2 *)(lds + ((c + k + it) % 22) * 6);
        const double2 a = row[0], b = row[1], d = row[2];
        acc0 += a.x + b.x + d.x; acc1 += a.y + b.y + d.y;
      }
    } else if (MODE == 5) { // 8 x ds_read2_b64 (two values 98 doubles apart)
#pragma unroll
      for (int k = 0; k < 8; ++k) { acc0 += E[base + o + k]; acc1 += E[base + o + k + 98]; }
    } else if (MODE == 6) { // candidate mix per 8 steps: 4 b128 E reads + 4 b128 writes + 24 global table loads
#pragma unroll
      for (int k = 0; k < 4; ++k) { const double2 v = *(const double2 *)(E + base + o + 2 * k); acc0 += v.x; acc1 += v.y; }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const double2 *row = (const double2 *)(tab + ((c + k + it) % 22) * 6);
        const double2 a = row[0], b = row[1], d = row[2];
        acc2 += a.x + b.x + d.x; acc3 += a.y + b.y + d.y;
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) { double2 v; v.x = acc0 + k; v.y = acc2 + k; *(double2 *)(E + 3200 + ((base + o + 2 * k) % 3200)) = v; }
    } else if (MODE == 7) { // candidate mix B: 4 b128 E reads + 4 b128 writes + 24 LDS table b128
#pragma unroll
      for (int k = 0; k < 4; ++k) { const double2 v = *(const double2 *)(E + base + o + 2 * k); acc0 += v.x; acc1 += v.y; }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const double2 *row = (const double2 *)(lds + ((c + k + it) % 22) * 6);
        const double2 a = row[0], b = row[1], d = row[2];
        acc2 += a.x + b.x + d.x; acc3 += a.y + b.y + d.y;
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) { double2 v; v.x = acc0 + k; v.y = acc2 + k; *(double2 *)(E + 3200 + ((base + o + 2 * k) % 3200)) = v; }
    }
  }
  long long t1 = __builtin_readcyclecounter();
  if (lane == 0) out[blockIdx.x * (blockDim.x >> 6) + w] = t1 - t0;
  sink[blockIdx.x * blockDim.x + threadIdx.x] = acc0 + acc1 + acc2 + acc3;
}
template <int MODE>
void run(const char *name, int threads) {
  long long *d; double *s, *tab; unsigned char *cls; const int blocks = 256, iters = 4000;
  hipMalloc(&d, 8 * 4096); hipMalloc(&s, 8 * blocks * threads); hipMalloc(&tab, 8 * 6 * 32); hipMalloc(&cls, 64);
  std::vector<double> ht(6 * 32, 0.25); hipMemcpy(tab, ht.data(), ht.size() * 8, hipMemcpyHostToDevice);
  unsigned char hc[64]; for (int i = 0; i < 64; ++i) hc[i] = (i % 9 == 0) ? (i % 22) : 5; hipMemcpy(cls, hc, 64, hipMemcpyHostToDevice);
  hipFuncSetAttribute((const void *)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), (1024 + 3 * 6400) * 8, 0, d, s, tab, cls, iters);
  hipDeviceSynchronize();
  std::vector<long long> h(blocks * (threads / 64));
  hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
  double avg = 0; for (auto v : h) avg += v; avg /= h.size();
  printf("%-60s waves/CU=%d: %.1f cycles per 8 steps\n", name, threads / 64, avg / iters);
  hipFree(d); hipFree(s); hipFree(tab); hipFree(cls);
}
int main() {
  for (int threads : {64, 192}) {
    run<0>("4 aligned ds_read_b128 (row walk, stride 98)", threads);
    run<1>("8 ds_read_b64 (row walk)", threads);
    run<2>("4 ds_write_b128 (row walk)", threads);
    run<3>("24 global_load_dwordx4 from 1KiB table by class", threads);
    run<4>("24 ds_read_b128 from LDS table by class", threads);
    run<5>("8 x (2 ds_read_b64 98 apart) [ds_read2?]", threads);
    run<6>("mix A: 4 b128 rd + 24 global tab + 4 b128 wr", threads);
    run<7>("mix B: 4 b128 rd + 24 LDS tab + 4 b128 wr", threads);
  }
  return 0;
}
