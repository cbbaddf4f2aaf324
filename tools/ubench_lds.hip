// Micro-benchmark (developer tool): LDS throughput of the sweep's access patterns, 3 waves/CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
extern __shared__ __attribute__((aligned(16))) double lds[];

template <int MODE>
__global__ void k(long long *out, double *sink, int iters) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  double *E = lds + 1024 + w * 6400;
  for (int i = lane; i < 6400; i += 64) E[i] = 1.0 + i * 1e-9;
  for (int i = threadIdx.x; i < 1024; i += blockDim.x) lds[i] = 0.25;
  __syncthreads();
  double acc0 = 0, acc1 = 0, acc2 = 0, acc3 = 0;
  const int base = lane * 97;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    const int o = (it * 8) % 96;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (MODE == 0) { // ds_read_b128, every lane the same address (table row of the common class)
        const double2 v = *(const double2 *)(lds + ((it + k) & 31) * 4);
        acc0 += v.x; acc1 += v.y;
      } else if (MODE == 1) { // ds_read_b128, lanes spread over 4 rows (classes)
        const double2 v = *(const double2 *)(lds + (((lane >> 4) * 5 + it + k) & 31) * 4);
        acc0 += v.x; acc1 += v.y;
      } else if (MODE == 2) { // ds_read_b64 skewed row-walk (stride 97 doubles between lanes)
        acc0 += E[base + o + k];
      } else if (MODE == 3) { // ds_write_b64 skewed row-walk
        E[base + o + k] = acc0 + k;
      } else if (MODE == 4) { // the sweep's mix: 3 b128 + 3 b64 + 1 write
        const double2 a = *(const double2 *)(lds + ((it + k) & 31) * 4);
        const double2 b = *(const double2 *)(lds + ((it + k) & 31) * 4 + 2);
        const double2 c = *(const double2 *)(lds + 512 + ((it + k) & 31) * 2);
        acc0 += a.x + b.x + c.x; acc1 += a.y + b.y + c.y;
        acc2 += E[base + o + k + 1] + E[base + o + k + 98];
        acc3 += E[(base + o + k + 6400 - 98) % 6400];
        E[base + o + k] = acc0;
      }
    }
  }
  long long t1 = __builtin_readcyclecounter();
  if (lane == 0) out[blockIdx.x * (blockDim.x >> 6) + w] = t1 - t0;
  sink[blockIdx.x * blockDim.x + threadIdx.x] = acc0 + acc1 + acc2 + acc3;
}
template <int MODE>
void run(const char *name, int threads) {
  long long *d; double *s; const int blocks = 256, iters = 4000;
  hipMalloc(&d, 8 * 4096); hipMalloc(&s, 8 * blocks * threads);
  hipFuncSetAttribute((const void *)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), (1024 + 3 * 6400) * 8, 0, d, s, iters);
  hipDeviceSynchronize();
  std::vector<long long> h(blocks * (threads / 64));
  hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
  double avg = 0; for (auto v : h) avg += v; avg /= h.size();
  printf("%-52s waves/CU=%d: %.1f cycles per op-group\n", name, threads / 64, avg / (iters * 8.0));
  hipFree(d); hipFree(s);
}
int main() {
  for (int threads : {64, 192}) {
    run<0>("ds_read_b128 same address (x1)", threads);
    run<1>("ds_read_b128 4 distinct rows (x1)", threads);
    run<2>("ds_read_b64 skewed stride-97 (x1)", threads);
    run<3>("ds_write_b64 skewed stride-97 (x1)", threads);
    run<4>("sweep mix: 3 b128 + 3 b64 + 1 write", threads);
  }
  return 0;
}
