// Micro-benchmarks (developer tool): per-iteration shader cycles of the building blocks of the
// sweep loop, one wave per CU-slot.  Prints cycles/iteration for each variant.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
extern __shared__ __attribute__((aligned(16))) double lds[];

__device__ __forceinline__ double shr1(double x, double seam) {
  int lo = __builtin_amdgcn_update_dpp(__double2loint(seam), __double2loint(x), 0x138, 0xf, 0xf, false);
  int hi = __builtin_amdgcn_update_dpp(__double2hiint(seam), __double2hiint(x), 0x138, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double rowshr1(double x, double seam) {
  int lo = __builtin_amdgcn_update_dpp(__double2loint(seam), __double2loint(x), 0x111, 0xf, 0xf, false);
  int hi = __builtin_amdgcn_update_dpp(__double2hiint(seam), __double2hiint(x), 0x111, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}

template <int MODE>
__global__ void k(long long *out, double *sink, int iters, double c0) {
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = 1.0 + i * 1e-9;
  __syncthreads();
  double nv = 1.0 + lane * 1e-6, acc = 0.0;
  double b0 = c0, b1 = c0 * 1.01, b2 = c0 * 0.99, b3 = c0 * 1.02;
  int li = lane * 97;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (MODE == 0) { // 4 dependent fp64 fmac
        double t = fma(b0, nv, acc); t = fma(b1, nv, t); t = fma(b2, nv, t); nv = fma(b3, nv, t) * 0.25;
      } else if (MODE == 1) { // 2 dependent fmac + wave_shr DPP
        double U = shr1(nv, 0.0);
        double t = fma(b2, nv, acc); nv = fma(b3, U, t) * 0.5;
      } else if (MODE == 2) { // 2 dependent fmac + row_shr DPP
        double U = rowshr1(nv, 0.0);
        double t = fma(b2, nv, acc); nv = fma(b3, U, t) * 0.5;
      } else if (MODE == 3) { // one ds_read_b64 (independent, prefetchable) + write per step
        double r = lds[(li + it * 8 + k) & 8191];
        nv = fma(b0, nv, r) * 0.5;
        lds[(li + it * 8 + k + 4096) & 8191] = nv;
      } else if (MODE == 4) { // 3 ds_read_b128 (table-like, same address) + 3 ds_read_b64 + write
        const double2 a = *(const double2 *)(lds + ((k * 4) & 63));
        const double2 b = *(const double2 *)(lds + ((k * 4 + 2) & 63));
        const double2 c = *(const double2 *)(lds + 64 + ((k * 2) & 63));
        double r = lds[(li + it * 8 + k) & 4095], d = lds[(li + 97 + it * 8 + k) & 4095], e = lds[(li + 194 + it * 8 + k) & 4095];
        double t = fma(a.x, r, c.y); t = fma(a.y, d, t); t = fma(b.x, e, t); nv = fma(b.y, nv, t) * 0.25 + c.x * 1e-9;
        lds[4096 + ((li + it * 8 + k) & 4095)] = nv;
      }
    }
  }
  long long t1 = __builtin_readcyclecounter();
  if (lane == 0) out[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
  sink[blockIdx.x * blockDim.x + threadIdx.x] = nv + acc;
}

template <int MODE>
void run(const char *name, int blocks, int threads) {
  long long *d; double *s;
  hipMalloc(&d, 8 * 4096); hipMalloc(&s, 8 * blocks * threads);
  const int iters = 2000;
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 65536, 0, d, s, iters, 0.2499);
  hipDeviceSynchronize();
  std::vector<long long> h(blocks * (threads / 64));
  hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
  double avg = 0; for (auto v : h) avg += v; avg /= h.size();
  printf("%-44s blocks=%d threads=%d: %.1f cycles per step\n", name, blocks, threads, avg / (iters * 8.0));
  hipFree(d); hipFree(s);
}
int main() {
  for (int threads : {64, 192}) {
    run<0>("4 dependent v_fma_f64 (+1 mul)", 256, threads);
    run<1>("2 dep fma + DPP wave_shr:1", 256, threads);
    run<2>("2 dep fma + DPP row_shr:1", 256, threads);
    run<3>("ds_read_b64 + fma + ds_write_b64", 256, threads);
    run<4>("3 b128 + 3 b64 reads + 4 fma + write", 256, threads);
  }
  return 0;
}
