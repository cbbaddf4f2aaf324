// tools/ubench_hbm.hip -- what this box's HBM delivers to hand-written streaming kernels (the numbers the
// kernels' "practical ceiling" is priced on: profiles/r07_hbm_ubench.txt).  16-byte accesses per lane (global_load_dwordx4 /
// global_store_dwordx4), a grid-stride loop over 4 GiB per stream, with and without the non-temporal hint, several grid sizes:
//   read    : sum of one stream (one 8-byte store per workgroup)
//   write   : one stream of stores
//   copy    : 1 read : 1 write
//   two2one : 2 reads : 1 write (k_sweep_stream's mix: grid + A in, grid out)
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_hbm.hip -o tools/bin/ubench_hbm && tools/bin/ubench_hbm
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef double d2 __attribute__((ext_vector_type(2)));

template <bool NT>
__device__ __forceinline__ d2 ld(const d2 *p) {
  if constexpr (NT) return __builtin_nontemporal_load(p);
  else return *p;
}
template <bool NT>
__device__ __forceinline__ void st(d2 *p, d2 v) {
  if constexpr (NT) __builtin_nontemporal_store(v, p);
  else *p = v;
}

constexpr int kU = 4; // 16-byte accesses in flight per lane and stream

template <bool NT>
__global__ void __launch_bounds__(256) k_read(const d2 *__restrict__ a, double *__restrict__ out, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  d2 acc = {0.0, 0.0};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i + (kU - 1) * stride < n; i += kU * stride) {
    d2 v[kU];
#pragma unroll
    for (int k = 0; k < kU; ++k) v[k] = ld<NT>(a + i + k * stride);
#pragma unroll
    for (int k = 0; k < kU; ++k) acc += v[k];
  }
  if (acc.x + acc.y == 123.456) out[blockIdx.x] = acc.x; // (never: the loads must not be dead)
}
template <bool NT>
__global__ void __launch_bounds__(256) k_write(d2 *__restrict__ a, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const d2 v = {1.0, 2.0};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i + (kU - 1) * stride < n; i += kU * stride)
#pragma unroll
    for (int k = 0; k < kU; ++k) st<NT>(a + i + k * stride, v);
}
template <bool NT>
__global__ void __launch_bounds__(256) k_copy(const d2 *__restrict__ a, d2 *__restrict__ b, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i + (kU - 1) * stride < n; i += kU * stride) {
    d2 v[kU];
#pragma unroll
    for (int k = 0; k < kU; ++k) v[k] = ld<NT>(a + i + k * stride);
#pragma unroll
    for (int k = 0; k < kU; ++k) st<NT>(b + i + k * stride, v[k]);
  }
}
template <bool NT>
__global__ void __launch_bounds__(256) k_two2one(const d2 *__restrict__ a, const d2 *__restrict__ c, d2 *__restrict__ b, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i + (kU - 1) * stride < n; i += kU * stride) {
    d2 v[kU], w[kU];
#pragma unroll
    for (int k = 0; k < kU; ++k) { v[k] = ld<NT>(a + i + k * stride); w[k] = ld<NT>(c + i + k * stride); }
#pragma unroll
    for (int k = 0; k < kU; ++k) st<NT>(b + i + k * stride, v[k] + w[k]);
  }
}

// copy in CHUNKS: a wavefront reads U contiguous KB (U 16-byte loads per lane in flight), then writes them -- the shape of
// k_sweep_roll's hand-over (a building's 48 KB of rows out, the next building's in), which sustains more than the
// element-interleaved copy above: longer same-direction bursts per HBM channel
template <bool NT, int U>
__global__ void __launch_bounds__(256) k_copy_chunk(const d2 *__restrict__ a, d2 *__restrict__ b, size_t n) {
  const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, waves = ((size_t)gridDim.x * blockDim.x) >> 6;
  const int lane = threadIdx.x & 63;
  for (size_t c = wave; (c + 1) * U * 64 <= n; c += waves) {
    const size_t base = c * U * 64 + lane;
    d2 v[U];
#pragma unroll
    for (int k = 0; k < U; ++k) v[k] = ld<NT>(a + base + k * 64);
#pragma unroll
    for (int k = 0; k < U; ++k) st<NT>(b + base + k * 64, v[k]);
  }
}

// the same with the next chunk's loads issued BEFORE the current chunk's stores (what the hand-over does: store a slot, load
// the same slot of the next building)
template <bool NT, int U>
__global__ void __launch_bounds__(256) k_copy_pipe(const d2 *__restrict__ a, d2 *__restrict__ b, size_t n) {
  const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, waves = ((size_t)gridDim.x * blockDim.x) >> 6;
  const int lane = threadIdx.x & 63;
  const size_t chunks = n / (U * 64);
  d2 v[U], w[U];
  size_t c = wave;
  if (c < chunks)
#pragma unroll
    for (int k = 0; k < U; ++k) v[k] = ld<NT>(a + c * U * 64 + lane + k * 64);
  for (; c < chunks; c += 2 * waves) {
    const size_t c1 = c + waves, c2 = c + 2 * waves;
    if (c1 < chunks)
#pragma unroll
      for (int k = 0; k < U; ++k) w[k] = ld<NT>(a + c1 * U * 64 + lane + k * 64);
#pragma unroll
    for (int k = 0; k < U; ++k) st<NT>(b + c * U * 64 + lane + k * 64, v[k]);
    if (c2 < chunks)
#pragma unroll
      for (int k = 0; k < U; ++k) v[k] = ld<NT>(a + c2 * U * 64 + lane + k * 64);
    if (c1 < chunks)
#pragma unroll
      for (int k = 0; k < U; ++k) st<NT>(b + c1 * U * 64 + lane + k * 64, w[k]);
  }
}

// 8-byte accesses (512 bytes per wavefront instruction: k_sweep_roll's rows), chunks of U x 512 B, next chunk's loads first
template <bool NT, int U>
__global__ void __launch_bounds__(256) k_copy_pipe8(const double *__restrict__ a, double *__restrict__ b, size_t n) {
  const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, waves = ((size_t)gridDim.x * blockDim.x) >> 6;
  const int lane = threadIdx.x & 63;
  const size_t chunks = n / (U * 64);
  double v[U], w[U];
  size_t c = wave;
  auto L = [&](const double *p) { if constexpr (NT) return __builtin_nontemporal_load(p); else return *p; };
  auto S = [&](double *p, double x) { if constexpr (NT) __builtin_nontemporal_store(x, p); else *p = x; };
  if (c < chunks)
#pragma unroll
    for (int k = 0; k < U; ++k) v[k] = L(a + c * U * 64 + lane + k * 64);
  for (; c < chunks; c += 2 * waves) {
    const size_t c1 = c + waves, c2 = c + 2 * waves;
    if (c1 < chunks)
#pragma unroll
      for (int k = 0; k < U; ++k) w[k] = L(a + c1 * U * 64 + lane + k * 64);
#pragma unroll
    for (int k = 0; k < U; ++k) S(b + c * U * 64 + lane + k * 64, v[k]);
    if (c2 < chunks)
#pragma unroll
      for (int k = 0; k < U; ++k) v[k] = L(a + c2 * U * 64 + lane + k * 64);
    if (c1 < chunks)
#pragma unroll
      for (int k = 0; k < U; ++k) S(b + c1 * U * 64 + lane + k * 64, w[k]);
  }
}

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char **argv) {
  const size_t bytes = (size_t)(argc > 1 ? atof(argv[1]) : 4.0) * (1ull << 30), n = bytes / 16;
  d2 *a, *b, *c;
  double *out;
  CHECK(hipMalloc(&a, bytes)); CHECK(hipMalloc(&b, bytes)); CHECK(hipMalloc(&c, bytes)); CHECK(hipMalloc(&out, 1 << 20));
  CHECK(hipMemset(a, 0, bytes)); CHECK(hipMemset(b, 0, bytes)); CHECK(hipMemset(c, 0, bytes));
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  printf("%s, %d CUs; %.1f GiB per stream, 16-byte accesses, %d per lane and stream in flight; best of 5 launches\n", prop.name, prop.multiProcessorCount, bytes / 1073741824.0, kU);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  auto time = [&](auto launch) {
    float best = 1e30f;
    for (int rep = 0; rep < 6; ++rep) {
      hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (rep > 0 && ms < best) best = ms;
    }
    return best;
  };
  for (int wg_per_cu : {4, 8, 16, 32}) {
    const int grid = prop.multiProcessorCount * wg_per_cu;
    struct { const char *name; double streams; float ms[2]; } rows[4] = {{"read", 1, {}}, {"write", 1, {}}, {"copy 1:1", 2, {}}, {"two reads : one write", 3, {}}};
    rows[0].ms[0] = time([&] { hipLaunchKernelGGL(k_read<false>, dim3(grid), dim3(256), 0, 0, a, out, n); });
    rows[0].ms[1] = time([&] { hipLaunchKernelGGL(k_read<true>, dim3(grid), dim3(256), 0, 0, a, out, n); });
    rows[1].ms[0] = time([&] { hipLaunchKernelGGL(k_write<false>, dim3(grid), dim3(256), 0, 0, b, n); });
    rows[1].ms[1] = time([&] { hipLaunchKernelGGL(k_write<true>, dim3(grid), dim3(256), 0, 0, b, n); });
    rows[2].ms[0] = time([&] { hipLaunchKernelGGL(k_copy<false>, dim3(grid), dim3(256), 0, 0, a, b, n); });
    rows[2].ms[1] = time([&] { hipLaunchKernelGGL(k_copy<true>, dim3(grid), dim3(256), 0, 0, a, b, n); });
    rows[3].ms[0] = time([&] { hipLaunchKernelGGL(k_two2one<false>, dim3(grid), dim3(256), 0, 0, a, c, b, n); });
    rows[3].ms[1] = time([&] { hipLaunchKernelGGL(k_two2one<true>, dim3(grid), dim3(256), 0, 0, a, c, b, n); });
    for (auto &r : rows)
      printf("%2d workgroups per CU  %-22s plain %6.3f ms = %5.2f TB/s    nt %6.3f ms = %5.2f TB/s\n", wg_per_cu, r.name, r.ms[0],
             r.streams * bytes / (r.ms[0] * 1e-3) / 1e12, r.ms[1], r.streams * bytes / (r.ms[1] * 1e-3) / 1e12);
  }
  for (int wg_per_cu : {1, 2, 4, 8}) {
    const int grid = prop.multiProcessorCount * wg_per_cu;
    float ms[6];
    ms[0] = time([&] { hipLaunchKernelGGL((k_copy_chunk<false, 16>), dim3(grid), dim3(256), 0, 0, a, b, n); });
    ms[1] = time([&] { hipLaunchKernelGGL((k_copy_chunk<true, 16>), dim3(grid), dim3(256), 0, 0, a, b, n); });
    ms[2] = time([&] { hipLaunchKernelGGL((k_copy_chunk<false, 32>), dim3(grid), dim3(256), 0, 0, a, b, n); });
    ms[3] = time([&] { hipLaunchKernelGGL((k_copy_chunk<true, 32>), dim3(grid), dim3(256), 0, 0, a, b, n); });
    ms[4] = time([&] { hipLaunchKernelGGL((k_copy_chunk<false, 48>), dim3(grid), dim3(256), 0, 0, a, b, n); });
    ms[5] = time([&] { hipLaunchKernelGGL((k_copy_chunk<true, 48>), dim3(grid), dim3(256), 0, 0, a, b, n); });
    float msp[4];
    msp[0] = time([&] { hipLaunchKernelGGL((k_copy_pipe<false, 24>), dim3(grid), dim3(256), 0, 0, a, b, n); });
    msp[1] = time([&] { hipLaunchKernelGGL((k_copy_pipe<true, 24>), dim3(grid), dim3(256), 0, 0, a, b, n); });
    msp[2] = time([&] { hipLaunchKernelGGL((k_copy_pipe<false, 24>), dim3(grid), dim3(256), 0, 0, a, a, n); });
    msp[3] = time([&] { hipLaunchKernelGGL((k_copy_pipe<true, 24>), dim3(grid), dim3(256), 0, 0, a, a, n); });
    printf("%2d workgroups per CU  copy, next chunk's loads ahead of the stores, 24 KB chunks: plain %5.2f TB/s  nt %5.2f TB/s   in place: plain %5.2f TB/s  nt %5.2f TB/s\n",
           wg_per_cu, 2.0 * bytes / (msp[0] * 1e-3) / 1e12, 2.0 * bytes / (msp[1] * 1e-3) / 1e12, 2.0 * bytes / (msp[2] * 1e-3) / 1e12, 2.0 * bytes / (msp[3] * 1e-3) / 1e12);
    float ms8[4];
    ms8[0] = time([&] { hipLaunchKernelGGL((k_copy_pipe8<false, 96>), dim3(grid), dim3(256), 0, 0, (const double *)a, (double *)b, 2 * n); });
    ms8[1] = time([&] { hipLaunchKernelGGL((k_copy_pipe8<true, 96>), dim3(grid), dim3(256), 0, 0, (const double *)a, (double *)b, 2 * n); });
    ms8[2] = time([&] { hipLaunchKernelGGL((k_copy_pipe8<false, 96>), dim3(grid), dim3(256), 0, 0, (const double *)a, (double *)a, 2 * n); });
    ms8[3] = time([&] { hipLaunchKernelGGL((k_copy_pipe8<true, 96>), dim3(grid), dim3(256), 0, 0, (const double *)a, (double *)a, 2 * n); });
    printf("%2d workgroups per CU  the same with 8-byte accesses, 96 x 512 B chunks:              plain %5.2f TB/s  nt %5.2f TB/s   in place: plain %5.2f TB/s  nt %5.2f TB/s\n",
           wg_per_cu, 2.0 * bytes / (ms8[0] * 1e-3) / 1e12, 2.0 * bytes / (ms8[1] * 1e-3) / 1e12, 2.0 * bytes / (ms8[2] * 1e-3) / 1e12, 2.0 * bytes / (ms8[3] * 1e-3) / 1e12);
    float msi[2];
    msi[0] = time([&] { hipLaunchKernelGGL((k_copy_chunk<false, 48>), dim3(grid), dim3(256), 0, 0, a, a, n); });
    msi[1] = time([&] { hipLaunchKernelGGL((k_copy_chunk<true, 48>), dim3(grid), dim3(256), 0, 0, a, a, n); });
    printf("%2d workgroups per CU  IN PLACE (read a chunk, write it back: the sweep kernels' state), 48 KB chunks  plain %6.3f ms = %5.2f TB/s   nt %6.3f ms = %5.2f TB/s\n",
           wg_per_cu, msi[0], 2.0 * bytes / (msi[0] * 1e-3) / 1e12, msi[1], 2.0 * bytes / (msi[1] * 1e-3) / 1e12);
    const char *names[6] = {"16 KB chunks", "16 KB chunks, nt", "32 KB chunks", "32 KB chunks, nt", "48 KB chunks", "48 KB chunks, nt"};
    for (int k = 0; k < 6; ++k)
      printf("%2d workgroups per CU  copy in %-18s %6.3f ms = %5.2f TB/s\n", wg_per_cu, names[k], ms[k], 2.0 * bytes / (ms[k] * 1e-3) / 1e12);
  }
  CHECK(hipGetLastError());
  return 0;
}
