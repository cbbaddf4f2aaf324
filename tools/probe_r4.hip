// Round-4 probes (developer tool):
//  1. does the gfx950 hardware apply wave_shr / wave_shl / row_shr DPP controls to 64-bit ("DP ALU") VALU
//     instructions?  The assembler refuses them ("DP ALU dpp only supports row_newbcast"), so the
//     instructions are hand-encoded.
//  2. v_max_f32_dpp with |src1| and wave_shr:1 bound_ctrl (the travelling max|delta| accumulator)
//  3. LDS cycles of ds_write_b64 / ds_write2_b64 / ds_write_b128 with 64 lanes or ONE lane active, four
//     wavefronts per CU (bulk publishing of row 63 by lane 63)
//   hipcc --offload-arch=gfx950 -O2 tools/probe_r4.hip -o /tmp/probe_r4 && timeout 60 /tmp/probe_r4
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>

extern __shared__ __attribute__((aligned(16))) double lds[];
typedef double d2 __attribute__((ext_vector_type(2)));

// MODE 0: v_mov_b64_dpp wave_shr:1 bound_ctrl ; 1: wave_shl:1 ; 2: row_shr:1 ; 3: v_fmac_f64_dpp wave_shr:1 (acc += dpp(x) * 2.0)
// 4: v_add_f64_dpp? (VOP3: no dpp encoding on gfx9) -- skipped
template <int MODE>
__global__ void k_dp_dpp(double *out) {
  const int lane = threadIdx.x;
  double x = 100.0 + lane, r = -1.0, two = 2.0;
  // v_mov_b64_dpp vdst, vsrc: VOP1 opcode 0x38 (v_mov_b64), encoding 0x7e000000 | vdst << 17 | op << 9 | 0xfa ; dword 2: src0 | dpp_ctrl << 8 | bound_ctrl << 19 | bank 0xf << 24 | row 0xf << 28
  if (MODE == 0)
    asm volatile("v_mov_b64 v[2:3], %1\n v_mov_b64 v[4:5], %2\n s_nop 4\n"
                 ".long 0x7e0470fa\n .long 0xff0b3804\n" // v_mov_b64_dpp v[2:3], v[4:5] wave_shr:1 bound_ctrl
                 "s_nop 4\n v_mov_b64 %0, v[2:3]\n" : "=v"(r) : "v"(r), "v"(x) : "v2", "v3", "v4", "v5");
  else if (MODE == 1)
    asm volatile("v_mov_b64 v[2:3], %1\n v_mov_b64 v[4:5], %2\n s_nop 4\n"
                 ".long 0x7e0470fa\n .long 0xff0b3004\n"
                 "s_nop 4\n v_mov_b64 %0, v[2:3]\n" : "=v"(r) : "v"(r), "v"(x) : "v2", "v3", "v4", "v5");
  else if (MODE == 2)
    asm volatile("v_mov_b64 v[2:3], %1\n v_mov_b64 v[4:5], %2\n s_nop 4\n"
                 ".long 0x7e0470fa\n .long 0xff091104\n"
                 "s_nop 4\n v_mov_b64 %0, v[2:3]\n" : "=v"(r) : "v"(r), "v"(x) : "v2", "v3", "v4", "v5");
  else if (MODE == 3) // v_fmac_f64_dpp v[2:3], v[4:5], v[6:7]: VOP2 op 4: 0x08 << 24 | vdst << 17 | vsrc1 << 9 | 0xfa
    asm volatile("v_mov_b64 v[2:3], %1\n v_mov_b64 v[4:5], %2\n v_mov_b64 v[6:7], %3\n s_nop 4\n"
                 ".long 0x08040cfa\n .long 0xff0b3804\n" // v[2:3] += dpp(v[4:5]) * v[6:7]
                 "s_nop 4\n v_mov_b64 %0, v[2:3]\n" : "=v"(r) : "v"(r), "v"(x), "v"(two) : "v2", "v3", "v4", "v5", "v6", "v7");
  out[lane] = r;
}

__global__ void k_dppmax(float *out) {
  const int lane = threadIdx.x;
  float acc = 1000.0f + lane, d = (lane & 1) ? -(float)lane * 3.0f : (float)lane * 3.0f;
  asm volatile("s_nop 4\n v_max_f32_dpp %0, %0, |%1| wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n s_nop 4" : "+v"(acc) : "v"(d));
  out[lane] = acc;
}

template <int MODE> // 0: ds_write_b64 all lanes; 1: lane 63 only; 2: ds_write2_b64 all; 3: lane 63; 4: ds_write_b128 all; 5: lane 63; 6: b64 lane 63 via per-lane address (others to a strip)
__global__ void k_ldsw(long long *out, int iters) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 19000; i += blockDim.x) lds[i] = 1.0;
  __syncthreads();
  const unsigned base = (unsigned)(size_t)(lds + w * 4600);
  unsigned addr = base + lane * 8;
  if (MODE == 4 || MODE == 5) addr = base + lane * 16;
  if (MODE == 6) addr = lane == 63 ? base : base + 2048 + lane * 8;
  double a = 1.0 + lane, b = 2.0 + lane;
  const bool on = (MODE & 1) && MODE != 6 ? lane == 63 : true;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if (on) {
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        if (MODE == 0 || MODE == 1 || MODE == 6) asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(addr), "v"(a), "n"(k * 16));
        else if (MODE == 2 || MODE == 3) asm volatile("ds_write2_b64 %0, %1, %2 offset0:%3 offset1:%4" ::"v"(addr), "v"(a), "v"(b), "n"(2 * k), "n"(2 * k + 1));
        else asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(addr), "v"(d2{a, b}), "n"(k * 16));
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  long long t1 = __builtin_readcyclecounter();
  if (lane == 0) out[blockIdx.x * (blockDim.x >> 6) + w] = t1 - t0;
}

template <int MODE>
void run_lds(const char *name) {
  long long *d;
  const int blocks = 256, threads = 256, iters = 2000;
  hipMalloc(&d, 8 * 4096);
  hipFuncSetAttribute((const void *)k_ldsw<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
  hipLaunchKernelGGL(k_ldsw<MODE>, dim3(blocks), dim3(threads), 19000 * 8, 0, d, iters);
  hipDeviceSynchronize();
  std::vector<long long> h(blocks * (threads / 64));
  hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
  double avg = 0;
  for (auto v : h) avg += v;
  avg /= h.size();
  printf("%-44s 4 waves/CU: %.1f cycles per instruction (per wavefront; the pipe serves four)\n", name, avg / (iters * 16.0));
  hipFree(d);
}

template <int MODE>
void run_dp(const char *name) {
  double *d, h[64];
  hipMalloc(&d, 64 * 8);
  hipLaunchKernelGGL(k_dp_dpp<MODE>, dim3(1), dim3(64), 0, 0, d);
  hipError_t e = hipDeviceSynchronize();
  printf("%s (%s):", name, hipGetErrorString(e));
  if (e == hipSuccess) {
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int i = 0; i < 64; ++i) printf(" %g", h[i]);
  }
  printf("\n");
  fflush(stdout);
}

int main(int argc, char **argv) {
  const bool dp = argc > 1 && !strcmp(argv[1], "dp");
  if (dp) { // may fault: run separately
    run_dp<0>("v_mov_b64_dpp wave_shr:1 bound_ctrl (expect lane l: 100+l-1, lane 0: 0)");
    run_dp<1>("v_mov_b64_dpp wave_shl:1 bound_ctrl (expect lane l: 100+l+1, lane 63: 0)");
    run_dp<2>("v_mov_b64_dpp row_shr:1 bound_ctrl");
    run_dp<3>("v_fmac_f64_dpp wave_shr:1 (expect -1 + 2*(100+l-1))");
    return 0;
  }
  {
    float *d, h[64];
    hipMalloc(&d, 256);
    hipLaunchKernelGGL(k_dppmax, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, 256, hipMemcpyDeviceToHost);
    printf("v_max_f32_dpp acc, acc(wave_shr:1, bound_ctrl), |d| (expect lane 0: 0, lane l: max(1000+l-1, 3l)):");
    for (int i = 0; i < 64; ++i) printf(" %g", h[i]);
    printf("\n");
  }
  run_lds<0>("ds_write_b64, 64 lanes");
  run_lds<1>("ds_write_b64, lane 63 only (EXEC)");
  run_lds<6>("ds_write_b64, 64 lanes, lane 63 -> row, others -> strip");
  run_lds<2>("ds_write2_b64, 64 lanes");
  run_lds<3>("ds_write2_b64, lane 63 only (EXEC)");
  run_lds<4>("ds_write_b128, 64 lanes");
  run_lds<5>("ds_write_b128, lane 63 only (EXEC)");
  return 0;
}
