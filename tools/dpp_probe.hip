// Probe: semantics of DPP wave_shr:1 / wave_shl:1 on gfx950 (used by the fast sweep).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int *out) {
  int lane = threadIdx.x;
  int v = lane + 100;
  int shr = __builtin_amdgcn_update_dpp(v, v, 0x138, 0xf, 0xf, false);
  int shl = __builtin_amdgcn_update_dpp(v, v, 0x130, 0xf, 0xf, false);
  int up = __shfl_up(v, 1, 64);
  int dn = __shfl_down(v, 1, 64);
  out[lane] = shr; out[64 + lane] = shl; out[128 + lane] = up; out[192 + lane] = dn;
}
int main() {
  int *d; hipMalloc(&d, 256 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  int h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char *names[4] = {"wave_shr1", "wave_shl1", "shfl_up", "shfl_down"};
  for (int r = 0; r < 4; ++r) { printf("%s:", names[r]); for (int i = 0; i < 64; ++i) printf(" %d", h[r * 64 + i] - 100); printf("\n"); }
  return 0;
}
