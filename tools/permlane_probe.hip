// Probe of v_permlane32_swap / v_permlane16_swap / DPP lane semantics on gfx950 (diagnostic tool).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(unsigned* out) {
  unsigned lane = threadIdx.x;
  unsigned x = 1000 + lane, y = 2000 + lane;
  auto r = __builtin_amdgcn_permlane32_swap(x, y, false, false);
  out[lane] = r[0]; out[64 + lane] = r[1];
  auto q = __builtin_amdgcn_permlane16_swap(x, y, false, false);
  out[128 + lane] = q[0]; out[192 + lane] = q[1];
  out[256 + lane] = __builtin_amdgcn_update_dpp(7777, (int)x, 0x101, 0xf, 0xf, false);  // row_shl:1
  out[320 + lane] = __builtin_amdgcn_update_dpp(7777, (int)x, 0x111, 0xf, 0xf, false);  // row_shr:1
}
int main() {
  unsigned* d; hipMalloc(&d, 384 * 4);
  probe<<<1, 64>>>(d);
  unsigned h[384]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char* names[] = {"swap32.vdst", "swap32.src0", "swap16.vdst", "swap16.src0", "row_shl1", "row_shr1"};
  for (int k = 0; k < 6; ++k) { printf("%s:", names[k]); for (int i = 0; i < 64; ++i) printf(" %u", h[k * 64 + i]); printf("\n"); }
  return 0;
}
