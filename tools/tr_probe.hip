// What ds_read_b64_tr_b16 hands each lane (gfx950): a [K rows][N cols] image of 16-bit values, value = 100 * row + col;
// lane 4q + p of a 16-lane group supplies the address of (row r0 + q, cols c0 + 4p ..); prints what every lane receives.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(short* out, int pitch) {
  __shared__ short lds[64 * 72];
  for (int i = threadIdx.x; i < 64 * 72; i += 64) lds[i] = (short)(100 * (i / pitch) + (i % pitch));
  __syncthreads();
  const int l = threadIdx.x, g = l >> 4, q = (l >> 2) & 3, p = l & 3;
  const int r0 = 8 * (g >> 1), c0 = 16 * (g & 1);
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(lds + (r0 + q) * pitch + c0 + 4 * p));
  for (int e = 0; e < 4; ++e) out[l * 4 + e] = v[e];
}
int main() {
  short* d; hipMalloc(&d, 64 * 4 * 2);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, 72);
  short h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) printf("lane %2d: %5d %5d %5d %5d\n", l, h[4*l], h[4*l+1], h[4*l+2], h[4*l+3]);
  return 0;
}
