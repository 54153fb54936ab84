// Issue rate of v_mfma_f32_32x32x2_f32 as a function of the number of independent accumulator chains a wave
// interleaves and of the waves per SIMD (tuning tool).  hipcc --offload-arch=gfx950 -O3 tools/mfma_probe.hip -o tools/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int kChains>
__global__ __launch_bounds__(64) void chain_kernel(float* out, int iters, float a, float b) {
  f32x16 acc[kChains];
#pragma unroll
  for (int c = 0; c < kChains; ++c)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
  const float av = a + threadIdx.x, bv = b - threadIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 16; ++k)
#pragma unroll
      for (int c = 0; c < kChains; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[c], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < kChains; ++c)
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[c][i];
  out[blockIdx.x * 64 + threadIdx.x] = s;
}

template <int kChains>
static int run(float* out, int waves_per_simd) {
  const int iters = 2000;
  const int grid = 256 * 4 * waves_per_simd;      // one-wave workgroups: 4 SIMDs x CUs x waves per SIMD
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(chain_kernel<kChains>, dim3(grid), dim3(64), 0, 0, out, 10, 1.f, 2.f);
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL(chain_kernel<kChains>, dim3(grid), dim3(64), 0, 0, out, iters, 1.f, 2.f);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double mfma_per_simd = static_cast<double>(iters) * 16 * kChains * waves_per_simd;
  const double tf = mfma_per_simd * 1024 * 4096 / (ms * 1e-3) * 1e-12;
  printf("chains/wave %d  waves/SIMD %d : %8.1f us  %6.1f ns per MFMA per SIMD  %6.1f TF/s (peak 157.3)\n", kChains,
         waves_per_simd, ms * 1e3, ms * 1e6 / mfma_per_simd, tf);
  return 0;
}

int main() {
  float* out;
  CK(hipMalloc(&out, 256 * 4 * 8 * 64 * 4));
  for (int w = 1; w <= 2; ++w) {
    if (run<1>(out, w)) return 1;
    if (run<2>(out, w)) return 1;
    if (run<4>(out, w)) return 1;
  }
  return 0;
}
