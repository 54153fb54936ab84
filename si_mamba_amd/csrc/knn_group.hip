// k-nearest-neighbour grouping of the tokeniser (SURVEY.md section 8f "next" row 1): for every patch centre the
// K = group_size nearest points of its cloud.  Replaces pytorch3d.ops.knn_points(center, xyz, K, return_sorted=False)
// as called at the reference's models/point_mamba.py:96 (and part_segmentation/models/pt_mamba.py:179).
//
// One wave per centre.  Lane l keeps the squared distances (direct differences, like pytorch3d; this file is built
// with -ffp-contract=off) of points l, l + 64, ... in kPer registers; each of the K rounds takes the wave-wide
// (value, index) minimum by DPP/shuffle butterfly, the owning lane retires that point.  The state loop is
// unrolled so the registers are statically indexed.  Output: indices in ascending distance, ties to the lower
// point index (the reference's order is unspecified: return_sorted=False, and every consumer is order-invariant).
#include "common.h"

namespace simamba {

constexpr int kKnnWaves = 4;     // centres per workgroup

template <int kPer>
__global__ __launch_bounds__(64 * kKnnWaves) void knn_group_kernel(const float* __restrict__ pts,
                                                                   const float* __restrict__ centers,
                                                                   long long* __restrict__ idx, int N, int G, int K) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int g = blockIdx.x * kKnnWaves + wave;
  const int b = blockIdx.y;
  if (g >= G) return;
  const float* P = pts + static_cast<size_t>(b) * N * 3;
  const float* c = centers + (static_cast<size_t>(b) * G + g) * 3;
  const float cx = c[0], cy = c[1], cz = c[2];
  float d[kPer];
#pragma unroll
  for (int k = 0; k < kPer; ++k) {
    const int i = lane + 64 * k;
    if (i < N) {
      const float dx = P[3 * i] - cx, dy = P[3 * i + 1] - cy, dz = P[3 * i + 2] - cz;
      d[k] = (dx * dx + dy * dy) + dz * dz;
    } else {
      d[k] = __builtin_inff();
    }
  }
  long long* out = idx + (static_cast<size_t>(b) * G + g) * K;
  for (int r = 0; r < K; ++r) {
    float bv = d[0];
    int bk = 0;
#pragma unroll
    for (int k = 1; k < kPer; ++k)
      if (d[k] < bv) { bv = d[k]; bk = k; }            // strict: lower k (= lower index within the lane) wins ties
    int bi = lane + 64 * bk;
    float v = bv;
    int i = bi;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      const float ov = __shfl_xor(v, off, 64);
      const int oi = __shfl_xor(i, off, 64);
      if (ov < v || (ov == v && oi < i)) { v = ov; i = oi; }
    }
    if (lane == 0) out[r] = i;
    if (i == bi) {                                      // this lane owned the winner: retire it
#pragma unroll
      for (int k = 0; k < kPer; ++k)
        if (k == bk) d[k] = __builtin_inff();
    }
  }
}

}  // namespace simamba

using namespace simamba;

extern "C" int simamba_knn_group(const float* points, const float* centers, long long* idx, int batch, int N, int G,
                                 int K, void* stream) {
  if (batch < 0 || N < 1 || G < 0 || K < 1 || K > N || N > 8192 || batch > 65535) return SIMAMBA_E_SHAPE;
  if (batch == 0 || G == 0) return SIMAMBA_OK;
  if (!points || !centers || !idx) return SIMAMBA_E_NULLPTR;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const dim3 grid((G + kKnnWaves - 1) / kKnnWaves, batch), block(64 * kKnnWaves);
  if (N <= 1024) hipLaunchKernelGGL(knn_group_kernel<16>, grid, block, 0, s, points, centers, idx, N, G, K);
  else if (N <= 2048) hipLaunchKernelGGL(knn_group_kernel<32>, grid, block, 0, s, points, centers, idx, N, G, K);
  else if (N <= 4096) hipLaunchKernelGGL(knn_group_kernel<64>, grid, block, 0, s, points, centers, idx, N, G, K);
  else hipLaunchKernelGGL(knn_group_kernel<128>, grid, block, 0, s, points, centers, idx, N, G, K);
  return static_cast<int>(hipGetLastError());
}
