// Chamfer distance between small point sets: the loss of the MAE pre-training step
// (reference models/point_mamba.py:2950 / :3203: pytorch3d.loss.chamfer_distance(pred, gt, batch_reduction=None)
// on (B*M, 32, 3) rebuilt / ground-truth patches; pytorch3d semantics: squared L2 to the nearest neighbour,
// mean over the points of each set, sum of the two directions).
//
// One wave per pair of sets (n, m <= 64 points): lane i keeps pred_i and sweeps the m ground-truth points staged
// in LDS, then lane j keeps gt_j and sweeps the n predictions; wave reductions by DPP/shuffle.  The arg-mins are
// stored for the backward, which is the analytic gradient w.r.t. the predictions
//   dpred_i = g * ( 2/n (p_i - gt[a(i)]) + 2/m sum_{j : b(j) = i} (p_i - gt_j) ).
#include "common.h"

namespace simamba {

constexpr int kChWaves = 4;     // pairs per workgroup

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

__global__ __launch_bounds__(64 * kChWaves) void chamfer_fwd_kernel(const float* __restrict__ pred,
                                                                     const float* __restrict__ gt,
                                                                     float* __restrict__ dist,
                                                                     unsigned char* __restrict__ idx1,
                                                                     unsigned char* __restrict__ idx2, long long pairs,
                                                                     int n, int m) {
  __shared__ float sP[kChWaves][64 * 3], sG[kChWaves][64 * 3];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long long pr = static_cast<long long>(blockIdx.x) * kChWaves + wave;
  if (pr >= pairs) return;                       // whole wave; no workgroup barrier below
  const float* P = pred + pr * n * 3;
  const float* G = gt + pr * m * 3;
  for (int e = lane; e < n * 3; e += 64) sP[wave][e] = P[e];
  for (int e = lane; e < m * 3; e += 64) sG[wave][e] = G[e];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  float s1 = 0.f, s2 = 0.f;
  if (lane < n) {
    const float x = sP[wave][3 * lane], y = sP[wave][3 * lane + 1], z = sP[wave][3 * lane + 2];
    float best = 3.0e38f; int bi = 0;
    for (int j = 0; j < m; ++j) {
      const float dx = x - sG[wave][3 * j], dy = y - sG[wave][3 * j + 1], dz = z - sG[wave][3 * j + 2];
      const float d = dx * dx + dy * dy + dz * dz;
      if (d < best) { best = d; bi = j; }
    }
    s1 = best;
    idx1[pr * n + lane] = static_cast<unsigned char>(bi);
  }
  if (lane < m) {
    const float x = sG[wave][3 * lane], y = sG[wave][3 * lane + 1], z = sG[wave][3 * lane + 2];
    float best = 3.0e38f; int bi = 0;
    for (int i = 0; i < n; ++i) {
      const float dx = x - sP[wave][3 * i], dy = y - sP[wave][3 * i + 1], dz = z - sP[wave][3 * i + 2];
      const float d = dx * dx + dy * dy + dz * dz;
      if (d < best) { best = d; bi = i; }
    }
    s2 = best;
    idx2[pr * m + lane] = static_cast<unsigned char>(bi);
  }
  const float t1 = wave_sum(s1), t2 = wave_sum(s2);
  if (lane == 0) dist[pr] = t1 / static_cast<float>(n) + t2 / static_cast<float>(m);
}

__global__ __launch_bounds__(64 * kChWaves) void chamfer_bwd_kernel(const float* __restrict__ pred,
                                                                     const float* __restrict__ gt,
                                                                     const float* __restrict__ ddist,
                                                                     const unsigned char* __restrict__ idx1,
                                                                     const unsigned char* __restrict__ idx2,
                                                                     float* __restrict__ dpred, long long pairs, int n,
                                                                     int m) {
  __shared__ float sG[kChWaves][64 * 3];
  __shared__ unsigned char sI2[kChWaves][64];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long long pr = static_cast<long long>(blockIdx.x) * kChWaves + wave;
  if (pr >= pairs) return;
  const float* G = gt + pr * m * 3;
  for (int e = lane; e < m * 3; e += 64) sG[wave][e] = G[e];
  if (lane < m) sI2[wave][lane] = idx2[pr * m + lane];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  if (lane >= n) return;
  const float g = ddist[pr];
  const float* p = pred + (pr * n + lane) * 3;
  const float x = p[0], y = p[1], z = p[2];
  const int a = idx1[pr * n + lane];
  const float k1 = 2.f * g / static_cast<float>(n), k2 = 2.f * g / static_cast<float>(m);
  float gx = k1 * (x - sG[wave][3 * a]), gy = k1 * (y - sG[wave][3 * a + 1]), gz = k1 * (z - sG[wave][3 * a + 2]);
  for (int j = 0; j < m; ++j) {
    if (sI2[wave][j] == lane) {
      gx += k2 * (x - sG[wave][3 * j]); gy += k2 * (y - sG[wave][3 * j + 1]); gz += k2 * (z - sG[wave][3 * j + 2]);
    }
  }
  float* o = dpred + (pr * n + lane) * 3;
  o[0] = gx; o[1] = gy; o[2] = gz;
}

}  // namespace simamba

using namespace simamba;

extern "C" int simamba_chamfer_fwd(const float* pred, const float* gt, float* dist, unsigned char* idx1,
                                   unsigned char* idx2, long long pairs, int n, int m, void* stream) {
  if (pairs < 0 || n < 1 || m < 1 || n > 64 || m > 64) return SIMAMBA_E_SHAPE;
  if (pairs == 0) return SIMAMBA_OK;
  if (!pred || !gt || !dist || !idx1 || !idx2) return SIMAMBA_E_NULLPTR;
  const long long grid = (pairs + kChWaves - 1) / kChWaves;
  if (grid > 0x7fffffffll) return SIMAMBA_E_SHAPE;
  hipLaunchKernelGGL(chamfer_fwd_kernel, dim3(static_cast<unsigned>(grid)), dim3(64 * kChWaves), 0,
                     static_cast<hipStream_t>(stream), pred, gt, dist, idx1, idx2, pairs, n, m);
  return static_cast<int>(hipGetLastError());
}

extern "C" int simamba_chamfer_bwd(const float* pred, const float* gt, const float* ddist, const unsigned char* idx1,
                                   const unsigned char* idx2, float* dpred, long long pairs, int n, int m,
                                   void* stream) {
  if (pairs < 0 || n < 1 || m < 1 || n > 64 || m > 64) return SIMAMBA_E_SHAPE;
  if (pairs == 0) return SIMAMBA_OK;
  if (!pred || !gt || !ddist || !idx1 || !idx2 || !dpred) return SIMAMBA_E_NULLPTR;
  const long long grid = (pairs + kChWaves - 1) / kChWaves;
  if (grid > 0x7fffffffll) return SIMAMBA_E_SHAPE;
  hipLaunchKernelGGL(chamfer_bwd_kernel, dim3(static_cast<unsigned>(grid)), dim3(64 * kChWaves), 0,
                     static_cast<hipStream_t>(stream), pred, gt, ddist, idx1, idx2, dpred, pairs, n, m);
  return static_cast<int>(hipGetLastError());
}
