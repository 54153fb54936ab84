// Farthest-point sampling for gfx950 -- first step of the tokeniser that feeds the hot path
// ("next" row 1).
//
// Replaces pytorch3d.ops.sample_farthest_points as called at the reference's models/point_mamba.py:93
// (K = num_group centres per cloud, start at point 0).  One workgroup per cloud: the cloud and the running
// minimum distances live in registers (4 points per lane for N = 1024), each of the K rounds is
//   broadcast the last pick through LDS -> update min-distance -> (value, index) arg-max by DPP wave
//   reduction -> 4-entry LDS combine,
// i.e. two workgroup barriers per round and no global traffic besides the initial 12 N bytes.
// Distances are accumulated exactly like the torch formulation ((dx^2 + dy^2) + dz^2, no FMA contraction;
// this file is built with -ffp-contract=off) and ties go to the lower index, so the picks are bit-identical
// to the oracle's.
#include "common.h"

namespace simamba {

constexpr int kFpsThreads = 256;
constexpr int kFpsMaxPer = 16;     // points per lane: N <= 4096

__device__ __forceinline__ void argmax_combine(float& v, int& i, float ov, int oi) {
  if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
}

__global__ __launch_bounds__(kFpsThreads) void fps_kernel(const float* __restrict__ pts, long long* __restrict__ idx,
                                                          float* __restrict__ centers, int N, int K) {
  __shared__ float sCur[3];
  __shared__ float sVal[kFpsThreads / 64];
  __shared__ int sIdx[kFpsThreads / 64];
  const float* P = pts + static_cast<size_t>(blockIdx.x) * N * 3;
  const int tid = threadIdx.x;
  float px[kFpsMaxPer], py[kFpsMaxPer], pz[kFpsMaxPer], md[kFpsMaxPer];
#pragma unroll
  for (int k = 0; k < kFpsMaxPer; ++k) {
    const int i = tid + k * kFpsThreads;
    const bool ok = i < N;
    px[k] = ok ? P[3 * i] : 0.f;
    py[k] = ok ? P[3 * i + 1] : 0.f;
    pz[k] = ok ? P[3 * i + 2] : 0.f;
    md[k] = ok ? __builtin_inff() : -1.f;    // padding never wins the arg-max
  }
  int cur = 0;
  for (int r = 0; r < K; ++r) {
    if (tid == 0) {
      idx[static_cast<size_t>(blockIdx.x) * K + r] = cur;
      const float cx = P[3 * cur], cy = P[3 * cur + 1], cz = P[3 * cur + 2];
      sCur[0] = cx; sCur[1] = cy; sCur[2] = cz;
      if (centers) {
        float* c = centers + (static_cast<size_t>(blockIdx.x) * K + r) * 3;
        c[0] = cx; c[1] = cy; c[2] = cz;
      }
    }
    __syncthreads();
    const float cx = sCur[0], cy = sCur[1], cz = sCur[2];
    float bv = -2.f;
    int bi = 0x7fffffff;
#pragma unroll
    for (int k = 0; k < kFpsMaxPer; ++k) {
      if (k * kFpsThreads < N) {
        const float dx = px[k] - cx, dy = py[k] - cy, dz = pz[k] - cz;
        const float d = (dx * dx + dy * dy) + dz * dz;
        md[k] = fminf(md[k], d);
        argmax_combine(bv, bi, md[k], tid + k * kFpsThreads);
      }
    }
    // wave arg-max (butterfly over 64 lanes)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const float ov = __shfl_xor(bv, off);
      const int oi = __shfl_xor(bi, off);
      argmax_combine(bv, bi, ov, oi);
    }
    if ((tid & 63) == 0) { sVal[tid >> 6] = bv; sIdx[tid >> 6] = bi; }
    __syncthreads();
    bv = sVal[0]; bi = sIdx[0];
#pragma unroll
    for (int w = 1; w < kFpsThreads / 64; ++w) argmax_combine(bv, bi, sVal[w], sIdx[w]);
    cur = bi;
  }
}

}  // namespace simamba

using namespace simamba;

extern "C" int simamba_farthest_point_sample(const float* points, long long* idx, float* centers, int B, int N, int K,
                                             void* stream) {
  if (B < 0 || N <= 0 || K < 0 || K > N || N > kFpsThreads * kFpsMaxPer) return SIMAMBA_E_SHAPE;
  if (B == 0 || K == 0) return SIMAMBA_OK;
  if (!points || !idx) return SIMAMBA_E_NULLPTR;
  hipLaunchKernelGGL(fps_kernel, dim3(B), dim3(kFpsThreads), 0, static_cast<hipStream_t>(stream), points, idx, centers,
                     N, K);
  return static_cast<int>(hipGetLastError());
}
