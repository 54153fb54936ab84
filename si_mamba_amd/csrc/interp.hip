// Feature propagation of the part-segmentation head: 3-nearest-centre inverse-distance interpolation
// (reference part_segmentation/models/pointnet2_utils.py:262-305, PointNetFeaturePropagation.forward:
//  square_distance :19-38, dists.sort()[:, :, :3] :291-292, weights 1/(d+1e-8) normalised :294-296, weighted sum of
//  the three gathered feature rows :297).
//
//   * three_nn_kernel: one lane per query point, the S centres of the sample staged in LDS as (x, y, z, |c|^2);
//     squared distance in the reference's expanded form -2 p.c + |p|^2 + |c|^2 (so the clamp-free near-zero /
//     slightly negative values of coincident points behave alike), three smallest kept in registers with ties to
//     the lower centre index (the reference's sort leaves ties unspecified); writes idx (int32) and the
//     normalised weights;
//   * interp_fwd_kernel: out[b, n, :] = sum_k w_k feats[b, idx_k, :], 4 channels per lane, coalesced rows;
//   * interp_bwd_kernel: dfeats[b, s, :] = sum over the points that name centre s of w dout[b, n, :], as a
//     deterministic gather (see the kernel).
// HBM-bound: per sample N x C output floats + 3 gathered rows per point (L2-resident: S x C floats).
#include "common.h"

namespace simamba {

constexpr int kNnThreads = 256;

__global__ __launch_bounds__(kNnThreads) void three_nn_kernel(const float* __restrict__ xyz1,
                                                              const float* __restrict__ xyz2, int* __restrict__ idx,
                                                              float* __restrict__ wgt, int N, int S) {
  extern __shared__ float4 sC[];                    // [S] (x, y, z, |c|^2)
  const int b = blockIdx.y;
  const float* c = xyz2 + static_cast<size_t>(b) * S * 3;
  for (int s = threadIdx.x; s < S; s += kNnThreads) {
    const float x = c[3 * s], y = c[3 * s + 1], z = c[3 * s + 2];
    sC[s] = make_float4(x, y, z, x * x + y * y + z * z);
  }
  __syncthreads();
  const int n = blockIdx.x * kNnThreads + threadIdx.x;
  if (n >= N) return;
  const float* p = xyz1 + (static_cast<size_t>(b) * N + n) * 3;
  const float px = p[0], py = p[1], pz = p[2];
  const float pp = px * px + py * py + pz * pz;
  float d0 = 3.0e38f, d1 = 3.0e38f, d2 = 3.0e38f;
  int i0 = 0, i1 = 0, i2 = 0;
  for (int s = 0; s < S; ++s) {
    const float4 q = sC[s];
    float d = -2.f * (px * q.x + py * q.y + pz * q.z);
    d += pp;
    d += q.w;
    if (d < d2) {                                   // strict: ties keep the earlier centre
      if (d < d1) {
        d2 = d1; i2 = i1;
        if (d < d0) { d1 = d0; i1 = i0; d0 = d; i0 = s; }
        else { d1 = d; i1 = s; }
      } else { d2 = d; i2 = s; }
    }
  }
  if (S < 3) {                                      // fewer than three centres: repeat the last one with weight 0
    if (S < 2) { i1 = i0; d1 = 3.0e38f; }
    i2 = i1; d2 = 3.0e38f;
  }
  const float r0 = 1.0f / (d0 + 1e-8f), r1 = 1.0f / (d1 + 1e-8f), r2 = 1.0f / (d2 + 1e-8f);
  const float norm = r0 + r1 + r2;
  const size_t o = (static_cast<size_t>(b) * N + n) * 3;
  idx[o] = i0; idx[o + 1] = i1; idx[o + 2] = i2;
  wgt[o] = r0 / norm; wgt[o + 1] = r1 / norm; wgt[o + 2] = r2 / norm;
}

template <typename T>
__device__ __forceinline__ void ld4(const T* p, float (&v)[4]) {
  const Pack<T, 4> pk = *reinterpret_cast<const Pack<T, 4>*>(p);
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = to_f32<T>(pk.v[i]);
}

// one workgroup row of C / 4 lanes per point (several points per workgroup)
template <typename T>
__global__ __launch_bounds__(256) void interp_fwd_kernel(const T* __restrict__ feats, const int* __restrict__ idx,
                                                         const float* __restrict__ wgt, T* __restrict__ out,
                                                         long long points, int N, int S, int C) {
  const int cq = C / 4;
  const long long e = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (e >= points * cq) return;
  const long long pt = e / cq;                       // b * N + n
  const int c = static_cast<int>(e - pt * cq) * 4;
  const long long b = pt / N;
  const int* id = idx + pt * 3;
  const float* w = wgt + pt * 3;
  const T* base = feats + b * S * C + c;
  float v[4], acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    ld4<T>(base + static_cast<long long>(id[k]) * C, v);
    const float wk = w[k];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = fmaf(wk, v[i], acc[i]);
  }
  Pack<T, 4> pk;
#pragma unroll
  for (int i = 0; i < 4; ++i) pk.v[i] = from_f32<T>(acc[i]);
  *reinterpret_cast<Pack<T, 4>*>(out + pt * C + c) = pk;
}

// Backward as a gather (deterministic, no atomics): one workgroup per (sample, centre).  Phase 1 scans the
// sample's 3 N neighbour entries and compacts, in order, the (point, weight) pairs that name this centre into LDS
// (wave ballots + a 4-entry prefix over the waves); phase 2 gives every lane 4 channels and sums
// weight * dout[point] over that list with coalesced row reads.  Every dout row is read 3 times in total -- the
// float-atomic scatter this replaces took 1.55 ms at (16, 2048, 1152) against 0.1 ms for the reads.
template <typename T>
__global__ __launch_bounds__(256) void interp_bwd_kernel(const T* __restrict__ dout, const int* __restrict__ idx,
                                                         const float* __restrict__ wgt, float* __restrict__ dfeats,
                                                         int N, int S, int C) {
  extern __shared__ int sm[];
  int* sPt = sm;                                       // [N] point of each match
  float* sWt = reinterpret_cast<float*>(sm + N);       // [N] its weight
  __shared__ int sWave[4];
  __shared__ int sBase;
  const int s = blockIdx.x, b = blockIdx.y;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int* id = idx + static_cast<size_t>(b) * N * 3;
  const float* w = wgt + static_cast<size_t>(b) * N * 3;
  if (tid == 0) sBase = 0;
  __syncthreads();
  for (int base = 0; base < 3 * N; base += 256) {
    const int e = base + tid;
    const bool match = e < 3 * N && id[e] == s;
    const unsigned long long bal = __ballot(match);
    if (lane == 0) sWave[wave] = __popcll(bal);
    __syncthreads();
    int off = sBase;
    for (int k = 0; k < wave; ++k) off += sWave[k];
    if (match) {
      const int pos = off + __popcll(bal & ((1ull << lane) - 1ull));
      sPt[pos] = e / 3;
      sWt[pos] = w[e];
    }
    __syncthreads();
    if (tid == 0) sBase += sWave[0] + sWave[1] + sWave[2] + sWave[3];
    __syncthreads();
  }
  const int count = sBase;
  const T* drow = dout + static_cast<size_t>(b) * N * C;
  float* orow = dfeats + (static_cast<size_t>(b) * S + s) * C;
  for (int c = 4 * tid; c < C; c += 4 * 256) {
    float acc[4] = {0.f, 0.f, 0.f, 0.f}, d[4];
    for (int m = 0; m < count; ++m) {
      ld4<T>(drow + static_cast<size_t>(sPt[m]) * C + c, d);
      const float wk = sWt[m];
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = fmaf(wk, d[i], acc[i]);
    }
    *reinterpret_cast<float4*>(orow + c) = make_float4(acc[0], acc[1], acc[2], acc[3]);
  }
}

}  // namespace simamba

using namespace simamba;

extern "C" int simamba_three_nn(const float* xyz1, const float* xyz2, int* idx, float* weight, int batch, int N, int S,
                                void* stream) {
  if (batch < 0 || N < 0 || S < 1 || S > 8192 || batch > 65535) return SIMAMBA_E_SHAPE;
  if (batch == 0 || N == 0) return SIMAMBA_OK;
  if (!xyz1 || !xyz2 || !idx || !weight) return SIMAMBA_E_NULLPTR;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const size_t smem = sizeof(float4) * S;
  if (smem > 64 * 1024) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(three_nn_kernel),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(smem));
    if (e != hipSuccess) return static_cast<int>(e);
  }
  hipLaunchKernelGGL(three_nn_kernel, dim3((N + kNnThreads - 1) / kNnThreads, batch), dim3(kNnThreads), smem, s, xyz1,
                     xyz2, idx, weight, N, S);
  return static_cast<int>(hipGetLastError());
}

extern "C" int simamba_three_interpolate_fwd(const void* feats, const int* idx, const float* weight, void* out,
                                             int batch, int N, int S, int C, int io_dtype, void* stream) {
  if (io_dtype != SIMAMBA_F32 && io_dtype != SIMAMBA_BF16) return SIMAMBA_E_DTYPE;
  if (batch < 0 || N < 0 || S < 1 || C < 4 || (C % 4) != 0) return SIMAMBA_E_SHAPE;
  if (batch == 0 || N == 0) return SIMAMBA_OK;
  if (!feats || !idx || !weight || !out) return SIMAMBA_E_NULLPTR;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const long long points = static_cast<long long>(batch) * N;
  const long long total = points * (C / 4);
  const dim3 grid(static_cast<unsigned>((total + 255) / 256));
  if (io_dtype == SIMAMBA_F32)
    hipLaunchKernelGGL(interp_fwd_kernel<float>, grid, dim3(256), 0, s, static_cast<const float*>(feats), idx, weight,
                       static_cast<float*>(out), points, N, S, C);
  else
    hipLaunchKernelGGL(interp_fwd_kernel<bf16_t>, grid, dim3(256), 0, s, static_cast<const bf16_t*>(feats), idx, weight,
                       static_cast<bf16_t*>(out), points, N, S, C);
  return static_cast<int>(hipGetLastError());
}

extern "C" int simamba_three_interpolate_bwd(const void* dout, const int* idx, const float* weight, float* dfeats,
                                             int batch, int N, int S, int C, int io_dtype, void* stream) {
  if (io_dtype != SIMAMBA_F32 && io_dtype != SIMAMBA_BF16) return SIMAMBA_E_DTYPE;
  if (batch < 0 || N < 0 || N > 8192 || S < 1 || S > 65535 || C < 4 || (C % 4) != 0 || batch > 65535)
    return SIMAMBA_E_SHAPE;
  if (batch == 0) return SIMAMBA_OK;
  if (!dfeats) return SIMAMBA_E_NULLPTR;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (N == 0) {
    const hipError_t e = hipMemsetAsync(dfeats, 0, sizeof(float) * static_cast<size_t>(batch) * S * C, s);
    return static_cast<int>(e);
  }
  if (!dout || !idx || !weight) return SIMAMBA_E_NULLPTR;
  const size_t smem = 8 * static_cast<size_t>(N);
  if (io_dtype == SIMAMBA_F32)
    hipLaunchKernelGGL(interp_bwd_kernel<float>, dim3(S, batch), dim3(256), smem, s, static_cast<const float*>(dout), idx,
                       weight, dfeats, N, S, C);
  else
    hipLaunchKernelGGL(interp_bwd_kernel<bf16_t>, dim3(S, batch), dim3(256), smem, s, static_cast<const bf16_t*>(dout),
                       idx, weight, dfeats, N, S, C);
  return static_cast<int>(hipGetLastError());
}
