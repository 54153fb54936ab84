// BatchNorm1d (+ per-group additive term) + ReLU, and the max over a group's points, for the patch encoder
// (reference models/point_mamba.py:46-73: Conv1d - BatchNorm1d - ReLU - Conv1d, max over the n points of a patch,
// concat [global, local] - Conv1d - BatchNorm1d - ReLU - Conv1d, max).  HBM-bound streaming kernels over the
// token-major (rows, C) activations the 1x1 convolutions produce as GEMMs:
//
//   * forward   : pass 1 per-channel shifted sums (one read), finalize (mean, 1/std, running statistics),
//                 pass 2 y = relu((x + g - mean) * invstd * w + b) (one read, one write): 3 passes instead of the 5
//                 of stock batch_norm + relu;
//   * backward  : pass 1 sums of dy*mask and dy*mask*xhat (two reads), finalize (dweight, dbias),
//                 pass 2 dx (two reads, one write) and the per-group sum of dx (the gradient of the additive
//                 term): 5 passes instead of 8;
//   * the additive term g[r / group][c] is the "global feature" half of the encoder's second convolution:
//     cat([max_n f, f]) @ W^T == f @ W_local^T + (max_n f) @ W_global^T, the second product having one row per
//     patch instead of one per point -- the 537 MB concat tensor and a quarter of the GEMM flops disappear.
//
// A thread owns 4 consecutive channels (one 16-byte access fp32 / 8-byte bf16) for the whole kernel, so the
// per-channel constants sit in registers; a workgroup is TPR = C / 4 lanes per row x RL row lanes and walks a
// chunk of kBnChunk rows.  Per-workgroup partial sums go to a (grid, 2, C) buffer and are combined in fp64 by
// a one-workgroup finalize kernel: deterministic, no atomics.  Sums are taken about the first row's value
// (shifted data), which removes the cancellation in E[x^2] - E[x]^2.
#include "common.h"

namespace simamba {

constexpr int kBnChunk = 256;        // rows per workgroup (a multiple of the group size)
constexpr int kBnMaxThreads = 256;

struct BnArgs {
  const void* x;
  const float* gterm;
  const float* weight;
  const float* bias;
  const float* mean;
  const float* invstd;
  const void* dy;
  void* y;            // forward output / backward dx
  float* dgterm;
  float* partial;     // (grid, 2, C)
  long long rows;
  long long ld;       // row stride (elements) of x / y / dy / dx: channel slices of a wider tensor are processed in place
  int C, group, dgroup, training;
};

template <typename T>
__device__ __forceinline__ void load_c4(const T* p, float (&v)[4]) {
  const Pack<T, 4> pk = *reinterpret_cast<const Pack<T, 4>*>(p);
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = to_f32<T>(pk.v[i]);
}
template <typename T>
__device__ __forceinline__ void store_c4(T* p, const float (&v)[4]) {
  Pack<T, 4> pk;
#pragma unroll
  for (int i = 0; i < 4; ++i) pk.v[i] = from_f32<T>(v[i]);
  *reinterpret_cast<Pack<T, 4>*>(p) = pk;
}

// combine the RL row lanes of a workgroup: s0 / s1 hold this thread's two 4-channel sums
__device__ __forceinline__ void reduce_row_lanes(float (&s0)[4], float (&s1)[4], float* sm, int tpr, int rl,
                                                 int cl, int rlane, float* out, int C) {
  // sm: [rl][2][C]
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    sm[(rlane * 2 + 0) * C + 4 * cl + i] = s0[i];
    sm[(rlane * 2 + 1) * C + 4 * cl + i] = s1[i];
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 2 * C; e += blockDim.x) {
    float acc = 0.f;
    for (int r = 0; r < rl; ++r) acc += sm[r * 2 * C + e];
    out[e] = acc;
  }
}

// ---- forward pass 1: shifted sums -----------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(kBnMaxThreads) void bn_stats_kernel(BnArgs p) {
  extern __shared__ float sm[];
  const int C = p.C, tpr = C / 4, rl = blockDim.x / tpr;
  const int cl = threadIdx.x % tpr, rlane = threadIdx.x / tpr;
  const T* __restrict__ x = static_cast<const T*>(p.x);
  float K[4];
  load_c4<T>(x + 4 * cl, K);                                   // shift: the first row (same for every workgroup)
  if (p.gterm) {
#pragma unroll
    for (int i = 0; i < 4; ++i) K[i] += p.gterm[4 * cl + i];
  }
  float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
  const long long r0 = static_cast<long long>(blockIdx.x) * kBnChunk;
  const long long r1 = min(r0 + kBnChunk, p.rows);
#pragma unroll 4
  for (long long r = r0 + rlane; r < r1; r += rl) {
    float v[4];
    load_c4<T>(x + r * p.ld + 4 * cl, v);
    if (p.gterm) {
      const float4 g = *reinterpret_cast<const float4*>(p.gterm + (r / p.group) * C + 4 * cl);
      v[0] += g.x; v[1] += g.y; v[2] += g.z; v[3] += g.w;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float d = v[i] - K[i];
      s1[i] += d;
      s2[i] = fmaf(d, d, s2[i]);
    }
  }
  reduce_row_lanes(s1, s2, sm, tpr, rl, cl, rlane, p.partial + static_cast<size_t>(blockIdx.x) * 2 * C, C);
}

// finalize: fp64 combination of the per-workgroup partials.  A workgroup of 256 threads owns kFinCh channels:
// thread (pl, c) sums partials pl, pl + kFinPl, ... of channel c, the kFinPl lanes are combined through LDS.
constexpr int kFinCh = 16, kFinPl = 16;

__device__ __forceinline__ void finalize_sums(const float* __restrict__ partial, int grid, int C, int c, int pl,
                                              bool valid, double (*sm)[kFinPl][kFinCh], double& S1, double& S2) {
  double a1 = 0.0, a2 = 0.0;
  if (valid) {
#pragma unroll 4
    for (int g = pl; g < grid; g += kFinPl) {
      a1 += partial[(static_cast<size_t>(g) * 2 + 0) * C + c];
      a2 += partial[(static_cast<size_t>(g) * 2 + 1) * C + c];
    }
  }
  const int cl = threadIdx.x % kFinCh;
  sm[0][pl][cl] = a1;
  sm[1][pl][cl] = a2;
  __syncthreads();
  S1 = 0.0; S2 = 0.0;
  for (int k = 0; k < kFinPl; ++k) { S1 += sm[0][k][cl]; S2 += sm[1][k][cl]; }
}

template <typename T>
__global__ __launch_bounds__(kFinCh * kFinPl) void bn_stats_finalize_kernel(
    const void* x, const float* gterm, const float* partial, int grid, long long rows, int C, float eps,
    float momentum, float* mean, float* invstd, float* running_mean, float* running_var) {
  __shared__ double sm[2][kFinPl][kFinCh];
  const int c = blockIdx.x * kFinCh + threadIdx.x % kFinCh, pl = threadIdx.x / kFinCh;
  double S1, S2;
  finalize_sums(partial, grid, C, c, pl, c < C, sm, S1, S2);
  if (c >= C || pl != 0) return;
  float K = to_f32<T>(static_cast<const T*>(x)[c]);
  if (gterm) K += gterm[c];
  const double n = static_cast<double>(rows);
  const double m1 = S1 / n;
  double var = S2 / n - m1 * m1;
  var = var < 0.0 ? 0.0 : var;
  const double mu = static_cast<double>(K) + m1;
  mean[c] = static_cast<float>(mu);
  invstd[c] = static_cast<float>(1.0 / sqrt(var + static_cast<double>(eps)));
  if (running_mean) running_mean[c] = static_cast<float>((1.0 - momentum) * running_mean[c] + momentum * mu);
  if (running_var) {
    const double unbiased = rows > 1 ? var * n / (n - 1.0) : var;
    running_var[c] = static_cast<float>((1.0 - momentum) * running_var[c] + momentum * unbiased);
  }
}

// eval mode: statistics are the running ones
__global__ void bn_eval_stats_kernel(const float* running_mean, const float* running_var, float eps, int C,
                                     float* mean, float* invstd) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  mean[c] = running_mean[c];
  invstd[c] = 1.0f / sqrtf(running_var[c] + eps);
}

// ---- forward pass 2: normalise + ReLU ---------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(kBnMaxThreads) void bn_relu_apply_kernel(BnArgs p) {
  const int C = p.C, tpr = C / 4, rl = blockDim.x / tpr;
  const int cl = threadIdx.x % tpr, rlane = threadIdx.x / tpr;
  const T* __restrict__ x = static_cast<const T*>(p.x);
  T* __restrict__ y = static_cast<T*>(p.y);
  float scale[4], shift[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = 4 * cl + i;
    scale[i] = p.invstd[c] * (p.weight ? p.weight[c] : 1.f);
    shift[i] = (p.bias ? p.bias[c] : 0.f) - p.mean[c] * scale[i];
  }
  const long long r0 = static_cast<long long>(blockIdx.x) * kBnChunk;
  const long long r1 = min(r0 + kBnChunk, p.rows);
#pragma unroll 4
  for (long long r = r0 + rlane; r < r1; r += rl) {
    float v[4];
    load_c4<T>(x + r * p.ld + 4 * cl, v);
    if (p.gterm) {
      const float4 g = *reinterpret_cast<const float4*>(p.gterm + (r / p.group) * C + 4 * cl);
      v[0] += g.x; v[1] += g.y; v[2] += g.z; v[3] += g.w;
    }
    float o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = fmaxf(fmaf(v[i], scale[i], shift[i]), 0.f);
    store_c4<T>(y + r * p.ld + 4 * cl, o);
  }
}

// ---- backward pass 1: sum dy*mask, sum dy*mask*xhat ---------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(kBnMaxThreads) void bn_relu_bwd_reduce_kernel(BnArgs p) {
  extern __shared__ float sm[];
  const int C = p.C, tpr = C / 4, rl = blockDim.x / tpr;
  const int cl = threadIdx.x % tpr, rlane = threadIdx.x / tpr;
  const T* __restrict__ x = static_cast<const T*>(p.x);
  const T* __restrict__ dy = static_cast<const T*>(p.dy);
  float scale[4], shift[4], mu[4], is[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = 4 * cl + i;
    mu[i] = p.mean[c]; is[i] = p.invstd[c];
    scale[i] = is[i] * (p.weight ? p.weight[c] : 1.f);
    shift[i] = (p.bias ? p.bias[c] : 0.f) - mu[i] * scale[i];
  }
  float sb[4] = {0.f, 0.f, 0.f, 0.f}, sw[4] = {0.f, 0.f, 0.f, 0.f};
  const long long r0 = static_cast<long long>(blockIdx.x) * kBnChunk;
  const long long r1 = min(r0 + kBnChunk, p.rows);
#pragma unroll 4
  for (long long r = r0 + rlane; r < r1; r += rl) {
    float v[4], d[4];
    load_c4<T>(x + r * p.ld + 4 * cl, v);
    load_c4<T>(dy + r * p.ld + 4 * cl, d);
    if (p.gterm) {
      const float4 g = *reinterpret_cast<const float4*>(p.gterm + (r / p.group) * C + 4 * cl);
      v[0] += g.x; v[1] += g.y; v[2] += g.z; v[3] += g.w;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float dr = fmaf(v[i], scale[i], shift[i]) > 0.f ? d[i] : 0.f;
      sb[i] += dr;
      sw[i] = fmaf(dr, (v[i] - mu[i]) * is[i], sw[i]);
    }
  }
  reduce_row_lanes(sb, sw, sm, tpr, rl, cl, rlane, p.partial + static_cast<size_t>(blockIdx.x) * 2 * C, C);
}

__global__ __launch_bounds__(kFinCh * kFinPl) void bn_bwd_finalize_kernel(const float* partial, int grid, int C,
                                                                             float* dweight, float* dbias) {
  __shared__ double sm[2][kFinPl][kFinCh];
  const int c = blockIdx.x * kFinCh + threadIdx.x % kFinCh, pl = threadIdx.x / kFinCh;
  double Sb, Sw;
  finalize_sums(partial, grid, C, c, pl, c < C, sm, Sb, Sw);
  if (c >= C || pl != 0) return;
  dbias[c] = static_cast<float>(Sb);
  dweight[c] = static_cast<float>(Sw);
}

// ---- backward pass 2: dx, and the per-group sum of dx -------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(kBnMaxThreads) void bn_relu_bwd_dx_kernel(BnArgs p, const float* dweight,
                                                                        const float* dbias) {
  extern __shared__ float sm[];                    // [rl][C] for the group sums
  const int C = p.C, tpr = C / 4, rl = blockDim.x / tpr;
  const int cl = threadIdx.x % tpr, rlane = threadIdx.x / tpr;
  const T* __restrict__ x = static_cast<const T*>(p.x);
  const T* __restrict__ dy = static_cast<const T*>(p.dy);
  T* __restrict__ dx = static_cast<T*>(p.y);
  float scale[4], shift[4], mu[4], is[4], kb[4], kw[4];
  const float inv_n = 1.f / static_cast<float>(p.rows);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = 4 * cl + i;
    mu[i] = p.mean[c]; is[i] = p.invstd[c];
    scale[i] = is[i] * (p.weight ? p.weight[c] : 1.f);
    shift[i] = (p.bias ? p.bias[c] : 0.f) - mu[i] * scale[i];
    kb[i] = p.training ? dbias[c] * inv_n : 0.f;
    kw[i] = p.training ? dweight[c] * inv_n : 0.f;
  }
  const long long r0 = static_cast<long long>(blockIdx.x) * kBnChunk;
  const long long r1 = min(r0 + kBnChunk, p.rows);
  const int group = p.dgterm ? p.dgroup : kBnChunk;
  for (long long gbase = r0; gbase < r1; gbase += group) {
    float gs[4] = {0.f, 0.f, 0.f, 0.f};
    const long long gend = min(gbase + group, r1);
#pragma unroll 4
    for (long long r = gbase + rlane; r < gend; r += rl) {
      float v[4], d[4], o[4];
      load_c4<T>(x + r * p.ld + 4 * cl, v);
      load_c4<T>(dy + r * p.ld + 4 * cl, d);
      if (p.gterm) {
        const float4 g = *reinterpret_cast<const float4*>(p.gterm + (r / p.group) * C + 4 * cl);
        v[0] += g.x; v[1] += g.y; v[2] += g.z; v[3] += g.w;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float dr = fmaf(v[i], scale[i], shift[i]) > 0.f ? d[i] : 0.f;
        const float xh = (v[i] - mu[i]) * is[i];
        o[i] = scale[i] * (dr - kb[i] - xh * kw[i]);
        gs[i] += o[i];
      }
      store_c4<T>(dx + r * p.ld + 4 * cl, o);
    }
    if (p.dgterm) {
      __syncthreads();                              // previous group's reads of sm are done
#pragma unroll
      for (int i = 0; i < 4; ++i) sm[rlane * C + 4 * cl + i] = gs[i];
      __syncthreads();
      for (int e = threadIdx.x; e < C; e += blockDim.x) {
        float acc = 0.f;
        for (int r = 0; r < rl; ++r) acc += sm[r * C + e];
        p.dgterm[(gbase / p.dgroup) * C + e] = acc;
      }
    }
  }
}

// ---- max over the n points of a group ------------------------------------------------------------------------------
template <typename T>
__global__ void group_max_fwd_kernel(const T* __restrict__ x, T* __restrict__ out, unsigned char* __restrict__ idx,
                                     long long groups, int n, int C) {
  const long long e = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;   // (group, channel quad)
  const int cq = C / 4;
  if (e >= groups * cq) return;
  const long long g = e / cq;
  const int c = static_cast<int>(e - g * cq) * 4;
  const T* base = x + g * n * C + c;
  float best[4];
  int bi[4] = {0, 0, 0, 0};
  load_c4<T>(base, best);
  for (int r = 1; r < n; ++r) {
    float v[4];
    load_c4<T>(base + static_cast<long long>(r) * C, v);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bool take = (v[i] > best[i]) || (v[i] != v[i] && best[i] == best[i]);   // first max; NaN wins
      best[i] = take ? v[i] : best[i];
      bi[i] = take ? r : bi[i];
    }
  }
  store_c4<T>(out + g * C + c, best);
  *reinterpret_cast<uchar4*>(idx + g * C + c) = make_uchar4(bi[0], bi[1], bi[2], bi[3]);
}

// same (group, channel quad) -> lane map as the forward: a lane walks the n rows of its group and writes dout to the
// arg-max row, zeros elsewhere (no per-element index arithmetic; the (group, row, quad) map of the first version
// spent its time in 64-bit divisions and reached 2.6 TB/s)
template <typename T>
__global__ void group_max_bwd_kernel(const T* __restrict__ dout, const unsigned char* __restrict__ idx,
                                     T* __restrict__ dx, long long groups, int n, int C) {
  const long long e = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;   // (group, channel quad)
  const int cq = C / 4;
  if (e >= groups * cq) return;
  const long long g = e / cq;
  const int c = static_cast<int>(e - g * cq) * 4;
  float d[4];
  load_c4<T>(dout + g * C + c, d);
  const uchar4 id = *reinterpret_cast<const uchar4*>(idx + g * C + c);
  T* base = dx + g * n * C + c;
#pragma unroll 4
  for (int r = 0; r < n; ++r) {
    float o[4];
    o[0] = id.x == r ? d[0] : 0.f; o[1] = id.y == r ? d[1] : 0.f;
    o[2] = id.z == r ? d[2] : 0.f; o[3] = id.w == r ? d[3] : 0.f;
    store_c4<T>(base + static_cast<long long>(r) * C, o);
  }
}

static bool bn_shape_ok(long long rows, int C, long long ld, int group, bool has_g) {
  if (rows <= 0 || C < 4 || C > 4 * kBnMaxThreads || (C % 4) != 0 || ld < C || (ld % 4) != 0) return false;
  if (has_g && (group <= 0 || (rows % group) != 0)) return false;
  return true;
}
static dim3 bn_block(int C) {
  const int tpr = C / 4;
  const int rl = kBnMaxThreads / tpr > 0 ? kBnMaxThreads / tpr : 1;
  return dim3(tpr * rl);
}
static int bn_row_lanes(int C) { return static_cast<int>(bn_block(C).x) / (C / 4); }

}  // namespace simamba

using namespace simamba;

extern "C" int simamba_bn_relu_grid(long long rows) {
  return rows <= 0 ? 0 : static_cast<int>((rows + kBnChunk - 1) / kBnChunk);
}

extern "C" int simamba_bn_relu_fwd(const void* x, const float* gterm, int group, const float* weight,
                                   const float* bias, float* running_mean, float* running_var, float momentum,
                                   float eps, int training, void* y, float* mean, float* invstd, float* partial,
                                   long long rows, int C, long long ld, int io_dtype, void* stream) {
  if (io_dtype != SIMAMBA_F32 && io_dtype != SIMAMBA_BF16) return SIMAMBA_E_DTYPE;
  if (rows == 0) return SIMAMBA_OK;
  if (ld == 0) ld = C;
  if (!bn_shape_ok(rows, C, ld, group, gterm != nullptr)) return SIMAMBA_E_SHAPE;
  if (!x || !y || !mean || !invstd) return SIMAMBA_E_NULLPTR;
  if (training && !partial) return SIMAMBA_E_NULLPTR;
  if (!training && (!running_mean || !running_var)) return SIMAMBA_E_NULLPTR;
  hipStream_t s = static_cast<hipStream_t>(stream);
  BnArgs a{};
  a.x = x; a.gterm = gterm; a.weight = weight; a.bias = bias; a.mean = mean; a.invstd = invstd;
  a.y = y; a.partial = partial; a.rows = rows; a.ld = ld; a.C = C; a.group = group; a.dgroup = group;
  a.training = training;
  const int grid = simamba_bn_relu_grid(rows);
  const dim3 block = bn_block(C);
  const size_t smem = sizeof(float) * 2 * C * bn_row_lanes(C);
  const bool f32 = io_dtype == SIMAMBA_F32;
  if (training) {
    if (f32) {
      hipLaunchKernelGGL(bn_stats_kernel<float>, dim3(grid), block, smem, s, a);
      hipLaunchKernelGGL(bn_stats_finalize_kernel<float>, dim3((C + kFinCh - 1) / kFinCh), dim3(kFinCh * kFinPl), 0, s, x, gterm, partial,
                         grid, rows, C, eps, momentum, mean, invstd, running_mean, running_var);
    } else {
      hipLaunchKernelGGL(bn_stats_kernel<bf16_t>, dim3(grid), block, smem, s, a);
      hipLaunchKernelGGL(bn_stats_finalize_kernel<bf16_t>, dim3((C + kFinCh - 1) / kFinCh), dim3(kFinCh * kFinPl), 0, s, x, gterm, partial,
                         grid, rows, C, eps, momentum, mean, invstd, running_mean, running_var);
    }
  } else {
    hipLaunchKernelGGL(bn_eval_stats_kernel, dim3((C + 127) / 128), dim3(128), 0, s, running_mean, running_var, eps, C,
                       mean, invstd);
  }
  if (f32) hipLaunchKernelGGL(bn_relu_apply_kernel<float>, dim3(grid), block, 0, s, a);
  else hipLaunchKernelGGL(bn_relu_apply_kernel<bf16_t>, dim3(grid), block, 0, s, a);
  return static_cast<int>(hipGetLastError());
}

extern "C" int simamba_bn_relu_bwd(const void* dy, const void* x, const float* gterm, int group,
                                   const float* weight, const float* bias, const float* mean, const float* invstd,
                                   void* dx, float* dgterm, int dgroup, float* dweight, float* dbias, float* partial,
                                   long long rows, int C, long long ld, int io_dtype, int training, void* stream) {
  if (io_dtype != SIMAMBA_F32 && io_dtype != SIMAMBA_BF16) return SIMAMBA_E_DTYPE;
  if (rows == 0) return SIMAMBA_OK;
  if (ld == 0) ld = C;
  if (!bn_shape_ok(rows, C, ld, group, gterm != nullptr)) return SIMAMBA_E_SHAPE;
  if (dgterm && (dgroup <= 0 || (kBnChunk % dgroup) != 0 || (rows % dgroup) != 0)) return SIMAMBA_E_SHAPE;
  if (!dy || !x || !mean || !invstd || !dx || !dweight || !dbias || !partial) return SIMAMBA_E_NULLPTR;
  if (dgterm && !gterm) return SIMAMBA_E_NULLPTR;
  hipStream_t s = static_cast<hipStream_t>(stream);
  BnArgs a{};
  a.x = x; a.gterm = gterm; a.weight = weight; a.bias = bias; a.mean = mean; a.invstd = invstd;
  a.dy = dy; a.y = dx; a.dgterm = dgterm; a.partial = partial; a.rows = rows; a.ld = ld; a.C = C; a.group = group;
  a.dgroup = dgroup; a.training = training;
  const int grid = simamba_bn_relu_grid(rows);
  const dim3 block = bn_block(C);
  const int rl = bn_row_lanes(C);
  const bool f32 = io_dtype == SIMAMBA_F32;
  if (f32) hipLaunchKernelGGL(bn_relu_bwd_reduce_kernel<float>, dim3(grid), block, sizeof(float) * 2 * C * rl, s, a);
  else hipLaunchKernelGGL(bn_relu_bwd_reduce_kernel<bf16_t>, dim3(grid), block, sizeof(float) * 2 * C * rl, s, a);
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + kFinCh - 1) / kFinCh), dim3(kFinCh * kFinPl), 0, s, partial, grid, C,
                     dweight, dbias);
  if (f32) hipLaunchKernelGGL(bn_relu_bwd_dx_kernel<float>, dim3(grid), block, sizeof(float) * C * rl, s, a, dweight, dbias);
  else hipLaunchKernelGGL(bn_relu_bwd_dx_kernel<bf16_t>, dim3(grid), block, sizeof(float) * C * rl, s, a, dweight, dbias);
  return static_cast<int>(hipGetLastError());
}

extern "C" int simamba_group_max_fwd(const void* x, void* out, unsigned char* idx, long long groups, int n, int C,
                                     int io_dtype, void* stream) {
  if (io_dtype != SIMAMBA_F32 && io_dtype != SIMAMBA_BF16) return SIMAMBA_E_DTYPE;
  if (groups == 0) return SIMAMBA_OK;
  if (groups < 0 || n < 1 || n > 256 || C < 4 || (C % 4) != 0) return SIMAMBA_E_SHAPE;
  if (!x || !out || !idx) return SIMAMBA_E_NULLPTR;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const long long total = groups * (C / 4);
  const dim3 grid(static_cast<unsigned>((total + 255) / 256));
  if (io_dtype == SIMAMBA_F32)
    hipLaunchKernelGGL(group_max_fwd_kernel<float>, grid, dim3(256), 0, s, static_cast<const float*>(x),
                       static_cast<float*>(out), idx, groups, n, C);
  else
    hipLaunchKernelGGL(group_max_fwd_kernel<bf16_t>, grid, dim3(256), 0, s, static_cast<const bf16_t*>(x),
                       static_cast<bf16_t*>(out), idx, groups, n, C);
  return static_cast<int>(hipGetLastError());
}

extern "C" int simamba_group_max_bwd(const void* dout, const unsigned char* idx, void* dx, long long groups, int n,
                                     int C, int io_dtype, void* stream) {
  if (io_dtype != SIMAMBA_F32 && io_dtype != SIMAMBA_BF16) return SIMAMBA_E_DTYPE;
  if (groups == 0) return SIMAMBA_OK;
  if (groups < 0 || n < 1 || n > 256 || C < 4 || (C % 4) != 0) return SIMAMBA_E_SHAPE;
  if (!dout || !idx || !dx) return SIMAMBA_E_NULLPTR;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const long long total = groups * (C / 4);
  if (total > 0x7fffffffll * 256) return SIMAMBA_E_SHAPE;
  const dim3 grid(static_cast<unsigned>((total + 255) / 256));
  if (io_dtype == SIMAMBA_F32)
    hipLaunchKernelGGL(group_max_bwd_kernel<float>, grid, dim3(256), 0, s, static_cast<const float*>(dout), idx,
                       static_cast<float*>(dx), groups, n, C);
  else
    hipLaunchKernelGGL(group_max_bwd_kernel<bf16_t>, grid, dim3(256), 0, s, static_cast<const bf16_t*>(dout), idx,
                       static_cast<bf16_t*>(dx), groups, n, C);
  return static_cast<int>(hipGetLastError());
}
