// Fused (DropPath-scaled) residual add + LayerNorm, forward and backward, for gfx950.
//
// Replaces, inside the reference's Block.forward (models/block.py:56-60) and the final norm of MixerModel
// (models/point_mamba.py:257-258), the chain  drop_path(h) -> + residual -> LayerNorm  (4 torch launches
// reading/writing ~700 MB per block forward at the model shape, ~1 GB backward) by one streaming pass each
// way: forward reads h, residual and writes residual_out, normed (4 row-sized tensors), backward reads
// dnormed, dresidual_out, residual_out and writes dresidual (+ dhidden when DropPath rescales it).
// One wave per row; the row sits in registers as 16-byte chunks (dim <= 2048, dim % 4 == 0); mean / variance
// and the two backward projections are wave all-reduces; the LayerNorm weight/bias gradients are accumulated
// per lane across the rows a wave walks, summed over the workgroup's 4 waves in LDS and written as one
// partial row per workgroup (summed by the caller -- deterministic, no atomics).
// HBM-bound: 4*R*dim*s bytes forward, 4-5 * R*dim*s backward (R = batch * rows_per_batch).
#include "common.h"

namespace simamba {

constexpr int kLnThreads = 256;
constexpr int kLnMaxChunks = 8;   // 16-byte chunks per lane: dim <= 64 * 4 * 8

struct LnArgs {
  const void* hidden;
  const float* residual;
  const float* rowscale;
  const float* weight;
  const float* bias;
  float* residual_out;
  void* normed;
  float* mean;
  float* rstd;
  // backward
  const void* dnormed;
  const float* dresidual_out;
  float* dresidual;
  void* dhidden;
  float* dwb_partial;   // (gridDim.x, 2, dim)
  int batch, rows_per_batch, dim;
  float eps;
};

__device__ __forceinline__ float wave_allreduce_sum(float v) {
  v = row_allreduce_sum(v);
  v += __shfl_xor(v, 16);
  v += __shfl_xor(v, 32);
  return v;
}

template <typename T>
__device__ __forceinline__ float4 ld4(const T* p) {
  const Pack<T, 4> pk = *reinterpret_cast<const Pack<T, 4>*>(p);
  return make_float4(to_f32<T>(pk.v[0]), to_f32<T>(pk.v[1]), to_f32<T>(pk.v[2]), to_f32<T>(pk.v[3]));
}
template <typename T>
__device__ __forceinline__ void st4(T* p, float4 v) {
  Pack<T, 4> pk;
  pk.v[0] = from_f32<T>(v.x); pk.v[1] = from_f32<T>(v.y); pk.v[2] = from_f32<T>(v.z); pk.v[3] = from_f32<T>(v.w);
  *reinterpret_cast<Pack<T, 4>*>(p) = pk;
}

template <typename TH, typename TO, int kChunks>
__global__ __launch_bounds__(kLnThreads) void add_ln_fwd_kernel(LnArgs p) {
  const int lane = threadIdx.x & 63;
  const int wave_global = (blockIdx.x * kLnThreads + threadIdx.x) >> 6;
  const int nwaves = (gridDim.x * kLnThreads) >> 6;
  const int rows = p.batch * p.rows_per_batch;
  const int nch = p.dim >> 2;
  const TH* __restrict__ hg = static_cast<const TH*>(p.hidden);
  TO* __restrict__ og = static_cast<TO*>(p.normed);
  float4 w[kChunks], bs[kChunks];
#pragma unroll
  for (int k = 0; k < kChunks; ++k) {
    const int c = lane + 64 * k;
    w[k] = c < nch ? *reinterpret_cast<const float4*>(p.weight + 4 * c) : make_float4(0, 0, 0, 0);
    bs[k] = (c < nch && p.bias) ? *reinterpret_cast<const float4*>(p.bias + 4 * c) : make_float4(0, 0, 0, 0);
  }
  const float inv_dim = 1.f / p.dim;
  for (int row = wave_global; row < rows; row += nwaves) {
    const size_t base = static_cast<size_t>(row) * p.dim;
    const float scale = (p.rowscale && p.residual) ? p.rowscale[row / p.rows_per_batch] : 1.f;
    float4 x[kChunks];
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < kChunks; ++k) {
      const int c = lane + 64 * k;
      if (c < nch) {
        float4 h = ld4<TH>(hg + base + 4 * c);
        if (p.residual) {
          const float4 r = *reinterpret_cast<const float4*>(p.residual + base + 4 * c);
          h.x = fmaf(h.x, scale, r.x); h.y = fmaf(h.y, scale, r.y);
          h.z = fmaf(h.z, scale, r.z); h.w = fmaf(h.w, scale, r.w);
        }
        if (p.residual_out) *reinterpret_cast<float4*>(p.residual_out + base + 4 * c) = h;
        x[k] = h;
        sum += (h.x + h.y) + (h.z + h.w);
      } else {
        x[k] = make_float4(0, 0, 0, 0);
      }
    }
    const float mean = wave_allreduce_sum(sum) * inv_dim;
    float sq = 0.f;
#pragma unroll
    for (int k = 0; k < kChunks; ++k) {
      if (lane + 64 * k < nch) {
        const float a = x[k].x - mean, b = x[k].y - mean, c = x[k].z - mean, d = x[k].w - mean;
        sq += (a * a + b * b) + (c * c + d * d);
      }
    }
    const float rstd = rsqrtf(wave_allreduce_sum(sq) * inv_dim + p.eps);
#pragma unroll
    for (int k = 0; k < kChunks; ++k) {
      const int c = lane + 64 * k;
      if (c < nch) {
        float4 y;
        y.x = fmaf((x[k].x - mean) * rstd, w[k].x, bs[k].x);
        y.y = fmaf((x[k].y - mean) * rstd, w[k].y, bs[k].y);
        y.z = fmaf((x[k].z - mean) * rstd, w[k].z, bs[k].z);
        y.w = fmaf((x[k].w - mean) * rstd, w[k].w, bs[k].w);
        st4<TO>(og + base + 4 * c, y);
      }
    }
    if (lane == 0) { p.mean[row] = mean; p.rstd[row] = rstd; }
  }
}

template <typename TH, typename TO, int kChunks>
__global__ __launch_bounds__(kLnThreads) void add_ln_bwd_kernel(LnArgs p) {
  extern __shared__ float sred[];          // [4 waves][2][dim]
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int wave_global = (blockIdx.x * kLnThreads + threadIdx.x) >> 6;
  const int nwaves = (gridDim.x * kLnThreads) >> 6;
  const int rows = p.batch * p.rows_per_batch;
  const int nch = p.dim >> 2;
  const TO* __restrict__ dyg = static_cast<const TO*>(p.dnormed);
  TH* __restrict__ dhg = static_cast<TH*>(p.dhidden);
  float4 w[kChunks], dw[kChunks], db[kChunks];
#pragma unroll
  for (int k = 0; k < kChunks; ++k) {
    const int c = lane + 64 * k;
    w[k] = c < nch ? *reinterpret_cast<const float4*>(p.weight + 4 * c) : make_float4(0, 0, 0, 0);
    dw[k] = make_float4(0, 0, 0, 0);
    db[k] = make_float4(0, 0, 0, 0);
  }
  const float inv_dim = 1.f / p.dim;
  for (int row = wave_global; row < rows; row += nwaves) {
    const size_t base = static_cast<size_t>(row) * p.dim;
    const float mean = p.mean[row], rstd = p.rstd[row];
    float4 xh[kChunks], g[kChunks];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < kChunks; ++k) {
      const int c = lane + 64 * k;
      if (c < nch) {
        const float4 x = *reinterpret_cast<const float4*>(p.residual_out + base + 4 * c);
        const float4 dy = ld4<TO>(dyg + base + 4 * c);
        xh[k] = make_float4((x.x - mean) * rstd, (x.y - mean) * rstd, (x.z - mean) * rstd, (x.w - mean) * rstd);
        g[k] = make_float4(dy.x * w[k].x, dy.y * w[k].y, dy.z * w[k].z, dy.w * w[k].w);
        dw[k].x = fmaf(dy.x, xh[k].x, dw[k].x); dw[k].y = fmaf(dy.y, xh[k].y, dw[k].y);
        dw[k].z = fmaf(dy.z, xh[k].z, dw[k].z); dw[k].w = fmaf(dy.w, xh[k].w, dw[k].w);
        db[k].x += dy.x; db[k].y += dy.y; db[k].z += dy.z; db[k].w += dy.w;
        s1 += (g[k].x + g[k].y) + (g[k].z + g[k].w);
        s2 += (g[k].x * xh[k].x + g[k].y * xh[k].y) + (g[k].z * xh[k].z + g[k].w * xh[k].w);
      } else {
        xh[k] = make_float4(0, 0, 0, 0);
        g[k] = make_float4(0, 0, 0, 0);
      }
    }
    const float c1 = wave_allreduce_sum(s1) * inv_dim;
    const float c2 = wave_allreduce_sum(s2) * inv_dim;
    const float scale = (p.rowscale && dhg) ? p.rowscale[row / p.rows_per_batch] : 1.f;
#pragma unroll
    for (int k = 0; k < kChunks; ++k) {
      const int c = lane + 64 * k;
      if (c < nch) {
        float4 dx;
        dx.x = (g[k].x - c1 - xh[k].x * c2) * rstd;
        dx.y = (g[k].y - c1 - xh[k].y * c2) * rstd;
        dx.z = (g[k].z - c1 - xh[k].z * c2) * rstd;
        dx.w = (g[k].w - c1 - xh[k].w * c2) * rstd;
        if (p.dresidual_out) {
          const float4 r = *reinterpret_cast<const float4*>(p.dresidual_out + base + 4 * c);
          dx.x += r.x; dx.y += r.y; dx.z += r.z; dx.w += r.w;
        }
        if (p.dresidual) *reinterpret_cast<float4*>(p.dresidual + base + 4 * c) = dx;
        if (dhg) st4<TH>(dhg + base + 4 * c, make_float4(dx.x * scale, dx.y * scale, dx.z * scale, dx.w * scale));
      }
    }
  }
  // weight / bias gradients: lanes -> LDS per wave -> sum over the 4 waves -> one partial row per workgroup
  float* mine = sred + wave * 2 * p.dim;
#pragma unroll
  for (int k = 0; k < kChunks; ++k) {
    const int c = lane + 64 * k;
    if (c < nch) {
      *reinterpret_cast<float4*>(mine + 4 * c) = dw[k];
      *reinterpret_cast<float4*>(mine + p.dim + 4 * c) = db[k];
    }
  }
  __syncthreads();
  float* out = p.dwb_partial + static_cast<size_t>(blockIdx.x) * 2 * p.dim;
  for (int i = threadIdx.x; i < 2 * p.dim; i += kLnThreads) {
    float v = 0.f;
#pragma unroll
    for (int wv = 0; wv < kLnThreads / 64; ++wv) v += sred[wv * 2 * p.dim + i];
    out[i] = v;
  }
}

static int ln_grid(int rows) {
  int g = (rows + 3) / 4;            // 4 rows (waves) per workgroup pass
  return g < 1 ? 1 : (g > 1024 ? 1024 : g);
}

template <typename TH, typename TO>
static int launch_ln(const LnArgs& a, bool bwd, int grid, hipStream_t s) {
  const int chunks = ((a.dim >> 2) + 63) / 64;
#define SIMAMBA_LN_CASE(K)                                                                                      \
  if (chunks <= K) {                                                                                             \
    if (bwd)                                                                                                     \
      hipLaunchKernelGGL((add_ln_bwd_kernel<TH, TO, K>), dim3(grid), dim3(kLnThreads),                           \
                         sizeof(float) * (kLnThreads / 64) * 2 * a.dim, s, a);                                   \
    else                                                                                                         \
      hipLaunchKernelGGL((add_ln_fwd_kernel<TH, TO, K>), dim3(grid), dim3(kLnThreads), 0, s, a);                 \
    return static_cast<int>(hipGetLastError());                                                                  \
  }
  SIMAMBA_LN_CASE(1)
  SIMAMBA_LN_CASE(2)
  SIMAMBA_LN_CASE(4)
  SIMAMBA_LN_CASE(8)
#undef SIMAMBA_LN_CASE
  return SIMAMBA_E_SHAPE;
}

static int dispatch_ln(const LnArgs& a, bool bwd, int grid, int hidden_dtype, int out_dtype, hipStream_t s) {
  if (hidden_dtype == SIMAMBA_F32 && out_dtype == SIMAMBA_F32) return launch_ln<float, float>(a, bwd, grid, s);
  if (hidden_dtype == SIMAMBA_BF16 && out_dtype == SIMAMBA_BF16) return launch_ln<bf16_t, bf16_t>(a, bwd, grid, s);
  if (hidden_dtype == SIMAMBA_F32 && out_dtype == SIMAMBA_BF16) return launch_ln<float, bf16_t>(a, bwd, grid, s);
  if (hidden_dtype == SIMAMBA_BF16 && out_dtype == SIMAMBA_F32) return launch_ln<bf16_t, float>(a, bwd, grid, s);
  return SIMAMBA_E_DTYPE;
}

}  // namespace simamba

using namespace simamba;

static int check_ln(int batch, int rows_per_batch, int dim) {
  if (batch < 0 || rows_per_batch < 0) return SIMAMBA_E_SHAPE;
  if (dim <= 0 || (dim & 3) || dim > 64 * 4 * kLnMaxChunks) return SIMAMBA_E_SHAPE;
  return SIMAMBA_OK;
}

extern "C" int simamba_add_layer_norm_grid(int batch, int rows_per_batch) {
  const long long rows = static_cast<long long>(batch) * rows_per_batch;
  return ln_grid(rows > 2000000000ll ? 2000000000 : static_cast<int>(rows));
}

extern "C" int simamba_add_layer_norm_fwd(const void* hidden, const float* residual, const float* rowscale,
                                          const float* weight, const float* bias, float* residual_out, void* normed,
                                          float* mean, float* rstd, int batch, int rows_per_batch, int dim, float eps,
                                          int hidden_dtype, int out_dtype, void* stream) {
  int rc = check_ln(batch, rows_per_batch, dim);
  if (rc) return rc;
  if (batch == 0 || rows_per_batch == 0) return SIMAMBA_OK;
  if (!hidden || !weight || !normed || !mean || !rstd) return SIMAMBA_E_NULLPTR;
  LnArgs a{};
  a.hidden = hidden; a.residual = residual; a.rowscale = rowscale; a.weight = weight; a.bias = bias;
  a.residual_out = residual_out; a.normed = normed; a.mean = mean; a.rstd = rstd;
  a.batch = batch; a.rows_per_batch = rows_per_batch; a.dim = dim; a.eps = eps;
  return dispatch_ln(a, false, simamba_add_layer_norm_grid(batch, rows_per_batch), hidden_dtype, out_dtype,
                     static_cast<hipStream_t>(stream));
}

extern "C" int simamba_add_layer_norm_bwd(const void* dnormed, const float* dresidual_out, const float* residual_out,
                                          const float* mean, const float* rstd, const float* weight,
                                          const float* rowscale, float* dresidual, void* dhidden, float* dwb_partial,
                                          int batch, int rows_per_batch, int dim, int hidden_dtype, int out_dtype,
                                          void* stream) {
  int rc = check_ln(batch, rows_per_batch, dim);
  if (rc) return rc;
  if (batch == 0 || rows_per_batch == 0) return SIMAMBA_OK;
  if (!dnormed || !residual_out || !mean || !rstd || !weight || !dwb_partial || (!dresidual && !dhidden))
    return SIMAMBA_E_NULLPTR;
  LnArgs a{};
  a.dnormed = dnormed; a.dresidual_out = dresidual_out; a.residual_out = const_cast<float*>(residual_out);
  a.mean = const_cast<float*>(mean); a.rstd = const_cast<float*>(rstd); a.weight = weight; a.rowscale = rowscale;
  a.dresidual = dresidual; a.dhidden = dhidden; a.dwb_partial = dwb_partial;
  a.batch = batch; a.rows_per_batch = rows_per_batch; a.dim = dim;
  return dispatch_ln(a, true, simamba_add_layer_norm_grid(batch, rows_per_batch), hidden_dtype, out_dtype,
                     static_cast<hipStream_t>(stream));
}
