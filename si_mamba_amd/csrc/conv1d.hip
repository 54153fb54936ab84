// Causal depthwise conv1d (+SiLU), forward and backward, for gfx950.
//
// Replaces causal_conv1d_cuda.causal_conv1d_{fwd,bwd} of causal-conv1d as used inside the mixer
// the reference calls at models/block.py:72.  Pure streaming op: every lane owns 4 consecutive
// timesteps of one (batch, channel) row (one 16-byte access for fp32), the 3-step halo comes
// from the neighbouring pack (an L1/L2 hit).  A workgroup owns a tile of channels and walks a
// slice of the batch, so the weight-gradient reduction over (batch, time) happens in registers,
// then over the 16-lane DPP row, then in LDS, and only one atomic per (channel, tap) and
// workgroup reaches HBM.
// Algorithmic HBM bytes: fwd 2*B*D*L*s, bwd 3*B*D*L*s (+ (W+1)*D*4 parameters).
#include "common.h"

namespace simamba {

constexpr int kConvThreads = 256;
constexpr int kPack = 4;

struct ConvArgs {
  const void* x;
  const float* w;
  const float* bias;
  void* out;       // fwd: out ; bwd: dx
  const void* dout;
  float* dw;
  float* dbias;
  int batch, dim, seqlen, width;
  int silu;
  int vec;
  int ppr;         // padded packs per row (power of two, >= 16)
  int bchunk;      // batch samples per workgroup
  long long x_bs;  // batch stride (elements) of x
  long long o_bs;  // batch stride (elements) of out (fwd: contiguous) / dx (bwd)
};

template <typename T, int P = kPack>
__device__ __forceinline__ void load_pack(const T* row, int t0, int L, bool vec, float (&v)[P]) {
  if (t0 < 0 || t0 >= L) {
#pragma unroll
    for (int j = 0; j < P; ++j) v[j] = 0.f;
    return;
  }
  load_items<T, P>(row + t0, L - t0, vec, v);
}

__device__ __forceinline__ void load_taps(const ConvArgs& p, int d, float (&w4)[4], float& bias) {
  const int W = p.width;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    int k = j - (4 - W);
    w4[j] = (k >= 0) ? p.w[d * W + k] : 0.f;
  }
  bias = p.bias ? p.bias[d] : 0.f;
}

template <typename T>
__global__ __launch_bounds__(kConvThreads) void conv1d_fwd_kernel(ConvArgs p) {
  const int L = p.seqlen, D = p.dim;
  const int rows_per_blk = p.ppr >= kConvThreads ? 1 : kConvThreads / p.ppr;
  const int drow = p.ppr >= kConvThreads ? 0 : threadIdx.x / p.ppr;
  const int d = blockIdx.x * rows_per_blk + drow;
  if (d >= D) return;
  float w4[4], bias;
  load_taps(p, d, w4, bias);
  const T* __restrict__ xg = static_cast<const T*>(p.x);
  T* __restrict__ og = static_cast<T*>(p.out);
  const int b0 = blockIdx.y * p.bchunk;
  const int b1 = min(b0 + p.bchunk, p.batch);
  const bool vec = p.vec != 0;
  for (int b = b0; b < b1; ++b) {
    const T* row = xg + static_cast<size_t>(b) * p.x_bs + static_cast<size_t>(d) * L;
    T* orow = og + static_cast<size_t>(b) * p.o_bs + static_cast<size_t>(d) * L;
    for (int pk = (p.ppr >= kConvThreads ? threadIdx.x : threadIdx.x % p.ppr); pk * kPack < L;
         pk += (p.ppr >= kConvThreads ? kConvThreads : p.ppr)) {
      const int t0 = pk * kPack;
      float xp[kPack], xc[kPack], o[kPack];
      load_pack<T>(row, t0 - kPack, L, vec, xp);
      load_pack<T>(row, t0, L, vec, xc);
      const float win[7] = {xp[1], xp[2], xp[3], xc[0], xc[1], xc[2], xc[3]};
#pragma unroll
      for (int i = 0; i < kPack; ++i) {
        float acc = bias;
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = fmaf(w4[j], win[i + j], acc);
        o[i] = p.silu ? acc * sigmoid_f(acc) : acc;
      }
      store_items<T, kPack>(orow + t0, L - t0, vec, o);
    }
  }
}

__device__ __forceinline__ void conv_bwd_reduce(const ConvArgs& p, float (&sacc)[16][5], float (&gw)[4], float gb, int drow,
                                                int rows_per_blk, bool dvalid) {
  // 16-lane rows never straddle channels (ppr is a multiple of 16)
#pragma unroll
  for (int j = 0; j < 4; ++j) gw[j] = row_allreduce_sum(gw[j]);
  gb = row_allreduce_sum(gb);
  if ((threadIdx.x & 15) == 0 && dvalid) {
#pragma unroll
    for (int j = 0; j < 4; ++j) atomicAdd(&sacc[drow][j], gw[j]);
    atomicAdd(&sacc[drow][4], gb);
  }
  __syncthreads();
  if (threadIdx.x < rows_per_blk * 5) {
    const int r = threadIdx.x / 5, j = threadIdx.x % 5;
    const int dd = blockIdx.x * rows_per_blk + r;
    if (dd < p.dim) {
      const float v = sacc[r][j];
      if (j < 4) {
        const int k = j - (4 - p.width);
        if (k >= 0) atomicAdd(&p.dw[dd * p.width + k], v);
      } else if (p.dbias) {
        atomicAdd(&p.dbias[dd], v);
      }
    }
  }
}

// P: timesteps per lane and step -- one 16-byte access: 4 for fp32, 8 for bf16 (with 4 the bf16 kernel issued twice the
// loads and recomputed 7 SiLU derivatives per 4 outputs instead of 11 per 8: 120 us where its bytes take 55)
template <typename T, int P>
__global__ __launch_bounds__(kConvThreads) void conv1d_bwd_kernel(ConvArgs p) {
  __shared__ float sacc[16][5];   // up to 16 channels per workgroup x (4 taps + bias)
  const int L = p.seqlen, D = p.dim;
  const int rows_per_blk = p.ppr >= kConvThreads ? 1 : kConvThreads / p.ppr;
  const int drow = p.ppr >= kConvThreads ? 0 : threadIdx.x / p.ppr;
  const int d = blockIdx.x * rows_per_blk + drow;
  const bool dvalid = d < D;
  const int dc = dvalid ? d : D - 1;
  if (threadIdx.x < 16 * 5) (&sacc[0][0])[threadIdx.x] = 0.f;
  __syncthreads();
  float w4[4], bias;
  load_taps(p, dc, w4, bias);
  const T* __restrict__ xg = static_cast<const T*>(p.x);
  const T* __restrict__ gg = static_cast<const T*>(p.dout);
  T* __restrict__ dxg = static_cast<T*>(p.out);
  const int b0 = blockIdx.y * p.bchunk;
  const int b1 = min(b0 + p.bchunk, p.batch);
  const bool vec = p.vec != 0;
  float gw[4] = {0.f, 0.f, 0.f, 0.f}, gb = 0.f;
  for (int b = b0; b < b1; ++b) {
    const size_t roff = (static_cast<size_t>(b) * D + dc) * L;                      // dout: contiguous
    const size_t xoff = static_cast<size_t>(b) * p.x_bs + static_cast<size_t>(dc) * L;
    const size_t doff = static_cast<size_t>(b) * p.o_bs + static_cast<size_t>(dc) * L;
    for (int pk = (p.ppr >= kConvThreads ? threadIdx.x : threadIdx.x % p.ppr); pk * P < L;
         pk += (p.ppr >= kConvThreads ? kConvThreads : p.ppr)) {
      const int t0 = pk * P;
      float xp[P], xc[P], xn[P], gc[P], gn[P];
      load_pack<T, P>(xg + xoff, t0 - P, L, vec, xp);
      load_pack<T, P>(xg + xoff, t0, L, vec, xc);
      load_pack<T, P>(xg + xoff, t0 + P, L, vec, xn);
      load_pack<T, P>(gg + roff, t0, L, vec, gc);
      load_pack<T, P>(gg + roff, t0 + P, L, vec, gn);
      // x[t0-3 .. t0+P+2], dout[t0 .. t0+P+2]
      float xw[P + 6], gv[P + 3];
#pragma unroll
      for (int i = 0; i < 3; ++i) { xw[i] = xp[P - 3 + i]; xw[P + 3 + i] = xn[i]; gv[P + i] = gn[i]; }
#pragma unroll
      for (int i = 0; i < P; ++i) { xw[3 + i] = xc[i]; gv[i] = gc[i]; }
      float dpre[P + 3];
#pragma unroll
      for (int i = 0; i < P + 3; ++i) {
        float g = gv[i];
        if (p.silu) {
          float pre = bias;
#pragma unroll
          for (int j = 0; j < 4; ++j) pre = fmaf(w4[j], xw[i + j], pre);
          float sg = sigmoid_f(pre);
          g = g * sg * (1.f + pre * (1.f - sg));
        }
        dpre[i] = (t0 + i < L) ? g : 0.f;
      }
      float dx[P];
#pragma unroll
      for (int i = 0; i < P; ++i) {
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = fmaf(w4[j], dpre[i + 3 - j], acc);
        dx[i] = acc;
        gb += dpre[i];
#pragma unroll
        for (int j = 0; j < 4; ++j) gw[j] = fmaf(dpre[i], xw[i + j], gw[j]);
      }
      if (dvalid) store_items<T, P>(dxg + doff + t0, L - t0, vec, dx);
    }
  }
  conv_bwd_reduce(p, sacc, gw, gb, drow, rows_per_blk, dvalid);
}

// ---- the aligned case (every pack one whole 16-byte access, L % P == 0): same arithmetic, software-pipelined ----------
// One (sample, pack) per iteration; the five raw packs of the NEXT iteration are requested before this one's arithmetic,
// so a wave has two iterations of loads in flight instead of stalling a full HBM latency per sample (the loop above:
// 103 us in bf16 where the bytes take 55).  Neighbouring packs are loaded from clamped addresses and zeroed by select:
// a load under a condition would hold the pipeline's registers live across both arms.
template <typename T, int P>
__device__ __forceinline__ void unpack16(const uint4& r, float (&v)[P]) {
  if constexpr (sizeof(T) == 4) {
    v[0] = __builtin_bit_cast(float, r.x); v[1] = __builtin_bit_cast(float, r.y);
    v[2] = __builtin_bit_cast(float, r.z); v[3] = __builtin_bit_cast(float, r.w);
  } else {
    const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      v[2 * i] = __builtin_bit_cast(float, w[i] << 16);
      v[2 * i + 1] = __builtin_bit_cast(float, w[i] & 0xffff0000u);
    }
  }
}

template <typename T, int P>
__global__ __launch_bounds__(kConvThreads) void conv1d_bwd_fast_kernel(ConvArgs p) {
  static_assert(sizeof(T) * P == 16, "one 16-byte access per pack");
  __shared__ float sacc[16][5];
  const int L = p.seqlen, D = p.dim;
  const int rows_per_blk = p.ppr >= kConvThreads ? 1 : kConvThreads / p.ppr;
  const int drow = p.ppr >= kConvThreads ? 0 : threadIdx.x / p.ppr;
  const int d = blockIdx.x * rows_per_blk + drow;
  const bool dvalid = d < D;
  const int dc = dvalid ? d : D - 1;
  if (threadIdx.x < 16 * 5) (&sacc[0][0])[threadIdx.x] = 0.f;
  __syncthreads();
  float w4[4], bias;
  load_taps(p, dc, w4, bias);
  const T* __restrict__ xg = static_cast<const T*>(p.x);
  const T* __restrict__ gg = static_cast<const T*>(p.dout);
  T* __restrict__ dxg = static_cast<T*>(p.out);
  const int b0 = blockIdx.y * p.bchunk;
  const int nb = min(b0 + p.bchunk, p.batch) - b0;
  const int packs = L / P;
  const int pk0 = p.ppr >= kConvThreads ? threadIdx.x : threadIdx.x % p.ppr;
  const int pstride = p.ppr >= kConvThreads ? kConvThreads : p.ppr;
  const int npk = pk0 < packs ? (packs - pk0 + pstride - 1) / pstride : 0;
  const int n = nb * npk;

  struct Raw { uint4 xp, xc, xn, gc, gn; };
  auto where = [&](int it, int& b, int& t0) {
    const int q = npk == 1 ? it : it / npk;
    b = b0 + q;
    t0 = (pk0 + (it - q * npk) * pstride) * P;
  };
  auto load_raw = [&](int it_) {
    const int it = it_ < n ? it_ : n - 1;                      // unconditional: the step past the end re-reads the last
    int b, t0;
    where(it, b, t0);
    const T* xr = xg + static_cast<size_t>(b) * p.x_bs + static_cast<size_t>(dc) * L;
    const T* gr = gg + (static_cast<size_t>(b) * D + dc) * L;
    const int tp = t0 >= P ? t0 - P : 0, tn = t0 + P < L ? t0 + P : t0;
    Raw r;
    r.xp = *reinterpret_cast<const uint4*>(xr + tp);
    r.xc = *reinterpret_cast<const uint4*>(xr + t0);
    r.xn = *reinterpret_cast<const uint4*>(xr + tn);
    r.gc = *reinterpret_cast<const uint4*>(gr + t0);
    r.gn = *reinterpret_cast<const uint4*>(gr + tn);
    return r;
  };

  float gw[4] = {0.f, 0.f, 0.f, 0.f}, gb = 0.f;
  if (n > 0) {
    Raw cur = load_raw(0);
    for (int it = 0; it < n; ++it) {
      const Raw nxt = load_raw(it + 1);
      int b, t0;
      where(it, b, t0);
      float xp[P], xc[P], xn[P], gc[P], gn[P];
      unpack16<T, P>(cur.xp, xp); unpack16<T, P>(cur.xc, xc); unpack16<T, P>(cur.xn, xn);
      unpack16<T, P>(cur.gc, gc); unpack16<T, P>(cur.gn, gn);
      const bool first = t0 == 0, last = t0 + P >= L;
      float xw[P + 6], gv[P + 3];
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        xw[i] = first ? 0.f : xp[P - 3 + i];
        xw[P + 3 + i] = last ? 0.f : xn[i];
        gv[P + i] = last ? 0.f : gn[i];
      }
#pragma unroll
      for (int i = 0; i < P; ++i) { xw[3 + i] = xc[i]; gv[i] = gc[i]; }
      float dpre[P + 3];
#pragma unroll
      for (int i = 0; i < P + 3; ++i) {
        float g = gv[i];
        if (p.silu) {
          float pre = bias;
#pragma unroll
          for (int j = 0; j < 4; ++j) pre = fmaf(w4[j], xw[i + j], pre);
          float sg = sigmoid_f(pre);
          g = g * sg * (1.f + pre * (1.f - sg));
        }
        dpre[i] = g;                                           // (positions >= L carry gv = 0)
      }
      float dx[P];
#pragma unroll
      for (int i = 0; i < P; ++i) {
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = fmaf(w4[j], dpre[i + 3 - j], acc);
        dx[i] = acc;
        gb += dpre[i];
#pragma unroll
        for (int j = 0; j < 4; ++j) gw[j] = fmaf(dpre[i], xw[i + j], gw[j]);
      }
      if (dvalid)
        store_items<T, P>(dxg + static_cast<size_t>(b) * p.o_bs + static_cast<size_t>(dc) * L + t0, P, true, dx);
      cur = nxt;
    }
  }
  conv_bwd_reduce(p, sacc, gw, gb, drow, rows_per_blk, dvalid);
}

static int fill_common(ConvArgs& a, int io_dtype, int pack = kPack) {
  int packs = (a.seqlen + pack - 1) / pack;
  int ppr = 16;
  while (ppr < packs && ppr < kConvThreads) ppr <<= 1;
  a.ppr = ppr;
  const int rows_per_blk = ppr >= kConvThreads ? 1 : kConvThreads / ppr;
  const int dblocks = (a.dim + rows_per_blk - 1) / rows_per_blk;
  // enough workgroups to fill 256 CUs several times over, but long batch slices for the reduction
  int bchunk = a.batch;
  while (bchunk > 1 && static_cast<long long>(dblocks) * ((a.batch + bchunk - 1) / bchunk) < 2048) bchunk = (bchunk + 1) / 2;
  a.bchunk = bchunk;
  const size_t esz = io_dtype == SIMAMBA_F32 ? 4 : 2;
  a.vec = (a.seqlen * esz) % 16 == 0;
  return dblocks;
}

}  // namespace simamba

using namespace simamba;

static bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

static int check_conv(const void* x, const float* w, int batch, int dim, int seqlen, int width, int io_dtype) {
  if (!x || !w) return SIMAMBA_E_NULLPTR;
  if (batch < 0 || dim <= 0 || seqlen < 0) return SIMAMBA_E_SHAPE;
  if (width < 2 || width > 4) return SIMAMBA_E_WIDTH;
  if (io_dtype != SIMAMBA_F32 && io_dtype != SIMAMBA_BF16) return SIMAMBA_E_DTYPE;
  return SIMAMBA_OK;
}

extern "C" int simamba_causal_conv1d_fwd(const void* x, const float* w, const float* bias, void* out,
                                         int batch, int dim, int seqlen, int width, int silu,
                                         int io_dtype, long long x_bstride, void* stream) {
  int rc = check_conv(x, w, batch, dim, seqlen, width, io_dtype);
  if (rc) return rc;
  if (!out) return SIMAMBA_E_NULLPTR;
  if (batch == 0 || seqlen == 0) return SIMAMBA_OK;
  ConvArgs a{};
  a.x = x; a.w = w; a.bias = bias; a.out = out;
  a.batch = batch; a.dim = dim; a.seqlen = seqlen; a.width = width; a.silu = silu;
  const int dblocks = fill_common(a, io_dtype);
  a.x_bs = x_bstride ? x_bstride : static_cast<long long>(dim) * seqlen;
  a.o_bs = static_cast<long long>(dim) * seqlen;
  a.vec = a.vec && al16(x) && al16(out) && (a.x_bs * (io_dtype == SIMAMBA_F32 ? 4 : 2)) % 16 == 0;
  dim3 grid(dblocks, (batch + a.bchunk - 1) / a.bchunk);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (io_dtype == SIMAMBA_F32)
    hipLaunchKernelGGL(conv1d_fwd_kernel<float>, grid, dim3(kConvThreads), 0, s, a);
  else
    hipLaunchKernelGGL(conv1d_fwd_kernel<bf16_t>, grid, dim3(kConvThreads), 0, s, a);
  return static_cast<int>(hipGetLastError());
}

extern "C" int simamba_causal_conv1d_bwd(const void* x, const float* w, const float* bias, const void* dout,
                                         void* dx, float* dw, float* dbias, int batch, int dim, int seqlen,
                                         int width, int silu, int io_dtype, long long x_bstride,
                                         long long dx_bstride, void* stream) {
  int rc = check_conv(x, w, batch, dim, seqlen, width, io_dtype);
  if (rc) return rc;
  if (!dout || !dx || !dw) return SIMAMBA_E_NULLPTR;
  hipStream_t s = static_cast<hipStream_t>(stream);
  // the two accumulators are zeroed here; a caller that carves dbias directly behind dw gets one memset node
  const bool joined = dbias == dw + static_cast<size_t>(dim) * width;
  hipError_t e = hipMemsetAsync(dw, 0, sizeof(float) * (static_cast<size_t>(dim) * width + (joined ? dim : 0)), s);
  if (e != hipSuccess) return static_cast<int>(e);
  if (dbias && !joined) {
    e = hipMemsetAsync(dbias, 0, sizeof(float) * dim, s);
    if (e != hipSuccess) return static_cast<int>(e);
  }
  if (batch == 0 || seqlen == 0) return SIMAMBA_OK;
  ConvArgs a{};
  a.x = x; a.w = w; a.bias = bias; a.out = dx; a.dout = dout; a.dw = dw; a.dbias = dbias;
  a.batch = batch; a.dim = dim; a.seqlen = seqlen; a.width = width; a.silu = silu;
  const int dblocks = fill_common(a, io_dtype, io_dtype == SIMAMBA_F32 ? 4 : 8);
  a.x_bs = x_bstride ? x_bstride : static_cast<long long>(dim) * seqlen;
  a.o_bs = dx_bstride ? dx_bstride : static_cast<long long>(dim) * seqlen;
  const long long esz_ = io_dtype == SIMAMBA_F32 ? 4 : 2;
  a.vec = a.vec && al16(x) && al16(dx) && al16(dout) && (a.x_bs * esz_) % 16 == 0 && (a.o_bs * esz_) % 16 == 0;
  dim3 grid(dblocks, (batch + a.bchunk - 1) / a.bchunk);
  const bool fast = a.vec && seqlen % (io_dtype == SIMAMBA_F32 ? 4 : 8) == 0;
  if (io_dtype == SIMAMBA_F32) {
    if (fast) hipLaunchKernelGGL((conv1d_bwd_fast_kernel<float, 4>), grid, dim3(kConvThreads), 0, s, a);
    else hipLaunchKernelGGL((conv1d_bwd_kernel<float, 4>), grid, dim3(kConvThreads), 0, s, a);
  } else {
    if (fast) hipLaunchKernelGGL((conv1d_bwd_fast_kernel<bf16_t, 8>), grid, dim3(kConvThreads), 0, s, a);
    else hipLaunchKernelGGL((conv1d_bwd_kernel<bf16_t, 8>), grid, dim3(kConvThreads), 0, s, a);
  }
  return static_cast<int>(hipGetLastError());
}
