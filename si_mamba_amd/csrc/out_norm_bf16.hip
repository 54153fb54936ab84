// out_proj -> (+ residual, DropPath-scaled) -> LayerNorm as ONE kernel, bf16 operands, for gfx950 (the
// "y * silu(z) -> out_proj epilogue" / "LN -> in_proj prologue" pair; reference models/block.py:72 then :56-58 of the
// next block, or models/point_mamba.py:257-258 after the last one).
//
//   hidden[b, t, c] = sum_d y[b, d, t] * W[c, d]                      (the mixer's out_proj; y = gated scan output)
//   res             = bf16(hidden) * rowscale[b] + residual[b, t, c]  (fp32; residual NULL: res = bf16(hidden))
//   normed          = LayerNorm(res) * gamma + beta                   (what the NEXT block's in_proj reads)
//
// The out_proj result never goes to memory: under autocast it is a bf16 tensor that the next kernel re-reads in fp32
// (50 MB written + read per layer at the model shape, plus one launch); here a workgroup owns WHOLE token rows -- 128
// tokens x all C channels -- so its accumulators can be rounded to bf16 (the rounding the reference's GEMM output has),
// parked in LDS token-major and normalised row by row with the arithmetic of csrc/add_norm.hip.
//
// Orientation: D[i = c][j = t] = sum_k A[i][k] B[k][j] with A = W (rows c, k = d contiguous in memory: every wave reads
// its own C / 8 channel rows straight from L2 into registers -- no other wave shares them, LDS staging would buy
// nothing) and B[k = d][j = t] = y[d][t], which memory holds t-contiguous: the y tile goes through LDS as loaded
// ([64 d][32 t] sub-images with 64-byte rows) and its B fragments -- 8 consecutive d for one t -- come back through
// ds_read_b64_tr_b16, the hardware's transposing read (tools/tr_probe.hip prints its lane map).  v_mfma_f32_16x16x32_bf16,
// fp32 accumulation; K walks in steps of 64 with the next step's loads in flight; one barrier per step.
//
// Tile: 128 tokens x all C channels per 512-thread workgroup, wave w = channels [w C/8, (w+1) C/8) of all 128 tokens.
// What sets the tile is W: every workgroup streams all of W (C x K bf16, 590 KB at the model shape) from its XCD's L2,
// which delivers ~70 GB/s per CU (MI355X_MICROARCH.md, 'Indexed rows'): at 64 tokens per workgroup that stream alone
// was 45 us of a 85 us main loop at the model shape (measured by loading W once instead: tools/bench_out_norm.py), at
// 128 it is half of that and about the MFMA time.
#include "common.h"

namespace simamba {

typedef float on_f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 on_bf16x8 __attribute__((ext_vector_type(8)));
typedef short on_s16x4 __attribute__((ext_vector_type(4)));

constexpr int kOnThreads = 512;
constexpr int kOnTok = 128;         // tokens per workgroup tile
constexpr int kOnKS = 64;           // d per step
constexpr int kOnImg = kOnKS * 32 + 32;   // elements per [64 d][32 t] sub-image: + 64 B, so that the four sub-images a
                                          // 16-lane write group spans start in different banks

struct OnArgs {
  const uint16_t* y;        // (batch, K, L) bf16
  const uint16_t* w;        // (C, K) bf16
  const float* residual;    // (batch, L, C) fp32 or NULL
  const float* rowscale;    // (batch) fp32 or NULL
  const float* gamma;       // (C) fp32
  const float* beta;        // (C) fp32 or NULL
  float* residual_out;      // (batch, L, C) fp32
  void* normed;             // (batch, L, C) bf16 or fp32
  float* mean;              // (batch * L)
  float* rstd;
  int batch, K, L, C;
  float eps;
};

using on_rsrc_t = __amdgpu_buffer_rsrc_t;
__device__ __forceinline__ on_rsrc_t on_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, static_cast<int>(bytes), 0x00020000);
}
__device__ __forceinline__ uint4 on_bload16(on_rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
__device__ __forceinline__ float on_wave_sum(float v) {
  v = row_allreduce_sum(v);
  v += __shfl_xor(v, 16);
  v += __shfl_xor(v, 32);
  return v;
}

// CB: 16-channel blocks per wave (C = 128 CB); TO: type of `normed`
template <int CB, typename TO>
__global__ __launch_bounds__(kOnThreads, 2) void out_proj_add_ln_kernel(OnArgs p) {
  constexpr int C = 128 * CB;
  constexpr int kHP = C + 4;                                   // staging pitch (bf16): rows 2 dwords apart in the banks
  constexpr int kTB = kOnTok / 16;                             // 16-token blocks
  __shared__ __attribute__((aligned(16))) uint16_t sY[2][4 * kOnImg];         // [buffer][32-token block][d][32 t]
  __shared__ __attribute__((aligned(16))) uint16_t sH[kOnTok * kHP];          // bf16(hidden), token-major
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, q = lane >> 4;
  const int K = p.K, L = p.L;
  const int tps = (L + kOnTok - 1) / kOnTok;
  const int b = static_cast<int>(blockIdx.x) / tps;
  const int t0 = (static_cast<int>(blockIdx.x) - b * tps) * kOnTok;
  const int nk = K / kOnKS;

  // ---- y staging: 64 d x 128 t = 1024 packs of 8 tokens; thread -> (d = tid >> 4 and + 32, t8 = 8 (tid & 15)).  Inside a
  // 64-byte image row the two 16-token halves are swapped for rows with bit 4 of d set, so that the two 16-lane groups
  // of one transposing read pass (rows 16 apart = 1 KiB = the same banks) fall into different halves.
  const int yd = tid >> 4, yt = 8 * (tid & 15);
  const on_rsrc_t rs_y = on_rsrc(p.y + static_cast<size_t>(b) * K * L, static_cast<unsigned>(K) * L * 2u);
  const bool yin = t0 + yt < L;                                // L % 8 == 0
  const unsigned yvoff0 = yin ? (static_cast<unsigned>(yd) * L + t0 + yt) * 2u : 0xfffff000u;
  const unsigned yvoff1 = yin ? (static_cast<unsigned>(yd + 32) * L + t0 + yt) * 2u : 0xfffff000u;
  uint16_t* const ydst0 =
      &sY[0][(yt >> 5) * kOnImg + yd * 32 + (((((yt >> 4) & 1) ^ ((yd >> 4) & 1))) << 4) + (yt & 15)];
  constexpr int kYBuf = 4 * kOnImg;                            // elements per buffer
  const unsigned ystep = static_cast<unsigned>(kOnKS) * L * 2u;

  // ---- W fragments.  The MFMA's k slots may be ANY 8 columns as long as both operands agree, so slot (q, e) of the
  // step's m-th MFMA is d = 64 ks + 16 q + 8 m + e: lane (li, q) then owns 32 CONTIGUOUS bytes of row c per step, the
  // four lanes of a row cover one whole 128-byte line, and the two 16-byte loads of it are issued back to back.
  const uint16_t* wrow[CB];
#pragma unroll
  for (int cb = 0; cb < CB; ++cb)
    wrow[cb] = p.w + static_cast<size_t>(16 * (CB * wave + cb) + li) * K + 16 * q;

  // ---- B fragment addresses (ds_read_b64_tr_b16): the 16-lane group q, lane 4 qq + pp of it supplies (row r0 + qq,
  // columns c0 + 4 pp ..) with r0 = 16 q + 8 m (+ 4 for the second half of the fragment), c0 = the token block's half
  // of the image row; lane i of the group receives column c0 + i of those 4 rows = y[d = 16 q + 8 m + e][t = li]
  const int tqq = (lane >> 2) & 3, tpp = lane & 3;
  const int troff0 = (16 * q + tqq) * 32 + ((q & 1) << 4) + 4 * tpp;          // even token blocks (half 0 ^ (q & 1))
  const int troff1 = (16 * q + tqq) * 32 + (((q & 1) ^ 1) << 4) + 4 * tpp;    // odd token blocks

  on_f32x4 acc[CB][kTB];
#pragma unroll
  for (int cb = 0; cb < CB; ++cb)
#pragma unroll
    for (int tb = 0; tb < kTB; ++tb)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[cb][tb][i] = 0.f;

  // One step ahead: while step ks computes (16 CB MFMAs per wave, two waves per SIMD) the W fragments of step ks + 1 are
  // in flight from L2 and the y packs of step ks + 2 from HBM.  The ring is unrolled by two so that both fragment arrays
  // are named statically.
  struct WF { uint4 f[CB][2]; };
  struct YF { uint4 v[2]; };
  auto load_w = [&](WF& w, int ks_) {                         // unconditional (a step past the end re-reads the last one):
    const int ks = ks_ < nk ? ks_ : nk - 1;                   // a load under `if` keeps the slot's OLD contents alive
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
      for (int m = 0; m < 2; ++m)
        w.f[cb][m] = *reinterpret_cast<const uint4*>(wrow[cb] + ks * kOnKS + 8 * m);
  };
  auto load_y = [&](YF& y, int ks) {                          // out-of-range steps: an out-of-range offset (zeros)
    const unsigned so = static_cast<unsigned>(ks < nk ? ks : 0) * ystep;
    y.v[0] = on_bload16(rs_y, ks < nk ? yvoff0 : 0xfffff000u, so);
    y.v[1] = on_bload16(rs_y, ks < nk ? yvoff1 : 0xfffff000u, so);
  };
  auto store_y = [&](const YF& y, int buf) {
    *reinterpret_cast<uint4*>(ydst0 + buf * kYBuf) = y.v[0];
    *reinterpret_cast<uint4*>(ydst0 + buf * kYBuf + 32 * 32) = y.v[1];
  };
  auto step = [&](const WF& w, int buf) {
    const uint16_t* img = &sY[buf][0];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
#pragma unroll
      for (int th = 0; th < 2; ++th) {                         // four token blocks at a time (16 fragment registers)
        on_bf16x8 bf[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int tb = 4 * th + j;
          const uint16_t* a0 = img + (tb >> 1) * kOnImg + 8 * m * 32 + ((tb & 1) ? troff1 : troff0);
          const on_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (on_s16x4 __attribute__((address_space(3)))*)(a0));
          const on_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (on_s16x4 __attribute__((address_space(3)))*)(a0 + 4 * 32));
          const uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
          bf[j] = __builtin_bit_cast(on_bf16x8, make_uint4(l2.x, l2.y, h2.x, h2.y));
        }
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[cb][4 * th + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                __builtin_bit_cast(on_bf16x8, w.f[cb][m]), bf[j], acc[cb][4 * th + j], 0, 0, 0);
      }
    }
  };

  // Iteration j: [barrier: buffer j & 1 holds step j] store the y packs of step j + 1 (requested one iteration ago) into
  // the other buffer -- last read in iteration j - 1, before the barrier -- request y of step j + 2 and W of step j + 1,
  // compute step j.  (Steps run in pairs: a step past the end multiplies zeros -- its y packs are out-of-range buffer
  // loads -- so the ring needs no early exit, which would fork every fragment array's live range.)
  WF wa, wb;
  YF ya, yb;
  load_y(ya, 0);
  load_w(wa, 0);
  load_y(yb, 1);
  store_y(ya, 0);
#ifndef ON_NK
#define ON_NK nk                            // timing-only A/B builds (tools/build_alt.sh): -DON_NK=2 = 2 K-steps, i.e. the
#endif                                   // epilogue's share; -DON_ROWS=1 below = the main loop's (profiles/r03c_out_norm.txt)
  for (int ks = 0; ks < ON_NK; ks += 2) {
    __syncthreads();
    store_y(yb, 1);
    load_w(wb, ks + 1); load_y(ya, ks + 2);
    step(wa, 0);
    __syncthreads();
    store_y(ya, 0);
    load_w(wa, ks + 2); load_y(yb, ks + 3);
    step(wb, 1);
  }

  // ---- residual rows of this wave's 16 tokens: requested now, all of them, so that their HBM latency runs under the
  // staging below (with two waves per SIMD nothing else would cover it; one row at a time the epilogue cost as much as
  // the whole product)
  constexpr int kRows = kOnTok / 8;
  constexpr int kChunks = (C / 4 + 63) / 64;                   // 4-channel chunks per lane
  constexpr int nch = C / 4;
  float4 rr[kRows][kChunks];
  {
    const bool hasr = p.residual != nullptr;
    const float* rbase = hasr ? p.residual : p.gamma;          // (no residual: any readable address, value unused)
#pragma unroll
    for (int r = 0; r < kRows; ++r) {
      const int t = t0 + kRows * wave + r;
      const size_t row = static_cast<size_t>(b) * L + (t < L ? t : L - 1);
#pragma unroll
      for (int k = 0; k < kChunks; ++k) {
        const int c = lane + 64 * k;
        const size_t off = hasr ? row * C + 4 * (c < nch ? c : 0) : 0;
        rr[r][k] = *reinterpret_cast<const float4*>(rbase + off);
      }
    }
  }

  // ---- accumulators -> bf16 (the out_proj output's rounding), token-major in LDS ---------------------------------
  // register e of acc[cb][tb] holds D[c = 16 (CB wave + cb) + 4 q + e][t = 16 tb + li]
#pragma unroll
  for (int cb = 0; cb < CB; ++cb)
#pragma unroll
    for (int tb = 0; tb < kTB; ++tb) {
      const int c = 16 * (CB * wave + cb) + 4 * q;
      const unsigned lo = static_cast<unsigned>(f32_to_bf16(acc[cb][tb][0])) |
                          (static_cast<unsigned>(f32_to_bf16(acc[cb][tb][1])) << 16);
      const unsigned hi = static_cast<unsigned>(f32_to_bf16(acc[cb][tb][2])) |
                          (static_cast<unsigned>(f32_to_bf16(acc[cb][tb][3])) << 16);
      *reinterpret_cast<uint2*>(&sH[(16 * tb + li) * kHP + c]) = make_uint2(lo, hi);
    }
  __syncthreads();

  // ---- residual add + LayerNorm, one wave per row, 16 rows per wave (the arithmetic of add_ln_fwd_kernel) ----------
  float4 gm[kChunks], bt[kChunks];
#pragma unroll
  for (int k = 0; k < kChunks; ++k) {
    const int c = lane + 64 * k;
    gm[k] = c < nch ? *reinterpret_cast<const float4*>(p.gamma + 4 * c) : make_float4(0, 0, 0, 0);
    bt[k] = (c < nch && p.beta) ? *reinterpret_cast<const float4*>(p.beta + 4 * c) : make_float4(0, 0, 0, 0);
  }
  const float scale = (p.rowscale && p.residual) ? p.rowscale[b] : 1.f;
  const float inv_dim = 1.f / C;
  TO* __restrict__ og = static_cast<TO*>(p.normed);
#ifndef ON_ROWS
#define ON_ROWS kRows
#endif
#pragma unroll
  for (int r = 0; r < ON_ROWS; ++r) {
    const int tl = kRows * wave + r;
    if (t0 + tl < L) {                                         // wave-uniform
      const size_t row = static_cast<size_t>(b) * L + t0 + tl;
      const size_t base = row * C;
      float4 x[kChunks];
      float sum = 0.f;
#pragma unroll
      for (int k = 0; k < kChunks; ++k) {
        const int c = lane + 64 * k;
        if (c < nch) {
          const uint2 hv = *reinterpret_cast<const uint2*>(&sH[tl * kHP + 4 * c]);
          float4 h = make_float4(bf16_to_f32(hv.x & 0xffffu), bf16_to_f32(hv.x >> 16), bf16_to_f32(hv.y & 0xffffu),
                                 bf16_to_f32(hv.y >> 16));
          if (p.residual) {
            h.x = fmaf(h.x, scale, rr[r][k].x); h.y = fmaf(h.y, scale, rr[r][k].y);
            h.z = fmaf(h.z, scale, rr[r][k].z); h.w = fmaf(h.w, scale, rr[r][k].w);
          }
          *reinterpret_cast<float4*>(p.residual_out + base + 4 * c) = h;
          x[k] = h;
          sum += (h.x + h.y) + (h.z + h.w);
        } else {
          x[k] = make_float4(0, 0, 0, 0);
        }
      }
      const float mean = on_wave_sum(sum) * inv_dim;
      float sq = 0.f;
#pragma unroll
      for (int k = 0; k < kChunks; ++k) {
        if (lane + 64 * k < nch) {
          const float a = x[k].x - mean, bb = x[k].y - mean, cc = x[k].z - mean, d = x[k].w - mean;
          sq += (a * a + bb * bb) + (cc * cc + d * d);
        }
      }
      const float rstd = rsqrtf(on_wave_sum(sq) * inv_dim + p.eps);
#pragma unroll
      for (int k = 0; k < kChunks; ++k) {
        const int c = lane + 64 * k;
        if (c < nch) {
          Pack<TO, 4> pk;
          pk.v[0] = from_f32<TO>(fmaf((x[k].x - mean) * rstd, gm[k].x, bt[k].x));
          pk.v[1] = from_f32<TO>(fmaf((x[k].y - mean) * rstd, gm[k].y, bt[k].y));
          pk.v[2] = from_f32<TO>(fmaf((x[k].z - mean) * rstd, gm[k].z, bt[k].z));
          pk.v[3] = from_f32<TO>(fmaf((x[k].w - mean) * rstd, gm[k].w, bt[k].w));
          *reinterpret_cast<Pack<TO, 4>*>(og + base + 4 * c) = pk;
        }
      }
      if (lane == 0) { p.mean[row] = mean; p.rstd[row] = rstd; }
    }
  }
}

template <int CB>
static void on_launch(const OnArgs& a, int out_dtype, hipStream_t s) {
  const int tps = (a.L + kOnTok - 1) / kOnTok;
  dim3 grid(static_cast<unsigned>(a.batch) * tps);
  if (out_dtype == SIMAMBA_BF16)
    hipLaunchKernelGGL((out_proj_add_ln_kernel<CB, bf16_t>), grid, dim3(kOnThreads), 0, s, a);
  else
    hipLaunchKernelGGL((out_proj_add_ln_kernel<CB, float>), grid, dim3(kOnThreads), 0, s, a);
}

}  // namespace simamba

using namespace simamba;

extern "C" int simamba_out_proj_add_ln_fwd(const void* y, const void* w, const float* residual, const float* rowscale,
                                           const float* gamma, const float* beta, float* residual_out, void* normed,
                                           float* mean, float* rstd, int batch, int K, int L, int C, float eps,
                                           int out_dtype, void* stream) {
  if (batch < 0 || K <= 0 || L < 0 || C <= 0) return SIMAMBA_E_SHAPE;
  if (out_dtype != SIMAMBA_F32 && out_dtype != SIMAMBA_BF16) return SIMAMBA_E_DTYPE;
  // whole 128-channel groups (one 16-channel block per wave each; 128 tokens x 512 channels of bf16 staging would not
  // leave LDS for the y tiles), K in 64-steps, 8-token packs, one buffer descriptor
  // per sample
  if (C % 128 || C > 384 || K % kOnKS || L % 8 || static_cast<long long>(K) * L * 2 >= (1LL << 32) - 65536)
    return SIMAMBA_E_SHAPE;
  if (batch == 0 || L == 0) return SIMAMBA_OK;
  if (!y || !w || !gamma || !residual_out || !normed || !mean || !rstd) return SIMAMBA_E_NULLPTR;
  uintptr_t al = reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(w) | reinterpret_cast<uintptr_t>(gamma) |
                 reinterpret_cast<uintptr_t>(residual_out) | reinterpret_cast<uintptr_t>(normed) |
                 reinterpret_cast<uintptr_t>(residual) | reinterpret_cast<uintptr_t>(beta);
  if (al & 15u) return SIMAMBA_E_ALIGN;
  if (static_cast<long long>(batch) * ((L + kOnTok - 1) / kOnTok) > 0x7fffffffLL) return SIMAMBA_E_SHAPE;
  OnArgs a{};
  a.y = static_cast<const uint16_t*>(y); a.w = static_cast<const uint16_t*>(w);
  a.residual = residual; a.rowscale = rowscale; a.gamma = gamma; a.beta = beta;
  a.residual_out = residual_out; a.normed = normed; a.mean = mean; a.rstd = rstd;
  a.batch = batch; a.K = K; a.L = L; a.C = C; a.eps = eps;
  hipStream_t s = static_cast<hipStream_t>(stream);
  switch (C / 128) {
    case 1: on_launch<1>(a, out_dtype, s); break;
    case 2: on_launch<2>(a, out_dtype, s); break;
    default: on_launch<3>(a, out_dtype, s); break;
  }
  return static_cast<int>(hipGetLastError());
}
