// Fused x_proj -> dt_proj of the Mamba mixer on the gfx950 matrix cores (fp32 in, fp32 accumulate, exact fp32:
// v_mfma_f32_32x32x2_f32 computes a k-ordered fmaf chain).
//
// Replaces, inside upstream's mamba_inner_fn (the fast path the reference's mixer takes from models/block.py:72),
//     x_dbl = x_proj.weight (S, D) @ x                       S = dt_rank + 2 * d_state = 56, D = d_inner = 768
//     delta = dt_proj.weight (D, R) @ x_dbl[:R]              R = dt_rank = 24
// with x = the conv output (batch, D, L), L contiguous.  Two library GEMMs become one pass over x:
//   * x_dbl leaves token-major, (batch, L, S): B_t | C_t sit where the scan kernels read them (state stride 1);
//   * delta leaves (batch, D, L), L contiguous, the layout the scan streams.
// One 256-thread workgroup owns 64 consecutive tokens of one sample.
//   phase 1  Y[s, t] = sum_d Wx[s, d] x[d, t]: K = D in steps of 32 through double-buffered LDS tiles; wave w owns the
//            32 x 32 block (s-block w & 1, t-block w >> 1).  A-operand lane map of the 32x32x2 MFMA is
//            A[i = lane & 31][k = lane >> 5]: which two d's an MFMA contracts is free as long as A and B agree, so
//            lanes < 32 take d = 0..15 of the step and lanes >= 32 take d = 16..31: a lane's 16 A values are 64
//            contiguous bytes of a Wx row (4 ds_read_b128; 36-float pitch: conflict-free), its 16 B values one
//            column of the x tile (16 ds_read_b32, lanes on consecutive t).
//   phase 2  delta[d, t] = sum_r Wdt[d, r] Y[r, t], K = R = 24 = 12 MFMAs per 32 x 32 block.  The dt rows of Y go
//            through LDS once (6 KB) and stay in registers as B operands; the A operand, Wdt[d][8g + 4h .. + 3], is
//            three 16-byte loads per lane and d-block straight from global memory (74 KB table, L2 resident), issued
//            one d-block ahead.  The accumulator's C/D layout (col = lane & 31, row = (reg & 3) + 8 (reg >> 2) +
//            4 (lane >> 5)) puts a store instruction on two rows x 128 contiguous bytes of delta.
// Algorithmic bytes per token: read 4 D, write 4 D + 4 S (6.4 KB); flops 2 S D + 2 D R = 123 K: close to the ridge
// of fp32 MFMA (157 TF/s) against HBM.
#include "common.h"

namespace simamba {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kXdtThreads = 256;
constexpr int kTok = 64;          // tokens per workgroup
constexpr int kKS = 32;           // K (= d) per LDS step
constexpr int kWP = kKS + 4;      // pitch of the Wx tile (floats)
constexpr int kSPad = 64;         // S padded to two 32-row blocks

struct XdtArgs {
  const float* x;        // (batch, D, L); with kConv: the conv INPUT (x half of the in_proj output)
  const float* cw;       // kConv: (D, 4) depthwise taps
  const float* cb;       // kConv: (D) bias or NULL
  float* xconv;          // kConv: (batch, D, L) silu(conv(x)) written as a by-product
  const float* wx;       // (S, D)
  const float* wdt;      // (D, R)
  float* xdbl;           // (batch, L, S)
  float* delta;          // (batch, D, L)
  int batch, D, L, S, R;
  long long x_bs;        // batch stride of x (elements)
};

// kConv: the causal depthwise conv1d (width 4) + SiLU that precedes x_proj in the mixer runs while the x tile is
// staged -- same arithmetic as conv1d_fwd_kernel (fmaf chain over the taps, x * sigmoid(x)) -- and its result is both
// the GEMM operand and, stored once, the conv output the scan and the backward read: the separate conv launch and its
// re-read of the activation disappear.  A thread's 4-step pack needs the 3 steps before it: the previous pack of the
// row sits in the neighbouring lane (DPP row_shr:1; a tile row is exactly one 16-lane DPP row) except for the row's
// first pack, which reads the 16 bytes in front of the tile (zeros at t = 0).
template <bool kConv>
__global__ __launch_bounds__(kXdtThreads, 2) void xdt_proj_f32_kernel(XdtArgs p) {
  __shared__ __attribute__((aligned(16))) float sX[2][kKS * kTok];       // [d][t]
  __shared__ __attribute__((aligned(16))) float sW[2][kSPad * kWP];      // [s][d], padded pitch
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, hh = lane >> 5;
  const int sblk = wave & 1, tblk = wave >> 1;
  const int D = p.D, L = p.L, S = p.S, R = p.R;
  // A workgroup walks tiles blockIdx.x, + gridDim.x, ...: the host sizes the grid so that every workgroup gets the same
  // number of tiles and all of them are resident at once (2 - 3 per CU).  With one tile per workgroup 1024 tiles on 768
  // resident slots ran as a full round plus a quarter-full one in which each SIMD's matrix pipe served a single wave.
  const int tps = (L + kTok - 1) / kTok;                  // tiles per sample
  for (int tile = blockIdx.x; tile < p.batch * tps; tile += gridDim.x) {
  const int b = tile / tps;
  const int t0 = (tile - b * tps) * kTok;
  const float* __restrict__ xg = p.x + static_cast<size_t>(b) * p.x_bs;

  // ---- staging identities ---------------------------------------------------------------------------------------
  // x tile: 32 d x 64 t = 512 float4; thread -> (d = tid >> 4 (+16), t4 = 4 (tid & 15))
  const int xd = tid >> 4, xt = 4 * (tid & 15);
  const bool xok = t0 + xt < L;                           // L % 4 == 0: a pack is all in or all out
  // Wx tile: 64 s x 32 d = 512 float4; thread -> (s = tid >> 3 (+32), d4 = 4 (tid & 7))
  const int ws = tid >> 3, wd = 4 * (tid & 7);
  float4 rx[2], rw[2], rh[2], rt[2];
  float rb[2];
  auto gload = [&](int k0) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int d = k0 + xd + 16 * j;
      rx[j] = xok ? *reinterpret_cast<const float4*>(xg + static_cast<size_t>(d) * L + t0 + xt)
                  : make_float4(0.f, 0.f, 0.f, 0.f);
      if (kConv) {
        // the pack in front of the tile (used by the row's first lane only; one address per 16-lane row) and the taps
        rh[j] = t0 > 0 ? *reinterpret_cast<const float4*>(xg + static_cast<size_t>(d) * L + t0 - 4)
                       : make_float4(0.f, 0.f, 0.f, 0.f);
        rt[j] = *reinterpret_cast<const float4*>(p.cw + static_cast<size_t>(d) * 4);
        rb[j] = p.cb ? p.cb[d] : 0.f;
      }
      const int s = ws + 32 * j;
      rw[j] = s < S ? *reinterpret_cast<const float4*>(p.wx + static_cast<size_t>(s) * D + k0 + wd)
                    : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto lstore = [&](int buf, int k0) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      float4 v = rx[j];
      if (kConv) {
        // previous pack of the row: the neighbouring lane's current pack (lane16 == 0: the halo load)
        // The three DPP reads run with EVERY lane active and are pinned in front of the select: a DPP read of a lane
        // that is masked off returns the `old` operand, and hipcc, left alone, predicates the false arm of a ternary
        // (lane 1 of every row then read zeros from the masked-off lane 0: t0 + 4 .. 6 of every tile came out wrong).
        const bool first = (tid & 15) == 0;
        float q1 = dpp<DPP_ROW_SHR + 1>(0.f, rx[j].y);
        float q2 = dpp<DPP_ROW_SHR + 1>(0.f, rx[j].z);
        float q3 = dpp<DPP_ROW_SHR + 1>(0.f, rx[j].w);
        asm volatile("" : "+v"(q1), "+v"(q2), "+v"(q3));
        const float p1 = first ? rh[j].y : q1;
        const float p2 = first ? rh[j].z : q2;
        const float p3 = first ? rh[j].w : q3;
        const float win[7] = {p1, p2, p3, rx[j].x, rx[j].y, rx[j].z, rx[j].w};
        const float w4[4] = {rt[j].x, rt[j].y, rt[j].z, rt[j].w};
        float o[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float acc = rb[j];
#pragma unroll
          for (int q = 0; q < 4; ++q) acc = fmaf(w4[q], win[i + q], acc);
          o[i] = acc * sigmoid_f(acc);
        }
        v = make_float4(o[0], o[1], o[2], o[3]);
        if (xok)
          *reinterpret_cast<float4*>(p.xconv + (static_cast<size_t>(b) * D + k0 + xd + 16 * j) * L + t0 + xt) = v;
        if (!xok) v = make_float4(0.f, 0.f, 0.f, 0.f);
      }
      *reinterpret_cast<float4*>(&sX[buf][(xd + 16 * j) * kTok + xt]) = v;
      *reinterpret_cast<float4*>(&sW[buf][(ws + 32 * j) * kWP + wd]) = rw[j];
    }
  };

  // ---- phase 1 --------------------------------------------------------------------------------------------------
  // Double-buffered LDS tiles, the next step's global loads in flight under this step's 16 MFMAs, one barrier per
  // step.  (A register-level pipeline that also reads the next step's operands under the MFMAs and keeps three
  // steps of global loads in flight measured no faster, 127 us against 120 us: three co-resident workgroups per CU
  // already cover those latencies.)
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  const int nk = D / kKS;
  gload(0);
  lstore(0, 0);
  __syncthreads();
  for (int ks = 0; ks < nk; ++ks) {
    const int buf = ks & 1;
    if (ks + 1 < nk) gload((ks + 1) * kKS);
    const float* aw = &sW[buf][(sblk * 32 + li) * kWP + 16 * hh];
    const float* bx = &sX[buf][(16 * hh) * kTok + tblk * 32 + li];
    float av[16], bv[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 v = *reinterpret_cast<const float4*>(aw + 4 * q);
      av[4 * q] = v.x; av[4 * q + 1] = v.y; av[4 * q + 2] = v.z; av[4 * q + 3] = v.w;
    }
#pragma unroll
    for (int m = 0; m < 16; ++m) bv[m] = bx[m * kTok];
#pragma unroll
    for (int m = 0; m < 16; ++m) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m], bv[m], acc, 0, 0, 0);
    if (ks + 1 < nk) lstore(buf ^ 1, (ks + 1) * kKS);
    __syncthreads();
  }

  // ---- x_dbl out (token-major) and the dt rows into LDS ----------------------------------------------------------
  // accumulator register r holds Y[s = sblk*32 + (r & 3) + 8 (r >> 2) + 4 hh][t = tblk*32 + li]
  float* sDt = &sX[0][0];                                  // [R <= 32][64 t]; the phase-1 tiles are dead
  {
    const int t = t0 + tblk * 32 + li;
    float* row = p.xdbl + (static_cast<size_t>(b) * L + t) * S;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int s = sblk * 32 + 8 * g + 4 * hh;
      if (t < L && s < S)                                  // S % 4 == 0: four states are all in or all out
        *reinterpret_cast<float4*>(row + s) = make_float4(acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]);
      if (sblk == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int r = 8 * g + 4 * hh + i;
          if (r < R) sDt[r * kTok + tblk * 32 + li] = acc[4 * g + i];
        }
      }
    }
  }
  __syncthreads();

  // ---- phase 2: delta = Wdt @ dt --------------------------------------------------------------------------------
  // B operands: MFMA m contracts r = (m & 3) + 8 (m >> 2) + 4 hh; rows >= R contribute zeros.  A wave keeps the dt
  // rows of both 32-token blocks in registers and takes the d-blocks w, w + 4, ...: every Wdt row is loaded once per
  // workgroup.  (Measured against giving a wave one token block and twice the d-blocks -- 90 VGPRs, four workgroups
  // per CU instead of three: 133 us instead of 120 us at (64, 768, 1024).)
  constexpr int kM2 = 12;                                  // R <= 24
  float bdt[2][kM2];
#pragma unroll
  for (int tb = 0; tb < 2; ++tb)
#pragma unroll
    for (int m = 0; m < kM2; ++m) {
      const int r = (m & 3) + 8 * (m >> 2) + 4 * hh;
      bdt[tb][m] = r < R ? sDt[r * kTok + tb * 32 + li] : 0.f;
    }
  const int ndb = D / 32;                                  // d-blocks of 32 channels; wave w takes w, w + 4, ...
  float4 wa[3], wn[3];
  auto wload = [&](int db, float4 (&dst)[3]) {
    const float* row = p.wdt + static_cast<size_t>(db * 32 + li) * R + 4 * hh;
#pragma unroll
    for (int g = 0; g < 3; ++g)
      dst[g] = (8 * g + 4 * hh < R) ? *reinterpret_cast<const float4*>(row + 8 * g) : make_float4(0.f, 0.f, 0.f, 0.f);
  };
  if (wave < ndb) wload(wave, wa);
  for (int db = wave; db < ndb; db += 4) {
    if (db + 4 < ndb) wload(db + 4, wn);
    const float a2[kM2] = {wa[0].x, wa[0].y, wa[0].z, wa[0].w, wa[1].x, wa[1].y, wa[1].z, wa[1].w,
                           wa[2].x, wa[2].y, wa[2].z, wa[2].w};
#pragma unroll
    for (int tb = 0; tb < 2; ++tb) {
      f32x16 o;
#pragma unroll
      for (int i = 0; i < 16; ++i) o[i] = 0.f;
#pragma unroll
      for (int m = 0; m < kM2; ++m) o = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[m], bdt[tb][m], o, 0, 0, 0);
      const int t = t0 + tb * 32 + li;
      if (t < L) {
        float* dst = p.delta + (static_cast<size_t>(b) * D + db * 32 + 4 * hh) * L + t;
#pragma unroll
        for (int r = 0; r < 16; ++r) dst[static_cast<size_t>((r & 3) + 8 * (r >> 2)) * L] = o[r];
      }
    }
#pragma unroll
    for (int g = 0; g < 3; ++g) wa[g] = wn[g];
  }
  __syncthreads();                                         // sDt (= the first x tile buffer) is refilled by the next tile
  }
}

}  // namespace simamba

using namespace simamba;

static int xdt_launch(const void* x, const float* cw, const float* cb, const float* wx, const float* wdt, void* xconv,
                      void* xdbl, void* delta, int batch, int D, int L, int S, int R, int io_dtype,
                      long long x_bstride, bool conv, void* stream) {
  if (batch < 0 || D <= 0 || L < 0 || batch > 65535) return SIMAMBA_E_SHAPE;
  if (io_dtype != SIMAMBA_F32) return SIMAMBA_E_DTYPE;
  if (D % 32 || L % 4 || S % 4 || S > kSPad || S < R || R % 4 || R > 24 || R < 4) return SIMAMBA_E_SHAPE;
  if (batch == 0 || L == 0) return SIMAMBA_OK;
  if (!x || !wx || !wdt || !xdbl || !delta || (conv && (!cw || !xconv))) return SIMAMBA_E_NULLPTR;
  uintptr_t al = reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(wx) | reinterpret_cast<uintptr_t>(wdt) |
                 reinterpret_cast<uintptr_t>(xdbl) | reinterpret_cast<uintptr_t>(delta);
  if (conv) al |= reinterpret_cast<uintptr_t>(cw) | reinterpret_cast<uintptr_t>(xconv);
  if (al & 15u) return SIMAMBA_E_ALIGN;
  XdtArgs a{};
  a.x = static_cast<const float*>(x); a.wx = wx; a.wdt = wdt; a.cw = cw; a.cb = cb;
  a.xconv = static_cast<float*>(xconv);
  a.xdbl = static_cast<float*>(xdbl); a.delta = static_cast<float*>(delta);
  a.batch = batch; a.D = D; a.L = L; a.S = S; a.R = R;
  a.x_bs = x_bstride ? x_bstride : static_cast<long long>(D) * L;
  if (a.x_bs % 4) return SIMAMBA_E_ALIGN;
  // grid: every workgroup the same number of tiles, all workgroups resident together (3 fit per CU, 256 CUs)
  const long long ntiles = static_cast<long long>(batch) * ((L + kTok - 1) / kTok);
  long long g = ntiles;
  if (ntiles > 768) {
    long long per = (ntiles + 767) / 768;                  // tiles per workgroup at full residency
    while (ntiles % per) ++per;                            // ... evened out (per divides ntiles)
    g = ntiles / per;
  }
  dim3 grid(static_cast<unsigned>(g));
  if (conv)
    hipLaunchKernelGGL(xdt_proj_f32_kernel<true>, grid, dim3(kXdtThreads), 0, static_cast<hipStream_t>(stream), a);
  else
    hipLaunchKernelGGL(xdt_proj_f32_kernel<false>, grid, dim3(kXdtThreads), 0, static_cast<hipStream_t>(stream), a);
  return static_cast<int>(hipGetLastError());
}

// fp32 only (bf16 mixers keep the library GEMMs); D % 32 == 0, L % 4 == 0, S % 4 == 0, S <= 64, R % 4 == 0, R <= 24.
extern "C" int simamba_xdt_proj_fwd(const void* x, const float* wx, const float* wdt, void* xdbl, void* delta,
                                    int batch, int D, int L, int S, int R, int io_dtype, long long x_bstride,
                                    void* stream) {
  return xdt_launch(x, nullptr, nullptr, wx, wdt, nullptr, xdbl, delta, batch, D, L, S, R, io_dtype, x_bstride, false,
                    stream);
}

// The same with the mixer's causal depthwise conv1d (width 4, + SiLU) applied to x on the way in; xconv receives it.
extern "C" int simamba_conv_xdt_proj_fwd(const void* x, const float* cw, const float* cb, const float* wx,
                                         const float* wdt, void* xconv, void* xdbl, void* delta, int batch, int D, int L,
                                         int S, int R, int io_dtype, long long x_bstride, void* stream) {
  return xdt_launch(x, cw, cb, wx, wdt, xconv, xdbl, delta, batch, D, L, S, R, io_dtype, x_bstride, true, stream);
}
