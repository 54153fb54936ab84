// Selective-scan forward for gfx950.  Decomposition: see scan_common.h.
//
// Replaces selective_scan_cuda.fwd of mamba-ssm as reached from the reference's
// models/block.py:72.  Algorithmic HBM bytes: 4*B*D*L*s (u, delta, z, out) + 2*B*N*L*s (B, C)
// + small; the (B_t, C_t) tile is re-read from L2 once per 16*R channels.
#include <cstdlib>
#include "scan_common.h"

namespace simamba {

template <typename T, int kItems>
__global__ __launch_bounds__(kScanThreads) void scan_fwd_kernel(ScanArgs p) {
  constexpr int LC = 16 * kItems;
  constexpr int LDP = LC + 4;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sB = smem;                       // [kMaxState][LDP]
  float* sC = sB + kMaxState * LDP;       // [kMaxState][LDP]
  float* sCarry = sC + kMaxState * LDP;   // [16 * passes][kMaxState]
  float* sMid = sCarry + kRowsPerPass * p.passes * kMaxState;   // kItems == 16 only: state at the chunk's middle

  int tile_id, b;
  xcd_tile(tile_id, b);
  const int tile_base = tile_id * (kRowsPerPass * p.passes);
  const int lane16 = threadIdx.x & 15;
  const int rowslot = threadIdx.x >> 4;   // 0..15: wave*4 + sub-row
  const int L = p.seqlen, D = p.dim, N = p.dstate;
  const T* __restrict__ ug = static_cast<const T*>(p.u);
  const T* __restrict__ dg = static_cast<const T*>(p.delta);
  const T* __restrict__ zg = static_cast<const T*>(p.z);
  T* __restrict__ og = static_cast<T*>(p.out);
  const bool vec = p.vec != 0;
  const bool keep_state = (p.nchunks > 1) || p.x_ckpt || p.last_state;

  const int nchunks_k = (L + LC - 1) / LC;        // chunks of THIS kernel (p.nchunks counts 128-step checkpoints)
  for (int c = 0; c < nchunks_k; ++c) {
    __syncthreads();
    stage_bc<T, LC>(static_cast<const T*>(p.B), static_cast<const T*>(p.C), sB, sC, b, N, L, c, p.bc_bs, p.bc_ns,
                    p.bc_ts);
    __syncthreads();

    for (int r = 0; r < p.passes; ++r) {
      const int slot = r * kRowsPerPass + rowslot;
      const int d = tile_base + slot;
      const bool dvalid = d < D;
      const int dc = dvalid ? d : D - 1;
      const int t0 = c * LC + lane16 * kItems;
      int nvalid = L - t0;
      nvalid = nvalid < 0 ? 0 : (nvalid > kItems ? kItems : nvalid);
      const size_t off = (static_cast<size_t>(b) * D + dc) * L + t0;

      float u[kItems], dl[kItems], zz[kItems], y[kItems], du[kItems];
      load_items<T, kItems>(ug + off, nvalid, vec, u);
      load_items<T, kItems>(dg + off, nvalid, vec, dl);
      if (zg) load_items<T, kItems>(zg + static_cast<size_t>(b) * p.z_bs + static_cast<size_t>(dc) * L + t0, nvalid, vec, zz);

      const float bias = p.delta_bias ? p.delta_bias[dc] : 0.f;
      const float Dd = p.D ? p.D[dc] : 0.f;
      float sumd = 0.f;
#pragma unroll
      for (int i = 0; i < kItems; ++i) {
        float x = dl[i] + bias;
        x = p.softplus ? softplus_f(x) : x;
        x = (i < nvalid) ? x : 0.f;           // padded steps are the identity map
        dl[i] = x;
        sumd += x;
        du[i] = x * u[i];
        y[i] = Dd * u[i];
      }

      float A2[kMaxState];
      load_A_row(p.A + static_cast<size_t>(dc) * N, N, A2);
#pragma unroll
      for (int n = 0; n < kMaxState; ++n) {
        if (n < N) {
          const float A2n = A2[n];
          const float* bp = sB + n * LDP;
          const float* cp = sC + n * LDP;
          float a[kItems], bb[kItems], cc[kItems];
#pragma unroll
          for (int i = 0; i < kItems; i += 4) {
            const int qo = bc_quad<kItems>((lane16 * kItems + i) >> 2) * 4;
            float4 vb = *reinterpret_cast<const float4*>(bp + qo);
            float4 vc = *reinterpret_cast<const float4*>(cp + qo);
            bb[i] = vb.x; bb[i + 1] = vb.y; bb[i + 2] = vb.z; bb[i + 3] = vb.w;
            cc[i] = vc.x; cc[i + 1] = vc.y; cc[i + 2] = vc.z; cc[i + 3] = vc.w;
          }
          float S = 0.f;
#pragma unroll
          for (int i = 0; i < kItems; ++i) {
            a[i] = fast_exp2(dl[i] * A2n);
            bb[i] = du[i] * bb[i];
            S = fmaf(a[i], S, bb[i]);
          }
          float P = fast_exp2(A2n * sumd);
          row_scan_inclusive(P, S);
          float h = row_prev(S, 0.f);
          if (c > 0 || keep_state) {
            const float carry = (c > 0) ? sCarry[slot * kMaxState + n] : 0.f;
            if (c > 0) h = fmaf(row_prev(P, 1.f), carry, h);
            if (keep_state) {
              const float st = fmaf(P, carry, S);       // state after this lane's last step
              if (lane16 == 15) sCarry[slot * kMaxState + n] = st;
              if (kItems == 16 && lane16 == 7) sMid[slot * kMaxState + n] = st;   // 128-step boundary inside the chunk
            }
          }
#pragma unroll
          for (int i = 0; i < kItems; ++i) {
            h = fmaf(a[i], h, bb[i]);
            y[i] = fmaf(cc[i], h, y[i]);
          }
        }
      }

      if (zg) {
#pragma unroll
        for (int i = 0; i < kItems; ++i) y[i] = y[i] * zz[i] * sigmoid_f(zz[i]);
      }
      if (dvalid) store_items<T, kItems>(og + off, nvalid, vec, y);

      if (keep_state) {
        __builtin_amdgcn_wave_barrier();
        if (dvalid && lane16 < N) {
          const float st = sCarry[slot * kMaxState + lane16];
          constexpr int kPer = LC / SIMAMBA_SCAN_CHUNK > 0 ? LC / SIMAMBA_SCAN_CHUNK : 1;   // checkpoints per chunk
          float* ck = p.x_ckpt ? p.x_ckpt + (static_cast<size_t>(b) * D + d) * p.nchunks * N + lane16 : nullptr;
          if (ck) {
            if (kPer == 2) {
              ck[static_cast<size_t>(2 * c) * N] = sMid[slot * kMaxState + lane16];
              if (2 * c + 1 < p.nchunks) ck[static_cast<size_t>(2 * c + 1) * N] = st;
            } else {
              ck[static_cast<size_t>(c) * N] = st;
            }
          }
          if (p.last_state && c == nchunks_k - 1)
            p.last_state[(static_cast<size_t>(b) * D + d) * N + lane16] = st;
        }
      }
    }
  }
}

static size_t fwd_smem_bytes(int kItems, int passes) {
  const int LC = 16 * kItems;
  return sizeof(float) * (2 * kMaxState * (LC + 4) + 2 * kRowsPerPass * passes * kMaxState);
}

template <typename T>
static int launch_fwd(const ScanArgs& a, hipStream_t s) {
  dim3 grid((a.dim + kRowsPerPass * a.passes - 1) / (kRowsPerPass * a.passes), a.batch);
  if (a.seqlen <= 64) {
    hipLaunchKernelGGL((scan_fwd_kernel<T, 4>), grid, dim3(kScanThreads), fwd_smem_bytes(4, a.passes), s, a);
  } else if (a.seqlen <= 128 || !a.long_items) {
    hipLaunchKernelGGL((scan_fwd_kernel<T, 8>), grid, dim3(kScanThreads), fwd_smem_bytes(8, a.passes), s, a);
  } else {
    // 16 steps per lane: the 4 scan steps and the per-lane bookkeeping amortise over twice the work
    hipLaunchKernelGGL((scan_fwd_kernel<T, 16>), grid, dim3(kScanThreads), fwd_smem_bytes(16, a.passes), s, a);
  }
  return static_cast<int>(hipGetLastError());
}

int scan_fwd_seq_dispatch(const void* u, const void* delta, const float* A, const void* B, const void* C, const float* D,
                          const void* z, const float* delta_bias, void* out, float* x_ckpt, int ckpt_step,
                          float* last_state,
                          int batch, int dim, int seqlen, int io_dtype, long long z_bs,
                          long long bc_bs, long long bc_ns, long long bc_ts, int nchunks128, int lpc, hipStream_t s,
                          const void* dt = nullptr, const void* wdt = nullptr, long long dt_bs = 0, long long dt_ts = 0,
                          int dt_rank = 0);
int scan_fwd_seq_bc_mode(const void* B, const void* C, int io_dtype, long long bc_bs, long long bc_ns, long long bc_ts);
int scan_fwd_seq_mix_c4(int batch, int dim);

// Kernel choice when the caller leaves it to the library (variant == SIMAMBA_SCAN_AUTO).  The lanes-per-channel
// kernel (scan_fwd_seq.hip) issues 5 VALU per (row, step, state) against ~7 for the row-scan kernel, but a wave
// covers 64 / lpc whole rows: it needs rows / (64 / lpc) waves to give every one of the 1024 SIMDs its 3 waves.
static int auto_variant(long long rows, int batch, int dim) {
  // Two lanes per channel from 49 152 rows on (1.5 waves per SIMD).  Measured kernel times, fp32, rocprofv3 medians:
  //   (64,768,1024): row-scan 329-334 us, LPC2 307-311 us, LPC4 313-320 us      (49 152 rows)
  //   (64,768, 512): row-scan 179 us,     LPC2 156 us,     LPC4 152 us
  //   (32,768,1024): row-scan 182 us,     LPC2 198 us,     LPC4 207 us          (24 576 rows: too few waves)
  //   (256,768,128): row-scan 165 us,     LPC2 115 us                           (196 608 rows, the north-star shape)
  // Where two lanes per channel leaves half a round of waves over (49 152 rows = 1536 waves) the mixed launch
  // (SIMAMBA_SCAN_MIX, scan_fwd_seq.hip) evens the SIMDs out and is 14 % faster on its own, but 15 % slower in the
  // place the mixer calls it from (right behind 400 MB of freshly written operands): not chosen here.
  (void)batch; (void)dim;
  if (rows >= 48 * 1024) return SIMAMBA_SCAN_LPC2;
  return SIMAMBA_SCAN_ROWSCAN;
}

}  // namespace simamba

using namespace simamba;

extern "C" int simamba_scan_num_chunks(int seqlen) {
  if (seqlen <= 64) return 1;
  return (seqlen + SIMAMBA_SCAN_CHUNK - 1) / SIMAMBA_SCAN_CHUNK;
}

extern "C" int simamba_scan_fwd_auto_variant(int batch, int dim) {
  if (batch <= 0 || dim <= 0) return SIMAMBA_SCAN_ROWSCAN;
  return auto_variant(static_cast<long long>(batch) * dim, batch, dim);
}

// Which checkpoints a forward / backward pair should use when the caller leaves it to the library.  The sequential
// backward (scan_bwd_seq.hip, 16-step checkpoints written by a lanes-per-channel forward) is taken from the row
// count on at which that forward is: below it neither has enough waves for the chip.
extern "C" int simamba_scan_ckpt_step(int batch, int dim, int seqlen, int dstate, int io_dtype) {
  if (batch <= 0 || dim <= 0 || seqlen <= 0) return SIMAMBA_SCAN_CKPT_ROW;
  const long long rows = static_cast<long long>(batch) * dim;
  const int pack = io_dtype == SIMAMBA_F32 ? 4 : 8;
  if (dstate == kMaxState && dim % 64 == 0 && seqlen % pack == 0 && rows * seqlen < (1ll << 30) &&
      auto_variant(rows, batch, dim) != SIMAMBA_SCAN_ROWSCAN)
    return SIMAMBA_SCAN_CKPT_SEQ;
  return SIMAMBA_SCAN_CKPT_ROW;
}

extern "C" long long simamba_scan_ckpt_floats(int batch, int dim, int seqlen, int dstate, int ckpt_step) {
  if (batch <= 0 || dim <= 0 || seqlen <= 0 || dstate <= 0) return 0;
  if (ckpt_step == SIMAMBA_SCAN_CKPT_SEQ)
    return seqlen <= 16 ? 0 : static_cast<long long>(batch) * ((seqlen + 15) / 16) * dim * kMaxState;
  const int nc = simamba_scan_num_chunks(seqlen);
  return nc <= 1 ? 0 : static_cast<long long>(batch) * dim * nc * dstate;
}

static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

extern "C" int simamba_selective_scan_fwd(const void* u, const void* delta, const float* A, const void* B,
                                          const void* C, const float* D, const void* z,
                                          const float* delta_bias, void* out, float* x_ckpt,
                                          float* last_state, int batch, int dim, int seqlen, int dstate,
                                          int io_dtype, int delta_softplus, long long z_bstride,
                                          long long bc_bstride, long long bc_nstride, long long bc_tstride,
                                          int ckpt_step, int variant, void* stream) {
  if (batch < 0 || dim <= 0 || seqlen < 0 || batch > 65535) return SIMAMBA_E_SHAPE;
  if (ckpt_step == 0) ckpt_step = SIMAMBA_SCAN_CKPT_ROW;
  if (ckpt_step != SIMAMBA_SCAN_CKPT_ROW && ckpt_step != SIMAMBA_SCAN_CKPT_SEQ) return SIMAMBA_E_VARIANT;
  if (dstate < 1 || dstate > kMaxState) return SIMAMBA_E_DSTATE;
  if (io_dtype != SIMAMBA_F32 && io_dtype != SIMAMBA_BF16) return SIMAMBA_E_DTYPE;
  if (batch == 0 || seqlen == 0) return SIMAMBA_OK;   // nothing to do (empty tensors carry NULL data)
  if (!u || !delta || !A || !B || !C || !out) return SIMAMBA_E_NULLPTR;
  ScanArgs a{};
  a.u = u; a.delta = delta; a.A = A; a.B = B; a.C = C; a.D = D; a.z = z; a.delta_bias = delta_bias;
  a.out = out; a.x_ckpt = x_ckpt; a.last_state = last_state;
  a.batch = batch; a.dim = dim; a.seqlen = seqlen; a.dstate = dstate;
  a.nchunks = simamba_scan_num_chunks(seqlen);
  a.softplus = delta_softplus;
  const size_t esz = io_dtype == SIMAMBA_F32 ? 4 : 2;
  a.z_bs = z_bstride ? z_bstride : static_cast<long long>(dim) * seqlen;
  if (!bc_bstride && !bc_nstride && !bc_tstride) {
    bc_bstride = static_cast<long long>(dstate) * seqlen; bc_nstride = seqlen; bc_tstride = 1;
  }
  a.bc_bs = bc_bstride; a.bc_ns = bc_nstride; a.bc_ts = bc_tstride;
  a.vec = ((seqlen * esz) % 16 == 0) && aligned16(u) && aligned16(delta) && aligned16(out) &&
          (!z || (aligned16(z) && (a.z_bs * esz) % 16 == 0));
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (variant != SIMAMBA_SCAN_AUTO && variant != SIMAMBA_SCAN_ROWSCAN && variant != SIMAMBA_SCAN_LPC2 &&
      variant != SIMAMBA_SCAN_LPC4 && variant != SIMAMBA_SCAN_MIX)
    return SIMAMBA_E_VARIANT;
  const long long rows = static_cast<long long>(batch) * dim;
  // what the lanes-per-channel kernel assumes: 16 states, softplus on (the only form the reference's mixer uses),
  // pack-aligned rows and B / C, 32-bit byte offsets
  const bool seq_ok = dstate == kMaxState && delta_softplus && a.vec && rows * seqlen < (1ll << 30) &&
                      scan_fwd_seq_bc_mode(B, C, io_dtype, a.bc_bs, a.bc_ns, a.bc_ts) != 0 &&
                      static_cast<long long>(batch) * a.z_bs < (1ll << 30) &&
                      (reinterpret_cast<uintptr_t>(A) & 15u) == 0 && a.bc_ns >= 0 && a.bc_ts >= 0 &&
                      (kMaxState - 1) * a.bc_ns + (seqlen - 1) * a.bc_ts < (1ll << 30);
  int v = variant == SIMAMBA_SCAN_AUTO ? auto_variant(rows, batch, dim) : variant;
  // 16-step checkpoints are written by the lanes-per-channel kernels only (four lanes per channel where two would
  // leave the chip short of waves)
  const bool want_seq_ckpt = x_ckpt && ckpt_step == SIMAMBA_SCAN_CKPT_SEQ;
  if (want_seq_ckpt && v == SIMAMBA_SCAN_ROWSCAN) {
    if (variant != SIMAMBA_SCAN_AUTO) return SIMAMBA_E_VARIANT;
    v = SIMAMBA_SCAN_LPC4;
  }
  if (v != SIMAMBA_SCAN_ROWSCAN && !seq_ok) {
    if (variant != SIMAMBA_SCAN_AUTO || want_seq_ckpt) return SIMAMBA_E_VARIANT;   // a request the shape cannot take
    v = SIMAMBA_SCAN_ROWSCAN;
  }
  if (v != SIMAMBA_SCAN_ROWSCAN)
    return scan_fwd_seq_dispatch(u, delta, A, B, C, D, z, delta_bias, out, x_ckpt, ckpt_step, last_state, batch, dim, seqlen,
                                 io_dtype, a.z_bs, a.bc_bs, a.bc_ns, a.bc_ts, a.nchunks,
                                 v == SIMAMBA_SCAN_MIX ? 6 : v == SIMAMBA_SCAN_LPC2 ? 2 : 4, s);
  // channels per workgroup: amortise the (B_t,C_t) staging, but keep >= ~3 workgroups per CU
  int passes = 4;
  while (passes > 1 && static_cast<long long>(batch) * ((dim + 16 * passes - 1) / (16 * passes)) < 768) passes >>= 1;
  a.passes = passes;
  a.long_items = seqlen >= 768;   // measured: -6 % at L = 1024, neutral at L = 512 (3 instead of 4 waves per SIMD)
  return io_dtype == SIMAMBA_F32 ? launch_fwd<float>(a, s) : launch_fwd<bf16_t>(a, s);
}

// The mixer's scan with delta formed inside the kernel (csrc/scan_fwd_seq.hip, kDt): u = the conv output, xdbl = the
// x_proj output (batch, seqlen, dt_rank + 2 * 16) token-major, wdt = dt_proj.weight (dim, dt_rank) in the I/O type.
extern "C" int simamba_selective_scan_dt_fwd(const void* u, const void* xdbl, const void* wdt, const float* A,
                                             const float* D, const void* z, const float* delta_bias, void* out,
                                             float* x_ckpt, float* last_state, int batch, int dim, int seqlen,
                                             int dstate, int dt_rank, int io_dtype, long long z_bstride,
                                             long long xdbl_bstride, long long xdbl_tstride, int ckpt_step, int variant,
                                             void* stream) {
  if (batch < 0 || dim <= 0 || seqlen < 0 || batch > 65535) return SIMAMBA_E_SHAPE;
  if (dstate != kMaxState) return SIMAMBA_E_DSTATE;
  if (io_dtype != SIMAMBA_F32 && io_dtype != SIMAMBA_BF16) return SIMAMBA_E_DTYPE;
  if (ckpt_step == 0) ckpt_step = SIMAMBA_SCAN_CKPT_ROW;
  if (ckpt_step != SIMAMBA_SCAN_CKPT_ROW && ckpt_step != SIMAMBA_SCAN_CKPT_SEQ) return SIMAMBA_E_VARIANT;
  if (variant != SIMAMBA_SCAN_AUTO && variant != SIMAMBA_SCAN_LPC2 && variant != SIMAMBA_SCAN_LPC4 &&
      variant != SIMAMBA_SCAN_MIX)
    return SIMAMBA_E_VARIANT;                               // the row-scan kernel reads a delta tensor
  const int pack = io_dtype == SIMAMBA_F32 ? 4 : 8;
  if (dt_rank < pack || dt_rank > 24 || dt_rank % pack) return SIMAMBA_E_SHAPE;
  if (batch == 0 || seqlen == 0) return SIMAMBA_OK;
  if (!u || !xdbl || !wdt || !A || !z || !out) return SIMAMBA_E_NULLPTR;
  const size_t esz = io_dtype == SIMAMBA_F32 ? 4 : 2;
  const long long S = dt_rank + 2 * kMaxState;
  const long long xb = xdbl_bstride ? xdbl_bstride : S * seqlen, xt = xdbl_tstride ? xdbl_tstride : S;
  const long long zb = z_bstride ? z_bstride : static_cast<long long>(dim) * seqlen;
  const long long rows = static_cast<long long>(batch) * dim;
  const char* Bp = static_cast<const char*>(xdbl) + static_cast<size_t>(dt_rank) * esz;
  const char* Cp = Bp + kMaxState * esz;
  const bool ok = ((seqlen * esz) % 16 == 0) && aligned16(u) && aligned16(out) && aligned16(z) && aligned16(xdbl) &&
                  aligned16(wdt) && aligned16(A) && (zb * esz) % 16 == 0 && (xb * esz) % 16 == 0 &&
                  (xt * esz) % 16 == 0 && (dt_rank * esz) % 16 == 0 && rows * seqlen < (1ll << 30) &&
                  static_cast<long long>(batch) * zb < (1ll << 30) && static_cast<long long>(batch) * xb < (1ll << 30) &&
                  scan_fwd_seq_bc_mode(Bp, Cp, io_dtype, xb, 1, xt) == 2;
  if (!ok) return SIMAMBA_E_VARIANT;
  int v = variant;
  if (v == SIMAMBA_SCAN_AUTO) {
    v = auto_variant(rows, batch, dim);
    if (v == SIMAMBA_SCAN_ROWSCAN) v = SIMAMBA_SCAN_LPC4;   // too few rows for two lanes per channel
  }
  hipStream_t s = static_cast<hipStream_t>(stream);
  return scan_fwd_seq_dispatch(u, nullptr, A, Bp, Cp, D, z, delta_bias, out, x_ckpt, ckpt_step, last_state, batch, dim,
                               seqlen, io_dtype, zb, xb, 1, xt, simamba_scan_num_chunks(seqlen),
                               v == SIMAMBA_SCAN_MIX ? 6 : v == SIMAMBA_SCAN_LPC2 ? 2 : 4, s,
                               xdbl, wdt, xb, xt, dt_rank);
}
