// Selective-scan backward for gfx950.  Decomposition: see scan_common.h.
//
// Replaces selective_scan_cuda.bwd of mamba-ssm (autograd of the call at the reference's
// models/block.py:72).  Per (row, state n) the kernel re-runs the forward scan to get h_t, runs
// the adjoint scan g_t = C_t dy_t + a_{t+1} g_{t+1} right-to-left over the same 16 lanes, and
// forms all gradients in registers.  Reductions:
//   dB, dC  (sum over channels): v_permlane32_swap/v_permlane16_swap transposing reduction over
//           the wave's 4 channel rows -> ds_add_f32 into an LDS tile shared by the workgroup's
//           16*R channels -> one coalesced float-atomic flush per workgroup and chunk;
//   dA, dD, ddelta_bias (sum over time and batch): DPP row all-reduce -> one atomic per row.
// Algorithmic HBM bytes: 7*B*D*L*s + 2*B*N*L*(s+4) + small.
#include "scan_common.h"
#include "scan_xlane.h"

namespace simamba {

template <typename T, int kItems>
__global__ __launch_bounds__(kScanThreads, 2) void scan_bwd_kernel(ScanArgs p) {
  constexpr int LC = 16 * kItems;
  constexpr int LDP = LC + 4;
  constexpr int KH = kItems / 2;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sB = smem;                          // [kMaxState][LDP]
  float* sC = sB + kMaxState * LDP;          // [kMaxState][LDP]
  float* sAcc = sC + kMaxState * LDP;        // [2][kMaxState][LC]   dB / dC of this chunk (cross-wave sum)
  float* sG = sAcc + 2 * kMaxState * LC;     // [16*passes][kMaxState] adjoint state entering from the right
  float* sDf = sG + kRowsPerPass * p.passes * kMaxState;  // [16*passes] delta of the next chunk's first step
  float* sCk = sDf + kRowsPerPass * p.passes;             // [16 rows][kMaxState] chunk-start states of this pass

  int tile_id, b;
  xcd_tile(tile_id, b);
  const int tile_base = tile_id * (kRowsPerPass * p.passes);
  const int lane16 = threadIdx.x & 15;
  const int rowslot = threadIdx.x >> 4;
  const int wave = threadIdx.x >> 6;
  const int sub = rowslot & 3;               // channel row inside the wave
  const int L = p.seqlen, D = p.dim, N = p.dstate;
  const T* __restrict__ ug = static_cast<const T*>(p.u);
  const T* __restrict__ dg = static_cast<const T*>(p.delta);
  const T* __restrict__ zg = static_cast<const T*>(p.z);
  const T* __restrict__ gg = static_cast<const T*>(p.dout);
  T* __restrict__ dug = static_cast<T*>(p.du);
  T* __restrict__ ddg = static_cast<T*>(p.ddelta);
  T* __restrict__ dzg = static_cast<T*>(p.dz);
  const bool vec = p.vec != 0;

  for (int c = p.nchunks - 1; c >= 0; --c) {
    __syncthreads();
    stage_bc<T, LC>(static_cast<const T*>(p.B), static_cast<const T*>(p.C), sB, sC, b, N, L, c, p.bc_bs, p.bc_ns,
                    p.bc_ts);
    __syncthreads();
    const bool last_chunk = (c == p.nchunks - 1);

    // dB / dC partial sums of this wave over its passes: after the transposing reduction, row `sub` of the
    // wave owns tensor (sub >> 1) and items [(sub & 1) * KH, +KH) of every state n.
    float acc[kMaxState][KH];
#pragma unroll
    for (int n = 0; n < kMaxState; ++n)
#pragma unroll
      for (int i = 0; i < KH; ++i) acc[n][i] = 0.f;

    for (int r = 0; r < p.passes; ++r) {
      const int slot = r * kRowsPerPass + rowslot;
      const int d = tile_base + slot;
      const bool dvalid = d < D;
      const int dc = dvalid ? d : D - 1;
      const int t0 = c * LC + lane16 * kItems;
      int nvalid = L - t0;
      nvalid = nvalid < 0 ? 0 : (nvalid > kItems ? kItems : nvalid);
      const size_t off = (static_cast<size_t>(b) * D + dc) * L + t0;

      // dy = dout * silu(z) feeds the adjoint scan; dzw = dout * silu'(z) is kept for the epilogue
      // (dz = dzw * y_pre), so neither dout nor z stays in registers through the state loop
      float u[kItems], dl[kItems], dy[kItems], dzw[kItems];
      float ypre[kItems], dxs[kItems], dda[kItems], du[kItems];
      load_items<T, kItems>(ug + off, nvalid, vec, u);
      load_items<T, kItems>(dg + off, nvalid, vec, dl);
      load_items<T, kItems>(gg + off, nvalid, vec, dy);
      const size_t zoff = static_cast<size_t>(dc) * L + t0;
      if (zg) load_items<T, kItems>(zg + static_cast<size_t>(b) * p.z_bs + zoff, nvalid, vec, dzw);
      float A2[kMaxState];
      load_A_row(p.A + static_cast<size_t>(dc) * N, N, A2);

      const float bias = p.delta_bias ? p.delta_bias[dc] : 0.f;
      const float Dd = p.D ? p.D[dc] : 0.f;
      float sumd = 0.f;
#pragma unroll
      for (int i = 0; i < kItems; ++i) {
        float x = dl[i] + bias;
        x = p.softplus ? softplus_f(x) : x;
        const bool ok = i < nvalid;
        x = ok ? x : 0.f;
        dl[i] = x;
        sumd += x;
        du[i] = x * u[i];
        const float go = (ok && dvalid) ? dy[i] : 0.f;
        if (zg) {
          const float z = dzw[i], sg = sigmoid_f(z);
          dy[i] = go * z * sg;
          dzw[i] = go * sg * (1.f + z * (1.f - sg));
        } else {
          dy[i] = go;
        }
        ypre[i] = Dd * u[i];
        dxs[i] = 0.f;
        dda[i] = 0.f;
      }
      // delta of the step right after this lane's last one (next lane, or next chunk for lane 15)
      const float dnext = row_next(dl[0], last_chunk ? 0.f : sDf[slot]);
      // chunk-start state h_{t0-1}[n] from the forward's checkpoints: lane n of the row fetches entry n (one
      // coalesced 64-byte read per row, issued with the tile loads) and parks it in LDS; the state loop then
      // reads it back row-broadcast.  (Reading x_ckpt inside the state loop cost one exposed global-load
      // round trip per state: s_waitcnt vmcnt(0) in the middle of the loop.)
      {
        const float* ck = p.x_ckpt + ((static_cast<size_t>(b) * D + dc) * p.nchunks + (c > 0 ? c - 1 : 0)) * N;
        sCk[rowslot * kMaxState + lane16] = (c > 0 && lane16 < N) ? ck[lane16] : 0.f;
      }
      float dAlane = 0.f;   // lane n of the row ends up holding dA[d][n] of this chunk

#pragma unroll
      for (int n = 0; n < kMaxState; ++n) {
        // (the runtime guard doubles as a boundary between the unrolled states: compiled away for dstate == 16,
        //  hipcc interleaves all 16 states and spills 3 KB per lane)
        if (n < N) {
          const float A2n = A2[n];
          const float* bp = sB + n * LDP;
          const float* cp = sC + n * LDP;
          float a[kItems], bw[kItems], cc[kItems], hp[kItems];
#pragma unroll
          for (int i = 0; i < kItems; i += 4) {
            const int qo = bc_quad<kItems>((lane16 * kItems + i) >> 2) * 4;
            float4 vb = *reinterpret_cast<const float4*>(bp + qo);
            float4 vc = *reinterpret_cast<const float4*>(cp + qo);
            bw[i] = vb.x; bw[i + 1] = vb.y; bw[i + 2] = vb.z; bw[i + 3] = vb.w;
            cc[i] = vc.x; cc[i + 1] = vc.y; cc[i + 2] = vc.z; cc[i + 3] = vc.w;
          }
          // ---- forward re-scan: h_t -----------------------------------------------------
          float S = 0.f;
#pragma unroll
          for (int i = 0; i < kItems; ++i) {
            a[i] = fast_exp2(dl[i] * A2n);
            S = fmaf(a[i], S, du[i] * bw[i]);
          }
          float P = fast_exp2(A2n * sumd);
          row_scan_inclusive(P, S);
          float h = row_prev(S, 0.f);
          {
            const float hin = sCk[rowslot * kMaxState + n];     // zero-filled when there is no chunk to the left
            h = fmaf(row_prev(P, 1.f), hin, h);
          }
          float red[2 * kItems];   // [0,K): dB terms, [K,2K): dC terms
#pragma unroll
          for (int i = 0; i < kItems; ++i) {
            hp[i] = h;
            h = fmaf(a[i], h, du[i] * bw[i]);
            ypre[i] = fmaf(cc[i], h, ypre[i]);
            red[kItems + i] = dy[i] * h;
            cc[i] = cc[i] * dy[i];          // c_t = C_t * dy_t, source term of the adjoint scan
          }
          // ---- adjoint scan, right to left: g_t = c_t + a_{t+1} g_{t+1} -----------------
          const float anext = fast_exp2(dnext * A2n);
          float G = 0.f;
#pragma unroll
          for (int i = kItems - 1; i >= 0; --i) {
            const float al = (i == kItems - 1) ? anext : a[i + 1];
            G = fmaf(al, G, cc[i]);
          }
          float Q = fast_exp2(A2n * (sumd - dl[0] + dnext));
          row_scan_inclusive_rev(Q, G);
          float g = row_next(G, 0.f);
          {
            // adjoint state entering from the chunk on the right (0 for the last chunk): one straight-line
            // sequence for every chunk position, the only predicated op is the LDS store of the carry
            const float gl = sG[slot * kMaxState + n];
            const float gcar = last_chunk ? 0.f : gl;
            g = fmaf(row_next(Q, 1.f), gcar, g);
            if (c > 0 && lane16 == 0) sG[slot * kMaxState + n] = fmaf(Q, gcar, G);
          }
          float dAacc = 0.f;
#pragma unroll
          for (int i = kItems - 1; i >= 0; --i) {
            const float al = (i == kItems - 1) ? anext : a[i + 1];
            g = fmaf(al, g, cc[i]);
            red[i] = g * du[i];
            dxs[i] = fmaf(g, bw[i], dxs[i]);
            const float q = g * a[i] * hp[i];
            dda[i] = fmaf(A2n, q, dda[i]);
            dAacc = fmaf(dl[i], q, dAacc);
          }
          dAacc = row_allreduce_sum(dAacc);
          dAlane = (lane16 == n) ? dAacc : dAlane;
          // ---- dB / dC: transposing reduction over the wave's 4 channel rows ---------------
          swap32_stage<kItems>(red, red + kItems);
          swap16_stage<KH>(red, red + KH);
#pragma unroll
          for (int i = 0; i < KH; ++i) acc[n][i] += red[i];
        }
      }

      // ---- per-timestep gradients ---------------------------------------------------------
      float ddl[kItems], dzv[kItems];
      float sD = 0.f, sBias = 0.f;
#pragma unroll
      for (int i = 0; i < kItems; ++i) {
        // d softplus(x)/dx = sigmoid(x) = 1 - exp(-softplus(x)); padded steps carry no gradient
        const float sgm = p.softplus ? (1.f - fast_exp2(-dl[i] * kLog2e)) : 1.f;
        const float gd = fmaf(u[i], dxs[i], kLn2 * dda[i]) * sgm;
        ddl[i] = gd;
        sBias += gd;
        sD = fmaf(dy[i], u[i], sD);
        du[i] = fmaf(dl[i], dxs[i], Dd * dy[i]);
        if (zg) dzv[i] = dzw[i] * ypre[i];
      }
      if (dvalid) {
        store_items<T, kItems>(dug + off, nvalid, vec, du);
        store_items<T, kItems>(ddg + off, nvalid, vec, ddl);
        if (zg) store_items<T, kItems>(dzg + static_cast<size_t>(b) * p.dz_bs + zoff, nvalid, vec, dzv);
      }
      sD = row_allreduce_sum(sD);
      sBias = row_allreduce_sum(sBias);
      if (dvalid) {
        if (lane16 < N) atomicAdd(&p.dA[static_cast<size_t>(d) * N + lane16], dAlane);
        if (lane16 == 0 && p.dD) atomicAdd(&p.dD[d], sD);
        if (lane16 == 1 && p.ddelta_bias) atomicAdd(&p.ddelta_bias[d], sBias);
      }
      // hand this chunk's first delta to the chunk on the left
      if (c > 0 && lane16 == 0) sDf[slot] = dl[0];
    }

    // ---- cross-wave sum of the dB / dC partials through one LDS tile, then one atomic flush -----
    float* accp = sAcc + (sub >> 1) * kMaxState * LC + lane16 * kItems + (sub & 1) * KH;
    for (int w = 0; w < kScanThreads / 64; ++w) {
      if (wave == w) {
#pragma unroll
        for (int n = 0; n < kMaxState; ++n) {
          if (n < N) {
#pragma unroll
            for (int i = 0; i < KH; ++i) {
              const float prev = (w == 0) ? 0.f : accp[n * LC + i];
              accp[n * LC + i] = prev + acc[n][i];
            }
          }
        }
      }
      __syncthreads();
    }
    for (int i = threadIdx.x; i < 2 * N * LC; i += kScanThreads) {
      const int tensor = i / (N * LC);
      const int rem = i - tensor * (N * LC);
      const int n = rem / LC, t = rem - n * LC;
      const int gt = c * LC + t;
      if (gt < L) {
        float* dst = (tensor == 0 ? p.dB : p.dC) + (static_cast<size_t>(b) * N + n) * L + gt;
        atomicAdd(dst, sAcc[(tensor * kMaxState + n) * LC + t]);
      }
    }
  }
}

static size_t bwd_smem_bytes(int kItems, int passes) {
  const int LC = 16 * kItems;
  return sizeof(float) * (2 * kMaxState * (LC + 4) + 2 * kMaxState * LC + kRowsPerPass * passes * kMaxState +
                          kRowsPerPass * passes + kRowsPerPass * kMaxState);
}

template <typename T>
static int launch_bwd(const ScanArgs& a, hipStream_t s) {
  dim3 grid((a.dim + kRowsPerPass * a.passes - 1) / (kRowsPerPass * a.passes), a.batch);
  if (a.seqlen <= 64) {
    hipLaunchKernelGGL((scan_bwd_kernel<T, 4>), grid, dim3(kScanThreads), bwd_smem_bytes(4, a.passes), s, a);
  } else {
    hipLaunchKernelGGL((scan_bwd_kernel<T, 8>), grid, dim3(kScanThreads), bwd_smem_bytes(8, a.passes), s, a);
  }
  return static_cast<int>(hipGetLastError());
}

struct BwdSeqArgs;
bool scan_bwd_seq_ok(int batch, int dim, int seqlen, int dstate, int softplus, int vec, long long z_bs, long long dz_bs,
                     bool has_z, int bc_mode, long long bc_ns, long long bc_ts);
int scan_bwd_seq_dispatch(const ScanArgs& a, int io_dtype, int bc_mode, hipStream_t s, const void* dt = nullptr,
                          const void* wdt = nullptr, long long dt_bs = 0, long long dt_ts = 0, int dt_rank = 0);
int scan_fwd_seq_bc_mode(const void* B, const void* C, int io_dtype, long long bc_bs, long long bc_ns, long long bc_ts);

}  // namespace simamba

using namespace simamba;

static bool aligned16b(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// dt != NULL: delta is formed inside the (sequential) kernel from the x_proj output; `delta` is then NULL
static int scan_bwd_impl(const void* u, const void* delta, const float* A, const void* B,
                         const void* C, const float* D, const void* z,
                         const float* delta_bias, const void* dout, const float* x_ckpt,
                         void* du, void* ddelta, float* dA, float* dB, float* dC, float* dD,
                         void* dz, float* ddelta_bias, int batch, int dim, int seqlen,
                         int dstate, int io_dtype, int delta_softplus, long long z_bstride,
                         long long dz_bstride, long long bc_bstride, long long bc_nstride,
                         long long bc_tstride, int ckpt_step, void* stream, const void* dt, const void* wdt,
                         long long dt_bs, long long dt_ts, int dt_rank) {
  if (batch < 0 || dim <= 0 || seqlen < 0 || batch > 65535) return SIMAMBA_E_SHAPE;
  if (ckpt_step == 0) ckpt_step = SIMAMBA_SCAN_CKPT_ROW;
  if (ckpt_step != SIMAMBA_SCAN_CKPT_ROW && ckpt_step != SIMAMBA_SCAN_CKPT_SEQ) return SIMAMBA_E_VARIANT;
  if (dstate < 1 || dstate > kMaxState) return SIMAMBA_E_DSTATE;
  if (io_dtype != SIMAMBA_F32 && io_dtype != SIMAMBA_BF16) return SIMAMBA_E_DTYPE;
  if (!A || !dA) return SIMAMBA_E_NULLPTR;
  if (batch > 0 && seqlen > 0) {
    if (!u || (!delta && !dt) || !B || !C || !dout || !du || !ddelta || !dB || !dC) return SIMAMBA_E_NULLPTR;
    if ((z != nullptr) != (dz != nullptr)) return SIMAMBA_E_NULLPTR;
  }
  const int nchunks = simamba_scan_num_chunks(seqlen);
  if (ckpt_step == SIMAMBA_SCAN_CKPT_SEQ ? (seqlen > 16 && !x_ckpt) : (nchunks > 1 && !x_ckpt)) return SIMAMBA_E_NULLPTR;
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipError_t e;
  // The five accumulators are zeroed here (the kernel adds into them).  Never a byte outside the five spans:
  // spans that are EXACTLY adjacent in memory (the next one starts where the previous one ends; the Python
  // callers carve them that way out of one allocation) are merged into one memset node, anything else --
  // separately allocated buffers, with whatever lives between them -- gets a memset of its own.
  {
    const size_t bc = sizeof(float) * static_cast<size_t>(batch) * dstate * seqlen;
    struct Span { char* p; size_t n; } sp[5] = {{reinterpret_cast<char*>(dA), sizeof(float) * dim * dstate},
                                                 {reinterpret_cast<char*>(dB), bc},
                                                 {reinterpret_cast<char*>(dC), bc},
                                                 {reinterpret_cast<char*>(dD), dD ? sizeof(float) * dim : 0},
                                                 {reinterpret_cast<char*>(ddelta_bias),
                                                  ddelta_bias ? sizeof(float) * dim : 0}};
    for (int i = 1; i < 5; ++i)                      // insertion sort by address
      for (int j = i; j > 0 && sp[j].p < sp[j - 1].p; --j) { const Span t = sp[j]; sp[j] = sp[j - 1]; sp[j - 1] = t; }
    for (int i = 0; i < 5;) {
      if (!sp[i].n) { ++i; continue; }
      char* lo = sp[i].p;
      char* hi = lo + sp[i].n;
      int j = i + 1;
      while (j < 5 && (sp[j].n == 0 || sp[j].p == hi)) { hi += sp[j].n; ++j; }
      if ((e = hipMemsetAsync(lo, 0, static_cast<size_t>(hi - lo), s)) != hipSuccess) return static_cast<int>(e);
      i = j;
    }
  }
  if (batch == 0 || seqlen == 0) return SIMAMBA_OK;
  ScanArgs a{};
  a.u = u; a.delta = delta; a.A = A; a.B = B; a.C = C; a.D = D; a.z = z; a.delta_bias = delta_bias;
  a.dout = dout; a.x_ckpt = const_cast<float*>(x_ckpt);
  a.du = du; a.ddelta = ddelta; a.dA = dA; a.dB = dB; a.dC = dC; a.dD = dD; a.dz = dz;
  a.ddelta_bias = ddelta_bias;
  a.batch = batch; a.dim = dim; a.seqlen = seqlen; a.dstate = dstate;
  a.nchunks = nchunks;
  a.softplus = delta_softplus;
  const size_t esz = io_dtype == SIMAMBA_F32 ? 4 : 2;
  a.z_bs = z_bstride ? z_bstride : static_cast<long long>(dim) * seqlen;
  a.dz_bs = dz_bstride ? dz_bstride : static_cast<long long>(dim) * seqlen;
  if (!bc_bstride && !bc_nstride && !bc_tstride) {
    bc_bstride = static_cast<long long>(dstate) * seqlen; bc_nstride = seqlen; bc_tstride = 1;
  }
  a.bc_bs = bc_bstride; a.bc_ns = bc_nstride; a.bc_ts = bc_tstride;
  a.vec = ((seqlen * esz) % 16 == 0) && aligned16b(u) && (dt || aligned16b(delta)) && aligned16b(dout) &&
          aligned16b(du) && aligned16b(ddelta) &&
          (!z || (aligned16b(z) && aligned16b(dz) && (a.z_bs * esz) % 16 == 0 && (a.dz_bs * esz) % 16 == 0));
  if (ckpt_step == SIMAMBA_SCAN_CKPT_SEQ) {
    const int bc_mode = scan_fwd_seq_bc_mode(B, C, io_dtype, a.bc_bs, a.bc_ns, a.bc_ts);
    if (!scan_bwd_seq_ok(batch, dim, seqlen, dstate, delta_softplus, a.vec, a.z_bs, a.dz_bs, z != nullptr, bc_mode,
                         a.bc_ns, a.bc_ts) ||
        (reinterpret_cast<uintptr_t>(A) & 15u) != 0 || (x_ckpt && (reinterpret_cast<uintptr_t>(x_ckpt) & 15u) != 0))
      return SIMAMBA_E_VARIANT;
    return scan_bwd_seq_dispatch(a, io_dtype, bc_mode, s, dt, wdt, dt_bs, dt_ts, dt_rank);
  }
  if (dt) return SIMAMBA_E_VARIANT;                          // only the sequential kernel forms delta itself
  // channels per workgroup (16 * passes): the more, the fewer dB/dC atomics reach HBM (measured at
  // (64,768,1024,16): 134 MB of flush traffic at passes = 3, 18 % on top of the 604 MB of gradient stores);
  // but keep >= 2 workgroups per CU, the number resident at this kernel's register footprint.
  static const int kCand[] = {12, 8, 6, 4, 3, 2, 1};
  int passes = 1;
  for (int cand : kCand) {
    if (static_cast<long long>(batch) * ((dim + 16 * cand - 1) / (16 * cand)) >= 512) { passes = cand; break; }
  }
  a.passes = passes;
  return io_dtype == SIMAMBA_F32 ? launch_bwd<float>(a, s) : launch_bwd<bf16_t>(a, s);
}

extern "C" int simamba_selective_scan_bwd(const void* u, const void* delta, const float* A, const void* B,
                                          const void* C, const float* D, const void* z,
                                          const float* delta_bias, const void* dout, const float* x_ckpt,
                                          void* du, void* ddelta, float* dA, float* dB, float* dC, float* dD,
                                          void* dz, float* ddelta_bias, int batch, int dim, int seqlen,
                                          int dstate, int io_dtype, int delta_softplus, long long z_bstride,
                                          long long dz_bstride, long long bc_bstride, long long bc_nstride,
                                          long long bc_tstride, int ckpt_step, void* stream) {
  return scan_bwd_impl(u, delta, A, B, C, D, z, delta_bias, dout, x_ckpt, du, ddelta, dA, dB, dC, dD, dz, ddelta_bias,
                       batch, dim, seqlen, dstate, io_dtype, delta_softplus, z_bstride, dz_bstride, bc_bstride,
                       bc_nstride, bc_tstride, ckpt_step, stream, nullptr, nullptr, 0, 0, 0);
}

// Backward of simamba_selective_scan_dt_fwd: same operands (+ dout, the forward's 16-step checkpoints); ddelta is the
// gradient with respect to the delta the kernels form (before bias and softplus), as in simamba_selective_scan_bwd.
extern "C" int simamba_selective_scan_dt_bwd(const void* u, const void* xdbl, const void* wdt, const float* A,
                                             const float* D, const void* z, const float* delta_bias, const void* dout,
                                             const float* x_ckpt, void* du, void* ddelta, float* dA, float* dB,
                                             float* dC, float* dD, void* dz, float* ddelta_bias, int batch, int dim,
                                             int seqlen, int dstate, int dt_rank, int io_dtype, long long z_bstride,
                                             long long dz_bstride, long long xdbl_bstride, long long xdbl_tstride,
                                             void* stream) {
  if (dstate != kMaxState) return SIMAMBA_E_DSTATE;
  if (io_dtype != SIMAMBA_F32 && io_dtype != SIMAMBA_BF16) return SIMAMBA_E_DTYPE;
  const int pack = io_dtype == SIMAMBA_F32 ? 4 : 8;
  if (dt_rank < pack || dt_rank > 24 || dt_rank % pack) return SIMAMBA_E_SHAPE;
  if (batch > 0 && seqlen > 0 && (!xdbl || !wdt || !z)) return SIMAMBA_E_NULLPTR;
  const size_t esz = io_dtype == SIMAMBA_F32 ? 4 : 2;
  const long long S = dt_rank + 2 * kMaxState;
  const long long xb = xdbl_bstride ? xdbl_bstride : S * seqlen, xt = xdbl_tstride ? xdbl_tstride : S;
  if (!aligned16b(xdbl) || !aligned16b(wdt) || (xb * esz) % 16 || (xt * esz) % 16 || (dt_rank * esz) % 16 ||
      static_cast<long long>(batch) * xb >= (1ll << 30))
    return SIMAMBA_E_VARIANT;
  const char* Bp = static_cast<const char*>(xdbl) + static_cast<size_t>(dt_rank) * esz;
  const char* Cp = Bp + kMaxState * esz;
  return scan_bwd_impl(u, nullptr, A, Bp, Cp, D, z, delta_bias, dout, x_ckpt, du, ddelta, dA, dB, dC, dD, dz,
                       ddelta_bias, batch, dim, seqlen, dstate, io_dtype, 1, z_bstride, dz_bstride, xb, 1, xt,
                       SIMAMBA_SCAN_CKPT_SEQ, stream, xdbl, wdt, xb, xt, dt_rank);
}
