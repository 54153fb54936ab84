// Selective-scan forward, "one lane per channel" form, for gfx950.
//
// Used when there are enough (batch, channel) rows to give every SIMD >= 2 waves with one row per lane
// (batch * dim >= kSeqMinRows): the recurrence h_t = a_t h_{t-1} + x_t B_t then needs only
// mul + exp2 + mul + fma + fma per (row, t, state) -- 5 VALU issues against ~9 for the row-scan form of
// scan_fwd.hip -- and the kernel is bound by the quarter-rate v_exp_f32 and the fp32 FMA pipe, the closest
// this op gets to the HBM roofline.
//
//   * a wave owns 64 consecutive channels of one sample and walks time in chunks of 16 steps;
//   * phase A (lane <-> time): 16-byte coalesced loads of delta / u / z rows, softplus and delta*u in the
//     loading lanes, transposed into two wave-private LDS tiles [64][16+4] (the row pitch of 20 dwords
//     makes the row-wise ds_read_b128 of phase B conflict-free);
//   * phase B (lane <-> channel): 16 sequential steps with the 16 states in VGPRs; B_t, C_t come from a
//     packed (batch, L, 32) fp32 image through SCALAR loads (wave-uniform address -> SGPR operands, no
//     VGPRs, no LDS bandwidth), y_t overwrites x_t in the tile;
//   * phase C (lane <-> time): y re-read in load order, (y + D u) * silu(z), 16-byte coalesced stores;
//   * the next chunk's global loads are issued before phase B, so HBM latency hides under the recurrence.
// No workgroup barrier anywhere: tiles are wave-private and LDS is in-order per wave.
#include "scan_common.h"

namespace simamba {

constexpr int kSeqTC = 16;                 // timesteps per chunk
constexpr int kSeqPitch = kSeqTC + 4;      // LDS row pitch (dwords)
constexpr int kSeqThreads = 256;

struct SeqArgs {
  const void* u;
  const void* delta;
  const void* z;
  void* out;
  const float* A;
  const float* D;
  const float* delta_bias;
  const float* bc;        // (batch, seqlen, 32): B_t[0..16) then C_t[0..16), fp32
  float* x_ckpt;
  float* last_state;
  int batch, dim, seqlen, nchunks128;
  int softplus, vec;
  long long z_bs;
};

// B, C (any strides, io dtype) -> packed (batch, L, 32) fp32
template <typename T>
__global__ void bc_pack_kernel(const T* __restrict__ Bg, const T* __restrict__ Cg, float* __restrict__ dst, int L,
                               int N, long long bs, long long ns, long long ts) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int b = blockIdx.y;
  if (t >= L) return;
  float v[32];
#pragma unroll
  for (int n = 0; n < 16; ++n) {
    const long long o = static_cast<long long>(b) * bs + n * ns + static_cast<long long>(t) * ts;
    v[n] = (n < N) ? to_f32<T>(Bg[o]) : 0.f;
    v[16 + n] = (n < N) ? to_f32<T>(Cg[o]) : 0.f;
  }
  float4* o4 = reinterpret_cast<float4*>(dst + (static_cast<size_t>(b) * L + t) * 32);
#pragma unroll
  for (int q = 0; q < 8; ++q) o4[q] = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
}

// ---- scalar (SMEM) loads of the packed B_t | C_t row --------------------------------------------------
// hipcc will not scalarise these loads by itself (the kernel also stores through other pointers, so the
// loads are not provably unclobbered), and as vector loads they cost 32 VGPRs per step.  The asm pair
// below issues two s_load_dwordx16 and later retires them; the "memory" clobbers keep the compiler's own
// LDS traffic out of the window in which the SMEM loads are in flight (lgkmcnt is shared and SMEM returns
// out of order, so a counted wait on an LDS read would be unsafe there), and the "+s" operands of the wait
// make every use of the loaded values depend on it.
typedef float f32x16 __attribute__((ext_vector_type(16)));

// `pin_before` / `pin_after` are VGPR values threaded through the asm only to pin the schedule: the step's
// arithmetic consumes pin_before (so it cannot be hoisted above the issue) and produces pin_after (so the
// retire cannot be hoisted above it) -- register-only VALU code is otherwise free to cross an asm volatile.
__device__ __forceinline__ void bc_issue(const float* rowp, f32x16& Bt, f32x16& Ct, float& pin_before) {
  asm volatile("s_load_dwordx16 %0, %3, 0x0\n\ts_load_dwordx16 %1, %3, 0x40"
               : "=&s"(Bt), "=&s"(Ct), "+v"(pin_before)
               : "s"(rowp)
               : "memory");
}
__device__ __forceinline__ void bc_wait(f32x16& Bt, f32x16& Ct, float& pin_after) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(Bt), "+s"(Ct), "+v"(pin_after)::"memory");
}
__device__ __forceinline__ void bc_wait(f32x16& Bt, f32x16& Ct) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(Bt), "+s"(Ct)::"memory");
}

// One aligned 4-element pack per lane (16 B fp32 / 8 B bf16).  The dispatcher only takes this kernel when
// rows are pack-aligned (L % 4 == 0 for fp32, L % 8 == 0 for bf16), so a pack is either entirely inside the
// sequence or entirely outside: out-of-range packs read element 0 of the tensor and are zeroed -- no
// per-element guards, no divergent branches.
template <typename T>
__device__ __forceinline__ void load4(const T* __restrict__ base, unsigned off, bool ok, float (&v)[4]) {
  const Pack<T, 4> pk = *reinterpret_cast<const Pack<T, 4>*>(base + (ok ? off : 0u));
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = ok ? to_f32<T>(pk.v[i]) : 0.f;
}
template <typename T>
__device__ __forceinline__ void store4(T* __restrict__ base, unsigned off, const float (&v)[4]) {
  Pack<T, 4> pk;
#pragma unroll
  for (int i = 0; i < 4; ++i) pk.v[i] = from_f32<T>(v[i]);
  *reinterpret_cast<Pack<T, 4>*>(base + off) = pk;
}

template <typename T, bool kHasZ>
__global__ __launch_bounds__(kSeqThreads, 3) void scan_fwd_seq_kernel(SeqArgs p) {
  __shared__ __attribute__((aligned(16))) float sTile[kSeqThreads / 64][2][64 * kSeqPitch];
  const int b = blockIdx.y;
  const int wave = threadIdx.x >> 6;
  const int lane = threadIdx.x & 63;
  const int L = p.seqlen, D = p.dim;
  const int ch_base = blockIdx.x * kSeqThreads + wave * 64;   // first channel of this wave
  if (ch_base >= D) return;                                    // whole wave idle (no barriers in this kernel)
  float* tD = &sTile[wave][0][0];
  float* tX = &sTile[wave][1][0];

  // ---- phase B identity: one channel per lane ---------------------------------------------------------
  const int d_own = min(ch_base + lane, D - 1);
  const bool own_valid = ch_base + lane < D;
  float A2[kMaxState], h[kMaxState];
  load_A_row(p.A + static_cast<size_t>(d_own) * kMaxState, kMaxState, A2);
#pragma unroll
  for (int n = 0; n < kMaxState; ++n) h[n] = 0.f;

  // ---- phase A / C identity: lane covers rows (lane/4 + 16 j), quarter q = lane % 4 of the chunk ------
  const int q = lane & 3;
  int rowA[4], dA[4];
  float biasA[4], DA[4];
  bool validA[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    rowA[j] = (lane >> 2) + 16 * j;
    validA[j] = ch_base + rowA[j] < D;
    dA[j] = min(ch_base + rowA[j], D - 1);
    biasA[j] = p.delta_bias ? p.delta_bias[dA[j]] : 0.f;
    DA[j] = p.D ? p.D[dA[j]] : 0.f;
  }
  const T* __restrict__ ug = static_cast<const T*>(p.u);
  const T* __restrict__ dg = static_cast<const T*>(p.delta);
  const T* __restrict__ zg = static_cast<const T*>(p.z);
  T* __restrict__ og = static_cast<T*>(p.out);
  const float* __restrict__ bc = p.bc + static_cast<size_t>(b) * L * 32;

  // 32-bit element offsets off one base pointer per tensor (the dispatcher guarantees < 2^31 elements):
  // keeps the address state at one VGPR per row instead of a 64-bit pointer per row and tensor.
  unsigned rowoff[4], zoff[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    rowoff[j] = (static_cast<unsigned>(b) * D + dA[j]) * L + 4 * q;
    zoff[j] = static_cast<unsigned>(b * p.z_bs) + static_cast<unsigned>(dA[j]) * L + 4 * q;
  }
  float dv[4][4], uv[4][4], zv[4][4];     // delta (reused for the NEXT chunk once phase A is done), u, z
  float un[4][4];                         // next chunk's u, in flight during phase B
  auto issue_loads = [&](int t0, float (&dd)[4][4], float (&uu)[4][4]) {
    const bool ok = t0 + 4 * q < L;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      load4<T>(dg, rowoff[j] + t0, ok, dd[j]);
      load4<T>(ug, rowoff[j] + t0, ok, uu[j]);
    }
  };
  auto issue_z = [&](int t0) {
    const bool ok = t0 + 4 * q < L;
#pragma unroll
    for (int j = 0; j < 4; ++j) load4<T>(zg, zoff[j] + t0, ok, zv[j]);
  };

  const int nchunks = (L + kSeqTC - 1) / kSeqTC;
  f32x16 B0, C0, B1, C1;
  float pin0 = 0.f;
  bc_issue(bc, B0, C0, pin0);
  bc_wait(B0, C0);
  issue_loads(0, dv, uv);
  for (int c = 0; c < nchunks; ++c) {
    const int t0 = c * kSeqTC;
    // ---- phase A: softplus, delta * u, transpose into the tiles ----------------------------------------
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float4 d4, x4;
      float dl[4], xx[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float x = dv[j][i] + biasA[j];
        x = p.softplus ? softplus_f(x) : x;
        x = (t0 + 4 * q < L) ? x : 0.f;          // padded packs: identity map, no input
        dl[i] = x;
        xx[i] = x * uv[j][i];
      }
      d4 = make_float4(dl[0], dl[1], dl[2], dl[3]);
      x4 = make_float4(xx[0], xx[1], xx[2], xx[3]);
      *reinterpret_cast<float4*>(tD + rowA[j] * kSeqPitch + 4 * q) = d4;
      *reinterpret_cast<float4*>(tX + rowA[j] * kSeqPitch + 4 * q) = x4;
    }
    if (kHasZ) issue_z(t0);
    if (c + 1 < nchunks) issue_loads(t0 + kSeqTC, dv, un);      // dv is dead after phase A
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();

    // ---- phase B: the recurrence, one channel per lane ---------------------------------------------------
    // B_t | C_t of step t sit in SGPRs (set 0 / set 1 alternate); the row of step t+1 is requested before
    // step t is computed and retired after it.
#pragma unroll
    for (int g = 0; g < kSeqTC / 4; ++g) {
      const float4 d4 = *reinterpret_cast<const float4*>(tD + lane * kSeqPitch + 4 * g);
      const float4 x4 = *reinterpret_cast<const float4*>(tX + lane * kSeqPitch + 4 * g);
      float dl[4] = {d4.x, d4.y, d4.z, d4.w};
      const float xx[4] = {x4.x, x4.y, x4.z, x4.w};
      float yy[4];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         // the two tile reads above are back
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int tnext = min(t0 + 4 * g + i + 1, L - 1);        // wave-uniform
        f32x16& Bc = (i & 1) ? B1 : B0;
        f32x16& Cc = (i & 1) ? C1 : C0;
        f32x16& Bn = (i & 1) ? B0 : B1;
        f32x16& Cn = (i & 1) ? C0 : C1;
        bc_issue(bc + static_cast<size_t>(tnext) * 32, Bn, Cn, dl[i]);
        float y = 0.f;
#pragma unroll
        for (int n = 0; n < kMaxState; ++n) {
          const float a = fast_exp2(dl[i] * A2[n]);
          h[n] = fmaf(a, h[n], xx[i] * Bc[n]);
          y = fmaf(h[n], Cc[n], y);
        }
        bc_wait(Bn, Cn, y);
        yy[i] = y;
      }
      *reinterpret_cast<float4*>(tX + lane * kSeqPitch + 4 * g) = make_float4(yy[0], yy[1], yy[2], yy[3]);
    }
    // state checkpoints at the 128-step boundaries the backward uses, and the final state
    const int tend = t0 + kSeqTC;
    if (own_valid) {
      if (p.x_ckpt && ((tend % SIMAMBA_SCAN_CHUNK) == 0 || tend >= L)) {
        const int c128 = (min(tend, L) - 1) / SIMAMBA_SCAN_CHUNK;
        float4* dst = reinterpret_cast<float4*>(
            p.x_ckpt + ((static_cast<size_t>(b) * D + d_own) * p.nchunks128 + c128) * kMaxState);
#pragma unroll
        for (int k = 0; k < 4; ++k) dst[k] = make_float4(h[4 * k], h[4 * k + 1], h[4 * k + 2], h[4 * k + 3]);
      }
      if (p.last_state && tend >= L) {
        float4* dst = reinterpret_cast<float4*>(p.last_state + (static_cast<size_t>(b) * D + d_own) * kMaxState);
#pragma unroll
        for (int k = 0; k < 4; ++k) dst[k] = make_float4(h[4 * k], h[4 * k + 1], h[4 * k + 2], h[4 * k + 3]);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();

    // ---- phase C: gate and store in load order -------------------------------------------------------------
    {
      const bool ok = t0 + 4 * q < L;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float4 y4 = *reinterpret_cast<const float4*>(tX + rowA[j] * kSeqPitch + 4 * q);
        const float yy[4] = {y4.x, y4.y, y4.z, y4.w};
        float o[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float v = fmaf(DA[j], uv[j][i], yy[i]);
          if (kHasZ) v = v * zv[j][i] * sigmoid_f(zv[j][i]);
          o[i] = v;
        }
        if (ok && validA[j]) store4<T>(og, rowoff[j] + t0, o);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) uv[j][i] = un[j][i];
  }
}

template <typename T>
static int launch_seq(const SeqArgs& a, const void* B, const void* C, float* ws, int dstate, long long bs, long long ns,
                      long long ts, hipStream_t s) {
  dim3 pgrid((a.seqlen + 127) / 128, a.batch);
  hipLaunchKernelGGL(bc_pack_kernel<T>, pgrid, dim3(128), 0, s, static_cast<const T*>(B), static_cast<const T*>(C), ws,
                     a.seqlen, dstate, bs, ns, ts);
  dim3 grid((a.dim + kSeqThreads - 1) / kSeqThreads, a.batch);
  if (a.z)
    hipLaunchKernelGGL((scan_fwd_seq_kernel<T, true>), grid, dim3(kSeqThreads), 0, s, a);
  else
    hipLaunchKernelGGL((scan_fwd_seq_kernel<T, false>), grid, dim3(kSeqThreads), 0, s, a);
  return static_cast<int>(hipGetLastError());
}

// Entry used by simamba_selective_scan_fwd (scan_fwd.hip) when the shape qualifies.
int scan_fwd_seq_dispatch(const void* u, const void* delta, const float* A, const void* B, const void* C, const float* D,
                          const void* z, const float* delta_bias, void* out, float* x_ckpt, float* last_state,
                          int batch, int dim, int seqlen, int dstate, int io_dtype, int delta_softplus,
                          long long z_bs, long long bc_bs, long long bc_ns, long long bc_ts, int vec, int nchunks128,
                          void* workspace, hipStream_t s) {
  SeqArgs a{};
  a.u = u; a.delta = delta; a.z = z; a.out = out; a.A = A; a.D = D; a.delta_bias = delta_bias;
  a.bc = static_cast<const float*>(workspace);
  a.x_ckpt = x_ckpt; a.last_state = last_state;
  a.batch = batch; a.dim = dim; a.seqlen = seqlen; a.nchunks128 = nchunks128;
  a.softplus = delta_softplus; a.vec = vec; a.z_bs = z_bs;
  float* ws = static_cast<float*>(workspace);
  return io_dtype == SIMAMBA_F32 ? launch_seq<float>(a, B, C, ws, dstate, bc_bs, bc_ns, bc_ts, s)
                                 : launch_seq<bf16_t>(a, B, C, ws, dstate, bc_bs, bc_ns, bc_ts, s);
}

}  // namespace simamba
