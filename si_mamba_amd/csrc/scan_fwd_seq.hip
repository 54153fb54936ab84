// Selective-scan forward, "lane per channel" form, for gfx950.
//
// Used when there are enough (batch, channel) rows to fill the chip with kLPC lanes per row
// (kLPC = 1, 2 or 4 adjacent lanes share a channel and split its 16 states): the recurrence
// h_t = a_t h_{t-1} + x_t B_t then needs only mul + exp2 + mul + fma + fma per (row, t, state) -- 5 VALU
// issues against ~9 for the row-scan form of scan_fwd.hip.  tools/valu_probe.hip prices that group at
// 8.6 - 10 ns per wave and SIMD (v_exp_f32 is quarter rate and does not overlap the FMA pipe).
// Measured at 256 x 768 x 128 x 16 fp32: 150 us (row-scan kernel: 165 us).  What bounds it is no longer the
// recurrence: built with -DSIMAMBA_SEQ_SKIP_B (phases A and C only) the kernel still takes 130 us, because a
// chunk touches 64 B of every 128-byte line and the other half is requested one chunk (~16 us, ~9 MB of
// traffic per XCD) later, after the line has left the 4 MB L2 -- PMC FETCH_SIZE shows 2x the algorithmic
// read bytes.  Full-line (32-step) chunks would need 2x the LDS or registers per wave and cost the occupancy
// the VALU side needs; DESIGN.md section 4.1 has the numbers.
//
//   * waves are independent (no workgroup barrier): a wave owns R = 64 / kLPC consecutive channels of one
//     sample and walks time in chunks of 16 steps through wave-private LDS tiles;
//   * phase A (lane <-> 4 time steps of a row): 16-byte coalesced loads of delta / u, softplus, into the
//     tiles tD = delta, tU = u, [R][16] floats with the 16-byte column groups XOR-swizzled by (row >> 2) & 3
//     so both the row-wise reads of phase B and the pack-wise accesses of A / C are conflict-free without
//     padding; the chunk's B_t | C_t (16 steps x 32 floats) is staged next to them straight from the
//     strided B / C operands (no packing pass, no workspace);
//   * phase B (lane <-> channel, NS = 16 / kLPC states in VGPRs): per step one ds_read_b128 each of B_t and
//     C_t at address 16 * (lane & 3): VGPR j then holds entry 4p + j at quad position p, identically in all
//     16 quads.  With one lane per channel the operand "B_t[4a + j] for every lane" is that VGPR read through
//     DPP quad_perm:[a,a,a,a] inside the multiply -- the broadcast costs no instruction, no SGPR, no scalar
//     cache traffic (a first version fed B_t / C_t through s_load: tools/smem_probe.hip measures 30 - 50 ns
//     per scalar-cache miss and CU, which capped that kernel at 163 us).  With two lanes per channel the
//     perm is [a,a+2,a,a+2] (the odd lane gets entry 8 + 4a + j), with four it is the identity (plain
//     operand).  The kLPC partial y_t are combined with 1 - 2 DPP quad adds per step;
//   * y_t overwrites u_t in the tile; phase C (same lane <-> pack map as A): y * silu(z), 16-byte stores;
//   * the next chunk's global loads are issued right after phase A, so HBM latency hides under phase B.
#include <cstdlib>
#include "scan_common.h"

namespace simamba {

constexpr int kSeqTC = 16;                 // timesteps per chunk
constexpr int kSeqThreads = 256;           // 4 independent waves
constexpr int kBcPitch = 36;               // floats per staged B_t | C_t row (32 + 4: staging writes 2-way, not 16-way)

struct SeqArgs {
  const void* u;
  const void* delta;
  const void* z;
  void* out;
  const float* A;
  const float* D;
  const float* delta_bias;
  const void* B;
  const void* C;
  float* x_ckpt;
  float* last_state;
  int batch, dim, seqlen, nchunks128;
  int softplus;
  long long z_bs;
  long long bc_bs, bc_ns, bc_ts;
};

// ---- multiply / multiply-accumulate with a quad-broadcast first operand ------------------------------------
// The DPP-selected source is always a VGPR written by ds_read (never by a VALU op), so the
// "VALU write -> DPP read: 2 wait states" hazard does not apply and no s_nop padding is needed.  Plain
// (non-volatile) asm: pure functions of their operands, free for the scheduler to interleave.
template <int P> __device__ __forceinline__ float mul_q(float s, float x);
template <int P> __device__ __forceinline__ void fmac_q(float& acc, float s, float x);
#define SIMAMBA_QUAD_OPS(P, PERM)                                                                               \
  template <> __device__ __forceinline__ float mul_q<P>(float s, float x) {                                     \
    float r;                                                                                                    \
    asm("v_mul_f32_dpp %0, %1, %2 quad_perm:" PERM " row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(s), "v"(x));  \
    return r;                                                                                                   \
  }                                                                                                             \
  template <> __device__ __forceinline__ void fmac_q<P>(float& acc, float s, float x) {                        \
    asm("v_fmac_f32_dpp %0, %1, %2 quad_perm:" PERM " row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(s), "v"(x)); \
  }
SIMAMBA_QUAD_OPS(0, "[0,0,0,0]")
SIMAMBA_QUAD_OPS(1, "[1,1,1,1]")
SIMAMBA_QUAD_OPS(2, "[2,2,2,2]")
SIMAMBA_QUAD_OPS(3, "[3,3,3,3]")
SIMAMBA_QUAD_OPS(4, "[0,2,0,2]")
SIMAMBA_QUAD_OPS(5, "[1,3,1,3]")
#undef SIMAMBA_QUAD_OPS

// One aligned 4-element pack per lane (16 B fp32 / 8 B bf16).  The dispatcher only takes this kernel when
// rows are pack-aligned (L % 4 == 0 for fp32, L % 8 == 0 for bf16), so a pack is either entirely inside the
// sequence or entirely outside: out-of-range packs read element 0 of the tensor and are zeroed -- no
// per-element guards, no divergent branches.
// Addressing is "uniform base pointer + 32-bit BYTE offset" throughout (the dispatcher guarantees every tensor
// spans < 4 GiB): global_load/store then take the base in SGPRs and one VGPR of offset, instead of a 64-bit
// VGPR address per access that the compiler hoists out of the chunk loop and spills.
template <typename T>
__device__ __forceinline__ void load4(const T* __restrict__ base, unsigned boff, bool ok, float (&v)[4]) {
  const Pack<T, 4> pk =
      *reinterpret_cast<const Pack<T, 4>*>(reinterpret_cast<const char*>(base) + (ok ? boff : 0u));
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = ok ? to_f32<T>(pk.v[i]) : 0.f;
}
template <typename T>
__device__ __forceinline__ void store4(T* __restrict__ base, unsigned boff, const float (&v)[4]) {
  Pack<T, 4> pk;
#pragma unroll
  for (int i = 0; i < 4; ++i) pk.v[i] = from_f32<T>(v[i]);
  *reinterpret_cast<Pack<T, 4>*>(reinterpret_cast<char*>(base) + boff) = pk;
}

// float offset of 16-byte column group g of a tile row
__device__ __forceinline__ int tile_off(int row, int g) { return row * kSeqTC + 4 * (g ^ ((row >> 2) & 3)); }

template <int kLPC> struct SeqCfg {
  static constexpr int NS = kMaxState / kLPC;    // states per lane
  static constexpr int R = 64 / kLPC;            // channels per wave
  static constexpr int kPacks = 4 / kLPC;        // 4-step packs per lane, tensor and chunk
  static constexpr int kTileFloats = 2 * R * kSeqTC + kSeqTC * kBcPitch;
  static constexpr int kWaves = kLPC == 1 ? 3 : 4;   // waves per SIMD the register budget targets (6 spills)
};

template <typename T, bool kHasZ, int kLPC>
__global__ __launch_bounds__(kSeqThreads, SeqCfg<kLPC>::kWaves) void scan_fwd_seq_kernel(SeqArgs p) {
  typedef SeqCfg<kLPC> Cfg;
  constexpr int NS = Cfg::NS, R = Cfg::R, kPacks = Cfg::kPacks;
  __shared__ __attribute__((aligned(16))) float sMem[kSeqThreads / 64][Cfg::kTileFloats];
  int tile_id, b;
  xcd_tile(tile_id, b);
  const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) >> 6);   // SGPR
  const int lane = threadIdx.x & 63;
  const int L = p.seqlen, D = p.dim;
  const int ch_base = (tile_id * (kSeqThreads / 64) + wave) * R;   // first channel of this wave
  if (ch_base >= D) return;                                            // whole wave idle (no barriers here)
  float* tD = &sMem[wave][0];
  float* tU = tD + R * kSeqTC;
  float* tBC = tU + R * kSeqTC;            // [16 steps][B_t(16) | C_t(16) | pad(4)]

  // ---- phase B identity: channel lane / kLPC, states NS * (lane % kLPC) .. + NS --------------------------
  const int rowB = lane / kLPC;
  const int n0 = NS * (lane % kLPC);
  const int d_own = min(ch_base + rowB, D - 1);
  const bool own_valid = ch_base + rowB < D;
  float A2[NS], h[NS];
#pragma unroll
  for (int q4 = 0; q4 < NS / 4; ++q4) {
    const float4 a4 = *reinterpret_cast<const float4*>(p.A + static_cast<size_t>(d_own) * kMaxState + n0 + 4 * q4);
    A2[4 * q4] = a4.x * kLog2e; A2[4 * q4 + 1] = a4.y * kLog2e; A2[4 * q4 + 2] = a4.z * kLog2e;
    A2[4 * q4 + 3] = a4.w * kLog2e;
  }
#pragma unroll
  for (int n = 0; n < NS; ++n) h[n] = 0.f;
  // the skip term D * u_t enters the partial sum of the first lane of a channel only
  const float Dl = (p.D && (lane % kLPC) == 0) ? p.D[d_own] : 0.f;

  // ---- phase A / C identity: pack pk = lane + 64 j covers row pk / 4, steps 4 (pk % 4) .. + 4 -------------
  const int q = lane & 3;
  float biasA[kPacks];
  unsigned rowoff[kPacks];
  // byte offsets off one base pointer per tensor; z sits at a wave-uniform distance from u (its batch stride
  // may differ)
  constexpr unsigned kEsz = sizeof(T);
  const unsigned zdelta = (static_cast<unsigned>(b * p.z_bs) - static_cast<unsigned>(b) * D * L) * kEsz;
#pragma unroll
  for (int j = 0; j < kPacks; ++j) {
    const int dA = min(ch_base + (lane >> 2) + 16 * j, D - 1);
    biasA[j] = p.delta_bias ? p.delta_bias[dA] : 0.f;
    rowoff[j] = ((static_cast<unsigned>(b) * D + dA) * L + 4 * q) * kEsz;
  }
  const T* __restrict__ ug = static_cast<const T*>(p.u);
  const T* __restrict__ dg = static_cast<const T*>(p.delta);
  const T* __restrict__ zg = static_cast<const T*>(p.z);
  T* __restrict__ og = static_cast<T*>(p.out);

  // ---- B_t | C_t staging identity: 512 scalars per chunk, 8 per lane, consecutive lanes on the contiguous
  // axis of the operand (time for (B,N,L) tensors, state for token-major views of the x_proj output) ---------
  const T* __restrict__ Bg = static_cast<const T*>(p.B) + static_cast<long long>(b) * p.bc_bs;
  const T* __restrict__ Cg = static_cast<const T*>(p.C) + static_cast<long long>(b) * p.bc_bs;
  const bool token_major = p.bc_ns == 1;
  const int bc_ns = static_cast<int>(p.bc_ns), bc_ts = static_cast<int>(p.bc_ts);
  // `lane_v` is an opaque copy of the lane id refreshed once per chunk: index arithmetic derived from it is
  // recomputed per chunk (a few VALU ops) instead of being hoisted into ~40 loop-invariant VGPRs and spilled
  int lane_v = lane;
  auto bc_index = [&](int i, int& n, int& t) {
    const int e = lane_v + 64 * i;                     // 0 .. 255
    n = token_major ? (e & 15) : (e >> 4);
    t = token_major ? (e >> 4) : (e & 15);
  };
  float bcv[2][4];
  auto issue_bc = [&](int t0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int n, tl;
      bc_index(i, n, tl);
      const int t = t0 + tl;
      const bool ok = t < L;
      const unsigned o = ok ? static_cast<unsigned>(n * bc_ns + t * bc_ts) * kEsz : 0u;   // bytes, < 2^32
      const float vb = to_f32<T>(*reinterpret_cast<const T*>(reinterpret_cast<const char*>(Bg) + o));
      const float vc = to_f32<T>(*reinterpret_cast<const T*>(reinterpret_cast<const char*>(Cg) + o));
      bcv[0][i] = ok ? vb : 0.f;
      bcv[1][i] = ok ? vc : 0.f;
    }
  };

  float dv[kPacks][4], uv[kPacks][4], zv[kPacks][4];
  auto issue_loads = [&](int t0) {
    const bool ok = t0 + 4 * q < L;
#pragma unroll
    for (int j = 0; j < kPacks; ++j) {
      load4<T>(dg, rowoff[j] + t0 * kEsz, ok, dv[j]);
      load4<T>(ug, rowoff[j] + t0 * kEsz, ok, uv[j]);
    }
  };

  const int nchunks = (L + kSeqTC - 1) / kSeqTC;
  issue_bc(0);
  issue_loads(0);
  for (int c = 0; c < nchunks; ++c) {
    const int t0 = c * kSeqTC;
    asm volatile("" : "+v"(lane_v));
    const int qv = lane_v & 3, prow = lane_v >> 2;
    const bool in_seq = t0 + 4 * qv < L;
    // ---- phase A: softplus(delta), u and the chunk's B | C into the tiles ------------------------------------
#pragma unroll
    for (int j = 0; j < kPacks; ++j) {
      float dl[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float x = dv[j][i] + biasA[j];
        x = p.softplus ? softplus_f(x) : x;
        dl[i] = in_seq ? x : 0.f;                  // padded packs: identity map (and u = 0: no input)
      }
      const int o = tile_off(prow + 16 * j, qv);
      *reinterpret_cast<float4*>(tD + o) = make_float4(dl[0], dl[1], dl[2], dl[3]);
      *reinterpret_cast<float4*>(tU + o) = make_float4(uv[j][0], uv[j][1], uv[j][2], uv[j][3]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int n, tl;
      bc_index(i, n, tl);
      tBC[tl * kBcPitch + n] = bcv[0][i];
      tBC[tl * kBcPitch + 16 + n] = bcv[1][i];
    }
    // dv / uv are dead after phase A: with registers to spare (kLPC = 1) the next chunk's loads fly during
    // the whole recurrence, otherwise they are requested after it (6 resident waves cover the latency)
    if (kLPC == 1 && c + 1 < nchunks) issue_loads(t0 + kSeqTC);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();

    // ---- phase B: the recurrence ------------------------------------------------------------------------------
    // LDS reads run one step (B_t | C_t) / one group (delta, u) ahead of their use; the sched_barriers keep
    // hipcc from sinking them back next to the consumer (it did, exposing ~200 cycles of LDS latency per step)
    float4 d4n, u4n;
    if (kLPC == 1) {
      d4n = *reinterpret_cast<const float4*>(tD + tile_off(rowB, 0));
      u4n = *reinterpret_cast<const float4*>(tU + tile_off(rowB, 0));
    }
    float4 b4n = *reinterpret_cast<const float4*>(tBC + 4 * q);
    float4 c4n = *reinterpret_cast<const float4*>(tBC + 16 + 4 * q);
#ifdef SIMAMBA_SEQ_SKIP_B        // timing experiment only: phases A and C without the recurrence
#pragma unroll 1
    for (int g = 0; g < 0; ++g) {
#else
#pragma unroll 1
    for (int g = 0; g < kSeqTC / 4; ++g) {
#endif
      const int o = tile_off(rowB, g);
      if (kLPC != 1) {
        d4n = *reinterpret_cast<const float4*>(tD + o);
        u4n = *reinterpret_cast<const float4*>(tU + o);
      }
      const float dl[4] = {d4n.x, d4n.y, d4n.z, d4n.w};
      const float uu[4] = {u4n.x, u4n.y, u4n.z, u4n.w};
      if (kLPC == 1) {
        const int on = tile_off(rowB, (g + 1) & 3);      // wraps on the last group: harmless re-read
        d4n = *reinterpret_cast<const float4*>(tD + on);
        u4n = *reinterpret_cast<const float4*>(tU + on);
      }
      float yy[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float vB[4] = {b4n.x, b4n.y, b4n.z, b4n.w};
        const float vC[4] = {c4n.x, c4n.y, c4n.z, c4n.w};
        {
          const int tn = (4 * g + i + 1) & (kSeqTC - 1);
          b4n = *reinterpret_cast<const float4*>(tBC + tn * kBcPitch + 4 * q);
          c4n = *reinterpret_cast<const float4*>(tBC + tn * kBcPitch + 16 + 4 * q);
        }
        __builtin_amdgcn_sched_barrier(0);
        const float xx = dl[i] * uu[i];
        float ys[2] = {Dl * uu[i], 0.f};                 // two partial sums: halves the dependent fmac chain
#pragma unroll
        for (int a = 0; a < NS / 4; ++a) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int n = 4 * a + j;
            float& y = ys[j & 1];
            const float e = fast_exp2(dl[i] * A2[n]);
            if constexpr (kLPC == 4) {
              h[n] = fmaf(e, h[n], xx * vB[j]);
              y = fmaf(h[n], vC[j], y);
            } else {
              constexpr int kSel = kLPC == 1 ? 0 : 4;
              float xb;
              if (a == 0) xb = mul_q<kSel + 0>(vB[j], xx);
              else if (a == 1) xb = mul_q<kSel + 1>(vB[j], xx);
              else if (a == 2) xb = mul_q<(kLPC == 1 ? 2 : 0)>(vB[j], xx);
              else xb = mul_q<(kLPC == 1 ? 3 : 0)>(vB[j], xx);
              h[n] = fmaf(e, h[n], xb);
              if (a == 0) fmac_q<kSel + 0>(y, vC[j], h[n]);
              else if (a == 1) fmac_q<kSel + 1>(y, vC[j], h[n]);
              else if (a == 2) fmac_q<(kLPC == 1 ? 2 : 0)>(y, vC[j], h[n]);
              else fmac_q<(kLPC == 1 ? 3 : 0)>(y, vC[j], h[n]);
            }
          }
        }
        float y = ys[0] + ys[1];
        if (kLPC >= 2) y += dpp<DPP_QUAD_XOR1>(0.f, y);
        if (kLPC == 4) y += dpp<DPP_QUAD_XOR2>(0.f, y);
        yy[i] = y;
        __builtin_amdgcn_sched_barrier(0);
      }
      // all kLPC lanes of a channel hold the same sums and store them to the same place (no EXEC games
      // next to DPP code)
      *reinterpret_cast<float4*>(tU + o) = make_float4(yy[0], yy[1], yy[2], yy[3]);
    }
    // z of this chunk and B | C of the next: requested only now so that their registers are free during the
    // recurrence; the latency hides under phase C and the other resident waves
    if (kHasZ) {
#pragma unroll
      for (int j = 0; j < kPacks; ++j) load4<T>(zg, rowoff[j] + zdelta + t0 * kEsz, in_seq, zv[j]);
    }
    if (c + 1 < nchunks) {
      issue_bc(t0 + kSeqTC);
      if (kLPC != 1) issue_loads(t0 + kSeqTC);
    }
    // state checkpoints at the 128-step boundaries the backward uses, and the final state
    const int tend = t0 + kSeqTC;
    if (own_valid) {
      if (p.x_ckpt && ((tend % SIMAMBA_SCAN_CHUNK) == 0 || tend >= L)) {
        const int c128 = (min(tend, L) - 1) / SIMAMBA_SCAN_CHUNK;
        float4* dst = reinterpret_cast<float4*>(
            p.x_ckpt + ((static_cast<size_t>(b) * D + d_own) * p.nchunks128 + c128) * kMaxState + n0);
#pragma unroll
        for (int k = 0; k < NS / 4; ++k) dst[k] = make_float4(h[4 * k], h[4 * k + 1], h[4 * k + 2], h[4 * k + 3]);
      }
      if (p.last_state && tend >= L) {
        float4* dst = reinterpret_cast<float4*>(p.last_state + (static_cast<size_t>(b) * D + d_own) * kMaxState + n0);
#pragma unroll
        for (int k = 0; k < NS / 4; ++k) dst[k] = make_float4(h[4 * k], h[4 * k + 1], h[4 * k + 2], h[4 * k + 3]);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();

    // ---- phase C: gate and store in load order ----------------------------------------------------------------
#pragma unroll
    for (int j = 0; j < kPacks; ++j) {
      const float4 y4 = *reinterpret_cast<const float4*>(tU + tile_off(prow + 16 * j, qv));
      const float yv[4] = {y4.x, y4.y, y4.z, y4.w};
      float o[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float v = yv[i];
        if (kHasZ) v = v * zv[j][i] * sigmoid_f(zv[j][i]);
        o[i] = v;
      }
      if (in_seq && ch_base + prow + 16 * j < D) store4<T>(og, rowoff[j] + t0 * kEsz, o);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}

template <typename T, int kLPC>
static void launch_seq_lpc(const SeqArgs& a, hipStream_t s) {
  const int ch_per_wg = (kSeqThreads / 64) * SeqCfg<kLPC>::R;
  dim3 grid((a.dim + ch_per_wg - 1) / ch_per_wg, a.batch);
  if (a.z)
    hipLaunchKernelGGL((scan_fwd_seq_kernel<T, true, kLPC>), grid, dim3(kSeqThreads), 0, s, a);
  else
    hipLaunchKernelGGL((scan_fwd_seq_kernel<T, false, kLPC>), grid, dim3(kSeqThreads), 0, s, a);
}

// lanes per channel: one while that still gives every one of the 1024 SIMDs 3 waves, two below
// (measured, fp32, MI355X: 256x768x128 150 us with 1 lane / 154 us with 2 / 176 us with 4;
//  128x768x1024 607 us with 2 lanes / 687 us with 4 -- the row-scan kernel needs 165 us / 666 us)
int seq_lanes_per_channel(long long rows) {
  if (const char* e = getenv("SIMAMBA_SEQ_LPC")) {       // tuning knob (tools/bench_scan.py)
    const int v = atoi(e);
    if (v == 1 || v == 2 || v == 4) return v;
  }
  return rows >= 3 * 1024 * 64 ? 1 : 2;
}

template <typename T>
static int launch_seq(const SeqArgs& a, hipStream_t s) {
  switch (seq_lanes_per_channel(static_cast<long long>(a.batch) * a.dim)) {
    case 1: launch_seq_lpc<T, 1>(a, s); break;
    case 2: launch_seq_lpc<T, 2>(a, s); break;
    default: launch_seq_lpc<T, 4>(a, s); break;
  }
  return static_cast<int>(hipGetLastError());
}

// Entry used by simamba_selective_scan_fwd (scan_fwd.hip) when the shape qualifies.
int scan_fwd_seq_dispatch(const void* u, const void* delta, const float* A, const void* B, const void* C, const float* D,
                          const void* z, const float* delta_bias, void* out, float* x_ckpt, float* last_state,
                          int batch, int dim, int seqlen, int io_dtype, int delta_softplus, long long z_bs,
                          long long bc_bs, long long bc_ns, long long bc_ts, int nchunks128, hipStream_t s) {
  SeqArgs a{};
  a.u = u; a.delta = delta; a.z = z; a.out = out; a.A = A; a.D = D; a.delta_bias = delta_bias;
  a.B = B; a.C = C;
  a.x_ckpt = x_ckpt; a.last_state = last_state;
  a.batch = batch; a.dim = dim; a.seqlen = seqlen; a.nchunks128 = nchunks128;
  a.softplus = delta_softplus; a.z_bs = z_bs;
  a.bc_bs = bc_bs; a.bc_ns = bc_ns; a.bc_ts = bc_ts;
  return io_dtype == SIMAMBA_F32 ? launch_seq<float>(a, s) : launch_seq<bf16_t>(a, s);
}

}  // namespace simamba
