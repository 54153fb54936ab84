// Selective-scan forward, "lanes per channel" form, for gfx950.
//
// kLPC = 2 or 4 adjacent lanes share a channel and split its 16 states, so the recurrence
// h_t = a_t h_{t-1} + x_t B_t costs mul + exp2 + mul + fma + fma per (row, t, state) -- 5 VALU issues against ~9 for
// the row-scan form of scan_fwd.hip -- and a wave walks time sequentially over R = 64 / kLPC channels of one sample.
//
// Memory side (what round 1's 16-step version got wrong): a chunk is 32 steps = ONE WHOLE 128-byte line of an fp32
// row.  With 16-step chunks every load touched 64 B of a line and came back for the other half one chunk later,
// after the line had left L2: PMC FETCH_SIZE showed 1.69x the algorithmic bytes and the kernel took 130 us with
// the recurrence compiled out.  tools/mem_probe.hip prices the access shapes on the box (403 MB in + out, no
// compute): 16 rows x 64 B per instruction 103 us, 8 rows x 128 B 77 us, a plain linear stream 76 us.
//
//   * waves are independent (no workgroup barrier); a wave owns wave-private LDS tiles tD = softplus(delta),
//     tU = u (overwritten by y), [R][32] floats, 16-byte column groups XOR-swizzled by (row >> 1) & 7: the pack-wise
//     accesses of phases A / C (8 lanes x 16 B = one row) and the row-wise reads of phase B (ds_read_b128, one row
//     per kLPC lanes) are both conflict-free without padding; plus the chunk's B_t | C_t (32 steps x 32 floats),
//     staged straight from the strided operands with 16-byte loads when their layout allows;
//   * phase A (lane <-> one 16-byte pack of a row): coalesced loads of delta / u, softplus, into the tiles;
//   * phase B (lane <-> channel, NS = 16 / kLPC states in VGPRs): per step one ds_read_b128 each of B_t and C_t at
//     address 16 * (lane & 3): VGPR j then holds entry 4p + j at quad position p, identically in all 16 quads, and
//     "B_t[n] for this lane's state n" is that VGPR read through DPP quad_perm inside v_mul / v_fmac ([a,a+2,a,a+2]
//     with two lanes per channel, the identity with four) -- no broadcast instruction, no SGPR, no scalar-cache
//     traffic.  tools/scan_probe.hip prices the alternatives (LDS broadcast reads with plain VALU operands, one or
//     two channels per lane): none is faster than this at 3 - 4 waves per SIMD.  The kLPC partial y_t are combined
//     with 1 - 2 DPP quad adds per step;
//   * y_t overwrites u_t in the tile; phase C (same lane <-> pack map as A): y * silu(z), 16-byte stores;
//   * the next chunk's delta / u loads are issued right after phase A and fly during the whole recurrence.
#include "scan_common.h"

namespace simamba {

constexpr int kSeqTC = 32;                 // timesteps per chunk: one 128-byte line of an fp32 row
#ifndef SIMAMBA_SEQ_WAVES
#define SIMAMBA_SEQ_WAVES 1
#endif
// The waves never synchronise with each other, so a workgroup is ONE wave: the dispatcher then balances at wave
// granularity (with 4-wave workgroups a grid of 384 workgroups put 8 waves on half of the CUs and 4 on the rest).
constexpr int kSeqThreads = 64 * SIMAMBA_SEQ_WAVES;
constexpr int kBcPitch = 36;               // floats per staged B_t | C_t row (32 + 4)

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
  // kDt kernels: delta is not read but formed here, delta[d][t] = sum_r wdt[d][r] * dt[t][r], from the dt columns of
  // the x_proj output (token-major rows, the same rows B_t | C_t are staged from) -- csrc/xdt_proj.hip then skips
  // its delta phase and the tensor never exists
  const void* dt;                          // element (b, t, r) at b * dt_bs + t * dt_ts + r
  const void* wdt;                         // (dim, dt_rank), I/O type
  long long dt_bs, dt_ts;
  int dt_rank;
  float* x_ckpt;
  float* ckpt16;                           // state at every 16-step boundary, (batch, nck16, dim, 16): what the
  int nck16;                               // sequential backward (scan_bwd_seq.hip) starts its segments from
  float* last_state;
  int batch, dim, seqlen, nchunks128;
  int bc_mode;                             // 1 time-major packs of B / C, 2 token-major packs
  int mix_c4;                              // mixed launch: the last mix_c4 channels of a sample run four lanes per channel
  long long z_bs;
  long long bc_bs, bc_ns, bc_ts;
};

// ---- multiply / multiply-accumulate with a quad-selected first operand -----------------------------------------
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
SIMAMBA_QUAD_OPS(0, "[0,2,0,2]")
SIMAMBA_QUAD_OPS(1, "[1,3,1,3]")
#undef SIMAMBA_QUAD_OPS

// One aligned 4-element pack per lane (16 B fp32 / 8 B bf16).  The dispatcher only takes this kernel when
// rows are pack-aligned (L % 4 == 0 for fp32, L % 8 == 0 for bf16), so a pack is either entirely inside the
// sequence or entirely outside.  A pack beyond the end of the sequence (last chunk of a ragged L) re-reads the
// FIRST pack of its own row -- valid memory, finite whenever the row is -- and is neutralised once per pack, not
// per element: its delta gets a bias of -1e30, which softplus maps to exactly 0 (a_t = 1, x_t = 0: the state
// passes through), and its outputs are not stored.  No per-element guards, no divergent branches.
// Addressing is "uniform base pointer + 32-bit BYTE offset" throughout (the dispatcher guarantees every tensor
// spans < 4 GiB): global_load/store then take the base in SGPRs and one VGPR of offset, instead of a 64-bit
// VGPR address per access that the compiler hoists out of the chunk loop and spills.
template <typename T>
__device__ __forceinline__ void load4(const T* __restrict__ base, unsigned boff, float (&v)[4]) {
  const Pack<T, 4> pk = *reinterpret_cast<const Pack<T, 4>*>(reinterpret_cast<const char*>(base) + boff);
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = to_f32<T>(pk.v[i]);
}
template <typename T>
__device__ __forceinline__ void store4(T* __restrict__ base, unsigned boff, const float (&v)[4]) {
  Pack<T, 4> pk;
#pragma unroll
  for (int i = 0; i < 4; ++i) pk.v[i] = from_f32<T>(v[i]);
  *reinterpret_cast<Pack<T, 4>*>(reinterpret_cast<char*>(base) + boff) = pk;
}

// softplus(x) with x handed over as x2 = x * log2(e) (the caller folds the scale into one fma with the bias):
// 2 transcendentals + 7 plain ops, branch-free.  Above torch's threshold (x > 20) the result is x itself (and the
// exp2 overflow beyond x ~ 88 never shows); below -15 the series log(1 + e) = e keeps the relative accuracy that
// 1 + e loses.  x2 = -inf-like (-1e30 from a padded pack) gives e = 0 and exactly 0.
__device__ __forceinline__ float softplus_log2(float x2) {
  const float e = fast_exp2(x2);
  float sp = fast_log2(1.f + e) * kLn2;
  sp = (x2 < -15.f * kLog2e) ? e : sp;
  return (x2 > 20.f * kLog2e) ? x2 * kLn2 : sp;
}

// float offset of 16-byte column group g (0..7) of a tile row
__device__ __forceinline__ int tile_off(int row, int g) { return row * kSeqTC + 4 * (g ^ ((row >> 1) & 7)); }

template <int kLPC> struct SeqCfg {
  static constexpr int NS = kMaxState / kLPC;    // states per lane
  static constexpr int R = 64 / kLPC;            // channels per wave
  static constexpr int kPacks = R / 8;           // 4-step packs per lane, tensor and chunk (8 packs = one row)
  static constexpr int kTileFloats = 2 * R * kSeqTC + kSeqTC * kBcPitch;
  // waves per SIMD: what the LDS footprint admits (4 waves x kTileFloats x 4 B per workgroup, 160 KiB per CU)
  static constexpr int kWaves = kLPC == 2 ? 3 : 4;
};

// One wave's work: channels ch_base .. ch_base + R of sample b, the whole sequence.  tD: the wave's LDS tiles.
using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

template <typename T, bool kHasZ, int kLPC, bool kDt = false>
__device__ __forceinline__ void seq_body(const SeqArgs& p, const int b, const int ch_base, float* tD) {
  typedef SeqCfg<kLPC> Cfg;
  constexpr int NS = Cfg::NS, R = Cfg::R, kPacks = Cfg::kPacks;
  const int lane = threadIdx.x & 63;
  const int L = p.seqlen, D = p.dim;
  float* tU = tD + R * kSeqTC;
  float* tBC = tU + R * kSeqTC;            // [32 steps][B_t(16) | C_t(16) | pad(4)]

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

  // ---- phase A / C identity: pack pk = lane + 64 j covers row pk / 8, steps 4 (pk % 8) .. + 4 -------------
  const int q = lane & 7;
  float biasA[kPacks];
  unsigned rowoff[kPacks];
  // byte offsets off one base pointer per tensor; z sits at a wave-uniform distance from u (its batch stride
  // may differ)
  constexpr unsigned kEsz = sizeof(T);
  const unsigned zdelta = (static_cast<unsigned>(b * p.z_bs) - static_cast<unsigned>(b) * D * L) * kEsz;
#pragma unroll
  for (int j = 0; j < kPacks; ++j) {
    const int dA = min(ch_base + (lane >> 3) + 8 * j, D - 1);
    biasA[j] = (p.delta_bias ? p.delta_bias[dA] : 0.f) * kLog2e;    // softplus works in the log2 domain
    rowoff[j] = ((static_cast<unsigned>(b) * D + dA) * L + 4 * q) * kEsz;
  }
  const T* __restrict__ ug = static_cast<const T*>(p.u);
  const T* __restrict__ dg = static_cast<const T*>(p.delta);
  const T* __restrict__ zg = static_cast<const T*>(p.z);
  T* __restrict__ og = static_cast<T*>(p.out);

  // ---- B_t | C_t staging: 32 steps x 16 states per tensor and chunk = 2 packs of 4 per lane and tensor.
  //   time-major (element (n, t) at n * ns + t): a pack = 4 steps of one state, transposed into the [t][n] image by
  //     4 ds_write_b32;
  //   token-major (state stride 1: the mixer's x_proj output): a pack = 4 states of one step, one ds_write_b128.
  // Other strides / alignments never reach this kernel (the dispatcher sends them to the row-scan kernel).
  const T* __restrict__ Bg = static_cast<const T*>(p.B) + static_cast<long long>(b) * p.bc_bs;
  const T* __restrict__ Cg = static_cast<const T*>(p.C) + static_cast<long long>(b) * p.bc_bs;
  const bool tok = p.bc_mode == 2;
  const int bc_ns = static_cast<int>(p.bc_ns), bc_ts = static_cast<int>(p.bc_ts);
  // `lane_v` is an opaque copy of the lane id refreshed once per chunk: index arithmetic derived from it is
  // recomputed per chunk (a few VALU ops) instead of being hoisted into loop-invariant VGPRs and spilled
  int lane_v = lane;
  // pack e = lane + 64 i (i = 0, 1): time-major -> state e >> 3, steps 4 (e & 7) ..; token-major -> step e >> 2,
  // states 4 (e & 3) ..  Packs beyond the sequence re-read step 0; their steps carry delta = 0, x = 0, so whatever
  // they hold is multiplied away.
  float bcv[2][2][4];
  auto issue_bc = [&](int t0) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int e = lane_v + 64 * i;
      const int n = tok ? 4 * (e & 3) : (e >> 3);
      const int tl = tok ? (e >> 2) : 4 * (e & 7);
      const int t = (t0 + tl < L) ? t0 + tl : 0;
      const unsigned o = static_cast<unsigned>(n * bc_ns + t * bc_ts) * kEsz;   // bytes, < 2^32
      load4<T>(Bg, o, bcv[0][i]);
      load4<T>(Cg, o, bcv[1][i]);
    }
  };
  auto stage_bc_tile = [&]() {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int e = lane_v + 64 * i;
      if (tok) {
        float* dst = tBC + (e >> 2) * kBcPitch + 4 * (e & 3);
        *reinterpret_cast<float4*>(dst) = make_float4(bcv[0][i][0], bcv[0][i][1], bcv[0][i][2], bcv[0][i][3]);
        *reinterpret_cast<float4*>(dst + 16) = make_float4(bcv[1][i][0], bcv[1][i][1], bcv[1][i][2], bcv[1][i][3]);
      } else {
        float* dst = tBC + 4 * (e & 7) * kBcPitch + (e >> 3);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          dst[k * kBcPitch] = bcv[0][i][k];
          dst[k * kBcPitch + 16] = bcv[1][i][k];
        }
      }
    }
  };

  float dv[kPacks][4], uv[kPacks][4], zv[kPacks][4];
  auto issue_loads = [&](int t0) {
    // beyond the end of the sequence: the row's first pack (rowoff already points 4 q steps into the row)
    const unsigned toff = (t0 + 4 * q < L) ? t0 * kEsz : 0u - 4u * q * kEsz;
#pragma unroll
    for (int j = 0; j < kPacks; ++j) {
      if (!kDt) load4<T>(dg, rowoff[j] + toff, dv[j]);
      load4<T>(ug, rowoff[j] + toff, uv[j]);
    }
  };

  // ---- kDt: delta of a chunk on the matrix pipe ---------------------------------------------------------------------
  // out[i = step][j = channel] = sum_r dt[step][r] * wdt[channel][r] as ONE 32 x 32 tile per wave and chunk:
  // v_mfma_f32_32x32x2_f32 (exact fp32: a k-ordered fmaf chain) with the k order of csrc/xdt_proj.hip's delta product
  // -- MFMA m = 4 g + e contracts r = 8 g + e (lanes < 32) then r = 8 g + 4 + e (lanes >= 32) -- so the values are bit
  // for bit the ones that kernel stored; bf16: two v_mfma_f32_32x32x16_bf16 (r = 16 j + 8 h .. + 7, zero past dt_rank)
  // and one rounding to bf16, as csrc/xdt_proj_bf16.hip does.  C/D layout: lane (column j = lane & 31, half h) holds rows
  // 8 g' + 4 h + e in register 4 g' + e: four whole 4-step packs (2 g' + h) of ONE channel -- the shape phase A wants.
  // With four lanes per channel the wave has 16 channels: columns 16-31 repeat them and are not stored.
  const float biasM = (kDt && p.delta_bias) ? p.delta_bias[min(ch_base + ((lane & 31) % R), D - 1)] * kLog2e : 0.f;
  f32x16 dacc;
  auto form_delta = [&](int t0) {
    const int mj = lane_v & 31, mh = lane_v >> 5;            // from the opaque lane id: nothing of this is loop-carried
    const int d_m = min(ch_base + (mj % R), D - 1);
    const int t = (t0 + mj < L) ? t0 + mj : 0;               // beyond the sequence: step 0 (its delta is overridden)
    const T* dtrow = static_cast<const T*>(p.dt) + static_cast<long long>(b) * p.dt_bs + static_cast<long long>(t) * p.dt_ts;
    const T* wrow = static_cast<const T*>(p.wdt) + static_cast<size_t>(d_m) * p.dt_rank;
#pragma unroll
    for (int i = 0; i < 16; ++i) dacc[i] = 0.f;
    if constexpr (sizeof(T) == 4) {
      float a[12], w[12];
#pragma unroll
      for (int g = 0; g < 3; ++g) {
        // unconditional loads (clamped column group, zeroed afterwards): a predicated load makes hipcc drain vmcnt(0)
        // before every use.  dt_rank % 4 == 0: four r are all in or all out
        const int rc = min(8 * g + 4 * mh, p.dt_rank - 4);
        float4 a4 = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(dtrow) + rc);
        const float4 w4 = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(wrow) + rc);
        if (8 * g + 4 * mh >= p.dt_rank) a4 = make_float4(0.f, 0.f, 0.f, 0.f);
        a[4 * g] = a4.x; a[4 * g + 1] = a4.y; a[4 * g + 2] = a4.z; a[4 * g + 3] = a4.w;
        w[4 * g] = w4.x; w[4 * g + 1] = w4.y; w[4 * g + 2] = w4.z; w[4 * g + 3] = w4.w;
      }
#pragma unroll
      for (int m = 0; m < 12; ++m) dacc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m], w[m], dacc, 0, 0, 0);
    } else {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int kc = min(16 * j + 8 * mh, p.dt_rank - 8);  // dt_rank % 8 == 0; clamped, zeroed below
        uint4 a8 = *reinterpret_cast<const uint4*>(reinterpret_cast<const uint16_t*>(dtrow) + kc);
        const uint4 w8 = *reinterpret_cast<const uint4*>(reinterpret_cast<const uint16_t*>(wrow) + kc);
        if (16 * j + 8 * mh >= p.dt_rank) a8 = make_uint4(0u, 0u, 0u, 0u);
        dacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a8), __builtin_bit_cast(bf16x8, w8),
                                                      dacc, 0, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) dacc[i] = bf16_to_f32(f32_to_bf16(dacc[i]));   // delta is a bf16 tensor there
    }
  };

  const int nchunks = (L + kSeqTC - 1) / kSeqTC;
  issue_bc(0);
  issue_loads(0);
  if (kDt) form_delta(0);
  for (int c = 0; c < nchunks; ++c) {
    const int t0 = c * kSeqTC;
    asm volatile("" : "+v"(lane_v));
    const int qv = lane_v & 7, prow = lane_v >> 3;
    const bool in_seq = t0 + 4 * qv < L;
    // ---- phase A: softplus(delta), u and the chunk's B | C into the tiles ------------------------------------
#pragma unroll
    for (int j = 0; j < kPacks; ++j) {
      const int o = tile_off(prow + 8 * j, qv);
      if (!kDt) {
        const float b2 = in_seq ? biasA[j] : -1e30f;          // padded pack: softplus -> 0, the identity map
        float dl[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) dl[i] = softplus_log2(fmaf(dv[j][i], kLog2e, b2));
        *reinterpret_cast<float4*>(tD + o) = make_float4(dl[0], dl[1], dl[2], dl[3]);
      }
      *reinterpret_cast<float4*>(tU + o) = make_float4(uv[j][0], uv[j][1], uv[j][2], uv[j][3]);
    }
    if (kDt) {                                                // delta from the accumulator: packs 2 g' + h of channel mj
      const int mj = lane_v & 31, mh = lane_v >> 5;
      const bool m_valid = mj < R;
#pragma unroll
      for (int gp = 0; gp < 4; ++gp) {
        const int qm = 2 * gp + mh;
        const float b2 = (t0 + 4 * qm < L) ? biasM : -1e30f;
        float dl[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) dl[e] = softplus_log2(fmaf(dacc[4 * gp + e], kLog2e, b2));
        if (m_valid) *reinterpret_cast<float4*>(tD + tile_off(mj, qm)) = make_float4(dl[0], dl[1], dl[2], dl[3]);
      }
    }
    stage_bc_tile();
    // dv / uv are dead after phase A: the next chunk's loads fly during the whole recurrence (kDt: its delta tile is
    // formed now -- 12 MFMAs on operands that come from L2 -- and waits in 16 registers where dv waited)
    // (kDt: unconditional -- the last iteration re-requests chunk 0, harmlessly: under `if (c + 1 < nchunks)` the PREVIOUS
    // contents of the accumulator would stay alive through the whole recurrence as the other arm of a phi)
    if (kDt) {
      const int tn = (c + 1 < nchunks) ? t0 + kSeqTC : 0;
      form_delta(tn);
      issue_loads(tn);
    } else if (c + 1 < nchunks) {
      issue_loads(t0 + kSeqTC);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();

    // ---- phase B: the recurrence ------------------------------------------------------------------------------
    // LDS reads run one step (B_t | C_t) ahead of their use; the sched_barriers keep hipcc from sinking them
    // back next to the consumer (it did, exposing ~200 cycles of LDS latency per step)
    const int q3 = lane_v & 3;
    // float offset of column group g of this lane's row: rowB * 32 + 4 (g ^ s) = (rowB * 32 + 4 s) ^ (4 g) -- the XOR
    // only touches the three bits below the row's stride
    const int rowBv = lane_v / kLPC;
    const int tbase = rowBv * kSeqTC + 4 * ((rowBv >> 1) & 7);
    float4 b4n = *reinterpret_cast<const float4*>(tBC + 4 * q3);
    float4 c4n = *reinterpret_cast<const float4*>(tBC + 16 + 4 * q3);
#pragma unroll 1
    for (int g = 0; g < kSeqTC / 4; ++g) {
      const int o = tbase ^ (4 * g);
      const float4 d4 = *reinterpret_cast<const float4*>(tD + o);
      const float4 u4 = *reinterpret_cast<const float4*>(tU + o);
      const float dl[4] = {d4.x, d4.y, d4.z, d4.w};
      const float uu[4] = {u4.x, u4.y, u4.z, u4.w};
      // z of this chunk is requested half way through the recurrence: early enough that its HBM latency is over
      // by phase C even when this wave is alone on its SIMD, late enough that its registers are free before
      if (kHasZ && g == kSeqTC / 8) {
#pragma unroll
        for (int j = 0; j < kPacks; ++j)
          load4<T>(zg, rowoff[j] + zdelta + (in_seq ? t0 * kEsz : 0u - 4u * qv * kEsz), zv[j]);
      }
      float yy[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float vB[4] = {b4n.x, b4n.y, b4n.z, b4n.w};
        const float vC[4] = {c4n.x, c4n.y, c4n.z, c4n.w};
        {
          const int tn = (4 * g + i + 1) & (kSeqTC - 1);      // wraps on the last step: harmless re-read
          b4n = *reinterpret_cast<const float4*>(tBC + tn * kBcPitch + 4 * q3);
          c4n = *reinterpret_cast<const float4*>(tBC + tn * kBcPitch + 16 + 4 * q3);
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
            if constexpr (kLPC == 4) {                   // quad position p holds entries 4p .. 4p+3: the lane's own
              h[n] = fmaf(e, h[n], xx * vB[j]);
              y = fmaf(h[n], vC[j], y);
            } else {                                     // even lanes: entries 4a + j, odd lanes: 8 + 4a + j
              float xb;
              if (a == 0) xb = mul_q<0>(vB[j], xx); else xb = mul_q<1>(vB[j], xx);
              h[n] = fmaf(e, h[n], xb);
              if (a == 0) fmac_q<0>(y, vC[j], h[n]); else fmac_q<1>(y, vC[j], h[n]);
            }
          }
        }
        float y = ys[0] + ys[1];
        // the kLPC partial sums of a channel, one fused DPP add per stage (hipcc makes v_mov_b32_dpp + v_add_f32 of the
        // update_dpp builtin); s_nop: "VALU write -> DPP read: 2 wait states", not padded inside asm
        if (kLPC == 4)
          asm("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
              "s_nop 1\n\tv_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf" : "+v"(y));
        else
          asm("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(y));
        yy[i] = y;
        __builtin_amdgcn_sched_barrier(0);
      }
      // all kLPC lanes of a channel hold the same sums and store them to the same place (no EXEC games
      // next to DPP code)
      *reinterpret_cast<float4*>(tU + o) = make_float4(yy[0], yy[1], yy[2], yy[3]);
      // the state after steps 15 and 31 of the chunk, for a backward that recomputes 16-step segments
      if (p.ckpt16 && (g & 3) == 3 && own_valid) {
        const int kb = (t0 >> 4) + (g >> 2);
        if (kb < p.nck16) {
          float4* dst = reinterpret_cast<float4*>(
              p.ckpt16 + ((static_cast<size_t>(b) * p.nck16 + kb) * D + d_own) * kMaxState + n0);
#pragma unroll
          for (int k = 0; k < NS / 4; ++k) dst[k] = make_float4(h[4 * k], h[4 * k + 1], h[4 * k + 2], h[4 * k + 3]);
        }
      }
    }
    // B | C of the next chunk: requested only now so that their registers are free during the recurrence; the
    // latency hides under phase C and phase A
    if (c + 1 < nchunks) issue_bc(t0 + kSeqTC);
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
      const float4 y4 = *reinterpret_cast<const float4*>(tU + tile_off(prow + 8 * j, qv));
      const float yv[4] = {y4.x, y4.y, y4.z, y4.w};
      float o[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float v = yv[i];
        if (kHasZ) v = v * zv[j][i] * sigmoid_f(zv[j][i]);
        o[i] = v;
      }
      if (in_seq && ch_base + prow + 8 * j < D) store4<T>(og, rowoff[j] + t0 * kEsz, o);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}

// (four lanes per channel with the delta tile: its 32 x 32 accumulator is 16 registers where the 16-channel wave kept 8
// of loaded delta -- held to three waves per SIMD instead of spilling at four)
template <typename T, bool kHasZ, int kLPC, bool kDt = false>
__global__ __launch_bounds__(kSeqThreads, SeqCfg<kLPC>::kWaves - ((kDt && kLPC == 4) ? 1 : 0)) void scan_fwd_seq_kernel(SeqArgs p) {
  typedef SeqCfg<kLPC> Cfg;
  __shared__ __attribute__((aligned(16))) float sMem[kSeqThreads / 64][Cfg::kTileFloats];
  int tile_id, b;
  xcd_tile(tile_id, b);
  const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) >> 6);   // SGPR
  const int ch_base = (tile_id * (kSeqThreads / 64) + wave) * Cfg::R;  // first channel of this wave
  if (ch_base >= p.dim) return;                                        // whole wave idle (no barriers here)
  seq_body<T, kHasZ, kLPC, kDt>(p, b, ch_base, &sMem[wave][0]);
}

// Both forms in one launch of one-wave workgroups: the first dim - mix_c4 channels of every sample two lanes per
// channel (32 channels a wave), the last mix_c4 four lanes per channel (16 a wave, 0.68 of the instructions).  At
// (64, 768, 1024) two lanes per channel is 1536 waves on 1024 SIMDs -- half of the SIMDs carry two waves, the kernel
// lasts two wave-times; with the last 256 channels of every sample on four lanes it is 1024 + 1024 waves, one of
// each per SIMD: 1.68 wave-times.  Block order: an XCD receives ids x, x + 8, ...; it is handed its two-lane waves
// first (one per SIMD while they last) and the four-lane ones after, and both parts of a sample stay on one XCD (they
// share B_t / C_t in its L2).
// Measured (tools/bench_scan_context.py, (64, 768, 1024) fp32): 245-254 us against 285 us for the two-lane launch when
// the launch follows itself or a small GEMM -- but 340-347 us against 303 us right after a kernel that left 400 MB of
// dirty lines in L2 / MALL, which is how the mixer calls it (conv + x_proj + dt_proj have just written x_conv and
// delta): there the launch competes with that write-back for HBM and finishing the same bytes in less time is not
// available.  Inside the model: 347 us against 302 us (rocprofv3).  So SIMAMBA_SCAN_AUTO does not pick it; it stays
// an explicit variant with its parity tests.
template <typename T, bool kHasZ, bool kDt = false>
__global__ __launch_bounds__(64, SeqCfg<2>::kWaves) void scan_fwd_seq_mix_kernel(SeqArgs p) {
  __shared__ __attribute__((aligned(16))) float sMem[SeqCfg<2>::kTileFloats];
  const int g2 = (p.dim - p.mix_c4) / SeqCfg<2>::R, g4 = p.mix_c4 / SeqCfg<4>::R;   // waves per sample of each form
  int x = 0, k = blockIdx.x, bx = p.batch;
  if (p.batch % kNumXcd == 0) { x = k % kNumXcd; k /= kNumXcd; bx = p.batch / kNumXcd; }
  const int n2 = bx * g2;
  if (k < n2) {
    const int sample = k / g2;
    seq_body<T, kHasZ, 2, kDt>(p, x * bx + sample, (k - sample * g2) * SeqCfg<2>::R, sMem);
  } else {
    k -= n2;
    const int sample = k / g4;
    seq_body<T, kHasZ, 4, kDt>(p, x * bx + sample, p.dim - p.mix_c4 + (k - sample * g4) * SeqCfg<4>::R, sMem);
  }
}

template <typename T, int kLPC>
static void launch_seq_lpc(const SeqArgs& a, hipStream_t s) {
  const int ch_per_wg = (kSeqThreads / 64) * SeqCfg<kLPC>::R;
  dim3 grid((a.dim + ch_per_wg - 1) / ch_per_wg, a.batch);
  if (a.dt) {                                              // the mixer's form: gated, delta formed in the kernel
    hipLaunchKernelGGL((scan_fwd_seq_kernel<T, true, kLPC, true>), grid, dim3(kSeqThreads), 0, s, a);
    return;
  }
  if (a.z)
    hipLaunchKernelGGL((scan_fwd_seq_kernel<T, true, kLPC>), grid, dim3(kSeqThreads), 0, s, a);
  else
    hipLaunchKernelGGL((scan_fwd_seq_kernel<T, false, kLPC>), grid, dim3(kSeqThreads), 0, s, a);
}

// How this kernel would read B and C: 1 = 16-byte (fp32) / 8-byte (bf16) packs along time (t is the unit-stride axis),
// 2 = packs along the state (n is: the mixer's x_proj output), 0 = neither keeps a pack aligned -- such operands go
// to the row-scan kernel, which gathers element by element.
int scan_fwd_seq_bc_mode(const void* B, const void* C, int io_dtype, long long bc_bs, long long bc_ns, long long bc_ts) {
  const size_t esz = io_dtype == SIMAMBA_F32 ? 4 : 2;
  const uintptr_t pm = 4 * esz - 1;
  if (((reinterpret_cast<uintptr_t>(B) | reinterpret_cast<uintptr_t>(C)) & pm) != 0 || bc_bs % 4 != 0) return 0;
  if (bc_ts == 1 && bc_ns % 4 == 0) return 1;
  if (bc_ns == 1 && bc_ts % 4 == 0) return 2;
  return 0;
}

// Channels per sample to run four lanes per channel in the mixed launch, 0 when mixing does not help.  With two lanes
// per channel the launch is n = rows / 32 waves on 1024 SIMDs: q full rounds and r left over.  0 < r <= 512: the
// left-over waves' channels go four lanes per channel instead (2 r waves of 0.68 wave-times each, at most one per
// SIMD) and the kernel lasts q + 0.68 wave-times instead of q + 1.  r > 512 has no such split (some SIMD would
// carry a two-lane wave of the last round anyway).
int scan_fwd_seq_mix_c4(int batch, int dim) {
  constexpr long long kSimds = 1024;
  if (kSeqThreads != 64 || dim % 32 || batch <= 0) return 0;
  const long long n = static_cast<long long>(batch) * dim / 32;
  const long long q = n / kSimds, r = n % kSimds;
  if (r == 0 || r > kSimds / 2 || q + 1 > SeqCfg<2>::kWaves) return 0;
  if ((32 * r) % batch) return 0;
  const long long c4 = 32 * r / batch;
  if (c4 % 32 || c4 > dim) return 0;
  return static_cast<int>(c4);
}

// Entry used by simamba_selective_scan_fwd (scan_fwd.hip) when the shape qualifies (16 states, softplus on,
// pack-aligned rows and B / C, 32-bit byte offsets); lpc = 2 or 4, or 6 = the mixed launch (scan_fwd_seq_mix_c4 != 0).
int scan_fwd_seq_dispatch(const void* u, const void* delta, const float* A, const void* B, const void* C, const float* D,
                          const void* z, const float* delta_bias, void* out, float* x_ckpt, int ckpt_step,
                          float* last_state,
                          int batch, int dim, int seqlen, int io_dtype, long long z_bs,
                          long long bc_bs, long long bc_ns, long long bc_ts, int nchunks128, int lpc, hipStream_t s,
                          const void* dt, const void* wdt, long long dt_bs, long long dt_ts, int dt_rank) {
  SeqArgs a{};
  a.dt = dt; a.wdt = wdt; a.dt_bs = dt_bs; a.dt_ts = dt_ts; a.dt_rank = dt_rank;
  a.u = u; a.delta = delta; a.z = z; a.out = out; a.A = A; a.D = D; a.delta_bias = delta_bias;
  a.B = B; a.C = C;
  if (ckpt_step == SIMAMBA_SCAN_CKPT_SEQ) { a.ckpt16 = x_ckpt; a.nck16 = (seqlen + 15) / 16; }
  else a.x_ckpt = x_ckpt;
  a.last_state = last_state;
  a.batch = batch; a.dim = dim; a.seqlen = seqlen; a.nchunks128 = nchunks128;
  a.z_bs = z_bs;
  a.bc_bs = bc_bs; a.bc_ns = bc_ns; a.bc_ts = bc_ts;
  a.bc_mode = scan_fwd_seq_bc_mode(B, C, io_dtype, bc_bs, bc_ns, bc_ts);
  if (a.bc_mode == 0) return SIMAMBA_E_VARIANT;
  if (lpc == 6) {
    a.mix_c4 = scan_fwd_seq_mix_c4(batch, dim);
    if (a.mix_c4 == 0) return SIMAMBA_E_VARIANT;
    const unsigned grid = static_cast<unsigned>(batch) * ((dim - a.mix_c4) / SeqCfg<2>::R + a.mix_c4 / SeqCfg<4>::R);
    if (a.dt) {
      if (io_dtype == SIMAMBA_F32) hipLaunchKernelGGL((scan_fwd_seq_mix_kernel<float, true, true>), dim3(grid), dim3(64), 0, s, a);
      else hipLaunchKernelGGL((scan_fwd_seq_mix_kernel<bf16_t, true, true>), dim3(grid), dim3(64), 0, s, a);
      return static_cast<int>(hipGetLastError());
    }
    if (io_dtype == SIMAMBA_F32) {
      if (z) hipLaunchKernelGGL((scan_fwd_seq_mix_kernel<float, true>), dim3(grid), dim3(64), 0, s, a);
      else hipLaunchKernelGGL((scan_fwd_seq_mix_kernel<float, false>), dim3(grid), dim3(64), 0, s, a);
    } else {
      if (z) hipLaunchKernelGGL((scan_fwd_seq_mix_kernel<bf16_t, true>), dim3(grid), dim3(64), 0, s, a);
      else hipLaunchKernelGGL((scan_fwd_seq_mix_kernel<bf16_t, false>), dim3(grid), dim3(64), 0, s, a);
    }
    return static_cast<int>(hipGetLastError());
  }
  if (io_dtype == SIMAMBA_F32) {
    if (lpc == 2) launch_seq_lpc<float, 2>(a, s); else launch_seq_lpc<float, 4>(a, s);
  } else {
    if (lpc == 2) launch_seq_lpc<bf16_t, 2>(a, s); else launch_seq_lpc<bf16_t, 4>(a, s);
  }
  return static_cast<int>(hipGetLastError());
}

}  // namespace simamba
