// Selective-scan backward, sequential "lanes per channel" form, for gfx950.
//
// Replaces selective_scan_cuda.bwd of mamba-ssm (autograd of the call at the reference's models/block.py:72) for
// the shapes the lanes-per-channel forward (scan_fwd_seq.hip) takes.  Where scan_bwd.hip spreads TIME over the 16
// lanes of a DPP row and pays two 4-step DPP scans, chunk carries through LDS and fix-ups per (row, state), this
// kernel walks time sequentially:
//
//   * four adjacent-stride lanes share a channel and split its 16 states (4 each, in VGPRs); a wave carries 16
//     channels of one sample, a 256-thread workgroup 64, and the workgroup's four waves share the B_t | C_t tile
//     of the chunk (staged once) and sum their dB / dC partials before ONE float-atomic flush per chunk;
//   * the forward has left the state h at every 16-step boundary (x_ckpt, layout (batch, ceil(L/16), dim, 16));
//     per 16-step segment, right to left:  (1) recompute h_t from the boundary state, keeping all 16 x 4 values
//     in registers (and y_t = C_t . h_t for dz on the way);  (2) walk the adjoint g_t = a_{t+1} g_{t+1} + C_t dy_t
//     backwards with g in registers across segments and chunks -- no scan, no carry, no LDS traffic for state;
//   * dB_t[n] = sum_d g x, dC_t[n] = sum_d dy h (sums over channels = over lanes): per step the lane's 8 products
//     go through a transposing reduction -- v_permlane32_swap, v_permlane16_swap, then two DPP stages inside the
//     quad -- 15 VALU for 8 values over 16 lanes, leaving ONE finished sum per lane pair;
//   * per-(channel, step) work that needs transcendentals (softplus, SiLU, sigmoid) runs in the coalesced
//     "load layout" (lane <-> one 16-byte pack of a row, phases A and C), everything else in the channel layout
//     (phase B); the two meet in wave-private LDS tiles [16 rows][32 steps], 16-byte column groups XOR-swizzled
//     by (row >> 1) & 7 (conflict-free from both sides), updated in place: delta stays, u -> du, dy -> the
//     ddelta pre-factor, dout silu'(z) -> dz.
//
// Lane map of phase B: lane = 16 r + 4 cq + p -> channel 4 r + cq of the wave, states 4 p .. 4 p + 3: a quad is one
// channel.  "B_t[n] for my states" is one ds_read_b128 at 16 p (plain VALU operands, no DPP), the sums over states
// are two quad_perm DPP adds, and the channel reduction's in-row stages are row rotations whose destination is
// masked per bank (bank = 4 consecutive lanes = one channel).
//
// Waves: (batch * dim / 16), e.g. 3072 at (64, 768): three per SIMD, all resident at once (the row-scan kernel and a
// two-lanes-per-channel form would leave 1536 waves on 1024 SIMDs).  Registers <= 168 (3 waves per SIMD), LDS
// 52.5 KB per workgroup (3 per CU).
// Algorithmic HBM bytes: 7*B*D*L*s + 2*B*N*L*(s+4) + the checkpoints B*D*L*4 (read) + small.
#include "scan_common.h"
#include "scan_xlane.h"

namespace simamba {

#ifndef SIMAMBA_BWDSEQ_OCC
#define SIMAMBA_BWDSEQ_OCC 3                   // waves per SIMD the register allocation is held to
#endif
constexpr int kBsTC = 32;                  // steps per chunk: one 128-byte line of an fp32 row
constexpr int kBsSeg = 16;                 // steps per segment = distance of the forward's checkpoints
constexpr int kBsR = 16;                   // channels per wave
constexpr int kBsWaves = 4;
constexpr int kBsThreads = 64 * kBsWaves;
constexpr int kBsCh = kBsR * kBsWaves;     // channels per workgroup
constexpr int kBsTile = kBsR * kBsTC;      // floats per tile
constexpr int kBsPPitch = 33;              // dB | dC partial tile [32 steps][32 + 1]
constexpr int kBsWaveFloats = 4 * kBsTile + kBsTC * kBsPPitch;
constexpr int kBsSmemFloats = kBsWaves * kBsWaveFloats + kBsTC * 32;

struct BwdSeqArgs {
  const void* u; const void* delta; const void* z; const void* dout;
  void* du; void* ddelta; void* dz;
  const float* A; const float* D; const float* delta_bias;
  const void* B; const void* C;
  // kDt kernels: delta is formed here from the dt columns of the x_proj output, as in csrc/scan_fwd_seq.hip
  const void* dt; const void* wdt;         // element (b, t, r) at b * dt_bs + t * dt_ts + r; (dim, dt_rank), I/O type
  long long dt_bs, dt_ts;
  int dt_rank;
  const float* ckpt;                       // (batch, nck, dim, 16)
  float* dA; float* dB; float* dC; float* dD; float* ddelta_bias;
  int batch, dim, seqlen, nck;
  int bc_mode;                             // 1 time-major packs of B / C, 2 token-major packs
  long long z_bs, dz_bs;
  long long bc_bs, bc_ns, bc_ts;
};

template <typename T>
__device__ __forceinline__ void bs_load4(const T* __restrict__ base, unsigned boff, float (&v)[4]) {
  const Pack<T, 4> pk = *reinterpret_cast<const Pack<T, 4>*>(reinterpret_cast<const char*>(base) + boff);
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = to_f32<T>(pk.v[i]);
}
template <typename T>
__device__ __forceinline__ void bs_store4(T* __restrict__ base, unsigned boff, const float (&v)[4]) {
  Pack<T, 4> pk;
#pragma unroll
  for (int i = 0; i < 4; ++i) pk.v[i] = from_f32<T>(v[i]);
  *reinterpret_cast<Pack<T, 4>*>(reinterpret_cast<char*>(base) + boff) = pk;
}

// softplus of x handed over as x * log2(e); see scan_fwd_seq.hip
__device__ __forceinline__ float bs_softplus_log2(float x2) {
  const float e = fast_exp2(x2);
  float sp = fast_log2(1.f + e) * kLn2;
  sp = (x2 < -15.f * kLog2e) ? e : sp;
  return (x2 > 20.f * kLog2e) ? x2 * kLn2 : sp;
}

__device__ __forceinline__ int bs_tile_off(int row, int g) { return row * kBsTC + 4 * (g ^ ((row >> 1) & 7)); }

// The last two stages of the channel reduction, inside a 16-lane row (4 channels = 4 banks of 4 lanes), in one block.
// Channel bit 1 (lanes 8 apart), transposing: banks 0,1 end with w0 summed over the pair, banks 2,3 with w1 -- two
// DPP adds into one register, each writing only its banks (bank_mask masks the destination write per group of 4
// lanes).  Channel bit 0 (lanes 4 apart): banks 1 and 3 add their left neighbour's value; they hold the finished
// sums.  s_nop: VALU write -> DPP read needs 2 wait states.
__device__ __forceinline__ float bs_row_finish(float w0, float w1) {
  float d;
  asm("s_nop 1\n\t"
      "v_add_f32_dpp %0, %1, %1 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
      "v_add_f32_dpp %0, %2, %2 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
      "s_nop 1\n\t"
      "v_add_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xa"
      : "=&v"(d)
      : "v"(w0), "v"(w1));
  return d;
}

// Sum over the 4 lanes of a quad (= the 16 states of a channel), result in all of them: one fused DPP add per stage
// (hipcc emits v_mov_b32_dpp + v_add_f32 for the update_dpp builtin).  The s_nops are the "VALU write -> DPP read:
// 2 wait states" hazard, which hipcc does not pad inside asm; the two-value form interleaves two reductions so that
// each one's instruction is the other's wait state.
__device__ __forceinline__ void bs_quad_sum1(float& a) {
  asm("s_nop 1\n\t"
      "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf"
      : "+v"(a));
}
__device__ __forceinline__ void bs_quad_sum2(float& a, float& b) {
  asm("s_nop 1\n\t"
      "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 0\n\t"
      "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf"
      : "+v"(a), "+v"(b));
}

using f32x4_t = __attribute__((ext_vector_type(4))) float;
using bf16x8_t = __attribute__((ext_vector_type(8))) __bf16;

template <typename T, bool kHasZ, bool kDt = false>
__global__ __launch_bounds__(kBsThreads, SIMAMBA_BWDSEQ_OCC) void scan_bwd_seq_kernel(BwdSeqArgs p) {
  __shared__ __attribute__((aligned(16))) float smem[kBsSmemFloats];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int tile_id, b;
  xcd_tile(tile_id, b);
  const int L = p.seqlen, D = p.dim;
  const int d0w = tile_id * kBsCh + wave * kBsR;           // first channel of this wave
  float* tD = smem + wave * kBsWaveFloats;                 // softplus(delta)
  float* tU = tD + kBsTile;                                // u            -> du
  float* tY = tU + kBsTile;                                // dy           -> u dxs + ln2 dda (ddelta before its sigmoid)
  float* tZ = tY + kBsTile;                                // dout silu'(z) -> dz
  float* tP = tZ + kBsTile;                                // this wave's dB | dC sums of the chunk, [t][f], pitch 33
  float* tBC = smem + kBsWaves * kBsWaveFloats;            // [32 steps][B_t(16) | C_t(16)], shared by the 4 waves

  // ---- phase B identity ---------------------------------------------------------------------------------------
  const int r4 = lane >> 4, cq = (lane >> 2) & 3, pq = lane & 3;
  const int ch = 4 * r4 + cq;
  const int dch = d0w + ch;
  float A2[4];
  {
    const float4 a4 = *reinterpret_cast<const float4*>(p.A + static_cast<size_t>(dch) * kMaxState + 4 * pq);
    A2[0] = a4.x * kLog2e; A2[1] = a4.y * kLog2e; A2[2] = a4.z * kLog2e; A2[3] = a4.w * kLog2e;
  }
  const float Dfull = p.D ? p.D[dch] : 0.f;
  float ga[4] = {0.f, 0.f, 0.f, 0.f};                      // a_{t+1} g_{t+1}: the adjoint state, carried through time
  float dAacc[4] = {0.f, 0.f, 0.f, 0.f};
  float dDacc = 0.f;
  const int sw = (ch >> 1) & 7;
  const int trow = ch * kBsTC;
  // element f of a dB | dC row that this lane's finished sum belongs to (see the reduction below)
  const int pf = (r4 >> 1) * 16 + 4 * pq + 2 * (r4 & 1) + (cq >> 1);
  const float* ckl = p.ckpt + (static_cast<size_t>(b) * p.nck * D + dch) * kMaxState + 4 * pq;   // + block * D * 16

  // ---- phase A / C identity: pack pk = lane + 64 j covers row pk / 8, steps 4 (pk % 8) .. + 4 ----------------
  // Phase B needs every register it can get for h (64) and its working set, so NOTHING of phases A / C lives through
  // it: byte / tile offsets are recomputed from the hardware lane id where they are used, the bias is re-read with the
  // chunk's operands, and the per-channel sum of ddelta goes out per chunk.
  constexpr unsigned kEsz = sizeof(T);
  // z / dz sit at a wave-uniform distance from u / du (their batch strides may differ)
  const unsigned zdelta = (static_cast<unsigned>(b * p.z_bs) - static_cast<unsigned>(b) * D * L) * kEsz;
  const unsigned dzdelta = (static_cast<unsigned>(b * p.dz_bs) - static_cast<unsigned>(b) * D * L) * kEsz;
  const T* __restrict__ ug = static_cast<const T*>(p.u);
  const T* __restrict__ dg = static_cast<const T*>(p.delta);
  const T* __restrict__ zg = static_cast<const T*>(p.z);
  const T* __restrict__ gg = static_cast<const T*>(p.dout);
  T* __restrict__ dug = static_cast<T*>(p.du);
  T* __restrict__ ddg = static_cast<T*>(p.ddelta);
  T* __restrict__ dzg = static_cast<T*>(p.dz);
  auto lane_now = [&]() -> int {                            // 2 VALU; never a loop-carried register
    int l = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    asm volatile("" : "+v"(l));
    return l;
  };
  auto row_off = [&](int l, int j) -> unsigned {
    return ((static_cast<unsigned>(b) * D + d0w + (l >> 3) + 8 * j) * L + 4 * (l & 7)) * kEsz;
  };

  // ---- B_t | C_t staging: one 4-pack per thread and chunk ------------------------------------------------------
  //   token-major (state stride 1: the mixer's x_proj output): thread -> step tid >> 3, states 4 (tid & 3) .. of
  //     B (tid & 4 == 0) or C: one 16-byte load, one ds_write_b128;
  //   time-major: thread -> tensor tid >> 7, state (tid >> 3) & 15, steps 4 (tid & 7) ..: transposed by 4 ds_write_b32.
  const bool tok = p.bc_mode == 2;
  const T* __restrict__ Bgp = static_cast<const T*>(p.B) + static_cast<long long>(b) * p.bc_bs;
  const T* __restrict__ Cgp = static_cast<const T*>(p.C) + static_cast<long long>(b) * p.bc_bs;
  float bcv[4];
  auto issue_bc = [&](int t0, int tv) {
    const bool isC = tok ? (tv & 4) != 0 : (tv >> 7) != 0;
    const int bc_n = tok ? 4 * (tv & 3) : ((tv >> 3) & 15);
    const int bc_tl = tok ? (tv >> 3) : 4 * (tv & 7);
    const int t = (t0 + bc_tl < L) ? t0 + bc_tl : 0;         // beyond the sequence: step 0 (multiplied away)
    const unsigned o = static_cast<unsigned>(bc_n * static_cast<int>(p.bc_ns) + t * static_cast<int>(p.bc_ts)) * kEsz;
    float vb[4], vc[4];                                      // both tensors from uniform bases, one kept
    if (isC) bs_load4<T>(Cgp, o, vc); else bs_load4<T>(Bgp, o, vb);
#pragma unroll
    for (int k = 0; k < 4; ++k) bcv[k] = isC ? vc[k] : vb[k];
  };
  auto stage_bc = [&](int tv) {
    if (tok) {
      *reinterpret_cast<float4*>(tBC + (tv >> 3) * 32 + 4 * (tv & 7)) = make_float4(bcv[0], bcv[1], bcv[2], bcv[3]);
    } else {
      float* dst = tBC + 4 * (tv & 7) * 32 + (tv >> 7) * 16 + ((tv >> 3) & 15);
#pragma unroll
      for (int k = 0; k < 4; ++k) dst[k * 32] = bcv[k];
    }
  };

  // operands of one chunk, requested one chunk ahead: the packs of the load layout, the biases of their rows and the
  // state entering the chunk's right segment
  float dv[2][4], uv[2][4], gv[2][4], zv[2][4], hsn[4], biasA[2];
  // kDt: delta[step][channel] = sum_r dt[step][r] * wdt[channel][r] of the chunk as two 16 x 16 tiles of the matrix pipe
  // (16 channels per wave): v_mfma_f32_16x16x4_f32, whose four k per instruction are chained in the order the forward's
  // 32x32x2 pairs are -- r = 8 g, 8 g + 4, 8 g + 1, 8 g + 5, then 8 g + 2, + 6, + 3, + 7 -- so the values are the forward's
  // bit for bit; bf16: one v_mfma_f32_16x16x32_bf16 per tile (r = 8 k .. + 7, zero past dt_rank), rounded to bf16.
  // C/D layout: lane (column = channel l & 15, k = l >> 4) holds rows 4 k .. 4 k + 3: ONE pack of 4 steps per tile,
  // dv[tile] = pack 4 tile + (l >> 4) of channel l & 15; biasA[0] its channel's bias.
  auto form_delta = [&](int t0, int l) {
    const int mc = l & 15, mk = l >> 4;
    const T* wrow = static_cast<const T*>(p.wdt) + static_cast<size_t>(d0w + mc) * p.dt_rank;
    biasA[0] = p.delta_bias ? p.delta_bias[d0w + mc] : 0.f;
#pragma unroll
    for (int tile = 0; tile < 2; ++tile) {
      const int tt = t0 + 16 * tile + mc;
      const T* dtrow = static_cast<const T*>(p.dt) + static_cast<long long>(b) * p.dt_bs +
                       static_cast<long long>(tt < L ? tt : 0) * p.dt_ts;          // beyond the sequence: step 0
      f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
      if constexpr (sizeof(T) == 4) {
        const int c0 = (mk & 1) * 4 + (mk >> 1);
        // all twelve loads go out unconditionally (clamped column, value zeroed afterwards): a predicated load makes
        // hipcc drain vmcnt(0) before every use -- twenty-four serial L2 round trips per chunk instead of one
        float a[6], w[6];
#pragma unroll
        for (int m = 0; m < 6; ++m) {
          const int r = 8 * (m >> 1) + c0 + 2 * (m & 1);
          const int rc = min(r, p.dt_rank - 1);
          a[m] = reinterpret_cast<const float*>(dtrow)[rc];
          w[m] = reinterpret_cast<const float*>(wrow)[rc];
        }
#pragma unroll
        for (int m = 0; m < 6; ++m) {
          const int r = 8 * (m >> 1) + c0 + 2 * (m & 1);
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(r < p.dt_rank ? a[m] : 0.f, w[m], acc, 0, 0, 0);
        }
      } else {
        const int kc = min(8 * mk, p.dt_rank - 8);           // dt_rank % 8 == 0; clamped, zeroed below (no predicated load)
        uint4 a8 = *reinterpret_cast<const uint4*>(reinterpret_cast<const uint16_t*>(dtrow) + kc);
        const uint4 w8 = *reinterpret_cast<const uint4*>(reinterpret_cast<const uint16_t*>(wrow) + kc);
        if (8 * mk >= p.dt_rank) a8 = make_uint4(0u, 0u, 0u, 0u);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a8), __builtin_bit_cast(bf16x8_t, w8),
                                                      acc, 0, 0, 0);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] = bf16_to_f32(f32_to_bf16(acc[e]));
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) dv[tile][e] = acc[e];
    }
  };
  auto load_state = [&](int ts) {                            // the forward's checkpoint at step ts - 1; zero at 0
    float4 h4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ts > 0 && ts < L) h4 = *reinterpret_cast<const float4*>(ckl + static_cast<size_t>(ts / kBsSeg - 1) * D * kMaxState);
    hsn[0] = h4.x; hsn[1] = h4.y; hsn[2] = h4.z; hsn[3] = h4.w;
  };
  auto issue_loads = [&](int t0, int l) {
    const int qv = l & 7;
    const bool in = t0 + 4 * qv < L;
    const unsigned to = in ? t0 * kEsz : 0u - 4u * qv * kEsz;    // beyond the sequence: the row's first pack
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const unsigned ro = row_off(l, j) + to;
      if (!kDt) bs_load4<T>(dg, ro, dv[j]);
      bs_load4<T>(ug, ro, uv[j]);
      bs_load4<T>(gg, ro, gv[j]);
      if (kHasZ) bs_load4<T>(zg, ro + zdelta, zv[j]);
      if (!kDt) biasA[j] = p.delta_bias ? p.delta_bias[d0w + (l >> 3) + 8 * j] : 0.f;
    }
    if (kDt) form_delta(t0, l);
    load_state(t0 + kBsSeg);
  };

  const int nchunks = (L + kBsTC - 1) / kBsTC;
  {
    const int t0 = (nchunks - 1) * kBsTC;
    const int l = lane_now();
    issue_bc(t0, wave * 64 + l);
    issue_loads(t0, l);
    stage_bc(wave * 64 + l);
  }

  for (int c = nchunks - 1; c >= 0; --c) {
    const int t0 = c * kBsTC;
    __syncthreads();            // tBC of this chunk is complete; every wave is done with the previous chunk's tP

    // ---- phase A: per-element transcendentals in the load layout, into the tiles ----------------------------
    {
      const int l = lane_now();
      const bool in_seq = t0 + 4 * (l & 7) < L;
      if (kDt) {                                             // delta in the accumulator's layout (form_delta)
        const int mc = l & 15, mk = l >> 4;
#pragma unroll
        for (int tile = 0; tile < 2; ++tile) {
          const int qm = 4 * tile + mk;
          const float b2 = (t0 + 4 * qm < L) ? biasA[0] * kLog2e : -1e30f;
          float dl[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) dl[i] = bs_softplus_log2(fmaf(dv[tile][i], kLog2e, b2));
          *reinterpret_cast<float4*>(tD + bs_tile_off(mc, qm)) = make_float4(dl[0], dl[1], dl[2], dl[3]);
        }
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const float b2 = in_seq ? biasA[j] * kLog2e : -1e30f;  // padded pack: softplus -> 0, the identity step
        float dl[4], dyv[4], dzw[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (!kDt) dl[i] = bs_softplus_log2(fmaf(dv[j][i], kLog2e, b2));
          const float go = in_seq ? gv[j][i] : 0.f;
          if (kHasZ) {
            const float z = zv[j][i], sg = sigmoid_f(z);
            dyv[i] = go * z * sg;
            dzw[i] = go * sg * (1.f + z * (1.f - sg));
          } else {
            dyv[i] = go;
          }
        }
        const int to_ = bs_tile_off((l >> 3) + 8 * j, l & 7);
        if (!kDt) *reinterpret_cast<float4*>(tD + to_) = make_float4(dl[0], dl[1], dl[2], dl[3]);
        *reinterpret_cast<float4*>(tU + to_) = make_float4(uv[j][0], uv[j][1], uv[j][2], uv[j][3]);
        *reinterpret_cast<float4*>(tY + to_) = make_float4(dyv[0], dyv[1], dyv[2], dyv[3]);
        if (kHasZ) *reinterpret_cast<float4*>(tZ + to_) = make_float4(dzw[0], dzw[1], dzw[2], dzw[3]);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();

    // ---- phase B: two 16-step segments, right to left --------------------------------------------------------
#pragma unroll 1
    for (int seg = 1; seg >= 0; --seg) {
      const int ts = t0 + kBsSeg * seg;                      // first step of the segment
      if (ts >= L) { load_state(t0); continue; }             // whole segment beyond the sequence (wave-uniform)
      const float* bcp = tBC + seg * (kBsSeg * 32) + 4 * pq;
      const int gb = 4 * seg;

      // (1) recompute h_t (kept), y_t -> dz.  hsn: the state entering the segment (requested a pass or more ago).
      float H[kBsSeg][4];
      {
        float h[4] = {hsn[0], hsn[1], hsn[2], hsn[3]};
        float d4[4], u4[4], z4[4];
#ifndef SIMAMBA_BWDSEQ_NOPREFETCH
        float4 b4n = *reinterpret_cast<const float4*>(bcp);
        float4 c4n = *reinterpret_cast<const float4*>(bcp + 16);
#endif
#pragma unroll
        for (int i = 0; i < kBsSeg; ++i) {
          const int o = trow + 4 * ((gb + (i >> 2)) ^ sw);
          if ((i & 3) == 0) {
            const float4 dd = *reinterpret_cast<const float4*>(tD + o);
            const float4 uu = *reinterpret_cast<const float4*>(tU + o);
            d4[0] = dd.x; d4[1] = dd.y; d4[2] = dd.z; d4[3] = dd.w;
            u4[0] = uu.x; u4[1] = uu.y; u4[2] = uu.z; u4[3] = uu.w;
            if (kHasZ) {
              const float4 zz = *reinterpret_cast<const float4*>(tZ + o);
              z4[0] = zz.x; z4[1] = zz.y; z4[2] = zz.z; z4[3] = zz.w;
            }
          }
#ifndef SIMAMBA_BWDSEQ_NOPREFETCH
          // LDS reads run one step (B_t | C_t) ahead of their use; the sched_barriers keep hipcc from sinking them
          // back next to the consumer or hoisting a whole pass of them
          const float vB[4] = {b4n.x, b4n.y, b4n.z, b4n.w};
          const float vC[4] = {c4n.x, c4n.y, c4n.z, c4n.w};
          {
            const int in = (i + 1) & (kBsSeg - 1);           // wraps on the last step: harmless re-read
            b4n = *reinterpret_cast<const float4*>(bcp + in * 32);
            c4n = *reinterpret_cast<const float4*>(bcp + in * 32 + 16);
          }
          __builtin_amdgcn_sched_barrier(0);
#else
          const float4 b4 = *reinterpret_cast<const float4*>(bcp + i * 32);
          const float4 c4 = *reinterpret_cast<const float4*>(bcp + i * 32 + 16);
          const float vB[4] = {b4.x, b4.y, b4.z, b4.w};
          const float vC[4] = {c4.x, c4.y, c4.z, c4.w};
#endif
          const float dl = d4[i & 3], uu = u4[i & 3];
          const float xx = dl * uu;
          float y = 0.f;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float a = fast_exp2(dl * A2[j]);
            h[j] = fmaf(a, h[j], xx * vB[j]);
            H[i][j] = h[j];
            y = fmaf(vC[j], h[j], y);
          }
          if (kHasZ) {
            bs_quad_sum1(y);
            const float dzv = z4[i & 3] * fmaf(Dfull, uu, y);  // dz = dout silu'(z) (C . h + D u)
            if (pq == 0) tZ[o + (i & 3)] = dzv;
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }

      // (2) the adjoint, right to left
      {
        float d4[4], u4[4], y4[4];
#ifndef SIMAMBA_BWDSEQ_NOPREFETCH
        float4 b4n = *reinterpret_cast<const float4*>(bcp + (kBsSeg - 1) * 32);
        float4 c4n = *reinterpret_cast<const float4*>(bcp + (kBsSeg - 1) * 32 + 16);
#endif
#pragma unroll
        for (int i = kBsSeg - 1; i >= 0; --i) {
          const int o = trow + 4 * ((gb + (i >> 2)) ^ sw);
          if ((i & 3) == 3) {
            const float4 dd = *reinterpret_cast<const float4*>(tD + o);
            const float4 uu = *reinterpret_cast<const float4*>(tU + o);
            const float4 yy = *reinterpret_cast<const float4*>(tY + o);
            d4[0] = dd.x; d4[1] = dd.y; d4[2] = dd.z; d4[3] = dd.w;
            u4[0] = uu.x; u4[1] = uu.y; u4[2] = uu.z; u4[3] = uu.w;
            y4[0] = yy.x; y4[1] = yy.y; y4[2] = yy.z; y4[3] = yy.w;
          }
          // the state entering the next (left) segment is requested once half of the h registers are free again:
          // the second half of this pass covers its latency, and nothing older is pending in the memory queue
          if (i == kBsSeg / 2 - 1) load_state(ts - kBsSeg);
#ifndef SIMAMBA_BWDSEQ_NOPREFETCH
          const float vB[4] = {b4n.x, b4n.y, b4n.z, b4n.w};
          const float vC[4] = {c4n.x, c4n.y, c4n.z, c4n.w};
          {
            const int in = (i + kBsSeg - 1) & (kBsSeg - 1);  // wraps on the last step: harmless re-read
            b4n = *reinterpret_cast<const float4*>(bcp + in * 32);
            c4n = *reinterpret_cast<const float4*>(bcp + in * 32 + 16);
          }
          __builtin_amdgcn_sched_barrier(0);
#else
          const float4 b4 = *reinterpret_cast<const float4*>(bcp + i * 32);
          const float4 c4 = *reinterpret_cast<const float4*>(bcp + i * 32 + 16);
          const float vB[4] = {b4.x, b4.y, b4.z, b4.w};
          const float vC[4] = {c4.x, c4.y, c4.z, c4.w};
#endif
          const float dl = d4[i & 3], uu = u4[i & 3], dy = y4[i & 3];
          const float xx = dl * uu;
          float dxs = 0.f, dda = 0.f;
          float red[8];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float g = fmaf(vC[j], dy, ga[j]);          // g_t = a_{t+1} g_{t+1} + C_t dy_t
            const float a = fast_exp2(dl * A2[j]);
            ga[j] = g * a;
            // g_t a_t h_{t-1}; at the segment's first step a_t h_{t-1} = h_t - x_t B_t (the recurrence itself), so the
            // state that entered the segment need not stay in registers through both passes
            const float qv = (i > 0) ? ga[j] * H[i > 0 ? i - 1 : 0][j] : g * fmaf(-xx, vB[j], H[0][j]);
            dda = fmaf(A2[j], qv, dda);
            dAacc[j] = fmaf(dl, qv, dAacc[j]);
            dxs = fmaf(g, vB[j], dxs);
            red[j] = g * xx;                                 // dB_t[n] term
            red[4 + j] = dy * H[i][j];                       // dC_t[n] term
          }
          bs_quad_sum2(dxs, dda);
          const float duv = fmaf(dl, dxs, Dfull * dy);
          const float t2v = fmaf(uu, dxs, kLn2 * dda);
          dDacc = fmaf(dy, uu, dDacc);
          if (pq == 0) {
            tU[o + (i & 3)] = duv;
            tY[o + (i & 3)] = t2v;
          }
          // channel reduction of the 8 products over the wave's 16 channels: halves, row pairs, channel bits 1 and 0
#ifdef SIMAMBA_BWDSEQ_NORED
          const float s = red[0] + red[1] + red[2] + red[3] + red[4] + red[5] + red[6] + red[7];
#else
          swap32_stage<4>(red, red + 4);                     // lanes 0-31: dB terms, 32-63: dC terms
          swap16_stage<2>(red, red + 2);                     // rows 0 / 2: states 0,1 of the group; 1 / 3: states 2,3
          const float s = bs_row_finish(red[0], red[1]);
#endif
          if ((cq & 1) == 1) tP[(kBsSeg * seg + i) * kBsPPitch + pf] = s;
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();

    // ---- phase C, own tiles: finish ddelta, store the three gradients in load order --------------------------
    {
      const int l = lane_now();
      const bool in_seq = t0 + 4 * (l & 7) < L;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int to_ = bs_tile_off((l >> 3) + 8 * j, l & 7);
        const float4 a4 = *reinterpret_cast<const float4*>(tU + to_);
        const float4 t4 = *reinterpret_cast<const float4*>(tY + to_);
        const float4 d4 = *reinterpret_cast<const float4*>(tD + to_);
        const float duv[4] = {a4.x, a4.y, a4.z, a4.w};
        const float t2[4] = {t4.x, t4.y, t4.z, t4.w};
        const float dl[4] = {d4.x, d4.y, d4.z, d4.w};
        float dd[4], dsum = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          // d softplus(x)/dx = sigmoid(x) = 1 - exp(-softplus(x)); padded steps (delta = 0) carry no gradient
          dd[i] = t2[i] * (1.f - fast_exp2(-dl[i] * kLog2e));
          dsum += dd[i];
        }
        const unsigned ro = row_off(l, j) + t0 * kEsz;
        if (in_seq) {
          bs_store4<T>(dug, ro, duv);
          bs_store4<T>(ddg, ro, dd);
          if (kHasZ) {
            const float4 z4 = *reinterpret_cast<const float4*>(tZ + to_);
            const float dzv[4] = {z4.x, z4.y, z4.z, z4.w};
            bs_store4<T>(dzg, ro + dzdelta, dzv);
          }
        }
        // ddelta_bias: this chunk's sum over the row's 8 packs, one atomic per row and chunk
        if (p.ddelta_bias) {
          dsum += dpp<DPP_QUAD_XOR1>(0.f, dsum);
          dsum += dpp<DPP_QUAD_XOR2>(0.f, dsum);
          dsum += dpp<DPP_ROW_HALF_MIRROR>(0.f, dsum);
          if ((l & 7) == 0) atomicAdd(&p.ddelta_bias[d0w + (l >> 3) + 8 * j], dsum);
        }
      }
      // the next chunk's operands: requested here, where phase B's registers are free, behind this chunk's stores in the
      // memory queue and AHEAD of the dB | dC atomics (a load issued behind those would wait for them: vmcnt retires
      // in order); they land during the flush below and under the other two waves of this SIMD
      // (unconditional: on the last iteration chunk 0 is simply read again -- a load under `if (c > 0)` would keep the
      // PREVIOUS values of these 44 registers alive through all of phase B, which is exactly where none are free)
      const int tn = c > 0 ? t0 - kBsTC : 0;
      issue_loads(tn, l);
      issue_bc(tn, wave * 64 + l);
    }
    __syncthreads();            // every wave's tP is complete and nobody reads this chunk's tBC any more

    // ---- phase C, workgroup: stage the next B | C tile, flush dB | dC ------------------------------------------
    {
      const int tv = wave * 64 + lane_now();
      stage_bc(tv);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int e = tv + kBsThreads * k;
        const int f = e >> 5, t = e & 31;
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < kBsWaves; ++w) v += smem[w * kBsWaveFloats + 4 * kBsTile + t * kBsPPitch + f];
#ifdef SIMAMBA_BWDSEQ_NOFLUSH
        if (t0 + t < L && v == 123.456f) {
#else
        if (t0 + t < L) {
#endif
          float* dst = ((f >> 4) ? p.dC : p.dB) + (static_cast<size_t>(b) * kMaxState + (f & 15)) * L + t0 + t;
          atomicAdd(dst, v);
        }
      }
    }
  }

  // ---- sums over time: one atomic per (channel, state) / channel and sample ---------------------------------
#pragma unroll
  for (int j = 0; j < 4; ++j) atomicAdd(&p.dA[static_cast<size_t>(dch) * kMaxState + 4 * pq + j], dAacc[j]);
  if (p.dD && pq == 0) atomicAdd(&p.dD[dch], dDacc);
}

// What the sequential backward assumes beyond what simamba_selective_scan_bwd has already checked: 16 states,
// softplus on, whole workgroups of 64 channels, pack-aligned rows and B / C, 32-bit byte offsets.
bool scan_bwd_seq_ok(int batch, int dim, int seqlen, int dstate, int softplus, int vec, long long z_bs, long long dz_bs,
                     bool has_z, int bc_mode, long long bc_ns, long long bc_ts) {
  const long long rows = static_cast<long long>(batch) * dim;
  return dstate == kMaxState && softplus && vec && dim % kBsCh == 0 && rows * seqlen < (1ll << 30) && bc_mode != 0 &&
         (!has_z || (static_cast<long long>(batch) * z_bs < (1ll << 30) &&
                     static_cast<long long>(batch) * dz_bs < (1ll << 30))) &&
         bc_ns >= 0 && bc_ts >= 0 && (kMaxState - 1) * bc_ns + (seqlen - 1) * bc_ts < (1ll << 30);
}

int scan_bwd_seq_dispatch(const ScanArgs& sa, int io_dtype, int bc_mode, hipStream_t s, const void* dt, const void* wdt,
                          long long dt_bs, long long dt_ts, int dt_rank) {
  BwdSeqArgs a{};
  a.u = sa.u; a.delta = sa.delta; a.z = sa.z; a.dout = sa.dout;
  a.du = sa.du; a.ddelta = sa.ddelta; a.dz = sa.dz;
  a.A = sa.A; a.D = sa.D; a.delta_bias = sa.delta_bias; a.B = sa.B; a.C = sa.C;
  a.ckpt = sa.x_ckpt;
  a.dA = sa.dA; a.dB = sa.dB; a.dC = sa.dC; a.dD = sa.dD; a.ddelta_bias = sa.ddelta_bias;
  a.batch = sa.batch; a.dim = sa.dim; a.seqlen = sa.seqlen; a.nck = (sa.seqlen + kBsSeg - 1) / kBsSeg;
  a.bc_mode = bc_mode;
  a.z_bs = sa.z_bs; a.dz_bs = sa.dz_bs; a.bc_bs = sa.bc_bs; a.bc_ns = sa.bc_ns; a.bc_ts = sa.bc_ts;
  a.dt = dt; a.wdt = wdt; a.dt_bs = dt_bs; a.dt_ts = dt_ts; a.dt_rank = dt_rank;
  dim3 grid(a.dim / kBsCh, a.batch);
  if (a.dt) {                                                // the mixer's form: gated, delta formed in the kernel
    if (io_dtype == SIMAMBA_F32) hipLaunchKernelGGL((scan_bwd_seq_kernel<float, true, true>), grid, dim3(kBsThreads), 0, s, a);
    else hipLaunchKernelGGL((scan_bwd_seq_kernel<bf16_t, true, true>), grid, dim3(kBsThreads), 0, s, a);
    return static_cast<int>(hipGetLastError());
  }
  if (io_dtype == SIMAMBA_F32) {
    if (a.z) hipLaunchKernelGGL((scan_bwd_seq_kernel<float, true>), grid, dim3(kBsThreads), 0, s, a);
    else hipLaunchKernelGGL((scan_bwd_seq_kernel<float, false>), grid, dim3(kBsThreads), 0, s, a);
  } else {
    if (a.z) hipLaunchKernelGGL((scan_bwd_seq_kernel<bf16_t, true>), grid, dim3(kBsThreads), 0, s, a);
    else hipLaunchKernelGGL((scan_bwd_seq_kernel<bf16_t, false>), grid, dim3(kBsThreads), 0, s, a);
  }
  return static_cast<int>(hipGetLastError());
}

}  // namespace simamba
