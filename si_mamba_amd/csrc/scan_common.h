// Shared pieces of the selective-scan forward and backward kernels.
//
// Work decomposition (both directions)
//   * one 16-lane DPP row  <->  one (batch, channel) recurrence over a chunk of
//     LC = 16 * kItems timesteps; each lane owns kItems consecutive timesteps, so a row reads
//     and writes whole contiguous segments of the (B, D, L) tensors (coalesced, no transposition).
//   * a wave64 carries 4 such rows (4 channels), a 256-thread workgroup 16, and the workgroup
//     walks `R` passes of 16 channels that all belong to ONE batch sample, so the (B_t, C_t)
//     tile of the chunk is staged in LDS once and broadcast-read by all of them.
//   * inside a lane the recurrence is sequential; across the 16 lanes it is a 4-step DPP
//     (row_shr) scan on the affine pair (P, S):  h_out = P * h_in + S.
//   * sequences longer than LC are walked chunk by chunk with the running state in LDS.
#pragma once
#include "common.h"

namespace simamba {

constexpr int kScanThreads = 256;
constexpr int kRowsPerPass = kScanThreads / 16;  // 16 channels per workgroup pass
constexpr int kMaxState = 16;
constexpr int kNumXcd = 8;    // MI355X: 8 XCDs, each with its own 4 MB L2; workgroups are dealt to them round-robin

// XCD-aware (tile, sample) of a workgroup in a (tiles, batch) grid.  The hardware sends consecutive workgroup
// ids to consecutive XCDs, so with the plain blockIdx mapping the tiles of ONE sample -- which all read that
// sample's B_t / C_t rows (and, in the backward, add into the same dB / dC rows) -- land on 8 different L2s.
// Re-labelling id -> (id % 8) * (total / 8) + id / 8 makes the ids that one XCD receives a contiguous range of
// (sample, tile) pairs: a sample's tiles share one L2.  Falls back to the identity when the grid does not
// split evenly over the XCDs.
__device__ __forceinline__ void xcd_tile(int& tile, int& sample) {
  const int gx = gridDim.x;
  const int total = gx * gridDim.y;
  int id = blockIdx.y * gx + blockIdx.x;
  if ((total % kNumXcd) == 0) id = (id % kNumXcd) * (total / kNumXcd) + id / kNumXcd;
  sample = id / gx;
  tile = id - sample * gx;
}

struct ScanArgs {
  const void* u;
  const void* delta;
  const float* A;
  const void* B;
  const void* C;
  const float* D;
  const void* z;
  const float* delta_bias;
  void* out;
  float* x_ckpt;
  float* last_state;
  // backward only
  const void* dout;
  void* du;
  void* ddelta;
  float* dA;
  float* dB;
  float* dC;
  float* dD;
  void* dz;
  float* ddelta_bias;
  int batch, dim, seqlen, dstate;
  int nchunks;
  int passes;     // R: channel passes per workgroup (tile = 16 * R channels)
  int long_items; // forward: 16 steps per lane for L > 128
  int vec;        // 16-byte vector access allowed
  int softplus;
  long long z_bs, dz_bs;            // batch strides (elements) of z / dz
  long long bc_bs, bc_ns, bc_ts;    // (batch, state, time) strides (elements) of B and C
};

// Inclusive scan over the 16 lanes of a row of the affine map (P,S) : h -> P*h + S, composed left to
// right (lane 0 first).  Each step is two fused DPP instructions: lanes whose source lane falls outside
// the row are simply not written (bound_ctrl:0), which is the identity of the composition -- no v_mov of
// identity values, no select.  Inline asm because hipcc does not fold update_dpp into the consumer here;
// the s_nop pads are the "VALU write -> DPP read of the same VGPR: 2 wait states" hazard.
__device__ __forceinline__ void row_scan_inclusive(float& P, float& S) {
  asm volatile(
      "s_nop 1\n\t"
      "v_fmac_f32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
      "v_mul_f32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 0\n\t"
      "v_fmac_f32_dpp %0, %0, %1 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
      "v_mul_f32_dpp %1, %1, %1 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 0\n\t"
      "v_fmac_f32_dpp %0, %0, %1 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
      "v_mul_f32_dpp %1, %1, %1 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 0\n\t"
      "v_fmac_f32_dpp %0, %0, %1 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
      "v_mul_f32_dpp %1, %1, %1 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1"
      : "+v"(S), "+v"(P));
}

// same, composed right to left (lane 15 first): g_out = Q * g_in + G with g flowing to lower lanes
__device__ __forceinline__ void row_scan_inclusive_rev(float& Q, float& G) {
  asm volatile(
      "s_nop 1\n\t"
      "v_fmac_f32_dpp %0, %0, %1 row_shl:1 row_mask:0xf bank_mask:0xf\n\t"
      "v_mul_f32_dpp %1, %1, %1 row_shl:1 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 0\n\t"
      "v_fmac_f32_dpp %0, %0, %1 row_shl:2 row_mask:0xf bank_mask:0xf\n\t"
      "v_mul_f32_dpp %1, %1, %1 row_shl:2 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 0\n\t"
      "v_fmac_f32_dpp %0, %0, %1 row_shl:4 row_mask:0xf bank_mask:0xf\n\t"
      "v_mul_f32_dpp %1, %1, %1 row_shl:4 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 0\n\t"
      "v_fmac_f32_dpp %0, %0, %1 row_shl:8 row_mask:0xf bank_mask:0xf\n\t"
      "v_mul_f32_dpp %1, %1, %1 row_shl:8 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1"
      : "+v"(G), "+v"(Q));
}

// value of the previous (SHR) / next (SHL) lane of the row; `edge` where there is none
__device__ __forceinline__ float row_prev(float v, float edge) { return dpp<DPP_ROW_SHR + 1>(edge, v); }
__device__ __forceinline__ float row_next(float v, float edge) { return dpp<DPP_ROW_SHL + 1>(edge, v); }

// the 16 per-state coefficients A[d][0..N) of a row, pre-scaled by log2(e), as registers
__device__ __forceinline__ void load_A_row(const float* __restrict__ Arow, int N, float (&A2)[kMaxState]) {
  if (N == kMaxState) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 v = *reinterpret_cast<const float4*>(Arow + 4 * q);
      A2[4 * q] = v.x * kLog2e; A2[4 * q + 1] = v.y * kLog2e; A2[4 * q + 2] = v.z * kLog2e; A2[4 * q + 3] = v.w * kLog2e;
    }
  } else {
#pragma unroll
    for (int n = 0; n < kMaxState; ++n) A2[n] = (n < N) ? Arow[n] * kLog2e : 0.f;
  }
}

// LDS image of a (state, time) tile row: 16-byte quads, XOR-swizzled by the 64-byte-group index.  A lane reads
// the kItems/4 consecutive quads of its time segment, one per ds_read_b128; unswizzled, lanes 64 bytes x 4
// apart would share bank quads on every read (measured at kItems = 8: SQ_LDS_BANK_CONFLICT = 49 % of
// SQ_LDS_IDX_ACTIVE); with q ^ ((q >> 4) & (kItems/4 - 1)) the 16 lanes of a row cover 16 distinct quads.
template <int kItems>
__device__ __forceinline__ int bc_quad(int q) { return q ^ ((q >> 4) & (kItems / 4 - 1)); }

// stage the chunk's B and C tiles (dstate x LC) of one batch sample into LDS, zero padded.
// The global side is read through (state, time) strides: time-major (B,N,L) tensors are walked with
// consecutive lanes on consecutive t, token-major ones (x_proj output, state stride 1) on consecutive n.
template <typename T, int LC>
__device__ __forceinline__ void stage_bc(const T* __restrict__ Bg, const T* __restrict__ Cg, float* sB,
                                         float* sC, int b, int dstate, int L, int chunk, long long bs,
                                         long long ns, long long ts) {
  constexpr int LDP = LC + 4;
  const int total = dstate * LC;
  const long long base = static_cast<long long>(b) * bs;
  const bool token_major = (ns == 1);
  for (int i = threadIdx.x; i < total; i += kScanThreads) {
    int n, t;
    if (token_major) { t = i / dstate; n = i - t * dstate; }
    else { n = i / LC; t = i - n * LC; }
    const int gt = chunk * LC + t;
    float vb = 0.f, vc = 0.f;
    if (gt < L) {
      const long long o = base + n * ns + gt * ts;
      vb = to_f32<T>(Bg[o]);
      vc = to_f32<T>(Cg[o]);
    }
    const int slot = n * LDP + bc_quad<LC / 16>(t >> 2) * 4 + (t & 3);
    sB[slot] = vb;
    sC[slot] = vc;
  }
}

}  // namespace simamba
