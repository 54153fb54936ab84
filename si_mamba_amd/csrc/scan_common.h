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
  int vec;        // 16-byte vector access allowed
  int softplus;
};

// inclusive scan over the 16 lanes of a row of the affine map (P,S) : h -> P*h + S,
// composed left to right (lane 0 first).
__device__ __forceinline__ void row_scan_inclusive(float& P, float& S) {
#define SIMAMBA_SCAN_STEP(N)                                   \
  {                                                            \
    float Pp = dpp<DPP_ROW_SHR + N>(1.f, P);                   \
    float Sp = dpp<DPP_ROW_SHR + N>(0.f, S);                   \
    S = fmaf(P, Sp, S);                                        \
    P = P * Pp;                                                \
  }
  SIMAMBA_SCAN_STEP(1)
  SIMAMBA_SCAN_STEP(2)
  SIMAMBA_SCAN_STEP(4)
  SIMAMBA_SCAN_STEP(8)
#undef SIMAMBA_SCAN_STEP
}

// same, composed right to left (lane 15 first): g_out = Q * g_in + G with g flowing to lower lanes
__device__ __forceinline__ void row_scan_inclusive_rev(float& Q, float& G) {
#define SIMAMBA_SCAN_STEP(N)                                   \
  {                                                            \
    float Qn = dpp<DPP_ROW_SHL + N>(1.f, Q);                   \
    float Gn = dpp<DPP_ROW_SHL + N>(0.f, G);                   \
    G = fmaf(Q, Gn, G);                                        \
    Q = Q * Qn;                                                \
  }
  SIMAMBA_SCAN_STEP(1)
  SIMAMBA_SCAN_STEP(2)
  SIMAMBA_SCAN_STEP(4)
  SIMAMBA_SCAN_STEP(8)
#undef SIMAMBA_SCAN_STEP
}

// stage the chunk's B and C tiles (dstate x LC) of one batch sample into LDS, zero padded
template <typename T, int LC>
__device__ __forceinline__ void stage_bc(const T* __restrict__ Bg, const T* __restrict__ Cg, float* sB,
                                         float* sC, int b, int dstate, int L, int chunk) {
  constexpr int LDP = LC + 4;
  const int total = dstate * LC;
  const size_t base = static_cast<size_t>(b) * dstate * L;
  for (int i = threadIdx.x; i < total; i += kScanThreads) {
    int n = i / LC, t = i - n * LC;
    int gt = chunk * LC + t;
    float vb = 0.f, vc = 0.f;
    if (gt < L) {
      vb = to_f32<T>(Bg[base + static_cast<size_t>(n) * L + gt]);
      vc = to_f32<T>(Cg[base + static_cast<size_t>(n) * L + gt]);
    }
    sB[n * LDP + t] = vb;
    sC[n * LDP + t] = vc;
  }
}

}  // namespace simamba
