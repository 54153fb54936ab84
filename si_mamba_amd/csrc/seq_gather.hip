// Token-sequence expansion of a (batch, channels, G) tensor along its last axis, and its adjoint.
//
// The reference feeds the block stack a sequence of L = 2 k G tokens that are the SAME G patch tokens in 2 k orders
// (models/point_mamba.py:889-898, :982-989).  Everything the first block does before its conv1d is per token
// (Add, LayerNorm, in_proj: models/block.py:56-60 and the first GEMM of the mixer), so it commutes with that
// gather: computed once on the G distinct tokens and expanded afterwards it costs 1/(2k) of the flops.  What is
// left of the 552 us in_proj GEMM is this kernel -- a pure copy, bound by the write of the (batch, channels, L) result:
//     out[b, c, l] = in[b, c, idx[b, l]]                     (forward, gather)
//     din[b, c, g] = sum_j dout[b, c, inv[b, g, j]]          (backward: every token sits at R = L / G positions)
// One wave per (b, channel row): the 512-byte source row (forward) / the 4 KB gradient row (backward) goes through a
// wave-private LDS buffer with 16-byte global accesses on both sides; the index of a sample is read once per
// workgroup pass.  The backward sums in a fixed order: deterministic, no atomics.
#include "common.h"

namespace simamba {

constexpr int kSgThreads = 256;
constexpr int kSgMaxG = 256;       // distinct tokens per sample
constexpr int kSgMaxL = 2048;      // sequence length

struct SgArgs {
  const void* in;      // fwd: (B, C, G)        bwd: dout (B, C, L) with batch stride
  void* out;           // fwd: (B, C, L) w/ bs   bwd: din (B, C, G)
  const int* idx;      // fwd: (B, L) token of every position      bwd: (B, G, R) positions of every token
  int B, C, G, L, R;
  long long seq_bs;    // batch stride (elements) of the (B, C, L) tensor
};

template <typename T>
__global__ __launch_bounds__(kSgThreads) void seq_gather_fwd_kernel(SgArgs p) {
  __shared__ __attribute__((aligned(16))) float sRow[kSgThreads / 64][kSgMaxG];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int b = blockIdx.y;
  const int G = p.G, L = p.L;
  // this lane's output packs: 4 consecutive positions each, pack q = lane + 64 j; their 4 source tokens
  constexpr int kMaxPacks = kSgMaxL / 4 / 64;
  const int npk = (L / 4 + 63) / 64;
  int tok[kMaxPacks][4];
#pragma unroll
  for (int j = 0; j < kMaxPacks; ++j) {
    const int q = lane + 64 * j;
    if (j < npk && 4 * q < L) {
      const int4 v = *reinterpret_cast<const int4*>(p.idx + static_cast<size_t>(b) * L + 4 * q);
      tok[j][0] = v.x; tok[j][1] = v.y; tok[j][2] = v.z; tok[j][3] = v.w;
    } else {
      tok[j][0] = tok[j][1] = tok[j][2] = tok[j][3] = 0;
    }
  }
  float* row = sRow[wave];
  const T* __restrict__ in = static_cast<const T*>(p.in);
  T* __restrict__ out = static_cast<T*>(p.out);
  for (int c = blockIdx.x * (kSgThreads / 64) + wave; c < p.C; c += gridDim.x * (kSgThreads / 64)) {
    const T* src = in + (static_cast<size_t>(b) * p.C + c) * G;
    for (int g4 = 4 * lane; g4 < G; g4 += 256) {
      const Pack<T, 4> pk = *reinterpret_cast<const Pack<T, 4>*>(src + g4);
      *reinterpret_cast<float4*>(row + g4) = make_float4(to_f32<T>(pk.v[0]), to_f32<T>(pk.v[1]), to_f32<T>(pk.v[2]),
                                                         to_f32<T>(pk.v[3]));
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    T* dst = out + static_cast<size_t>(b) * p.seq_bs + static_cast<size_t>(c) * L;
#pragma unroll
    for (int j = 0; j < kMaxPacks; ++j) {
      const int q = lane + 64 * j;
      if (j < npk && 4 * q < L) {
        Pack<T, 4> o;
#pragma unroll
        for (int i = 0; i < 4; ++i) o.v[i] = from_f32<T>(row[tok[j][i]]);
        *reinterpret_cast<Pack<T, 4>*>(dst + 4 * q) = o;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}

template <typename T>
__global__ __launch_bounds__(kSgThreads) void seq_gather_bwd_kernel(SgArgs p) {
  __shared__ __attribute__((aligned(16))) float sRow[kSgThreads / 64][kSgMaxL];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int b = blockIdx.y;
  const int G = p.G, L = p.L, R = p.R;
  float* row = sRow[wave];
  const T* __restrict__ dout = static_cast<const T*>(p.in);
  T* __restrict__ din = static_cast<T*>(p.out);
  // the positions of this lane's tokens (g = lane + 64 k), read once per workgroup: R <= 8 of them per token
  constexpr int kGPL = kSgMaxG / 64, kMaxR = 8;
  int pos[kGPL][kMaxR];
#pragma unroll
  for (int k = 0; k < kGPL; ++k) {
    const int g = lane + 64 * k;
#pragma unroll
    for (int j = 0; j < kMaxR; ++j)
      pos[k][j] = (g < G && j < R) ? p.idx[(static_cast<size_t>(b) * G + g) * R + j] : -1;
  }
  for (int c = blockIdx.x * (kSgThreads / 64) + wave; c < p.C; c += gridDim.x * (kSgThreads / 64)) {
    const T* src = dout + static_cast<size_t>(b) * p.seq_bs + static_cast<size_t>(c) * L;
    for (int l4 = 4 * lane; l4 < L; l4 += 256) {
      const Pack<T, 4> pk = *reinterpret_cast<const Pack<T, 4>*>(src + l4);
      *reinterpret_cast<float4*>(row + l4) = make_float4(to_f32<T>(pk.v[0]), to_f32<T>(pk.v[1]), to_f32<T>(pk.v[2]),
                                                         to_f32<T>(pk.v[3]));
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    T* dst = din + (static_cast<size_t>(b) * p.C + c) * G;
#pragma unroll
    for (int k = 0; k < kGPL; ++k) {
      const int g = lane + 64 * k;
      if (g < G) {
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < kMaxR; ++j) s += (pos[k][j] >= 0) ? row[pos[k][j]] : 0.f;
        dst[g] = from_f32<T>(s);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}

}  // namespace simamba

using namespace simamba;

static int sg_check(int B, int C, int G, int L, int io_dtype) {
  if (B < 0 || C <= 0 || G <= 0 || L <= 0 || B > 65535) return SIMAMBA_E_SHAPE;
  if (io_dtype != SIMAMBA_F32 && io_dtype != SIMAMBA_BF16) return SIMAMBA_E_DTYPE;
  if (G > kSgMaxG || L > kSgMaxL || G % 4 || L % 4) return SIMAMBA_E_SHAPE;
  return SIMAMBA_OK;
}

extern "C" int simamba_seq_gather_fwd(const void* in, const int* idx, void* out, int batch, int channels, int G, int L,
                                      long long out_bstride, int io_dtype, void* stream) {
  if (int rc = sg_check(batch, channels, G, L, io_dtype)) return rc;
  if (batch == 0) return SIMAMBA_OK;
  if (!in || !idx || !out) return SIMAMBA_E_NULLPTR;
  SgArgs a{};
  a.in = in; a.out = out; a.idx = idx; a.B = batch; a.C = channels; a.G = G; a.L = L; a.R = 0;
  a.seq_bs = out_bstride ? out_bstride : static_cast<long long>(channels) * L;
  const size_t esz = io_dtype == SIMAMBA_F32 ? 4 : 2;
  if ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) & (4 * esz - 1) ||
      reinterpret_cast<uintptr_t>(idx) & 15u || a.seq_bs % 4)
    return SIMAMBA_E_ALIGN;
  // >= 2048 workgroups when the work allows: a workgroup walks channel rows of ONE sample (its index stays in registers)
  int gx = (channels + 3) / 4;
  while (gx > 1 && static_cast<long long>(gx) * batch > 4096) gx = (gx + 1) / 2;
  dim3 grid(gx, batch);
  if (io_dtype == SIMAMBA_F32)
    hipLaunchKernelGGL(seq_gather_fwd_kernel<float>, grid, dim3(kSgThreads), 0, static_cast<hipStream_t>(stream), a);
  else
    hipLaunchKernelGGL(seq_gather_fwd_kernel<bf16_t>, grid, dim3(kSgThreads), 0, static_cast<hipStream_t>(stream), a);
  return static_cast<int>(hipGetLastError());
}

extern "C" int simamba_seq_gather_bwd(const void* dout, const int* inv, void* din, int batch, int channels, int G, int L,
                                      int R, long long dout_bstride, int io_dtype, void* stream) {
  if (int rc = sg_check(batch, channels, G, L, io_dtype)) return rc;
  if (R <= 0 || R > 8 || static_cast<long long>(G) * R != L) return SIMAMBA_E_SHAPE;
  if (batch == 0) return SIMAMBA_OK;
  if (!dout || !inv || !din) return SIMAMBA_E_NULLPTR;
  SgArgs a{};
  a.in = dout; a.out = din; a.idx = inv; a.B = batch; a.C = channels; a.G = G; a.L = L; a.R = R;
  a.seq_bs = dout_bstride ? dout_bstride : static_cast<long long>(channels) * L;
  const size_t esz = io_dtype == SIMAMBA_F32 ? 4 : 2;
  if (reinterpret_cast<uintptr_t>(dout) & (4 * esz - 1) || a.seq_bs % 4) return SIMAMBA_E_ALIGN;
  int gx = (channels + 3) / 4;
  while (gx > 1 && static_cast<long long>(gx) * batch > 4096) gx = (gx + 1) / 2;
  dim3 grid(gx, batch);
  if (io_dtype == SIMAMBA_F32)
    hipLaunchKernelGGL(seq_gather_bwd_kernel<float>, grid, dim3(kSgThreads), 0, static_cast<hipStream_t>(stream), a);
  else
    hipLaunchKernelGGL(seq_gather_bwd_kernel<bf16_t>, grid, dim3(kSgThreads), 0, static_cast<hipStream_t>(stream), a);
  return static_cast<int>(hipGetLastError());
}
