// Shared device helpers for the gfx950 kernels (wave64, DPP row ops, bf16 I/O).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/simamba.h"

namespace simamba {

constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;

struct bf16_t { uint16_t v; };

__device__ __forceinline__ float bf16_to_f32(uint16_t h) {
  return __builtin_bit_cast(float, static_cast<uint32_t>(h) << 16);
}
__device__ __forceinline__ uint16_t f32_to_bf16(float f) {
  // plain cast path: hipcc emits v_cvt_pk_bf16_f32 (RNE, NaN stays NaN) on gfx950
  __bf16 b = static_cast<__bf16>(f);
  return __builtin_bit_cast(uint16_t, b);
}

template <typename T> __device__ __forceinline__ float to_f32(T x);
template <> __device__ __forceinline__ float to_f32<float>(float x) { return x; }
template <> __device__ __forceinline__ float to_f32<bf16_t>(bf16_t x) { return bf16_to_f32(x.v); }
template <typename T> __device__ __forceinline__ T from_f32(float x);
template <> __device__ __forceinline__ float from_f32<float>(float x) { return x; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float x) { return bf16_t{f32_to_bf16(x)}; }

// ---- K consecutive elements per lane, 16-byte vector access when aligned -------------------
template <typename T, int K> struct alignas(sizeof(T) * K) Pack { T v[K]; };

template <typename T, int K>
__device__ __forceinline__ void load_items(const T* __restrict__ p, int nvalid, bool vec, float (&out)[K]) {
  if (vec && nvalid >= K) {
    constexpr int kChunk = (sizeof(T) * K >= 16) ? 16 / sizeof(T) : K;  // elements per 16-B load
#pragma unroll
    for (int c = 0; c < K; c += kChunk) {
      Pack<T, kChunk> pk = *reinterpret_cast<const Pack<T, kChunk>*>(p + c);
#pragma unroll
      for (int j = 0; j < kChunk; ++j) out[c + j] = to_f32<T>(pk.v[j]);
    }
  } else {
#pragma unroll
    for (int j = 0; j < K; ++j) out[j] = (j < nvalid) ? to_f32<T>(p[j]) : 0.f;
  }
}

template <typename T, int K>
__device__ __forceinline__ void store_items(T* __restrict__ p, int nvalid, bool vec, const float (&in)[K]) {
  if (vec && nvalid >= K) {
    constexpr int kChunk = (sizeof(T) * K >= 16) ? 16 / sizeof(T) : K;
#pragma unroll
    for (int c = 0; c < K; c += kChunk) {
      Pack<T, kChunk> pk;
#pragma unroll
      for (int j = 0; j < kChunk; ++j) pk.v[j] = from_f32<T>(in[c + j]);
      *reinterpret_cast<Pack<T, kChunk>*>(p + c) = pk;
    }
  } else {
#pragma unroll
    for (int j = 0; j < K; ++j)
      if (j < nvalid) p[j] = from_f32<T>(in[j]);
  }
}

// ---- DPP (data-parallel primitives) on 16-lane rows ----------------------------------------
// update_dpp(old, src, ctrl, row_mask, bank_mask, bound_ctrl): lanes whose source lane is out
// of range (bound_ctrl = 0) or that are masked off keep `old`.
template <int CTRL, int ROW_MASK = 0xf, int BANK_MASK = 0xf>
__device__ __forceinline__ float dpp(float old, float src) {
  return __builtin_bit_cast(
      float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, src), CTRL,
                                         ROW_MASK, BANK_MASK, false));
}
constexpr int DPP_QUAD_XOR1 = 0xB1;         // quad_perm [1,0,3,2]
constexpr int DPP_QUAD_XOR2 = 0x4E;         // quad_perm [2,3,0,1]
constexpr int DPP_ROW_SHL = 0x100;          // + n (1..15): lane i reads lane i+n of its row
constexpr int DPP_ROW_SHR = 0x110;          // + n (1..15): lane i reads lane i-n of its row
constexpr int DPP_ROW_MIRROR = 0x140;       // lane i reads lane 15-i
constexpr int DPP_ROW_HALF_MIRROR = 0x141;  // lane i reads lane 7-i within its half row

// sum over the 16 lanes of a row, result in every lane
__device__ __forceinline__ float row_allreduce_sum(float v) {
  v += dpp<DPP_QUAD_XOR1>(0.f, v);
  v += dpp<DPP_QUAD_XOR2>(0.f, v);
  v += dpp<DPP_ROW_HALF_MIRROR>(0.f, v);
  v += dpp<DPP_ROW_MIRROR>(0.f, v);
  return v;
}

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float fast_log2(float x) { return __builtin_amdgcn_logf(x); }
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

// softplus with torch's threshold (x > 20 -> x); tiny-x branch keeps log1p accuracy
__device__ __forceinline__ float softplus_f(float x) {
  float e = fast_exp2(x * kLog2e);
  float sp = fast_log2(1.f + e) * kLn2;
  sp = (x < -15.f) ? e : sp;
  return (x > 20.f) ? x : sp;
}
__device__ __forceinline__ float sigmoid_f(float x) { return fast_rcp(1.f + fast_exp2(-x * kLog2e)); }

}  // namespace simamba
