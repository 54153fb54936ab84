// Cross-lane reduction pieces shared by the two selective-scan backward kernels (gfx950, wave64).
#pragma once
#include "common.h"

namespace simamba {

// x[i] <- (x[i] + y[i]) after exchanging half of the lanes: lanes 0-31 end with the sum over halves of x,
// lanes 32-63 with that of y (swap32); rows 0/2 with x summed over row pairs, rows 1/3 with y (swap16).
// Inline asm on purpose: with ROCm 7.2's hipcc the two-result builtin
// (__builtin_amdgcn_permlane{16,32}_swap) followed by r[0] + r[1] is register-coalesced into
// "v_add v, v, v" (2 * r[0]); verified in the .s and on hardware (tools/permlane_probe.hip).
// The s_nop pads cover the VALU-write -> permlane-swap read hazard, which hipcc does not see in asm.
// One reduction stage for N register pairs: all N swaps go out back to back inside ONE asm block (they touch
// disjoint registers, so only the block's inputs and outputs need the hazard padding), then N adds.
template <int N>
__device__ __forceinline__ void swap32_stage(float* x, float* y) {
  static_assert(N == 8 || N == 4, "pairs per stage");
  if constexpr (N == 8) {
    asm volatile(
        "s_nop 1\n\t"
        "v_permlane32_swap_b32 %0, %8\n\tv_permlane32_swap_b32 %1, %9\n\t"
        "v_permlane32_swap_b32 %2, %10\n\tv_permlane32_swap_b32 %3, %11\n\t"
        "v_permlane32_swap_b32 %4, %12\n\tv_permlane32_swap_b32 %5, %13\n\t"
        "v_permlane32_swap_b32 %6, %14\n\tv_permlane32_swap_b32 %7, %15\n\t"
        "s_nop 1"
        : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]),
          "+v"(y[0]), "+v"(y[1]), "+v"(y[2]), "+v"(y[3]), "+v"(y[4]), "+v"(y[5]), "+v"(y[6]), "+v"(y[7]));
  } else {
    asm volatile(
        "s_nop 1\n\t"
        "v_permlane32_swap_b32 %0, %4\n\tv_permlane32_swap_b32 %1, %5\n\t"
        "v_permlane32_swap_b32 %2, %6\n\tv_permlane32_swap_b32 %3, %7\n\t"
        "s_nop 1"
        : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(y[0]), "+v"(y[1]), "+v"(y[2]), "+v"(y[3]));
  }
#pragma unroll
  for (int i = 0; i < N; ++i) x[i] += y[i];
}
template <int N>
__device__ __forceinline__ void swap16_stage(float* x, float* y) {
  static_assert(N == 4 || N == 2, "pairs per stage");
  if constexpr (N == 4) {
    asm volatile(
        "s_nop 1\n\t"
        "v_permlane16_swap_b32 %0, %4\n\tv_permlane16_swap_b32 %1, %5\n\t"
        "v_permlane16_swap_b32 %2, %6\n\tv_permlane16_swap_b32 %3, %7\n\t"
        "s_nop 1"
        : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(y[0]), "+v"(y[1]), "+v"(y[2]), "+v"(y[3]));
  } else {
    asm volatile(
        "s_nop 1\n\t"
        "v_permlane16_swap_b32 %0, %2\n\tv_permlane16_swap_b32 %1, %3\n\t"
        "s_nop 1"
        : "+v"(x[0]), "+v"(x[1]), "+v"(y[0]), "+v"(y[1]));
  }
#pragma unroll
  for (int i = 0; i < N; ++i) x[i] += y[i];
}

}  // namespace simamba
