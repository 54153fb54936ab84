// Shared declarations of the spectral kernels (spectral.hip, spectral_tridiag.hip).
#pragma once
#include "common.h"

namespace simamba {

constexpr int kSpecMaxG = 128;
constexpr int kTdMaxSel = 8;      // eigenpairs the tridiagonal path extracts at most (k, +1 for MATRIX_SYM)

struct EigArgs {
  const float* adj;
  float* evals;
  float* evecs;
  long long* order;
  float* all_evals;
  float* all_evecs;
  int B, G, k;
  unsigned flags;
};

int launch_tridiag_topk(const EigArgs& a, hipStream_t s);

}  // namespace simamba
