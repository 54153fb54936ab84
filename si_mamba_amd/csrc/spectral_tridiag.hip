// Top-k eigenpairs of the graph Laplacian: Householder tridiagonalisation + bisection + inverse iteration.
//
// Same contract as laplacian_eig_kernel (spectral.hip) for the outputs the reference's forward consumes
// (models/point_mamba.py:884 uses only the k selected pairs): evals (B,k), evecs (B,G,k), order (B,k,G).
// The Jacobi kernel diagonalises the whole matrix (8-9 sweeps x 127 barrier-separated steps, 3.6 ms per
// workgroup); only k <= 7 pairs are needed, so this kernel does what LAPACK's ?syevx does, one workgroup
// (512 lanes) per sample, everything in LDS:
//   1. S = mirrored lower triangle of I - D^-1 A (the reference's eigh quirk), fp32, 66 KiB;
//   2. Householder reduction to tridiagonal T (G-2 reflectors; per reflector one matvec and one rank-2 update
//      of the trailing block, 16-lane DPP rows own matrix rows, reflectors stay in the zeroed columns;
//      4 barriers per reflector);
//   3. the wanted eigenvalues of T by multisection on Sturm counts in fp64 (32 shifts per eigenvalue and
//      round, 8 rounds -> interval 33^-8 of the Gershgorin range);
//   4. eigenvectors of T by inverse iteration with a pivoted tridiagonal LU in fp64 (one lane per vector),
//      modified Gram-Schmidt among them, back-transformation through the reflectors (one wave per vector);
//   5. sign convention, normalisation, rank-sort argsort -> order.
// Compiled with -ffp-contract=off (the Laplacian entries must round like the reference's torch ops).
#include "spectral_common.h"

namespace simamba {

constexpr int kTdThreads = 512;
constexpr int kTdGroups = kTdThreads / 16;   // 16-lane DPP rows
constexpr int kTdLD = kSpecMaxG + 1;      // odd pitch: row- and column-wise walks are both conflict-free

__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

// per-group partial sums of the squares of column `col` below the diagonal (rows col+1 .. G-1) -> sPart[grp]
__device__ __forceinline__ void colnorm_partials(const float* S, int LD, int G, int col, float* sPart) {
  const int lane16 = threadIdx.x & 15, grp = threadIdx.x >> 4;
  if (lane16 == 0) {
    float acc = 0.f;
    for (int r = col + 1 + grp; r < G; r += kTdGroups) {
      const float x = S[r * LD + col];
      acc = fmaf(x, x, acc);
    }
    sPart[grp] = acc;
  }
}

// sum of the kTdGroups (= 32) group partials, read as 8 broadcast float4
__device__ __forceinline__ float sum_partials(const float* sPart) {
  float acc = 0.f;
#pragma unroll
  for (int q = 0; q < kTdGroups / 4; ++q) {
    const float4 v = *reinterpret_cast<const float4*>(sPart + 4 * q);
    acc += (v.x + v.y) + (v.z + v.w);
  }
  return acc;
}

// 1/x to ~1 ulp: hardware v_rcp_f64 seed + one Newton step (the IEEE division sequence is ~30 dependent
// instructions; the Sturm recurrences below are pure latency chains of divisions)
__device__ __forceinline__ double fast_rcp_f64(double x) {
  double r = __builtin_amdgcn_rcp(x);
  return fma(fma(-x, r, 1.0), r, r);
}

// number of eigenvalues of the tridiagonal (d, e) below sigma
__device__ __forceinline__ int sturm_count(const double* d, const double* e2, int n, double sigma, double pivmin) {
  double q = d[0] - sigma;
  if (fabs(q) < pivmin) q = -pivmin;
  int cnt = q < 0.0;
  for (int i = 1; i < n; ++i) {
    q = d[i] - sigma - e2[i - 1] * fast_rcp_f64(q);
    if (fabs(q) < pivmin) q = -pivmin;
    cnt += q < 0.0;
  }
  return cnt;
}

__global__ __launch_bounds__(kTdThreads) void laplacian_tridiag_kernel(EigArgs p) {
  extern __shared__ __attribute__((aligned(16))) float S[];     // [G][kTdLD]
  __shared__ float sDeg[kSpecMaxG];
  __shared__ float sV[kSpecMaxG];           // current reflector
  __shared__ float sW[kSpecMaxG];           // p = tau S22 v
  __shared__ float sTau[kSpecMaxG];
  __shared__ double sD[kSpecMaxG], sE[kSpecMaxG], sE2[kSpecMaxG];
  __shared__ __attribute__((aligned(16))) float sPart[kTdGroups], sPd[kTdGroups];
  __shared__ double sLam[kTdMaxSel];
  __shared__ double sZ[kTdMaxSel][kSpecMaxG];
  __shared__ double sLa[kTdMaxSel][kSpecMaxG], sLb[kTdMaxSel][kSpecMaxG], sLc[kTdMaxSel][kSpecMaxG],
      sLl[kTdMaxSel][kSpecMaxG];
  __shared__ unsigned char sPiv[kTdMaxSel][kSpecMaxG];
  __shared__ float sSign[kTdMaxSel];

  const int G = p.G;
  constexpr int LD = kTdLD;
  const int tid = threadIdx.x;
  const int lane16 = tid & 15, grp = tid >> 4;      // 16 groups of 16 lanes
  const float* A = p.adj + static_cast<size_t>(blockIdx.x) * G * G;
  const bool msym = p.flags & SIMAMBA_SPEC_MATRIX_SYM;
  const bool smallest = p.flags & SIMAMBA_SPEC_SMALLEST;
  const int skip = msym ? 1 : 0;
  const int nsel = p.k;
  const int ntot = nsel + skip;                     // eigenpairs actually extracted (<= kTdMaxSel)

  // ---- 1. Laplacian ------------------------------------------------------------------------------------
  if (tid < G) {
    float s = 0.f;
    for (int j = 0; j < G; ++j) s = s + (A[tid * G + j] + A[j * G + tid]) / 2.f;
    sDeg[tid] = s;
  }
  __syncthreads();
  for (int e = tid; e < G * G; e += kTdThreads) {
    const int i = e / G, j = e - i * G;
    if (i >= j) {   // eigh(UPLO='L'): only the lower triangle of the (unsymmetric) L is read
      const float aij = (A[i * G + j] + A[j * G + i]) / 2.f;
      float l;
      if (msym) {
        const float di = powf(sDeg[i], -0.5f), dj = powf(sDeg[j], -0.5f);
        l = (i == j ? 1.f : 0.f) - (di * aij) * dj;
      } else {
        const float dinv = 1.0f / (sDeg[i] + 1e-6f);
        l = (i == j ? 1.f : 0.f) - dinv * aij;
      }
      S[i * LD + j] = l;
      S[j * LD + i] = l;
    }
  }
  __syncthreads();

  // ---- 2. Householder tridiagonalisation ------------------------------------------------------------------
  // Three barriers per reflector: every lane derives (beta, tau, scale) itself from the group partials of the
  // column norm, the p.v product rides on the matvec, and the NEXT column's norm partials ride on the update.
  colnorm_partials(S, LD, G, 0, sPart);
  __syncthreads();
  for (int k = 0; k + 2 < G; ++k) {
    const int m = G - k - 1;                        // order of the trailing block
    const float nrm2 = sum_partials(sPart);
    const float x0 = S[(k + 1) * LD + k];
    const float rest = nrm2 - x0 * x0;
    float tau = 0.f, scale = 0.f, beta = x0;
    if (rest > 1e-30f && rest > 1e-12f * nrm2) {    // identical in every lane
      beta = -copysignf(sqrtf(nrm2), x0);
      tau = (beta - x0) / beta;
      scale = 1.0f / (x0 - beta);
    }
    if (tid == 0) { sTau[k] = tau; sE[k] = beta; sD[k] = S[k * LD + k]; }
    if (tid < m) {
      const float xi = S[(k + 1 + tid) * LD + k];
      const float v = (tid == 0) ? 1.f : xi * scale;
      sV[tid] = v;
      // keep the reflector for the back-transformation; its leading 1 stays implicit (the slot still holds
      // x0, which the other lanes are reading right now)
      if (tid > 0) S[(k + 1 + tid) * LD + k] = v;
    }
    __syncthreads();                                                              // (1) v, params
    if (tau != 0.f) {
      // p = tau * S22 v : rows owned by 16-lane groups; partial p.v per group
      float pdot = 0.f;
      for (int i = grp; i < m; i += kTdGroups) {
        const float* row = S + (k + 1 + i) * LD + (k + 1);
        float acc = 0.f;
        for (int j = lane16; j < m; j += 16) acc = fmaf(row[j], sV[j], acc);
        acc = row_allreduce_sum(acc);
        const float pi = tau * acc;
        if (lane16 == 0) { sW[i] = pi; pdot = fmaf(pi, sV[i], pdot); }
      }
      if (lane16 == 0) sPd[grp] = pdot;
      __syncthreads();                                                            // (2) p, partial dots
      // w = p - (tau/2)(p.v) v is formed where it is used (same fmaf everywhere, so every lane sees the same w):
      // no write-back of w and no barrier for it
      const float alpha = -0.5f * tau * sum_partials(sPd);
      // S22 -= v w^T + w v^T ; lane 0 of a group also sees the new column k+1 -> next reflector's norm
      float sq = 0.f;
      for (int i = grp; i < m; i += kTdGroups) {
        float* row = S + (k + 1 + i) * LD + (k + 1);
        const float vi = sV[i], wi = fmaf(alpha, vi, sW[i]);
        for (int j = lane16; j < m; j += 16) {
          const float vj = sV[j], wj = fmaf(alpha, vj, sW[j]);
          const float nv = row[j] - (vi * wj + wi * vj);
          row[j] = nv;
          if (j == 0 && i >= 1) sq = fmaf(nv, nv, sq);
        }
      }
      if (lane16 == 0) sPart[grp] = sq;
    } else {
      colnorm_partials(S, LD, G, k + 1, sPart);
    }
    __syncthreads();                                                              // (3) trailing block, norms
  }
  if (tid == 0) {
    if (G >= 2) {
      sD[G - 2] = S[(G - 2) * LD + (G - 2)];
      sE[G - 2] = S[(G - 1) * LD + (G - 2)];
    }
    sD[G - 1] = S[(G - 1) * LD + (G - 1)];
    sE[G - 1] = 0.0;
  }
  __syncthreads();
  if (tid < G) sE2[tid] = sE[tid] * sE[tid];
  // Gershgorin range and pivmin
  double glo = 1e300, ghi = -1e300, emax = 0.0;
  if (tid < G) {
    const double el = tid > 0 ? fabs(sE[tid - 1]) : 0.0, er = tid + 1 < G ? fabs(sE[tid]) : 0.0;
    glo = sD[tid] - el - er;
    ghi = sD[tid] + el + er;
    emax = er * er;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    glo = fmin(glo, __shfl_xor(glo, off));
    ghi = fmax(ghi, __shfl_xor(ghi, off));
    emax = fmax(emax, __shfl_xor(emax, off));
  }
  __syncthreads();
  if ((tid & 63) == 0) { sLa[0][tid >> 6] = glo; sLb[0][tid >> 6] = ghi; sLc[0][tid >> 6] = emax; }
  __syncthreads();
  glo = sLa[0][0]; ghi = sLb[0][0]; emax = sLc[0][0];
#pragma unroll
  for (int w = 1; w < kTdThreads / 64; ++w) {       // waves 2.. hold the neutral elements (G <= 128 lanes)
    glo = fmin(glo, sLa[0][w]); ghi = fmax(ghi, sLb[0][w]); emax = fmax(emax, sLc[0][w]);
  }
  const double tnorm = fmax(fabs(glo), fabs(ghi));
  const double pivmin = fmax(emax, 1.0) * 2.2250738585072014e-308 * 4.0 + 1e-290;
  glo -= 1e-12 * tnorm + 1e-300;
  ghi += 1e-12 * tnorm + 1e-300;
  __syncthreads();

  // ---- 3. wanted eigenvalues: multisection, one wave = 64 shifts per eigenvalue and round -------------------
  const int sidx = tid >> 6, l64 = tid & 63;          // 8 waves
  const int want = smallest ? sidx : (G - 1 - sidx);  // ascending index of the eigenvalue this wave finds
  {
    double lo = glo, hi = ghi;
    if (sidx < ntot) {
      for (int round = 0; round < 5; ++round) {       // 65^-5 ~ 1e-9 of the Gershgorin range
        const double step = (hi - lo) * (1.0 / 65.0);
        const double sigma = lo + step * (l64 + 1);
        const int cnt = sturm_count(sD, sE2, G, sigma, pivmin);
        // lanes whose shift already has more than `want` eigenvalues below it; the first of them bounds lambda
        const unsigned long long above = __ballot(cnt > want);
        const int t = above ? __builtin_ctzll(above) : 64;
        const double nlo = lo + step * t;
        hi = (t == 64) ? hi : lo + step * (t + 1);
        lo = nlo;
      }
      if (l64 == 0) sLam[sidx] = 0.5 * (lo + hi);
    }
  }
  __syncthreads();

  // ---- 4a. eigenvectors of T: inverse iteration, one lane per vector (pivoted tridiagonal LU, fp64) ---------
  // All recurrences carry their running values in registers, so the LDS traffic (d, e in; factors out) is off
  // the dependent chain; the pivots are stored as reciprocals so the back-substitution has no division.
  if ((tid & 63) == 0 && (tid >> 6) < ntot) {        // one wave per vector: the chains run side by side
    const int s = tid >> 6, n = G;
    double* ra = sLa[s]; double* ub = sLb[s]; double* uc = sLc[s]; double* l = sLl[s]; double* z = sZ[s];
    unsigned char* piv = sPiv[s];
    const double lam = sLam[s];
    const double tiny = fmax(tnorm, 1.0) * 1.1e-16;
    // LU of (T - lam I) with row interchanges between neighbours: U = (1/ra, ub, uc), L = l, P = piv
    double ai = sD[0] - lam;                          // running diagonal / super-diagonal of row i
    double bi = (n > 1) ? sE[0] : 0.0;
    for (int i = 0; i + 1 < n; ++i) {
      const double sub = sE[i];
      const double a1 = sD[i + 1] - lam;
      const double b1 = (i + 2 < n) ? sE[i + 1] : 0.0;
      if (fabs(ai) >= fabs(sub)) {
        if (fabs(ai) < tiny) ai = tiny;
        const double r = fast_rcp_f64(ai);
        const double mult = sub * r;
        ra[i] = r; ub[i] = bi; uc[i] = 0.0; l[i] = mult; piv[i] = 0;
        ai = a1 - mult * bi;
        bi = b1;
      } else {
        const double r = fast_rcp_f64(sub);
        const double mult = ai * r;
        ra[i] = r; ub[i] = a1; uc[i] = b1; l[i] = mult; piv[i] = 1;      // row i <- old row i+1
        ai = bi - mult * a1;                                                // row i+1 <- old row i - mult * it
        bi = -mult * b1;
      }
    }
    if (fabs(ai) < tiny) ai = tiny;
    ra[n - 1] = fast_rcp_f64(ai); ub[n - 1] = 0.0; uc[n - 1] = 0.0;
    unsigned rng = 12345u + 977u * s;
    for (int i = 0; i < n; ++i) {                     // deterministic start vector in (-1, 1)
      rng = rng * 1664525u + 1013904223u;
      z[i] = (static_cast<double>(rng >> 8) / 8388608.0) - 1.0;
    }
    for (int it = 0; it < 3; ++it) {
      double zi = z[0];                               // forward: apply P, L^-1
      for (int i = 0; i + 1 < n; ++i) {
        double zn = z[i + 1];
        if (piv[i]) { const double t = zi; zi = zn; zn = t; }
        z[i] = zi;
        zi = zn - l[i] * zi;
      }
      double z1 = zi * ra[n - 1], z2 = 0.0, nr = z1 * z1;   // backward: U z = rhs (two super-diagonals)
      z[n - 1] = z1;
      for (int i = n - 2; i >= 0; --i) {
        const double zc = (z[i] - ub[i] * z1 - uc[i] * z2) * ra[i];
        z[i] = zc;
        nr = fma(zc, zc, nr);
        z2 = z1; z1 = zc;
      }
      nr = 1.0 / sqrt(nr);
      for (int i = 0; i < n; ++i) z[i] *= nr;
    }
  }
  __syncthreads();
  // ---- 4b. modified Gram-Schmidt among the vectors (wave 0; exact eigenvectors are orthogonal already) ------
  if (tid < 64) {
    for (int s = 1; s < ntot; ++s) {
      for (int t = 0; t < s; ++t) {
        double dot = 0.0;
        for (int i = tid; i < G; i += 64) dot += sZ[s][i] * sZ[t][i];
        dot = wave_sum_f64(dot);
        for (int i = tid; i < G; i += 64) sZ[s][i] -= dot * sZ[t][i];
      }
      double nr = 0.0;
      for (int i = tid; i < G; i += 64) nr += sZ[s][i] * sZ[s][i];
      nr = 1.0 / sqrt(wave_sum_f64(nr));
      for (int i = tid; i < G; i += 64) sZ[s][i] *= nr;
    }
  }
  __syncthreads();
  // ---- 4c. back-transformation v = H_0 H_1 ... H_{G-3} z : one wave per vector -------------------------------
  {
    const int wave = tid >> 6, lane = tid & 63;
    for (int s = wave; s < ntot; s += kTdThreads / 64) {
      double* z = sZ[s];
      for (int k = G - 3; k >= 0; --k) {
        const double tau = sTau[k];
        if (tau == 0.0) continue;                      // uniform
        const int m = G - k - 1;
        double dot = 0.0;
        for (int i = lane; i < m; i += 64) {
          const double vi = (i == 0) ? 1.0 : static_cast<double>(S[(k + 1 + i) * LD + k]);
          dot += vi * z[k + 1 + i];
        }
        dot = wave_sum_f64(dot) * tau;
        for (int i = lane; i < m; i += 64) {
          const double vi = (i == 0) ? 1.0 : static_cast<double>(S[(k + 1 + i) * LD + k]);
          z[k + 1 + i] -= dot * vi;
        }
      }
      // sign convention: component of largest magnitude positive (first such index on ties)
      double best = -1.0; int bi = 0x7fffffff;
      for (int i = lane; i < G; i += 64) {
        const double v = fabs(static_cast<double>(static_cast<float>(z[i])));
        if (v > best) { best = v; bi = i; }
      }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        const double ob = __shfl_xor(best, off);
        const int oi = __shfl_xor(bi, off);
        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
      }
      if (lane == 0) sSign[s] = z[bi] < 0.0 ? -1.f : 1.f;
    }
  }
  __syncthreads();

  // ---- 5. outputs (the first `skip` extracted pairs are dropped: MATRIX_SYM) ---------------------------------
  if (p.evals && tid < nsel)
    p.evals[static_cast<size_t>(blockIdx.x) * nsel + tid] = static_cast<float>(sLam[tid + skip]);
  if (p.evecs) {
    float* out = p.evecs + static_cast<size_t>(blockIdx.x) * G * nsel;
    for (int e = tid; e < G * nsel; e += kTdThreads) {
      const int i = e / nsel, mm = e - i * nsel;
      out[e] = static_cast<float>(sZ[mm + skip][i]) * sSign[mm + skip];
    }
  }
  if (p.order) {
    long long* out = p.order + static_cast<size_t>(blockIdx.x) * nsel * G;
    for (int e = tid; e < G * nsel; e += kTdThreads) {
      const int mm = e / G, i = e - mm * G;
      const float sg = sSign[mm + skip];
      const float vi = static_cast<float>(sZ[mm + skip][i]) * sg;
      int rk = 0;
      for (int j = 0; j < G; ++j) {
        const float vj = static_cast<float>(sZ[mm + skip][j]) * sg;
        rk += (vj < vi) || (vj == vi && j < i);
      }
      out[mm * G + rk] = i;
    }
  }
}

int launch_tridiag_topk(const EigArgs& a, hipStream_t s) {
  static const bool once = [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(laplacian_tridiag_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, kSpecMaxG * kTdLD * 4);
    return true;
  }();
  (void)once;
  hipLaunchKernelGGL(laplacian_tridiag_kernel, dim3(a.B), dim3(kTdThreads), sizeof(float) * kSpecMaxG * kTdLD, s, a);
  return static_cast<int>(hipGetLastError());
}

}  // namespace simamba
