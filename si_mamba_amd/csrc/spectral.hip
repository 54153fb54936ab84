// k-NN graph, Laplacian eigen-decomposition and token ordering for gfx950.
//
// Replaces, per sample and in one workgroup each, the reference's
//   models/point_mamba.py:620-661 / :664-715   (pairwise distances, topk, index_put adjacency)
//   models/point_mamba.py:717-761 / :764-814   (Python loop of torch.linalg.eigh -> cuSOLVER)
//   models/point_mamba.py:820                  (torch.sort of each selected eigenvector)
// The G x G problem (G <= 128) lives entirely in LDS (matrix + eigenvector basis = 2*G*(G+1)*4
// <= 129 KiB of the CU's 160 KiB); the eigensolver is a cyclic two-sided Jacobi with the
// round-robin (tournament) ordering, G/2 disjoint rotations per step applied by all 1024 lanes.
// This file is compiled WITHOUT fast-math and with -ffp-contract=off so that distances and the
// Laplacian entries round exactly like the reference's unfused torch ops.
#include "common.h"

namespace simamba {

constexpr int kSpecMaxG = 128;
constexpr int kGraphThreads = 256;
constexpr int kEigThreads = 1024;

// ---------------------------------------------------------------------------------------------
// pass 0 (SIGMA_MEAN only): sum of all pairwise distances of the batch -> ws[0] (double)
__global__ void dist_sum_kernel(const float* __restrict__ pts, double* __restrict__ acc, int G, int F) {
  __shared__ double sred[kGraphThreads];
  const float* P = pts + static_cast<size_t>(blockIdx.x) * G * F;
  double s = 0.0;
  for (int e = threadIdx.x; e < G * G; e += kGraphThreads) {
    const int i = e / G, j = e - i * G;
    float d2 = 0.f;
    for (int f = 0; f < F; ++f) {
      const float df = P[i * F + f] - P[j * F + f];
      d2 = d2 + df * df;
    }
    s += static_cast<double>(sqrtf(d2));
  }
  sred[threadIdx.x] = s;
  __syncthreads();
  for (int w = kGraphThreads / 2; w > 0; w >>= 1) {
    if (threadIdx.x < w) sred[threadIdx.x] += sred[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) atomicAdd(acc, sred[0]);
}

// ---------------------------------------------------------------------------------------------
// k-NN adjacency of one sample per workgroup
__global__ __launch_bounds__(kGraphThreads) void knn_graph_kernel(const float* __restrict__ pts,
                                                                    float* __restrict__ adj,
                                                                    const double* __restrict__ dist_sum, int B,
                                                                    int G, int F, int knn, float alpha,
                                                                    unsigned flags) {
  extern __shared__ float sm[];
  float* sDist = sm;            // [G][G+1]
  float* sAdj = sm + G * (G + 1);   // [G][G+1]
  const int LD = G + 1;
  const float* P = pts + static_cast<size_t>(blockIdx.x) * G * F;
  for (int e = threadIdx.x; e < G * G; e += kGraphThreads) {
    const int i = e / G, j = e - i * G;
    float d2 = 0.f;
    for (int f = 0; f < F; ++f) {
      const float df = P[i * F + f] - P[j * F + f];
      d2 = d2 + df * df;
    }
    sDist[i * LD + j] = sqrtf(d2);
    sAdj[i * LD + j] = 0.f;
  }
  __syncthreads();
  const bool self_loop = flags & SIMAMBA_SPEC_SELF_LOOP;
  const bool binary = flags & SIMAMBA_SPEC_BINARY;
  const bool symmetric = flags & SIMAMBA_SPEC_SYMMETRIC;
  float inv2s2 = 0.f;
  if (flags & SIMAMBA_SPEC_SIGMA_MEAN) {
    const float sigma = static_cast<float>(*dist_sum / (static_cast<double>(B) * G * G));
    inv2s2 = 2.f * (sigma * sigma);
  }
  // two phases like the reference's two index_put calls: A[i, nn] = w, then A[nn, i] = w
  for (int phase = 0; phase < (symmetric ? 2 : 1); ++phase) {
    if (threadIdx.x < G) {
      const int i = threadIdx.x;
      float pv = -1.f;   // previous (value, index) in ascending lexicographic order
      int pi = -1;
      for (int m = 0; m <= knn; ++m) {
        float bv = 3.0e38f;
        int bi = -1;
        for (int j = 0; j < G; ++j) {
          const float v = sDist[i * LD + j];
          const bool after_prev = (v > pv) || (v == pv && j > pi);
          if (after_prev && (v < bv)) { bv = v; bi = j; }
        }
        pv = bv; pi = bi;
        if (bi < 0) break;                       // NaN distances: nothing left to pick
        if (m == 0 && !self_loop) continue;      // drop the nearest (the point itself)
        float w = 1.f;
        if (!binary) {
          const float dd = bv * bv;
          w = (flags & SIMAMBA_SPEC_SIGMA_MEAN) ? expf(-dd / inv2s2) : expf(-1.f * alpha * dd);
        }
        if (phase == 0) sAdj[i * LD + bi] = w; else sAdj[bi * LD + i] = w;
      }
    }
    __syncthreads();
  }
  float* out = adj + static_cast<size_t>(blockIdx.x) * G * G;
  for (int e = threadIdx.x; e < G * G; e += kGraphThreads) {
    const int i = e / G, j = e - i * G;
    out[e] = sAdj[i * LD + j];
  }
}

// ---------------------------------------------------------------------------------------------
// Laplacian + Jacobi eigensolver + selection + argsort, one sample per workgroup
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

__global__ __launch_bounds__(kEigThreads) void laplacian_eig_kernel(EigArgs p) {
  extern __shared__ float sm[];
  const int G = p.G, LD = G + 1;
  float* S = sm;                  // [G][LD]  matrix being diagonalised
  float* V = S + G * LD;          // [G][LD]  accumulated rotations (columns = eigenvectors)
  __shared__ float sDeg[kSpecMaxG];
  __shared__ float sCs[kSpecMaxG];        // (s, tau = s / (1 + c)) per pair
  __shared__ float sPiv[2 * kSpecMaxG];   // rotated pivot diagonal (a_pp', a_qq') per pair
  __shared__ int sPq[kSpecMaxG];          // (p, q) per pair
  __shared__ int sRank[kSpecMaxG];        // eigenvalue index by ascending rank
  __shared__ int sSel[kSpecMaxG];
  __shared__ float sSign[kSpecMaxG];
  __shared__ int sFlag;
  const int tid = threadIdx.x;
  const float* A = p.adj + static_cast<size_t>(blockIdx.x) * G * G;

  // A <- (A + A^T) / 2 into V (scratch), degrees, Laplacian lower triangle mirrored into S
  for (int e = tid; e < G * G; e += kEigThreads) {
    const int i = e / G, j = e - i * G;
    V[i * LD + j] = (A[i * G + j] + A[j * G + i]) / 2.f;
  }
  __syncthreads();
  if (tid < G) {
    float s = 0.f;
    for (int j = 0; j < G; ++j) s = s + V[tid * LD + j];
    sDeg[tid] = s;
  }
  __syncthreads();
  const bool msym = p.flags & SIMAMBA_SPEC_MATRIX_SYM;
  for (int e = tid; e < G * G; e += kEigThreads) {
    const int i = e / G, j = e - i * G;
    if (i >= j) {   // eigh(UPLO='L'): only the lower triangle of the (unsymmetric) L is read
      float l;
      if (msym) {
        const float di = powf(sDeg[i], -0.5f), dj = powf(sDeg[j], -0.5f);
        l = (i == j ? 1.f : 0.f) - (di * V[i * LD + j]) * dj;
      } else {
        const float dinv = 1.0f / (sDeg[i] + 1e-6f);
        l = (i == j ? 1.f : 0.f) - dinv * V[i * LD + j];
      }
      S[i * LD + j] = l;
      S[j * LD + i] = l;
    }
  }
  __syncthreads();
  for (int e = tid; e < G * G; e += kEigThreads) {
    const int i = e / G, j = e - i * G;
    V[i * LD + j] = (i == j) ? 1.f : 0.f;
  }
  __syncthreads();

  // ---- cyclic Jacobi, tournament ordering ------------------------------------------------
  const int M = G + (G & 1);      // players (one dummy when G is odd)
  const int half = M / 2;
  for (int sweep = 0; sweep < 30; ++sweep) {
    if (tid == 0) sFlag = 0;
    __syncthreads();
    for (int step = 0; step < M - 1; ++step) {
      if (tid < half) {
        int a, b;
        if (tid == 0) { a = M - 1; b = step % (M - 1); }
        else { a = (step + tid) % (M - 1); b = (step + M - 1 - tid) % (M - 1); }
        int pp = a < b ? a : b, qq = a < b ? b : a;
        // Rutishauser's form: x' = x - s (y + tau x), y' = y + s (x - tau y), tau = s / (1 + c).
        // (c x - s y with c rounded to 1 for small angles grows every norm by t^2/2 per rotation.)
        float s = 0.f, tq = 0.f, dpp_ = 0.f, dqq_ = 0.f;
        if (qq < G) {
          const float apq = S[pp * LD + qq];
          const float app = S[pp * LD + pp], aqq = S[qq * LD + qq];
          dpp_ = app; dqq_ = aqq;
          if (fabsf(apq) > 1e-12f * (fabsf(app) + fabsf(aqq)) + 1e-37f) {
            const float theta = (aqq - app) / (2.f * apq);
            const float t = (theta >= 0.f ? 1.f : -1.f) / (fabsf(theta) + sqrtf(1.f + theta * theta));
            const float c = 1.f / sqrtf(1.f + t * t);
            s = t * c;
            tq = s / (1.f + c);
            dpp_ = app - t * apq;
            dqq_ = aqq + t * apq;
            if (fabsf(apq) > 5e-8f * (fabsf(app) + fabsf(aqq)) + 1e-10f) sFlag = 1;
          }
        } else {
          qq = -1;
        }
        sPq[2 * tid] = pp; sPq[2 * tid + 1] = qq;
        sCs[2 * tid] = s; sCs[2 * tid + 1] = tq;
        sPiv[2 * tid] = dpp_; sPiv[2 * tid + 1] = dqq_;
      }
      __syncthreads();
      // rows: S <- J^T S
      for (int e = tid; e < half * G; e += kEigThreads) {
        const int pr = e / G, j = e - pr * G;
        const int pp = sPq[2 * pr], qq = sPq[2 * pr + 1];
        const float s = sCs[2 * pr], tq = sCs[2 * pr + 1];
        if (qq >= 0 && s != 0.f) {
          const float x = S[pp * LD + j], y = S[qq * LD + j];
          S[pp * LD + j] = x - s * (y + tq * x);
          S[qq * LD + j] = y + s * (x - tq * y);
        }
      }
      __syncthreads();
      // columns: S <- S J, V <- V J
      for (int e = tid; e < half * G; e += kEigThreads) {
        const int pr = e / G, i = e - pr * G;
        const int pp = sPq[2 * pr], qq = sPq[2 * pr + 1];
        const float s = sCs[2 * pr], tq = sCs[2 * pr + 1];
        if (qq >= 0 && s != 0.f) {
          const float x = S[i * LD + pp], y = S[i * LD + qq];
          float xn = x - s * (y + tq * x), yn = y + s * (x - tq * y);
          // the 2x2 pivot block is known in closed form: exact zero off-diagonal, a_pp - t a_pq, a_qq + t a_pq
          if (i == pp) { xn = sPiv[2 * pr]; yn = 0.f; }
          if (i == qq) { xn = 0.f; yn = sPiv[2 * pr + 1]; }
          S[i * LD + pp] = xn;
          S[i * LD + qq] = yn;
          const float vx = V[i * LD + pp], vy = V[i * LD + qq];
          V[i * LD + pp] = vx - s * (vy + tq * vx);
          V[i * LD + qq] = vy + s * (vx - tq * vy);
        }
      }
      __syncthreads();
    }
    if (sFlag == 0) break;
    __syncthreads();
  }

  // ---- sort eigenvalues ascending (rank sort, ties by index) ------------------------------
  if (tid < G) {
    const float li = S[tid * LD + tid];
    int rk = 0;
    for (int j = 0; j < G; ++j) {
      const float lj = S[j * LD + j];
      rk += (lj < li) || (lj == li && j < tid);
    }
    sRank[rk] = tid;
  }
  __syncthreads();
  if (p.all_evals && tid < G) p.all_evals[static_cast<size_t>(blockIdx.x) * G + tid] = S[sRank[tid] * LD + sRank[tid]];

  // selected columns: smallest -> ranks 0..; largest -> ranks G-1, G-2, ...; MATRIX_SYM drops the first
  const bool smallest = p.flags & SIMAMBA_SPEC_SMALLEST;
  const int skip = msym ? 1 : 0;
  const int nsel = p.k;
  if (tid < nsel) {
    const int r = smallest ? (tid + skip) : (G - 1 - tid - skip);
    sSel[tid] = sRank[r];
  }
  __syncthreads();
  // sign convention: component of largest magnitude positive (first such index on ties).
  // columns handled: all G when all_evecs is wanted, else the selected ones.
  const int ncols = p.all_evecs ? G : nsel;
  if (tid < ncols) {
    const int col = p.all_evecs ? sRank[tid] : sSel[tid];
    float best = -1.f, sgn = 1.f;
    for (int i = 0; i < G; ++i) {
      const float v = V[i * LD + col];
      if (fabsf(v) > best) { best = fabsf(v); sgn = v < 0.f ? -1.f : 1.f; }
    }
    sSign[col] = sgn;
  }
  __syncthreads();
  if (p.all_evecs) {
    float* out = p.all_evecs + static_cast<size_t>(blockIdx.x) * G * G;
    for (int e = tid; e < G * G; e += kEigThreads) {
      const int i = e / G, r = e - i * G;
      const int col = sRank[r];
      out[e] = V[i * LD + col] * sSign[col];
    }
  }
  if (p.evals && tid < nsel) p.evals[static_cast<size_t>(blockIdx.x) * nsel + tid] = S[sSel[tid] * LD + sSel[tid]];
  if (p.evecs) {
    float* out = p.evecs + static_cast<size_t>(blockIdx.x) * G * nsel;
    for (int e = tid; e < G * nsel; e += kEigThreads) {
      const int i = e / nsel, m = e - i * nsel;
      out[e] = V[i * LD + sSel[m]] * sSign[sSel[m]];
    }
  }
  if (p.order) {
    long long* out = p.order + static_cast<size_t>(blockIdx.x) * nsel * G;
    for (int e = tid; e < G * nsel; e += kEigThreads) {
      const int m = e / G, i = e - m * G;
      const int col = sSel[m];
      const float sg = sSign[col];
      const float vi = V[i * LD + col] * sg;
      int rk = 0;
      for (int j = 0; j < G; ++j) {
        const float vj = V[j * LD + col] * sg;
        rk += (vj < vi) || (vj == vi && j < i);
      }
      out[m * G + rk] = i;
    }
  }
}

// ---------------------------------------------------------------------------------------------
__global__ void argsort_rows_kernel(const float* __restrict__ vals, long long* __restrict__ idx, int n) {
  extern __shared__ float sv[];
  const float* row = vals + static_cast<size_t>(blockIdx.x) * n;
  for (int i = threadIdx.x; i < n; i += blockDim.x) sv[i] = row[i];
  __syncthreads();
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const float vi = sv[i];
    int rk = 0;
    for (int j = 0; j < n; ++j) rk += (sv[j] < vi) || (sv[j] == vi && j < i);
    idx[static_cast<size_t>(blockIdx.x) * n + rk] = i;
  }
}

}  // namespace simamba

using namespace simamba;

extern "C" size_t simamba_spectral_workspace_bytes(int B, int G) {
  if (B < 0 || G < 0) return 0;
  return 256 + sizeof(float) * static_cast<size_t>(B) * G * G;
}

// raise the dynamic-LDS cap of the two big-tile kernels once per process (thread-safe static init;
// the value never changes afterwards, so this is not observable state)
static void ensure_lds_attrs() {
  static const bool once = [] {
    const int cap = 2 * kSpecMaxG * (kSpecMaxG + 1) * 4;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(laplacian_eig_kernel),
                        hipFuncAttributeMaxDynamicSharedMemorySize, cap);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(knn_graph_kernel),
                        hipFuncAttributeMaxDynamicSharedMemorySize, cap);
    return true;
  }();
  (void)once;
}

static int check_groups(int B, int G) {
  if (B < 0) return SIMAMBA_E_SHAPE;
  if (G < 2 || G > kSpecMaxG) return SIMAMBA_E_GROUPS;
  return SIMAMBA_OK;
}

extern "C" int simamba_knn_graph(const float* points, float* adj, void* workspace, size_t ws_bytes, int B, int G,
                                 int F, int knn, float alpha, unsigned flags, void* stream) {
  if (!points || !adj) return SIMAMBA_E_NULLPTR;
  int rc = check_groups(B, G);
  if (rc) return rc;
  if (F < 1 || knn < 0 || knn + 1 > G) return SIMAMBA_E_GROUPS;
  if (B == 0) return SIMAMBA_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  double* acc = nullptr;
  if (flags & SIMAMBA_SPEC_SIGMA_MEAN) {
    if (!workspace || ws_bytes < 256) return SIMAMBA_E_WORKSPACE;
    acc = static_cast<double*>(workspace);
    hipError_t e = hipMemsetAsync(acc, 0, sizeof(double), s);
    if (e != hipSuccess) return static_cast<int>(e);
    hipLaunchKernelGGL(dist_sum_kernel, dim3(B), dim3(kGraphThreads), 0, s, points, acc, G, F);
  }
  ensure_lds_attrs();
  const size_t smem = sizeof(float) * 2 * G * (G + 1);
  hipLaunchKernelGGL(knn_graph_kernel, dim3(B), dim3(kGraphThreads), smem, s, points, adj, acc, B, G, F, knn, alpha,
                     flags);
  return static_cast<int>(hipGetLastError());
}

extern "C" int simamba_laplacian_topk(const float* adj, float* evals, float* evecs, long long* order,
                                      float* all_evals, float* all_evecs, int B, int G, int k, unsigned flags,
                                      void* stream) {
  if (!adj) return SIMAMBA_E_NULLPTR;
  int rc = check_groups(B, G);
  if (rc) return rc;
  const int need = k + ((flags & SIMAMBA_SPEC_MATRIX_SYM) ? 1 : 0);
  if (k < 0 || need > G) return SIMAMBA_E_GROUPS;
  if (B == 0) return SIMAMBA_OK;
  EigArgs a{adj, evals, evecs, order, all_evals, all_evecs, B, G, k, flags};
  const size_t smem = sizeof(float) * 2 * G * (G + 1);
  ensure_lds_attrs();
  hipLaunchKernelGGL(laplacian_eig_kernel, dim3(B), dim3(kEigThreads), smem, static_cast<hipStream_t>(stream), a);
  return static_cast<int>(hipGetLastError());
}

extern "C" int simamba_spectral_topk(const float* centers, float* evals, float* evecs, long long* order,
                                     void* workspace, size_t ws_bytes, int B, int G, int knn, float alpha, int k,
                                     unsigned flags, void* stream) {
  if (!centers || !workspace) return SIMAMBA_E_NULLPTR;
  if (ws_bytes < simamba_spectral_workspace_bytes(B, G)) return SIMAMBA_E_WORKSPACE;
  float* adj = reinterpret_cast<float*>(static_cast<char*>(workspace) + 256);
  int rc = simamba_knn_graph(centers, adj, workspace, ws_bytes, B, G, 3, knn, alpha, flags, stream);
  if (rc) return rc;
  return simamba_laplacian_topk(adj, evals, evecs, order, nullptr, nullptr, B, G, k, flags, stream);
}

extern "C" int simamba_argsort_rows(const float* vals, long long* idx, int rows, int n, void* stream) {
  if (!vals || !idx) return SIMAMBA_E_NULLPTR;
  if (rows < 0 || n < 0 || n > 1024) return SIMAMBA_E_SHAPE;
  if (rows == 0 || n == 0) return SIMAMBA_OK;
  hipLaunchKernelGGL(argsort_rows_kernel, dim3(rows), dim3(256), sizeof(float) * n,
                     static_cast<hipStream_t>(stream), vals, idx, n);
  return static_cast<int>(hipGetLastError());
}
