// k-NN graph, Laplacian eigen-decomposition and token ordering for gfx950.
//
// Replaces, per sample and in one workgroup each, the reference's
//   models/point_mamba.py:620-661 / :664-715   (pairwise distances, topk, index_put adjacency)
//   models/point_mamba.py:717-761 / :764-814   (Python loop of torch.linalg.eigh -> cuSOLVER)
//   models/point_mamba.py:820                  (torch.sort of each selected eigenvector)
// The G x G problem (G <= 128) lives entirely in LDS (matrix + eigenvector basis = 2*128*132*4
// = 132 KiB of the CU's 160 KiB); the eigensolver is a cyclic one-sided (Hestenes) Jacobi with the
// round-robin (tournament) ordering: G/2 disjoint column pairs per step, one 16-lane DPP row per pair.
// This file is compiled WITHOUT fast-math and with -ffp-contract=off so that distances and the
// Laplacian entries round exactly like the reference's unfused torch ops.
#include "spectral_common.h"

namespace simamba {

constexpr int kGraphThreads = 256;   // dist_sum_kernel
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
// k-NN adjacency of one sample per workgroup.  FOUR lanes (one quad) per graph node: lane p of the quad keeps the
// distances to nodes p, p + 4, p + 8, ... in 32 registers (the loops are unrolled, every access is statically
// indexed); each of the knn+1 selection passes is a compare/select sweep over those registers followed by a
// two-step DPP quad reduction of the (value, index) pair, lexicographic so that ties go to the lower index like
// a stable sort.  The picked (index, weight) lists go to LDS once and feed both index_put phases of the
// reference.  (History: sweeping an LDS copy of the matrix cost 0.64 ms per batch of 64; one lane per node with
// 128 registers 0.32 ms -- two waves per CU are latency-bound.)
constexpr int kKnnMaxK = 32;   // knn + 1 <= 32 list entries per node
constexpr int kKnnLanes = 4;   // lanes per node
constexpr int kKnnPer = kSpecMaxG / kKnnLanes;   // distances per lane

// lexicographic (value, index) minimum over the 4 lanes of a quad, result in every lane
__device__ __forceinline__ void quad_argmin(float& v, int& i) {
#pragma unroll
  for (int step = 0; step < 2; ++step) {
    const float ov = step == 0 ? dpp<DPP_QUAD_XOR1>(v, v) : dpp<DPP_QUAD_XOR2>(v, v);
    const int oi = step == 0 ? __builtin_amdgcn_update_dpp(i, i, DPP_QUAD_XOR1, 0xf, 0xf, false)
                             : __builtin_amdgcn_update_dpp(i, i, DPP_QUAD_XOR2, 0xf, 0xf, false);
    const bool take = (ov < v) || (ov == v && static_cast<unsigned>(oi) < static_cast<unsigned>(i));
    v = take ? ov : v;
    i = take ? oi : i;
  }
}

__global__ __launch_bounds__(kSpecMaxG * kKnnLanes) void knn_graph_kernel(const float* __restrict__ pts,
                                                                            float* __restrict__ adj,
                                                                            const double* __restrict__ dist_sum,
                                                                            int B, int G, int F, int knn, float alpha,
                                                                            unsigned flags) {
  extern __shared__ float sm[];
  float* sAdj = sm;                                   // [G][G+1]
  float* sP = sAdj + G * (G + 1);                     // [G][F]
  float* sWt = sP + G * F;                            // [G][kKnnMaxK]
  int* sNb = reinterpret_cast<int*>(sWt + G * kKnnMaxK);   // [G][kKnnMaxK]
  const int LD = G + 1;
  const int tid = threadIdx.x, nthr = blockDim.x;
  const float* P = pts + static_cast<size_t>(blockIdx.x) * G * F;
  for (int e = tid; e < G * F; e += nthr) sP[e] = P[e];
  for (int e = tid; e < G * LD; e += nthr) sAdj[e] = 0.f;
  __syncthreads();
  const bool self_loop = flags & SIMAMBA_SPEC_SELF_LOOP;
  const bool binary = flags & SIMAMBA_SPEC_BINARY;
  const bool symmetric = flags & SIMAMBA_SPEC_SYMMETRIC;
  float inv2s2 = 0.f;
  if (flags & SIMAMBA_SPEC_SIGMA_MEAN) {
    const float sigma = static_cast<float>(*dist_sum / (static_cast<double>(B) * G * G));
    inv2s2 = 2.f * (sigma * sigma);
  }
  const int i = tid >> 2, part = tid & 3;             // node, lane of its quad (blockDim = 4 G: every quad is whole)
  int nlist = 0;
  {
    float d[kKnnPer];
#pragma unroll
    for (int jj = 0; jj < kKnnPer; ++jj) {
      const int j = part + kKnnLanes * jj;
      float d2 = 0.f;
      if (j < G) {
        for (int f = 0; f < F; ++f) {
          const float df = sP[i * F + f] - sP[j * F + f];
          d2 = d2 + df * df;
        }
        d[jj] = sqrtf(d2);
      } else {
        d[jj] = __builtin_inff();
      }
    }
    float pv = -1.f;   // previous pick, ascending lexicographic (value, index) order
    int pi = -1;
    for (int m = 0; m <= knn; ++m) {
      float bv = 3.0e38f;
      int bi = 0x7fffffff;
#pragma unroll
      for (int jj = 0; jj < kKnnPer; ++jj) {
        const float v = d[jj];
        const int j = part + kKnnLanes * jj;
        const bool after_prev = (v > pv) || (v == pv && j > pi);
        if (after_prev && (v < bv)) { bv = v; bi = j; }     // ascending j inside the lane: first hit is the lowest
      }
      quad_argmin(bv, bi);
      pv = bv; pi = bi;
      if (bi == 0x7fffffff) break;             // NaN distances: nothing left to pick (quad-uniform)
      if (m == 0 && !self_loop) continue;      // drop the nearest (the point itself)
      float w = 1.f;
      if (!binary) {
        const float dd = bv * bv;
        w = (flags & SIMAMBA_SPEC_SIGMA_MEAN) ? expf(-dd / inv2s2) : expf(-1.f * alpha * dd);
      }
      if (part == 0) {
        sNb[i * kKnnMaxK + nlist] = bi;
        sWt[i * kKnnMaxK + nlist] = w;
      }
      ++nlist;
    }
  }
  __syncthreads();
  // two phases like the reference's two index_put calls: A[i, nn] = w, then A[nn, i] = w; lane 0 of each quad
  for (int phase = 0; phase < (symmetric ? 2 : 1); ++phase) {
    if (part == 0) {
      for (int q = 0; q < nlist; ++q) {
        const int bi = sNb[i * kKnnMaxK + q];
        const float w = sWt[i * kKnnMaxK + q];
        if (phase == 0) sAdj[i * LD + bi] = w; else sAdj[bi * LD + i] = w;
      }
    }
    __syncthreads();
  }
  float* out = adj + static_cast<size_t>(blockIdx.x) * G * G;
  for (int e = tid; e < G * G; e += nthr) {
    const int r = e / G, c = e - r * G;
    out[e] = sAdj[r * LD + c];
  }
}

// ---------------------------------------------------------------------------------------------
// Laplacian + Jacobi eigensolver + selection + argsort, one sample per workgroup.
//
// One-sided (Hestenes) Jacobi on W = S + 2I (S = the mirrored lower triangle of the Laplacian; its
// spectrum lies in [-1, 3] by Gershgorin, so the shifted matrix is positive definite): column pairs (p,q)
// of W are rotated until all columns are mutually orthogonal.  Then W = (S + 2I) V with V orthogonal and
// W^T W diagonal, i.e. w_j = lambda'_j v_j: the eigenvector is the normalised column, the eigenvalue its
// norm minus the shift -- no eigenvector matrix is accumulated, the whole state is ONE G x G LDS image.
// Why one-sided: a 16-lane DPP row owns one pair -- it forms the three inner products (alpha, beta, gamma)
// with a DPP all-reduce, derives the rotation redundantly in every lane and applies it to its slice of the
// two columns (16-byte LDS accesses), so a step of G/2 disjoint pairs needs ONE workgroup barrier (the
// two-sided form needs three plus a serial parameter phase and measured 6.2 ms per batch of 64).
// Columns are stored as contiguous rows Wt[p][:].

constexpr int kEigLD = kSpecMaxG + 4;     // row stride of the LDS image (floats), keeps rows 16-B aligned
constexpr int kEigVec = kSpecMaxG / 64;   // 16-byte chunks of a column per lane (lane l: floats 4l.., 64+4l..)
constexpr float kEigShift = 2.0f;

__global__ __launch_bounds__(kEigThreads) void laplacian_eig_kernel(EigArgs p) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int G = p.G;
  constexpr int LD = kEigLD;
  float* Wt = sm;                 // [kSpecMaxG][LD]  column p of W at Wt[p*LD + i]; rows >= G unused
  float* Vt = Wt;                 // after convergence the columns are normalised in place: V
  __shared__ float sDeg[kSpecMaxG];
  __shared__ float sEval[kSpecMaxG];
  __shared__ int sRank[kSpecMaxG];        // eigenvalue index by ascending rank
  __shared__ int sSel[kSpecMaxG];
  __shared__ float sSign[kSpecMaxG];
  __shared__ int sFlag;
  const int tid = threadIdx.x;
  const int lane16 = tid & 15;
  const int grp = tid >> 4;               // 64 groups of 16 lanes: one column pair each
  const float* A = p.adj + static_cast<size_t>(blockIdx.x) * G * G;

  // degrees of (A + A^T)/2, then the Laplacian's lower triangle mirrored (+ shift) into Wt
  if (tid < G) {
    float s = 0.f;
    for (int j = 0; j < G; ++j) s = s + (A[tid * G + j] + A[j * G + tid]) / 2.f;
    sDeg[tid] = s;
  }
  for (int e = tid; e < kSpecMaxG * LD; e += kEigThreads) Wt[e] = 0.f;
  __syncthreads();
  const bool msym = p.flags & SIMAMBA_SPEC_MATRIX_SYM;
  for (int e = tid; e < G * G; e += kEigThreads) {
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
      const float w = l + (i == j ? kEigShift : 0.f);
      Wt[i * LD + j] = w;
      Wt[j * LD + i] = w;
    }
  }
  __syncthreads();

  // ---- cyclic one-sided Jacobi, tournament ordering: group `grp` owns pair `grp` of each step ---------
  const int M = G + (G & 1);      // players (one dummy when G is odd)
  const int half = M / 2;         // <= 64 pairs per step
  // |cos(w_p, w_q)| below which a pair counts as orthogonal.  The fp32 inner product of two 128-vectors
  // carries ~sqrt(G) * 2^-24 ~ 7e-7 of relative noise, so a tighter bound never terminates (measured: all
  // 24 sweeps); rotations are still applied down to kApply so a sweep only refines.
  const float kDone = 1.5e-6f, kApply = 2.0e-7f;
  for (int sweep = 0; sweep < 24; ++sweep) {
    if (tid == 0) sFlag = 0;
    __syncthreads();
    for (int step = 0; step < M - 1; ++step) {
      if (grp < half) {
        // step < M-1 and grp < M/2, so one conditional subtraction replaces each modulo
        int a = step + grp, b = step + (M - 1) - grp;
        a = a >= M - 1 ? a - (M - 1) : a;
        b = b >= M - 1 ? b - (M - 1) : b;
        if (grp == 0) { a = M - 1; b = step; }
        const int pp = a < b ? a : b, qq = a < b ? b : a;
        if (qq < G) {
          float4 wp[kEigVec], wq[kEigVec];
          float al = 0.f, be = 0.f, ga = 0.f;
#pragma unroll
          for (int k = 0; k < kEigVec; ++k) {    // entries >= G of a row are zero and stay zero
            wp[k] = *reinterpret_cast<const float4*>(Wt + pp * LD + 64 * k + 4 * lane16);
            wq[k] = *reinterpret_cast<const float4*>(Wt + qq * LD + 64 * k + 4 * lane16);
            al += wp[k].x * wp[k].x + wp[k].y * wp[k].y + wp[k].z * wp[k].z + wp[k].w * wp[k].w;
            be += wq[k].x * wq[k].x + wq[k].y * wq[k].y + wq[k].z * wq[k].z + wq[k].w * wq[k].w;
            ga += wp[k].x * wq[k].x + wp[k].y * wq[k].y + wp[k].z * wq[k].z + wp[k].w * wq[k].w;
          }
          al = row_allreduce_sum(al);
          be = row_allreduce_sum(be);
          ga = row_allreduce_sum(ga);
          const float g2 = ga * ga, ab = al * be;
          if (g2 > (kApply * kApply) * ab && fabsf(ga) > 1e-30f) {   // uniform over the 16 lanes
            // Any (s, tq) with tq = s / (1 + sqrt(1 - s^2)) is an exactly orthogonal rotation in the
            // Rutishauser form below, so the hardware rcp / rsq approximations only perturb the ANGLE.
            const float zeta = (be - al) * __builtin_amdgcn_rcpf(2.f * ga);
            const float t = (zeta >= 0.f ? 1.f : -1.f) *
                            __builtin_amdgcn_rcpf(fabsf(zeta) + __builtin_amdgcn_sqrtf(1.f + zeta * zeta));
            const float s = t * __builtin_amdgcn_rsqf(1.f + t * t);
            const float c = __builtin_amdgcn_sqrtf(fmaxf(1.f - s * s, 0.f));
            const float tq = s * __builtin_amdgcn_rcpf(1.f + c);
#pragma unroll
            for (int k = 0; k < kEigVec; ++k) {
              float4 x = wp[k], y = wq[k], xn, yn;
              xn.x = x.x - s * (y.x + tq * x.x); yn.x = y.x + s * (x.x - tq * y.x);
              xn.y = x.y - s * (y.y + tq * x.y); yn.y = y.y + s * (x.y - tq * y.y);
              xn.z = x.z - s * (y.z + tq * x.z); yn.z = y.z + s * (x.z - tq * y.z);
              xn.w = x.w - s * (y.w + tq * x.w); yn.w = y.w + s * (x.w - tq * y.w);
              *reinterpret_cast<float4*>(Wt + pp * LD + 64 * k + 4 * lane16) = xn;
              *reinterpret_cast<float4*>(Wt + qq * LD + 64 * k + 4 * lane16) = yn;
            }
            if (lane16 == 0 && g2 > (kDone * kDone) * ab) sFlag = 1;
          }
        }
      }
      __syncthreads();
    }
    if (sFlag == 0) break;      // measured: 7-9 sweeps at G = 64 / 128
    __syncthreads();
  }

  // ---- eigenvalue = column norm - shift; eigenvector = normalised column (in place) -------------------
  for (int col = grp; col < G; col += kEigThreads / 16) {
    float4 w[kEigVec];
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < kEigVec; ++k) {
      w[k] = *reinterpret_cast<const float4*>(Wt + col * LD + 64 * k + 4 * lane16);
      acc += w[k].x * w[k].x + w[k].y * w[k].y + w[k].z * w[k].z + w[k].w * w[k].w;
    }
    acc = row_allreduce_sum(acc);
    const float nrm = sqrtf(acc);
    const float inv = 1.0f / nrm;
#pragma unroll
    for (int k = 0; k < kEigVec; ++k) {
      w[k].x *= inv; w[k].y *= inv; w[k].z *= inv; w[k].w *= inv;
      *reinterpret_cast<float4*>(Wt + col * LD + 64 * k + 4 * lane16) = w[k];
    }
    if (lane16 == 0) sEval[col] = nrm - kEigShift;
  }
  __syncthreads();
  // ---- sort eigenvalues ascending (rank sort, ties by index) ------------------------------
  if (tid < G) {
    const float li = sEval[tid];
    int rk = 0;
    for (int j = 0; j < G; ++j) {
      const float lj = sEval[j];
      rk += (lj < li) || (lj == li && j < tid);
    }
    sRank[rk] = tid;
  }
  __syncthreads();
  if (p.all_evals && tid < G) p.all_evals[static_cast<size_t>(blockIdx.x) * G + tid] = sEval[sRank[tid]];

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
      const float v = Vt[col * LD + i];
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
      out[e] = Vt[col * LD + i] * sSign[col];
    }
  }
  if (p.evals && tid < nsel) p.evals[static_cast<size_t>(blockIdx.x) * nsel + tid] = sEval[sSel[tid]];
  if (p.evecs) {
    float* out = p.evecs + static_cast<size_t>(blockIdx.x) * G * nsel;
    for (int e = tid; e < G * nsel; e += kEigThreads) {
      const int i = e / nsel, m = e - i * nsel;
      out[e] = Vt[sSel[m] * LD + i] * sSign[sSel[m]];
    }
  }
  if (p.order) {
    long long* out = p.order + static_cast<size_t>(blockIdx.x) * nsel * G;
    for (int e = tid; e < G * nsel; e += kEigThreads) {
      const int m = e / G, i = e - m * G;
      const int col = sSel[m];
      const float sg = sSign[col];
      const float vi = Vt[col * LD + i] * sg;
      int rk = 0;
      for (int j = 0; j < G; ++j) {
        const float vj = Vt[col * LD + j] * sg;
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
    const int cap = 4 * (kSpecMaxG * (kSpecMaxG + 1) + kSpecMaxG * 64 + 2 * kSpecMaxG * kKnnMaxK);   // knn_graph_kernel
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
  if (F < 1 || F > 64 || knn < 0 || knn + 1 > G || knn + 1 > kKnnMaxK) return SIMAMBA_E_GROUPS;
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
  const size_t smem = sizeof(float) * (static_cast<size_t>(G) * (G + 1) + G * F + 2 * G * kKnnMaxK);
  hipLaunchKernelGGL(knn_graph_kernel, dim3(B), dim3(G * kKnnLanes), smem, s, points, adj, acc, B, G, F, knn, alpha,
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
  // top-k only: Householder tridiagonalisation + bisection + inverse iteration (spectral_tridiag.hip);
  // the full spectrum / full basis (never used by the reference's forward) stays on the Jacobi kernel
  if (!all_evals && !all_evecs && need <= kTdMaxSel && G >= 3)
    return launch_tridiag_topk(a, static_cast<hipStream_t>(stream));
  const size_t smem = sizeof(float) * kSpecMaxG * kEigLD;
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
