// Scalar-load throughput probe for gfx950 (diagnostic tool): ns per s_load_dwordx16 per CU for
//   mode 0: every wave re-reads one 64-byte line (pure scalar-cache hits)
//   mode 1: every wave streams its own lines (cold misses, nothing shared)
//   mode 2: the waves of a workgroup stream the SAME lines in lockstep
//   mode 3: the waves of a workgroup stream the same lines, wave w delayed by w steps
//   mode 4: like 1 but each wave walks a 16 KB window repeatedly (fits the scalar cache only if alone)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void probe(const float* src, float* out, int iters, int mode, int wg_stride_floats) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nw = blockDim.x >> 6;
  const float* base = src + (size_t)blockIdx.x * wg_stride_floats;
  float acc = 0.f;
  for (int it = 0; it < iters; ++it) {
    int line;
    if (mode == 0) line = 0;
    else if (mode == 1) line = it * nw + wave;
    else if (mode == 2) line = it;
    else if (mode == 3) line = (it + 4 * (nw - 1 - wave));
    else line = (it & 255) + 256 * wave;
    const float* p = base + (size_t)line * 16;
    f32x16 v;
    asm volatile("s_load_dwordx16 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory");
    acc += v[0] + v[7] + v[15];
  }
  if (threadIdx.x % 64 == 0) out[blockIdx.x * nw + wave] = acc;
}

int main() {
  const size_t bytes = 1ull << 30;
  float *src, *out;
  (void)hipMalloc(&src, bytes);
  (void)hipMemset(src, 0, bytes);
  (void)hipMalloc(&out, 1 << 20);
  const int iters = 2000;
  for (int mode = 0; mode < 5; ++mode)
    for (int wpc : {4, 8, 16}) {          // waves per CU: 1 workgroup per CU of wpc waves
      const int blocks = 256;
      const int stride = (iters + 128) * 16 * 24;     // floats between workgroups: 256 x 3.27 MB < 1 GiB, > any window
      hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
      probe<<<blocks, 64 * wpc>>>(src, out, 10, mode, stride);
      (void)hipEventRecord(e0);
      probe<<<blocks, 64 * wpc>>>(src, out, iters, mode, stride);
      (void)hipEventRecord(e1);
      (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      printf("mode %d waves/CU %2d: %.1f ns per s_load per CU (%.2f ns per load per wave = latency-ish)\n", mode, wpc,
             ms * 1e6 / (double(iters) * wpc), ms * 1e6 / iters);
    }
  return 0;
}
