// Access-pattern probe for the lane-per-channel scan forward (diagnostic tool, not part of the library).
// Streams three (rows, 128) fp32 tensors in and one out -- the byte count of the selective scan at L = 128 -- with
// the SEGMENT shapes a wave of the scan kernel can use, and an optional block of dependent VALU work between
// chunks that stands for the recurrence (so that the two halves of a 128-byte line are requested microseconds
// apart, as in the real kernel).  Output: us and GB/s per variant.
//   seg64    16-step chunks: each load instruction covers 16 rows x 64 B
//   seg128   32-step chunks: 8 rows x 128 B (whole lines)
//   pair64   16-step consumption, but the two 64-B halves of a line are requested back to back (2 instructions)
//   seg512   whole rows (2 rows x 512 B per instruction)
//   linear   plain coalesced stream (1 KiB contiguous per instruction): the copy-kernel ceiling
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int L = 128;

__device__ __forceinline__ float burn(float x, int n) {
  for (int i = 0; i < n; ++i) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(x));
  return x;
}

// variant: 0 seg64, 1 seg128, 2 pair64, 3 seg512
template <int V>
__global__ __launch_bounds__(256) void probe(const float4* __restrict__ a, const float4* __restrict__ b,
                                             const float4* __restrict__ c, float4* __restrict__ o, int rows, int delay) {
  const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  const size_t row0 = (size_t)wave * 64;
  if (row0 >= (size_t)rows) return;
  constexpr int LQ = L / 4;                       // float4 per row
  float acc = 0.f;
  if (V == 0 || V == 2) {
    // chunk = 16 steps = 4 float4 per row; instruction j covers rows 16 j .. 16 j + 15
    for (int sc = 0; sc < L / 32; ++sc) {
      float4 va[2][4], vb[2][4], vc[2][4];
      if (V == 2) {        // both halves of the 128-B line requested together
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const size_t q = (row0 + 16 * j + (lane >> 2)) * LQ + (2 * sc + h) * 4 + (lane & 3);
            va[h][j] = a[q]; vb[h][j] = b[q]; vc[h][j] = c[q];
          }
      }
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        if (V == 0) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const size_t q = (row0 + 16 * j + (lane >> 2)) * LQ + (2 * sc + h) * 4 + (lane & 3);
            va[h][j] = a[q]; vb[h][j] = b[q]; vc[h][j] = c[q];
          }
        }
        float t = va[h][0].x;
        t = burn(t, delay);
        acc += t * 1e-30f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const size_t q = (row0 + 16 * j + (lane >> 2)) * LQ + (2 * sc + h) * 4 + (lane & 3);
          float4 r;
          r.x = va[h][j].x + vb[h][j].x + vc[h][j].x + acc; r.y = va[h][j].y + vb[h][j].y + vc[h][j].y;
          r.z = va[h][j].z + vb[h][j].z + vc[h][j].z; r.w = va[h][j].w + vb[h][j].w + vc[h][j].w;
          o[q] = r;
        }
      }
    }
  } else if (V == 1) {
    for (int sc = 0; sc < L / 32; ++sc) {
      float4 va[8], vb[8], vc[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const size_t q = (row0 + 8 * j + (lane >> 3)) * LQ + sc * 8 + (lane & 7);
        va[j] = a[q]; vb[j] = b[q]; vc[j] = c[q];
      }
      float t = va[0].x;
      t = burn(t, 2 * delay);
      acc += t * 1e-30f;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const size_t q = (row0 + 8 * j + (lane >> 3)) * LQ + sc * 8 + (lane & 7);
        float4 r;
        r.x = va[j].x + vb[j].x + vc[j].x + acc; r.y = va[j].y + vb[j].y + vc[j].y;
        r.z = va[j].z + vb[j].z + vc[j].z; r.w = va[j].w + vb[j].w + vc[j].w;
        o[q] = r;
      }
    }
  } else {
    // whole rows: instruction j covers rows 2 j, 2 j + 1; processed 16 rows at a time
    for (int g = 0; g < 4; ++g) {
      float4 va[8], vb[8], vc[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const size_t q = (row0 + 16 * g + 2 * j + (lane >> 5)) * LQ + (lane & 31);
        va[j] = a[q]; vb[j] = b[q]; vc[j] = c[q];
      }
      float t = va[0].x;
      t = burn(t, 2 * delay);
      acc += t * 1e-30f;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const size_t q = (row0 + 16 * g + 2 * j + (lane >> 5)) * LQ + (lane & 31);
        float4 r;
        r.x = va[j].x + vb[j].x + vc[j].x + acc; r.y = va[j].y + vb[j].y + vc[j].y;
        r.z = va[j].z + vb[j].z + vc[j].z; r.w = va[j].w + vb[j].w + vc[j].w;
        o[q] = r;
      }
    }
  }
}

__global__ __launch_bounds__(256) void linear(const float4* __restrict__ a, const float4* __restrict__ b,
                                              const float4* __restrict__ c, float4* __restrict__ o, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const float4 x = a[i], y = b[i], z = c[i];
    float4 r; r.x = x.x + y.x + z.x; r.y = x.y + y.y + z.y; r.z = x.z + y.z + z.z; r.w = x.w + y.w + z.w;
    o[i] = r;
  }
}

int main(int argc, char** argv) {
  const int rows = 256 * 768;
  const size_t n = (size_t)rows * L, bytes = n * 4;
  float *a, *b, *c, *o;
  (void)hipMalloc(&a, bytes); (void)hipMalloc(&b, bytes); (void)hipMalloc(&c, bytes); (void)hipMalloc(&o, bytes);
  std::vector<float> h(n);
  for (size_t i = 0; i < n; ++i) h[i] = (float)(i % 977) * 1e-3f;
  (void)hipMemcpy(a, h.data(), bytes, hipMemcpyHostToDevice);
  (void)hipMemcpy(b, h.data(), bytes, hipMemcpyHostToDevice);
  (void)hipMemcpy(c, h.data(), bytes, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int waves = rows / 64, blocks = (waves + 3) / 4;
  auto time = [&](const char* name, int delay, auto launch) {
    for (int i = 0; i < 3; ++i) launch();
    (void)hipEventRecord(e0);
    const int reps = 20;
    for (int i = 0; i < reps; ++i) launch();
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / reps;
    printf("%-8s delay=%5d  %8.1f us  %7.1f GB/s\n", name, delay, us, 4.0 * bytes / us * 1e-3);
  };
  for (int delay : {0, 400, 1600, 6400}) {
    time("seg64", delay, [&] { probe<0><<<blocks, 256>>>((float4*)a, (float4*)b, (float4*)c, (float4*)o, rows, delay); });
    time("pair64", delay, [&] { probe<2><<<blocks, 256>>>((float4*)a, (float4*)b, (float4*)c, (float4*)o, rows, delay); });
    time("seg128", delay, [&] { probe<1><<<blocks, 256>>>((float4*)a, (float4*)b, (float4*)c, (float4*)o, rows, delay); });
    time("seg512", delay, [&] { probe<3><<<blocks, 256>>>((float4*)a, (float4*)b, (float4*)c, (float4*)o, rows, delay); });
  }
  time("linear", 0, [&] { linear<<<2048, 256>>>((float4*)a, (float4*)b, (float4*)c, (float4*)o, n / 4); });
  return 0;
}
