// Instruction-throughput probe for gfx950 (diagnostic tool): cycles per wave-instruction per SIMD for the
// VALU ops the scan kernels are made of, at 1/2/4/8 waves per SIMD.  Output: table on stdout.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP16(X) X X X X X X X X X X X X X X X X
typedef float float2_ __attribute__((ext_vector_type(2)));

template <int OP>
__global__ void probe(float* out, int iters, unsigned long long* cyc) {
  float a[16], q[16], e[16], y = 0.f;
  float sb = __builtin_amdgcn_readfirstlane(iters) * 1e-3f, sc = sb + 1.f;
  float2_ p[16];
  for (int i = 0; i < 16; ++i) { a[i] = threadIdx.x * 1e-3f + i; p[i] = float2_{a[i], a[i] + 1.f}; q[i] = a[i] * 0.5f; e[i] = 0.f; }
  float m = 1.0001f, c = 1e-6f;
  float2_ pm = {1.0001f, 0.9999f}, pc = {1e-6f, 2e-6f};
  __syncthreads();
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
      if (OP == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pm), "v"(pc));
      if (OP == 2) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
      if (OP == 3) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
      if (OP == 4) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pm));
      if (OP == 5) asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
      if (OP == 6) asm volatile("v_fmac_f32_dpp %0, %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(m), "v"(c));
      if (OP == 7) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
      if (OP == 8) asm volatile("v_log_f32 %0, %0" : "+v"(a[i]));
      if (OP == 9) {     // 1 exp + 3 independent fma: serial (sum) or overlapped (max)?
        asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(p[i].x) : "v"(m), "v"(c));
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(p[i].y) : "v"(m), "v"(c));
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(q[i]) : "v"(m), "v"(c));
      }
      if (OP == 10) {    // the one-lane-per-channel scan group: mul, exp, mul(sgpr), fmac, fmac(sgpr)
        float t, xb;
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(t) : "v"(m), "v"(q[i]));
        asm volatile("v_exp_f32 %0, %0" : "+v"(t));
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(xb) : "s"(sb), "v"(c));
        asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(xb) : "v"(t), "v"(a[i]));
        a[i] = xb;
        asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(y) : "s"(sc), "v"(xb));
      }
      if (OP == 11) {    // same, 4-byte encodings only, exp results consumed 16 groups later
        float t;
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(t) : "v"(m), "v"(q[i]));
        asm volatile("v_exp_f32 %0, %0" : "+v"(t));
        e[i] = t;
      }
    }
    if (OP == 11) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        float xb;
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(xb) : "s"(sb), "v"(c));
        asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(xb) : "v"(e[i]), "v"(a[i]));
        a[i] = xb;
        asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(y) : "s"(sc), "v"(xb));
      }
    }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += a[i] + p[i].x + p[i].y + q[i] + e[i] + y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int OP>
void run(const char* name) {
  const int iters = 2000;
  float* out; unsigned long long* cyc;
  (void)hipMalloc(&out, 256 * 2048 * sizeof(float));
  (void)hipMalloc(&cyc, 2048 * sizeof(unsigned long long));
  printf("%-18s", name);
  for (int wps : {1, 2, 3, 4, 8}) {            // waves per SIMD: block = 256 threads (1 wave / SIMD), wps blocks per CU
    int blocks = 256 * wps;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    probe<OP><<<blocks, 256>>>(out, 10, cyc);
    (void)hipEventRecord(e0);
    probe<OP><<<blocks, 256>>>(out, iters, cyc);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks);
    (void)hipMemcpy(h.data(), cyc, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double avg = 0; for (auto v : h) avg += v; avg /= blocks;
    // per SIMD: wps waves each issuing iters*16 instructions
    double ns_per_inst = ms * 1e6 / (double(iters) * 16 * wps);
    printf("  wps=%d: %.2f ns/inst/SIMD (%.2f cyc@2.4GHz, ctr %.1f/inst)", wps, ns_per_inst, ns_per_inst * 2.4,
           avg / (double(iters) * 16));
  }
  printf("\n");
}

int main() {
  run<0>("v_fma_f32");
  run<1>("v_pk_fma_f32");
  run<3>("v_mul_f32");
  run<4>("v_pk_mul_f32");
  run<2>("v_exp_f32");
  run<7>("v_rcp_f32");
  run<8>("v_log_f32");
  run<5>("v_mov_b32_dpp");
  run<6>("v_fmac_f32_dpp");
  run<9>("exp+3fma (x4 inst)");
  run<10>("scan group (x5)");
  run<11>("scan group split");
  return 0;
}
