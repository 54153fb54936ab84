// Prices the operand feed of the lane-per-channel scan recurrence on gfx950 (diagnostic tool).
// Every variant runs the same arithmetic per (channel, step, state):  a = exp2(dl * A2); h = a h + (dl u) B; y += h C
// and differs only in how B_t[n] / C_t[n] (shared by all channels of a sample) reach the VALU:
//   0  wave-uniform scalars (SGPR operands): the floor, no feed cost at all
//   1  one ds_read_b128 each + DPP quad_perm broadcast inside v_mul / v_fmac (round-1 kernel)
//   2  LDS broadcast reads, one lane per channel: 8 ds_read_b128 per step, plain VALU
//   3  LDS broadcast reads, lane = 2 channels x 8 states: 4 ds_read_b128 per step serve 2 channels
//   4  LDS broadcast reads, lane = 1 channel x 8 states (two lanes per channel): 4 ds_read_b128 per step
// dl / u arrive by ds_read_b128 per 4 steps and y leaves by ds_write_b128 per 4 steps in variants 1-4, as in the
// kernel.  Output: ns per (state, step) wave-instruction group per SIMD at 2, 3, 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr float kLog2e = 1.4426950408889634f;
constexpr int kPitch = 36;

template <int P> __device__ __forceinline__ float mul_q(float s, float x);
template <int P> __device__ __forceinline__ void fmac_q(float& acc, float s, float x);
#define QUAD_OPS(P, PERM)                                                                                       \
  template <> __device__ __forceinline__ float mul_q<P>(float s, float x) {                                     \
    float r;                                                                                                    \
    asm("v_mul_f32_dpp %0, %1, %2 quad_perm:" PERM " row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(s), "v"(x));  \
    return r;                                                                                                   \
  }                                                                                                             \
  template <> __device__ __forceinline__ void fmac_q<P>(float& acc, float s, float x) {                        \
    asm("v_fmac_f32_dpp %0, %1, %2 quad_perm:" PERM " row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(s), "v"(x)); \
  }
QUAD_OPS(0, "[0,0,0,0]")
QUAD_OPS(1, "[1,1,1,1]")
QUAD_OPS(2, "[2,2,2,2]")
QUAD_OPS(3, "[3,3,3,3]")
#undef QUAD_OPS

struct Uni { float b[16], c[16]; };

template <int MODE>
__global__ __launch_bounds__(256, MODE == 3 ? 2 : (MODE == 2 ? 3 : 4)) void probe(float* out, Uni uni, int steps) {
  constexpr int CH = MODE == 3 ? 2 : 1;                 // channels per lane
  constexpr int NS = (MODE == 3 || MODE == 4) ? 8 : 16; // states per lane and channel
  __shared__ __attribute__((aligned(16))) float sBC[4][16 * kPitch];
  __shared__ __attribute__((aligned(16))) float sD[4][96 * 16];      // [row][16 steps] delta (64 rows) | u / y (32 rows)
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float* bc = sBC[wave];
  float* td = sD[wave];
  for (int i = lane; i < 16 * kPitch; i += 64) bc[i] = 0.5f + 1e-3f * i;
  for (int i = lane; i < 96 * 16; i += 64) td[i] = 0.01f + 1e-4f * (i & 63);
  __builtin_amdgcn_wave_barrier();
  float h[CH][NS], A2[CH][NS];
#pragma unroll
  for (int c = 0; c < CH; ++c)
#pragma unroll
    for (int n = 0; n < NS; ++n) { h[c][n] = 0.f; A2[c][n] = -(n + 1 + 0.01f * lane) * kLog2e; }
  const int half = (MODE == 3 || MODE == 4) ? (lane & 1) : 0;
  float ysum = 0.f;
  for (int g = 0; g < steps / 4; ++g) {
    float4 d4[CH], u4[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int row = MODE == 0 ? 0 : ((MODE == 3 ? 2 * (lane >> 1) + c : (MODE == 4 ? lane >> 1 : lane)));
      if (MODE == 0) { d4[c] = make_float4(0.011f, 0.012f, 0.013f, 0.014f); u4[c] = make_float4(1.f, 0.9f, 1.1f, 0.8f); }
      else {
        d4[c] = *reinterpret_cast<const float4*>(td + row * 16 + 4 * (g & 3));
        u4[c] = *reinterpret_cast<const float4*>(td + 64 * 16 + (row & 31) * 16 + 4 * (g & 3));
      }
    }
    float yy[CH][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int t = (4 * g + i) & 15;
      float vB[NS], vC[NS];
      float4 b4, c4;
      if (MODE == 1) {
        b4 = *reinterpret_cast<const float4*>(bc + t * kPitch + 4 * (lane & 3));
        c4 = *reinterpret_cast<const float4*>(bc + t * kPitch + 16 + 4 * (lane & 3));
      } else if (MODE >= 2) {
#pragma unroll
        for (int q = 0; q < NS / 4; ++q) {
          const float4 x = *reinterpret_cast<const float4*>(bc + t * kPitch + NS * half + 4 * q);
          const float4 y = *reinterpret_cast<const float4*>(bc + t * kPitch + 16 + NS * half + 4 * q);
          vB[4 * q] = x.x; vB[4 * q + 1] = x.y; vB[4 * q + 2] = x.z; vB[4 * q + 3] = x.w;
          vC[4 * q] = y.x; vC[4 * q + 1] = y.y; vC[4 * q + 2] = y.z; vC[4 * q + 3] = y.w;
        }
      }
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        const float dl = i == 0 ? d4[c].x : i == 1 ? d4[c].y : i == 2 ? d4[c].z : d4[c].w;
        const float uu = i == 0 ? u4[c].x : i == 1 ? u4[c].y : i == 2 ? u4[c].z : u4[c].w;
        const float xx = dl * uu;
        float ys[2] = {0.9f * uu, 0.f};
#pragma unroll
        for (int n = 0; n < NS; ++n) {
          const float e = __builtin_amdgcn_exp2f(dl * A2[c][n]);
          float& y = ys[n & 1];
          if (MODE == 0) {
            h[c][n] = fmaf(e, h[c][n], xx * uni.b[n & 3]);
            y = fmaf(h[c][n], uni.c[n & 3], y);
          } else if (MODE == 1) {
            const float bj = (n & 3) == 0 ? b4.x : (n & 3) == 1 ? b4.y : (n & 3) == 2 ? b4.z : b4.w;
            const float cj = (n & 3) == 0 ? c4.x : (n & 3) == 1 ? c4.y : (n & 3) == 2 ? c4.z : c4.w;
            float xb;
            if (n < 4) xb = mul_q<0>(bj, xx); else if (n < 8) xb = mul_q<1>(bj, xx);
            else if (n < 12) xb = mul_q<2>(bj, xx); else xb = mul_q<3>(bj, xx);
            h[c][n] = fmaf(e, h[c][n], xb);
            if (n < 4) fmac_q<0>(y, cj, h[c][n]); else if (n < 8) fmac_q<1>(y, cj, h[c][n]);
            else if (n < 12) fmac_q<2>(y, cj, h[c][n]); else fmac_q<3>(y, cj, h[c][n]);
          } else {
            h[c][n] = fmaf(e, h[c][n], xx * vB[n]);
            y = fmaf(h[c][n], vC[n], y);
          }
        }
        float y = ys[0] + ys[1];
        if (MODE == 3 || MODE == 4)
          y += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, y), 0xB1, 0xf, 0xf, false));
        yy[c][i] = y;
        __builtin_amdgcn_sched_barrier(0);    // one (channel, step)'s work stays together (the kernel does the same)
      }
    }
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      if (MODE == 0) ysum += yy[c][0] + yy[c][1] + yy[c][2] + yy[c][3];
      else {
        const int row = MODE == 3 ? 2 * (lane >> 1) + c : (MODE == 4 ? lane >> 1 : lane);
        *reinterpret_cast<float4*>(td + 64 * 16 + (row & 31) * 16 + 4 * (g & 3)) =
            make_float4(yy[c][0], yy[c][1], yy[c][2], yy[c][3]);
      }
    }
  }
  float s = ysum;
#pragma unroll
  for (int c = 0; c < CH; ++c)
#pragma unroll
    for (int n = 0; n < NS; ++n) s += h[c][n];
  out[blockIdx.x * 256 + threadIdx.x] = s + td[lane];
}

template <int MODE>
void run(const char* name) {
  constexpr int CH = MODE == 3 ? 2 : 1;
  constexpr int NS = (MODE == 3 || MODE == 4) ? 8 : 16;
  const int steps = 4096;
  float* out;
  (void)hipMalloc(&out, 256 * 256 * 8 * sizeof(float));
  Uni uni;
  for (int i = 0; i < 16; ++i) { uni.b[i] = 0.5f + 0.01f * i; uni.c[i] = 0.7f - 0.01f * i; }
  printf("%-34s", name);
  for (int wps : {2, 3, 4}) {
    if ((MODE == 2 || MODE == 3) && wps == 4) continue;
    const int blocks = 256 * wps;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    probe<MODE><<<blocks, 256>>>(out, uni, 64);
    (void)hipEventRecord(e0);
    probe<MODE><<<blocks, 256>>>(out, uni, steps);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    // per SIMD: wps waves x steps x CH*NS (state, step) groups
    printf("  wps=%d: %6.2f ns/group (%7.1f us)", wps, ms * 1e6 / (double(steps) * CH * NS * wps), ms * 1e3);
  }
  printf("\n");
}

int main() {
  run<0>("0 scalar operands (floor)");
  run<1>("1 ds_read x2 + DPP quad");
  run<2>("2 LDS broadcast x8, 1 lane/ch");
  run<3>("3 LDS broadcast x4, 2ch x 8st/lane");
  run<4>("4 LDS broadcast x4, 2 lanes/ch");
  return 0;
}
