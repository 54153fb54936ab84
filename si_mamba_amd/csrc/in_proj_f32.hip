// in_proj of the mixer (and out_proj's input gradient) as a hand-written fp32 MFMA kernel for gfx950 -- the fp32 form of
// csrc/in_proj_bf16.hip (reference: self.in_proj of the Mamba mixer, reached from models/block.py:72).
//
//   xz[b, j, t] = sum_c W[j, c] * x[b, t, c]        x: (batch, L, C) fp32, token-major;  W: (M, C) fp32;
//                                                    xz: (batch, M, L) fp32, L contiguous
//
// v_mfma_f32_32x32x2_f32: exact fp32 (a k-ordered fmaf chain), 64 cycles per instruction and SIMD -- the product is bound
// by the matrix pipe (491 us at the 157 TF/s peak for the model shape's 77 GFLOP), not by its 450 MB of traffic, so the
// kernel's job is to keep one MFMA per 64 cycles issuing on every SIMD and hide everything else in the 56 free issue
// cycles between them:
//
//   * a 256-thread workgroup owns 128 tokens of one sample and ALL M output channels; wave w owns tokens 32 w .. 32 w + 31
//     and keeps their B fragments -- the whole K = C <= 384 -- in registers for the life of the kernel (192 registers).
//     Which k an MFMA's two slots contract is free as long as A and B agree: a lane (li, hh) loads 4 consecutive floats
//     at k = 8 g + 4 hh for both operands, and the m-th MFMA of group g contracts k = 8 g + m and 8 g + 4 + m;
//   * the output channels are walked in blocks of 32: the W block (32 x C floats, 48 KB) is shared by the four waves through
//     LDS (rows pitched 16 bytes past their length: the A reads -- 32 rows, 16 bytes each -- are conflict-free and a k
//     group is an immediate offset), block j + 1 arrives in three register batches during block j;
//   * a finished block waits in the second accumulator set, is parked channel-major in LDS ([32 channels][128 tokens]) between
//     the MFMAs of the next block and stored with 16 bytes per lane -- 512 contiguous bytes per channel row -- at the end of
//     the block after that.  One barrier per block (192 MFMAs per wave).
//
// MEASURED SLOWER than the tuned library GEMM and therefore NOT on the default fp32 route (_lib.in_proj_hand_enabled): 646-655 us
// against 534-537 us at (64, 1024, 384 -> 1536) -- 118 against 144 TF/s.  The exact-fp32 MFMA issues on the vector port, so the
// ~500 other vector instructions per block (hipcc keeps the B fragments in AGPRs and copies four of them in front of every
// group of MFMAs; parking, addressing) are not hidden behind the matrix work as they are in the bf16 form but added to
// it; the library's kernel is hand-scheduled assembly at 0.92 of the pipe.  Kept as an explicit, parity-tested variant
// (tests/test_gpu_in_proj.py; _lib.hand_in_proj(True) selects it).
//
// The A-fragment LDS reads are inline asm with counted waits for the reason given in csrc/in_proj_bf16.hip (hipcc sinks
// them in front of their MFMAs otherwise); same contract: a read's destination is only used through ipf_lds_wait.
#include <type_traits>

#include "common.h"

namespace simamba {

typedef float ipf_f32x16 __attribute__((ext_vector_type(16)));
typedef float ipf_f4 __attribute__((ext_vector_type(4)));        // native vectors: see csrc/in_proj_bf16.hip (no uint4)

constexpr int kIpfThreads = 256;
constexpr int kIpfTok = 128;         // tokens per workgroup
constexpr int kIpfCb = 32;           // output channels per block
constexpr int kIpfOPitch = kIpfTok * 4 + 16;     // bytes per channel row of the output staging

template <int I, int N, typename F>
__device__ __forceinline__ void ipf_static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    ipf_static_for<I + 1, N>(f);
  }
}
template <int OFF>
__device__ __forceinline__ void ipf_lds_read16(ipf_f4& dst, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF) : "memory");
}
template <int N>
__device__ __forceinline__ void ipf_lds_wait(ipf_f4& v) {
  asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(v) : "n"(N) : "memory");
}

struct IpfArgs {
  const float* x;        // (batch, L, C)
  const float* w;        // (M, C)
  float* xz;             // (batch, M, L)
  int batch, L, C, M;
};

// NG: k groups of 8 (C = 8 NG)
template <int NG>
__global__ __launch_bounds__(kIpfThreads, 1) void in_proj_f32_kernel(IpfArgs p) {
  constexpr int C = 8 * NG;
  constexpr int kRowB = C * 4;                               // bytes per W row
  constexpr int kChunks = kRowB / 16;                        // 16-byte chunks per W row
  constexpr int kWPitch = kRowB + 16;
  constexpr int kWBuf = kIpfCb * kWPitch;
  constexpr int kWBlk = kIpfCb * kRowB;                      // bytes per W block in memory
  constexpr int kLd = kWBlk / 16 / kIpfThreads;              // 16-byte pieces per thread and W block (NG / 4)
  constexpr int kBatch = 4;                                  // pieces per register batch
  constexpr int kNB = (kLd + kBatch - 1) / kBatch;           // batches per block
  constexpr int kOBuf = kIpfCb * kIpfOPitch;
  constexpr int kNO = kIpfCb * kIpfTok * 4 / 16 / kIpfThreads;   // 16-byte store chunks per thread and block (4)
  constexpr int kAhead = 4;
  static_assert(NG % 8 == 0 && kLd * 4 == NG, "C % 64 == 0");
  __shared__ __attribute__((aligned(16))) unsigned char sW[2][kWBuf];
  __shared__ __attribute__((aligned(16))) unsigned char sO[2][kOBuf];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, hh = lane >> 5;
  const int L = p.L, M = p.M;
  const int tps = (L + kIpfTok - 1) / kIpfTok;
  const int b = static_cast<int>(blockIdx.x) / tps;
  const int t0 = (static_cast<int>(blockIdx.x) - b * tps) * kIpfTok;
  const int nblk = M / kIpfCb;

  // ---- B fragments: group g: x[b][t0 + 32 wave + li][8 g + 4 hh .. + 3]; tokens past L read token L - 1, never stored
  ipf_f4 bq[NG];
  {
    const int t = t0 + 32 * wave + li;
    const float* row = p.x + (static_cast<size_t>(b) * L + (t < L ? t : L - 1)) * C + 4 * hh;
#pragma unroll
    for (int g = 0; g < NG; ++g) bq[g] = *reinterpret_cast<const ipf_f4*>(row + 8 * g);
  }

  // ---- W block staging: piece i = chunk q = tid + 256 i of the block (row q / kChunks, chunk q % kChunks)
  unsigned wsrc[kLd], wdst[kLd];
#pragma unroll
  for (int i = 0; i < kLd; ++i) {
    const int q = tid + kIpfThreads * i;
    const int r = q / kChunks, c = q - r * kChunks;
    wsrc[i] = static_cast<unsigned>(r) * kRowB + 16u * c;
    wdst[i] = static_cast<unsigned>(r) * kWPitch + 16u * c;
  }
  const unsigned char* wbase = reinterpret_cast<const unsigned char*>(p.w);
  const unsigned aaddr = static_cast<unsigned>(reinterpret_cast<uintptr_t>(&sW[0][0])) +
                         static_cast<unsigned>(li) * kWPitch + 16u * hh;

  // ---- output: register 4 gg + e of an accumulator = channel 8 gg + 4 hh + e, token 32 wave + li
  const unsigned ooff = static_cast<unsigned>(4 * hh) * kIpfOPitch + (32u * wave + li) * 4u;
  float* const obase = p.xz + static_cast<size_t>(b) * M * L;
  unsigned oaddr[kNO];
  int orow[kNO], otok[kNO];
#pragma unroll
  for (int i = 0; i < kNO; ++i) {
    const int q = tid + kIpfThreads * i;
    orow[i] = q / (kIpfTok / 4);
    otok[i] = t0 + 4 * (q % (kIpfTok / 4));
    oaddr[i] = static_cast<unsigned>(reinterpret_cast<uintptr_t>(&sO[0][0])) +
               static_cast<unsigned>(orow[i]) * kIpfOPitch + 16u * (q % (kIpfTok / 4));
  }
  auto out_store = [&](ipf_f4& v, int blk, int i) __attribute__((always_inline)) {
    if (otok[i] < L)                                         // L % 4 == 0: a chunk is all in or all out
      *reinterpret_cast<ipf_f4*>(obase + static_cast<size_t>(kIpfCb * blk + orow[i]) * L + otok[i]) = v;
  };
  auto park = [&](const ipf_f32x16& acc, int buf, int i) __attribute__((always_inline)) {
    const int gg = i >> 2, e = i & 3;
    *reinterpret_cast<float*>(&sO[buf][ooff + static_cast<unsigned>(8 * gg + e) * kIpfOPitch]) = acc[i];
  };

  // prologue: W block 0 into buffer 0
  {
    ipf_f4 v[kLd];
#pragma unroll
    for (int i = 0; i < kLd; ++i) v[i] = *reinterpret_cast<const ipf_f4*>(wbase + wsrc[i]);
#pragma unroll
    for (int i = 0; i < kLd; ++i) *reinterpret_cast<ipf_f4*>(&sW[0][wdst[i]]) = v[i];
  }
  __syncthreads();

  ipf_f32x16 acc2[2];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc2[1][i] = 0.f;

  // Iteration blk (parity P): MFMAs of block blk from sW[P]; between them: W block blk + 1 in kNB register batches (loaded
  // at group kG0 * n, written to sW[P ^ 1] kLag groups later -- that buffer was last read in block blk - 1), the parking of
  // block blk - 1 (accumulators acc2[P ^ 1]) into sO[P ^ 1], and at the end the stores of block blk - 2 from sO[P].
  auto block = [&](int blk, auto par_tag) __attribute__((always_inline)) {
    constexpr int P = decltype(par_tag)::value;
    ipf_f32x16& acc = acc2[P];
    const ipf_f32x16& prev = acc2[P ^ 1];
    const bool so = blk > 1;
    const bool more = blk + 1 < nblk;
    const unsigned char* wnext = wbase + static_cast<size_t>(more ? blk + 1 : blk) * kWBlk;   // (re-read, unused, at the end)
    constexpr int kG0 = NG / kNB;                            // groups between batches
    constexpr int kLag = kG0 - 2 > 0 ? kG0 - 2 : 1;
    ipf_f4 o[kNO];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    ipf_f4 a[NG];
    ipf_f4 wv[kBatch];
    ipf_static_for<0, kAhead>([&](auto gc) {
      constexpr int G = decltype(gc)::value;
      ipf_lds_read16<P * kWBuf + 32 * G>(a[G], aaddr);
    });
    ipf_static_for<0, NG>([&](auto gc) {
      constexpr int G = decltype(gc)::value;
      constexpr int kInFlight = (G + kAhead < NG ? kAhead : NG - G) - 1;
      ipf_lds_wait<kInFlight>(a[G]);
      // (one accumulation chain: two chains over even / odd k groups, summed at the end, measured the same)
#pragma unroll
      for (int m = 0; m < 4; ++m) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[G][m], bq[G][m], acc, 0, 0, 0);
      if constexpr (G + kAhead < NG) ipf_lds_read16<P * kWBuf + 32 * (G + kAhead)>(a[G + kAhead], aaddr);
      // W block blk + 1: batch n = pieces kBatch n .. of it
      if constexpr (G % kG0 == 0 && G / kG0 < kNB) {
        constexpr int n = G / kG0;
#pragma unroll
        for (int j = 0; j < kBatch; ++j)
          if (kBatch * n + j < kLd) wv[j] = *reinterpret_cast<const ipf_f4*>(wnext + wsrc[kBatch * n + j]);
      }
      if constexpr (G % kG0 == kLag && G / kG0 < kNB) {
        constexpr int n = G / kG0;
#pragma unroll
        for (int j = 0; j < kBatch; ++j)
          if (kBatch * n + j < kLd) *reinterpret_cast<ipf_f4*>(&sW[P ^ 1][wdst[kBatch * n + j]]) = wv[j];
      }
      // the accumulator registers of block blk - 1, one (or, for a short K, a few) per group
      if constexpr (G >= 1) {
        constexpr int kPer = (16 + NG - 2) / (NG - 1);
        if (blk > 0) {
#pragma unroll
          for (int j = 0; j < kPer; ++j)
            if (kPer * (G - 1) + j < 16) park(prev, P ^ 1, kPer * (G - 1) + j);
        }
      }
      // the parked block blk - 2: read two groups before it is stored -- an asm read's destination must not live long:
      // held from the top of the block, two of these four registers were moved aside (copied before their data had
      // landed) in one of the two unrolled parities, and a quarter of that block's rows went out as garbage
      if constexpr (G == NG - 3) {
        if (so) {
#pragma unroll
          for (int i = 0; i < kNO; ++i) ipf_lds_read16<P * kOBuf>(o[i], oaddr[i]);
        }
      }
      if constexpr (G == NG - 1) {
        if (so) {
#pragma unroll
          for (int i = 0; i < kNO; ++i) {
            ipf_lds_wait<0>(o[i]);
            out_store(o[i], blk - 2, i);
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    });
    __syncthreads();
  };

  for (int blk = 0; blk < nblk; blk += 2) {
    block(blk, std::integral_constant<int, 0>{});
    if (blk + 1 < nblk) block(blk + 1, std::integral_constant<int, 1>{});
  }
  // tail: the last block is still in registers, the one before it parked but not stored
  const int last = nblk - 1;
  auto store_parked = [&](int blk, auto buf_tag) __attribute__((always_inline)) {
    constexpr int buf = decltype(buf_tag)::value;
    ipf_f4 o[kNO];
#pragma unroll
    for (int i = 0; i < kNO; ++i) ipf_lds_read16<buf * kOBuf>(o[i], oaddr[i]);
#pragma unroll
    for (int i = 0; i < kNO; ++i) {
      ipf_lds_wait<0>(o[i]);
      out_store(o[i], blk, i);
    }
  };
  if (last > 0) {
    if ((last - 1) & 1) store_parked(last - 1, std::integral_constant<int, 1>{});
    else store_parked(last - 1, std::integral_constant<int, 0>{});
  }
  if (last & 1) {
#pragma unroll
    for (int i = 0; i < 16; ++i) park(acc2[1], 1, i);
  } else {
#pragma unroll
    for (int i = 0; i < 16; ++i) park(acc2[0], 0, i);
  }
  __syncthreads();
  if (last & 1) store_parked(last, std::integral_constant<int, 1>{});
  else store_parked(last, std::integral_constant<int, 0>{});
}

template <int NG>
static void ipf_launch(const IpfArgs& a, hipStream_t s) {
  const int tps = (a.L + kIpfTok - 1) / kIpfTok;
  hipLaunchKernelGGL((in_proj_f32_kernel<NG>), dim3(static_cast<unsigned>(a.batch) * tps), dim3(kIpfThreads), 0, s, a);
}

int in_proj_f32_dispatch(const void* x, const void* w, void* xz, int batch, int L, int C, int M, hipStream_t s) {
  IpfArgs a{static_cast<const float*>(x), static_cast<const float*>(w), static_cast<float*>(xz), batch, L, C, M};
  switch (C / 64) {
    case 1: ipf_launch<8>(a, s); break;
    case 2: ipf_launch<16>(a, s); break;
    case 3: ipf_launch<24>(a, s); break;
    case 4: ipf_launch<32>(a, s); break;
    case 5: ipf_launch<40>(a, s); break;
    default: ipf_launch<48>(a, s); break;
  }
  return static_cast<int>(hipGetLastError());
}

}  // namespace simamba
