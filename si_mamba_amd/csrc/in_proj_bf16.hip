// in_proj of the mixer as a hand-written bf16 MFMA kernel for gfx950 (reference: self.in_proj of the Mamba
// mixer, the first product of upstream's forward, reached from models/block.py:72).
//
//   xz[b, j, t] = sum_c W[j, c] * x[b, t, c]        x: (batch, L, C) bf16, token-major (the LayerNorm output)
//                                                    W: (M, C) bf16 (M = 2 d_inner),  xz: (batch, M, L) bf16, L contiguous
//
// The output is what the conv / scan kernels stream along time, so it leaves channel-major -- 4x the bytes of the
// input: 201 MB written against 50 MB read at the model shape, an HBM-bound product (37 us of stores at 5.5 TB/s, 31 us
// of MFMA at the bf16 peak).  Shape of the kernel:
//
//   * a 256-thread workgroup owns 256 tokens of one sample and ALL M output channels; wave w owns tokens 64 w .. 64 w + 63
//     and keeps their B fragments -- the whole K = C <= 384 of them -- in registers for the life of the kernel
//     (v_mfma_f32_32x32x16_bf16, two 32-token column sets per wave: set u, column li <-> token 64 w + 2 li + u, so that the
//     two accumulators of a lane hold two ADJACENT tokens of a channel and pack into one dword);
//   * the output channels are walked in blocks of 32: the W block (32 x C, 24 KB) is shared by the four waves through LDS
//     (register-staged two blocks ahead; rows pitched 16 bytes past their length, so that the A-fragment reads -- 32
//     rows, one 16-byte chunk each -- are conflict-free and a k step is an immediate offset), every workgroup streams W (1.2 MB) once from L2 for its 256 tokens;
//   * a finished block is rounded to bf16, parked channel-major in LDS ([32 channels][256 tokens]) and stored with 16
//     bytes per lane: 512 contiguous bytes per channel row.  Block j is parked between the MFMAs of block j + 1 and stored
//     in front of those of block j + 2; one barrier per block.
//
// One wave per SIMD (the B fragments alone are 192 registers), 256 workgroups at the model shape: one per CU, one round.
#include <type_traits>

#include "common.h"

namespace simamba {

typedef float ip_f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 ip_bf16x8 __attribute__((ext_vector_type(8)));
// 16-byte register values as a NATIVE vector: HIP's ip_u4 is a struct whose copies hipcc lowers to memcpy between address
// spaces, and an array of them that lives across loop iterations then stays in scratch memory instead of registers
typedef unsigned ip_u4 __attribute__((ext_vector_type(4)));

// Timing-only A/B switches (tools/build_alt.sh; the elimination numbers of DESIGN 4.10): -DIP_NOSTORE drops the global stores,
// -DIP_NOWLOAD reloads W block 0 for every block, -DIP_NOMFMA issues one MFMA per block.  Outputs are wrong by construction.
constexpr int kIpThreads = 256;
constexpr int kIpCb = 32;            // output channels per block
#ifndef SIMAMBA_INPROJ_SETS
#define SIMAMBA_INPROJ_SETS 2        // 32-token column sets per wave: 2 -> 256 tokens per workgroup, one per CU, 192 registers
#endif                               // of B fragments; 1 -> 128 tokens, two workgroups per CU, twice the LDS reads per flop
                                     // (LDS-bound: 114-124 us where the two-set form takes 102-104, library GEMM 116)
constexpr int kIpSets = SIMAMBA_INPROJ_SETS;
constexpr int kIpTok = 128 * kIpSets;            // tokens per workgroup
constexpr int kIpOPitch = kIpTok * 2 + 16;       // bytes per channel row of the output staging (+16: rows shift banks)

template <int I, int N, typename F>
__device__ __forceinline__ void ip_static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    ip_static_for<I + 1, N>(f);
  }
}
// A 16-byte LDS read the compiler can neither sink next to its use nor count: issued here, waited for by ip_lds_wait.
// (Left to itself hipcc moved every A-fragment read directly in front of its MFMAs and waited lgkmcnt(0) each time --
// one LDS latency per k step, 2.5 us per channel block at one wave per SIMD.)
// Contract: the destination may only be READ through ip_lds_wait (which hands the compiler a new value): to hipcc the
// register is defined the moment this statement issues.  A copy of it made before the wait -- a live-range split or a
// spill into an AGPR under register pressure -- would copy stale contents; the current builds make none (checked in the
// .s: every ds_read_b128 destination is next touched by the s_waitcnt statement or the MFMA behind it), and
// tests/test_gpu_in_proj.py compares every output element on every build.
template <int OFF>
__device__ __forceinline__ void ip_lds_read16(ip_u4& dst, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF) : "memory");
}
// LDS operations of a wave return in order: at most N of them still outstanding means every read older than the N youngest
// has landed.  `v` ties the wait to the value about to be used.
template <int N>
__device__ __forceinline__ void ip_lds_wait(ip_u4& v) {
  asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(v) : "n"(N) : "memory");
}

struct IpArgs {
  const uint16_t* x;     // (batch, L, C)
  const uint16_t* w;     // (M, C)
  uint16_t* xz;          // (batch, M, L)
  int batch, L, C, M;
};

// NS: 16-deep k steps (C = 16 NS)
template <int NS>
__global__ __launch_bounds__(kIpThreads, kIpSets == 1 ? 2 : 1) void in_proj_bf16_kernel(IpArgs p) {
  constexpr int C = 16 * NS;
  constexpr int kRowB = C * 2;                               // bytes per W row
  constexpr int kChunks = kRowB / 16;                        // 16-byte chunks per W row (2 NS)
  constexpr int kWBlk = kIpCb * kRowB;                       // bytes per W block in memory
  constexpr int kWPitch = kRowB + 16;                        // LDS row pitch: rows 4 banks apart, so the 16 rows a read
                                                             // pass covers are conflict-free and a step is an immediate
  constexpr int kWBuf = kIpCb * kWPitch;                     // bytes per LDS buffer
  constexpr int kLd = kWBlk / 16 / kIpThreads;               // 16-byte loads per thread and W block (NS / 4)
  static_assert(NS % 4 == 0, "whole 16-byte loads per thread");
  __shared__ __attribute__((aligned(16))) unsigned char sW[2][kWBuf];
  __shared__ __attribute__((aligned(16))) unsigned char sO[2][kIpCb * kIpOPitch];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, hh = lane >> 5;
  const int L = p.L, M = p.M;
  const int tps = (L + kIpTok - 1) / kIpTok;
  const int b = static_cast<int>(blockIdx.x) / tps;
  const int t0 = (static_cast<int>(blockIdx.x) - b * tps) * kIpTok;
  const int nblk = M / kIpCb;

  // ---- B fragments: set u, k step s: x[b][t0 + 64 wave + 2 li + u][16 s + 8 hh .. + 7]; tokens past L read token L - 1
  // and are never stored ---------------------------------------------------------------------------------------------
  ip_u4 bf[kIpSets][NS];
#pragma unroll
  for (int u = 0; u < kIpSets; ++u) {
    const int t = t0 + 32 * kIpSets * wave + kIpSets * li + u;
    const uint16_t* row = p.x + (static_cast<size_t>(b) * L + (t < L ? t : L - 1)) * C + 8 * hh;
#pragma unroll
    for (int s = 0; s < NS; ++s) bf[u][s] = *reinterpret_cast<const ip_u4*>(row + 16 * s);
  }

  // ---- W block staging: chunk q = tid + 256 i of the block (row q / kChunks, chunk q % kChunks) -> LDS row-major with the
  // rows pitched kWPitch apart ------------------------------------------------------------------------------------
  unsigned wsrc[kLd], wdst[kLd];
#pragma unroll
  for (int i = 0; i < kLd; ++i) {
    const int q = tid + kIpThreads * i;
    const int r = q / kChunks, c = q - r * kChunks;
    wsrc[i] = static_cast<unsigned>(r) * kRowB + 16u * c;
    wdst[i] = static_cast<unsigned>(r) * kWPitch + 16u * c;
  }
  struct WS { ip_u4 v[kLd]; };
  const unsigned char* wbase = reinterpret_cast<const unsigned char*>(p.w);
  auto load_w = [&](WS& st, int blk_) __attribute__((always_inline)) {                      // unconditional: a block past the end re-reads the last one
#ifdef IP_NOWLOAD
    const int blk = 0 * blk_;
#else
    const int blk = blk_ < nblk ? blk_ : nblk - 1;
#endif
    const unsigned char* src = wbase + static_cast<size_t>(blk) * kWBlk;
#pragma unroll
    for (int i = 0; i < kLd; ++i) st.v[i] = *reinterpret_cast<const ip_u4*>(src + wsrc[i]);
  };
  auto store_w = [&](const WS& st, int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < kLd; ++i) *reinterpret_cast<ip_u4*>(&sW[buf][wdst[i]]) = st.v[i];
  };
  // A fragment of step s: row li, chunk 2 s + hh: a per-lane LDS address + the immediate 32 s (+ the buffer)
  const unsigned aaddr = static_cast<unsigned>(reinterpret_cast<uintptr_t>(&sW[0][0])) +
                         static_cast<unsigned>(li) * kWPitch + 16u * hh;

  // ---- output staging: register 4 g + e of acc[u] = channel 8 g + 4 hh + e, token 64 wave + 2 li + u -> one dword per
  // channel row; store-out: chunk q = tid + 256 i of the [32][256 tokens] block, 16 bytes = 8 tokens ------------------
  const unsigned ooff = static_cast<unsigned>(4 * hh) * kIpOPitch + (32u * kIpSets * wave + kIpSets * li) * 2u;
  uint16_t* const obase = p.xz + static_cast<size_t>(b) * M * L;

  // store-out of a parked block: chunk q = tid + 256 i of the [32 channels][tokens] image; the LDS reads are issued first
  // (ip_lds_read16: asm, so the value must pass through ip_lds_wait before it is used), the stores some MFMAs later
  constexpr int kNO = 2 * kIpSets;
  constexpr int kOBuf = kIpCb * kIpOPitch;
  unsigned oaddr[kNO];
  int orow[kNO], otok[kNO];
#pragma unroll
  for (int i = 0; i < kNO; ++i) {
    const int q = tid + kIpThreads * i;
    orow[i] = q / (16 * kIpSets);
    otok[i] = t0 + 8 * (q % (16 * kIpSets));
    oaddr[i] = static_cast<unsigned>(reinterpret_cast<uintptr_t>(&sO[0][0])) +
               static_cast<unsigned>(orow[i]) * kIpOPitch + 16u * (q % (16 * kIpSets));
  }
  auto out_read = [&](ip_u4 (&o)[kNO], auto buf_tag) __attribute__((always_inline)) {
    constexpr int buf = decltype(buf_tag)::value;
#pragma unroll
    for (int i = 0; i < kNO; ++i) ip_lds_read16<buf * kOBuf>(o[i], oaddr[i]);
  };
  auto out_store = [&](ip_u4& v, int blk, int i) __attribute__((always_inline)) {
#ifdef IP_NOSTORE
    if (otok[i] < L && v.x == 0x12345678u)
#else
    if (otok[i] < L)                                         // L % 8 == 0: a chunk is all in or all out
#endif
      *reinterpret_cast<ip_u4*>(obase + static_cast<size_t>(kIpCb * blk + orow[i]) * L + otok[i]) = v;
  };
  auto store_out = [&](int blk, auto buf_tag) __attribute__((always_inline)) {   // the unpipelined form (tail)
    ip_u4 o[kNO];
    out_read(o, buf_tag);
#pragma unroll
    for (int i = 0; i < kNO; ++i) {
      ip_lds_wait<0>(o[i]);
      out_store(o[i], blk, i);
    }
  };

  // park a finished block (rounded to bf16) channel-major: one (g, e) register pair = one dword per lane
  // (one set per wave: the neighbouring token sits in the neighbouring lane -- a quad_perm DPP read, even lanes write)
  auto park = [&](const ip_f32x16 (&acc)[kIpSets], int buf, int i) __attribute__((always_inline)) {
    const int g = i >> 2, e = i & 3;
    const float lo = acc[0][4 * g + e];
    float hi;
    if constexpr (kIpSets == 2) hi = acc[kIpSets - 1][4 * g + e];
    else hi = dpp<0xb1>(lo, lo);                               // quad_perm [1, 0, 3, 2]: lane ^ 1
    const unsigned v = static_cast<unsigned>(f32_to_bf16(lo)) | (static_cast<unsigned>(f32_to_bf16(hi)) << 16);
    if (kIpSets == 2 || (li & 1) == 0)
      *reinterpret_cast<unsigned*>(&sO[buf][ooff + static_cast<unsigned>(8 * g + e) * kIpOPitch]) = v;
  };

  // One wave per SIMD: nothing else hides a latency, so everything is issued ahead in program order.  Iteration blk:
  //   stores of block blk - 2 (parked during iteration blk - 1, visible since its barrier);
  //   MFMAs of block blk with the A fragments read kAhead steps ahead, and between them the parking of block blk - 1
  //   (its accumulators stay live in the other register set);
  //   W block blk + 1 (in registers since iteration blk - 2) into the other LDS buffer, request of block blk + 3; barrier.
  constexpr int kAhead = kIpSets == 1 ? 4 : 6;
  // (the two register sets -- accumulators and W stages -- are indexed by a compile-time parity: handed around as
  // references they ended up behind a run-time pointer, i.e. in scratch memory)
  WS st[2];
  ip_f32x16 acc2[2][kIpSets];
  auto block = [&](int blk, auto par_tag) __attribute__((always_inline)) {
    constexpr int P = decltype(par_tag)::value;              // = blk & 1
    ip_f32x16 (&acc)[kIpSets] = acc2[P];
    const ip_f32x16 (&prev)[kIpSets] = acc2[P ^ 1];
    WS& stg = st[P ^ 1];                                     // holds W block blk + 1
    constexpr int buf = P;
    constexpr bool kSpread = NS >= kLd + kNO + 6;            // room to hand the side work out one piece per step
    const bool so = blk > 1;
    ip_u4 o[kNO];
#pragma unroll
    for (int u = 0; u < kIpSets; ++u)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[u][i] = 0.f;
    ip_u4 a[NS];
    ip_static_for<0, (kAhead < NS ? kAhead : NS)>([&](auto sc) {
      constexpr int S = decltype(sc)::value;
      ip_lds_read16<buf * kWBuf + 32 * S>(a[S], aaddr);
    });
    if (!kSpread && so) store_out(blk - 2, par_tag);         // block blk - 2, parked in sO[buf] during block blk - 1
    ip_static_for<0, NS>([&](auto sc) {
      constexpr int S = decltype(sc)::value;
      // reads S + 1 .. min(S + kAhead, NS) - 1 may still be in flight (plus whatever the compiler has issued since: the
      // count is then merely stricter)
      constexpr int kInFlight = (S + kAhead < NS ? kAhead : NS - S) - 1;
      ip_lds_wait<kInFlight>(a[S]);
#ifdef IP_NOMFMA
      if (S == 0)
#endif
#pragma unroll
      for (int u = 0; u < kIpSets; ++u)
        acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(ip_bf16x8, a[S]),
                                                         __builtin_bit_cast(ip_bf16x8, bf[u][S]), acc[u], 0, 0, 0);
      if constexpr (S + kAhead < NS) ip_lds_read16<buf * kWBuf + 32 * (S + kAhead)>(a[S + kAhead], aaddr);
      if (blk > 0) {                                           // 16 dwords of the previous block over the first steps
        constexpr int kPer = (16 + NS - 1) / NS;
#pragma unroll
        for (int j = 0; j < kPer; ++j)
          if (kPer * S + j < 16) park(prev, buf ^ 1, kPer * S + j);
      }
      if constexpr (kSpread) {
        // towards the end W block blk + 1 into the other LDS buffer (last read in block blk - 1), one 16-byte
        // piece per step, then the request for block blk + 3 into the same registers
        constexpr int kW0 = NS - 2 - kLd;
        if constexpr (S >= kW0 && S < kW0 + kLd)
          *reinterpret_cast<ip_u4*>(&sW[buf ^ 1][wdst[S - kW0]]) = stg.v[S - kW0];
        if constexpr (S == kW0 + kLd) load_w(stg, blk + 3);
        // The stores of block blk - 2 go LAST: hipcc cannot count VMEM operations across the loop's back edge and waits
        // for all but the youngest few before it touches the W registers above -- with this block's stores already in
        // flight that wait sat on their retirement (~1-2 us under a chip-wide write stream), every block.  Issued here
        // they are a whole block old when the next wait comes.
        // (read two steps before the stores: the destination of an asm read must not live long enough to be moved
        // aside before its data has landed -- the fp32 form did exactly that with reads held from the top of the block)
        if constexpr (S == NS - 3) {
          if (so) out_read(o, par_tag);                      // block blk - 2, parked in sO[buf] during block blk - 1
        }
        if constexpr (S == NS - 1) {
          if (so) {
#pragma unroll
            for (int i = 0; i < kNO; ++i) {
              ip_lds_wait<0>(o[i]);
              out_store(o[i], blk - 2, i);
            }
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    });
    if (!kSpread) {
      store_w(stg, buf ^ 1);
      load_w(stg, blk + 3);
    }
    __syncthreads();
  };

  load_w(st[0], 0);
  load_w(st[1], 1);
  store_w(st[0], 0);
  load_w(st[0], 2);
  __syncthreads();
#pragma unroll
  for (int u = 0; u < kIpSets; ++u)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc2[1][u][i] = 0.f;
  // even blocks compute into acc2[0] and carry W block blk + 1 in st[1], odd ones the other way round
  for (int blk = 0; blk < nblk; blk += 2) {
    block(blk, std::integral_constant<int, 0>{});
    if (blk + 1 < nblk) block(blk + 1, std::integral_constant<int, 1>{});
  }
  // tail: the last block is still in registers, the one before it parked but not stored
  const int last = nblk - 1;
  if (last > 0) {
    if ((last - 1) & 1) store_out(last - 1, std::integral_constant<int, 1>{});
    else store_out(last - 1, std::integral_constant<int, 0>{});
  }
  if (last & 1) {
#pragma unroll
    for (int i = 0; i < 16; ++i) park(acc2[1], 1, i);
  } else {
#pragma unroll
    for (int i = 0; i < 16; ++i) park(acc2[0], 0, i);
  }
  __syncthreads();
  if (last & 1) store_out(last, std::integral_constant<int, 1>{});
  else store_out(last, std::integral_constant<int, 0>{});
}

template <int NS>
static void ip_launch(const IpArgs& a, hipStream_t s) {
  const int tps = (a.L + kIpTok - 1) / kIpTok;
  hipLaunchKernelGGL((in_proj_bf16_kernel<NS>), dim3(static_cast<unsigned>(a.batch) * tps), dim3(kIpThreads), 0, s, a);
}

int in_proj_f32_dispatch(const void* x, const void* w, void* xz, int batch, int L, int C, int M, hipStream_t s);   // in_proj_f32.hip

}  // namespace simamba

using namespace simamba;

extern "C" int simamba_in_proj_fwd(const void* x, const void* w, void* xz, int batch, int L, int C, int M, int io_dtype,
                                   void* stream) {
  if (io_dtype != SIMAMBA_F32 && io_dtype != SIMAMBA_BF16) return SIMAMBA_E_DTYPE;
  if (batch < 0 || L < 0 || C <= 0 || M <= 0) return SIMAMBA_E_SHAPE;
  // whole 64-deep groups of k steps (one 16-byte W load per thread and 64 k), 32-channel blocks, 16-byte store chunks
  if (C % 64 || C > 384 || M % kIpCb || L % (io_dtype == SIMAMBA_F32 ? 4 : 8)) return SIMAMBA_E_SHAPE;
  if (batch == 0 || L == 0) return SIMAMBA_OK;
  if (!x || !w || !xz) return SIMAMBA_E_NULLPTR;
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w) | reinterpret_cast<uintptr_t>(xz)) & 15u)
    return SIMAMBA_E_ALIGN;
  if (static_cast<long long>(batch) * ((L + 127) / 128) > 0x7fffffffLL) return SIMAMBA_E_SHAPE;
  if (io_dtype == SIMAMBA_F32)
    return in_proj_f32_dispatch(x, w, xz, batch, L, C, M, static_cast<hipStream_t>(stream));
  IpArgs a{static_cast<const uint16_t*>(x), static_cast<const uint16_t*>(w), static_cast<uint16_t*>(xz), batch, L, C, M};
  hipStream_t s = static_cast<hipStream_t>(stream);
  switch (C / 64) {
    case 1: ip_launch<4>(a, s); break;
    case 2: ip_launch<8>(a, s); break;
    case 3: ip_launch<12>(a, s); break;
    case 4: ip_launch<16>(a, s); break;
    case 5: ip_launch<20>(a, s); break;
    default: ip_launch<24>(a, s); break;
  }
  return static_cast<int>(hipGetLastError());
}
