// Fused x_proj -> dt_proj of the Mamba mixer on the gfx950 matrix cores (fp32 in, fp32 accumulate, exact fp32:
// v_mfma_f32_32x32x2_f32 computes a k-ordered fmaf chain).
//
// Replaces, inside upstream's mamba_inner_fn (the fast path the reference's mixer takes from models/block.py:72),
//     x_dbl = x_proj.weight (S, D) @ x                       S = dt_rank + 2 * d_state = 56, D = d_inner = 768
//     delta = dt_proj.weight (D, R) @ x_dbl[:R]              R = dt_rank = 24
// with x = the conv output (batch, D, L), L contiguous.  Two library GEMMs become one pass over x:
//   * x_dbl leaves token-major, (batch, L, S): B_t | C_t sit where the scan kernels read them (state stride 1);
//   * delta leaves (batch, D, L), L contiguous, the layout the scan streams.
// A 256-thread workgroup walks 64-token tiles of one sample each (all workgroups resident: 2 per CU).
//   phase 1  Y[s, t] = sum_d Wx[s, d] x[d, t]: K = D in steps of 32 through double-buffered LDS tiles; wave w owns the
//            32 x 32 block (s-block w & 1, t-block w >> 1).  A-operand lane map of the 32x32x2 MFMA is
//            A[i = lane & 31][k = lane >> 5]: which two d's an MFMA contracts is free as long as A and B agree, so
//            lanes < 32 take d = 0..15 of the step and lanes >= 32 take d = 16..31: a lane's 16 A values are 64
//            contiguous bytes of a Wx row (4 ds_read_b128; 36-float pitch: conflict-free), its 16 B values one
//            column of the x tile (8 ds_read2st64_b32, lanes on consecutive t).
//   phase 2  delta[d, t] = sum_r Wdt[d, r] Y[r, t], K = R = 24 = 12 MFMAs per 32 x 32 block, run for the PREVIOUS tile
//            inside phase 1 of the next one (see Schedule below).  The dt rows of Y wait in LDS (6 KB); the A operand,
//            Wdt[d][8g + 4h .. + 3], is three 16-byte loads per lane and d-block straight from global memory (74 KB
//            table, L2 resident), issued one unit ahead.  The accumulator's C/D layout (col = lane & 31, row =
//            (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)) puts a store instruction on two rows x 128 contiguous bytes.
// Algorithmic bytes per token: read 4 D, write 4 D + 4 S (6.4 KB; with the conv fused another 4 D written); flops
// 2 S D + 2 D R = 123 K: close to the ridge of fp32 MFMA (157 TF/s) against HBM.
#include "common.h"
#include <type_traits>

namespace simamba {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kXdtThreads = 256;
constexpr int kTok = 64;          // tokens per workgroup
constexpr int kKS = 32;           // K (= d) per LDS step
constexpr int kWP = kKS + 4;      // pitch of the Wx tile (floats)
constexpr int kSPad = 64;         // S padded to two 32-row blocks

struct XdtArgs {
  const float* x;        // (batch, D, L); with kConv: the conv INPUT (x half of the in_proj output)
  const float* cw;       // kConv: (D, 4) depthwise taps
  const float* cb;       // kConv: (D) bias or NULL
  float* xconv;          // kConv: (batch, D, L) silu(conv(x)) written as a by-product
  const float* wx;       // (S, D)
  const float* wdt;      // (D, R)
  float* xdbl;           // (batch, L, S)
  float* delta;          // (batch, D, L)
  int batch, D, L, S, R;
  long long x_bs;        // batch stride of x (elements)
};

// kConv: the causal depthwise conv1d (width 4) + SiLU that precedes x_proj in the mixer runs while the x tile is
// staged -- same arithmetic as conv1d_fwd_kernel (fmaf chain over the taps, x * sigmoid(x)) -- and its result is both
// the GEMM operand and, stored once, the conv output the scan and the backward read: the separate conv launch and its
// re-read of the activation disappear.  A thread's 4-step pack needs the 3 steps before it: the previous pack of the
// row sits in the neighbouring lane (DPP row_shr:1; a tile row is exactly one 16-lane DPP row) except for the row's
// first pack, which reads the 16 bytes in front of the tile (zeros at t = 0).  Taps and bias sit in LDS.
//
// Schedule.  The first form of this kernel ran load -> LDS -> 16 MFMAs -> barrier per step and phase 2 after phase 1:
// 51 % matrix-pipe busy, phase 1 + phase 2 = 72 + 50 us with nothing of the one under the other (a dependent chain of
// these MFMAs alone reaches 89 % of the pipe: tools/mfma_probe.hip, so the chain is not the limit -- what sits between
// the bursts is).  Now every wave overlaps its own work, four steps deep (a step = 32 d of one tile; g counts steps
// across the tiles a workgroup walks):
//     iteration g:   16 MFMAs of step g                          (operand registers read in iteration g - 1)
//                    one 32 x 32 block of delta of the PREVIOUS tile every other iteration (12 MFMAs, 16 row stores)
//                    conv + LDS store of step g + 2              (global loads issued in iteration g - 2 -> buffer g & 1)
//                    issue the global loads of step g + 4
//                    ds_read the operands of step g + 1          (tile buffer (g + 1) & 1 -> the registers just used)
//                    one barrier
// so the x reads, the x_conv / delta writes and the matrix pipe run for the whole kernel instead of taking turns; only
// the last tile's delta blocks are left for a tail.  The body of an iteration is ONE basic block: every global access
// is a buffer instruction whose out-of-range lanes (tile edge, rows past S or R, steps past the last tile: the
// descriptor of such a step has zero records) load zeros / store nothing, so nothing in it branches, hipcc's scheduler
// lays the VALU and memory work between the MFMAs, and its s_waitcnt bookkeeping comes out counted (vmcnt(N), the
// loads of the two steps ahead stay in flight) where predicated loads made it drain to vmcnt(0) at every use.
// Measured at (64, 768, 1024) by elimination (tools/xdt_probe.hip): with the MFMAs removed the kernel
// takes 101 us (plain) / 167 us (conv) of its 118 / 180 us -- it is bound by how many bytes 8 waves per CU keep in
// flight around a per-step barrier, not by the matrix pipe or the instruction order (sched_group_barrier / iglp_opt
// interleaves: no change).
constexpr int kMaxDConv = 1024;   // taps + bias of the fused conv live in LDS (20 KB)
constexpr unsigned kOob = 0xfffff000u;   // a byte offset past every descriptor's range

using rsrc_t = __amdgpu_buffer_rsrc_t;

__device__ __forceinline__ rsrc_t make_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, static_cast<int>(bytes), 0x00020000);
}
// The 128-bit builtins' own vector type, reached by bit_cast only: initialising an ext_vector_type(4) from the
// builtin's result compiles (hipcc, ROCm 7.2) to a ONE-dword load splatted over the four lanes.
using bvec4_t = decltype(__builtin_amdgcn_raw_buffer_load_b128(make_rsrc(nullptr, 0u), 0u, 0u, 0));
__device__ __forceinline__ float4 bload4(rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
__device__ __forceinline__ void bstore4(float4 f, rsrc_t r, unsigned voff, unsigned soff) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(bvec4_t, f), r, voff, soff, 0);
}
__device__ __forceinline__ void bstore1(float f, rsrc_t r, unsigned voff, unsigned soff) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, f), r, voff, soff, 0);
}

// A step of the walk: tile `j` of this workgroup (sample q, tile r of the sample), step ks of it.  Advancing never divides.
struct Cursor {
  int j, ks, q, r;
};

template <bool kConv>
__global__ __launch_bounds__(kXdtThreads, 2) void xdt_proj_f32_kernel(XdtArgs p) {
  __shared__ __attribute__((aligned(16))) float sX[2][kKS * kTok];       // [d][t]
  __shared__ __attribute__((aligned(16))) float sW[2][kSPad * kWP];      // [s][d], padded pitch
  __shared__ __attribute__((aligned(16))) float sDt[24 * kTok];          // dt rows of the tile just finished [r][t]
  __shared__ __attribute__((aligned(16))) float sTap[kConv ? kMaxDConv * 4 : 4];
  __shared__ float sBias[kConv ? kMaxDConv : 1];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, hh = lane >> 5;
  const int sblk = wave & 1, tblk = wave >> 1;
  const int D = p.D, L = p.L, S = p.S, R = p.R;
  const int tps = __builtin_amdgcn_readfirstlane((L + kTok - 1) / kTok);   // tiles per sample
  const int ntiles = p.batch * tps;
  const int nwg = static_cast<int>(gridDim.x), wg = static_cast<int>(blockIdx.x);
  // this workgroup's tiles: wg, wg + nwg, ... (the host evens the counts out and keeps all workgroups resident)
  const int ntw = __builtin_amdgcn_readfirstlane((ntiles - wg + nwg - 1) / nwg);
  if (ntw <= 0) return;
  const int nk = D / kKS;                                 // steps per tile; even (host: D % 64 == 0)
  const int ndb = D / 32;                                 // 32-channel blocks of delta; wave w owns w, w + 4, ...
  const int nunits = wave < ndb ? 2 * ((ndb - wave + 3) / 4) : 0;   // (d-block, token block) units of this wave per tile
  const int gq = __builtin_amdgcn_readfirstlane(nwg / tps), gr = nwg - gq * tps;          // the tile stride nwg as (samples, tiles of a sample)
  const unsigned sample_bytes = static_cast<unsigned>(D) * static_cast<unsigned>(L) * 4u;   // host: < 2^32

  for (int i = tid; i < 24 * kTok; i += kXdtThreads) sDt[i] = 0.f;   // rows >= R stay zero
  if (kConv) {
    for (int d = tid; d < D; d += kXdtThreads) {
      *reinterpret_cast<float4*>(&sTap[4 * d]) = *reinterpret_cast<const float4*>(p.cw + 4 * static_cast<size_t>(d));
      sBias[d] = p.cb ? p.cb[d] : 0.f;
    }
    __syncthreads();
  }

  auto advance = [&](Cursor& c, int n) {                   // n <= nk steps forward
    c.ks += n;
    const bool wrap = c.ks >= nk;
    c.ks -= wrap ? nk : 0;
    c.j += wrap ? 1 : 0;
    c.r += wrap ? gr : 0;
    c.q += wrap ? gq : 0;
    const bool carry = c.r >= tps;
    c.r -= carry ? tps : 0;
    c.q += carry ? 1 : 0;
  };

  // ---- staging identities ---------------------------------------------------------------------------------------
  // x tile: 32 d x 64 t = 512 float4; thread -> (d = tid >> 4 (+16), t4 = 4 (tid & 15)); byte offsets inside a sample
  const int xd = tid >> 4, xt = 4 * (tid & 15);
  const bool first = (tid & 15) == 0;                     // first pack of its tile row
  unsigned xoff[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) xoff[i] = (static_cast<unsigned>(xd + 16 * i) * L + xt) * 4u;
  // Wx tile: 64 s x 32 d = 512 float4; thread -> (s = tid >> 3 (+32), d4 = 4 (tid & 7)); rows >= S are out of range
  const int ws = tid >> 3, wd = 4 * (tid & 7);
  unsigned woff[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) woff[i] = (ws + 32 * i < S) ? (static_cast<unsigned>(ws + 32 * i) * D + wd) * 4u : kOob;
  const rsrc_t rs_wx = make_rsrc(p.wx, static_cast<unsigned>(S) * D * 4u);
  const rsrc_t rs_wdt = make_rsrc(p.wdt, static_cast<unsigned>(D) * R * 4u);

  struct Stage { float4 rx[2], rw[2], rh[2]; };
  struct Ops { float a[16], b[16]; };

  // global loads of a step (zeros past the tile's end, past row S, and for a step past the last tile)
  // Cursor fields enter descriptors and scalar offsets.  In the loop they live in SGPRs anyway; in the prologue hipcc
  // had them in VGPRs, could not prove them uniform and wrapped those buffer loads in waterfall loops -- whose loads
  // its s_waitcnt bookkeeping counts once although the counter sees every trip: the first conv read its neighbours'
  // packs (DPP) before they had landed, on some workgroups, on some runs (tests/test_gpu_xdt_proj.py caught it on
  // the bf16 form; the fp32 form had the same loops).  readfirstlane makes the uniformity explicit: no waterfall.
  auto uniform = [](const Cursor& c) {                    // the same values, provably wave-uniform (SGPRs)
    return Cursor{__builtin_amdgcn_readfirstlane(c.j), __builtin_amdgcn_readfirstlane(c.ks),
                  __builtin_amdgcn_readfirstlane(c.q), __builtin_amdgcn_readfirstlane(c.r)};
  };
  auto issue = [&](Stage& st, const Cursor& c_) {
    const Cursor c = uniform(c_);
    const rsrc_t rs = make_rsrc(p.x + static_cast<size_t>(c.q) * p.x_bs, c.j < ntw ? sample_bytes : 0u);
    const int t0 = c.r * kTok;
    const unsigned soff = (static_cast<unsigned>(c.ks) * kKS * L + t0) * 4u;
    const bool xok = t0 + xt < L;                          // L % 4 == 0: a pack is all in or all out
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      st.rx[i] = bload4(rs, xok ? xoff[i] : kOob, soff);
      // the pack in front of the tile, for the row's first lane only (t0 = 0: nothing in front -- zeros)
      // (the 16 bytes back go into the scalar offset: t0 > 0 makes soff >= 256, while xoff - 16 would wrap for d = 0)
      if (kConv) st.rh[i] = bload4(rs, (first && t0 > 0) ? xoff[i] : kOob, soff - 16u);
      st.rw[i] = bload4(rs_wx, woff[i], static_cast<unsigned>(c.ks) * kKS * 4u);
    }
  };
  // conv + SiLU (kConv), x_conv out, LDS tiles of the step
  auto stage = [&](const Stage& st, const Cursor& c_, int buf) {
    const Cursor c = uniform(c_);
    rsrc_t rs_xc;
    unsigned soff = 0;
    bool xok = true;
    if (kConv) {
      rs_xc = make_rsrc(p.xconv + static_cast<size_t>(c.q) * D * L, c.j < ntw ? sample_bytes : 0u);
      const int t0 = c.r * kTok;
      soff = (static_cast<unsigned>(c.ks) * kKS * L + t0) * 4u;
      xok = t0 + xt < L;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      float4 v = st.rx[i];
      if (kConv) {
        const int d = c.ks * kKS + xd + 16 * i;
        // previous pack of the row: the neighbouring lane's current pack (row's first lane: the halo load).
        // The three DPP reads run with EVERY lane active and are pinned in front of the select: a DPP read of a lane
        // that is masked off returns the `old` operand, and hipcc, left alone, predicates the false arm of a ternary.
        float q1 = dpp<DPP_ROW_SHR + 1>(0.f, v.y);
        float q2 = dpp<DPP_ROW_SHR + 1>(0.f, v.z);
        float q3 = dpp<DPP_ROW_SHR + 1>(0.f, v.w);
        asm volatile("" : "+v"(q1), "+v"(q2), "+v"(q3));
        const float p1 = first ? st.rh[i].y : q1;
        const float p2 = first ? st.rh[i].z : q2;
        const float p3 = first ? st.rh[i].w : q3;
        const float win[7] = {p1, p2, p3, v.x, v.y, v.z, v.w};
        const float4 tp = *reinterpret_cast<const float4*>(&sTap[4 * d]);
        const float w4[4] = {tp.x, tp.y, tp.z, tp.w};
        const float bias = sBias[d];
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float acc = bias;
#pragma unroll
          for (int q = 0; q < 4; ++q) acc = fmaf(w4[q], win[e + q], acc);
          o[e] = acc * sigmoid_f(acc);
        }
        v = make_float4(o[0], o[1], o[2], o[3]);
        bstore4(v, rs_xc, xok ? xoff[i] : kOob, soff);
        if (!xok) v = make_float4(0.f, 0.f, 0.f, 0.f);
      }
      *reinterpret_cast<float4*>(&sX[buf][(xd + 16 * i) * kTok + xt]) = v;
      *reinterpret_cast<float4*>(&sW[buf][(ws + 32 * i) * kWP + wd]) = st.rw[i];
    }
  };
  // operands of a step: A = 64 contiguous bytes of a Wx row per lane, B = one column of the x tile
  auto read_ops = [&](Ops& o, int buf) {
    const float* aw = &sW[buf][(sblk * 32 + li) * kWP + 16 * hh];
    const float* bx = &sX[buf][(16 * hh) * kTok + tblk * 32 + li];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 v = *reinterpret_cast<const float4*>(aw + 4 * q);
      o.a[4 * q] = v.x; o.a[4 * q + 1] = v.y; o.a[4 * q + 2] = v.z; o.a[4 * q + 3] = v.w;
    }
#pragma unroll
    for (int m = 0; m < 16; ++m) o.b[m] = bx[m * kTok];
  };

  // ---- delta = Wdt @ dt of a finished tile, one (d-block, token block) unit at a time ------------------------------
  // B operands: MFMA m contracts r = (m & 3) + 8 (m >> 2) + 4 hh; rows >= R contribute zeros.  A wave keeps the dt
  // rows of both 32-token blocks in registers and takes the d-blocks w, w + 4, ...: every Wdt row is loaded once per
  // workgroup and tile.  The accumulator's C/D layout puts a store instruction on two rows x 128 contiguous bytes.
  // A unit past the wave's last one addresses d-block >= ndb: its Wdt loads and delta stores fall out of range.
  constexpr int kM2 = 12;                                  // R <= 24
  float4 wa[3];
  unsigned wdoff[3];
#pragma unroll
  for (int g = 0; g < 3; ++g) wdoff[g] = (8 * g + 4 * hh < R) ? (static_cast<unsigned>(li) * R + 4 * hh + 8 * g) * 4u : kOob;
  const unsigned dvoff = (static_cast<unsigned>(4 * hh) * L + li) * 4u;   // delta: row 4 hh, column li of a 32 x 32 block
  auto wload = [&](int u_) {
    const int u = __builtin_amdgcn_readfirstlane(u_);                                // the Wdt rows of unit u (and u + 1: same d-block)
    const int db = wave + 4 * (u >> 1);
    const unsigned soff = db < ndb ? static_cast<unsigned>(db) * 32u * R * 4u : 0u;
#pragma unroll
    for (int g = 0; g < 3; ++g) wa[g] = bload4(rs_wdt, db < ndb ? wdoff[g] : kOob, soff);
  };
  // the dt rows of the unit's token block come from LDS (rows >= R: lanes read row 0 and contribute zeros)
  const int dtoff = 4 * hh * kTok + li;
  auto unit_mfma = [&](int u, f32x16& o) {
    const float a2[kM2] = {wa[0].x, wa[0].y, wa[0].z, wa[0].w, wa[1].x, wa[1].y, wa[1].z, wa[1].w,
                           wa[2].x, wa[2].y, wa[2].z, wa[2].w};
    const float* dt = &sDt[dtoff + (u & 1) * 32];
    float b2[kM2];
#pragma unroll
    for (int m = 0; m < kM2; ++m) b2[m] = dt[((m & 3) + 8 * (m >> 2)) * kTok];
#pragma unroll
    for (int i = 0; i < 16; ++i) o[i] = 0.f;
#pragma unroll
    for (int m = 0; m < kM2; ++m) o = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[m], b2[m], o, 0, 0, 0);
  };
  auto unit_store = [&](int u_, const f32x16& o, int q_, int r_) {
    const int u = __builtin_amdgcn_readfirstlane(u_), q = __builtin_amdgcn_readfirstlane(q_),
              r = __builtin_amdgcn_readfirstlane(r_);   // tile (sample q, tile r of it)
    const int db = wave + 4 * (u >> 1);
    const rsrc_t rs = make_rsrc(p.delta + static_cast<size_t>(q) * D * L, db < ndb ? sample_bytes : 0u);
    const int tb0 = r * kTok + (u & 1) * 32;
    const unsigned voff = tb0 + li < L ? dvoff : kOob;
    const unsigned base = (static_cast<unsigned>(db) * 32u * L + tb0) * 4u;
#pragma unroll
    for (int i = 0; i < 16; ++i) bstore1(o[i], rs, voff, base + static_cast<unsigned>((i & 3) + 8 * (i >> 2)) * L * 4u);
  };

  // ---- the pipeline ----------------------------------------------------------------------------------------------
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  Stage s0, s1;
  Ops o0;
  // uniform values kept in SGPRs (integer division runs on the VALU): descriptors built from them are provably
  // wave-uniform, no waterfall loop wraps the prologue's buffer loads and hipcc's vmcnt counts stay exact
  const int q0 = __builtin_amdgcn_readfirstlane(wg / tps);
  Cursor cm{0, 0, q0, wg - q0 * tps};                     // step g            (MFMAs)
  Cursor cr = cm, cs = cm, cl = cm;
  advance(cr, 1);                                          // step g + 1        (operand reads)
  advance(cs, 1); advance(cs, 1);                          // step g + 2        (conv + LDS store)
  {
    // prologue: steps 0 and 1 staged, 2 and 3 in flight, operands of step 0 in registers
    issue(s0, cm);
    issue(s1, cr);
    // every load above has landed before the first conv reads its neighbours' packs through DPP: the one place where a
    // miscounted wait (hipcc's waterfall loops) would go unnoticed; paid once per workgroup
    if (kConv) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    stage(s0, cm, 0);
    issue(s0, cs);
    stage(s1, cr, 1);
    cl = cs; advance(cl, 1);
    issue(s1, cl);
    advance(cl, 1);                                        // step g + 4        (global loads)
    __syncthreads();
    read_ops(o0, 0);
    __syncthreads();                                       // buffer 0 is restaged in iteration 0
  }
  int pq = 0, pr = 0;                                      // the previous tile (delta units)
  // One iteration; kOdd: the odd step of a pair (delta unit ks >> 1 of the previous tile when kUnits).
  auto iteration = [&](Ops& cur, Stage& st, const int buf, auto odd_tag, auto units_tag) {
    constexpr bool kOdd = decltype(odd_tag)::value, kUnits = decltype(units_tag)::value;
    f32x16 o;
    const int u = cm.ks >> 1;
    if (kOdd && kUnits) unit_mfma(u, o);
#pragma unroll
    for (int m = 0; m < 16; ++m) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.a[m], cur.b[m], acc, 0, 0, 0);
    stage(st, cs, buf);
    if (kOdd && kUnits) {
      unit_store(u, o, pq, pr);
      wload(u + 1);
    }
    issue(st, cl);
    read_ops(cur, buf ^ 1);                                // step g + 1, once the MFMAs above have taken step g's
    advance(cm, 1); advance(cr, 1); advance(cs, 1); advance(cl, 1);
    __syncthreads();
  };
  using T = std::true_type;
  using F = std::false_type;
  const bool want_delta = p.delta != nullptr;             // NULL: the scan forms delta itself (csrc/scan_fwd_seq.hip)
  for (int j = 0; j < ntw; ++j) {
    if (j == 0 || !want_delta) {
      for (int k = 0; k < nk; k += 2) {                    // nk is even: a tile starts on an even step
        iteration(o0, s0, 0, F{}, F{});
        iteration(o0, s1, 1, T{}, F{});
      }
    } else {
      wload(0);
      for (int k = 0; k < nk; k += 2) {
        iteration(o0, s0, 0, F{}, T{});
        iteration(o0, s1, 1, T{}, T{});
      }
      for (int u = nk / 2; u < nunits; ++u) {              // ndb % 4 != 0: units past nk / 2 (none at D = 768)
        f32x16 o;
        if (!(u & 1)) wload(u);
        unit_mfma(u, o);
        unit_store(u, o, pq, pr);
      }
      // those units read the previous tile's dt rows, which the waves that have none are about to overwrite below
      if (2 * ((ndb + 3) / 4) > nk / 2) __syncthreads();
    }
    // ---- the tile is complete: x_dbl out (token-major), its dt rows into LDS, accumulator cleared ---------------------
    // accumulator register r holds Y[s = sblk*32 + (r & 3) + 8 (r >> 2) + 4 hh][t = tblk*32 + li]
    {
      const int tile = wg + j * nwg;
      const int b = tile / tps, t0 = (tile - b * tps) * kTok;
      pq = b; pr = tile - b * tps;
      const int t = t0 + tblk * 32 + li;
      float* row = p.xdbl + (static_cast<size_t>(b) * L + t) * S;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int s = sblk * 32 + 8 * g + 4 * hh;
        if (t < L && s < S)                                // S % 4 == 0: four states are all in or all out
          *reinterpret_cast<float4*>(row + s) = make_float4(acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]);
        if (sblk == 0) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int r = 8 * g + 4 * hh + i;
            if (r < R) sDt[r * kTok + tblk * 32 + li] = acc[4 * g + i];
          }
        }
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    }
    __syncthreads();
  }

  // ---- tail: delta of the last tile --------------------------------------------------------------------------------
  for (int u = 0; want_delta && u < nunits; ++u) {
    f32x16 o;
    if (!(u & 1)) wload(u);
    unit_mfma(u, o);
    unit_store(u, o, pq, pr);
  }
}

}  // namespace simamba

using namespace simamba;

// xdt_proj_bf16.hip
int xdt_launch_bf16(const void* x, const float* cw, const float* cb, const void* wx, const void* wdt, void* xconv,
                    void* xdbl, void* delta, int batch, int D, int L, int S, int R, long long x_bs, bool conv,
                    hipStream_t stream);

static int xdt_launch(const void* x, const float* cw, const float* cb, const void* wx_, const void* wdt_, void* xconv,
                      void* xdbl, void* delta, int batch, int D, int L, int S, int R, int io_dtype,
                      long long x_bstride, bool conv, void* stream) {
  if (batch < 0 || D <= 0 || L < 0 || batch > 65535) return SIMAMBA_E_SHAPE;
  if (io_dtype != SIMAMBA_F32 && io_dtype != SIMAMBA_BF16) return SIMAMBA_E_DTYPE;
  const int pack = io_dtype == SIMAMBA_F32 ? 4 : 8;      // elements per 16-byte access
  if (D % 64 || L % pack || S % 4 || S > kSPad || S < R || R % 4 || R > 24 || R < 4) return SIMAMBA_E_SHAPE;
  if (conv && D > kMaxDConv) return SIMAMBA_E_SHAPE;
  if (static_cast<long long>(D) * L * 4 >= (1LL << 32) - 65536 || static_cast<long long>(S) * D * 4 >= (1LL << 31)) return SIMAMBA_E_SHAPE;
  if (batch == 0 || L == 0) return SIMAMBA_OK;
  const float* wx = static_cast<const float*>(wx_);
  const float* wdt = static_cast<const float*>(wdt_);
  if (!x || !wx || !wdt || !xdbl || (conv && (!cw || !xconv))) return SIMAMBA_E_NULLPTR;   // delta may be NULL
  uintptr_t al = reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(wx) | reinterpret_cast<uintptr_t>(wdt) |
                 reinterpret_cast<uintptr_t>(xdbl) | reinterpret_cast<uintptr_t>(delta);
  if (conv) al |= reinterpret_cast<uintptr_t>(cw) | reinterpret_cast<uintptr_t>(xconv);
  if (al & 15u) return SIMAMBA_E_ALIGN;
  XdtArgs a{};
  a.x = static_cast<const float*>(x); a.wx = wx; a.wdt = wdt; a.cw = cw; a.cb = cb;
  a.xconv = static_cast<float*>(xconv);
  a.xdbl = static_cast<float*>(xdbl); a.delta = static_cast<float*>(delta);
  a.batch = batch; a.D = D; a.L = L; a.S = S; a.R = R;
  a.x_bs = x_bstride ? x_bstride : static_cast<long long>(D) * L;
  if (a.x_bs % pack) return SIMAMBA_E_ALIGN;
  if (static_cast<long long>(batch) * ((L + kTok - 1) / kTok) > 0x7fffffffLL) return SIMAMBA_E_SHAPE;
  if (io_dtype == SIMAMBA_BF16)
    return xdt_launch_bf16(x, cw, cb, wx_, wdt_, xconv, xdbl, delta, batch, D, L, S, R, a.x_bs, conv,
                           static_cast<hipStream_t>(stream));
  // grid: every workgroup the same number of tiles, all workgroups resident together (2 per CU, 256 CUs)
  const long long ntiles = static_cast<long long>(batch) * ((L + kTok - 1) / kTok);
  if (ntiles > 0x7fffffffLL) return SIMAMBA_E_SHAPE;
  long long g = ntiles;
  if (ntiles > 512) {
    // tiles per workgroup at full residency, evened out (per divides ntiles) when a divisor is near: a prime tile count
    // must not collapse the grid to a few workgroups -- the kernel takes uneven counts (ntw), evenness is only tidier
    const long long per0 = (ntiles + 511) / 512;
    long long per = per0;
    while (ntiles % per && per < 2 * per0) ++per;
    g = (ntiles % per == 0) ? ntiles / per : 512;
  }
  dim3 grid(static_cast<unsigned>(g));
  if (conv)
    hipLaunchKernelGGL(xdt_proj_f32_kernel<true>, grid, dim3(kXdtThreads), 0, static_cast<hipStream_t>(stream), a);
  else
    hipLaunchKernelGGL(xdt_proj_f32_kernel<false>, grid, dim3(kXdtThreads), 0, static_cast<hipStream_t>(stream), a);
  return static_cast<int>(hipGetLastError());
}

// D % 64 == 0, L % 4 == 0 (bf16: 8), S % 4 == 0, S <= 64, R % 4 == 0, R <= 24; wx / wdt in the I/O type.
extern "C" int simamba_xdt_proj_fwd(const void* x, const void* wx, const void* wdt, void* xdbl, void* delta,
                                    int batch, int D, int L, int S, int R, int io_dtype, long long x_bstride,
                                    void* stream) {
  return xdt_launch(x, nullptr, nullptr, wx, wdt, nullptr, xdbl, delta, batch, D, L, S, R, io_dtype, x_bstride, false,
                    stream);
}

// The same with the mixer's causal depthwise conv1d (width 4, + SiLU) applied to x on the way in; xconv receives it.
extern "C" int simamba_conv_xdt_proj_fwd(const void* x, const float* cw, const float* cb, const void* wx,
                                         const void* wdt, void* xconv, void* xdbl, void* delta, int batch, int D, int L,
                                         int S, int R, int io_dtype, long long x_bstride, void* stream) {
  return xdt_launch(x, cw, cb, wx, wdt, xconv, xdbl, delta, batch, D, L, S, R, io_dtype, x_bstride, true, stream);
}
