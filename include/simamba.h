/*
 * simamba.h -- C ABI of libsimamba_hip.so, the MI355X (gfx950) implementation of the
 * SI-Mamba hot path: selective scan (fwd+bwd), causal depthwise conv1d (fwd+bwd) and
 * the k-NN graph-Laplacian eigen-ordering.
 *
 * The reference (denix56/SI-Mamba) is pure Python; the native code on this path lives in
 * un-vendored wheels.  Each entry point below names the reference call site whose native
 * callee it replaces (paths relative to the reference repo):
 *
 *   simamba_selective_scan_fwd/bwd   selective_scan_cuda.fwd/bwd of mamba-ssm, reached from
 *                                    models/block.py:72 (self.mixer(...)) with the mixer built
 *                                    at models/point_mamba.py:162.
 *   simamba_causal_conv1d_fwd/bwd    causal_conv1d_cuda.causal_conv1d_fwd/bwd of causal-conv1d,
 *                                    same call site (inside the mixer).
 *   simamba_xdt_proj_fwd             the x_proj / dt_proj GEMM pair of the same mixer (cuBLAS calls inside
 *                                    upstream's mamba_inner_fn).
 *   simamba_add_layer_norm_fwd/bwd   the Add -> LayerNorm of models/block.py:56-60 (torch ops there).
 *   simamba_out_proj_add_ln_fwd      out_proj of the mixer + that Add -> LayerNorm, one kernel (bf16).
 *   simamba_in_proj_fwd              in_proj of the same mixer (a cuBLAS GEMM upstream), fp32 and bf16.
 *   simamba_selective_scan_dt_fwd/bwd  the scan with delta formed in the kernel (same call site, bf16 path).
 *   simamba_knn_graph                models/point_mamba.py:620-661 and :664-715
 *                                    (create_graph_from_centers / ..._feature_space_...).
 *   simamba_laplacian_topk           models/point_mamba.py:717-761 and :764-814
 *                                    (per-sample torch.linalg.eigh loop, cuSOLVER underneath).
 *   simamba_spectral_topk            the two above fused: centres -> top-k eigenpairs + orders.
 *   simamba_argsort_rows             the torch.sort of models/point_mamba.py:820.
 *   simamba_farthest_point_sample    pytorch3d.ops.sample_farthest_points, models/point_mamba.py:93.
 *
 * Conventions
 *   - Plain pointers and sizes only.  All pointers are DEVICE pointers unless noted.
 *   - The caller owns every buffer (incl. workspace); the library allocates nothing and keeps
 *     no mutable global state.  Calls only enqueue work on `stream` (a hipStream_t passed as
 *     void*); no device-wide synchronisation, no default-stream use, no hipSetDevice.
 *   - Return 0 on success; <0 = argument error detected before any launch (SIMAMBA_E_*);
 *     >0 = hipError_t reported by the runtime for the launch.  simamba_strerror() decodes both.
 *   - Tensors are contiguous in the layouts stated per function, except where a stride argument
 *     (in ELEMENTS) is given: z / dz / conv x / conv dx may be batch-strided views (e.g. halves of
 *     the mixer's (batch, 2*dim, seqlen) in_proj output), and B, C may be read through arbitrary
 *     (batch, state, time) strides (e.g. straight out of the (batch, seqlen, R+2N) x_proj output).
 *     A stride of 0 means "contiguous default".
 *   - io_dtype: SIMAMBA_F32 or SIMAMBA_BF16 for activation-sized tensors; all state,
 *     accumulation and parameter gradients are fp32.
 */
#ifndef SIMAMBA_H_
#define SIMAMBA_H_

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SIMAMBA_ABI_VERSION 9

#define SIMAMBA_F32  0
#define SIMAMBA_BF16 1

#define SIMAMBA_OK            0
#define SIMAMBA_E_NULLPTR    -1
#define SIMAMBA_E_SHAPE      -2
#define SIMAMBA_E_DTYPE      -3
#define SIMAMBA_E_DSTATE     -4   /* dstate must be in [1,16] */
#define SIMAMBA_E_WIDTH      -5   /* conv width must be in [2,4] */
#define SIMAMBA_E_WORKSPACE  -6
#define SIMAMBA_E_GROUPS     -7   /* G in [2,128], knn + 1 <= min(G,32), k (+1) <= G, F in [1,64] */
#define SIMAMBA_E_ALIGN      -8
#define SIMAMBA_E_VARIANT    -9   /* unknown forward-scan variant, or one the shape / alignment cannot take */

/* timesteps per scan chunk; simamba_scan_num_chunks(L) = ceil(L / chunk) */
#define SIMAMBA_SCAN_CHUNK 128

/*
 * State checkpoints handed from the forward to the backward (`x_ckpt`, `ckpt_step` of both calls; 0 = _ROW):
 *   SIMAMBA_SCAN_CKPT_ROW  (batch, dim, nchunks, dstate): state at the end of every 128-step chunk; written by every
 *                          forward kernel, read by the row-scan backward (any shape).
 *   SIMAMBA_SCAN_CKPT_SEQ  (batch, ceil(seqlen / 16), dim, 16): state after every 16th step; written by the
 *                          lanes-per-channel forward kernels, read by the sequential backward (dstate == 16,
 *                          dim % 64 == 0, delta_softplus, 16-byte aligned rows and B / C packs, tensors below 2^30
 *                          elements; anything else returns SIMAMBA_E_VARIANT).
 * simamba_scan_ckpt_step() is the library's own choice for a shape (host arithmetic only; the caller still has to
 * meet the alignment rules above or pass _ROW); simamba_scan_ckpt_floats() the size of x_ckpt in floats (0: none
 * needed, x_ckpt may be NULL).
 */
#define SIMAMBA_SCAN_CKPT_ROW 128
#define SIMAMBA_SCAN_CKPT_SEQ 16

int         simamba_abi_version(void);
const char* simamba_strerror(int rc);           /* host string, static storage */
int         simamba_scan_num_chunks(int seqlen);
/* the forward kernel SIMAMBA_SCAN_AUTO picks for (batch, dim) when the operands qualify for all of them (one of the
 * SIMAMBA_SCAN_* values below; host-side arithmetic only, no GPU work): what a profile's kernel name should be */
int         simamba_scan_fwd_auto_variant(int batch, int dim);
int         simamba_scan_ckpt_step(int batch, int dim, int seqlen, int dstate, int io_dtype);
long long   simamba_scan_ckpt_floats(int batch, int dim, int seqlen, int dstate, int ckpt_step);

/*
 * Selective scan forward.
 *   u, delta, z, out : (batch, dim, seqlen)  io_dtype      z may be NULL (no gating)
 *   A                : (dim, dstate) fp32                   (already -exp(A_log))
 *   B, C             : (batch, dstate, seqlen) io_dtype, element (b,n,t) at b*bc_bstride + n*bc_nstride
 *                      + t*bc_tstride (all three 0 => contiguous (batch, dstate, seqlen))
 *   z_bstride        : elements between consecutive batch samples of z (0 => dim*seqlen)
 *   D, delta_bias    : (dim) fp32, may be NULL
 *   x_ckpt, ckpt_step: state checkpoints for the backward (layouts above) or NULL: required by the backward
 *                      when simamba_scan_ckpt_floats() > 0.  With _SEQ the forward runs a lanes-per-channel kernel
 *                      (SIMAMBA_E_VARIANT if the operands cannot take one or `variant` names the row-scan kernel).
 *   last_state       : (batch, dim, dstate) fp32 or NULL.
 *   variant          : SIMAMBA_SCAN_AUTO in production: the library picks the kernel from the shape (no
 *                      environment variables, no global state).  The explicit values select one kernel for
 *                      benchmarks and parity tests: ROWSCAN (16 lanes per row, any shape), LPC2 / LPC4 (2 / 4
 *                      lanes per channel, 32-step chunks; need dstate == 16, 16-byte aligned rows and tensors
 *                      below 2^30 elements, else SIMAMBA_E_VARIANT), MIX (one launch with the first channels of
 *                      every sample on 2 and the last ones on 4 lanes per channel, for shapes whose 2-lane launch
 *                      leaves at most half a round of waves over, e.g. batch * dim = 49 152; else SIMAMBA_E_VARIANT).
 *                      All variants compute the same function.
 *   out = (scan(u, softplus?(delta + delta_bias), A, B, C) + D*u) * silu(z)
 */
#define SIMAMBA_SCAN_AUTO    0
#define SIMAMBA_SCAN_ROWSCAN 1
#define SIMAMBA_SCAN_LPC2    2
#define SIMAMBA_SCAN_LPC4    4
#define SIMAMBA_SCAN_MIX     6
int simamba_selective_scan_fwd(const void* u, const void* delta, const float* A,
                               const void* B, const void* C, const float* D, const void* z,
                               const float* delta_bias, void* out, float* x_ckpt,
                               float* last_state, int batch, int dim, int seqlen, int dstate,
                               int io_dtype, int delta_softplus, long long z_bstride,
                               long long bc_bstride, long long bc_nstride, long long bc_tstride,
                               int ckpt_step, int variant, void* stream);

/*
 * Selective scan backward.  Inputs as forward (+ dout, and x_ckpt / ckpt_step exactly as given to the
 * forward; ckpt_step selects the kernel: _ROW the row-scan backward, _SEQ the sequential one).
 * du, ddelta, dz : io_dtype (dz NULL iff z NULL).
 * dA (dim,dstate), dB, dC (batch,dstate,seqlen), dD, ddelta_bias (dim): fp32; the library
 * zeroes them on `stream` and then accumulates (float atomics: last-bit run-to-run jitter).
 * Not one byte outside these five spans is written: spans that are exactly adjacent in memory
 * (one ends where the next begins) are cleared by a single memset node, all others one by one.
 * dD / ddelta_bias may be NULL when D / delta_bias are NULL.  dB / dC are always contiguous
 * (batch, dstate, seqlen); z, dz and B, C take strides as in the forward.
 */
int simamba_selective_scan_bwd(const void* u, const void* delta, const float* A,
                               const void* B, const void* C, const float* D, const void* z,
                               const float* delta_bias, const void* dout, const float* x_ckpt,
                               void* du, void* ddelta, float* dA, float* dB, float* dC,
                               float* dD, void* dz, float* ddelta_bias,
                               int batch, int dim, int seqlen, int dstate,
                               int io_dtype, int delta_softplus, long long z_bstride,
                               long long dz_bstride, long long bc_bstride, long long bc_nstride,
                               long long bc_tstride, int ckpt_step, void* stream);

/*
 * The mixer's scan with delta formed INSIDE the scan kernels (no (batch, dim, seqlen) delta tensor exists): what
 * upstream's mamba_inner_fn computes between x_proj and out_proj, reached from models/block.py:72.
 *   u      : (batch, dim, seqlen) io_dtype -- the conv output
 *   xdbl   : (batch, seqlen, dt_rank + 2 * 16) io_dtype, token-major -- the x_proj output [dt | B_t | C_t]; element
 *            (b, t, s) at b * xdbl_bstride + t * xdbl_tstride + s (0 => contiguous)
 *   wdt    : (dim, dt_rank) io_dtype -- dt_proj.weight
 *   delta[b, d, t] = sum_r wdt[d, r] * xdbl[b, t, r] on the matrix pipe, with the instruction and k order of
 *   simamba_xdt_proj_fwd: bit for bit the tensor that call would have stored (bf16: rounded to bf16 like it), then
 *   out = (scan(u, softplus(delta + delta_bias), A, B_t, C_t) + D * u) * silu(z) exactly as simamba_selective_scan_fwd.
 * Lanes-per-channel kernels only: dstate == 16, z required, dt_rank % 4 == 0 (bf16: % 8), 4 <= dt_rank <= 24, 16-byte
 * aligned rows; anything else returns SIMAMBA_E_VARIANT / _SHAPE and the caller materialises delta
 * (simamba_xdt_proj_fwd + simamba_selective_scan_fwd).  x_ckpt / ckpt_step / variant / last_state as in
 * simamba_selective_scan_fwd (variant: AUTO, LPC2, LPC4 or MIX).
 * Backward: the sequential kernel (16-step checkpoints of the forward, SIMAMBA_SCAN_CKPT_SEQ; dim % 64 == 0); outputs
 * and accumulators as simamba_selective_scan_bwd, ddelta = gradient w.r.t. the delta formed above.
 */
int simamba_selective_scan_dt_fwd(const void* u, const void* xdbl, const void* wdt, const float* A, const float* D,
                                  const void* z, const float* delta_bias, void* out, float* x_ckpt, float* last_state,
                                  int batch, int dim, int seqlen, int dstate, int dt_rank, int io_dtype,
                                  long long z_bstride, long long xdbl_bstride, long long xdbl_tstride, int ckpt_step,
                                  int variant, void* stream);
int simamba_selective_scan_dt_bwd(const void* u, const void* xdbl, const void* wdt, const float* A, const float* D,
                                  const void* z, const float* delta_bias, const void* dout, const float* x_ckpt,
                                  void* du, void* ddelta, float* dA, float* dB, float* dC, float* dD, void* dz,
                                  float* ddelta_bias, int batch, int dim, int seqlen, int dstate, int dt_rank,
                                  int io_dtype, long long z_bstride, long long dz_bstride, long long xdbl_bstride,
                                  long long xdbl_tstride, void* stream);

/*
 * Fused x_proj -> dt_proj of the mixer on the matrix cores (the two skinny GEMMs between the conv and the scan
 * inside upstream's mamba_inner_fn, reached from models/block.py:72):
 *   xdbl[b, t, s]  = sum_d wx[s, d] * x[b, d, t]            (batch, seqlen, S) token-major, S = dt_rank + 2 * dstate
 *   delta[b, d, t] = sum_r wdt[d, r] * xdbl[b, t, r], r < R (batch, D, seqlen)
 *   x : (batch, D, seqlen), seqlen contiguous, batch stride x_bstride elements (0 => D * seqlen);
 *   wx : (S, D), wdt : (D, R), both in the I/O type.  SIMAMBA_F32: exact fp32 MFMA (v_mfma_f32_32x32x2_f32);
 *   SIMAMBA_BF16: v_mfma_f32_32x32x16_bf16 with fp32 accumulation and the roundings of the reference's autocast
 *   run (x_conv, x_dbl, delta each rounded once; delta formed from the rounded dt rows).
 *   D % 64 == 0, D * seqlen * 4 < 2^32 (one sample is one buffer descriptor), seqlen % 4 == 0 (bf16: % 8), S % 4 == 0, S <= 64,
 *   R % 4 == 0, 4 <= R <= 24, 16-byte aligned pointers; anything else returns SIMAMBA_E_SHAPE / _ALIGN (the host
 *   mirror then takes the two library GEMMs).
 *   delta == NULL: only xdbl (and xconv) are produced -- the delta product is left to simamba_selective_scan_dt_fwd.
 */
int simamba_xdt_proj_fwd(const void* x, const void* wx, const void* wdt, void* xdbl, void* delta,
                         int batch, int D, int seqlen, int S, int R, int io_dtype, long long x_bstride,
                         void* stream);
/* The same with the causal depthwise conv1d (width 4, bias cb or NULL, + SiLU) of the mixer applied to x on the way
 * in (causal_conv1d_fn inside the same mamba_inner_fn): xconv (batch, D, seqlen) receives silu(conv(x)) -- what the
 * scan and the backward read -- so conv, x_proj and dt_proj are one pass over the in_proj output's x half.
 * D <= 1024 (taps and bias are held in LDS). */
int simamba_conv_xdt_proj_fwd(const void* x, const float* cw, const float* cb, const void* wx, const void* wdt,
                              void* xconv, void* xdbl, void* delta, int batch, int D, int seqlen, int S, int R,
                              int io_dtype, long long x_bstride, void* stream);

/*
 * Token-sequence expansion along the last axis and its adjoint.  The reference's block stack sees L = 2 k G tokens
 * that are the same G patch tokens in 2 k orders (models/point_mamba.py:889-898, :982-989); the per-token head of the
 * first block (Add, LayerNorm, in_proj: models/block.py:56-60) is computed on the G distinct tokens and expanded:
 *   fwd: out[b, c, l] = in[b, c, idx[b, l]]              in (batch, channels, G), idx (batch, L) int32,
 *                                                        out (batch, channels, L) with batch stride out_bstride
 *   bwd: din[b, c, g] = sum_j dout[b, c, inv[b, g, j]]   inv (batch, G, R) int32: the R = L / G positions of token g
 *   G <= 256, L <= 2048, G % 4 == 0, L % 4 == 0, R <= 8; idx 16-byte aligned.  Deterministic (no atomics).
 */
int simamba_seq_gather_fwd(const void* in, const int* idx, void* out, int batch, int channels, int G, int L,
                           long long out_bstride, int io_dtype, void* stream);
int simamba_seq_gather_bwd(const void* dout, const int* inv, void* din, int batch, int channels, int G, int L,
                           int R, long long dout_bstride, int io_dtype, void* stream);

/*
 * Causal depthwise conv1d (+ optional SiLU).
 *   x, out, dout, dx : (batch, dim, seqlen) io_dtype
 *   w : (dim, width) fp32; bias : (dim) fp32 or NULL; dw, dbias fp32, zeroed then accumulated.
 *   out[b,d,t] = act(bias[d] + sum_k w[d,k] * x[b,d,t-(width-1)+k])
 *   x_bstride / dx_bstride: elements between batch samples of x / dx (0 => dim*seqlen).
 */
int simamba_causal_conv1d_fwd(const void* x, const float* w, const float* bias, void* out,
                              int batch, int dim, int seqlen, int width, int silu,
                              int io_dtype, long long x_bstride, void* stream);
int simamba_causal_conv1d_bwd(const void* x, const float* w, const float* bias,
                              const void* dout, void* dx, float* dw, float* dbias,
                              int batch, int dim, int seqlen, int width, int silu,
                              int io_dtype, long long x_bstride, long long dx_bstride, void* stream);

/*
 * Fused (DropPath-scaled) residual add + LayerNorm: the "Add -> LayerNorm" half of the reference's
 * Block.forward (models/block.py:56-60) and the final norm of MixerModel (models/point_mamba.py:257-258).
 *   hidden        : (batch, rows_per_batch, dim) hidden_dtype
 *   residual      : same shape fp32, or NULL (first block: residual_out = hidden, not written when
 *                   residual_out is NULL)
 *   rowscale      : (batch) fp32 or NULL -- DropPath keep-mask / keep-prob per sample, applied to hidden
 *                   only when residual != NULL (the reference does not drop the first block's input)
 *   residual_out  : fp32 ; normed : out_dtype ; mean, rstd : (batch * rows_per_batch) fp32
 *   residual_out = hidden * rowscale + residual ;  normed = LayerNorm(residual_out; weight, bias, eps)
 * Backward: dresidual (fp32, gradient w.r.t. `residual`) = LN'(dnormed) + dresidual_out ;
 *   dhidden (hidden_dtype) = dresidual * rowscale ; either may be NULL (not both).
 *   dwb_partial : (simamba_add_layer_norm_grid(batch, rows_per_batch), 2, dim) fp32 -- per-workgroup partial
 *   sums of (dweight, dbias); the caller reduces over the first axis.
 * dim % 4 == 0, dim <= 2048.
 */
int simamba_add_layer_norm_grid(int batch, int rows_per_batch);
int simamba_add_layer_norm_fwd(const void* hidden, const float* residual, const float* rowscale,
                               const float* weight, const float* bias, float* residual_out,
                               void* normed, float* mean, float* rstd, int batch, int rows_per_batch,
                               int dim, float eps, int hidden_dtype, int out_dtype, void* stream);
int simamba_add_layer_norm_bwd(const void* dnormed, const float* dresidual_out,
                               const float* residual_out, const float* mean, const float* rstd,
                               const float* weight, const float* rowscale, float* dresidual,
                               void* dhidden, float* dwb_partial, int batch, int rows_per_batch,
                               int dim, int hidden_dtype, int out_dtype, void* stream);

/*
 * out_proj -> (+ DropPath-scaled residual) -> LayerNorm in one kernel on the matrix cores, bf16 operands: the mixer's
 * out_proj (upstream mamba_inner_fn, models/block.py:72) fused with the Add -> LayerNorm that opens the next block
 * (models/block.py:56-58) or closes the stack (models/point_mamba.py:257-258).  The out_proj result is rounded to bf16
 * (the rounding the reference's autocast GEMM output has) and never written:
 *   y : (batch, K, L) bf16, L contiguous ; w : (C, K) bf16 ; residual : (batch, L, C) fp32 or NULL ;
 *   rowscale : (batch) fp32 or NULL (applied to the out_proj result when residual != NULL) ; gamma, beta : (C) fp32 ;
 *   residual_out : (batch, L, C) fp32 = bf16(y^T w^T) * rowscale + residual ; normed : out_dtype = LayerNorm(residual_out) ;
 *   mean, rstd : (batch * L) fp32 for simamba_add_layer_norm_bwd.
 * C % 128 == 0, C <= 384, K % 64 == 0, L % 8 == 0, K * L * 2 < 2^32, 16-byte aligned pointers.
 */
int simamba_out_proj_add_ln_fwd(const void* y, const void* w, const float* residual, const float* rowscale,
                                const float* gamma, const float* beta, float* residual_out, void* normed, float* mean,
                                float* rstd, int batch, int K, int L, int C, float eps, int out_dtype, void* stream);

/*
 * in_proj of the mixer on the matrix cores (the first product of upstream's Mamba.forward, reached from
 * models/block.py:72):  xz[b, j, t] = sum_c w[j, c] * x[b, t, c].  Also the shape of out_proj's input gradient.
 *   x : (batch, L, C) io_dtype, token-major (the LayerNorm output) ; w : (M, C) io_dtype ; xz : (batch, M, L) io_dtype,
 *   L contiguous (the layout the conv / scan kernels stream).  fp32: exact fp32 MFMA (an fmaf chain); bf16: fp32
 *   accumulation, one rounding to bf16.
 * C % 64 == 0, C <= 384, M % 32 == 0, L % 8 == 0 (fp32: L % 4 == 0), 16-byte aligned pointers.  A workgroup owns 256
 * (fp32: 128) tokens of one sample: grids of fewer than ~200 workgroups leave CUs idle and are better served by the
 * library GEMM.
 */
int simamba_in_proj_fwd(const void* x, const void* w, void* xz, int batch, int L, int C, int M, int io_dtype,
                        void* stream);

/*
 * Patch-encoder streaming ops (reference models/point_mamba.py:46-73, Encoder: Conv1d - BatchNorm1d - ReLU -
 * Conv1d, max over the n points of a patch; the 1x1 convolutions are GEMMs on token-major (rows, C) tensors).
 *
 * simamba_bn_relu_fwd: y = relu(BatchNorm(x + g)) with g[r / group][c] an optional per-group additive term
 * (gterm == NULL: none).  training != 0: batch statistics over all rows (biased variance for the
 * normalisation, unbiased for running_var; running_* updated with `momentum` when non-NULL), written to
 * mean / invstd (C) for the backward; training == 0: running statistics.  weight / bias may be NULL (1 / 0).
 *   x, y : (rows, C) io_dtype with row stride ld elements (0 = C; a channel slice of a wider tensor is
 *   processed in place: C <= 1024 per call, wider layers are done slice by slice) ; gterm : (rows / group, C)
 *   fp32 ; partial : (simamba_bn_relu_grid(rows), 2, C) fp32 scratch.  C % 4 == 0, ld % 4 == 0,
 *   rows % group == 0.
 * simamba_bn_relu_bwd: dx (rows, C) io_dtype (same ld), dgterm (rows / dgroup, C) fp32 or NULL (= sum of dx over
 *   runs of dgroup rows; 256 % dgroup == 0 -- for group > 256 the caller sums group / 256 consecutive rows of
 *   it), dweight, dbias (C) fp32 ; same partial scratch.
 * simamba_group_max_fwd/bwd: out[g][c] = max_r x[g][r][c] over r < n (first maximum; NaN propagates), idx the
 *   arg max as uint8 (n <= 256); backward routes dout to that row and writes zeros elsewhere (one pass).
 */
int simamba_bn_relu_grid(long long rows);
int simamba_bn_relu_fwd(const void* x, const float* gterm, int group, const float* weight, const float* bias,
                        float* running_mean, float* running_var, float momentum, float eps, int training,
                        void* y, float* mean, float* invstd, float* partial, long long rows, int C,
                        long long ld, int io_dtype, void* stream);
int simamba_bn_relu_bwd(const void* dy, const void* x, const float* gterm, int group, const float* weight,
                        const float* bias, const float* mean, const float* invstd, void* dx, float* dgterm,
                        int dgroup, float* dweight, float* dbias, float* partial, long long rows, int C,
                        long long ld, int io_dtype, int training, void* stream);
int simamba_group_max_fwd(const void* x, void* out, unsigned char* idx, long long groups, int n, int C,
                          int io_dtype, void* stream);
int simamba_group_max_bwd(const void* dout, const unsigned char* idx, void* dx, long long groups, int n, int C,
                          int io_dtype, void* stream);

/*
 * Feature propagation of the part-segmentation head (reference part_segmentation/models/pointnet2_utils.py:262-305,
 * PointNetFeaturePropagation.forward): for every query point the three nearest of the sample's S centres
 * (squared distance in the reference's expanded form, ties to the lower index), weights 1/(d + 1e-8) normalised
 * to sum 1, and the weighted sum of the three centres' feature rows.
 *   xyz1 : (batch, N, 3) fp32 query points ; xyz2 : (batch, S, 3) fp32 centres ; idx : (batch, N, 3) int32 ;
 *   weight : (batch, N, 3) fp32 ; feats : (batch, S, C) io_dtype ; out : (batch, N, C) io_dtype ;
 *   dfeats : (batch, S, C) fp32, every element written (deterministic gather, no atomics).  C % 4 == 0, S <= 8192,
 *   N <= 8192 for the backward.
 */
int simamba_three_nn(const float* xyz1, const float* xyz2, int* idx, float* weight, int batch, int N, int S,
                     void* stream);
int simamba_three_interpolate_fwd(const void* feats, const int* idx, const float* weight, void* out, int batch,
                                  int N, int S, int C, int io_dtype, void* stream);
int simamba_three_interpolate_bwd(const void* dout, const int* idx, const float* weight, float* dfeats, int batch,
                                  int N, int S, int C, int io_dtype, void* stream);

/*
 * Chamfer distance of the MAE pre-training loss (reference models/point_mamba.py:2950, :3203:
 * pytorch3d.loss.chamfer_distance(pred, gt, batch_reduction=None) -- squared L2 to the nearest neighbour, mean over
 * each set's points, both directions summed; one value per pair of sets).
 *   pred : (pairs, n, 3) fp32 ; gt : (pairs, m, 3) fp32 ; dist : (pairs) fp32 ; idx1 : (pairs, n) uint8 nearest gt
 *   of every prediction ; idx2 : (pairs, m) uint8 nearest prediction of every gt point ; n, m <= 64.
 * Backward: dpred (pairs, n, 3) fp32 from ddist (pairs) (gt carries no gradient).
 */
int simamba_chamfer_fwd(const float* pred, const float* gt, float* dist, unsigned char* idx1, unsigned char* idx2,
                        long long pairs, int n, int m, void* stream);
int simamba_chamfer_bwd(const float* pred, const float* gt, const float* ddist, const unsigned char* idx1,
                        const unsigned char* idx2, float* dpred, long long pairs, int n, int m, void* stream);

/*
 * k-NN grouping of the tokeniser (reference models/point_mamba.py:96: pytorch3d.ops.knn_points(center, xyz,
 * K=group_size, return_sorted=False)): idx[b][g][0..K) = the K points of cloud b nearest to centre g, ascending
 * squared distance (direct differences), ties to the lower point index.
 *   points : (batch, N, 3) fp32 ; centers : (batch, G, 3) fp32 ; idx : (batch, G, K) int64.  K <= N <= 8192.
 */
int simamba_knn_group(const float* points, const float* centers, long long* idx, int batch, int N, int G, int K,
                      void* stream);

/* ---- spectral ordering ---------------------------------------------------------------- */
#define SIMAMBA_SPEC_SYMMETRIC   0x01u  /* also write A[j,i] for every kNN edge (i,j)          */
#define SIMAMBA_SPEC_SELF_LOOP   0x02u  /* keep the nearest neighbour (the point itself)        */
#define SIMAMBA_SPEC_BINARY      0x04u  /* edge weight 1 instead of exp(-alpha d^2)             */
#define SIMAMBA_SPEC_MATRIX_SYM  0x08u  /* L = I - D^-1/2 A D^-1/2, take k+1, drop the first    */
#define SIMAMBA_SPEC_SMALLEST    0x10u  /* k smallest eigenvalues (else k largest)              */
#define SIMAMBA_SPEC_SIGMA_MEAN  0x20u  /* weight exp(-d^2 / 2 sigma^2), sigma = mean distance
                                           over the whole batch (reference alpha == 0 branch)  */

/*
 * k-NN graph adjacency.  points (B,G,F) fp32 -> adj (B,G,G) fp32.
 * workspace: >= simamba_spectral_workspace_bytes(B,G) bytes (used by SIGMA_MEAN only).
 */
int simamba_knn_graph(const float* points, float* adj, void* workspace, size_t ws_bytes,
                      int B, int G, int F, int knn, float alpha, unsigned flags, void* stream);

/*
 * Laplacian eigen-decomposition, one workgroup per sample (cyclic Jacobi, LDS-resident).
 *   adj        : (B,G,G) fp32
 *   evals      : (B,k)      evecs : (B,G,k)      order : (B,k,G) int64 (ascending argsort of
 *                each selected eigenvector, ties by index); any of the three may be NULL.
 *   all_evals  : (B,G) ascending, all_evecs : (B,G,G) columns = eigenvectors; may be NULL.
 * Reproduces the reference's quirk of decomposing the LOWER TRIANGLE of I - D^-1 A.
 * Eigenvector sign: the component of largest magnitude is made positive (LAPACK/cuSOLVER
 * leave the sign unspecified).
 */
int simamba_laplacian_topk(const float* adj, float* evals, float* evecs, long long* order,
                           float* all_evals, float* all_evecs, int B, int G, int k,
                           unsigned flags, void* stream);

size_t simamba_spectral_workspace_bytes(int B, int G);

/* centres (B,G,3) -> top-k eigenpairs + orders; workspace holds the (B,G,G) adjacency. */
int simamba_spectral_topk(const float* centers, float* evals, float* evecs, long long* order,
                          void* workspace, size_t ws_bytes, int B, int G, int knn, float alpha,
                          int k, unsigned flags, void* stream);

/*
 * Farthest-point sampling (tokeniser step ahead of the hot path; replaces
 * pytorch3d.ops.sample_farthest_points at reference models/point_mamba.py:93).
 *   points (B, N, 3) fp32 -> idx (B, K) int64 (first pick = point 0, ties to the lower index),
 *   centers (B, K, 3) fp32 or NULL.  N <= 4096, K <= N.
 */
int simamba_farthest_point_sample(const float* points, long long* idx, float* centers, int B, int N,
                                  int K, void* stream);

/* vals (rows, n) fp32 -> idx (rows, n) int64, ascending, ties by index.  n <= 1024. */
int simamba_argsort_rows(const float* vals, long long* idx, int rows, int n, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SIMAMBA_H_ */
