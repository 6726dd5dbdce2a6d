/*
 * coral_amd.h — C ABI of libcoral_amd.so, the MI355X (gfx950) hot path behind CoRal's
 * ModelSetup boundary.
 *
 * The reference (alexandrainst/coral) has no FFI of its own: everything below the
 * ModelSetup ABC (R/src/coral/data_models.py:44-82) runs inside HuggingFace Transformers
 * + torch ATen.  Each entry point here therefore cites the *library* call it replaces
 * ($TF = transformers/, line numbers for v5.15.0) and the CoRal line that causes the call.
 *
 * Conventions (all entry points):
 *   - plain pointers are DEVICE pointers unless the name ends in _h;
 *   - `stream` is a hipStream_t passed as void*; kernels are enqueued, never synchronised;
 *   - no allocation, no ownership transfer: the caller allocates outputs and workspaces;
 *   - return 0 on success, negative CA_ERR_* on failure; ca_last_error() gives the text;
 *   - activations are row-major "channels-last" [rows, channels] bf16 unless stated;
 *   - no CPU fallback exists in this library.
 */
#ifndef CORAL_AMD_H
#define CORAL_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CA_OK 0
#define CA_ERR_ARG (-1)
#define CA_ERR_LAUNCH (-2)
#define CA_ERR_UNSUPPORTED (-3)

/* library / diagnostics */
int ca_version(void);
const char* ca_last_error(void);
/* number of HIP devices visible (does not initialise a context beyond the count) */
int ca_device_count(void);

/* ------------------------------------------------------------------------------------
 * GEMM family (MFMA bf16, fp32 accumulate).  Replaces every nn.Linear / Conv1d-as-GEMM /
 * attention matmul on the path:
 *   $TF/models/wav2vec2/modeling_wav2vec2.py:495-498,546 (q,k,v,o)  :565-572 (FFN)
 *   :429-434 (feature projection) :282-299 (conv layers 1..6, as implicit GEMM)
 *   :326-368 (positional conv, grouped implicit GEMM) :1700 (lm_head)
 *   $TF/models/whisper/modeling_whisper.py:284-356,379-413,448-505 (encoder/decoder)
 *
 *   C[m,n] = epilogue( alpha * sum_k opA(m,k) * opB(n,k) )
 *
 * Operand layouts (bf16):
 *   CA_KMAJOR  : element (m,k) at  A[m*lda + k]   (reduction index contiguous)
 *   CA_MNMAJOR : element (m,k) at  A[row(k) + m]  (output index contiguous);
 *                row(k) = k*lda, or with kseg>0: (k / kseg)*kseg_stride + (k % kseg)*lda,
 *                which lets a reduction run over (batch, time) of a strided/overlapping
 *                window view (conv weight-gradients) without materialising im2col.
 * Overlapping rows are legal (lda smaller than the row length): a channels-last Conv1d
 * with stride s is exactly a KMAJOR GEMM with lda = s*C_in and K = k*C_in.
 * Constraints: all leading dimensions and offsets multiples of 8 elements (16 bytes);
 * KMAJOR operands must be readable (and zero) up to K rounded up to 8; MNMAJOR operand rows
 * must be readable up to M (resp. N) rounded up to 8 (those extra outputs are discarded).
 * Batch: blockIdx.z = z1*batch2 + z2, pointer offsets z1*s?1 + z2*s?2 (elements).
 * ---------------------------------------------------------------------------------- */
#define CA_KMAJOR 0
#define CA_MNMAJOR 1

#define CA_EPI_NONE 0      /* C = v                         (v = alpha*acc + bias)      */
#define CA_EPI_GELU 1      /* C = v (pre-activation, may be NULL), C2 = gelu_erf(v)     */
#define CA_EPI_RESIDUAL 2  /* C = v + R                                                 */
#define CA_EPI_DGELU 3     /* C = v * gelu_erf'(R)          (R = saved pre-activation)  */
#define CA_EPI_GELU_RESIDUAL 4 /* C = v (may be NULL), C2 = gelu_erf(v) + R  (pos-conv add) */

typedef struct CaGemmDesc {
  const void* A;
  const void* B;
  void* C;
  void* C2;          /* second output for CA_EPI_GELU (bf16), else NULL */
  const void* R;     /* bf16 side input for RESIDUAL / DGELU, else NULL */
  const float* bias; /* [N] fp32 or NULL */
  int32_t M, N, K;
  int32_t a_layout, b_layout;
  int64_t lda, ldb, ldc, ldr;
  int32_t a_kseg, b_kseg;
  int64_t a_kseg_stride, b_kseg_stride;
  int32_t batch1, batch2;
  int64_t sA1, sA2, sB1, sB2, sC1, sC2, sR1, sR2;
  int64_t sBias1, sBias2; /* per-batch offset of the bias vector (grouped conv) */
  int32_t epilogue;
  int32_t out_f32;    /* 0: C is bf16; 1: C is fp32 */
  int32_t accumulate; /* 1: C += result (read-modify-write) */
  float alpha;
  /* activation dropout fused into CA_EPI_GELU / CA_EPI_DGELU
   * ($TF/models/wav2vec2/modeling_wav2vec2.py:556,567; R/config/model/wav2vec2-large.yaml:12):
   * keep-mask = hash(seed, m*N+n) >= p; kept values scaled by 1/(1-p). p = 0 disables.
   * With CA_EPI_RESIDUAL: C = R + dropout(alpha A.B + bias), the hidden-state dropout in front of a residual add
   * ($TF/models/whisper/modeling_whisper.py:398,406,479,493,502; R/config/model/whisper-large-turbo.yaml:12);
   * ca_dropout_bf16 regenerates the same mask for the backward. */
  float dropout_p;
  uint64_t dropout_seed;
  /* Optional, weight-gradient form only (a_layout = b_layout = MNMAJOR, served by the 256x256 kernel): partial
   * column sums of A — the bias gradient that autograd's Linear backward computes as dY.sum(0) — taken from the A
   * tiles as they stream through the kernel instead of a second pass over dY.  The K-steps are dealt round-robin
   * to the first P = min(8, ceil(N/256)) tile columns, and tile column p stores its share to
   * a_colsum[p * a_colsum_ld + m] (plain stores; rows p >= P are not touched).  The caller adds the P rows
   * (ca_reduce_rows_f32).  NULL = off. */
  float* a_colsum;
  int64_t a_colsum_ld;
  /* Optional, skinny form only (M <= 32, both operands K-major, un-batched: one decoded token per clip,
   * R/src/coral/whisper.py generate path): output row m goes to row m * c_row_mul + c_row_index[m] of C (device-side
   * positions: the new token's K|V rows appended to a cache at a position that is data, not a launch argument).
   * NULL = rows in place. */
  const int32_t* c_row_index;
  int64_t c_row_mul;
  /* Optional, skinny form only: columns [c_split_n, N) are written to C_hi (leading dimension ldc_hi, column index
   * n - c_split_n) instead of C, and c_row_index then applies to those columns only - the q projection and the
   * cached K|V rows of a decoded token from one launch over the adjacent q|k|v weights.  0 = off; a multiple of 16;
   * CA_EPI_NONE only. */
  int32_t c_split_n;
  void* C_hi;
  int64_t ldc_hi;
  /* fp8 form only (ca_gemm_fp8): per-tensor dequantisation scales of A and B, device scalars (NULL = 1): the
   * accumulator is multiplied by alpha * a_scale[0] * b_scale[0]. */
  const float* a_scale;
  const float* b_scale;
  const float* a_row_scale; /* fp8 form only: [M] per-row factors of A (ca_layernorm_fwd_fp8), NULL = none */
  /* Optional, plain epilogue, un-batched, fp32 or bf16 output: the sum of squares of the values stored to C (after
   * `accumulate`; of the ROUNDED values when C is bf16),
   * one partial per 64 x 64 output block at c_sumsq[(m / 64) * ceil(N / 64) + n / 64] (plain stores, every block of the
   * output written once per launch: no atomics, the same bits on every run).  The squared norm of a weight gradient
   * then costs no second pass over it: Trainer's clip_grad_norm_ ($TF/trainer.py:1778-1796) adds the partials of all
   * matrices (ca_sum_f32) to the squared norm of the small tensors (ca_sumsq_ranges_f32).  NULL = off. */
  float* c_sumsq;
  /* != 0: C is written once and not read again soon (the FFN pre-activation kept for the backward, a weight gradient
   * the optimiser reads a backward later): interior tiles store it with the non-temporal hint, so that it does not
   * displace the operands of the next GEMMs from the L2s / Infinity Cache (measured: -0.5 ms of the 73-ms XLS-R-2B step
   * for the pre-activation alone).  Results are unchanged.  C2 always takes the default policy. */
  int32_t c_stream_out;
  /* Optional, CA_EPI_GELU / CA_EPI_DGELU: one more output C8[m * ldc + n] = e4m3(clamp(w * c8_scale[0])), w = gelu(v)
   * (after dropout; CA_EPI_GELU) or the gradient v * gelu'(R) (CA_EPI_DGELU) - the fp8 A operand of the NEXT projection
   * / data gradient (ca_gemm_fp8 with a_scale = the matching inv_scale), written by the tile that computes it: no
   * quantisation pass.  c8_scale is a device scalar from the previous step's
   * amax (ca_fp8_amax_rotate); max|gelu(v)| of this launch is accumulated into the CA_FP8_AMAX_SLOTS words at c8_amax.
   * NULL = off. */
  void* C8;
  const float* c8_scale;
  uint32_t* c8_amax;
  /* Optional, skinny form only (M <= 32, both operands K-major, un-batched; K <= 2048): the A operand is
   * LayerNorm(A rows) over the K elements of each row with these fp32 [K] vectors, formed in every workgroup's prologue
   * and rounded to bf16 exactly as ca_layernorm_fwd would have stored it (bit-identical to the two launches): the
   * LayerNorm in front of the q|k|v and fc1 projections of a decoded token ($TF/models/whisper/modeling_whisper.py:
   * 459-460,497-498 on the R/src/coral/whisper.py generate path) costs no launch of its own.  NULL = off. */
  const float* a_ln_gamma;
  const float* a_ln_beta;
  float a_ln_eps;
  int32_t xcd_balanced; /* internal: overwritten by the library (ca_gemm_set_compute_cus); callers leave it 0 */
} CaGemmDesc;

int ca_gemm_bf16(const CaGemmDesc* desc, void* stream);
/* The same descriptor with A [M][K] and B [N][K] holding OCP fp8 e4m3 bytes (both K-major, un-batched; lda, ldb, K in
 * elements = bytes, multiples of 16): C = epilogue(alpha * a_scale * b_scale * A B^T) on the CDNA4 fp8 matrix
 * instruction (BASELINE.json configs[4]: fp8 weights; replaces the bf16 `F.linear` of a forward projection whose
 * input and weight went through ca_quantize_fp8).  Bias, epilogues, C / C2 / R as for ca_gemm_bf16. */
int ca_gemm_fp8(const CaGemmDesc* desc, void* stream);
/* Per-tensor fp8 quantisation of a bf16 tensor of n elements: amax -> scale = 448 / amax, q = e4m3(x * scale)
 * (round to nearest even, saturating), inv_scale[0] = amax / 448 (the dequantisation factor ca_gemm_fp8 takes).
 * amax_ws: one float of workspace.  n must be a multiple of 8. */
int ca_quantize_fp8(const void* x_bf16, int64_t n, void* q_fp8, float* inv_scale, float* amax_ws, void* stream);
/* Delayed scaling (the fp8 training recipe: the scale of step t comes from the amax of step t - 1): ONE pass that
 * quantises x with the device scalar scale[0] (q = e4m3(clamp(x * scale)), saturating) and accumulates max|x| into
 * amax_next[0 .. CA_FP8_AMAX_SLOTS) (bits of non-negative floats; atomic max, order-independent).  ca_fp8_amax_rotate then turns `count`
 * accumulated amax accumulators (CA_FP8_AMAX_SLOTS words each) into the next step's scales: scale[i] = 448 / (margin * amax), inv_scale[i] =
 * margin * amax / 448 (what ca_gemm_fp8 takes as a_scale / b_scale), amax_next[i] = 0; a zero word keeps the old scale.
 * The same scale / amax pair serves activations produced by a GEMM epilogue (CaGemmDesc.C8). */
#define CA_FP8_AMAX_SLOTS 64 /* an amax accumulator is this many words (the writers spread their atomics over them:
                              * atomics on one address serialise); the tensor's amax is the maximum over the words */
int ca_quantize_fp8_delayed(const void* x_bf16, int64_t n, void* q_fp8, const float* scale, uint32_t* amax_next,
                            void* stream);
int ca_fp8_amax_rotate(uint32_t* amax_next, float* scale, float* inv_scale, int32_t count, float margin, void* stream);
/* Data gradients on the fp8 path (dX = dY W through ca_gemm_fp8: A = dY as e4m3 with row scales, B = W transposed):
 * ca_dropout_rows_fp8: y = dropout(x) exactly as ca_dropout_bf16 (mask of element i from (seed, i), i = row * C + col;
 *   p = 0: no mask, y may be NULL) and, in the same pass, q[row] = e4m3(y[row] * 448 / amax(y[row])), row_scale[row] =
 *   amax / 448 (CaGemmDesc.a_row_scale) - the gradient entering a sub-layer whose output went through hidden-state
 *   dropout ($TF/models/whisper/modeling_whisper.py:398,406) as the fp8 operand of that sub-layer's data gradient.
 * ca_quantize_fp8_transposed: q_t [cols, rows] = e4m3(clamp(x [rows, cols] * scale[0])) - the transposed e4m3 copy of a
 *   weight matrix, quantised with the scale of its untransposed copy (ca_quantize_fp8_delayed). */
int ca_dropout_rows_fp8(const void* x, void* y, void* q_fp8, float* row_scale, int64_t rows, int32_t C, float p,
                        uint64_t seed, void* stream);
int ca_quantize_fp8_transposed(const void* x_bf16, int32_t rows, int32_t cols, void* q_fp8_t, const float* scale,
                               void* stream);
/* The per-step refresh of a layer's e4m3 weight copies in ONE launch: for each task, q_fp8 [rows, cols] =
 * e4m3(clamp(x * scale[0])) exactly as ca_quantize_fp8_delayed (amax of x accumulated into amax_next, may be NULL) and,
 * when q_fp8_t is not NULL, the transposed copy [cols, rows] exactly as ca_quantize_fp8_transposed.  cols % 8 == 0;
 * a transposed copy also needs rows % 8 == 0. */
#define CA_FP8_GROUP_MAX 8
typedef struct CaFp8RefreshTask {
  const void* x_bf16;
  void* q_fp8;
  void* q_fp8_t;       /* or NULL */
  const float* scale;  /* device scalar */
  uint32_t* amax_next; /* CA_FP8_AMAX_SLOTS words, or NULL */
  int32_t rows, cols;
} CaFp8RefreshTask;
int ca_fp8_refresh_group(const CaFp8RefreshTask* tasks, int32_t count, void* stream);
/* Up to eight independent plain GEMMs of the same operand form (same a_layout / b_layout, un-batched, no epilogue,
 * no bias) in one launch of the 256x256 kernel: for problems that under-fill the chip one by one, e.g. the four
 * weight gradients of an encoder layer or the six token-side ones of a Whisper decoder layer (each replaces a
 * `torch.mm(dY.T, X)` of autograd's Linear backward). */
int ca_gemm_bf16_group(const CaGemmDesc* descs, int32_t count, void* stream);

/* Live kernel timing for bench.py's roofline line: between ca_prof_begin() and ca_prof_end()
 * every ca_gemm_bf16 launch is bracketed by hipEvents on its own stream.  ca_prof_end fills
 * three arrays of 24 entries indexed by kernel*8 + segmented*4 + a_layout*2 + b_layout (kernel 0 =
 * ca_gemm_kernel, 1 = ca_gemm_kernel_l, 2 = ca_gemm_kernel_x; segmented = a_kseg or b_kseg set):
 * summed kernel milliseconds, launch count, summed algorithmic FLOPs (2*M*N*K*batch).  Not for use
 * inside graph capture. */
int ca_prof_begin(void);
int ca_prof_end(double* ms, int64_t* count, double* flops);

/* ---- CA_DEBUG_API: test and tuning hooks.  Not part of the product interface - no host code of the path calls them;
 * tests/ and tools/ do, to run every tile shape through the same parity cases. ---------------------------------------
 * ca_gemm_force_kernel: 0 = automatic kernel choice, 1 = force the 128x128 kernel, 2 = force the 256x128 pipelined
 * kernel, 3 = force the 256x256 kernel, 5 = force the 128x128 tile on 8 waves (kernel M: what the automatic choice runs
 * when the grid has at most one 128x128 tile per CU).  (4 was the one-wave-per-SIMD experiment of round 3 and 6 the
 * deferred-store kernel P of round 5 - NOTEBOOK.md 4.1 / R5.1: measured, not adopted, removed.)
 * ca_gemm_debug_general_epilogue: on != 0 sends every wave tile through the general epilogue walk (interior tiles
 * normally take a specialised, predicate-free form that must give the same bits). */
int ca_gemm_force_kernel(int which);
int ca_gemm_debug_general_epilogue(int on);
/* ca_debug_cu_hog: `blocks` idle workgroups (`threads` threads, `lds_bytes` of LDS each) resident for `ms` milliseconds on
 * `stream` - a stand-in for the ring kernel of a collective that holds CUs during the backward of an N > 1 run
 * (accelerate/accelerator.py:1892,2053), so that its effect on the GEMM launches can be measured on one GPU. */
int ca_debug_cu_hog(int32_t blocks, int32_t threads, int32_t lds_bytes, double ms, void* stream);

/* CUs the compute kernels may count on: 0 (default) = the whole chip.  With n > 0 - what the trainer of an N > 1 run
 * sets, because the gradient exchange's kernel occupies CUs for most of the backward - persistent GEMM launches use at
 * most n workgroups (rounded down to a multiple of 8) and hand out EVERY tile through their counters (a workgroup that
 * only starts when others have exited finds the counter exhausted and exits, instead of running a statically assigned
 * tile at the end of the launch), and the tile-shape choice counts rounds on n CUs.  Results never depend on it. */
int ca_gemm_set_compute_cus(int n);

/* ------------------------------------------------------------------------------------
 * LayerNorm over the channel axis.  $TF/models/wav2vec2/modeling_wav2vec2.py:429-434,
 * :611-654,:791 (nn.LayerNorm, eps 1e-5); whisper :379-413.
 * x,y bf16 [rows, C]; gamma,beta fp32 [C]; stats fp32 [rows,2] = (mean, rstd) saved for bwd.
 * act: 0 none, 1 exact-erf GELU applied after the affine (conv-block form, :291-298).
 * ---------------------------------------------------------------------------------- */
int ca_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y,
                     float* stats, int64_t rows, int32_t C, float eps, int32_t act,
                     void* stream);
/* The same with fp32 rows on either side (x_f32: x is fp32 [rows, C]; y_f32: y is fp32), C <= 1024: the conv stack's
 * pre-norm tensors and the last conv block's output, which the reference's autocast keeps in fp32 as well (nn.LayerNorm
 * and the GELU behind it run in fp32 there: $TF/models/wav2vec2/modeling_wav2vec2.py:291-298,429-434) - seven bf16
 * roundings less in front of the transformer. */
int ca_layernorm_fwd_ex(const void* x, const float* gamma, const float* beta, void* y,
                        float* stats, int64_t rows, int32_t C, float eps, int32_t act,
                        int32_t x_f32, int32_t y_f32, void* stream);
/* The same LayerNorm with the output also quantised to OCP fp8 e4m3, one scale per row (taken by the wave that holds
 * the row, no extra pass): q[row] = e4m3(y[row] * 448 / amax(y[row])), row_scale[row] = amax / 448 for
 * CaGemmDesc.a_row_scale.  y (bf16) and stats ([rows][2] mean, rstd for ca_layernorm_bwd) may be NULL.  C must be a
 * multiple of 16. */
int ca_layernorm_fwd_fp8(const void* x, const float* gamma, const float* beta, void* y, void* q_fp8,
                         float* row_scale, float* stats, int64_t rows, int32_t C, float eps, void* stream);
/* dx bf16 [rows,C] (+ dres if non-NULL: the residual-stream gradient that bypasses the LN);
 * dgamma/dbeta fp32 [C] are ACCUMULATED into (+=) through the fp32 partial buffer `partial`
 * of ca_layernorm_bwd_partial_floats(rows, C) floats. */
int64_t ca_layernorm_bwd_partial_floats(int64_t rows, int32_t C);
int ca_layernorm_bwd(const void* dy, const void* x, const float* gamma, const float* beta,
                     const float* stats, const void* dres, void* dx, float* dgamma,
                     float* dbeta, float* partial, int64_t rows, int32_t C, int32_t act,
                     void* stream);

/* x_f32 != 0: the saved input x is fp32 (see ca_layernorm_fwd_ex); dy, dres and dx stay bf16. */
int ca_layernorm_bwd_ex(const void* dy, const void* x, const float* gamma, const float* beta,
                        const float* stats, const void* dres, void* dx, float* dgamma,
                        float* dbeta, float* partial, int64_t rows, int32_t C, int32_t act,
                        int32_t x_f32, void* stream);

/* column sums: out[n] (+)= sum_m x[m*ld + n]  (bias gradients). x bf16, out fp32.
 * rowmask uint8 [rows] or NULL: only rows with a non-zero mask byte are summed (gradient of
 * masked_spec_embed).  partial: fp32 workspace of ca_colsum_partial_floats(rows, N) floats. */
int64_t ca_colsum_partial_floats(int64_t rows, int32_t N);
int ca_colsum_bf16(const void* x, int64_t ld, int64_t rows, int32_t N, const uint8_t* rowmask,
                   float* out, int32_t accumulate, float* partial, void* stream);
/* out[i] (+)= sum_{p<nparts} partial[p*stride + i], i < n  (fp32; split-batch weight gradients) */
int ca_reduce_rows_f32(const float* partial, int32_t nparts, int64_t stride, int32_t n, float* out,
                       int32_t accumulate, void* stream);
/* Up to CA_REDUCE_MAX such reductions (the many-parts form: nparts rows of partial sums, n <= a few thousand columns)
 * in one launch - the second stages of one layer's backward.  Bit-identical to ca_reduce_rows_f32 per reduction. */
#define CA_REDUCE_MAX 4
typedef struct CaReduceDesc {
  const float* partial;
  float* out;
  int64_t stride;
  int32_t nparts, n, accumulate;
} CaReduceDesc;
int ca_reduce_rows_multi(const CaReduceDesc* descs, int32_t count, void* stream);
/* out = dy * gelu_erf'(u), bf16 elementwise (backward of the pos-conv GELU, :374). */
int ca_dgelu_mul(const void* dy, const void* u, void* out, int64_t n, void* stream);
/* y = x * keep / (1 - p), keep of element i from (seed, i): hidden-state dropout as its own pass and the mask the
 * CA_EPI_RESIDUAL epilogue (dropout_p > 0) applies to element (m, n) of a dense [M, N] output at i = m * N + n.
 * Replaces nn.functional.dropout at $TF/models/whisper/modeling_whisper.py:398,406,479,493,502,625,763 (backward: the
 * same call on the incoming gradient).  n % 8 == 0; x == y allowed. */
int ca_dropout_bf16(const void* x, void* y, int64_t n, float p, uint64_t seed, void* stream);

/* ------------------------------------------------------------------------------------
 * Waveform front end.
 * ca_wave_normalize: zero-mean / unit-variance over the valid samples of each utterance,
 *   padding -> 0.  $TF/models/wav2vec2/feature_extraction_wav2vec2.py:77-97 (called from
 *   R/src/coral/data.py:747).  x,y fp32 [B,N]; lengths int32 [B].
 * ca_conv0_*: feature-encoder layer 0 — Conv1d(1,C,k,stride,bias) + LayerNorm(C) + GELU
 *   fused ($TF/models/wav2vec2/modeling_wav2vec2.py:275-299 with layer_id 0).
 *   x fp32 [B,N] -> y bf16 [B,T0,C], T0 = (N-k)/stride+1.  w fp32 [C,k].
 *   The backward recomputes the conv from x (10 MACs/channel) instead of saving it.
 * ---------------------------------------------------------------------------------- */
int ca_wave_normalize(const float* x, const int32_t* lengths, float* y, int32_t B,
                      int64_t N, float eps, void* stream);

/* out[b] = _get_feat_extract_output_lengths(attention_mask[b].sum()) for the conv stack (kernels[i], strides[i]),
 * i < nconv <= 8: floor((n - k) / s) + 1 per layer ($TF/models/wav2vec2/modeling_wav2vec2.py:1093-1108).  kernels /
 * strides are HOST arrays; attention_mask is int32 [B, N] on the device. */
int ca_frame_lengths(const int32_t* attention_mask, int32_t B, int64_t N, const int32_t* kernels,
                     const int32_t* strides, int32_t nconv, int32_t* out, void* stream);
/* ca_pcm_prepare: raw PCM batch -> model input on the device (replaces the per-example
 *   host featurisation `processor(audio)` at R/src/coral/data.py:747 plus the collator's padding at
 *   R/src/coral/data_collators.py:72-77).  pcm: int16 (is_int16 != 0, scaled by 1/32768) or fp32, B rows of
 *   ld_in samples; lengths int32 [B] (NULL = full rows).  Optional peak normalisation (x / max|x|,
 *   R/src/coral/data.py:710), optional zero-mean / unit-variance over the valid samples
 *   ($TF/models/wav2vec2/feature_extraction_wav2vec2.py:77-97, eps 1e-7); y fp32 [B,N] with padding 0,
 *   mask int32 [B,N] (attention_mask) or NULL. */
int ca_pcm_prepare(const void* pcm, int32_t is_int16, int64_t ld_in, const int32_t* lengths, float* y,
                   int32_t* mask, int32_t B, int64_t N, int32_t peak_normalize, int32_t zero_mean_unit_var,
                   float eps, void* stream);
/* ------------------------------------------------------------------------------------
 * On-device waveform augmentation (the chain torch_audiomentations builds at
 * R/src/coral/data.py:708-738: Gain, AddBackgroundNoise, AddColoredNoise, Band/High/Low-pass, BandStop).
 * Random draws and filter designs happen on the host (coral_amd/augment.py); these kernels are deterministic.
 * x,y fp32 [B,N]; lengths int32 [B] (samples past the length are written as 0).
 * ca_wave_scale:  y = x * scale[b].
 * ca_fir_filter:  odd-length FIR centred on the output sample, edges replicated (julius low-pass filters);
 *                 taps fp32 [B, ld_taps], ntaps int32 [B]; mode int32 [B]: 0 copy, 1 fir(x), 2 x - fir(x).
 * ca_mix_noise:   y = x + noise * rms(x)/rms(noise) * 10^(-snr_db[b]/20) where active[b] != 0; noise rows of
 *                 noise_len samples (row stride noise_ld, 0 = one shared row) are read cyclically from noise_off[b].
 * ca_white_noise: n standard-normal samples from a counter hash of (seed, index).
 * ---------------------------------------------------------------------------------- */
int ca_wave_scale(const float* x, const float* scale, float* y, int32_t B, int64_t N, void* stream);
int ca_fir_filter(const float* x, const int32_t* lengths, const float* taps, const int32_t* ntaps,
                  const int32_t* mode, int64_t ld_taps, int32_t max_taps, float* y, int32_t B, int64_t N,
                  void* stream);
int ca_mix_noise(const float* x, const int32_t* lengths, const float* noise, int64_t noise_ld, int64_t noise_len,
                 const int64_t* noise_off, const float* snr_db, const int32_t* active, float* y, int32_t B,
                 int64_t N, void* stream);
int ca_white_noise(float* out, int64_t n, uint64_t seed, void* stream);

int ca_conv0_ln_gelu_fwd(const float* x, const float* w, const float* bias,
                         const float* gamma, const float* beta, void* y, int32_t B,
                         int64_t N, int32_t C, int32_t k, int32_t stride, float eps,
                         void* stream);
int64_t ca_conv0_bwd_partial_floats(int32_t B, int64_t N, int32_t C, int32_t k,
                                    int32_t stride);
int ca_conv0_ln_gelu_bwd(const float* x, const float* w, const float* bias,
                         const float* gamma, const float* beta, const void* dy, float* dw,
                         float* dbias, float* dgamma, float* dbeta, float* partial,
                         int32_t B, int64_t N, int32_t C, int32_t k, int32_t stride,
                         float eps, void* stream);

/* col2im for a strided channels-last Conv1d data-gradient:
 * dx[b,p,c] = sum_{j, t*stride+j==p} dcol[b,t,j*C+c].  dcol bf16 [B,T,k*C] (output of the
 * dgrad GEMM), dx bf16 [B,L,C]. */
int ca_col2im_1d(const void* dcol, void* dx, int32_t B, int64_t T, int64_t L, int32_t C,
                 int32_t k, int32_t stride, void* stream);

/* ------------------------------------------------------------------------------------
 * Attention softmax (the middle of SDPA, $TF/models/wav2vec2/modeling_wav2vec2.py:438-463,
 * $TF/models/whisper/modeling_whisper.py:215-238).  scores fp32 [BH, Tq, ld] (alpha already
 * applied by the GEMM), key padding by klen[b] (b = bh / H), optional causal mask;
 * probs bf16 [BH, Tq, ld] with columns >= Tk written as zero (so it can feed a KMAJOR GEMM).
 * bwd: ds = p * (dp - sum_j dp_j p_j) * scale, bf16, zero padded.
 * ---------------------------------------------------------------------------------- */
int ca_softmax_fwd(const float* scores, void* probs, const int32_t* klen, int32_t BH,
                   int32_t H, int32_t Tq, int32_t Tk, int64_t ld, int32_t causal,
                   void* stream);
int ca_softmax_bwd(const float* dprobs, const void* probs, void* dscores, float scale,
                   int32_t BH, int32_t Tq, int32_t Tk, int64_t ld, void* stream);

/* ------------------------------------------------------------------------------------
 * Fused multi-head attention (flash-style, no [T,T] matrix in HBM): forward and backward.
 *   $TF/models/wav2vec2/modeling_wav2vec2.py:438-463,529-543 (SDPA) and
 *   $TF/models/whisper/modeling_whisper.py:215-238,284-356.
 * Q [B,Tq,*], K,V [B,Tk,*] bf16 with row strides ldq/ldk/ldv, batch strides s?b (elements) and head h
 * at column offset h*hd; O/dO/dQ/dK/dV likewise.  softmax(scale * Q K^T + masks) V with key padding
 * klen[b] (NULL = none) and an optional causal mask.  lse fp32 [B,H,Tqp] (row log-sum-exp, written
 * by the forward, read by the backward), Dq fp32 [B,H,Tqp] scratch for rowsum(dO*O); Tqp % 32 == 0.
 * head_dim: any multiple of 8 up to 128.
 * ---------------------------------------------------------------------------------- */
typedef struct CaAttnDesc {
  const void *Q, *K, *V;
  void* O;          /* forward: output; backward: the saved forward output */
  const void* dO;   /* backward only */
  void *dQ, *dK, *dV;
  float* lse;
  float* Dq;
  const int32_t* klen;
  int64_t ldq, ldk, ldv, ldo, lddo, lddq, lddk, lddv;
  int64_t sqb, skb, svb, sob, sdob, sdqb, sdkb, sdvb;
  int32_t B, H, Tq, Tk, hd, Tqp, causal;
  float scale;
  /* dropout on the attention probabilities (training only; $TF/models/whisper/modeling_whisper.py:234,
   * nn.functional.dropout(attn_weights, p=attention_dropout)): P' = P * keep / (1 - p) after the softmax, keep of
   * (b, h, q, k) from a hash of (seed, flat index); forward and backward must be given the same p and seed. 0 = off. */
  float dropout_p;
  uint64_t dropout_seed;
  /* Optional, forward with at least 100 queries per head: the output also as OCP e4m3, O8[same element offsets as O] =
   * e4m3(clamp(bf16(O) * o8_scale[0])) - the fp8 A operand of the out-projection (ca_gemm_fp8, a_scale = the matching
   * inv_scale) written by the attention kernel's output stage, no quantisation pass.  o8_scale: device scalar from the
   * previous step's amax (delayed scaling, ca_fp8_amax_rotate); this launch's max |O| goes to the CA_FP8_AMAX_SLOTS words
   * at o8_amax.  NULL = off. */
  void* O8;
  const float* o8_scale;
  uint32_t* o8_amax;
  /* Optional, forward with Tq <= 16 and head_dim <= 64 (greedy decoding, ca_attn_fwd and ca_decode_attn_qproj): a
   * workspace of CA_ATTN_SPLIT_WS_BYTES(B, H) bytes, zero-filled ONCE by the caller (the kernels leave its counters at
   * zero), lets the keys of one (clip, head) be dealt to up to CA_ATTN_SPLIT_MAX workgroups when B x H workgroups would
   * leave half of the CUs idle and Tk >= 1024 (the 1500 cross-attention keys of $TF/models/whisper/modeling_whisper.py:
   * 312-335 at the evaluation batch sizes of R/config/evaluation.yaml) - or, with more than one round of (clip, head)
   * items and fewer than two (whisper-large at 16 clips: 320 on 256 CUs), the items of the second round only, so that
   * their parts fill it; partial results are merged in a fixed order by the workgroup that finishes last.  One launch at
   * a time per workspace.  NULL = one workgroup per (clip, head). */
  void* split_ws;
  int64_t split_ws_bytes;
} CaAttnDesc;
#define CA_ATTN_SPLIT_MAX 4
#define CA_ATTN_SPLIT_WS_BYTES(B, H) \
  ((((int64_t)(B) * (H) * 4 + 255) / 256 * 256) + (int64_t)(B) * (H) * CA_ATTN_SPLIT_MAX * 16 * 66 * 4)
int ca_attn_fwd(const CaAttnDesc* desc, void* stream);
int ca_attn_bwd(const CaAttnDesc* desc, void* stream);
/* Greedy decoding, one new token per clip: the pre-attention LayerNorm, the query projection and the single-query
 * attention over a K|V cache in ONE launch - what `WhisperDecoderLayer` does between the residual stream and the
 * attention output for its cross-attention ($TF/models/whisper/modeling_whisper.py:476-486: encoder_attn_layer_norm,
 * q_proj, attention over the cached encoder K|V).  x bf16 [B, d_model] (row stride ldx); Wq bf16 [H*hd, d_model]
 * row-major (row stride ldw), bq fp32; desc: K / V / O / klen / strides / B / H / Tk / hd / scale as for ca_attn_fwd
 * with Tq = 1 (Q is not read).  Bit-identical to ca_layernorm_fwd + ca_gemm_bf16 + ca_attn_fwd on the same inputs. */
int ca_decode_attn_qproj(const CaAttnDesc* desc, const void* x, int64_t ldx, const float* ln_gamma,
                         const float* ln_beta, float ln_eps, const void* Wq, int64_t ldw, const float* bq,
                         int32_t d_model, void* stream);

/* ------------------------------------------------------------------------------------
 * CTC head: log_softmax(fp32) + CTC loss (+ gradient wrt logits) + greedy decode.
 *   $TF/models/wav2vec2/modeling_wav2vec2.py:1705-1728 (F.ctc_loss, blank = pad id,
 *   zero_infinity) ← R/src/coral/wav2vec2.py:120,125; greedy: R/src/coral/compute_metrics.py:68-69
 *   + $TF/models/wav2vec2/tokenization_wav2vec2.py:311-323 (collapse repeats, drop blank).
 * logits fp32 [B,T,ldv] (first V columns valid); labels int32 [B,Lmax] (-100 padded);
 * in_len int32 [B] frames; nll fp32 [B] per-utterance loss (0 where infeasible and
 * zero_infinity); grad fp32 [B,T,ldv] = d(sum_b nll_b * gscale[b])/dlogits
 * (gscale NULL = 1).  ws: ca_ctc_workspace_bytes(B,T,Lmax) bytes.
 * ---------------------------------------------------------------------------------- */
int64_t ca_ctc_workspace_bytes(int32_t B, int32_t T, int32_t Lmax);
int ca_ctc_loss_fwd_bwd(const float* logits, const int32_t* labels, const int32_t* in_len,
                        float* nll, float* grad, const float* gscale, void* ws, int32_t B,
                        int32_t T, int32_t V, int64_t ldv, int32_t Lmax, int32_t blank,
                        int32_t zero_infinity, void* stream);
/* ids int32 [B,T] (collapsed ids, -1 padded), out_len int32 [B]; raw int32 [B,T] argmax. */
int ca_ctc_greedy_decode(const float* logits, const int32_t* in_len, int32_t* raw,
                         int32_t* ids, int32_t* out_len, int32_t B, int32_t T, int32_t V,
                         int64_t ldv, int32_t blank, void* stream);

/* ------------------------------------------------------------------------------------
 * Small elementwise pieces of the wav2vec2 encoder.
 * ca_mask_frames: SpecAugment + padding ($TF/.../modeling_wav2vec2.py:1272-1316, 752-755):
 *   h[b,t,:] = embed where tmask[b,t]; h[b,t,c] = 0 where fmask[b,c]; h[b,t,:] = 0 for
 *   t >= flen[b].  tmask/fmask uint8 (NULL = none), embed bf16 [C].
 * ca_regroup_pad: x [B,T,G*Cg] -> xg [B,G,T+2*pad,Cg] with zero time padding (layout for
 *   the grouped positional conv as an overlapping-row GEMM, :326-368).
 * ca_posconv_weight: weight-norm (dim=2) + reorder for the implicit GEMMs:
 *   w = g * v / ||v||_(0,1);  wf[g][co][j][ci] = w[g*Cg+co, ci, j]  (forward)
 *   wb[g][ci][j][co] = w[g*Cg+co, ci, K-1-j]                        (data gradient)
 *   v fp32 [d, Cg, K], g fp32 [K]; norm fp32 [K] saved.
 * ca_posconv_weight_bwd: dwf fp32 [G][Cg][K][Cg] -> dv, dg (+=).
 * ---------------------------------------------------------------------------------- */
int ca_mask_frames(void* h, const uint8_t* tmask, const uint8_t* fmask, const void* embed,
                   const int32_t* flen, int32_t B, int32_t T, int32_t C, void* stream);
int ca_regroup_pad(const void* x, void* xg, int32_t B, int32_t T, int32_t G, int32_t Cg,
                   int32_t pad, void* stream);
int64_t ca_posconv_partial_floats(int32_t K); /* size of `partial` below, in floats */
int ca_posconv_weight(const float* v, const float* g, void* wf, void* wb, float* norm,
                      float* partial, int32_t d, int32_t Cg, int32_t K, void* stream);
int ca_posconv_weight_bwd(const float* dwf, const float* v, const float* g,
                          const float* norm, float* dv, float* dg, float* partial, int32_t d,
                          int32_t Cg, int32_t K, void* stream);

/* casts / transposes used when refreshing bf16 compute copies from fp32 masters */
int ca_cast_f32_bf16(const float* x, void* y, int64_t n, void* stream);
int ca_cast_bf16_f32(const void* x, float* y, int64_t n, void* stream);
/* y[c*rows + r] = x[r*cols + c] with cast; x fp32 [rows, cols] -> y bf16 [cols, rows] */
int ca_transpose_f32_bf16(const float* x, void* y, int32_t rows, int32_t cols, void* stream);
/* conv weight reorder: w fp32 [Co,Ci,k] -> wr bf16 [Co][k][Ci] */
int ca_conv_weight_reorder(const float* w, void* wr, int32_t Co, int32_t Ci, int32_t k,
                           void* stream);
/* inverse for the gradient: dwr fp32 [Co][k][Ci] -> dw fp32 [Co,Ci,k] (+=) */
int ca_conv_weight_grad_reorder(const float* dwr, float* dw, int32_t Co, int32_t Ci,
                                int32_t k, void* stream);

/* ------------------------------------------------------------------------------------
 * Optimiser step.  $TF/trainer.py:1778-1796 (clip_grad_norm_ 1.0, AdamW β=(0.9,0.98))
 * ← R/src/coral/wav2vec2.py:216-240, R/config/asr_finetuning.yaml:64-75.
 * Flat fp32 buffers.  ca_sumsq: out[0] (+)= sum g^2 (partial: >= 4096 floats).
 * ca_adamw_step: reads *gnorm_sq (device), clip = min(1, max_norm/(sqrt+1e-6)); updates
 * p,m,v in place and writes the bf16 compute copy p16 (may be NULL). grad_scale multiplies
 * the gradient first (1/world for DDP-mean semantics).
 * ---------------------------------------------------------------------------------- */
int ca_sumsq_f32(const float* g, int64_t n, float* out, int32_t accumulate, float* partial,
                 void* stream);
/* Zero a list of ranges of one device buffer in ONE launch: the per-step `optimizer.zero_grad()` of
 * $TF/trainer.py:1796 restricted to what the next backward accumulates into (the weight-matrix gradients are
 * overwritten by their GEMMs), and the engines' other per-step clears.  ranges_bytes: device array of nranges
 * (offset, length) pairs in BYTES relative to base, multiples of 4; max_bytes: the longest range (sizes the grid). */
int ca_clear_ranges(void* base, const int64_t* ranges_bytes, int32_t nranges, int64_t max_bytes, void* stream);
/* out[0] (+)= sum of g^2 over the listed chunks: chunks = device array of nchunks (offset, length) pairs in floats,
 * offsets multiples of 4; partial: >= nchunks floats.  ca_sum_f32: out[0] (+)= sum x[i] in a fixed order (partial: >= 256
 * floats) - the per-tile partials CaGemmDesc.c_sumsq collects.  Together they are clip_grad_norm_'s squared norm
 * ($TF/trainer.py:1778-1796) without a pass over the weight-matrix gradients. */
int ca_sumsq_ranges_f32(const float* g, const int64_t* chunks, int32_t nchunks, float* out, int32_t accumulate,
                        float* partial, void* stream);
int ca_sum_f32(const float* x, int64_t n, float* out, int32_t accumulate, float* partial, void* stream);
int ca_adamw_step(float* p, float* m, float* v, const float* g, void* p16, int64_t n,
                  float lr, float beta1, float beta2, float eps, float weight_decay,
                  int32_t step, float grad_scale, float max_norm, const float* gnorm_sq,
                  void* stream);
/* The same update with the grid capped at max_blocks 256-thread workgroups (0 = no cap).  One workgroup per CU lets the
 * update run UNDER the next step's forward GEMMs (its waves fit into the registers they leave free) instead of
 * alternating with them - what the trainer passes for the per-bucket updates it overlaps with the forward
 * (coral_amd/trainer.py; $TF/trainer.py:1796 optimizer.step has no such notion: the result is the same bits). */
int ca_adamw_step_ex(float* p, float* m, float* v, const float* g, void* p16, int64_t n,
                     float lr, float beta1, float beta2, float eps, float weight_decay,
                     int32_t step, float grad_scale, float max_norm, const float* gnorm_sq,
                     int32_t max_blocks, void* stream);
/* The same update reading the gradient as a bf16 tensor: under the reference's bf16 autocast the weight gradient of a
 * Linear IS bf16 (the matmul's output dtype; cast to fp32 only when it lands in .grad), so a single-micro-batch step can
 * keep the weight-matrix gradients in bf16 from the weight-gradient GEMM's epilogue to here (coral_amd/trainer.py). */
int ca_adamw_step_g16(float* p, float* m, float* v, const void* g_bf16, void* p16, int64_t n,
                      float lr, float beta1, float beta2, float eps, float weight_decay,
                      int32_t step, float grad_scale, float max_norm, const float* gnorm_sq,
                      int32_t max_blocks, void* stream);
/* Does a wave of the update kernel fit into the registers the forward GEMMs' workgroups (two waves per SIMD) leave free?
 * The capped-grid update (max_blocks = the CU count) only runs UNDER the next step's forward while it does
 * (2 x registers of the forward kernel + the update's <= 512 per SIMD, in allocation granules of 8); one register too
 * many in the forward kernel and the capped update starves instead (+ 19 ms per XLS-R-2B step, round 5) - the trainer
 * asks at start-up and falls back to the full-grid update.  regs (optional, 2 words): registers per lane of the forward
 * GEMM kernel and of the update kernel as the loaded code object reports them.  Returns 1 / 0, negative on an error. */
int ca_background_update_fits(int32_t* regs);

/* ------------------------------------------------------------------------------------
 * Whisper log-mel front end.  $TF/models/whisper/feature_extraction_whisper.py:135-168
 * (torch.stft n_fft 400 hop 160 hann, reflect pad, |.|^2 of frames[:-1], mel, log10 clamp
 * 1e-10, max-8 floor, (x+4)/4) ← R/src/coral/whisper.py:51-55, R/src/coral/data.py:747.
 * wave fp32 [B,N] (already padded/truncated to N = 480000), mel_filters fp32 [201,n_mels],
 * out fp32 [B,n_mels,frames], frames = N/160.  ws: ca_logmel_workspace_bytes(B).
 * ---------------------------------------------------------------------------------- */
int64_t ca_logmel_workspace_bytes(int32_t B);
int ca_logmel(const float* wave, const float* mel_filters, float* out, void* ws, int32_t B,
              int64_t N, int32_t n_mels, void* stream);

/* cross-entropy over the vocabulary with ignore_index (-100), fp32 logits [rows, ldv]:
 * $TF/models/whisper/modeling_whisper.py:1084-1087.  loss_sum fp32[1] += sum nll,
 * count int32[1] += #valid; grad fp32 [rows, ldv] = (softmax - onehot) (unscaled). */
int ca_cross_entropy_fwd_bwd(const float* logits, const int32_t* labels, float* loss_sum,
                             int32_t* count, float* grad, int64_t rows, int32_t V,
                             int64_t ldv, int32_t ignore_index, void* stream);
/* masked argmax over the vocabulary for greedy generation
 * ($TF/models/whisper/generation_whisper.py:1774-1812 suppress processors + argmax):
 * out[r] = argmax_v (logits[r,v] if !suppress[v]); suppress uint8 [V] or NULL. */
int ca_argmax_masked(const float* logits, const uint8_t* suppress, int32_t* out,
                     int64_t rows, int32_t V, int64_t ldv, void* stream);
/* The same argmax and, in the same launch, the bookkeeping of a greedy step
 * ($TF/generation/utils.py `_sample`: finished rows take pad, `unfinished_sequences` falls on eos): with t = done[r] ?
 * pad_id : out[r]:  ids[r, pos[r] + 1] = t;  done[r] |= t == eos_id;  tok[r] = t;  pos[r] += 1;  klen[r] += 1.  All state
 * lives in device memory, so the step replays as part of a graph. */
int ca_argmax_advance(const float* logits, const uint8_t* suppress, int32_t* out, int64_t rows, int32_t V, int64_t ldv,
                      uint8_t* done, int64_t* ids, int64_t ld_ids, int32_t* tok, int32_t* pos, int32_t* klen,
                      int32_t pad_id, int32_t eos_id, void* stream);
/* embedding gather: y[r,:] = table[ids[r],:] + pos[pos_ids[r],:]  (bf16 tables)
 * $TF/models/whisper/modeling_whisper.py:204-212,676. */
int ca_embed_tokens(const void* table, const void* pos, const int32_t* ids,
                    const int32_t* pos_ids, void* y, int64_t rows, int32_t C, void* stream);
/* its backward: dtable[ids[r],:] += dy[r,:], dpos[pos_ids[r],:] += dy[r,:]  (fp32 tables, bf16 dy) */
int ca_embed_tokens_bwd(const void* dy, const int32_t* ids, const int32_t* pos_ids, float* dtable,
                        float* dpos, int64_t rows, int32_t C, void* stream);

/* ------------------------------------------------------------------------------------
 * Greedy decoding: ONE launch per token for a whole Whisper decoder (round 6).
 *
 * Replaces, for a batch of at most 16 clips, the ~170 dependent launches of a decoded token - per layer
 * `WhisperDecoderLayer.forward` ($TF/models/whisper/modeling_whisper.py:451-518: self_attn_layer_norm, q|k|v, cached
 * self-attention, out_proj + residual, encoder_attn_layer_norm, q_proj, attention over the cached encoder K|V, out_proj
 * + residual, final_layer_norm, fc1 + GELU, fc2 + residual), then `layer_norm`, the tied `proj_out` and the greedy pick
 * of `generate` ($TF/generation/utils.py `_sample`; call sites R/src/coral/evaluate.py:56-60, R/config/evaluation.yaml:20) -
 * by a persistent kernel of one 256-thread workgroup per CU.  The phases of a layer are separated by all-to-all seams
 * inside the launch (every workgroup publishes a progress word with a write-through store behind its drained
 * write-through payload stores; one wave per workgroup polls all of them with one 16-byte load per lane; every spin is
 * bounded); each workgroup's slices of the weight matrices arrive in LDS ahead of the seam that needs them (a ring per
 * wave, LDS-DMA), and the activations every workgroup needs are a few tens of KB from L2.  A dependent launch costs
 * ~4.7 us on this chip before it moves a byte, a seam ~3.2 us (tools/r06/seam_bench.hip).
 *
 * Per output element the arithmetic is that of the launch sequence it replaces (ca_gemm_bf16's weight-streaming form
 * with the LayerNorm prologue, ca_attn_fwd's single-query form with the same key split, ca_argmax_advance): logits and
 * token ids are bit-identical to it.
 *
 * All pointers are DEVICE pointers.  `layers`: n_layers records in device memory.  Weights bf16 row-major [N, K] with
 * row stride K; biases / LayerNorm vectors fp32 (16-byte aligned); a NULL bias reads as zeros.  State as for
 * ca_argmax_advance.  `ws`: CA_DECODE_WS_BYTES(B, d, f, H, n_layers) bytes of workspace (16-byte aligned, any contents);
 * `status`: 4 words the CALLER zeroes once; a launch only ever raises word 0 (0 = ok, else 1 + the phase whose seam timed
 * out - sticky over a series of launches; the host reads it after synchronising: a launch that gave up has written no token).
 * A launch that finds every clip finished (`done`) records pad for each, advances the positions and returns at once - a
 * host may queue launches ahead of its own all-finished check.
 * Limits: B <= 16, head_dim 64, d_model a multiple of 64 up to 1536, at most 256 CUs, and every CU of the device free
 * (the workgroups must all be resident: nothing else may run beside the launch).
 * ---------------------------------------------------------------------------------- */
typedef struct CaDecodeLayer {
  const float *ln1_g, *ln1_b;
  const void* wqkv;   /* [3d, d]: q | k | v rows */
  const float* bqkv;  /* [3d] (zeros in the k part) */
  const void* wo;
  const float* bo;
  const float *ln2_g, *ln2_b;
  const void* wq2;
  const float* bq2;
  const void* wo2;
  const float* bo2;
  const float *ln3_g, *ln3_b;
  const void* w1;     /* [f, d] */
  const float* b1;
  const void* w2;     /* [d, f] */
  const float* b2;
  void* self_kv;        /* bf16 [B, max_len, 2d]: K | V of the tokens decoded so far; this token's row is written */
  const void* cross_kv; /* bf16 [B, Te, 2d]: K | V projections of the encoder states */
} CaDecodeLayer;
typedef struct CaDecodeDesc {
  const CaDecodeLayer* layers; /* device array */
  int32_t n_layers, B, d, f, H, Te, max_len, V;
  const void *embed, *embed_pos; /* bf16 [V, d] (also the output projection), [max positions, d] */
  const float *lnf_g, *lnf_b;
  float eps;
  float* logits;      /* fp32 [B, ld_logits]: written for every row */
  int64_t ld_logits;
  const uint8_t* suppress; /* [V] or NULL */
  int32_t* out;       /* [B] the tokens picked */
  uint8_t* done;
  int64_t* ids;
  int64_t ld_ids;
  int32_t *tok, *pos, *klen;
  int32_t pad_id, eos_id;
  void* ws;
  int64_t ws_bytes;
  uint32_t* status;
  /* 1: CaDecodeLayer.cross_kv is the head-major copy [B][2 (K, V)][H][Te][64] of the encoder K|V (a (clip, head)'s keys
   * and values are two contiguous strips: the cross-attention streams them without 4 KB strides); 0: [B, Te, 2d] */
  int32_t cross_head_major;
} CaDecodeDesc;
#define CA_DECODE_MAX_B 16
#define CA_DECODE_WS_BYTES(B, d, f, H, n_layers)                                                              \
  (8192 + (int64_t)(n_layers) * 16 * (H) * 4 + (int64_t)16 * (7 * (int64_t)(d) + (f)) * 2 +                     \
   (int64_t)16 * (H) * CA_ATTN_SPLIT_MAX * 16 * 66 * 4 + 256 * 16 * 8 + 4096)
int ca_whisper_decode_token(const CaDecodeDesc* desc, void* stream);
/* 1 when ca_whisper_decode_token takes this shape on the current device, else 0 (the caller keeps the launch sequence) */
int ca_whisper_decode_token_supported(int32_t B, int32_t d, int32_t f, int32_t H, int32_t V);
/* debug / measurement: launches of ca_whisper_decode_token from now on leave per-workgroup phase stamps (shader clock) in
 * device_buf, [CUs][nph][2] 64-bit words (seam passed, phase done); NULL switches them off (tools/r06/persist_stamps.py) */
int ca_debug_decode_stamps(void* device_buf, int32_t nph);

/* ------------------------------------------------------------------------------------
 * Data-parallel exchange: bucket collectives over RCCL (xGMI), one context per rank.  Replaces what the reference gets
 * from accelerate's DDP wrapper / DeepSpeed ZeRO-2 under `Trainer` (accelerate/accelerator.py:1892 prepare_model ->
 * DistributedDataParallel, :2053 backward; launch lines R/makefile:79-84): a SUM all-reduce per gradient bucket
 * overlapped with the backward, or - sharded optimiser - a reduce-scatter of the gradients and an all-gather of the
 * updated bf16 weights.  (`cm_allreduce_bucket(ctx, ...)`.)
 *
 * A context owns an RCCL communicator and ONE HIP stream on which all of its collectives are enqueued, in call order;
 * nothing here synchronises the host.  Rendezvous is the caller's: rank 0 calls ca_comm_unique_id and hands the 128
 * bytes to every rank (any channel), then every rank calls ca_comm_init with the device it computes on current.
 * RCCL is bound at run time on the first ca_comm_* call (CA_ERR_UNSUPPORTED when librccl.so.1 cannot be found); the
 * rest of the library does not depend on it.
 *   ca_comm_after(ctx, producer):  the communication stream waits for the work enqueued so far on `producer`
 *                                  (the bucket's gradients are complete);
 *   ca_comm_before(ctx, consumer): `consumer` waits for the collectives enqueued so far (the optimiser may read).
 * Buffers are DEVICE pointers; all three collectives work IN PLACE:
 *   ca_allreduce_bucket:      buf[0:n] <- sum over ranks;
 *   ca_reduce_scatter_bucket: rank r ends with the sum over ranks of buf[r * n_per_rank : (r + 1) * n_per_rank] in
 *                             that slice (the other slices are left undefined);
 *   ca_allgather_bucket:      buf[r * n_per_rank : (r + 1) * n_per_rank] of every rank r -> all of buf on every rank.
 * ---------------------------------------------------------------------------------- */
#define CA_COMM_ID_BYTES 128
#define CA_COMM_F32 0
#define CA_COMM_BF16 1
typedef struct CaComm CaComm;
int ca_comm_unique_id(void* id128_h);
int ca_comm_init(CaComm** ctx, const void* id128_h, int32_t rank, int32_t world);
int ca_comm_destroy(CaComm* ctx); /* waits for the context's stream, then frees communicator, stream and context */
int ca_comm_abort(CaComm* ctx);   /* the same WITHOUT waiting (ncclCommAbort): for a context whose peers never joined a
                                   * collective - the rendezvous fallback of the host side (coral_amd/trainer.py) */
void* ca_comm_stream(CaComm* ctx); /* the context's hipStream_t */
int ca_comm_rank(CaComm* ctx);
int ca_comm_world(CaComm* ctx);
int ca_comm_after(CaComm* ctx, void* producer_stream);
int ca_comm_before(CaComm* ctx, void* consumer_stream);
int ca_allreduce_bucket(CaComm* ctx, void* buf, int64_t n, int32_t dtype);
int ca_reduce_scatter_bucket(CaComm* ctx, void* buf, int64_t n_per_rank, int32_t dtype);
int ca_allgather_bucket(CaComm* ctx, void* buf, int64_t n_per_rank, int32_t dtype);

#ifdef __cplusplus
}
#endif
#endif /* CORAL_AMD_H */
