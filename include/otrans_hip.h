/* libotrans_hip.so -- C ABI of the MI355X (gfx950) hot path for ZhengkunTian/OpenTransformer.
 *
 * The reference has no FFI (it is pure PyTorch); every entry below names the reference code it
 * replaces (paths relative to the reference root).  The reference-side binding a maintainer would
 * add is the ctypes stub shown in INTEGRATION.md (opentransformer_amd/_lib.py is that stub).
 *
 * Conventions:
 *  - plain pointers + sizes; the CALLER owns every buffer (inputs, outputs, saved-for-backward,
 *    workspace).  The library never allocates or frees device memory and keeps no pointer after
 *    a call returns.
 *  - every launch is asynchronous on the hipStream_t passed in (as void*); no hidden syncs, so all
 *    entries are legal inside hipGraph stream capture.
 *  - return 0 = OK, negative = bad argument (nothing launched), positive = hipError_t.
 *    otr_last_error_string() describes the last failure on the calling thread.
 *  - dtype codes: OTR_F32 = 0, OTR_BF16 = 1 (raw bf16 bits), OTR_F16 = 2 (IEEE binary16 bits).  The library is built
 *    twice from one source: libotrans_hip.so has bf16 as its 16-bit storage / MFMA-input type and accepts OTR_BF16,
 *    libotrans_hip_f16.so has fp16 and accepts OTR_F16 (otr_half_type() says which; the other code is rejected).
 *    Below, "bf16" in a parameter name or comment means "the 16-bit type of the build".  fp16 carries 3 more mantissa
 *    bits (logits 5e-4 from the fp32 reference instead of 3.9e-3, tools/precision_study.py) at the same MFMA rate;
 *    its gradients need loss scaling (see otr_optimizer_step).  `compute` selects the MFMA type: the 16-bit code ->
 *    v_mfma_f32_{16x16x32,32x32x16}_{bf16,f16} (fp32 accumulate), OTR_F32 -> v_mfma_f32_16x16x4_f32 (exact fp32,
 *    the parity mode).
 */
#ifndef OTRANS_HIP_H
#define OTRANS_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define OTR_F32 0
#define OTR_BF16 1
#define OTR_F16 2
#define OTR_ACT_NONE 0
#define OTR_ACT_RELU 1

/* ABI version of THIS header: bumped whenever a signature, a descriptor struct or the meaning of an argument changes (600: round 6;
 * 300 was rounds 3-5, during which otr_optimizer_step, otr_ln_desc_t, otr_wgrad_item_t and otr_beam_prune_cached changed without a
 * bump).  A binding compares otr_version() with the OTR_ABI_VERSION it was written against BEFORE its first call and refuses a
 * library that answers anything else: descriptors are passed by pointer and read at the library's idea of their size. */
#define OTR_ABI_VERSION 601
int32_t otr_version(void);
/* OTR_BF16 or OTR_F16: the 16-bit type this library was built for */
int32_t otr_half_type(void);
/* tuning hook for benchmarks: key 0 = force GEMM tile (0 auto / 64 / 128), key 1 = force split-K (0 auto),
 * key 2 = 1: generic (bounds-checked) loaders only, key 3 = 1: no persistent tile loop, key 4 = ablation bits of the split FFN
 * kernels (ffn3.hip: 1 no weight DMA, 2 no MFMA, 4 no tile stores, 8 no tile loads, 16 clock stamps into otr_debug_trace's buffer;
 * timing only), key 5 retired, key 6 = 0/1: 256-wide
 * weight-gradient launch off / on (-1: environment OTR_WGRAD256), key 7 = its workgroup count (0 = one per CU), key 8 = its ablation / cache
 * policy switches (wgrad256.h), key 9 = the shortest contraction it takes, key 10 = ablations of otr_conv2_dgrad (1 = no mask loads /
 * result stores, 2 = every operand load from one line: timing only, results are garbage), key 11 = bound of every in-kernel
 * turnstile / arrival spin (<= 0: the default 2^22; tests force a give-up with 1), key 12 = 1: the split FFN kernels exchange their
 * partial sums with write-through stores / memory-served loads only (the path a row block split across XCDs takes), key 13 = 1:
 * attention backward as two launches (dQ, then dK/dV) instead of one, key 15 = workgroup mapping of the split FFN kernels (1, the
 * default: an XCD owns one weight slice; 0: the four slices of a row block share an XCD), key 16 = workgroup mapping of the
 * attention launches (1, the default: the blocks of one (head, utterance) on one XCD; 0: the plain 3-D grid), key 17 = 1: conv1
 * forward on the VALU stencil for every shape (default: the fp32 matrix pipe for 64 channels and 16-bit activations), key 18 = 0:
 * the 768-column row-block projection on one workgroup per row block (default: two, 384 columns each), key 19 = 0: 4-wave workgroups
 * for the 256-column row-block kernels (default: 8 waves), key 20 = 0: 4-wave (64 queries / keys) workgroups for the attention launches
 * (default: 8 waves, 128 queries / keys, for aligned 16-bit operands with head dim 64), key 21 = 0: the streamed attention backward instead of
 * csrc/encattn.hip, key 22 = conv2 forward on the weight-stationary kernel (1) or the implicit GEMM (0), key 23 = utterances per workgroup of
 * the fused decoder launches (0 = the library's choice), key 24 / 25 = forms of the cached decode self-attention / the beam top-k (25 = 2: the
 * arg-max rounds), key 26 = 0: natural GEMM tile order, key 27 = 0: scalar loads of the relative-position score term, key 28 = resident
 * workgroups of the persistent 64 x 64-tile GEMM (1024; 512 = round 5), key 29 = 0: conv2's weight gradient of a 256-channel frontend on the
 * transposing GEMM instead of the gathered-row form of wgrad256.hip, key 30 = 0: no sliced parity-class input gradient for 256 output channels,
 * key 31 = 0: otr_conv2_dgrad_wide answers "not served", key 32 = ablation bits of csrc/conv2wide.hip (1 no MFMAs, 2 one fragment read per chunk,
 * 4 no weight DMA, 8 no row reloads), key 33 = csrc/encattn96.hip: bit 0 = it serves (0: the streamed dQ + dK/dV pair), bits 1-3 = its ablations
 * (2 no score-term loads, 4 no d bias stores, 8 no tiles), key 34 = 0: conv2 forward's implicit-im2col operand on the bounds-checked loader, key 37 = the narrowest x operand the 256-wide weight-gradient
 * launch takes (96; 128 = round 5).
 * Ablations are for timing only: results are garbage. */
int32_t otr_debug_set(int32_t key, int32_t value);
/* Register the caller-owned, zero-initialised DEVICE word that spin-bounded kernels (the turnstile of the 256-wide
 * weight-gradient launch; NULL = none) add 1 to whenever a wait gives up -- the results of such a launch may be wrong sums.
 * otr_optimizer_step reads and clears it: a non-zero count skips the update exactly like a non-finite gradient norm and is
 * accumulated in state[10].  Process-wide like the compute type; the pointer is baked into captured graphs, so register it
 * before capturing and keep the word alive.  Replaces nothing in the reference (train/trainer.py:229 only guards NaNs). */
int32_t otr_set_fault_counter(void* device_word);
/* hardware probe used by the tests of the 256-wide weight-gradient kernel: one wave copies image[2048] (16-bit words)
 * to LDS and issues ONE ds_read_b64_tr_b16 with lane l at byte address addr[l]; out[l*4 + j] = element j lane l got. */
int32_t otr_debug_trread(const void* image, const int32_t* addr, void* out, void* stream);
/* tuning hook: when buf != NULL every GEMM workgroup writes 4 shader-clock timestamps (start, operands staged,
 * k-loop done, stores issued) to buf[(blockIdx.y*gridDim.x + blockIdx.x)*4 ..]; otr_conv2_dgrad writes, per workgroup, the 100 MHz
 * real time at start / operand fragments built / end and its parity class; NULL disables.  Not for production. */
int32_t otr_debug_trace(void* buf);
const char* otr_last_error_string(void);

/* ---- nn.Linear and its gradients (module/attention.py:43,68,128-129; module/ffn.py:39-41;
 *      frontend/conv.py:146; decoder/transformer.py:181; model/ctc.py:47).
 * All matrices row-major with leading dimensions in elements.  `accumulate` != 0 adds into out.
 * workspace (fp32, caller owned, may be NULL): enables split-K for contraction-heavy / output-small
 * shapes; partial [M,N] slabs are reduced in a FIXED order by a second kernel, so results stay
 * deterministic.  64 MiB covers every shape of the AISHELL configs. */
typedef struct {
  int32_t M, N, K;              /* y[M,N] = x[M,K] * w[N,K]^T */
  int32_t x_dtype, w_dtype, y_dtype, compute;
  int64_t ldx, ldw, ldy;
  int32_t act;                  /* OTR_ACT_* applied after bias (forward only) */
  int32_t accumulate;
} otr_linear_desc_t;
/* y = act(x w^T + bias); bias f32[N] or NULL */
int32_t otr_linear_fwd(const otr_linear_desc_t* d, const void* x, const void* w, const float* bias, void* y,
                       void* workspace, int64_t workspace_bytes, void* stream);
/* dx[M,K] (dtype x_dtype, ld ldx) = dy[M,N] (dtype y_dtype, ld ldy) * w[N,K] */
/* otr_linear_fwd_batched: nbatch products of ONE shape in one launch, y_b = x_b w_b^T with x_b = x + b*bsx, w_b = w + b*bsw,
 * y_b = y + b*bsy (ELEMENT strides): the per-head products of module/attention.py:217-253 (MultiHeadedSelfAttentionWithRelPos:
 * (q+v) p^T per head, and its input gradient).  No bias, activation, accumulation or split-K.  Returns 1 WITHOUT launching when
 * the operands do not qualify (fp32 compute, unaligned rows / strides, operand types other than 16-bit x 16-bit -> f32 and
 * f32 x 16-bit -> 16-bit): the caller then loops over otr_linear_fwd. */
int32_t otr_linear_fwd_batched(const otr_linear_desc_t* d, const void* x, const void* w, void* y, int32_t nbatch, int64_t bsx, int64_t bsw,
                               int64_t bsy, void* stream);
int32_t otr_linear_dgrad(const otr_linear_desc_t* d, const void* dy, const void* w, void* dx, void* workspace,
                         int64_t workspace_bytes, void* stream);
/* dw[N,K] (dtype w_dtype, ld ldw) = dy[M,N]^T * x[M,K] */
int32_t otr_linear_wgrad(const otr_linear_desc_t* d, const void* dy, const void* x, void* dw, void* workspace,
                         int64_t workspace_bytes, void* stream);
/* out[N] (f32) (+)= sum over M rows of a[M,N]  (bias gradients) */
int32_t otr_colsum(const void* a, int32_t dtype, int64_t M, int64_t N, int64_t lda, float* out, int32_t accumulate,
                   void* stream);

/* ---- FFN forward through w_1 and the GLU in ONE launch (module/ffn.py:38-40): t[M,2F] = x[M,d] . w1[2F,d]^T + b1,
 *      u[M,F] = t[:, :F] * sigmoid(t[:, F:]); h[M,2F] = (t[:, :F] | sigmoid(t[:, F:])) is kept for backward (the
 *      sigmoid is computed once; otr_ffn_glu_bwd / otr_glu_bwd take it with h_has_sigmoid = 1).  Each GEMM tile holds 64 value columns and their 64 gate
 *      columns, so the GLU runs in the epilogue on data that is already on chip.  bf16 operands and outputs.
 *      Returns 1 without launching anything when the operands do not qualify; use otr_linear_fwd + otr_glu_fwd then. */
int32_t otr_ffn_glu_fwd(const void* x, int64_t ldx, const void* w1, int64_t ldw, const float* b1, void* h, void* u, int32_t M,
                        int32_t F, int32_t d_model, void* stream);
/* ---- FFN backward through w_2 and the GLU in ONE launch (module/ffn.py:38-41 backward): du = dy[M,d] . w2 (w2t = the
 *      [F,d] transposed bf16 shadow of w_2; du is never stored), dh[M,2F] = GLU'(h) * du with h[M,2F] the saved GLU
 *      input (h_has_sigmoid != 0: its second half already holds sigmoid(gate), as otr_ffn_glu_fwd writes it), and
 *      dbias_partial[rows, 2F] = column sums of dh per row tile (column-sum them for the w_1 bias gradient;
 *      *partial_rows receives the number of rows written, <= partial_rows_cap, size it ceil(M/64)).  bf16 operands.
 *      Returns 1 without launching anything when the operands do not qualify (alignment); use otr_linear_dgrad +
 *      otr_glu_bwd then. */
int32_t otr_ffn_glu_bwd(const void* dy, int32_t dy_dtype, int64_t ldy, const void* w2t, int64_t ldw, const void* h,
                        int32_t h_has_sigmoid, void* dh, float* dbias_partial, int32_t partial_rows_cap,
                        int32_t* partial_rows, int32_t M, int32_t F, int32_t d_model, void* stream);

/* ---- row-block fused FFN sub-layer of the post-norm layers (encoder/transformer.py:58-63, decoder/transformer.py:82-86,
 *      module/ffn.py:38-41 with activation 'glu'; SURVEY.md K7 + K8), d_model = 256, d_ff % 256 == 0, 16-bit operands.
 *      A workgroup owns 32 rows and streams the PACKED weights (otr_pack_frags) from L2 into registers; the d_ff-wide
 *      hidden lives in MFMA accumulators only.
 * otr_pack_frags: fragment-major copies of 16-bit matrices, many per launch.  table: DEVICE int64 [n_items][8] rows of
 *   {src element offset, row stride, col stride, rows (%32), cols (%16), perm, dst element offset, first block}; item i
 *   views src+offset as A[r][c] = src[r*rs + c*cs] (r = output index, c = contraction index) and writes one 1 KiB MFMA
 *   A-operand per (32 rows, 16 contraction indices) in (row tile, k-step) order; perm = 1 orders the contraction
 *   indices the way an accumulator tile presents them when it is fed back as an operand.  A block packs 4 fragments;
 *   total_blocks = sum of ceil(rows/32 * cols/16 / 4).
 * otr_ffn_ln_fwd:  y = LayerNorm(x + dropout(w_2(glu(w_1 x + b_1)) + b_2)) in ONE launch.  x f32 [M,256] and its 16-bit
 *   twin x16; w1_pack = pack(w_1 [2F,256] as A[f][k], perm 0); w2_pack = pack(w_2 [256,F] as A[n][f], perm 1).
 *   Outputs as otr_add_layernorm_fwd: y, y16 (may be NULL), z = x + dropout(branch) (may be NULL), mean, rstd; the
 *   dropout mask is the one otr_add_layernorm_bwd regenerates from (seed, rng_offset).
 * otr_ffn_bwd:  recomputes the pre-activations from x16, then dh[M,2F] = GLU'(.) * (dy16 . w_2), u[M,F] = glu(.)
 *   (row-major, the operands of the two weight gradients) and dx[M,256] f32 = skip + dh . w_1 in ONE launch.
 *   w2t_pack = pack(w_2 as A[f][n], perm 0); w1t_pack = pack(w_1 as A[k][f'], perm 1); skip may be NULL or alias dx.
 *   db1_part f32 [ceil(M/32)][2F]: column sums of dh per 32-row block (column-sum them for the w_1 bias gradient; the
 *   reduction over each block's rows happens on the accumulator registers, dh is not read again). */
int32_t otr_pack_frags(const void* src, void* dst, const int64_t* table, int32_t n_items, int64_t total_blocks,
                       void* stream);
int32_t otr_ffn_ln_fwd(const float* x, const void* x16, const void* w1_pack, const float* b1, const void* w2_pack,
                       const float* b2, const float* gamma, const float* beta, const uint64_t* seed, float p_drop,
                       uint64_t rng_offset, float eps, float* y, void* y16, float* z, float* mean, float* rstd, int64_t M,
                       int32_t F, int32_t d_model, void* stream);
int32_t otr_ffn_bwd(const void* x16, const void* dy16, const void* w1_pack, const float* b1, const void* w2t_pack,
                    const void* w1t_pack, void* dh, void* u, float* db1_part, const float* skip, float* dx, int64_t M,
                    int32_t F, int32_t d_model, void* stream);

/* ---- row-block projections of the attention sub-layers (module/attention.py:62-75 qvk_proj / output_proj, :120-140 q_proj /
 *      output_proj with the residual + LayerNorm of encoder/transformer.py:47-56, decoder/transformer.py:58-80), d_model = 256,
 *      16-bit operands.  Same structure as the fused FFN: 32 rows per workgroup, packed weights (otr_pack_frags, perm 0)
 *      streamed from L2, whole-row epilogues.
 * otr_rb_linear:  out[M,N] = x16[M,K] . W^T (+ bias) (+ skip), (N,K) in {(256,256), (768,256), (256,768)}; w_pack =
 *   pack(W as A[n][k]); for an input gradient pass dy16 as x16 and pack(W as A[k][n]) (rows = K inputs).  out f32 or 16-bit.
 * otr_proj_ln_fwd:  y = LayerNorm(x + dropout(c16 . W^T + bias)); outputs as otr_add_layernorm_fwd (y16, z may be NULL).
 * otr_ln_bwd_proj:  the LayerNorm backward of that sub-layer -- dx[M,256] f32 (may be NULL), da16 = dropout-masked gradient
 *   of the projection output (16-bit, the weight-gradient operand; may be NULL), partial f32
 *   [otr_ln_bwd_proj_partial_rows(M)][3][256] = per-workgroup sums of dgamma | dbeta | da (may be NULL) -- followed in
 *   the same launch by dc16 = da . W (16-bit, row stride ldc).  wt_pack = pack(W as A[k][n]). */
int32_t otr_rb_linear(const void* x16, int64_t ldx, const void* w_pack, const float* bias, const float* skip, int64_t lds,
                      void* out, int32_t out_dtype, int64_t ldo, int64_t M, int32_t N, int32_t K, void* stream);
int32_t otr_proj_ln_fwd(const float* x, const void* c16, int64_t ldc, const void* w_pack, const float* bias, const float* gamma,
                        const float* beta, const uint64_t* seed, float* y, void* y16, float* z, float* mean, float* rstd,
                        int64_t M, int32_t d_model, float eps, float p_drop, uint64_t rng_offset, void* stream);
int64_t otr_ln_bwd_proj_partial_rows(int64_t M);
/* otr_rb_linear_ln_bwd:  the input gradient of a projection whose INPUT is the output of a LayerNorm y = LN(z), z = x + dropout(a)
 *   (layer l's q|k|v projection reads layer l-1's FFN sub-layer: encoder/transformer.py:47-63), followed in the same launch by
 *   that LayerNorm's backward: dy = skip + g16[M,K] . W stays on chip; dx = d z f32 [M,256], da16 = dropout-masked branch
 *   gradient, partial as otr_ln_bwd_proj ([otr_ln_bwd_proj_partial_rows(M)][3][256]).  wt_pack = pack(W as A[k][n]); N = 256,
 *   K in {256, 768}. */
int32_t otr_rb_linear_ln_bwd(const void* g16, int64_t ldg, const void* wt_pack, const float* skip, int64_t lds, const float* z,
                             const float* mean, const float* rstd, const float* gamma, const uint64_t* seed, float p_drop,
                             uint64_t rng_offset, float* dx, void* da16, float* partial, int64_t M, int32_t N, int32_t K,
                             void* stream);
/* The same with a PREFETCH range: while the epilogue runs every workgroup touches its share of [prefetch, prefetch + prefetch_bytes)
 * (one dword per 64 bytes, no consumer) so that the NEXT launch finds those lines in the memory-side cache -- used for the two
 * input-gradient packs of the split FFN's backward launch that follows this one in the backward pass (3 MB it would otherwise
 * fetch from HBM in scattered 1 KiB pieces).  prefetch may be NULL (then exactly otr_rb_linear_ln_bwd). */
/* Ranges the NEXT otr_ln_bwd_proj / otr_ln_bwd_proj_slabs call of the calling thread has its kernel touch the same way (then the hint
 * is forgotten): the saved q|k|v and context the attention backward launch that follows it would otherwise fetch cold. */
/* Read [p, p + bytes) once and consume nothing: the lines are then in the memory-side cache for the launches that follow (the fused
 * decoder's packed weights, touched once before the stack's forward pass). */
int32_t otr_touch(const void* p, int64_t bytes, void* stream);
/* The start of a training step in one launch: buf[0..n) = 0 (the flat gradient buffer; f32, 16-byte aligned) and, when counter is not
 * NULL, counter[0] += inc (the device-resident dropout seed, advanced once per step so that forward and backward of one step draw
 * the same masks).  Replaces a fill launch + an 8-byte add launch (train/trainer.py:206-208 optimizer.zero_grad()). */
int32_t otr_zero_tick(float* buf, int64_t n, int64_t* counter, int64_t inc, void* stream);
int32_t otr_touch_hint(const void* p0, int64_t bytes0, const void* p1, int64_t bytes1);
int32_t otr_rb_linear_ln_bwd_pf(const void* g16, int64_t ldg, const void* wt_pack, const float* skip, int64_t lds, const float* z,
                                const float* mean, const float* rstd, const float* gamma, const uint64_t* seed, float p_drop,
                                uint64_t rng_offset, float* dx, void* da16, float* partial, int64_t M, int32_t N, int32_t K,
                                const void* prefetch, int64_t prefetch_bytes, void* stream);
int32_t otr_ln_bwd_proj(const float* dy, const float* z, const float* mean, const float* rstd, const float* gamma,
                        const uint64_t* seed, const void* wt_pack, float* dx, void* da16, void* dc16, int64_t ldc,
                        float* partial, int64_t M, int32_t d_model, float p_drop, uint64_t rng_offset, void* stream);

/* ---- the same sub-layer on 128-row workgroups (csrc/ffn3.hip): a workgroup owns 128 rows x 1/4 of the hidden units, the
 *      packed weights reach it ONCE through an LDS-DMA ring shared by its four waves (64 rows per wave, activations in the
 *      accumulator half of the register file), and the four workgroups of a row block exchange their fp32 partial sums
 *      through `scratch` (write-through stores, one arrival counter per row block in `sync`) and finish a quarter of the rows
 *      each.  d_ff % 256 == 0, d_model 256.
 *      scratch: >= otr_ffn_split_scratch_bytes(M) bytes, contents don't care; sync: >= otr_ffn_split_sync_ints(M) ints, ZERO
 *      before the first call and owned by these kernels afterwards (monotonic arrival counters: launches on one buffer must
 *      be stream-ordered).  The arrival wait is bounded (otr_debug_set(11, v)); a
 *      give-up is reported through otr_set_fault_counter and leaves wrong rows behind.  Needs the four workgroups of a row
 *      block co-resident: any grid on an otherwise idle GPU (they sit next to each other in dispatch order).
 * otr_ffn_ln_fwd_split: y = LayerNorm(x + dropout(w_2 glu(w_1 x + b_1) + b_2)), outputs as otr_ffn_ln_fwd.  With hsave / usave
 *      (both or neither) the pass also leaves what otr_ffn_bwd_split needs instead of recomputing it: hsave
 *      [otr_ffn_split_hsave_bytes(M, F) bytes] = (value + bias, sigmoid(gate)) of every hidden unit, 16-bit, in accumulator-tile
 *      order (opaque); usave [otr_ffn_split_padded_rows(M), F] 16-bit row-major = the glu output (operand of the w_2 weight
 *      gradient; rows past M are scratch).
 * otr_ffn_bwd_split: dh [otr_ffn_split_padded_rows(M), 2F] 16-bit row-major = GLU'(saved tiles, dy . w_2) (operand of the w_1
 *      weight gradient; its column sums are the w_1 bias gradient); dx = skip (or 0) + dh . w_1, f32 [M, 256], may alias skip. */
int64_t otr_ffn_split_scratch_bytes(int64_t M);
int64_t otr_ffn_split_sync_ints(int64_t M);
int64_t otr_ffn_split_hsave_bytes(int64_t M, int32_t F);
int64_t otr_ffn_split_padded_rows(int64_t M);
int32_t otr_ffn_ln_fwd_split(const float* x, const void* x16, const void* w1_pack, const float* b1, const void* w2_pack,
                             const float* b2, const float* gamma, const float* beta, const uint64_t* seed, float p_drop,
                             uint64_t rng_offset, float eps, float* y, void* y16, float* z, float* mean, float* rstd,
                             void* hsave, void* usave, void* scratch, int64_t scratch_bytes, int32_t* sync, int64_t sync_ints,
                             int64_t M, int32_t F, int32_t d_model, void* stream);
int32_t otr_ffn_bwd_split(const void* dy16, const void* hsave, const void* w2t_pack, const void* w1t_pack, void* dh,
                          const float* skip, float* dx, void* scratch, int64_t scratch_bytes, int32_t* sync, int64_t sync_ints,
                          int64_t M, int32_t F, int32_t d_model, void* stream);
/* ---- the same split FFN kernels in SLAB mode: no partial-sum exchange, no LayerNorm inside the launch.  The four hidden slices of
 *      a row block leave their shares as 16-bit slabs [4][M][256] and the launch that reads the sub-layer's output finishes
 *      y = LayerNorm(x + dropout(sum of slabs + b_2))  in its prologue (the in-launch exchange is a chain of far round trips: a third
 *      of the forward kernel's cycles; a launch boundary is cheaper).
 * otr_ffn_fwd_split_slab:  slabs = w_2[:, slice] glu(w_1[slice] x + b_1[slice]); hsave / usave as otr_ffn_ln_fwd_split.
 * otr_rb_linear_ln:        the q|k|v projection of the NEXT layer with that LayerNorm in its prologue: out[M,768] = y16 . W^T + bias,
 *      y and the LayerNorm's saved z / mean / rstd written on the way (otr_dec_ln_t, nslab <= 4).
 * otr_dec_ln (declared with the fused decoder below) is the LayerNorm alone, for an output nobody projects (the last encoder layer).
 * otr_ffn_bwd_split_slab:  dh as otr_ffn_bwd_split; slabs = the slices' shares dh[slice] . w_1[slice] of the input gradient.
 * otr_ln_bwd_proj_slabs:   otr_ln_bwd_proj with the LayerNorm's output gradient given as dskip f32 [M,256] + nslab (<= 4) such slabs. */
int32_t otr_ffn_fwd_split_slab(const void* x16, const void* w1_pack, const float* b1, const void* w2_pack, void* hsave, void* usave,
                               void* slabs, int64_t M, int32_t F, int32_t d_model, void* stream);
int32_t otr_ffn_bwd_split_slab(const void* dy16, const void* hsave, const void* w2t_pack, const void* w1t_pack, void* dh, void* slabs,
                               int64_t M, int32_t F, int32_t d_model, void* stream);
int32_t otr_ln_bwd_proj_slabs(const float* dskip, const void* slabs, int32_t nslab, const float* z, const float* mean, const float* rstd,
                              const float* gamma, const uint64_t* seed, const void* wt_pack, float* dx, void* da16, void* dc16, int64_t ldc,
                              float* partial, int64_t M, int32_t d_model, float p_drop, uint64_t rng_offset, void* stream);
/* ---- grouped weight / bias gradients: every dw_i[N,K] += dy_i[M,N]^T x_i[M,K] of a backward pass in a few launches
 *      (one per operand-type group), likewise every bias gradient out_i[N] += column sums of a_i[M,N].  The per-layer
 *      launches they replace are latency bound (a 4-k-step GEMM workgroup lives ~10 us, every launch costs 2-3 us);
 *      grouped, the long-contraction tiles of all layers fill the chip at once, so no split-K, no workspace, no
 *      reduce kernels.  Items that miss the alignment rules of the fast loaders run through otr_linear_wgrad
 *      (accumulate) with the workspace given here.  Results are always ACCUMULATED. */
typedef struct {
  const void* dy;
  const void* x;
  float* dw;
  int32_t M, N, K;
  int64_t ldy, ldx, ldw;
  int32_t dy_dtype, x_dtype;
  float* dbias;          /* NULL, or [N]: dbias += column sums of dy (the bias gradient of the same Linear), folded into the
                          * 256-wide launch -- allowed only on items for which otr_wgrad256_takes() returns 1 */
  int32_t overwrite;     /* != 0: the caller guarantees that dw holds ZEROS which nothing else has written in this backward pass (a
                          * gradient buffer cleared at the start of the step, this item its only writer): the 256-wide launch then
                          * STORES the first partial sum of every tile instead of reading the zeros back (7 % of that launch's
                          * bytes).  The result is the same sum; other paths ignore the flag and accumulate. */
} otr_wgrad_item_t;
/* 1 when otr_linear_wgrad_grouped would run this item on the 256-wide kernel (csrc/wgrad256.hip), else 0 */
int32_t otr_wgrad256_takes(const otr_wgrad_item_t* item, int32_t compute);
/* the schedule the 256-wide launch would use for these items (host only; tests replay the kernel's work decoding on it):
 * out[0..7] = {mode (1 rounds / 0 stream-K), grid, chunk, full rounds, tiles left, row ranges per left tile, total slabs, n},
 * out[8 + i] = first slab of problem i; grid_cap as otr_debug_set(7, v) */
/* host-only views of two launch geometries, for tests (no device work): the (row block, slice) of every workgroup of the split FFN
 * kernels under mapping `map` (otr_debug_set(15, .)) -- out[2 b], out[2 b + 1], returns the grid size (out may be NULL) -- and the
 * (block, head, utterance) of every workgroup of the XCD-aware attention grid (-1 for padding workgroups), out[3 i ..] */
int32_t otr_debug_ffn_split_map(int64_t M, int32_t map, int32_t* out, int32_t cap);
int32_t otr_debug_attention_grid(int32_t nx, int32_t H, int32_t B, int32_t* out, int32_t cap);
int32_t otr_debug_wgrad256_plan(const otr_wgrad_item_t* items, int32_t n, int32_t grid_cap, int32_t* out);
/* number of pieces of the last 256-wide launch on `workspace` that gave up waiting for their turn (bounded spin; 0 in any
 * healthy run; < 0 on error).  Synchronises the device. */
int32_t otr_debug_wgrad256_errors(const void* workspace);
int32_t otr_linear_wgrad_grouped(const otr_wgrad_item_t* items, int32_t n, int32_t compute, void* workspace,
                                 int64_t workspace_bytes, void* stream);
typedef struct {
  const void* a;
  float* out;
  int64_t M, N, lda;
  int32_t dtype;
} otr_colsum_item_t;
int32_t otr_colsum_grouped(const otr_colsum_item_t* items, int32_t n, void* stream);

/* ---- scaled-dot-product attention, flash style: scores are never materialised
 *      (module/attention.py:23-46 compute_context, :76-82 self, :137-143 cross).
 * q/k/v/o are [B, T, H*dk] slices addressed by (batch stride, time stride) in elements; head h
 * occupies columns [h*dk, (h+1)*dk).  key_mask: uint8 [B, Tk] (1 = valid) or NULL; causal != 0
 * additionally masks key > query (decoder/utils.py:7-11).  lse: f32 [B,H,Tq] (saved for bwd). */
typedef struct {
  int32_t B, H, Tq, Tk, dk;
  int32_t dtype;                /* element type of q,k,v,o and their grads; also the MFMA type */
  int64_t q_bs, q_ts, k_bs, k_ts, v_bs, v_ts, o_bs, o_ts;
  int32_t causal;
  float scale;                  /* 1/sqrt(dk) */
} otr_attn_desc_t;
int32_t otr_attention_fwd(const otr_attn_desc_t* d, const void* q, const void* k, const void* v,
                          const uint8_t* key_mask, void* o, float* lse, void* stream);
/* do_/dq/dk/dv use the strides of o/q/k/v.  delta: f32 [B,H,Tq] workspace. */
int32_t otr_attention_bwd(const otr_attn_desc_t* d, const void* q, const void* k, const void* v,
                          const uint8_t* key_mask, const void* o, const void* do_, const float* lse,
                          float* delta, void* dq, void* dk, void* dv, void* stream);

/* Variant with an additive fp32 score bias, S = (q.k + bias[b,h,i,col]) * scale, addressed as
 * bias[b*bias_bs + h*bias_hs + i*bias_rs + col], col = j, or col = j - i + Tq - 1 when rel_shift != 0: the
 * Transformer-XL relative-position term of MultiHeadedSelfAttentionWithRelPos (module/attention.py:196-253; the
 * reference materialises [B,h,T,2T-1] and gathers it at :209-215).  dbias (same addressing, caller pre-zeroed when
 * rel_shift) receives d loss / d bias, in f32 or (r06, dbias_dtype) in the library's 16-bit type: the band tensor of one Conformer block is
 * 64 MB in f32, written once and read by two GEMMs. */
int32_t otr_attention_bias_fwd(const otr_attn_desc_t* d, const void* q, const void* k, const void* v,
                               const uint8_t* key_mask, const float* bias, int64_t bias_bs, int64_t bias_hs,
                               int64_t bias_rs, int32_t rel_shift, void* o, float* lse, void* stream);
int32_t otr_attention_bias_bwd(const otr_attn_desc_t* d, const void* q, const void* k, const void* v,
                               const uint8_t* key_mask, const float* bias, void* dbias, int32_t dbias_dtype, int64_t bias_bs,
                               int64_t bias_hs, int64_t bias_rs, int32_t rel_shift, const void* o, const void* do_,
                               const float* lse, float* delta, void* dq, void* dk, void* dv, void* stream);

/* ---- y = LayerNorm(x + dropout(a)) (encoder/transformer.py:54-56,61-63; decoder/transformer.py:
 *      66-68,76-78,84-86).  x f32 [M,d]; a [M,d] of a_dtype or NULL; z (= x+drop(a), f32) and
 *      mean/rstd (f32 [M]) are saved for backward.  Dropout masks come from a counter RNG keyed by
 *      (*seed, rng_offset + element index) and are regenerated, never stored. */
typedef struct {
  int64_t M;
  int32_t d;
  int32_t a_dtype;
  float eps, p_drop;            /* p_drop = 0 -> no dropout */
  uint64_t rng_offset;
  float a_scale;                /* r05: y = LN(x + a_scale * dropout(a)), da = a_scale * dropout'(dz); 0 means 1 (older callers) */
  const uint8_t* a_row_mask;    /* r05: [M] or NULL; rows with 0 take no branch: a row := 0 forward, da row := 0 backward (the
                                 * masked_fill of module/conformer.py:109 folded into the residual add that consumes the branch) */
  int32_t dy_dtype;             /* r06, backward entries: OTR_F32 (0) or the library's 16-bit type -- the LayerNorm's output fed a 16-bit
                                 * reader only (a Linear), whose input gradient then comes back in that type (half the bytes of the
                                 * input-gradient GEMM's output); forward entries may pass y = NULL then and write the 16-bit twin alone */
} otr_ln_desc_t;
/* y_bf16 (may be NULL): bf16 copy of y, the GEMM-operand form of the residual stream */
int32_t otr_add_layernorm_fwd(const otr_ln_desc_t* d, const float* x, const void* a, const float* gamma,
                              const float* beta, const uint64_t* seed, float* y, void* y_bf16, float* z, float* mean,
                              float* rstd, void* stream);
/* dx f32 [M,d] (residual grad), da [M,d] a_dtype (branch grad, may be NULL), dgamma/dbeta f32[d] +=.
 * da_colsum (f32[d] +=, may be NULL): column sums of da, i.e. the bias gradient of the Linear that produced the
 * branch (module/attention.py:43 output_proj, module/ffn.py:41 w_2) without a separate reduction launch. */
int32_t otr_add_layernorm_bwd(const otr_ln_desc_t* d, const float* dy, const float* z, const float* mean,
                              const float* rstd, const float* gamma, const uint64_t* seed, float* dx, void* da,
                              float* dgamma, float* dbeta, float* da_colsum, float* partial, void* stream);
/* partial (may be NULL): f32 [otr_add_layernorm_bwd_partial_rows(M), 3, d].  When given, NOTHING is added to dgamma / dbeta /
 * da_colsum (they may be NULL); each workgroup writes its own sums (dgamma | dbeta | da column sums) to its row and the
 * caller column-sums the rows (otr_colsum / otr_colsum_grouped): no atomics, deterministic. */
int64_t otr_add_layernorm_bwd_partial_rows(int64_t M);
/* the same with dx = skip + (gradient w.r.t. the LayerNorm input), skip f32 [M,d] or NULL: a pre-norm residual
 * x + f(LN(x)) (encoder/conformer.py:50-73; normalize_before layers) hands x two gradients, and this saves the elementwise
 * add autograd would launch to join them.  skip may alias dx.  With a branch (da != NULL) the skip gradient is part of the
 * gradient of the sum x + a_scale * dropout(a): da = a_scale * dropout'(skip + LayerNorm input gradient) (r05, ops.ResidualLnFn). */
int32_t otr_add_layernorm_bwd_skip(const otr_ln_desc_t* d, const float* dy, const float* z, const float* mean,
                                   const float* rstd, const float* gamma, const uint64_t* seed, const float* skip, float* dx,
                                   void* da, float* dgamma, float* dbeta, float* da_colsum, float* partial, void* stream);

/* Two LayerNorms back to back, y2 = LN2(LN1(x + a_scale * dropout(a))) -- encoder/conformer.py:87-89 (post_ffn_norm, then final_norm, on
 * the result of the convolution branch's residual add) -- in one launch each way (r05).  Forward writes y2 (+ 16-bit twin), the pre-norm
 * sum z and both LayerNorms' row statistics; backward takes d y2, recomputes y1 = LN1(z) from z / mean / rstd, and leaves
 * partial f32 [otr_add_layernorm_bwd_partial_rows(M)][5 d] = per-workgroup sums of dgamma | dbeta | da | dgamma2 | dbeta2 (the
 * caller column-sums them).  skip / dx / da as otr_add_layernorm_bwd_skip. */
int32_t otr_add_layernorm2_fwd(const otr_ln_desc_t* d, const float* x, const void* a, const float* gamma, const float* beta,
                               const float* gamma2, const float* beta2, const uint64_t* seed, float* y2, void* y2_bf16, float* z,
                               float* mean, float* rstd, float* mean2, float* rstd2, void* stream);
int32_t otr_add_layernorm2_bwd(const otr_ln_desc_t* d, const float* dy2, const float* z, const float* mean, const float* rstd,
                               const float* gamma, const float* beta, const float* mean2, const float* rstd2, const float* gamma2,
                               const uint64_t* seed, const float* skip, float* dx, void* da, float* partial, void* stream);
/* Three LayerNorms back to back (r06): y2 = LN2(LN1(x + a_scale * dropout(a))) as above AND y3 = LN3(y2) -- final_norm of a Conformer block
 * is followed by the NEXT block's macaron_ffn_norm (encoder/conformer.py:89, :50), whose output only feeds a Linear: y3 (NULL: not written)
 * and its 16-bit twin.  Backward takes d y2 AND d y3 (autograd hands the node both; d y3 in f32 or in the library's 16-bit type, dy3_dtype:
 * a 16-bit consumer's input gradient comes back in that type), recomputes y1 and y2 from z and the saved statistics,
 * and leaves partial f32 [rows][7 d] = dgamma | dbeta | da | dgamma2 | dbeta2 | dgamma3 | dbeta3. */
int32_t otr_add_layernorm3_fwd(const otr_ln_desc_t* d, const float* x, const void* a, const float* gamma, const float* beta,
                               const float* gamma2, const float* beta2, const float* gamma3, const float* beta3, const uint64_t* seed,
                               float* y2, float* y3, void* y3_bf16, float* z, float* mean, float* rstd, float* mean2, float* rstd2,
                               float* mean3, float* rstd3, void* stream);
int32_t otr_add_layernorm3_bwd(const otr_ln_desc_t* d, const float* dy2, const void* dy3, int32_t dy3_dtype, const float* z, const float* mean,
                               const float* rstd, const float* gamma, const float* beta, const float* mean2, const float* rstd2,
                               const float* gamma2, const float* beta2, const float* mean3, const float* rstd3, const float* gamma3,
                               const uint64_t* seed, const float* skip, float* dx, void* da, float* partial, void* stream);

/* ---- F.glu / F.relu on the FFN hidden (module/ffn.py:15-21,40): u[M,F] = h[:, :F]*sigmoid(h[:, F:]) */
/* row_mask (uint8 [M], may be NULL): rows with mask 0 produce u = 0 / dh = 0 (module/conformer.py:46) */
int32_t otr_glu_fwd(const void* h, void* u, int32_t dtype, int64_t M, int64_t F, const uint8_t* row_mask, void* stream);
/* dh[M,2F] from du[M,F]; if dbias_partial != NULL it receives per-row-block partial column sums of dh,
 * f32 [ceil(M/32), 2F] (deterministic; column-sum them with otr_colsum to get the w_1 bias gradient) */
int32_t otr_glu_bwd(const void* h, const void* du, void* dh, float* dbias, int32_t dtype, int64_t M, int64_t F,
                    const uint8_t* row_mask, int32_t h_has_sigmoid, void* stream);

/* ---- PositionalEncoding (module/pos.py:30-57): y = x*scale + PE[t], t = row % T.
 *      x may alias y. */
int32_t otr_posenc_fwd(const float* x, float* y, void* y_bf16, int64_t rows, int32_t T, int32_t d, float scale,
                       void* stream);
/* the same launch also casts the encoder's key mask: mask_out[r] (uint8 [rows]) = mask_in[(r / T) * mask_bs + (r % T) * mask_ts] != 0,
 * mask_in = the bytes of a bool / uint8 mask read through its strides (the frame mask behind the two stride-2 convolutions is a
 * strided view, frontend/conv.py:78-83); both NULL = otr_posenc_fwd.  d % 4 == 0, x / y 16-byte aligned. */
int32_t otr_posenc_mask_fwd(const float* x, float* y, void* y_bf16, int64_t rows, int32_t T, int32_t d, float scale,
                            const uint8_t* mask_in, int64_t mask_bs, int64_t mask_ts, uint8_t* mask_out, void* stream);
/* decoder embedding + posenc (decoder/transformer.py:163-169): y[r,:] = E[tok[r],:]*scale + PE[r % L] */
int32_t otr_embed_posenc_fwd(const int64_t* tok, const float* E, float* y, void* y_bf16, int64_t rows, int32_t L,
                             int32_t d, int32_t vocab, float scale, void* stream);
/* dE[tok[r],:] += scale * dy[r,:]  (atomic f32) */
int32_t otr_embed_bwd(const int64_t* tok, const float* dy, float* dE, int64_t rows, int32_t d, int32_t vocab,
                      float scale, void* stream);
/* The two above on a token VIEW: tok is addressed as tok[(r / L) * ld_tok + r % L] (`truth[:, :-1]` of the [B, L + 1] target matrix,
 * model/speech2text.py:53: no copy).  otr_embed_bwd_ld also takes the gradient as partial sums: dE[tok[r],:] += scale * (dy[r,:] +
 * sum_s slabs[s][r,:]) with dy f32 [rows, d] or NULL and slabs 16-bit [nslab][rows][d] or NULL (what otr_dec_self_bwd leaves). */
int32_t otr_embed_posenc_fwd_ld(const int64_t* tok, int64_t ld_tok, const float* E, float* y, void* y_bf16, int64_t rows, int32_t L,
                                int32_t d, int32_t vocab, float scale, void* stream);
int32_t otr_embed_bwd_ld(const int64_t* tok, int64_t ld_tok, int32_t L, const float* dy, const void* slabs, int32_t nslab, float* dE,
                         int64_t rows, int32_t d, int32_t vocab, float scale, void* stream);
/* dst (bf16) = src (f32), n elements: bf16 shadows of weights / activations */
int32_t otr_cast_f32_to_bf16(const float* src, void* dst, int64_t n, void* stream);
/* y = x * (*s_dev) * s_host, elementwise f32 (n elements); x may alias y; s_dev may be NULL */
int32_t otr_scale(const float* x, float* y, int64_t n, const float* s_dev, float s_host, void* stream);
/* y16 (the library's 16-bit type) = x * s, n elements; x 16-byte, y16 8-byte aligned.  The positional encoding's backward
 * (module/pos.py:44-57) when its gradient goes on as a GEMM operand. */
int32_t otr_scale_cast(const float* x, void* y16, int64_t n, float s, void* stream);

/* ---- Conv2d-subsampling frontend (frontend/conv.py:50-83 Conv2dLayer, :131-153 ConvFrontEnd).
 *      Activations are channel-last: act1 [B,T1,F1,C1], act2 [B,T2,F2,C2] == [B*T2, F2*C2] rows
 *      (the flatten of conv.py:145 becomes a column permutation of output_layer.weight). */
typedef struct {
  int32_t B, T, F;              /* input fbank [B,T,F] f32 */
  int32_t C1, C2;               /* mid / out channels */
  int32_t T1, F1, T2, F2;       /* derived: T1=(T-3)/2+1, F1=(F-1)/2+1, ... */
  int32_t act_dtype, compute;
  int32_t w_dtype;              /* dtype of w2r as passed (f32 master or its bf16 shadow) */
} otr_conv_desc_t;
/* conv1: 1->C1, 3x3, stride 2, pad (0,1), +bias, ReLU.  w1 f32 [C1,1,3,3] */
int32_t otr_conv1_fwd(const otr_conv_desc_t* d, const float* x, const float* w1, const float* b1, void* act1,
                      void* stream);
/* dw1 [C1,9] f32 +=, db1 [C1] f32 += ; dact1 already masked by ReLU.  partial (may be NULL): f32
 * [otr_conv1_wgrad_partial_rows()][10*C1]; when given NOTHING is added to dw1 / db1 (they may be NULL): every workgroup
 * writes its own sums (9*C1 weight-gradient values in dw1's layout, then C1 bias-gradient values) and the caller
 * column-sums the rows (otr_colsum / otr_colsum_grouped): no atomics, deterministic, 4x the workgroups. */
int32_t otr_conv1_wgrad(const otr_conv_desc_t* d, const float* x, const void* dact1, float* dw1, float* db1,
                        float* partial, void* stream);
int32_t otr_conv1_wgrad_partial_rows(void);
/* conv2 as implicit GEMM on MFMA: w2r = w2 permuted to [C2,3,3,C1] (f32). +bias, ReLU */
int32_t otr_conv2_fwd(const otr_conv_desc_t* d, const void* act1, const void* w2r, const float* b2, void* act2,
                      void* stream);
/* Both Conv2dLayers of frontend/conv.py:141-142 in ONE launch (csrc/conv2fwd.hip): conv2 on a weight-stationary kernel whose input
 * rows are computed from the filterbank frames in place (conv1's arithmetic, bit for bit); act1 is written for the backward pass.
 * 16-bit activations and w2r, (C1, C2) = (64, 128), F1 = 40 or 20.  Returns 1 without launching anything when the operands do not
 * qualify: call otr_conv1_fwd + otr_conv2_fwd then.  otr_debug_set(22, 0 | 1 | 2): 0 = the generic paths, 2 = conv2 alone on the new kernel. */
int32_t otr_conv12_fwd(const otr_conv_desc_t* d, const float* x, const float* w1, const float* b1, void* act1, const void* w2r,
                       const float* b2, void* act2, void* stream);
/* dcol[B*T2*F2, 9*C1] (act dtype) = dact2 * w2r  (dact2 already masked by ReLU) */
int32_t otr_conv2_dgrad_cols(const otr_conv_desc_t* d, const void* dact2, const void* w2r, void* dcol,
                             void* stream);
/* dact1 = col2im(dcol) * (act1 > 0) */
int32_t otr_conv2_col2im(const otr_conv_desc_t* d, const void* dcol, const void* act1, void* dact1, void* stream);
/* The two above in ONE launch, implicitly (no column matrix): dact1 = [act1 > 0] * conv2's input gradient of dact2.  The
 * stride-2 output pixels fall into four parity classes with 4 / 2 / 2 / 1 taps; each is a GEMM whose B operand is read
 * straight from dact2's channel rows (csrc/conv.hip).  16-bit operands, (C1, C2) = (64, 128) or (32, 64).
 * Returns 1 without launching anything when the operands do not qualify: use otr_conv2_dgrad_cols + otr_conv2_col2im then. */
int32_t otr_conv2_dgrad(const otr_conv_desc_t* d, const void* dact2, const void* w2r, const void* act1, void* dact1,
                        void* stream);
/* r06 (ABI 601) -- the input gradient of a WIDE frontend, C1 == C2 == 256 (conformer_baseline.yaml; frontend/conv.py:50-83): the parity
 * classes as an implicit GEMM whose weights stream through LDS in MFMA-fragment order, all 256 channels of a pixel per workgroup
 * (csrc/conv2wide.hip); the ReLU mask of act1 is fused.  `scratch` (>= otr_conv2_wide_scratch_bytes(), 16-byte aligned) receives the
 * re-ordered weights on the launch's stream.  16-bit activations and weights.
 * Returns 1 without launching anything when the operands do not qualify: use otr_conv2_dgrad (or its column-matrix form). */
int64_t otr_conv2_wide_scratch_bytes(void);
int32_t otr_conv2_dgrad_wide(const otr_conv_desc_t* d, const void* dact2, const void* w2r, const void* act1, void* dact1, void* scratch,
                             int64_t scratch_bytes, void* stream);
/* host-side plan of that launch, no device work (tests): out[10] = {served 0/1, first workgroup of the parity classes
 * c = 2*(t1&1) + (f1&1) and the total (5 values), 256-pixel tiles per class (4 values)} */
int32_t otr_debug_conv2_dgrad_plan(const otr_conv_desc_t* d, int32_t* out);
/* dw2r [C2, 9*C1] f32 = dact2^T * im2col(act1) */
int32_t otr_conv2_wgrad(const otr_conv_desc_t* d, const void* dact2, const void* act1, float* dw2r, void* workspace,
                        int64_t workspace_bytes, void* stream);
/* g = g * (y > 0) elementwise (ReLU backward through a stored post-ReLU activation); g may alias out */
int32_t otr_relu_bwd(const void* y, const void* g, void* out, int32_t dtype, int64_t n, void* stream);
/* The same over a [rows, cols] matrix plus the bias gradient of the layer that produced y (column sums of out) in one pass:
 * partial [otr_relu_bwd_colsum_partial_rows(rows, cols, dtype)][cols] f32 receives per-workgroup sums (no atomics) for the
 * caller's column sum.  The second Conv2dLayer of frontend/conv.py:63-66 (relu then bias, C2 channels) is the user.
 * partial_rows returns 0 when the shape is not served (cols must be a multiple of the 16-byte vector, cols / vector dividing 256). */
int32_t otr_relu_bwd_colsum_partial_rows(int64_t rows, int32_t cols, int32_t dtype);
int32_t otr_relu_bwd_colsum(const void* y, const void* g, void* out, float* partial, int32_t dtype, int64_t rows, int32_t cols,
                            void* stream);

/* The other FFN activations of module/ffn.py:15-21 (relu runs in the GEMM epilogue, glu has otr_glu_* / otr_ffn_glu_*):
 * kind 1 = gelu (erf form, F.gelu's default), 2 = tanh, 3 = swish (x * sigmoid(x)).  y = act(x);  dx = dy * act'(x) from
 * the saved pre-activation x.  f32 or bf16, n elements, 16-byte aligned buffers; dx may alias dy. */
int32_t otr_act_fwd(const void* x, void* y, int32_t dtype, int64_t n, int32_t kind, void* stream);
int32_t otr_act_bwd(const void* x, const void* dy, void* dx, int32_t dtype, int64_t n, int32_t kind, void* stream);

/* ---- LabelSmoothingLoss (module/loss.py:21-48): logits f32 [R,V], target int64 [R].
 *      loss (f32 scalar) = sum_nonpad KL(conf || softmax) / #nonpad ; dlogits = d loss / d logits.
 *      scratch: f32[R+2] workspace (per-row losses are reduced in a fixed order: deterministic). */
int32_t otr_label_smoothing_loss(const float* logits, const int64_t* target, int64_t R, int32_t V, float smoothing,
                                 int32_t pad_idx, float* loss, float* dlogits, float* scratch, void* stream);
/* The same on rows longer than V (leading dimensions ld_logits, ld_dlogits >= V, in elements): the logits of an output layer
 * whose width is not a multiple of 8 (decoder/transformer.py:153, 4234 tokens) arrive as the head of a [R, V8] product and the
 * gradient leaves the same way, its columns V .. ld_dlogits-1 written as zeros, so that the layer's backward GEMMs read aligned
 * rows at the padded width. */
int32_t otr_label_smoothing_loss_ld(const float* logits, int64_t ld_logits, const int64_t* target, int64_t R, int32_t V,
                                    float smoothing, int32_t pad_idx, float* loss, float* dlogits, int64_t ld_dlogits,
                                    float* scratch, void* stream);

/* The whole loss in ONE launch, for the training step (module/loss.py:21-48 + the two scalar multiplications around it): every row
 * is read once (16-byte loads: logits and dlogits rows 16-byte aligned, ld % 4 == 0, V <= 8192, R <= 8192), the block that finishes
 * last adds up the row losses in the fixed order of otr_label_smoothing_loss (deterministic), and dlogits leaves already multiplied
 * by the device scalar *grad_scale (NULL = 1: the factor the caller's backward pass will be seeded with, e.g. the dynamic loss
 * scale of otr_optimizer_step's state block).  target: int64, addressed as target[(r / L) * ld_target + r % L] for r in [0, R): a
 * [B, L] view with row stride ld_target (model/speech2text.py:57 `truth[:, 1:]`) needs no copy; R % L == 0.
 * dlogits_dtype OTR_F32, or OTR_H16: the gradient in the library's 16-bit type (ld_dlogits % 8 == 0) -- the operand type of the output
 * layer's three backward GEMMs.  scratch: f32[R+2]; ticket: one uint32 the caller zero-initialises ONCE (the launch leaves it at
 * zero again). */
int32_t otr_label_smoothing_loss_fused(const float* logits, int64_t ld_logits, const int64_t* target, int64_t ld_target, int32_t L,
                                       int64_t R, int32_t V, float smoothing, int32_t pad_idx, const float* grad_scale, float* loss,
                                       void* dlogits, int32_t dlogits_dtype, int64_t ld_dlogits, float* scratch, uint32_t* ticket,
                                       void* stream);

/* ---- log_softmax over the last dim, f32 [R,V] (model/ctc.py:51,66; decoder/transformer.py:206) */
int32_t otr_log_softmax(const float* x, float* y, int64_t R, int32_t V, void* stream);

/* ---- Conformer encoder pieces (BASELINE configs[3]; encoder/conformer.py:20-114, module/conformer.py:12-57,
 *      module/attention.py:176-253).  All [M = B*T, C] row-major, C % 4 == 0. */
/* y = x + scale * dropout(a)  (pre-norm residual branches, ffn_scale 0.5 for the macaron FFN) and its branch grad */
int32_t otr_residual_add_fwd(const float* x, const void* a, int32_t a_dtype, float* y, int64_t n, float scale,
                             float p_drop, const uint64_t* seed, uint64_t rng_offset, void* stream);
int32_t otr_residual_add_bwd(const float* dy, void* da, int32_t a_dtype, int64_t n, float scale, float p_drop,
                             const uint64_t* seed, uint64_t rng_offset, void* stream);
/* y = dropout(x): x * mask / (1 - p), mask from the counter RNG keyed by (*seed, rng_offset + element index); applied to dy
 * with the same (seed, offset) it is the backward pass.  nn.Dropout of module/attention.py:46 (projected context),
 * module/ffn.py:40 (hidden), frontend/conv.py:66, module/conformer.py (end of the convolution module).  n % 4 == 0. */
int32_t otr_dropout(const void* x, void* y, int32_t dtype, int64_t n, float p_drop, const uint64_t* seed,
                    uint64_t rng_offset, void* stream);
/* out[M, 2d] = [q + u | q + v] (pos_bias_u / pos_bias_v flattened to [d]); q has leading dimension ldq */
int32_t otr_head_bias_add(const void* q, int64_t ldq, const float* u, const float* v, void* out, int32_t dtype,
                          int64_t M, int32_t d, void* stream);
/* dst[r, c, f] += src[r, f, c] (fp32; dst [rows, C, F], src [rows, F, C], both dense); clear_src != 0 leaves src ZERO.  Regroups a
 * weight gradient produced in a kernel's column order into the parameter's layout in one launch: the frontend Linear's staging image
 * (frontend/conv.py:145-146: flatten c*F+f, kernel order f*C+c) and conv2's channel-last taps (frontend/conv.py:56).  Replaces
 * torch's strided add (AccumulateGrad / add_) in the step. */
int32_t otr_regroup_add(float* dst, float* src, int64_t rows, int32_t C, int32_t F, int32_t clear_src, void* stream);
/* out[r, :cols] = a[r, :cols] + b[r, :cols] with independent leading dimensions */
int32_t otr_add2_strided(const void* a, int64_t lda, const void* b, int64_t ldb, void* out, int64_t ldo, int32_t dtype,
                         int64_t M, int32_t cols, void* stream);
/* r06: the same sum with the column sums of a and of b left as per-workgroup partials, partial [otr_add2_colsum_partial_rows(M)][2 cols] f32
 * (a's sums, then b's): d(q+u) + d(q+v) -> the packed qkv gradient, and the gradients of pos_bias_u / pos_bias_v (module/attention.py:241-245)
 * without a second pass over the two operands. */
int64_t otr_add2_colsum_partial_rows(int64_t M);
int32_t otr_add2_strided_colsum(const void* a, int64_t lda, const void* b, int64_t ldb, void* out, int64_t ldo, int32_t dtype, int64_t M,
                                int32_t cols, float* partial, void* stream);
/* out = x where mask[row] else 0 (f32) */
int32_t otr_row_mask(const float* x, const uint8_t* mask, float* out, int64_t M, int32_t C, void* stream);
/* the same with a type change on the way (OTR_F32 / OTR_H16 either side): the Conformer convolution module hands its
 * branch to the residual add, and takes the gradient back, in the activation type (module/conformer.py:56) */
int32_t otr_row_mask_cast(const void* x, int32_t x_dtype, const uint8_t* mask, void* out, int32_t out_dtype, int64_t M,
                          int32_t C, void* stream);
/* depthwise Conv1d over time, channel-last, per utterance, zero padded:
 *   y[b,t,c] = bias[c] + sum_{j<k} w[c,j] * g[b, t + j - pad, c]          k <= 7, 0 <= pad < k
 * pad = (k-1)/2 is the Conformer's 'same' convolution (module/conformer.py:26-28); pad = 0 with k = lookahead_steps + 1 is
 * the CTC head's look-ahead convolution (model/ctc.py:17-24,35-39: right-padded, no bias).  y f32 [B,T,C]; stats f32 [2C]
 * (may be NULL) receives per-channel sum / sum of squares of y over all B*T rows (BatchNorm batch statistics) */
int32_t otr_dwconv_fwd(const void* g, int32_t dtype, const float* w, const float* bias, float* y, float* stats,
                       int32_t B, int32_t T, int32_t C, int32_t k, int32_t pad, void* stream);
/* dg (dtype), dw f32 [C,k] +=, db f32 [C] += (may be NULL) */
int32_t otr_dwconv_bwd(const float* dy, const void* g, int32_t dtype, const float* w, void* dg, float* dw, float* db,
                       int32_t B, int32_t T, int32_t C, int32_t k, int32_t pad, void* stream);
/* otr_dwconv_bwd_part: the same with the weight / bias gradient sums left PER WORKGROUP in part [otr_dwconv_bwd_partial_rows(B*T)][C*k + C]
 * (columns: dw as [C][k], then db) instead of atomic adds on dw / db: the caller column-sums them (otr_colsum_grouped at the end of the
 * backward pass): 249 workgroups x 2304 atomics on 2304 addresses were half of the launch at the bench batch. */
int64_t otr_dwconv_bwd_partial_rows(int64_t M);
int32_t otr_dwconv_bwd_part(const float* dy, const void* g, int32_t dtype, const float* w, void* dg, float* part, int32_t B, int32_t T,
                            int32_t C, int32_t k, int32_t pad, void* stream);
/* BatchNorm1d (training: batch statistics from `stats`, running stats updated in place; eval: running stats) fused
 * with swish; saved f32 [2C] = mean | rstd for backward */
int32_t otr_bn_swish_fwd(const float* y, const float* stats, const float* gamma, const float* beta, float* running_mean,
                         float* running_var, float* saved, void* out, int32_t out_dtype, int64_t M, int32_t C, float eps,
                         float momentum, int32_t training, void* stream);
/* r05: the BatchNorm batch statistics without a zeroing launch and without atomics: otr_dwconv_fwd_part leaves the depthwise
 * convolution's per-workgroup sums in spart [otr_dwconv_fwd_partial_rows(B*T)][sum y (C) | sum y^2 (C)], otr_bn_swish_fwd_part adds
 * them up in its statistics launch (training mode; module/conformer.py:40-46,103-110). */
int64_t otr_dwconv_fwd_partial_rows(int64_t M);
int32_t otr_dwconv_fwd_part(const void* g, int32_t dtype, const float* w, const float* bias, float* y, float* spart, int32_t B, int32_t T,
                            int32_t C, int32_t k, int32_t pad, void* stream);
int32_t otr_bn_swish_fwd_part(const float* y, const float* spart, int32_t nblk, const float* gamma, const float* beta, float* running_mean,
                              float* running_var, float* saved, void* out, int32_t out_dtype, int64_t M, int32_t C, float eps,
                              float momentum, void* stream);
/* red f32 [2C]: on return d beta | d gamma of THIS call; dy f32 [M,C] = gradient w.r.t. the BatchNorm input.
 * partial: f32 scratch [otr_bn_swish_bwd_partial_rows(M)][2C] (per-strip sums: no atomics, deterministic).
 * dgamma_acc / dbeta_acc (f32 [C], may be NULL): the parameter gradients, += the same sums in the reduction launch. */
int32_t otr_bn_swish_bwd_partial_rows(int64_t M);
int32_t otr_bn_swish_bwd(const float* y, const void* ds, int32_t ds_dtype, const float* saved, const float* gamma,
                         const float* beta, float* red, float* partial, float* dgamma_acc, float* dbeta_acc, float* dy,
                         int64_t M, int32_t C, int32_t training, void* stream);
/* r06 (ABI 601): the middle of ConformerConvolutionModule's backward (module/conformer.py:36-57) as two steps instead of five launches.
 * otr_bn_swish_bwd_sums = the first two launches of otr_bn_swish_bwd (per-strip sums, their reduction into red [2C] and, when given, into
 * the BatchNorm parameter gradients); otr_conformer_conv_bwd_mid then does BatchNorm's apply step, the depthwise convolution's backward and
 * the GLU's backward in ONE launch: dh [M, 2C] from y, ds, red, g (the GLU output) and h (the GLU input).  The [M, C] fp32 dy and the
 * [M, C] dg tensors of the stand-alone chain are never written.  part [otr_dwconv_bwd_partial_rows(M)][C*k + C] and gpart [same rows][2C]
 * receive per-workgroup sums (no atomics): the depthwise conv's dw | db and the columns of dh (pointwise_conv1's bias gradient). */
int32_t otr_bn_swish_bwd_sums(const float* y, const void* ds, int32_t ds_dtype, const float* saved, const float* gamma, const float* beta,
                              float* red, float* partial, float* dgamma_acc, float* dbeta_acc, int64_t M, int32_t C, void* stream);
int32_t otr_conformer_conv_bwd_mid(const float* y, const void* ds, const float* saved, const float* gamma, const float* beta,
                                   const float* red, const void* g, const float* w, const void* h, const uint8_t* row_mask, void* dh,
                                   float* part, float* gpart, int32_t dtype, int32_t training, int32_t B, int32_t T, int32_t C, int32_t k,
                                   int32_t pad, void* stream);

/* ---- CTC loss with gradient w.r.t. the logits (nn.CTCLoss(blank, zero_infinity=True), reduction
 *      'mean', as built at model/ctc.py:30 and called at :50-53).  log_probs f32 [B,T,V] (already
 *      log-softmaxed), targets int64 [B, ldt], in_len/tgt_len int32 [B], max_tgt >= max(tgt_len) (<=127).
 *      alpha_ws: f32 [B, T, 2*max_tgt+1] workspace; nll: f32 [B] (0 where infeasible);
 *      loss: f32 scalar; dlogits f32 [B,T,V] or NULL. */
int32_t otr_ctc_loss(const float* log_probs, const int64_t* targets, int64_t ldt, const int32_t* in_len,
                     const int32_t* tgt_len, int32_t B, int32_t T, int32_t V, int32_t max_tgt, int32_t blank,
                     float* alpha_ws, float* nll, float* loss, float* dlogits, void* stream);

/* ---- one optimizer update over a replica's FLAT buffers (train/trainer.py:221-234 clip_grad_norm_(5) + gradient noise +
 *      NaN guard + scheduler.step + optimizer.step; train/scheduler.py:129-138 Noam lr; torch Adam with
 *      L2 weight decay, train/scheduler.py:10-13).  grad_scale (1/world_size after the all-reduce-sum)
 *      is folded into clip + update.  state: f32[OTR_OPT_STATE_FLOATS = 528] device block, zero-initialised by the caller once;
 *      [16..527] is scratch (per-workgroup partial sums of the gradient norm, added up in a fixed order: the clip factor is
 *      bit-reproducible, so data-parallel replicas that hold identical all-reduced gradients stay identical):
 *        [0] step  [1] lr  [2] bc1  [3] bc2  [4] sqnorm  [5] skipped  [6] loss_scale  [7] good_steps  [8] unscale
 *        [9] growth_interval  [10] faults (give-ups reported through otr_set_fault_counter; each skipped its update)
 *        [11..15] reserved.
 *      Loss scaling (fp16 builds): when state[6] > 0 the gradients in `grad` are state[6] times too large (the caller
 *      seeded its backward pass with that device scalar); the update divides it out, HALVES it and skips the update when
 *      the gradient norm is not finite, and doubles it after state[9] consecutive finite updates (0 = never) -- all on
 *      the device, so the step stays hipGraph-replayable.  grad_noise_std > 0 adds N(0, std) to every gradient element
 *      after clipping (trainer.py:223-227; pass train.grad_noise / accum_steps) -- except where gradient AND parameter are
 *      exactly zero (padding cells of the flat buffers stay zero).  noam_warmup <= 0 selects base_lr.  All arguments are
 *      checked before the first launch: a refused call leaves the device state untouched. */
#define OTR_OPT_STATE_FLOATS 528
int32_t otr_optimizer_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                           float* state, int32_t state_floats /* floats the caller allocated for `state`: refused below OTR_OPT_STATE_FLOATS */,
                           void* param_bf16 /* NULL or 16-bit shadow[n] refreshed in the same pass */,
                           float base_lr, float beta1, float beta2, float eps, float weight_decay,
                           float grad_scale, float clip_norm, float noam_model_size, float noam_warmup,
                           float noam_factor, float noam_step_offset, float grad_noise_std, void* stream);

/* ---- gradient all-reduce over RCCL / xGMI (replaces nn.DataParallel's per-step scatter / replicate / gather /
 *      reduce-add, train/trainer.py:56-66).  One in-place sum over the replica's flat gradient buffer,
 *      issued on the stream passed in (the compute stream: nothing to synchronise before the optimizer).  librccl is
 *      opened lazily (dlopen); the communicator is the one object the library owns.
 *      Rendezvous: rank 0 calls otr_allreduce_unique_id, the HOST ships the 128 bytes to every rank, every rank (having
 *      selected its GPU with hipSetDevice) calls otr_allreduce_init; dtype is OTR_F32 / OTR_BF16 / OTR_F16 of `buf`. */
int32_t otr_allreduce_unique_id(void* id128);
int32_t otr_allreduce_init(void** handle, const void* id128, int32_t rank, int32_t world);
int32_t otr_allreduce_run(void* handle, void* buf, int64_t count, int32_t dtype, void* stream);
int32_t otr_allreduce_destroy(void* handle);

/* ---- batch beam search step (recognize/speech2text.py:95-192).
 * beam_topk: rows = batch*beam hypotheses; logits f32 [rows, V] (row stride ld) are the decoder logits of
 *   the last position; optional LM logits are fused as log_softmax(dec) + lm_weight*log_softmax(lm)
 *   (speech2text.py:102-105); writes the k best (score, token) per row, descending (:112).
 * beam_prune: finished-beam masking (:156-192), score update, top-k over beam^2 candidates, prefix
 *   gather + token append (:118-146).  preds_* int64 [batch*beam, ldp] hold t tokens on entry, t+1 on
 *   exit; n_finished (int32 device scalar) = number of hypotheses ending in EOS after the step. */
int32_t otr_beam_topk(const float* logits, int64_t ld, const float* lm_logits, int64_t ld_lm, float lm_weight,
                      int64_t rows, int32_t V, int32_t k, float* out_score, int64_t* out_idx, void* stream);
int32_t otr_beam_prune(const float* k_score, const int64_t* k_idx, const float* scores_in, const uint8_t* flag_in,
                       const int64_t* preds_in, int64_t ldp, int32_t batch, int32_t beam, int32_t t, int32_t eos,
                       float* scores_out, uint8_t* flag_out, int64_t* preds_out, int32_t* n_finished, void* stream);

/* ---- transposed copies of many matrices in ONE launch (the W^T bf16 shadows that turn dx = dy.W into a
 *      forward-type GEMM; refreshed after every optimizer step).  table: DEVICE int64 [n_mats,4] rows of
 *      {element offset, rows, cols, first tile}; matrix i is src+offset [rows,cols] row-major and is written to
 *      dst+offset as [cols,rows]; tiles are 64x64, total_tiles = sum of ceil(rows/64)*ceil(cols/64). */
int32_t otr_transpose_batched(const void* src, void* dst, const int64_t* table, int32_t n_mats, int64_t total_tiles,
                              int32_t elem_bytes, void* stream);

/* ---- SpecAugment on the device (data/augment.py:9-41): zero x[b,t,f] (f32 [B,T,F]) inside NR
 *      rectangles per utterance, ranges int32 [B,NR,4] = {t0,t1,f0,f1} half-open.  The rectangles are drawn on the host
 *      with the reference's own random calls (opentransformer_amd/data.py), so the masks are bit-identical. */
int32_t otr_spec_mask(float* x, const int32_t* ranges, int32_t B, int32_t NR, int32_t T, int32_t F, void* stream);

/* ---- incremental (KV-cached) decoding, SURVEY.md 8f rank 1.  The reference threads a `cache` argument through
 *      decoder.inference / attention.inference but never fills it (decoder/transformer.py:185-208,
 *      module/attention.py:86-104, README.md:13 TODO) and re-runs the decoder over the whole prefix each step.
 *      Here one token per hypothesis is fed per step; every step-dependent scalar is read from DEVICE memory
 *      (pos = index of the token being fed) so one captured hipGraph serves all steps.
 * decode_embed: y[r,:] = E[preds[r, *pos], :]*scale + PE[*pos]  (decoder/transformer.py:163-169, model/lm.py:143-150)
 * decode_self_attention: qkv [rows, 3*H*dk] (q|k|v of the new position); kcache/vcache [rows, maxlen, H*dk] are
 *   write-once: the new k,v are stored at [r, *pos]; anc int32 [rows, maxlen]: anc[r,j] = cache row that holds
 *   position j < *pos of hypothesis r (hypotheses form a tree under beam pruning, so caches are never gathered);
 *   out [rows, H*dk] = softmax(q.k/sqrt(dk)) v over positions 0..*pos.
 * beam_prune_cached: otr_beam_prune with t = *pos_in + 1, plus anc_out[r'] = anc_in[parent(r')] ++ parent(r') and
 *   *pos_out = *pos_in + 1 (in/out buffers must be distinct). */
int32_t otr_decode_embed(const int64_t* preds, int64_t ldp, const int32_t* pos, const float* E, float* y, void* y_bf16,
                         int64_t rows, int32_t d, int32_t vocab, float scale, void* stream);
/* ---- the recurrent language model of the shallow fusion (model/lm.py:33-91: nn.Embedding -> nn.LSTM -> Linear; reached through
 *      recognize/base.py:26-37, which feeds it the LAST token of every hypothesis).
 * otr_decode_lookup: y[r,:] = E[preds[r, *pos],:] (f32 + optional 16-bit twin), pos a device scalar or NULL (= column 0).
 * otr_lstm_cell:     one cell update in torch.nn.LSTM's gate order i | f | g | o: gates = gates_a [rows,4H] + gates_b [rows,4H] (NULL)
 *                    + bias_b [4H] (NULL);  c = sigmoid(f) c_prev + sigmoid(i) tanh(g) (c_prev NULL = zeros);  h = sigmoid(o) tanh(c).
 *                    The two GEMMs (W_ih x + b_ih, W_hh h + b_hh) are otr_linear_fwd calls. */
int32_t otr_decode_lookup(const int64_t* preds, int64_t ldp, const int32_t* pos, const float* E, float* y, void* y_bf16, int64_t rows,
                          int32_t d, int32_t vocab, void* stream);
int32_t otr_lstm_cell(const float* gates_a, const float* gates_b, const float* bias_b, const float* c_prev, float* h, void* h_bf16,
                      float* c, int64_t rows, int32_t hidden, void* stream);
int32_t otr_decode_self_attention(const void* qkv, void* kcache, void* vcache, const int32_t* anc, const int32_t* pos,
                                  void* out, int32_t dtype, int64_t rows, int32_t H, int32_t dk, int32_t maxlen,
                                  float scale, void* stream);
/* otr_beam_prune_cached: n_finished is int32[2] here -- [0] receives the number of finished hypotheses, [1] is the kernel's arrival
 * word: zero it once when the buffer is made, the launch leaves it zero (no zeroing launch per step, r05).  batch * beam < 65536. */
int32_t otr_beam_prune_cached(const float* k_score, const int64_t* k_idx, const float* scores_in, const uint8_t* flag_in,
                              const int64_t* preds_in, int64_t ldp, int32_t batch, int32_t beam, int32_t eos,
                              const int32_t* pos_in, int32_t* pos_out, const int32_t* anc_in, int32_t* anc_out,
                              int32_t ld_anc, float* scores_out, uint8_t* flag_out, int64_t* preds_out,
                              int32_t* n_finished, void* stream);

/* ---- fused decoder layer for few rows (decoder/transformer.py:47-90 TransformerDecoderLayer.forward, post-norm, and :161-183
 *      the stack around it; module/attention.py:60-84,120-145; module/ffn.py:38-41 'glu'), csrc/declayer.hip.  d_model 256, 4 heads
 *      of 64, 16-bit operands, L <= 32 decoder rows per utterance.  Three launches per layer, cut along (utterance group, head) /
 *      (32-row block, hidden slice); a sub-layer leaves PARTIAL sums of its branch in `slabs` 16-BIT [n][R][256] (R = B * L rows) and the
 *      next launch finishes  y = LayerNorm(xres + dropout(sum_n slabs[n] + bias))  in its prologue -- described by otr_dec_ln_t --
 *      writing y / y16 / z / mean / rstd (each may be NULL) once per row.  nslab == 0: nothing to finish, the rows are taken from x16.
 *      Weight packs are otr_pack_frags packs with perm 0, rows = outputs (the forward packs of otr_rb_linear / otr_ffn_ln_fwd).
 * otr_dec_self_fwd:  q|k|v of every head (qkv16 [R,768], columns q | k | v), causal self-attention inside each utterance (ctx16 [R,256],
 *      lse f32 [B,4,L]), slabs [4][R][256] = per-head shares of ctx . W_o^T (no bias), 16-bit.
 * otr_dec_cross_fwd: q (q16 [R,256]) against the utterance's encoder keys / values, element (b, t, c) at kv[b*kv_bs + t*kv_ts + c], keys
 *      from column koff, values from voff (head h adds 64 h); key_mask uint8 [B,Tk] or NULL; ctx16, lse, slabs as above.
 * otr_dec_ffn_fwd:   slabs [S][R][256] = w_2 glu(w_1 y + b_1) over 1/S of the hidden units each (no b_2); F % (128 S) == 0.  hsave
 *      (may be NULL; otr_dec_ffn_hsave_bytes(R, F) bytes, opaque) receives (value + bias, sigmoid(gate)) of every hidden unit for
 *      otr_dec_ffn_bwd.  Every slab is 16-bit (the shares are rounded once and added up in fp32 by the consumer): the prologues are
 *      bound by what a CU can ingest, and the slabs were most of it.
 * otr_dec_ln:        the LayerNorm alone (closes the last layer). */
typedef struct {
  const float* xres; const void* x16; const void* slabs; int32_t nslab;
  const float* bias; const float* gamma; const float* beta; const uint64_t* seed;
  float p_drop, eps; uint64_t rng_offset;
  float* y; void* y16; float* z; float* mean; float* rstd;
} otr_dec_ln_t;
int32_t otr_rb_linear_ln(const otr_dec_ln_t* ln, const void* w_pack, const float* bias, void* out, int32_t out_dtype, int64_t ldo, int64_t M,
                         int32_t N, int32_t K, void* stream);
/* utterances per (group, head) workgroup of the four attention launches for a batch of B utterances x L decoder rows (host only): the
 * per-group partial buffers of otr_dec_cross_bwd / otr_dec_self_bwd have ceil(B / this) rows.  0 for shapes the launches refuse. */
int32_t otr_dec_group_size(int32_t B, int32_t L);
int32_t otr_dec_self_fwd(const otr_dec_ln_t* ln, int32_t B, int32_t L, const void* wqkv_pack, const float* bqkv, const void* wo_pack,
                         void* qkv16, void* ctx16, float* lse, void* slabs, void* stream);
/* otr_dec_self_step: the self-attention sub-layer of ONE cached beam-search step (recognize/speech2text.py:95-146 with the KV cache the
 *      reference leaves as a TODO, README.md:13): R hypothesis rows, one new position *pos each.  Finishes `ln` (the layer below), projects
 *      q | k | v, appends k, v to kcache / vcache [R, maxlen, 256] at position *pos, attends over positions 0..*pos of the row's ancestors
 *      (anc int32 [R, maxlen]: anc[r][j] = the row whose cache holds position j of r's prefix; as otr_decode_self_attention), and leaves
 *      slabs [4][R][256] = per-head shares of ctx . W_o^T for the next launch's prologue.  Replaces otr_dec_ln + otr_linear_fwd +
 *      otr_decode_self_attention + otr_proj_ln_fwd of the step. */
int32_t otr_dec_self_step(const otr_dec_ln_t* ln, int64_t R, const void* wqkv_pack, const float* bqkv, const void* wo_pack, void* kcache,
                          void* vcache, const int32_t* anc, const int32_t* pos, int32_t maxlen, void* slabs, void* stream);
int32_t otr_dec_cross_fwd(const otr_dec_ln_t* ln, int32_t B, int32_t L, const void* wq_pack, const float* bq, const void* wo_pack,
                          const void* kv, int64_t kv_bs, int64_t kv_ts, int32_t koff, int32_t voff, const uint8_t* key_mask, int32_t Tk,
                          void* q16, void* ctx16, float* lse, void* slabs, void* stream);
int64_t otr_dec_ffn_hsave_bytes(int64_t R, int32_t F);
int32_t otr_dec_ffn_fwd(const otr_dec_ln_t* ln, int64_t R, const void* w1_pack, const float* b1, const void* w2_pack, int32_t F, int32_t S,
                        void* slabs, void* hsave, void* stream);
int32_t otr_dec_ln(const otr_dec_ln_t* ln, int64_t R, void* stream);
/* PAIR launches (r06): two independent problems of the same launch in ONE grid -- the decoder's and the language model's layer of a
 * beam-search step (recognize/speech2text.py:100-113 runs them one after the other; they only meet in the top-k).  Block ids
 * 0 .. n_a - 1 work on problem a, the rest on b; each descriptor holds exactly the arguments of the single entry.  The LM chain then
 * rides in the decoder's launches: no second stream, no branch in the captured step. */
typedef struct {
  otr_dec_ln_t ln; int64_t R; const void* wqkv_pack; const float* bqkv; const void* wo_pack; void* kcache; void* vcache;
  const int32_t* anc; const int32_t* pos; int32_t maxlen; void* slabs;
} otr_dec_self_step_t;
typedef struct {
  otr_dec_ln_t ln; int64_t R; const void* w1_pack; const float* b1; const void* w2_pack; int32_t F, S; void* slabs; void* hsave;
} otr_dec_ffn_fwd_t;
int32_t otr_dec_self_step_pair(const otr_dec_self_step_t* a, const otr_dec_self_step_t* b, void* stream);
int32_t otr_dec_ffn_fwd_pair(const otr_dec_ffn_fwd_t* a, const otr_dec_ffn_fwd_t* b, void* stream);
int32_t otr_dec_ln_pair(const otr_dec_ln_t* a, int64_t Ra, const otr_dec_ln_t* b, int64_t Rb, void* stream);
/* Backward, the same cut mirrored (csrc/declayer.hip).  The gradient of a sub-layer's LayerNorm output arrives as dskip f32 [R,256]
 * (may be NULL) plus nslab partial slabs; every launch finishes it and runs that LayerNorm's backward in its prologue
 * (otr_dec_lnb_t; z / mean / rstd as saved by the forward prologue), writing -- once per row -- dz (the gradient of the residual input =
 * the skip part for the launch below), da16 (the dropout-masked branch gradient: weight-gradient operand of output_proj / w_2) and
 * partial f32 [row blocks][3][256] = per-block sums of dgamma | dbeta | d branch-bias (column-sum them; a row block is a group for the
 * attention launches (ceil(B / (32 / L)) of them), 32 rows for the FFN launch).
 * otr_dec_ffn_bwd:   the hidden comes from hsave (otr_dec_ffn_fwd); dh [R,2F], u [R,F] (weight-gradient operands), db1_part
 *      [ceil(R/32)][2F] (column-sum for the w_1 bias gradient), slabs 16-BIT [S][R][256] = shares of dh . w_1.  Packs as otr_ffn_bwd.
 *      (all slabs 16-bit, as in the forward launches)
 * otr_dec_cross_bwd: wo / wq input-gradient packs (otr_pack_frags of W as A[k][n]); q16, ctx16, lse as the forward left them; dkv has kv's
 *      geometry and receives d keys / d values of this layer's columns (every (utterance, key) once); dq16 [R,256]; slabs [4][R][256] =
 *      per-head shares of dq . W_q.
 * otr_dec_self_bwd:  dqkv16 [R,768]; slabs [4][R][256] = per-head shares of dqkv . W_qkv.
 * otr_dec_sum:       out = skip + sum of slabs (the gradient that leaves the stack). */
typedef struct {
  const float* dskip; const void* slabs; int32_t nslab;
  const float* z; const float* mean; const float* rstd; const float* gamma; const uint64_t* seed;
  float p_drop; uint64_t rng_offset;
  float* dz; void* da16; float* partial;
} otr_dec_lnb_t;
int32_t otr_dec_ffn_bwd(const otr_dec_lnb_t* ln, int64_t R, const void* hsave, const void* w2t_pack, const void* w1t_pack, int32_t F,
                        int32_t S, void* dh, void* u, float* db1_part, void* slabs, void* stream);
int32_t otr_dec_cross_bwd(const otr_dec_lnb_t* ln, int32_t B, int32_t L, const void* wo_dgrad_pack, const void* wq_dgrad_pack, const void* q16,
                          const void* ctx16, const float* lse, const void* kv, void* dkv, int64_t kv_bs, int64_t kv_ts, int32_t koff,
                          int32_t voff, const uint8_t* key_mask, int32_t Tk, void* dq16, void* slabs, void* stream);
int32_t otr_dec_self_bwd(const otr_dec_lnb_t* ln, int32_t B, int32_t L, const void* wo_dgrad_pack, const void* wqkv_dgrad_pack,
                         const void* qkv16, const void* ctx16, const float* lse, void* dqkv16, void* slabs, void* stream);
int32_t otr_dec_sum(const float* skip, const void* slabs, int32_t nslab, int64_t R, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* OTRANS_HIP_H */
