// C-ABI entry points: error plumbing, nn.Linear family (on the GEMM core), conv2 implicit GEMMs.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

#include "gemm_kernel.h"
#include "wgrad256.h"

static thread_local char g_err[512] = "";

void otr_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int32_t otr_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    otr_set_error("%s: launch failed: %s", what, hipGetErrorString(e));
    return (int32_t)e;
  }
  return 0;
}

static __global__ void zero_f32_kernel(float* p, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = 0.f;
}
void otr_zero_f32(float* p, int64_t n, hipStream_t s) {
  if (n <= 0) return;
  unsigned g = (unsigned)((n + 255) / 256 > 1024 ? 1024 : (n + 255) / 256);
  hipLaunchKernelGGL(zero_f32_kernel, dim3(g), dim3(256), 0, s, p, n);
}

int g_otr_force_tile = 0;
int g_otr_force_ksplit = 0;
int g_otr_gemm_xcd_map = 1;        // gemm_kernel.h gemm_tile_of (otr_debug_set(26, 0) = natural tile order)
int g_otr_bias_vec4 = 1;           // attention.hip: relative-position score bias as 16-byte loads where the rows allow (otr_debug_set(27, 0) = scalar)
int g_otr_gemm_resident64 = 1024;   // gemm_kernel.h: resident workgroups of the persistent 64 x 64-tile GEMM (otr_debug_set(28, v); 512 = r05)
int g_otr_conv2_wgrad256 = 1;       // conv2 weight gradient of a C1 % 256 == 0 frontend on wgrad256.hip's gather form (otr_debug_set(29, 0) = the transposing GEMM)
int g_otr_attn_enc96 = 1;          // encattn96.hip: the Conformer's attention backward on the whole-utterance kernel (otr_debug_set(33, 0) = the streamed dQ / dK,dV pair)
int g_otr_im2k_fast = 1;            // gemm_kernel.h TileLoader RAWK (otr_debug_set(34, 0) = the bounds-checked loader)
int g_otr_force_generic = 0;
int g_otr_no_persist = 0;
int g_otr_ffn2_ablate = 0;   // tuning hook (otr_debug_set(4, v)): bit 0 = no weight DMA after the first chunk, bit 1 = no MFMA work
int g_otr_wgrad256 = -1;     // 256x256-tile weight-gradient launch (wgrad256.hip): -1 = environment OTR_WGRAD256 (default on), 0 / 1 (otr_debug_set(6, v))
int g_otr_wgrad256_ablate = 0;   // tuning hook (otr_debug_set(8, v)), see wgrad256.h
int g_otr_wgrad256_min_k = 96;      // narrowest x operand the 256-wide launch takes (otr_debug_set(37, v); 128 = round 5: the relative-position attention's 504 x 96 products then stay on the 128-wide grouped kernel)
int g_otr_wgrad256_min_rows = 256;   // shortest contraction the 256-wide launch takes (otr_debug_set(9, v))
extern int g_otr_conv2_dgrad_ablate;   // conv.hip (otr_debug_set(10, v))
extern int g_otr_conv2_dgrad_wide;     // conv.hip (otr_debug_set(30, v))
extern int g_otr_conv2_wide;           // conv.hip (otr_debug_set(31, v))
extern int g_otr_conv2wide_ablate;     // conv2wide.hip (otr_debug_set(32, v))
int g_otr_wgrad256_grid = 0; // workgroups of that launch; 0 = one per CU (otr_debug_set(7, v))
int g_otr_conv2_fwd_direct = 1;  // conv2 forward on the weight-stationary kernel where it serves (otr_debug_set(22, v))
int g_otr_attn_enc = 1;          // encoder-shape attention backward with the whole (utterance, head) in LDS (otr_debug_set(21, v))
int g_otr_rb_waves8 = 1;         // 256-column row-block kernels on 8-wave workgroups (otr_debug_set(19, v))
int g_otr_rb_nsplit = 1;         // q|k|v row-block projection: 1 = two workgroups per row block, 384 columns each (otr_debug_set(18, v))
int g_otr_conv1_stencil = 0;     // 1: conv1 forward on the VALU stencil instead of the fp32 matrix pipe (otr_debug_set(17, v))
int g_otr_attn_waves8 = 1;       // merged attention backward on 8-wave workgroups (128 queries / keys each; otr_debug_set(20, v))
int g_otr_attn_xmap = 1;         // attention launches: the blocks of one (head, utterance) on one XCD (otr_debug_set(16, v))
int g_otr_attn_bwd_split = 0;    // attention backward as two launches (dQ, then dK/dV) instead of one (otr_debug_set(13, v))
int g_otr_beam_reg = 1;          // beam_topk with the row in registers and wave-level arg-max rounds (beam.hip; otr_debug_set(25, v))
int g_otr_decode_attn64 = 1;     // cached decode self-attention: 16-byte loads, all value rows of a chunk in flight (decode.hip; otr_debug_set(24, v))
int g_otr_dec_group = 0;         // fused decoder: utterances per (group, head) workgroup; 0 = declayer.hip's choice, > 0 = at most that many (otr_debug_set(23, v))
int g_otr_ffn_map = 1;           // split FFN kernels: workgroup -> (row block, slice) mapping, ffn3.hip f3_block_map (otr_debug_set(15, v))
int g_otr_ffn_coh_only = 0;      // split FFN kernels: 1 = every partial-sum transfer writes through / reads past the L2 (otr_debug_set(12, v))
int g_otr_spin_limit = 1 << 22;  // bound of every in-kernel turnstile / arrival spin (otr_debug_set(11, v): tests force a give-up with 1)
int32_t* g_otr_fault = nullptr;  // device word the spin-bounded kernels add 1 to when they give up (otr_set_fault_counter)
unsigned long long* g_otr_trace = nullptr;
extern "C" int32_t otr_debug_trace(void* buf) { g_otr_trace = (unsigned long long*)buf; return 0; }
extern "C" int32_t otr_debug_set(int32_t key, int32_t value) {
  if (key == 0) g_otr_force_tile = value;
  else if (key == 1) g_otr_force_ksplit = value;
  else if (key == 26) g_otr_gemm_xcd_map = value;
  else if (key == 27) g_otr_bias_vec4 = value;
  else if (key == 29) g_otr_conv2_wgrad256 = value;
  else if (key == 30) g_otr_conv2_dgrad_wide = value;
  else if (key == 31) g_otr_conv2_wide = value;
  else if (key == 32) g_otr_conv2wide_ablate = value;
  else if (key == 33) g_otr_attn_enc96 = value;
  else if (key == 34) g_otr_im2k_fast = value;
  else if (key == 37) g_otr_wgrad256_min_k = value > 0 ? value : 96;
  else if (key == 28) g_otr_gemm_resident64 = value > 0 ? value : 512;
  else if (key == 2) g_otr_force_generic = value;
  else if (key == 3) g_otr_no_persist = value;
  else if (key == 4) g_otr_ffn2_ablate = value;
  else if (key == 5) { /* retired (8-wave form of the 32-row FFN kernel) */ }
  else if (key == 6) g_otr_wgrad256 = value;
  else if (key == 7) g_otr_wgrad256_grid = value;
  else if (key == 8) g_otr_wgrad256_ablate = value;
  else if (key == 9) g_otr_wgrad256_min_rows = value;
  else if (key == 10) g_otr_conv2_dgrad_ablate = value;
  else if (key == 11) g_otr_spin_limit = value > 0 ? value : 1 << 22;
  else if (key == 12) g_otr_ffn_coh_only = value;
  else if (key == 13) g_otr_attn_bwd_split = value;
  else if (key == 16) g_otr_attn_xmap = value;
  else if (key == 20) g_otr_attn_waves8 = value;
  else if (key == 17) g_otr_conv1_stencil = value;
  else if (key == 18) g_otr_rb_nsplit = value;
  else if (key == 19) g_otr_rb_waves8 = value;
  else if (key == 21) g_otr_attn_enc = value;
  else if (key == 22) g_otr_conv2_fwd_direct = value;
  else if (key == 15) g_otr_ffn_map = value;
  else if (key == 23) g_otr_dec_group = value;
  else if (key == 24) g_otr_decode_attn64 = value;
  else if (key == 25) g_otr_beam_reg = value;
  else { otr_set_error("debug_set: unknown key %d", key); return -1; }
  return 0;
}

extern "C" int32_t otr_set_fault_counter(void* device_word) { g_otr_fault = (int32_t*)device_word; return 0; }

extern "C" int32_t otr_version(void) { return OTR_ABI_VERSION; }
extern "C" int32_t otr_half_type(void) { return OTR_H16; }
extern "C" const char* otr_last_error_string(void) { return g_err; }

static inline int esize(int dtype) { return dtype == OTR_F32 ? 4 : 2; }
static inline bool dtype_ok(int d) { return d == OTR_F32 || d == OTR_H16; }
// vector path of the row-major loaders: 16-byte aligned base and rows
static inline int kc_vec(const void* p, int64_t ld, int dtype) {
  return ((uintptr_t)p % 16 == 0) && (ld % (16 / esize(dtype)) == 0);
}
// PM-element vector path of the transposing loaders (PM = 4 rows for bf16 compute, 2 for fp32)
static inline int mc_vec(const void* p, int64_t ld, int dtype, int compute) {
  int pm = compute == OTR_H16 ? 4 : 2;
  return ((uintptr_t)p % (pm * esize(dtype)) == 0) && (ld % pm == 0);
}

static int32_t run_gemm(const GemmArgs& a, int compute, int ad, int bd, int cd, int amode, int bmode, void* stream) {
  if (a.M <= 0 || a.N <= 0) return 0;
  OTR_REQUIRE(a.K > 0, "gemm: K must be positive (got %d)", a.K);
  hipStream_t s = (hipStream_t)stream;
  if (compute == OTR_H16) return gemm_dispatch_bf16(a, ad, bd, cd, amode, bmode, s);
  if (compute == OTR_F32) return gemm_dispatch_f32(a, ad, bd, cd, amode, bmode, s);
  otr_set_error("gemm: bad compute type %d", compute);
  return -1;
}

static int32_t check_linear(const otr_linear_desc_t* d) {
  OTR_REQUIRE(d != nullptr, "linear: null descriptor");
  OTR_REQUIRE(d->M >= 0 && d->N >= 0 && d->K > 0, "linear: bad shape M=%d N=%d K=%d", d->M, d->N, d->K);
  OTR_REQUIRE(dtype_ok(d->x_dtype) && dtype_ok(d->w_dtype) && dtype_ok(d->y_dtype), "linear: bad dtype code");
  OTR_REQUIRE(d->ldx >= d->K && d->ldw >= d->K && d->ldy >= d->N, "linear: leading dimension smaller than row");
  return 0;
}

extern "C" int32_t otr_linear_fwd(const otr_linear_desc_t* d, const void* x, const void* w, const float* bias,
                                  void* y, void* workspace, int64_t workspace_bytes, void* stream) {
  if (int32_t e = check_linear(d)) return e;
  OTR_REQUIRE(x && w && y, "linear_fwd: null pointer");
  GemmArgs a{};
  a.A = x; a.B = w; a.C = y; a.bias = bias;
  a.M = d->M; a.N = d->N; a.K = d->K;
  a.lda = d->ldx; a.ldb = d->ldw; a.ldc = d->ldy;
  a.act = d->act; a.accumulate = d->accumulate;
  a.a_vec = kc_vec(x, d->ldx, d->x_dtype);
  a.b_vec = kc_vec(w, d->ldw, d->w_dtype);
  a.allow_split = 1; a.ws = (float*)workspace; a.ws_bytes = workspace_bytes; a.trace = g_otr_trace;
  return run_gemm(a, d->compute, d->x_dtype, d->w_dtype, d->y_dtype, MODE_KC, MODE_KC, stream);
}

extern "C" int32_t otr_linear_fwd_batched(const otr_linear_desc_t* d, const void* x, const void* w, void* y, int32_t nbatch, int64_t bsx,
                                          int64_t bsw, int64_t bsy, void* stream) {
  if (int32_t e = check_linear(d)) return e;
  OTR_REQUIRE(x && w && y, "linear_fwd_batched: null pointer");
  OTR_REQUIRE(nbatch >= 1 && nbatch <= 65535 && bsx >= 0 && bsw >= 0 && bsy > 0, "linear_fwd_batched: bad batch (n=%d)", nbatch);
  OTR_REQUIRE(d->act == OTR_ACT_NONE && !d->accumulate, "linear_fwd_batched: plain products only");
  if (nbatch == 1) return 1;
  GemmArgs a{};
  a.A = x; a.B = w; a.C = y; a.bias = nullptr;
  a.M = d->M; a.N = d->N; a.K = d->K;
  a.lda = d->ldx; a.ldb = d->ldw; a.ldc = d->ldy;
  a.act = OTR_ACT_NONE; a.accumulate = 0;
  a.a_vec = kc_vec(x, d->ldx, d->x_dtype);
  a.b_vec = kc_vec(w, d->ldw, d->w_dtype);
  a.allow_split = 0; a.ws = nullptr; a.ws_bytes = 0; a.trace = nullptr;
  a.nbatch = nbatch; a.bsa = bsx * esize(d->x_dtype); a.bsb = bsw * esize(d->w_dtype); a.bsc = bsy * esize(d->y_dtype);
  if (d->compute != OTR_H16) return 1;                                 // 16-bit compute only; 1 = not served
  return run_gemm(a, d->compute, d->x_dtype, d->w_dtype, d->y_dtype, MODE_KC, MODE_KC, stream);
}

extern "C" int32_t otr_linear_dgrad(const otr_linear_desc_t* d, const void* dy, const void* w, void* dx,
                                    void* workspace, int64_t workspace_bytes, void* stream) {
  if (int32_t e = check_linear(d)) return e;
  OTR_REQUIRE(dy && w && dx, "linear_dgrad: null pointer");
  GemmArgs a{};  // dx[M,K] = dy[M,N] * w[N,K]: contraction over N; w is "rows(K)-contiguous"
  a.A = dy; a.B = w; a.C = dx; a.bias = nullptr;
  a.M = d->M; a.N = d->K; a.K = d->N;
  a.lda = d->ldy; a.ldb = d->ldw; a.ldc = d->ldx;
  a.act = OTR_ACT_NONE; a.accumulate = d->accumulate;
  a.a_vec = kc_vec(dy, d->ldy, d->y_dtype);
  a.b_vec = mc_vec(w, d->ldw, d->w_dtype, d->compute);
  a.allow_split = 1; a.ws = (float*)workspace; a.ws_bytes = workspace_bytes; a.trace = g_otr_trace;
  return run_gemm(a, d->compute, d->y_dtype, d->w_dtype, d->x_dtype, MODE_KC, MODE_MC, stream);
}

extern "C" int32_t otr_linear_wgrad(const otr_linear_desc_t* d, const void* dy, const void* x, void* dw,
                                    void* workspace, int64_t workspace_bytes, void* stream) {
  if (int32_t e = check_linear(d)) return e;
  OTR_REQUIRE(dy && x && dw, "linear_wgrad: null pointer");
  GemmArgs a{};  // dw[N,K] = dy[M,N]^T * x[M,K]: contraction over M; both operands rows-contiguous
  a.A = dy; a.B = x; a.C = dw; a.bias = nullptr;
  a.M = d->N; a.N = d->K; a.K = d->M;
  a.lda = d->ldy; a.ldb = d->ldx; a.ldc = d->ldw;
  a.act = OTR_ACT_NONE; a.accumulate = d->accumulate;
  a.a_vec = mc_vec(dy, d->ldy, d->y_dtype, d->compute);
  a.b_vec = mc_vec(x, d->ldx, d->x_dtype, d->compute);
  a.allow_split = 1; a.ws = (float*)workspace; a.ws_bytes = workspace_bytes; a.trace = g_otr_trace;
  if (d->M == 0) return 0;
  return run_gemm(a, d->compute, d->y_dtype, d->x_dtype, d->w_dtype, MODE_MC, MODE_MC, stream);
}

// ------------------------------------------------------------------------------------------------
// FFN forward through w_1 and the GLU in one launch: h = x . w1^T + b1 (saved for backward), u = h[:, :F] * sigmoid(h[:, F:]).
// Returns 1 (nothing launched) when the operands do not qualify; the caller then runs otr_linear_fwd + otr_glu_fwd.
extern "C" int32_t otr_ffn_glu_fwd(const void* x, int64_t ldx, const void* w1, int64_t ldw, const float* b1, void* h, void* u,
                                   int32_t M, int32_t F, int32_t d_model, void* stream) {
  OTR_REQUIRE(x && w1 && h && u, "ffn_glu_fwd: null pointer");
  OTR_REQUIRE(M >= 0 && F > 0 && d_model > 0 && ldx >= d_model && ldw >= d_model, "ffn_glu_fwd: bad shape");
  if (M == 0) return 0;
  const int64_t t128 = (int64_t)((M + 127) / 128) * ((2 * F + 127) / 128);
  const bool big = M >= 128 && F >= 128 && t128 >= 256 && g_otr_force_tile != 64;
  const int half = big ? 64 : 32;                          // value columns per tile
  const bool ok = kc_vec(x, ldx, OTR_H16) && kc_vec(w1, ldw, OTR_H16) && d_model % 8 == 0 && F % half == 0 &&
                  (uintptr_t)h % 16 == 0 && (uintptr_t)u % 16 == 0 && g_otr_force_generic == 0;
  if (!ok) return 1;
  GemmArgs a{};
  a.A = x; a.B = w1; a.C = h; a.bias = b1;
  a.M = M; a.N = 2 * F; a.K = d_model;
  a.lda = ldx; a.ldb = ldw; a.ldc = 2 * (int64_t)F;
  a.act = OTR_ACT_GLU_FWD; a.accumulate = 0;
  a.a_vec = 1; a.b_vec = 1;
  a.allow_split = 0; a.ws = nullptr; a.ws_bytes = 0; a.trace = g_otr_trace;
  a.aux_out = u;
  const int keep = g_otr_force_tile;
  g_otr_force_tile = big ? 128 : 64;                       // F % (tile/2) == 0 was checked for this tile width
  const int32_t e = run_gemm(a, OTR_H16, OTR_H16, OTR_H16, OTR_H16, MODE_KC, MODE_KC, stream);
  g_otr_force_tile = keep;
  return e;
}

// ------------------------------------------------------------------------------------------------
// FFN backward through w_2 and the GLU in one launch: du = dy . w2 (never stored), dh = GLU'(h) * du, plus per-row-tile
// column sums of dh (the w_1 bias gradient).  Returns 1 (nothing launched) when the operands do not qualify for the
// fused kernel; the caller then runs otr_linear_dgrad / otr_glu_bwd.
extern "C" int32_t otr_ffn_glu_bwd(const void* dy, int32_t dy_dtype, int64_t ldy, const void* w2t, int64_t ldw, const void* h,
                                   int32_t h_has_sigmoid, void* dh, float* dbias_partial, int32_t partial_rows_cap,
                                   int32_t* partial_rows, int32_t M, int32_t F, int32_t d_model, void* stream) {
  OTR_REQUIRE(dy && w2t && h && dh && dbias_partial && partial_rows, "ffn_glu_bwd: null pointer");
  OTR_REQUIRE(M >= 0 && F > 0 && d_model > 0 && ldy >= d_model && ldw >= d_model, "ffn_glu_bwd: bad shape");
  *partial_rows = 0;
  if (M == 0) return 0;
  const bool ok = dy_dtype == OTR_H16 && kc_vec(dy, ldy, OTR_H16) && kc_vec(w2t, ldw, OTR_H16) && d_model % 8 == 0 &&
                  F % 8 == 0 && (uintptr_t)h % 16 == 0 && (uintptr_t)dh % 16 == 0 && g_otr_force_generic == 0;
  const int64_t t128 = (int64_t)((M + 127) / 128) * ((F + 127) / 128);
  const bool big = M >= 128 && F >= 128 && t128 >= 256 && g_otr_force_tile != 64;   // few tiles: 64x64 fills the chip better
  const int rows = (M + (big ? 127 : 63)) / (big ? 128 : 64);
  if (!ok || rows > partial_rows_cap) return 1;
  GemmArgs a{};
  a.A = dy; a.B = w2t; a.C = dh /* unused, must be aligned */; a.bias = nullptr;
  a.M = M; a.N = F; a.K = d_model;
  a.lda = ldy; a.ldb = ldw; a.ldc = 2 * (int64_t)F;
  a.act = OTR_ACT_GLU_BWD; a.accumulate = 0; a.aux_flag = h_has_sigmoid ? 1 : 0;
  a.a_vec = 1; a.b_vec = 1;
  a.allow_split = 0; a.ws = nullptr; a.ws_bytes = 0; a.trace = g_otr_trace;
  a.aux_in = h; a.aux_out = dh; a.aux_part = dbias_partial;
  const int keep = g_otr_force_tile;
  g_otr_force_tile = big ? 128 : 64;                  // the partial layout depends on the tile height: pin it
  const int32_t e = run_gemm(a, OTR_H16, OTR_H16, OTR_H16, OTR_H16, MODE_KC, MODE_KC, stream);
  g_otr_force_tile = keep;
  if (e) return e;
  *partial_rows = rows;
  return 0;
}

// ------------------------------------------------------------------------------------------------
// All weight gradients of a backward pass in (a few) grouped launches: dw_i[N,K] += dy_i[M,N]^T x_i[M,K].
// the problems the 256-wide launch takes: long contraction, 16-bit operands in 16-byte aligned rows, a few tiles at least
static bool wgrad256_ok(const otr_wgrad_item_t& it, int compute) {
  return compute == OTR_H16 && it.dy && it.x && it.dw && it.dy_dtype == OTR_H16 && it.x_dtype == OTR_H16 && it.M >= g_otr_wgrad256_min_rows && it.N >= 128 &&
         it.K >= g_otr_wgrad256_min_k && it.N % 8 == 0 && it.K % 8 == 0 && it.ldy >= it.N && it.ldx >= it.K && it.ldw >= it.K && it.ldy % 8 == 0 &&
         it.ldx % 8 == 0 && it.ldw % 4 == 0 && (uintptr_t)it.dy % 16 == 0 && (uintptr_t)it.x % 16 == 0 && (uintptr_t)it.dw % 16 == 0 &&
         it.ldy < (1ll << 24) && it.ldx < (1ll << 24) && (int64_t)(it.N + 256) * it.ldw * 4 < (1ll << 31) &&
         (!it.dbias || (uintptr_t)it.dbias % 4 == 0);
}
static void wgrad256_env() {
  if (g_otr_wgrad256 < 0) {
    const char* e = getenv("OTR_WGRAD256");
    g_otr_wgrad256 = (e && e[0] == '0') ? 0 : 1;
  }
}
extern "C" int32_t otr_wgrad256_takes(const otr_wgrad_item_t* item, int32_t compute) {
  wgrad256_env();
  return (item && g_otr_wgrad256 && wgrad256_ok(*item, compute)) ? 1 : 0;
}

extern "C" int32_t otr_linear_wgrad_grouped(const otr_wgrad_item_t* items, int32_t n, int32_t compute, void* workspace,
                                            int64_t workspace_bytes, void* stream) {
  OTR_REQUIRE(n >= 0 && (items || n == 0), "linear_wgrad_grouped: null items");
  OTR_REQUIRE(compute == OTR_H16 || compute == OTR_F32, "linear_wgrad_grouped: bad compute type");
  const int pm = compute == OTR_H16 ? 4 : 2;
  wgrad256_env();
  hipStream_t s = (hipStream_t)stream;
  // Long-contraction 16-bit problems whose output is made of whole 256 x 256 tiles: ONE persistent launch with 256-wide
  // tiles fed by direct-to-LDS DMA (wgrad256.hip); everything else stays on the 128 / 64-wide grouped kernel below.
  std::vector<char> taken((size_t)n, 0);
  if (g_otr_wgrad256 && compute == OTR_H16) {
    std::vector<W256Item> big;
    std::vector<int> idx;
    for (int i = 0; i < n; ++i) {
      const otr_wgrad_item_t& it = items[i];
      if (!(it.dy && it.x && it.dw)) continue;                        // reported below
      const bool ok = wgrad256_ok(it, compute);
      if (ok) idx.push_back(i);
    }
    // longest contraction first; problems of equal row count share a launch (its rounds schedule needs tiles of one length).  Narrow
    // problems (K < 128: a 256-column tile of theirs is mostly zero lines) go behind the wide ones of their row count and into launches of
    // their own: mixed in, they would push wide problems over the table's 56 entries into a small, badly filled tail launch
    auto narrow = [&](int a) { return items[a].K < 128 ? 1 : 0; };
    std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) {
      if (items[a].M != items[b].M) return items[a].M > items[b].M;
      return narrow(a) < narrow(b);
    });
    size_t c0 = 0;
    while (c0 < idx.size()) {
      size_t c1 = c0;
      while (c1 < idx.size() && c1 - c0 < (size_t)W256_MAX_PROBS && items[idx[c1]].M == items[idx[c0]].M && narrow(idx[c1]) == narrow(idx[c0])) ++c1;
      big.clear();
      for (size_t c = c0; c < c1; ++c) {
        const otr_wgrad_item_t& it = items[idx[c]];
        big.push_back(W256Item{it.dy, it.x, it.dw, it.dbias, it.M, it.N, it.K, it.ldy, it.ldx, it.ldw, it.overwrite != 0});
      }
      if (!workspace || wgrad256_workspace_bytes(big.data(), (int)big.size()) > workspace_bytes) break;   // the grouped kernel takes them
      if (int32_t e = wgrad256_launch(big.data(), (int)big.size(), workspace, workspace_bytes, g_otr_wgrad256_grid, g_otr_wgrad256_ablate, s)) return e;
      for (size_t c = c0; c < c1; ++c) taken[(size_t)idx[c]] = 1;
      c0 = c1;
    }
  }
  std::vector<int> order[16];   // key = big(1) | dy dtype(1) | x dtype(1)
  for (int i = 0; i < n; ++i) {
    if (taken[(size_t)i]) continue;
    const otr_wgrad_item_t& it = items[i];
    OTR_REQUIRE(!it.dbias, "linear_wgrad_grouped: item %d carries a bias gradient but does not run on the 256-wide kernel "
                "(check otr_wgrad256_takes first)", i);
    OTR_REQUIRE(it.dy && it.x && it.dw, "linear_wgrad_grouped: item %d has a null pointer", i);
    OTR_REQUIRE(it.M >= 0 && it.N > 0 && it.K > 0 && it.ldy >= it.N && it.ldx >= it.K && it.ldw >= it.K,
                "linear_wgrad_grouped: item %d has a bad shape", i);
    OTR_REQUIRE(dtype_ok(it.dy_dtype) && dtype_ok(it.x_dtype), "linear_wgrad_grouped: item %d has a bad dtype", i);
    if (it.M == 0) continue;
    const bool fast = mc_vec(it.dy, it.ldy, it.dy_dtype, compute) && mc_vec(it.x, it.ldx, it.x_dtype, compute) &&
                      it.N % pm == 0 && it.K % pm == 0 && it.M % (compute == OTR_H16 ? 8 : 4) == 0 &&
                      (uintptr_t)it.dw % 16 == 0 && it.ldw % 4 == 0 &&
                      it.ldy < (1ll << 31) && it.ldx < (1ll << 31) && it.ldw < (1ll << 31);
    if (!fast) {   // odd alignment: the stand-alone path (generic loaders, split-K through the workspace)
      otr_linear_desc_t d{};
      d.M = it.M; d.N = it.N; d.K = it.K;
      d.x_dtype = it.x_dtype; d.w_dtype = OTR_F32; d.y_dtype = it.dy_dtype; d.compute = compute;
      d.ldx = it.ldx; d.ldw = it.ldw; d.ldy = it.ldy; d.act = OTR_ACT_NONE; d.accumulate = 1;
      if (int32_t e = otr_linear_wgrad(&d, it.dy, it.x, it.dw, workspace, workspace_bytes, stream)) return e;
      continue;
    }
    // (K >= 96: the relative-position attention's dp products, 504 x 96 over 7968 rows of an fp32 operand, are bound by reading that
    //  operand -- twice on 64-wide tiles, once on 128-wide ones, a quarter of whose MFMA columns then idle: r05)
    const int big = (it.N >= 128 && it.K >= 96) ? 1 : 0;
    order[(big << 2) | ((it.dy_dtype != OTR_F32) << 1) | (it.x_dtype != OTR_F32)].push_back(i);   // dtype codes -> 0 / 1
  }
  for (int key = 4; key < 8; ++key) {
    // a group whose 128 x 128 tiles would not cover half the chip (the 4240 x 256 output layer over 480 rows: 68 tiles) runs on
    // 64 x 64 tiles instead: four times the workgroups
    int64_t t128 = 0;
    for (int i : order[key]) t128 += (int64_t)((items[i].N + 127) / 128) * ((items[i].K + 127) / 128);
    if (!order[key].empty() && t128 < 128) {
      order[key & 3].insert(order[key & 3].end(), order[key].begin(), order[key].end());
      order[key].clear();
    }
  }
  for (int key = 0; key < 8; ++key) {
    std::vector<int>& v = order[key];
    if (v.empty()) continue;
    // longest contraction first: the long tiles start early, the short ones fill the tail
    std::stable_sort(v.begin(), v.end(), [&](int a, int b) { return items[a].M > items[b].M; });
    std::vector<GroupDesc> d(v.size());
    for (size_t c = 0; c < v.size(); ++c) {
      const otr_wgrad_item_t& it = items[v[c]];
      d[c].A = it.dy; d[c].B = it.x; d[c].C = it.dw;
      d[c].M = it.N; d[c].N = it.K; d[c].K = it.M;
      d[c].lda = (int)it.ldy; d[c].ldb = (int)it.ldx; d[c].ldc = (int)it.ldw;
      d[c].a_vec = 1; d[c].b_vec = 1;
    }
    const int ad = ((key >> 1) & 1) ? OTR_H16 : OTR_F32, bd = (key & 1) ? OTR_H16 : OTR_F32, big = key >> 2;
    OTR_REQUIRE(workspace && workspace_bytes >= 4096, "linear_wgrad_grouped: needs a workspace (descriptor table)");
    int32_t e = compute == OTR_H16 ? gemm_grouped_wgrad_bf16(d.data(), (int)d.size(), ad, bd, big, workspace, workspace_bytes, s)
                                    : gemm_grouped_wgrad_f32(d.data(), (int)d.size(), ad, bd, big, workspace, workspace_bytes, s);
    if (e) return e;
  }
  return 0;
}

// ------------------------------------------------------------------------------------------------
// conv2 as implicit GEMM (frontend/conv.py:63-64 second Conv2dLayer)
static int32_t conv_geom(const otr_conv_desc_t* d, ConvGeom& g) {
  OTR_REQUIRE(d != nullptr, "conv: null descriptor");
  OTR_REQUIRE(d->B > 0 && d->T >= 7 && d->F >= 3, "conv: bad input shape B=%d T=%d F=%d", d->B, d->T, d->F);
  OTR_REQUIRE(d->T1 == (d->T - 3) / 2 + 1 && d->T2 == (d->T1 - 3) / 2 + 1, "conv: T1/T2 inconsistent with T");
  OTR_REQUIRE(d->F1 == (d->F - 1) / 2 + 1 && d->F2 == (d->F1 - 1) / 2 + 1, "conv: F1/F2 inconsistent with F");
  OTR_REQUIRE(d->C1 > 0 && d->C2 > 0 && d->C1 % 8 == 0, "conv: C1 must be a positive multiple of 8 (got %d)", d->C1);
  OTR_REQUIRE(dtype_ok(d->act_dtype), "conv: bad act dtype");
  OTR_REQUIRE((int64_t)d->B * d->T1 * d->F1 * d->C1 < (1ll << 31), "conv: act1 too large for 32-bit pixel index");
  g.T1 = d->T1; g.F1 = d->F1; g.C1 = d->C1; g.T2 = d->T2; g.F2 = d->F2;
  g.divF2 = make_fastdiv((uint32_t)d->F2);
  g.divT2 = make_fastdiv((uint32_t)d->T2);
  g.divC1 = make_fastdiv((uint32_t)d->C1);
  g.a1_elems = (int64_t)d->B * d->T1 * d->F1 * d->C1;
  return 0;
}

int32_t conv2_fwd_direct(const void* act1, const void* w2r, const float* b2, void* act2, int B, int T1, int F1, int T2, int F2, int C1, int C2,
                         int act_is_h16, int w_is_h16, hipStream_t stream);      // conv2fwd.hip: weight-stationary kernel (64 -> 128 channels)

extern "C" int32_t otr_conv2_fwd(const otr_conv_desc_t* d, const void* act1, const void* w2r, const float* b2,
                                 void* act2, void* stream) {
  GemmArgs a{};
  if (int32_t e = conv_geom(d, a.cg)) return e;
  OTR_REQUIRE(act1 && w2r && act2, "conv2_fwd: null pointer");
  if (b2) {
    const int32_t rc = conv2_fwd_direct(act1, w2r, b2, act2, d->B, d->T1, d->F1, d->T2, d->F2, d->C1, d->C2, d->act_dtype == OTR_H16,
                                        d->w_dtype == OTR_H16, (hipStream_t)stream);
    if (rc != 1) return rc;
  }
  a.A = act1; a.B = w2r; a.C = act2; a.bias = b2;
  a.M = d->B * d->T2 * d->F2; a.N = d->C2; a.K = 9 * d->C1;
  a.lda = 0; a.ldb = a.K; a.ldc = d->C2;
  a.act = OTR_ACT_RELU; a.accumulate = 0;
  a.a_vec = ((uintptr_t)act1 % 16 == 0);
  OTR_REQUIRE(dtype_ok(d->w_dtype), "conv2_fwd: bad w_dtype");
  a.b_vec = kc_vec(w2r, a.ldb, d->w_dtype);
  return run_gemm(a, d->compute, d->act_dtype, d->w_dtype, d->act_dtype, MODE_IM2K, MODE_KC, stream);
}

int32_t conv12_fwd_direct(const float* x, const float* w1, const float* b1, void* act1, const void* w2r, const float* b2, void* act2, int B, int T,
                          int F, int T1, int F1, int T2, int F2, int C1, int C2, int act_is_h16, int w_is_h16, hipStream_t stream);

extern "C" int32_t otr_conv12_fwd(const otr_conv_desc_t* d, const float* x, const float* w1, const float* b1, void* act1, const void* w2r,
                                  const float* b2, void* act2, void* stream) {
  ConvGeom g{};
  if (int32_t e = conv_geom(d, g)) return e;
  OTR_REQUIRE(x && w1 && b1 && act1 && w2r && b2 && act2, "conv12_fwd: null pointer");
  return conv12_fwd_direct(x, w1, b1, act1, w2r, b2, act2, d->B, d->T, d->F, d->T1, d->F1, d->T2, d->F2, d->C1, d->C2, d->act_dtype == OTR_H16,
                           d->w_dtype == OTR_H16, (hipStream_t)stream);
}

extern "C" int32_t otr_conv2_dgrad_cols(const otr_conv_desc_t* d, const void* dact2, const void* w2r, void* dcol,
                                        void* stream) {
  GemmArgs a{};
  if (int32_t e = conv_geom(d, a.cg)) return e;
  OTR_REQUIRE(dact2 && w2r && dcol, "conv2_dgrad_cols: null pointer");
  a.A = dact2; a.B = w2r; a.C = dcol; a.bias = nullptr;  // dcol[M2, 9*C1] = dact2[M2,C2] * w2r[C2, 9*C1]
  a.M = d->B * d->T2 * d->F2; a.N = 9 * d->C1; a.K = d->C2;
  a.lda = d->C2; a.ldb = 9 * d->C1; a.ldc = 9 * d->C1;
  a.act = OTR_ACT_NONE; a.accumulate = 0;
  a.a_vec = kc_vec(dact2, a.lda, d->act_dtype);
  OTR_REQUIRE(dtype_ok(d->w_dtype), "conv2_dgrad_cols: bad w_dtype");
  a.b_vec = mc_vec(w2r, a.ldb, d->w_dtype, d->compute);
  return run_gemm(a, d->compute, d->act_dtype, d->w_dtype, d->act_dtype, MODE_KC, MODE_MC, stream);
}

extern "C" int32_t otr_conv2_wgrad(const otr_conv_desc_t* d, const void* dact2, const void* act1, float* dw2r,
                                   void* workspace, int64_t workspace_bytes, void* stream) {
  GemmArgs a{};
  if (int32_t e = conv_geom(d, a.cg)) return e;
  OTR_REQUIRE(dact2 && act1 && dw2r, "conv2_wgrad: null pointer");
  if (g_otr_conv2_wgrad256 && d->act_dtype == OTR_H16 && d->compute == OTR_H16) {
    // wide frontends (C1 a multiple of 256: the Conformer's 256 -> 256): the 256 x 256-tile kernel with gathered x rows (wgrad256.hip);
    // the transposing GEMM below ran this 178 GFLOP problem in 631 us
    const int32_t rc = wgrad256_conv_launch(dact2, act1, dw2r, d->B, d->T1, d->F1, d->T2, d->F2, d->C1, d->C2, workspace, workspace_bytes,
                                            (hipStream_t)stream);
    if (rc != 1) return rc;
  }
  a.A = dact2; a.B = act1; a.C = dw2r; a.bias = nullptr;  // dw2r[C2, 9*C1] = dact2^T * im2col(act1)
  a.M = d->C2; a.N = 9 * d->C1; a.K = d->B * d->T2 * d->F2;
  a.lda = d->C2; a.ldb = 0; a.ldc = 9 * d->C1;
  a.act = OTR_ACT_NONE; a.accumulate = 0;
  a.a_vec = mc_vec(dact2, a.lda, d->act_dtype, d->compute);
  a.b_vec = ((uintptr_t)act1 % 16 == 0);
  a.allow_split = 1; a.ws = (float*)workspace; a.ws_bytes = workspace_bytes; a.trace = g_otr_trace;
  return run_gemm(a, d->compute, d->act_dtype, d->act_dtype, OTR_F32, MODE_MC, MODE_IM2M, stream);
}
