// CTC loss (alpha/beta recursion) with the gradient taken straight to the logits, i.e. fused with the
// log_softmax backward.  Replaces nn.CTCLoss(blank=0, zero_infinity=True) (third-party torch kernel)
// as used by model/ctc.py:30,50-53; default reduction 'mean' = mean_b( nll_b / max(tgt_len_b, 1) ).
//
// One workgroup per utterance, one thread per extended-label state s (S = 2L+1 <= 256); the
// previous time step lives in a double-buffered LDS row, alpha is kept in a caller workspace for
// the beta pass.  d loss / d logit[b,t,c] = coef_b * ( softmax[b,t,c] - sum_{s: ext_s=c} gamma_t(s) ),
// gamma_t(s) = exp(alpha_t(s) + beta_t(s) - lp[t,ext_s] + nll_b),  coef_b = 1 / (B * max(L_b,1)).
#include "common.h"

#define NEG_INF (-__builtin_huge_valf())

__device__ __forceinline__ float lse3(float a, float b, float c) {
  float m = fmaxf(a, fmaxf(b, c));
  if (m == NEG_INF) return NEG_INF;
  return m + logf(expf(a - m) + expf(b - m) + expf(c - m));
}

// dense part: dlogits = coef_b * exp(lp) for t < in_len[b], else 0; also zeroes *loss
__global__ void ctc_dense_kernel(const float* lp, float* dlogits, const int32_t* in_len, const int32_t* tgt_len,
                                 int B, int T, int V, float* loss) {
  const int64_t total = (int64_t)B * T * V;
  if (blockIdx.x == 0 && threadIdx.x == 0) *loss = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t bt = i / V;
    int b = (int)(bt / T), t = (int)(bt - (int64_t)b * T);
    float coef = 1.f / ((float)B * (float)max(tgt_len[b], 1));
    dlogits[i] = (t < in_len[b]) ? coef * expf(lp[i]) : 0.f;
  }
}

__global__ __launch_bounds__(256) void ctc_alpha_beta_kernel(const float* lp, const int64_t* targets, int64_t ldt,
                                                            const int32_t* in_len, const int32_t* tgt_len, int B, int T,
                                                            int V, int blank, float* alpha_ws, int Smax, float* nll_out,
                                                            float* loss, float* dlogits) {
  __shared__ float sh[2][256];
  __shared__ float s_ll;
  const int b = blockIdx.x, s = threadIdx.x;
  // a target length beyond the caller's bound (Smax = 2*max_tgt + 1 alpha slots per frame, `targets` rows of at least
  // max_tgt labels) would read `targets` out of bounds and write alpha rows over neighbouring (t, b) slots: such an
  // utterance is treated like an infeasible alignment (nll 0, zero gradient: zero_infinity semantics), never computed
  const int Lraw = tgt_len[b];
  const bool bad_len = Lraw < 0 || 2 * Lraw + 1 > Smax;
  const int Tb = max(min(in_len[b], T), 0), L = bad_len ? 0 : Lraw, S = 2 * L + 1;
  const float* lpb = lp + (int64_t)b * T * V;
  float* aw = alpha_ws + (int64_t)b * T * Smax;
  const int64_t* tg = targets + (int64_t)b * ldt;
  const bool live = s < S;
  const int ext = (live && (s & 1)) ? (int)tg[s >> 1] : blank;
  const int ext_m2 = (live && s >= 2 && (s & 1)) ? (int)tg[(s - 2) >> 1] : blank;
  const int ext_p2 = (s + 2 < S && (s & 1)) ? (int)tg[(s + 2) >> 1] : blank;
  const bool skip_b = live && s >= 2 && ext != blank && ext != ext_m2;       // s-2 -> s allowed
  const bool skip_f = (s + 2 < S) && ext_p2 != blank && ext_p2 != ext;       // s -> s+2 allowed
  const float coef = 1.f / ((float)B * (float)max(L, 1));

  // ---------------- alpha
  float a = NEG_INF;
  if (Tb > 0) {
    if (s == 0) a = lpb[blank];
    else if (s == 1 && S > 1) a = lpb[ext];
  }
  sh[0][s] = a;
  if (live && Tb > 0) aw[s] = a;
  for (int t = 1; t < Tb; ++t) {
    __syncthreads();
    const float* prev = sh[(t - 1) & 1];
    float a0 = prev[s];
    float a1 = s >= 1 ? prev[s - 1] : NEG_INF;
    float a2 = skip_b ? prev[s - 2] : NEG_INF;
    a = live ? lse3(a0, a1, a2) + lpb[(int64_t)t * V + ext] : NEG_INF;
    sh[t & 1][s] = a;
    if (live) aw[(int64_t)t * Smax + s] = a;
  }
  __syncthreads();
  if (s == 0) {
    float ll = NEG_INF;
    if (Tb > 0) {
      const float* fin = sh[(Tb - 1) & 1];
      ll = lse3(fin[S - 1], S > 1 ? fin[S - 2] : NEG_INF, NEG_INF);
    }
    if (bad_len) ll = NEG_INF;
    s_ll = ll;
    bool ok = ll != NEG_INF;                      // zero_infinity=True
    nll_out[b] = ok ? -ll : 0.f;
    if (ok) atomicAdd(loss, -ll * coef);
  }
  __syncthreads();
  const float ll = s_ll;
  if (!dlogits) return;
  if (ll == NEG_INF) {                            // infeasible alignment: zero the whole gradient slab
    for (int64_t i = s; i < (int64_t)T * V; i += 256) dlogits[(int64_t)b * T * V + i] = 0.f;
    return;
  }
  // ---------------- beta + occupancy
  float* dl = dlogits + (int64_t)b * T * V;
  float be = NEG_INF;
  {
    int t = Tb - 1;
    if (live && (s == S - 1 || s == S - 2)) be = lpb[(int64_t)t * V + ext];
    __syncthreads();
    sh[t & 1][s] = be;
    if (live && be != NEG_INF) {
      float g = expf(aw[(int64_t)t * Smax + s] + be - lpb[(int64_t)t * V + ext] - ll);
      atomicAdd(dl + (int64_t)t * V + ext, -coef * g);
    }
  }
  for (int t = Tb - 2; t >= 0; --t) {
    __syncthreads();
    const float* nxt = sh[(t + 1) & 1];
    float b0 = nxt[s];
    float b1 = (s + 1 < S) ? nxt[s + 1] : NEG_INF;
    float b2 = skip_f ? nxt[s + 2] : NEG_INF;
    float lpe = live ? lpb[(int64_t)t * V + ext] : 0.f;
    be = live ? lse3(b0, b1, b2) + lpe : NEG_INF;
    sh[t & 1][s] = be;
    if (live && be != NEG_INF) {
      float al = aw[(int64_t)t * Smax + s];
      if (al != NEG_INF) atomicAdd(dl + (int64_t)t * V + ext, -coef * expf(al + be - lpe - ll));
    }
  }
}

extern "C" int32_t otr_ctc_loss(const float* log_probs, const int64_t* targets, int64_t ldt, const int32_t* in_len,
                                const int32_t* tgt_len, int32_t B, int32_t T, int32_t V, int32_t max_tgt, int32_t blank,
                                float* alpha_ws, float* nll, float* loss, float* dlogits, void* stream) {
  OTR_REQUIRE(log_probs && targets && in_len && tgt_len && alpha_ws && nll && loss, "ctc_loss: null pointer");
  OTR_REQUIRE(B > 0 && T > 0 && V > 1, "ctc_loss: bad shape B=%d T=%d V=%d", B, T, V);
  OTR_REQUIRE(max_tgt >= 0 && 2 * max_tgt + 1 <= 256, "ctc_loss: target length %d > 127 not supported", max_tgt);
  OTR_REQUIRE(blank >= 0 && blank < V, "ctc_loss: blank out of range");
  hipStream_t s = (hipStream_t)stream;
  if (dlogits) {
    int64_t total = (int64_t)B * T * V;
    unsigned g = (unsigned)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
    hipLaunchKernelGGL(ctc_dense_kernel, dim3(g), dim3(256), 0, s, log_probs, dlogits, in_len, tgt_len, B, T, V, loss);
  } else {
    otr_zero_f32(loss, 1, s);
  }
  hipLaunchKernelGGL(ctc_alpha_beta_kernel, dim3(B), dim3(256), 0, s, log_probs, targets, ldt, in_len, tgt_len, B, T, V,
                     blank, alpha_ws, 2 * max_tgt + 1, nll, loss, dlogits);
  return otr_check_launch("ctc_loss");
}
