// Batch beam search scoring step (recognize/speech2text.py:95-192), device resident:
//  * beam_topk:  per hypothesis row: log_softmax(decoder logits) [+ lm_weight * log_softmax(LM logits)]
//                fused with top-k(beam) over the vocabulary -- the [rows, V] log-prob tensor is never
//                written (speech2text.py:100-112).
//  * beam_prune: per utterance: mask finished beams (one live branch with score 0 emitting EOS:
//                speech2text.py:156-192), add to the running scores, top-k(beam) over beam^2
//                candidates, gather the surviving prefixes and append the new token (:118-146).
// Ties are broken towards the lower candidate index.
#include "common.h"

#define NEG_INF (-__builtin_huge_valf())
constexpr int MAXK = 16;
extern int g_otr_beam_reg;     // api.hip (otr_debug_set(25, v)): 1 = the register-resident top-k kernel where the vocabulary fits

__device__ __forceinline__ void block_lse(const float* x, int V, float* sh, float& mx, float& lse) {
  float m = NEG_INF;
  for (int v = threadIdx.x; v < V; v += blockDim.x) m = fmaxf(m, x[v]);
  m = wave_max(m);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
  __syncthreads();
  m = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
  float s = 0.f;
  for (int v = threadIdx.x; v < V; v += blockDim.x) s += expf(x[v] - m);
  s = wave_sum(s);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  s = sh[0] + sh[1] + sh[2] + sh[3];
  mx = m;
  lse = m + logf(s);
}

__global__ __launch_bounds__(256) void beam_topk_kernel(const float* logits, int64_t ld, const float* lm_logits,
                                                       int64_t ld_lm, float lm_weight, int V, int k, float* out_score,
                                                       int64_t* out_idx) {
  __shared__ float sh[4];
  __shared__ float cand_s[256];
  __shared__ int cand_i[256];
  const int64_t row = blockIdx.x;
  const float* x = logits + row * ld;
  const float* y = lm_logits ? lm_logits + row * ld_lm : nullptr;
  float mx, lse, lmx, llse = 0.f;
  block_lse(x, V, sh, mx, lse);
  if (y) block_lse(y, V, sh, lmx, llse);
  // per-thread sorted top-k over its strided slice (descending, ties -> lower index first)
  float ts[MAXK];
  int ti[MAXK];
#pragma unroll
  for (int j = 0; j < MAXK; ++j) { ts[j] = NEG_INF; ti[j] = 0x7fffffff; }
  for (int v = threadIdx.x; v < V; v += blockDim.x) {
    float s = x[v] - lse;
    if (y) s += lm_weight * (y[v] - llse);
    if (s > ts[k - 1] || (s == ts[k - 1] && v < ti[k - 1])) {
      ts[k - 1] = s; ti[k - 1] = v;
#pragma unroll
      for (int j = MAXK - 1; j > 0; --j) {
        if (j < k && (ts[j] > ts[j - 1] || (ts[j] == ts[j - 1] && ti[j] < ti[j - 1]))) {
          float a = ts[j]; ts[j] = ts[j - 1]; ts[j - 1] = a;
          int b = ti[j]; ti[j] = ti[j - 1]; ti[j - 1] = b;
        }
      }
    }
  }
  // k rounds of block-wide argmax over the threads' current heads
  int head = 0;
  for (int r = 0; r < k; ++r) {
    float hs = NEG_INF;
    int hi = 0x7fffffff;
#pragma unroll
    for (int j = 0; j < MAXK; ++j)
      if (j == head) { hs = ts[j]; hi = ti[j]; }
    __syncthreads();
    cand_s[threadIdx.x] = hs;
    cand_i[threadIdx.x] = hi;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
      if ((int)threadIdx.x < off) {
        float a = cand_s[threadIdx.x], b = cand_s[threadIdx.x + off];
        int ia = cand_i[threadIdx.x], ib = cand_i[threadIdx.x + off];
        if (b > a || (b == a && ib < ia)) { cand_s[threadIdx.x] = b; cand_i[threadIdx.x] = ib; }
      }
      __syncthreads();
    }
    float ws = cand_s[0];
    int wi = cand_i[0];
    if (threadIdx.x == 0) { out_score[row * k + r] = ws; out_idx[row * k + r] = wi; }
    if (hi == wi && hs == ws && head < k) ++head;   // the owner pops its head
  }
}

// The same for V <= 256 * BT_NV (the shipped vocabularies), r05: the row (and the LM's row) is read ONCE into registers -- the kernel
// above reads each five times with 4-byte loads -- and a round of the block-wide arg-max is a 6-step wave butterfly + one exchange
// of the four waves' winners (2 barriers) instead of an 8-level shared-memory tree (18 barriers): 53 us per decode step before; the number now is in DESIGN.md 5.7 at
// 80 rows x 4234 (profiles/r05_decode_kernels.txt).  Same scores, same tie rule (lower index first): same selection.
constexpr int BT_NV = 20;
constexpr int BT_CAP = 16 * BT_NV;       // the counting selection's candidates: only the <= k threads whose maximum is at least tau hold any, BT_NV each
__device__ __forceinline__ bool bt_better(float s, int i, float t, int j) { return s > t || (s == t && i < j); }
__global__ __launch_bounds__(256) void beam_topk_reg_kernel(const float* logits, int64_t ld, const float* lm_logits, int64_t ld_lm, float lm_weight,
                                                           int V, int k, float* out_score, int64_t* out_idx, int g_rank) {
  __shared__ float shf[8];
  __shared__ float ws_s[4];
  __shared__ int ws_i[4];
  __shared__ __attribute__((aligned(16))) float tm_s[256];
  __shared__ __attribute__((aligned(16))) int tm_i[256];
  __shared__ float cand_s[BT_CAP];
  __shared__ int cand_i[BT_CAP];
  __shared__ float thr_s;
  __shared__ int thr_i, ncand;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int64_t row = blockIdx.x;
  const float* x = logits + row * ld;
  const float* y = lm_logits ? lm_logits + row * ld_lm : nullptr;
  float xs[BT_NV], ys[BT_NV];
  float mx = NEG_INF, my = NEG_INF;
#pragma unroll
  for (int j = 0; j < BT_NV; ++j) {
    const int v = tid + 256 * j;
    xs[j] = v < V ? x[v] : NEG_INF;
    ys[j] = (y && v < V) ? y[v] : NEG_INF;
    mx = fmaxf(mx, xs[j]);
    my = fmaxf(my, ys[j]);
  }
  mx = wave_max(mx); my = wave_max(my);
  if (lane == 0) { shf[wid] = mx; shf[4 + wid] = my; }
  __syncthreads();
  mx = fmaxf(fmaxf(shf[0], shf[1]), fmaxf(shf[2], shf[3]));
  my = fmaxf(fmaxf(shf[4], shf[5]), fmaxf(shf[6], shf[7]));
  float sx = 0.f, sy = 0.f;
#pragma unroll
  for (int j = 0; j < BT_NV; ++j) {
    const int v = tid + 256 * j;
    if (v < V) { sx += expf(xs[j] - mx); if (y) sy += expf(ys[j] - my); }
  }
  sx = wave_sum(sx); sy = wave_sum(sy);
  __syncthreads();
  if (lane == 0) { shf[wid] = sx; shf[4 + wid] = sy; }
  __syncthreads();
  const float lse = mx + logf(shf[0] + shf[1] + shf[2] + shf[3]);
  const float llse = y ? my + logf(shf[4] + shf[5] + shf[6] + shf[7]) : 0.f;
  // k rounds of a block-wide arg-max over the live elements (ties -> lower index first).  Each thread keeps the best of ITS live
  // elements; only the thread that owned a round's winner retires it and rescans its 20 registers.  (The first r05 form kept a
  // sorted k-list per thread: 19k instructions once unrolled, larger than the instruction cache -- 40 us; this is ~1k.)
  float sc[BT_NV];
  uint32_t alive = 0;
#pragma unroll
  for (int j = 0; j < BT_NV; ++j) {
    const int v = tid + 256 * j;
    sc[j] = xs[j] - lse;
    if (y) sc[j] += lm_weight * (ys[j] - llse);
    // a NaN never compares (bt_better is false both ways): a row of NaNs left the winner index at its 0x7fffffff sentinel, the shift
    // below undefined and a sentinel as a token id.  NaN ranks as -inf: such a row yields its lowest indices, like the
    // shared-memory kernel
    if (!(sc[j] == sc[j])) sc[j] = NEG_INF;
    if (v < V) alive |= 1u << j;
  }
  float hs = NEG_INF;
  int hi = 0x7fffffff;
#pragma unroll
  for (int j = 0; j < BT_NV; ++j)
    if ((alive >> j & 1u) && bt_better(sc[j], tid + 256 * j, hs, hi)) { hs = sc[j]; hi = tid + 256 * j; }
  // ---- r06: selection by COUNTING instead of k serial arg-max rounds (2 barriers + a butterfly each: 22 us at k = 10).
  //  1. tau = the k-th best of the 256 thread maxima (each thread ranks its own maximum against all: keys (score, index) are
  //     distinct, so exactly one thread finds rank k - 1).  Every element of the true top-k is at least tau: k distinct elements --
  //     the k best thread maxima -- are at least tau, so anything below tau has k elements above it.
  //  2. the candidates = all elements at least tau (k of them when every thread holds at most one, a few more otherwise) are
  //     compacted into LDS; a candidate's rank among the candidates is its rank in the row, and ranks 0 .. k-1 write the output.
  // Same keys, same tie rule (lower index first): the same selection as the rounds.  At most k threads have a maximum >= tau and only
  // they hold candidates: nc <= k * BT_NV <= BT_CAP always; the rounds below stay as a guarded fall-back (and as otr_debug_set(25, 2)).
  if (g_rank) {
    tm_s[tid] = hs; tm_i[tid] = hi;
    if (tid == 0) { ncand = 0; thr_i = 0x7fffffff; thr_s = NEG_INF; }
    __syncthreads();
    int rank = 0;
#pragma unroll 8
    for (int j = 0; j < 256; j += 4) {
      const float4 a = *reinterpret_cast<const float4*>(tm_s + j);
      const int4 b = *reinterpret_cast<const int4*>(tm_i + j);
      rank += (int)bt_better(a.x, b.x, hs, hi) + (int)bt_better(a.y, b.y, hs, hi) + (int)bt_better(a.z, b.z, hs, hi) + (int)bt_better(a.w, b.w, hs, hi);
    }
    if (rank == k - 1 && hi != 0x7fffffff) { thr_s = hs; thr_i = hi; }
    __syncthreads();
    const float ts_ = thr_s;
    const int ti_ = thr_i;
#pragma unroll
    for (int j = 0; j < BT_NV; ++j) {
      const int v = tid + 256 * j;
      if ((alive >> j & 1u) && !bt_better(ts_, ti_, sc[j], v)) {           // at least tau
        const int pos = atomicAdd(&ncand, 1);
        if (pos < BT_CAP) { cand_s[pos] = sc[j]; cand_i[pos] = v; }
      }
    }
    __syncthreads();
    const int nc = ncand;
    if (nc >= k && nc <= BT_CAP && ti_ != 0x7fffffff) {                    // (uniform over the workgroup)
      for (int c = tid; c < nc; c += 256) {
        const float ms = cand_s[c];
        const int mi = cand_i[c];
        int r = 0;
        for (int q = 0; q < nc; ++q) r += (int)bt_better(cand_s[q], cand_i[q], ms, mi);
        if (r < k) { out_score[row * k + r] = ms; out_idx[row * k + r] = mi; }
      }
      return;
    }
    __syncthreads();                                                        // fall back: the rounds below
  }
  for (int r = 0; r < k; ++r) {
    float bs = hs;
    int bi = hi;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const float os = __shfl_xor(bs, off);
      const int oi = __shfl_xor(bi, off);
      if (bt_better(os, oi, bs, bi)) { bs = os; bi = oi; }
    }
    __syncthreads();                                  // the previous round's readers are done with ws_*
    if (lane == 0) { ws_s[wid] = bs; ws_i[wid] = bi; }
    __syncthreads();
    float wsc = ws_s[0];
    int wi = ws_i[0];
#pragma unroll
    for (int w = 1; w < 4; ++w)
      if (bt_better(ws_s[w], ws_i[w], wsc, wi)) { wsc = ws_s[w]; wi = ws_i[w]; }
    if (tid == 0) { out_score[row * k + r] = wsc; out_idx[row * k + r] = wi; }
    if (hi == wi && wi < V) {                         // the owner retires the winner and rescans (wi < V: never shift by a sentinel)
      alive &= ~(1u << ((wi - tid) >> 8));
      hs = NEG_INF; hi = 0x7fffffff;
#pragma unroll
      for (int j = 0; j < BT_NV; ++j)
        if ((alive >> j & 1u) && bt_better(sc[j], tid + 256 * j, hs, hi)) { hs = sc[j]; hi = tid + 256 * j; }
    }
  }
}

extern "C" int32_t otr_beam_topk(const float* logits, int64_t ld, const float* lm_logits, int64_t ld_lm, float lm_weight,
                                 int64_t rows, int32_t V, int32_t k, float* out_score, int64_t* out_idx, void* stream) {
  OTR_REQUIRE(logits && out_score && out_idx, "beam_topk: null pointer");
  OTR_REQUIRE(k >= 1 && k <= MAXK && k <= V, "beam_topk: k=%d must be in [1, %d] and <= V", k, MAXK);
  OTR_REQUIRE(rows >= 0 && V > 0 && ld >= V, "beam_topk: bad shape");
  if (rows == 0) return 0;
  if (V <= 256 * BT_NV && g_otr_beam_reg)
    hipLaunchKernelGGL(beam_topk_reg_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, logits, ld, lm_logits, ld_lm, lm_weight, V,
                       k, out_score, out_idx, g_otr_beam_reg == 1 ? 1 : 0);     // otr_debug_set(25, 2): the register kernel with the serial rounds
  else
    hipLaunchKernelGGL(beam_topk_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, logits, ld, lm_logits,
                       ld_lm, lm_weight, V, k, out_score, out_idx);
  return otr_check_launch("beam_topk");
}

// one block per utterance; beam*beam <= 256 candidates
__global__ __launch_bounds__(256) void beam_prune_kernel(const float* k_score, const int64_t* k_idx, const float* scores_in,
                                                        const uint8_t* flag_in, const int64_t* preds_in, int64_t ldp,
                                                        int beam, int t, int eos, float* scores_out, uint8_t* flag_out,
                                                        int64_t* preds_out, int32_t* n_finished, const int32_t* pos_in,
                                                        int32_t* pos_out, const int32_t* anc_in, int32_t* anc_out,
                                                        int ld_anc, int32_t* arrive) {
  __shared__ float c0[256];
  __shared__ int win[MAXK];
  const int b = blockIdx.x, tid = threadIdx.x, nc = beam * beam;
  if (pos_in) t = *pos_in + 1;                      // cached decoding: prefix length lives on the device
  float s = NEG_INF;
  if (tid < nc) {
    int hyp = b * beam + tid / beam, br = tid % beam;
    bool fin = flag_in[hyp] != 0;
    float ks = k_score[(int64_t)hyp * beam + br];
    if (fin) ks = (br == 0) ? 0.f : NEG_INF;        // mask_finished_scores
    s = scores_in[hyp] + ks;
  }
  // r06: selection by COUNTING (it was `beam` rounds of a block-wide arg-max, two barriers each: 12.8 us per decode step): every
  // candidate's rank = the number of candidates that beat it (ties -> lower candidate index; keys are distinct), ranks 0 .. beam-1 are the
  // winners in order.  A NaN score ranks as -inf (it never compares).
  if (!(s == s)) s = NEG_INF;
  c0[tid] = s;
  __syncthreads();
  if (tid < nc) {
    int rank = 0;
    for (int q = 0; q < nc; ++q) {
      const float o = c0[q];
      rank += (o > s || (o == s && q < tid)) ? 1 : 0;
    }
    if (rank < beam) {
      win[rank] = tid;
      scores_out[b * beam + rank] = s;
    }
  }
  __syncthreads();
  // gather prefixes (+ re-parent the KV-cache ancestor table, decode.hip): every (output hypothesis, position) pair at once -- one
  // level of dependent loads; it was a serial loop over the beam, 10 x [winner -> flag -> token -> copy]: 20.6 us per decode step
  const int64_t ob = (int64_t)b * beam;
  for (int e = tid; e < beam * t; e += 256) {
    const int r = e / t, i = e - r * t;
    const int64_t src = ob + win[r] / beam;
    preds_out[(ob + r) * ldp + i] = preds_in[src * ldp + i];
    if (anc_in && i < t - 1) anc_out[(ob + r) * ld_anc + i] = anc_in[src * ld_anc + i];
  }
  if (tid < 64) {                                   // wave 0: lane r < beam appends output hypothesis r's token
    bool f = false;
    if (tid < beam) {
      const int w = win[tid];
      const int64_t src = ob + w / beam;
      const bool fin = flag_in[src] != 0;
      const int64_t tok = fin ? (int64_t)eos : k_idx[src * beam + (w % beam)];   // mask_finished_preds
      preds_out[(ob + tid) * ldp + t] = tok;
      if (anc_in) anc_out[(ob + tid) * ld_anc + t - 1] = (int32_t)src;
      f = tok == eos;
      flag_out[ob + tid] = f ? 1 : 0;
    }
    const int cnt = __popcll(__ballot(f));
    if (tid == 0) {
      if (!arrive) {
        if (cnt) atomicAdd(n_finished, cnt);        // n_finished zeroed by the launch before this one
      } else {
        // cached loop: no zeroing launch.  *arrive packs (blocks arrived << 16 | finished so far); the last block to arrive publishes
        // the total and resets the word for the next step (which is a later launch on the same stream / graph branch)
        const int old = atomicAdd(arrive, (1 << 16) | cnt);
        if ((old >> 16) == (int)gridDim.x - 1) {
          *n_finished = (old & 0xffff) + cnt;
          *arrive = 0;
        }
      }
    }
  }
  if (pos_out && b == 0 && tid == 0) *pos_out = t;
}

extern "C" int32_t otr_beam_prune(const float* k_score, const int64_t* k_idx, const float* scores_in,
                                  const uint8_t* flag_in, const int64_t* preds_in, int64_t ldp, int32_t batch,
                                  int32_t beam, int32_t t, int32_t eos, float* scores_out, uint8_t* flag_out,
                                  int64_t* preds_out, int32_t* n_finished, void* stream) {
  OTR_REQUIRE(k_score && k_idx && scores_in && flag_in && preds_in && scores_out && flag_out && preds_out && n_finished,
              "beam_prune: null pointer");
  OTR_REQUIRE(beam >= 1 && beam <= MAXK && beam * beam <= 256, "beam_prune: beam=%d must be in [1, 16]", beam);
  OTR_REQUIRE(batch > 0 && t >= 1 && t < ldp, "beam_prune: bad shape batch=%d t=%d ldp=%lld", batch, t, (long long)ldp);
  hipStream_t s = (hipStream_t)stream;
  otr_zero_f32(reinterpret_cast<float*>(n_finished), 1, s);   // int32 0 == float 0 bit pattern
  hipLaunchKernelGGL(beam_prune_kernel, dim3(batch), dim3(256), 0, s, k_score, k_idx, scores_in, flag_in, preds_in, ldp,
                     beam, t, eos, scores_out, flag_out, preds_out, n_finished, (const int32_t*)nullptr, (int32_t*)nullptr,
                     (const int32_t*)nullptr, (int32_t*)nullptr, 0, (int32_t*)nullptr);
  return otr_check_launch("beam_prune");
}

extern "C" int32_t otr_beam_prune_cached(const float* k_score, const int64_t* k_idx, const float* scores_in,
                                         const uint8_t* flag_in, const int64_t* preds_in, int64_t ldp, int32_t batch,
                                         int32_t beam, int32_t eos, const int32_t* pos_in, int32_t* pos_out,
                                         const int32_t* anc_in, int32_t* anc_out, int32_t ld_anc, float* scores_out,
                                         uint8_t* flag_out, int64_t* preds_out, int32_t* n_finished, void* stream) {
  OTR_REQUIRE(k_score && k_idx && scores_in && flag_in && preds_in && scores_out && flag_out && preds_out && n_finished,
              "beam_prune_cached: null pointer");
  OTR_REQUIRE(pos_in && pos_out && anc_in && anc_out, "beam_prune_cached: null position / ancestor pointer");
  OTR_REQUIRE(pos_in != pos_out && anc_in != anc_out && preds_in != preds_out,
              "beam_prune_cached: in/out buffers must be distinct (ping-pong)");
  OTR_REQUIRE(beam >= 1 && beam <= MAXK && beam * beam <= 256, "beam_prune_cached: beam=%d must be in [1, 16]", beam);
  OTR_REQUIRE(batch > 0 && ld_anc > 0 && ld_anc < ldp, "beam_prune_cached: bad shape batch=%d ld_anc=%d ldp=%lld", batch,
              ld_anc, (long long)ldp);
  OTR_REQUIRE(batch < 32768 && (int64_t)batch * beam < 65536, "beam_prune_cached: batch=%d x beam=%d too large for the packed arrival word",
              batch, beam);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(beam_prune_kernel, dim3(batch), dim3(256), 0, s, k_score, k_idx, scores_in, flag_in, preds_in, ldp,
                     beam, 0, eos, scores_out, flag_out, preds_out, n_finished, pos_in, pos_out, anc_in, anc_out, ld_anc,
                     n_finished + 1);
  return otr_check_launch("beam_prune_cached");
}
