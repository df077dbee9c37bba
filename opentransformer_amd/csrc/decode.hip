// Incremental (KV-cached) decoding step for the batch beam search.
//
// The reference threads a `cache` argument through decoder.inference / attention.inference but never fills it
// (decoder/transformer.py:185-208, module/attention.py:86-104): every step re-runs the decoder over the whole
// prefix for every beam.  Here each step feeds ONE token per hypothesis:
//   * decode_embed:            y[r] = E[preds[r, *pos]] * sqrt(d) + PE[*pos]        (decoder/transformer.py:163-169)
//   * decode_self_attention:   the new position's q against the cached k/v of its ANCESTORS.  Hypotheses form a
//                              tree (beam pruning re-parents rows), so instead of gathering the caches after every
//                              prune, an ancestor table anc[r, j] = cache row holding position j of hypothesis r is
//                              carried along (updated by beam_prune_cached); the caches themselves are write-once.
//   * beam_prune_cached:       otr_beam_prune + ancestor-table update, prefix length read from device memory.
// Every step-dependent scalar (position, prefix length) lives in device memory, so ONE captured hipGraph replays
// for all steps.  These kernels are HBM/latency bound (one query row per hypothesis): no MFMA.
#include "common.h"

#define NEG_INF (-__builtin_huge_valf())
extern int g_otr_decode_attn64;    // api.hip (otr_debug_set(24, v)): the vector-load form of the cached self-attention step

__global__ void decode_embed_kernel(const int64_t* preds, int64_t ldp, const int32_t* pos, const float* E, float* y,
                                    bf16_t* y_lp, int d, int vocab, float scale) {
  const int64_t r = blockIdx.x;
  const int p = *pos;
  const int64_t t = preds[r * ldp + p];
  const float nl = -logf(10000.f) / (float)d;
  const bool ok = t >= 0 && t < vocab;
  for (int col = threadIdx.x; col < d; col += blockDim.x) {
    float e = ok ? E[t * d + col] : 0.f;
    float v = e * scale + pe_value(p, col, nl);
    y[r * d + col] = v;
    if (y_lp) y_lp[r * d + col] = f2bf(v);
  }
}

extern "C" int32_t otr_decode_embed(const int64_t* preds, int64_t ldp, const int32_t* pos, const float* E, float* y,
                                    void* y_bf16, int64_t rows, int32_t d, int32_t vocab, float scale, void* stream) {
  OTR_REQUIRE(preds && pos && E && y, "decode_embed: null pointer");
  OTR_REQUIRE(rows >= 0 && d > 0 && vocab > 0 && ldp > 0, "decode_embed: bad shape");
  if (rows == 0) return 0;
  hipLaunchKernelGGL(decode_embed_kernel, dim3((unsigned)rows), dim3(d >= 256 ? 256 : 64), 0, (hipStream_t)stream, preds,
                     ldp, pos, E, y, (bf16_t*)y_bf16, d, vocab, scale);
  return otr_check_launch("decode_embed");
}

// ---- recurrent language model in the fused search (model/lm.py:72-79 through recognize/base.py:26-37)
// y[r] = E[preds[r, *pos]]: the plain nn.Embedding row of each hypothesis' LAST token (no scale, no positional term), position read
// from device memory like decode_embed (pos == NULL: column 0 of preds).
__global__ void decode_lookup_kernel(const int64_t* preds, int64_t ldp, const int32_t* pos, const float* E, float* y, bf16_t* y_lp, int d,
                                     int vocab) {
  const int64_t r = blockIdx.x;
  const int64_t t = preds[r * ldp + (pos ? *pos : 0)];
  const bool ok = t >= 0 && t < vocab;
  for (int col = threadIdx.x; col < d; col += blockDim.x) {
    const float v = ok ? E[t * d + col] : 0.f;
    y[r * d + col] = v;
    if (y_lp) y_lp[r * d + col] = f2bf(v);
  }
}
extern "C" int32_t otr_decode_lookup(const int64_t* preds, int64_t ldp, const int32_t* pos, const float* E, float* y, void* y_bf16,
                                     int64_t rows, int32_t d, int32_t vocab, void* stream) {
  OTR_REQUIRE(preds && E && y, "decode_lookup: null pointer");
  OTR_REQUIRE(rows >= 0 && d > 0 && vocab > 0 && ldp > 0, "decode_lookup: bad shape");
  if (rows == 0) return 0;
  hipLaunchKernelGGL(decode_lookup_kernel, dim3((unsigned)rows), dim3(d >= 256 ? 256 : 64), 0, (hipStream_t)stream, preds, ldp, pos, E, y,
                     (bf16_t*)y_bf16, d, vocab);
  return otr_check_launch("decode_lookup");
}

// One LSTM cell update (torch.nn.LSTM's equations, gate order i | f | g | o): gates = ga + gb + bias_b with ga = W_ih x + b_ih the
// caller's first GEMM, gb = W_hh h + b_hh its second (NULL when the previous state is zero: then bias_b = b_hh is all that is left
// of that term), c_prev NULL = zeros.
//   c = sigmoid(f) c_prev + sigmoid(i) tanh(g);  h = sigmoid(o) tanh(c)        -> h (f32 + 16-bit twin for the next GEMM), c (f32)
__global__ void lstm_cell_kernel(const float* ga, const float* gb, const float* bias_b, const float* c_prev, float* h, bf16_t* h_lp, float* c,
                                 int64_t rows, int H) {
  const int64_t total = rows * H;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / H;
    const int j = (int)(i - r * H);
    float g4[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float v = ga[r * 4 * H + q * H + j];
      if (gb) v += gb[r * 4 * H + q * H + j];
      if (bias_b) v += bias_b[q * H + j];
      g4[q] = v;
    }
    const float si = 1.f / (1.f + expf(-g4[0])), sf = 1.f / (1.f + expf(-g4[1])), so = 1.f / (1.f + expf(-g4[3]));
    const float cn = sf * (c_prev ? c_prev[i] : 0.f) + si * tanhf(g4[2]);
    const float hn = so * tanhf(cn);
    c[i] = cn;
    h[i] = hn;
    if (h_lp) h_lp[i] = f2bf(hn);
  }
}
extern "C" int32_t otr_lstm_cell(const float* gates_a, const float* gates_b, const float* bias_b, const float* c_prev, float* h, void* h_bf16,
                                 float* c, int64_t rows, int32_t hidden, void* stream) {
  OTR_REQUIRE(gates_a && h && c, "lstm_cell: null pointer");
  OTR_REQUIRE(rows >= 0 && hidden > 0, "lstm_cell: bad shape");
  if (rows == 0) return 0;
  const int64_t n = rows * hidden, g = (n + 255) / 256;
  hipLaunchKernelGGL(lstm_cell_kernel, dim3((unsigned)(g > 4096 ? 4096 : g)), dim3(256), 0, (hipStream_t)stream, gates_a, gates_b, bias_b, c_prev, h,
                     (bf16_t*)h_bf16, c, rows, hidden);
  return otr_check_launch("lstm_cell");
}

// One wave per (hypothesis row, head).  qkv: [R, 3d] of the new position (columns q|k|v, module/attention.py:73).
// Phase 0 stores the new k,v into cache[r, p]; phase 1: lane = key position (chunks of 64, online softmax),
// each lane dots its ancestor's cached key with q (q staged in LDS); phase 2: lane = head dimension, the
// probabilities are broadcast with shuffles and the value rows are read coalesced.
template <class T>
__global__ __launch_bounds__(64) void decode_self_attn_kernel(const T* __restrict__ qkv, T* __restrict__ kc,
                                                             T* __restrict__ vc, const int32_t* __restrict__ anc,
                                                             const int32_t* __restrict__ pos, T* __restrict__ out,
                                                             int H, int dk, int maxlen, float scale) {
  __shared__ float qs[128];
  const int r = blockIdx.x / H, h = blockIdx.x % H, lane = threadIdx.x;
  const int d = H * dk;
  const int p = *pos;                                  // keys 0..p
  const T* q = qkv + (int64_t)r * 3 * d + h * dk;
  const T* kn = q + d;
  const T* vn = q + 2 * d;
  for (int i = lane; i < dk; i += 64) {
    qs[i] = ElemIO<T>::ld(q + i);
    int64_t o = ((int64_t)r * maxlen + p) * d + h * dk + i;
    kc[o] = kn[i];
    vc[o] = vn[i];
  }
  __syncthreads();
  float m = NEG_INF, l = 0.f, acc0 = 0.f, acc1 = 0.f;   // acc: dims lane, lane+64
  for (int c0 = 0; c0 <= p; c0 += 64) {
    const int j = c0 + lane;
    float s = NEG_INF;
    int row = r;
    if (j <= p) {
      const T* kp;
      if (j == p) kp = kn;                              // not read back through the cache: no store->load hazard
      else {
        row = anc[(int64_t)r * maxlen + j];
        kp = kc + ((int64_t)row * maxlen + j) * d + h * dk;
      }
      float a = 0.f;
      for (int i = 0; i < dk; ++i) a += qs[i] * ElemIO<T>::ld(kp + i);
      s = a * scale;
    }
    const float mn = fmaxf(m, wave_max(s));
    const float pj = (j <= p) ? expf(s - mn) : 0.f;
    const float corr = expf(m - mn);                    // m = -inf on the first chunk -> 0
    l = l * corr + wave_sum(pj);
    acc0 *= corr;
    acc1 *= corr;
    m = mn;
    const int n = min(64, p + 1 - c0);
    for (int jj = 0; jj < n; ++jj) {
      const float pv = __shfl(pj, jj);
      const int rw = __shfl(row, jj);
      const int jabs = c0 + jj;
      const T* vp = (jabs == p) ? vn : vc + ((int64_t)rw * maxlen + jabs) * d + h * dk;
      if (lane < dk) acc0 += pv * ElemIO<T>::ld(vp + lane);
      if (lane + 64 < dk) acc1 += pv * ElemIO<T>::ld(vp + lane + 64);
    }
  }
  const float inv = 1.f / l;
  T* o = out + (int64_t)r * d + h * dk;
  if (lane < dk) ElemIO<T>::st(o + lane, acc0 * inv);
  if (lane + 64 < dk) ElemIO<T>::st(o + lane + 64, acc1 * inv);
}

// The same for 16-bit operands and head dim 64 (the shipped models), without the two serial loops: the kernel above dots a key with
// 64 two-byte loads per lane and then walks the positions one value row at a time, a dependent global load per position -- 14 us
// per launch at 30 cached positions, 25 us at 60, ten launches per decode step (rocprofv3, profiles/r05_decode_kernels.txt).  Here a
// lane fetches its key as eight 16-byte loads in flight, and the value rows of a 64-position chunk are read as (row, 16-byte piece)
// pairs, eight per lane, all in flight: lane -> piece lane & 7 of rows (lane >> 3) + 8 t; the eight row groups meet in a 3-step
// butterfly.  Probabilities and ancestor rows travel through LDS.  Same arithmetic per score; the context sums in a different order.
__global__ __launch_bounds__(64) void decode_self_attn64_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ kc, bf16_t* __restrict__ vc,
                                                               const int32_t* __restrict__ anc, const int32_t* __restrict__ pos,
                                                               bf16_t* __restrict__ out, int H, int maxlen, float scale) {
  constexpr int DK = 64;
  __shared__ float qs[DK];
  __shared__ float ps[64];
  __shared__ int rws[64];
  const int r = blockIdx.x / H, h = blockIdx.x % H, lane = threadIdx.x;
  const int d = H * DK;
  const int p = *pos;
  const bf16_t* q = qkv + (int64_t)r * 3 * d + h * DK;
  const bf16_t* kn = q + d;
  const bf16_t* vn = q + 2 * d;
  qs[lane] = bf2f(q[lane]);
  if (lane < 16) {                                      // the new position's key / value into the caches: 16 bytes per lane
    const int64_t o = ((int64_t)r * maxlen + p) * d + h * DK + 8 * (lane & 7);
    if (lane < 8) st_global_b128(kc + o, ld_global_b128(kn + 8 * lane));
    else st_global_b128(vc + o, ld_global_b128(vn + 8 * (lane & 7)));
  }
  __syncthreads();
  float m = NEG_INF, l = 0.f, acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
  const int ch = lane & 7, g = lane >> 3;
  for (int c0 = 0; c0 <= p; c0 += 64) {
    const int j = c0 + lane, jc = min(j, p);            // lanes past the last position fetch a valid row and are masked
    int row = r;
    const bf16_t* kp = kn;
    if (jc != p) {
      row = anc[(int64_t)r * maxlen + jc];
      kp = kc + ((int64_t)row * maxlen + jc) * d + h * DK;
    }
    uint4 kk[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) kk[i] = ld_global_b128(kp + 8 * i);
    float a = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const uint32_t w[4] = {kk[i].x, kk[i].y, kk[i].z, kk[i].w};
#pragma unroll
      for (int e = 0; e < 4; ++e) { a += qs[8 * i + 2 * e] * h2f_lo(w[e]); a += qs[8 * i + 2 * e + 1] * h2f_hi(w[e]); }
    }
    const float s = (j <= p) ? a * scale : NEG_INF;
    const float mn = fmaxf(m, wave_max(s));
    const float pj = (j <= p) ? expf(s - mn) : 0.f;
    const float corr = expf(m - mn);                    // m = -inf on the first chunk -> 0
    l = l * corr + wave_sum(pj);
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] *= corr;
    m = mn;
    __syncthreads();                                    // the previous chunk's readers are done with ps / rws
    ps[lane] = pj;
    rws[lane] = row;
    __syncthreads();
    const int n = min(64, p + 1 - c0);
    uint4 vv[8];
    float pw[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int jj = g + 8 * t, jv = min(jj, n - 1), jabs = c0 + jv;
      const bf16_t* vp = (jabs == p) ? vn : vc + ((int64_t)rws[jv] * maxlen + jabs) * d + h * DK;
      vv[t] = ld_global_b128(vp + 8 * ch);
      pw[t] = jj < n ? ps[jv] : 0.f;
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const uint32_t w[4] = {vv[t].x, vv[t].y, vv[t].z, vv[t].w};
#pragma unroll
      for (int e = 0; e < 4; ++e) { acc[2 * e] += pw[t] * h2f_lo(w[e]); acc[2 * e + 1] += pw[t] * h2f_hi(w[e]); }
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    acc[e] += __shfl_xor(acc[e], 8);
    acc[e] += __shfl_xor(acc[e], 16);
    acc[e] += __shfl_xor(acc[e], 32);
  }
  if (lane < 8) {
    const float inv = 1.f / l;
    st_global_b128(out + (int64_t)r * d + h * DK + 8 * lane,
                   make_uint4(pack2h(acc[0] * inv, acc[1] * inv), pack2h(acc[2] * inv, acc[3] * inv), pack2h(acc[4] * inv, acc[5] * inv),
                              pack2h(acc[6] * inv, acc[7] * inv)));
  }
}

extern "C" int32_t otr_decode_self_attention(const void* qkv, void* kcache, void* vcache, const int32_t* anc,
                                             const int32_t* pos, void* out, int32_t dtype, int64_t rows, int32_t H,
                                             int32_t dk, int32_t maxlen, float scale, void* stream) {
  OTR_REQUIRE(qkv && kcache && vcache && anc && pos && out, "decode_self_attention: null pointer");
  OTR_REQUIRE(dtype == OTR_F32 || dtype == OTR_H16, "decode_self_attention: dtype must be f32 or bf16");
  OTR_REQUIRE(H > 0 && dk > 0 && dk <= 128 && maxlen > 0 && rows >= 0, "decode_self_attention: bad shape (dk <= 128)");
  OTR_REQUIRE(rows * H < (1ll << 31), "decode_self_attention: too many rows");
  if (rows == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == OTR_F32)
    hipLaunchKernelGGL(decode_self_attn_kernel<float>, dim3((unsigned)(rows * H)), dim3(64), 0, s, (const float*)qkv,
                       (float*)kcache, (float*)vcache, anc, pos, (float*)out, H, dk, maxlen, scale);
  else if (dk == 64 && g_otr_decode_attn64 && (((uintptr_t)qkv | (uintptr_t)kcache | (uintptr_t)vcache | (uintptr_t)out) % 16) == 0)
    hipLaunchKernelGGL(decode_self_attn64_kernel, dim3((unsigned)(rows * H)), dim3(64), 0, s, (const bf16_t*)qkv, (bf16_t*)kcache, (bf16_t*)vcache,
                       anc, pos, (bf16_t*)out, H, maxlen, scale);
  else
    hipLaunchKernelGGL(decode_self_attn_kernel<bf16_t>, dim3((unsigned)(rows * H)), dim3(64), 0, s, (const bf16_t*)qkv,
                       (bf16_t*)kcache, (bf16_t*)vcache, anc, pos, (bf16_t*)out, H, dk, maxlen, scale);
  return otr_check_launch("decode_self_attention");
}
