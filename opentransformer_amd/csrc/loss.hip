// Row-reduction kernels over the vocabulary axis:
//  * LabelSmoothingLoss forward + gradient in one pass (module/loss.py:21-48)
//  * log_softmax (model/ctc.py:51,66; decoder/transformer.py:206)
// One 256-thread block per row; the row (V <= ~32k fp32) is read twice from L2/HBM.
#include "common.h"

__device__ __forceinline__ float block_reduce(float v, float* sh, bool is_max) {
  v = is_max ? wave_max(v) : wave_sum(v);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sh[wid] = v;
  __syncthreads();
  float r = sh[0];
  for (int i = 1; i < (int)(blockDim.x >> 6); ++i) r = is_max ? fmaxf(r, sh[i]) : r + sh[i];
  return r;
}

// number of non-pad rows: every row's block counts them itself (R targets, a few loads per thread, an exact integer-valued sum in a
// fixed order: the same value in every block) -- a counting launch of its own was 5 us in front of the rows kernel
// (beyond LS_COUNT_INLINE rows the R^2 / 256 loads stop being free: the counting launch is kept for those)
constexpr int64_t LS_COUNT_INLINE = 8192;
__device__ __forceinline__ float ls_count(const int64_t* target, int64_t R, int pad_idx, float* sh) {
  float c = 0.f;
  for (int64_t i = threadIdx.x; i < R; i += blockDim.x) c += (target[i] != pad_idx) ? 1.f : 0.f;
  return block_reduce(c, sh, false);
}
// the same over a [B, L] view with row stride ldt (elements): target[(i / L) * ldt + i % L]
__device__ __forceinline__ float ls_count_ld(const int64_t* target, int64_t ldt, int L, int64_t R, int pad_idx, float* sh) {
  float c = 0.f;
  for (int64_t i = threadIdx.x; i < R; i += blockDim.x) c += (target[(i / L) * ldt + (i % L)] != pad_idx) ? 1.f : 0.f;
  return block_reduce(c, sh, false);
}
__global__ void ls_count_kernel(const int64_t* target, int64_t R, int pad_idx, float* scratch) {      // scratch[0] = number of non-pad rows
  __shared__ float sh[4];
  const float c = ls_count(target, R, pad_idx, sh);
  if (threadIdx.x == 0) scratch[0] = c;
}

// deterministic final reduction: loss = sum_r row_loss[r] (fixed order, one block)
__global__ void ls_finalize_kernel(const float* row_loss, int64_t R, float* loss) {
  __shared__ float sh[4];
  float c = 0.f;
  for (int64_t i = threadIdx.x; i < R; i += blockDim.x) c += row_loss[i];
  c = block_reduce(c, sh, false);
  if (threadIdx.x == 0) *loss = c;
}

// KL(conf || softmax(logits)) per row, conf = 1-eps on the target, eps/(V-1) elsewhere.
//   row_loss = C - [ (1-eps) logp_t + eps/(V-1) (sum_v logp_v - logp_t) ],
//   C = (1-eps) log(1-eps) + eps log(eps/(V-1))        (0 log 0 := 0, as torch's kl_div)
//   d row_loss / d logit_v = softmax_v - conf_v
// Rows of logits / dlogits may be longer than V (ld_x, ld_dx >= V: the head of a row-padded product, ops.padded_rows): the
// columns of dlogits behind V are written as zeros, so that the buffer can be read at its full width.
__global__ __launch_bounds__(256) void ls_rows_kernel(const float* logits, int64_t ld_x, const int64_t* target, int64_t R, int V, float eps,
                                                     int pad_idx, const float* count, float* row_loss, float* dlogits, int64_t ld_dx) {
  __shared__ float sh[4];
  const int64_t row = blockIdx.x;
  const float* x = logits + row * ld_x;
  float* dx = dlogits ? dlogits + row * ld_dx : nullptr;
  const int64_t t = target[row];
  if (dx) for (int64_t v = V + threadIdx.x; v < ld_dx; v += blockDim.x) dx[v] = 0.f;
  if (t == pad_idx) {
    if (dx) for (int v = threadIdx.x; v < V; v += blockDim.x) dx[v] = 0.f;
    if (threadIdx.x == 0) row_loss[row] = 0.f;
    return;
  }
  float mx = -__builtin_huge_valf();
  for (int v = threadIdx.x; v < V; v += blockDim.x) mx = fmaxf(mx, x[v]);
  mx = block_reduce(mx, sh, true);
  float se = 0.f, sx = 0.f;
  for (int v = threadIdx.x; v < V; v += blockDim.x) { float xv = x[v]; se += expf(xv - mx); sx += xv; }
  se = block_reduce(se, sh, false);
  sx = block_reduce(sx, sh, false);
  const float lse = mx + logf(se);
  const float inv_cnt = 1.f / (count ? count[0] : ls_count(target, R, pad_idx, sh));
  const float off = eps / (float)(V - 1), on = 1.f - eps;
  if (dx) {
    for (int v = threadIdx.x; v < V; v += blockDim.x) {
      float sm = expf(x[v] - lse);
      dx[v] = (sm - (v == t ? on : off)) * inv_cnt;
    }
  }
  if (threadIdx.x == 0) {
    float logp_t = x[t] - lse;
    float sum_logp = sx - (float)V * lse;
    float C = (on > 0.f ? on * logf(on) : 0.f) + (eps > 0.f ? eps * logf(off) : 0.f);
    float rl = C - (on * logp_t + off * (sum_logp - logp_t));
    row_loss[row] = rl * inv_cnt;
  }
}

extern "C" int32_t otr_label_smoothing_loss_ld(const float* logits, int64_t ld_logits, const int64_t* target, int64_t R, int32_t V,
                                               float smoothing, int32_t pad_idx, float* loss, float* dlogits, int64_t ld_dlogits,
                                               float* scratch, void* stream) {
  OTR_REQUIRE(logits && target && loss && scratch, "label_smoothing_loss: null pointer");
  OTR_REQUIRE(R > 0 && V > 1, "label_smoothing_loss: bad shape R=%lld V=%d", (long long)R, V);
  OTR_REQUIRE(ld_logits >= V && (!dlogits || ld_dlogits >= V), "label_smoothing_loss: leading dimensions %lld / %lld shorter than V=%d",
              (long long)ld_logits, (long long)ld_dlogits, V);
  OTR_REQUIRE(smoothing >= 0.f && smoothing < 1.f, "label_smoothing_loss: smoothing out of [0,1)");
  hipStream_t s = (hipStream_t)stream;
  const float* count = nullptr;
  if (R > LS_COUNT_INLINE) {
    hipLaunchKernelGGL(ls_count_kernel, dim3(1), dim3(256), 0, s, target, R, pad_idx, scratch);
    count = scratch;
  }
  hipLaunchKernelGGL(ls_rows_kernel, dim3((unsigned)R), dim3(256), 0, s, logits, ld_logits, target, R, V, smoothing, pad_idx, count, scratch + 2,
                     dlogits, ld_dlogits);
  hipLaunchKernelGGL(ls_finalize_kernel, dim3(1), dim3(256), 0, s, scratch + 2, R, loss);
  return otr_check_launch("label_smoothing_loss");
}

// three sums in one exchange (the waves' partials in a fixed order, like block_reduce)
__device__ __forceinline__ void block_reduce3(float& a, float& b, float& c, float* sh3) {
  a = wave_sum(a); b = wave_sum(b); c = wave_sum(c);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) { sh3[wid] = a; sh3[4 + wid] = b; sh3[8 + wid] = c; }
  __syncthreads();
  a = sh3[0] + sh3[1] + sh3[2] + sh3[3];
  b = sh3[4] + sh3[5] + sh3[6] + sh3[7];
  c = sh3[8] + sh3[9] + sh3[10] + sh3[11];
}

// ---- the whole loss in ONE launch (round 5).  What the three-kernel form above spends around its 8 MB of logits at the AISHELL
// shape (480 rows x 4234): three passes over every row with 4-byte loads (17 us), a finalize launch (5 us), and -- in the training
// step -- two launches that only multiply by a scalar (the loss scale seeding the backward pass, 5 us; the gradient of the loss
// times it, 6 us).  Here a row is read ONCE into registers (16-byte loads: NV4 float4 per thread), its gradient leaves already
// multiplied by the device scalar *gscale (the factor the backward pass will be seeded with), and the block that finishes LAST adds
// up the per-row losses in a fixed order (the order of ls_finalize_kernel: the result is bit-identical to the three-kernel form's
// reduction of the same row losses).  Inter-block hand-off: row losses travel as agent-scope relaxed atomic stores / loads
// (global_store / load sc1: write-through, served past the L1), the arrival count is one agent-scope atomic add behind an
// explicit vmcnt(0) (cdna_hip_programming.md G16, "sc1 stores and loads both sides"); the last block puts the ticket back to 0.
// Targets are addressed as target[(row / L) * ldt + row % L]: the [B, L] view truth[:, 1:] of a [B, L + 1] matrix needs no copy.
// H16: the gradient leaves in the library's 16-bit type (8 bytes per quad) -- the operand type of the three GEMMs of the output layer
// behind it (decoder/transformer.py:153), which then run on the 16-bit paths (its weight gradient in the grouped 256-wide launch).
template <int NV4, bool H16>
__global__ __launch_bounds__(256) void ls_rows_fused_kernel(const float* logits, int64_t ld_x, const int64_t* target, int64_t ldt, int L,
                                                           int64_t R, int V, float eps, int pad_idx, const float* gscale, float* row_loss,
                                                           void* dlogits, int64_t ld_dx, float* loss, unsigned int* ticket) {
  __shared__ float sh[4];
  __shared__ float sh3[12];
  __shared__ int last_flag;
  const int tid = threadIdx.x;
  const int64_t row = blockIdx.x;
  const float* x = logits + row * ld_x;
  float* dx = dlogits ? reinterpret_cast<float*>(dlogits) + row * ld_dx : nullptr;                     // !H16
  bf16_t* dxh = dlogits ? reinterpret_cast<bf16_t*>(dlogits) + row * ld_dx : nullptr;                  //  H16
  auto put4 = [&](int64_t v, float a, float b, float c, float d) {
    if constexpr (H16) *reinterpret_cast<uint2*>(dxh + v) = make_uint2(pack2bf(a, b), pack2bf(c, d));
    else *reinterpret_cast<float4*>(dx + v) = make_float4(a, b, c, d);
  };
  const int64_t t = target[(row / L) * ldt + (row % L)];
  float rl = 0.f;
  if (t == pad_idx) {
    if (dlogits) for (int64_t v = 4 * tid; v < ld_dx; v += 1024) put4(v, 0.f, 0.f, 0.f, 0.f);
  } else {
    float4 xv[NV4];
    const float ninf = -__builtin_huge_valf();
    float mx = ninf;
#pragma unroll
    for (int k = 0; k < NV4; ++k) {
      const int v = 4 * (tid + 256 * k);
      float4 q = make_float4(ninf, ninf, ninf, ninf);
      if (v + 3 < V) {
        q = *reinterpret_cast<const float4*>(x + v);
      } else if (v < V) {              // the row's last, partial quad: element loads (ld_x may equal V: nothing behind the row is touched)
        q.x = x[v];
        if (v + 1 < V) q.y = x[v + 1];
        if (v + 2 < V) q.z = x[v + 2];
      }
      xv[k] = q;
      mx = fmaxf(mx, fmaxf(fmaxf(q.x, q.y), fmaxf(q.z, q.w)));
    }
    // the non-pad count rides in the same reductions (its target loads were issued beside the row's): two block exchanges in all
    float cnt = 0.f;
    for (int64_t i = tid; i < R; i += 256) cnt += (target[(i / L) * ldt + (i % L)] != pad_idx) ? 1.f : 0.f;
    mx = block_reduce(mx, sh, true);
    float se = 0.f, sx = 0.f;
#pragma unroll
    for (int k = 0; k < NV4; ++k) {
      const int v = 4 * (tid + 256 * k);
      const float e[4] = {xv[k].x, xv[k].y, xv[k].z, xv[k].w};
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (v + j < V) { se += expf(e[j] - mx); sx += e[j]; }
    }
    block_reduce3(se, sx, cnt, sh3);
    const float lse = mx + logf(se);
    const float inv_cnt = 1.f / cnt;
    const float off = eps / (float)(V - 1), on = 1.f - eps;
    if (dlogits) {
      const float g = inv_cnt * (gscale ? gscale[0] : 1.f);
#pragma unroll
      for (int k = 0; k < NV4; ++k) {
        const int v = 4 * (tid + 256 * k);
        if (v < ld_dx) {
          const float e[4] = {xv[k].x, xv[k].y, xv[k].z, xv[k].w};
          float o[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) o[j] = (v + j < V) ? (expf(e[j] - lse) - ((int64_t)(v + j) == t ? on : off)) * g : 0.f;
          put4(v, o[0], o[1], o[2], o[3]);
        }
      }
    }
    if (tid == 0) {
      const float logp_t = x[t] - lse;
      const float sum_logp = sx - (float)V * lse;
      const float Cc = (on > 0.f ? on * logf(on) : 0.f) + (eps > 0.f ? eps * logf(off) : 0.f);
      rl = (Cc - (on * logp_t + off * (sum_logp - logp_t))) * inv_cnt;
    }
  }
  if (tid == 0) {
    __hip_atomic_store(row_loss + row, rl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned int arrived = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    last_flag = (arrived == (unsigned int)(R - 1));
  }
  __syncthreads();
  if (!last_flag) return;
  float c = 0.f;
  for (int64_t i = tid; i < R; i += 256) c += __hip_atomic_load(row_loss + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  c = block_reduce(c, sh, false);
  if (tid == 0) {
    *loss = c;
    __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

extern "C" int32_t otr_label_smoothing_loss_fused(const float* logits, int64_t ld_logits, const int64_t* target, int64_t ld_target,
                                                  int32_t L, int64_t R, int32_t V, float smoothing, int32_t pad_idx,
                                                  const float* grad_scale, float* loss, void* dlogits, int32_t dlogits_dtype,
                                                  int64_t ld_dlogits, float* scratch, uint32_t* ticket, void* stream) {
  OTR_REQUIRE(logits && target && loss && scratch && ticket, "label_smoothing_loss_fused: null pointer");
  OTR_REQUIRE(R > 0 && V > 1 && L > 0 && R % L == 0 && ld_target >= L, "label_smoothing_loss_fused: bad shape R=%lld L=%d ld_target=%lld V=%d",
              (long long)R, L, (long long)ld_target, V);
  OTR_REQUIRE(R <= LS_COUNT_INLINE, "label_smoothing_loss_fused: more than %lld rows (use otr_label_smoothing_loss_ld)", (long long)LS_COUNT_INLINE);
  OTR_REQUIRE(V <= 8192, "label_smoothing_loss_fused: V = %d > 8192 does not fit a row in registers (use otr_label_smoothing_loss_ld)", V);
  OTR_REQUIRE(ld_logits >= V && ld_logits % 4 == 0 && (uintptr_t)logits % 16 == 0, "label_smoothing_loss_fused: logits rows must be 16-byte aligned (ld %% 4 == 0)");
  OTR_REQUIRE(dlogits_dtype == OTR_F32 || dlogits_dtype == OTR_H16, "label_smoothing_loss_fused: bad dlogits dtype %d", dlogits_dtype);
  OTR_REQUIRE(!dlogits || (ld_dlogits >= V && ld_dlogits % 4 == 0 && ld_dlogits <= 8192 && (uintptr_t)dlogits % 16 == 0 &&
                           (dlogits_dtype == OTR_F32 || ld_dlogits % 8 == 0)),
              "label_smoothing_loss_fused: dlogits rows must be 16-byte aligned (ld %% 4 == 0 for f32, %% 8 for 16-bit; ld <= 8192)");
  OTR_REQUIRE(smoothing >= 0.f && smoothing < 1.f, "label_smoothing_loss_fused: smoothing out of [0,1)");
  hipStream_t s = (hipStream_t)stream;
  const int64_t width = dlogits && ld_dlogits > V ? ld_dlogits : V;
#define OTR_LS_LAUNCH(NV4, H)                                                                                                            \
  hipLaunchKernelGGL((ls_rows_fused_kernel<NV4, H>), dim3((unsigned)R), dim3(256), 0, s, logits, ld_logits, target, ld_target, L, R, V, smoothing, \
                     pad_idx, grad_scale, scratch + 2, dlogits, ld_dlogits, loss, ticket)
  const bool h16 = dlogits_dtype == OTR_H16;
  if (width <= 2048) { if (h16) OTR_LS_LAUNCH(2, true); else OTR_LS_LAUNCH(2, false); }
  else if (width <= 5120) { if (h16) OTR_LS_LAUNCH(5, true); else OTR_LS_LAUNCH(5, false); }
  else { if (h16) OTR_LS_LAUNCH(8, true); else OTR_LS_LAUNCH(8, false); }
#undef OTR_LS_LAUNCH
  return otr_check_launch("label_smoothing_loss_fused");
}

extern "C" int32_t otr_label_smoothing_loss(const float* logits, const int64_t* target, int64_t R, int32_t V,
                                            float smoothing, int32_t pad_idx, float* loss, float* dlogits,
                                            float* scratch, void* stream) {
  return otr_label_smoothing_loss_ld(logits, V, target, R, V, smoothing, pad_idx, loss, dlogits, V, scratch, stream);
}

__global__ __launch_bounds__(256) void log_softmax_kernel(const float* x, float* y, int V) {
  __shared__ float sh[4];
  const float* xr = x + (int64_t)blockIdx.x * V;
  float* yr = y + (int64_t)blockIdx.x * V;
  float mx = -__builtin_huge_valf();
  for (int v = threadIdx.x; v < V; v += blockDim.x) mx = fmaxf(mx, xr[v]);
  mx = block_reduce(mx, sh, true);
  float se = 0.f;
  for (int v = threadIdx.x; v < V; v += blockDim.x) se += expf(xr[v] - mx);
  se = block_reduce(se, sh, false);
  const float lse = mx + logf(se);
  for (int v = threadIdx.x; v < V; v += blockDim.x) yr[v] = xr[v] - lse;
}

extern "C" int32_t otr_log_softmax(const float* x, float* y, int64_t R, int32_t V, void* stream) {
  OTR_REQUIRE(x && y, "log_softmax: null pointer");
  OTR_REQUIRE(R >= 0 && V > 0, "log_softmax: bad shape");
  if (R == 0) return 0;
  hipLaunchKernelGGL(log_softmax_kernel, dim3((unsigned)R), dim3(256), 0, (hipStream_t)stream, x, y, V);
  return otr_check_launch("log_softmax");
}
