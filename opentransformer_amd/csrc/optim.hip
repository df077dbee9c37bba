// Fused optimizer step over the flat parameter / gradient buffers of one data-parallel replica:
// grad-norm -> clip(5.0) -> NaN guard -> Adam with L2 weight decay -> Noam learning rate.
// Replaces train/trainer.py:221-234 (clip_grad_norm_, NaN guard, scheduler.step, optimizer.step) and
// train/scheduler.py:129-138 (TransformerScheduler) + torch.optim.Adam (train/scheduler.py:10-13).
// All scalars that change per step live in a small device-resident state block so the whole update
// is hipGraph-replayable:  state = {step, lr, bias_corr1, bias_corr2, sqnorm, skipped}.
#include "common.h"

extern int32_t* g_otr_fault;   // api.hip: sticky device fault word (otr_set_fault_counter) or NULL

struct OptState {
  float step;            // [0] number of optimizer updates applied so far (Adam's t)
  float lr;              // [1] learning rate used by the last update
  float bc1, bc2;        // [2,3] 1 - beta^t
  float sqnorm;          // [4] sum of squares of the flat gradient as it sits in memory (loss-scaled, not yet / world)
  float skipped;         // [5] count of updates skipped by the NaN guard
  float loss_scale;      // [6] 0 = loss scaling off; else the factor the backward pass was seeded with (fp16 builds:
                         //     ops.ScaleGradFn multiplies the loss gradient by this device scalar)
  float good_steps;      // [7] consecutive finite updates since loss_scale last changed
  float unscale;         // [8] grad_scale / loss_scale of THIS update (tick kernel -> adam kernel)
  float growth_interval; // [9] double loss_scale after this many finite updates (0 = never); set by the caller
  float faults;          // [10] total give-ups of spin-bounded kernels (otr_set_fault_counter) seen by the updates so far: each
                         //      such update was skipped like a non-finite one -- its gradients may be wrong sums
  float reserved[5];
  float norm_part[512];  // [16 .. 527] scratch: per-workgroup sums of squares of the gradient, summed by the tick kernel in a FIXED
                         //      order -- every rank of a data-parallel job then derives bit-identical clip factors from its
                         //      (bit-identical, all-reduced) gradients; an atomic sum over 512 workgroups differed in the last
                         //      bits from rank to rank and let the replicas drift apart
};
constexpr int OPT_NORM_WG = 512;

__global__ void sqnorm_kernel(const float* g, int64_t n, OptState* st) {
  __shared__ float sh[4];
  float s = 0.f;
  const int64_t n4 = n / 4;
  const float4* g4 = reinterpret_cast<const float4*>(g);
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + 3 * stride < n4; i += 4 * stride) {          // four 16-byte loads in flight per lane (one was latency-bound: 45 us for 150 MB)
    float4 v0 = g4[i], v1 = g4[i + stride], v2 = g4[i + 2 * stride], v3 = g4[i + 3 * stride];
    s += v0.x * v0.x + v0.y * v0.y + v0.z * v0.z + v0.w * v0.w;
    s += v1.x * v1.x + v1.y * v1.y + v1.z * v1.z + v1.w * v1.w;
    s += v2.x * v2.x + v2.y * v2.y + v2.z * v2.z + v2.w * v2.w;
    s += v3.x * v3.x + v3.y * v3.y + v3.z * v3.z + v3.w * v3.w;
  }
  for (; i < n4; i += stride) {
    float4 v = g4[i];
    s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) { float v = g[n4 * 4 + threadIdx.x]; s += v * v; }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) st->norm_part[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}

// one thread: NaN guard + dynamic loss scale, then advance the step counter and evaluate the schedule (Noam if
// warmup > 0, else constant lr)
__global__ void opt_tick_kernel(OptState* st, float base_lr, float model_size, float warmup, float factor,
                                float step_offset, float beta1, float beta2, float grad_scale, int32_t* fault, int nparts) {
  // one wave: the partial sums in a fixed order (lane l takes parts l, l + 64, ...; then the butterfly)
  float part = 0.f;
  for (int i = threadIdx.x; i < nparts; i += 64) part += st->norm_part[i];
  part = wave_sum(part);
  if (threadIdx.x != 0) return;
  st->sqnorm = part;
  const float ls = st->loss_scale > 0.f ? st->loss_scale : 1.f;
  const float us = grad_scale / ls;
  st->unscale = us;
  if (fault) {
    // a kernel of this step gave up a bounded wait (wgrad256 turnstile, fused-FFN arrival): a finite but WRONG gradient would
    // pass the NaN guard below, so the update is skipped here and the adam kernel is told through a NaN unscale factor
    const int32_t nf = __hip_atomic_exchange(fault, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (nf != 0) {
      st->faults += (float)nf;
      st->skipped += 1.f;
      st->unscale = __builtin_nanf("");
      return;
    }
  }
  float norm = sqrtf(st->sqnorm) * us;
  if (!isfinite(norm)) {                                   // trainer.py:229: skip the update
    st->skipped += 1.f;
    if (st->loss_scale > 0.f) { st->loss_scale = fmaxf(ls * 0.5f, 1.f); st->good_steps = 0.f; }   // fp16 overflow: back off
    return;
  }
  if (st->loss_scale > 0.f) {
    st->good_steps += 1.f;
    if (st->growth_interval > 0.f && st->good_steps >= st->growth_interval) {
      st->loss_scale = fminf(ls * 2.f, 65536.f);
      st->good_steps = 0.f;
    }
  }
  float t = st->step + 1.f;
  st->step = t;
  st->bc1 = 1.f - powf(beta1, t);
  st->bc2 = 1.f - powf(beta2, t);
  if (warmup > 0.f) {
    float s = t + step_offset;                             // reference quirk: first update uses step 3
    st->lr = factor * rsqrtf(model_size) * fminf(rsqrtf(s), s * powf(warmup, -1.5f));
  } else {
    st->lr = base_lr;
  }
}

// N(0,1) from the counter RNG (Box-Muller): gradient noise of train/trainer.py:223-227
__device__ __forceinline__ float otr_gauss(uint64_t seed, uint64_t idx) {
  const float u1 = ((float)otr_rand32_sm64(seed, 2 * idx) + 1.f) * (1.f / 4294967296.f);
  const float u2 = (float)otr_rand32_sm64(seed, 2 * idx + 1) * (1.f / 4294967296.f);
  return sqrtf(-2.f * __logf(u1)) * __cosf(6.2831853f * u2);
}

// One element of the update (torch.optim.Adam with L2 weight decay, train/trainer.py:221-234)
__device__ __forceinline__ void adam_one(float& pi, float gi, float& mi, float& vi, int64_t i, float coef, float wd, float beta1, float beta2,
                                         float step_size, float rbc2, float eps, float noise_std, uint64_t nseed) {
  // Cells where BOTH the gradient and the parameter are exactly zero are structural padding of the flat buffers (alignment gaps, the
  // extra rows of a row-padded Linear: include/otrans_hip.h): they get no noise, so they stay zero -- the invariant the padded
  // GEMMs rely on.  (A real parameter that is exactly 0 with an exactly 0 gradient is skipped too: measure zero.)
  const bool pad = gi == 0.f && pi == 0.f;
  gi *= coef;
  if (noise_std > 0.f && !pad) gi += noise_std * otr_gauss(nseed, (uint64_t)i);   // added after clipping, as the reference does
  gi += wd * pi;
  mi = beta1 * mi + (1.f - beta1) * gi;
  vi = beta2 * vi + (1.f - beta2) * gi * gi;
  pi = pi - step_size * mi / (sqrtf(vi) * rbc2 + eps);
}
// 16 bytes per lane and stream; p / m / v are read once and written once per step and nobody reads them before the next step's
// optimizer: non-temporal both ways, so 0.58 GB of state does not push the 16-bit weights (written here, read by the next step's
// first launches) and the packs out of the caches.  n4 = whole float4s, the tail (< 4 elements) is done by workgroup 0.
__global__ __launch_bounds__(256) void adam_kernel(float* p, const float* g, float* m, float* v, int64_t n, const OptState* st, bf16_t* p_lp,
                                                   float beta1, float beta2, float eps, float wd, float clip, float noise_std) {
  const float us = st->unscale;
  const float norm = sqrtf(st->sqnorm) * us;
  if (!isfinite(norm)) return;
  const float coef = us * (clip > 0.f ? fminf(1.f, clip / (norm + 1e-6f)) : 1.f);
  const float lr = st->lr, bc1 = st->bc1, rbc2 = rsqrtf(st->bc2);
  const float step_size = lr / bc1;
  const uint64_t nseed = 0x6E015Eull + (uint64_t)st->step * 0x9E3779B97F4A7C15ull;
  typedef float f4 __attribute__((ext_vector_type(4)));
  const int64_t n4 = n >> 2;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    f4 pv = __builtin_nontemporal_load(reinterpret_cast<const f4*>(p) + i);
    const f4 gv = *(reinterpret_cast<const f4*>(g) + i);
    f4 mv = __builtin_nontemporal_load(reinterpret_cast<const f4*>(m) + i);
    f4 vv = __builtin_nontemporal_load(reinterpret_cast<const f4*>(v) + i);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float pi = pv[e], mi = mv[e], vi = vv[e];
      adam_one(pi, gv[e], mi, vi, 4 * i + e, coef, wd, beta1, beta2, step_size, rbc2, eps, noise_std, nseed);
      pv[e] = pi; mv[e] = mi; vv[e] = vi;
    }
    __builtin_nontemporal_store(mv, reinterpret_cast<f4*>(m) + i);
    __builtin_nontemporal_store(vv, reinterpret_cast<f4*>(v) + i);
    __builtin_nontemporal_store(pv, reinterpret_cast<f4*>(p) + i);
    if (p_lp) *reinterpret_cast<uint2*>(p_lp + 4 * i) = make_uint2((uint32_t)f2bf(pv[0]) | ((uint32_t)f2bf(pv[1]) << 16), (uint32_t)f2bf(pv[2]) | ((uint32_t)f2bf(pv[3]) << 16));
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const int64_t i = 4 * n4 + threadIdx.x;
    float pi = p[i], mi = m[i], vi = v[i];
    adam_one(pi, g[i], mi, vi, i, coef, wd, beta1, beta2, step_size, rbc2, eps, noise_std, nseed);
    p[i] = pi; m[i] = mi; v[i] = vi;
    if (p_lp) p_lp[i] = f2bf(pi);
  }
}

extern "C" int32_t otr_optimizer_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                                      float* state, int32_t state_floats, void* param_bf16, float base_lr, float beta1, float beta2, float eps,
                                      float weight_decay, float grad_scale, float clip_norm, float noam_model_size,
                                      float noam_warmup, float noam_factor, float noam_step_offset, float grad_noise_std,
                                      void* stream) {
  OTR_REQUIRE(param && grad && exp_avg && exp_avg_sq && state, "optimizer_step: null pointer");
  OTR_REQUIRE(n > 0, "optimizer_step: empty parameter buffer");
  OTR_REQUIRE(grad_noise_std >= 0.f, "optimizer_step: grad_noise_std must be >= 0");
  OTR_REQUIRE((uintptr_t)grad % 16 == 0, "optimizer_step: grad buffer must be 16-byte aligned");
  // every argument check sits in front of the FIRST launch: a refused call must not have advanced the device state (step count,
  // loss scale, the consumed fault word) without applying an update
  OTR_REQUIRE(state_floats >= OTR_OPT_STATE_FLOATS, "optimizer_step: the state block must hold OTR_OPT_STATE_FLOATS floats "
              "(the gradient-norm partial sums live behind the 16 state slots)");
  OTR_REQUIRE(((uintptr_t)param | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) % 16 == 0 && (!param_bf16 || (uintptr_t)param_bf16 % 8 == 0),
              "optimizer_step: parameter / moment buffers must be 16-byte aligned (the 16-bit shadow 8-byte)");
  hipStream_t s = (hipStream_t)stream;
  OptState* st = reinterpret_cast<OptState*>(state);
  // <= 512 workgroups, each leaves ONE partial sum in the state block (no atomics: the sum order is fixed, see OptState)
  unsigned grid = (unsigned)((n / 4 + 255) / 256 > OPT_NORM_WG ? OPT_NORM_WG : (n / 4 + 255) / 256);
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL(sqnorm_kernel, dim3(grid), dim3(256), 0, s, grad, n, st);
  hipLaunchKernelGGL(opt_tick_kernel, dim3(1), dim3(64), 0, s, st, base_lr, noam_model_size, noam_warmup, noam_factor,
                     noam_step_offset, beta1, beta2, grad_scale, g_otr_fault, (int)grid);
  unsigned g2 = (unsigned)((n / 4 + 255) / 256 > 4096 ? 4096 : (n / 4 + 255) / 256);
  if (g2 < 1) g2 = 1;
  hipLaunchKernelGGL(adam_kernel, dim3(g2), dim3(256), 0, s, param, grad, exp_avg, exp_avg_sq, n, st, (bf16_t*)param_bf16, beta1, beta2, eps,
                     weight_decay, clip_norm, grad_noise_std);
  return otr_check_launch("optimizer_step");
}
