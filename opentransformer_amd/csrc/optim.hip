// Fused optimizer step over the flat parameter / gradient buffers of one data-parallel replica:
// grad-norm -> clip(5.0) -> NaN guard -> Adam with L2 weight decay -> Noam learning rate.
// Replaces train/trainer.py:221-234 (clip_grad_norm_, NaN guard, scheduler.step, optimizer.step) and
// train/scheduler.py:129-138 (TransformerScheduler) + torch.optim.Adam (train/scheduler.py:10-13).
// All scalars that change per step live in a small device-resident state block so the whole update
// is hipGraph-replayable:  state = {step, lr, bias_corr1, bias_corr2, sqnorm, skipped}.
#include "common.h"

struct OptState {
  float step;        // number of optimizer updates applied so far (Adam's t)
  float lr;          // learning rate used by the last update
  float bc1, bc2;    // 1 - beta^t
  float sqnorm;      // sum of squares of the (unscaled) flat gradient
  float skipped;     // count of updates skipped by the NaN guard
};

__global__ void sqnorm_kernel(const float* g, int64_t n, OptState* st) {
  __shared__ float sh[4];
  float s = 0.f;
  const int64_t n4 = n / 4;
  const float4* g4 = reinterpret_cast<const float4*>(g);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    float4 v = g4[i];
    s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) { float v = g[n4 * 4 + threadIdx.x]; s += v * v; }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(&st->sqnorm, sh[0] + sh[1] + sh[2] + sh[3]);
}

// one thread: advance the step counter, evaluate the schedule (Noam if warmup > 0, else constant lr)
__global__ void opt_tick_kernel(OptState* st, float base_lr, float model_size, float warmup, float factor,
                                float step_offset, float beta1, float beta2, float grad_scale) {
  float norm = sqrtf(st->sqnorm) * grad_scale;
  if (!isfinite(norm)) { st->skipped += 1.f; return; }     // trainer.py:229: skip the update
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

__global__ void adam_kernel(float* p, const float* g, float* m, float* v, int64_t n, const OptState* st, bf16_t* p_lp,
                            float beta1, float beta2, float eps, float wd, float grad_scale, float clip) {
  const float norm = sqrtf(st->sqnorm) * grad_scale;
  if (!isfinite(norm)) return;
  const float coef = grad_scale * (clip > 0.f ? fminf(1.f, clip / (norm + 1e-6f)) : 1.f);
  const float lr = st->lr, bc1 = st->bc1, rbc2 = rsqrtf(st->bc2);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float pi = p[i];
    float gi = g[i] * coef + wd * pi;
    float mi = beta1 * m[i] + (1.f - beta1) * gi;
    float vi = beta2 * v[i] + (1.f - beta2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    pi = pi - (lr / bc1) * mi / (sqrtf(vi) * rbc2 + eps);
    p[i] = pi;
    if (p_lp) p_lp[i] = f2bf(pi);
  }
}

extern "C" int32_t otr_optimizer_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                                      float* state, void* param_bf16, float base_lr, float beta1, float beta2, float eps,
                                      float weight_decay, float grad_scale, float clip_norm, float noam_model_size,
                                      float noam_warmup, float noam_factor, float noam_step_offset, void* stream) {
  OTR_REQUIRE(param && grad && exp_avg && exp_avg_sq && state, "optimizer_step: null pointer");
  OTR_REQUIRE(n > 0, "optimizer_step: empty parameter buffer");
  OTR_REQUIRE((uintptr_t)grad % 16 == 0, "optimizer_step: grad buffer must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  OptState* st = reinterpret_cast<OptState*>(state);
  otr_zero_f32(&st->sqnorm, 1, s);
  unsigned grid = (unsigned)((n / 4 + 255) / 256 > 2048 ? 2048 : (n / 4 + 255) / 256);
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL(sqnorm_kernel, dim3(grid), dim3(256), 0, s, grad, n, st);
  hipLaunchKernelGGL(opt_tick_kernel, dim3(1), dim3(1), 0, s, st, base_lr, noam_model_size, noam_warmup, noam_factor,
                     noam_step_offset, beta1, beta2, grad_scale);
  unsigned g2 = (unsigned)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
  hipLaunchKernelGGL(adam_kernel, dim3(g2), dim3(256), 0, s, param, grad, exp_avg, exp_avg_sq, n, st, (bf16_t*)param_bf16, beta1, beta2, eps,
                     weight_decay, grad_scale, clip_norm);
  return otr_check_launch("optimizer_step");
}
