// Conv2d-subsampling frontend, the parts that are NOT GEMM-shaped (frontend/conv.py:50-83):
//  * conv1: C_in = 1 -> a 9-tap stencil, pure HBM streaming (reads B*T*F fp32, writes B*T1*F1*C1).
//    Channel-last output so conv2's implicit-GEMM gathers are contiguous 16-byte chunks.
//  * conv1 weight/bias gradient: a reduction over all output pixels.
//  * col2im of the conv2 input gradient (gather form, fused with the ReLU mask of act1).
// Thread layout everywhere: a thread owns 8 consecutive channels of one pixel (16 B of bf16), so the
// 8-16 lanes of a pixel write one contiguous C1 row and consecutive pixels are contiguous in memory.
#include "common.h"

struct ConvArgs {
  const float* x; const float* w1; const float* b1;
  void* act1; const void* dact1_in; float* dw1; float* db1; float* partial;
  const void* dcol; const void* act1_in; void* dact1_out;
  int B, T, F, C1, C2, T1, F1, T2, F2;
};

template <class T> __device__ __forceinline__ void store8(T* p, const float* v) {
  if constexpr (sizeof(T) == 4) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
  } else {
    *reinterpret_cast<uint4*>(p) = MMA<bf16_t>::pack(v);
  }
}

// ------------------------------------------------------------------------------------------------ conv1 forward
template <class T> __global__ __launch_bounds__(256) void conv1_fwd_kernel(ConvArgs p) {
  const int CG = p.C1 / 8;                       // channel groups per pixel
  const int cg = threadIdx.x % CG;
  const int pix_per_iter = blockDim.x / CG;
  float w[8][9], bias[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    bias[c] = p.b1[cg * 8 + c];
#pragma unroll
    for (int t = 0; t < 9; ++t) w[c][t] = p.w1[(cg * 8 + c) * 9 + t];
  }
  const int64_t npix = (int64_t)p.B * p.T1 * p.F1;
  T* out = reinterpret_cast<T*>(p.act1);
  const int64_t stride = (int64_t)gridDim.x * pix_per_iter;
  auto gather = [&](int64_t pix, float* in) {
    const uint32_t pu = (uint32_t)pix, bt = pu / (uint32_t)p.F1;       // 32-bit: conv_check bounds the pixel count
    const int f1 = (int)(pu - bt * (uint32_t)p.F1), b = (int)(bt / (uint32_t)p.T1), t1 = (int)(bt - (uint32_t)b * (uint32_t)p.T1);
    const float* xin = p.x + ((int64_t)b * p.T + 2 * t1) * p.F;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        int f = 2 * f1 + kw - 1;
        in[kh * 3 + kw] = (f >= 0 && f < p.F) ? xin[kh * p.F + f] : 0.f;
      }
  };
  auto emit = [&](int64_t pix, const float* in) {
    float o[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      float a = bias[c];
#pragma unroll
      for (int t = 0; t < 9; ++t) a = fmaf(w[c][t], in[t], a);
      o[c] = fmaxf(a, 0.f);
    }
    store8<T>(out + pix * p.C1 + cg * 8, o);
  };
  int64_t pix = (int64_t)blockIdx.x * pix_per_iter + threadIdx.x / CG;
  for (; pix + 3 * stride < npix; pix += 4 * stride) {      // four pixels per lane in flight: one left the 9 taps' latency exposed (49 us for 92 MB)
    float in[4][9];
#pragma unroll
    for (int u = 0; u < 4; ++u) gather(pix + u * stride, in[u]);
#pragma unroll
    for (int u = 0; u < 4; ++u) emit(pix + u * stride, in[u]);
  }
  for (; pix < npix; pix += stride) {
    float in[9];
    gather(pix, in);
    emit(pix, in);
  }
}

// ------------------------------------------------------------------------------------------------ conv1 wgrad
template <class T> __global__ __launch_bounds__(256) void conv1_wgrad_kernel(ConvArgs p) {
  __shared__ float red[256][10];                 // one tap-vector (or bias) at a time, padded
  const int CG = p.C1 / 8;
  const int cg = threadIdx.x % CG;
  const int pix_per_iter = blockDim.x / CG;
  float dw[8][9], db[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    db[c] = 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t) dw[c][t] = 0.f;
  }
  const int64_t npix = (int64_t)p.B * p.T1 * p.F1;
  const T* g = reinterpret_cast<const T*>(p.dact1_in);
  const int64_t stride = (int64_t)gridDim.x * pix_per_iter;
  auto gather = [&](int64_t pix, float* in, float* gv) {
    const uint32_t pu = (uint32_t)pix, bt = pu / (uint32_t)p.F1;       // 32-bit: conv_check bounds the pixel count
    const int f1 = (int)(pu - bt * (uint32_t)p.F1), b = (int)(bt / (uint32_t)p.T1), t1 = (int)(bt - (uint32_t)b * (uint32_t)p.T1);
    const float* xin = p.x + ((int64_t)b * p.T + 2 * t1) * p.F;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        int f = 2 * f1 + kw - 1;
        in[kh * 3 + kw] = (f >= 0 && f < p.F) ? xin[kh * p.F + f] : 0.f;
      }
    load_row<T, 8>(g + pix * p.C1 + cg * 8, 8, true, gv);
  };
  auto accum = [&](const float* in, const float* gv) {
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      db[c] += gv[c];
#pragma unroll
      for (int t = 0; t < 9; ++t) dw[c][t] = fmaf(gv[c], in[t], dw[c][t]);
    }
  };
  int64_t pix = (int64_t)blockIdx.x * pix_per_iter + threadIdx.x / CG;
  for (; pix + 1 * stride < npix; pix += 2 * stride) {      // two pixels per lane in flight (80 accumulators leave room for no more)
    float in[2][9], gv[2][8];
#pragma unroll
    for (int u = 0; u < 2; ++u) gather(pix + u * stride, in[u], gv[u]);
#pragma unroll
    for (int u = 0; u < 2; ++u) accum(in[u], gv[u]);
  }
  for (; pix < npix; pix += stride) {
    float in[9], gv[8];
    gather(pix, in, gv);
    accum(in, gv);
  }
  // reduce over the threads that share a channel group: per channel c, 10 values (9 taps + bias)
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 9; ++t) red[threadIdx.x][t] = dw[c][t];
    red[threadIdx.x][9] = db[c];
    __syncthreads();
    // thread (g2, v) for g2 < CG, v < 10 sums column v over threads with tid % CG == g2
    const int v = threadIdx.x % 10;
    for (int g2 = threadIdx.x / 10; g2 < CG && threadIdx.x < 250; g2 += 25) {
      float s = 0.f;
      for (int t = g2; t < 256; t += CG) s += red[t][v];
      int ch = g2 * 8 + c;
      if (p.partial) {             // per-workgroup sums, reduced by the caller's column sum: no atomics, 4x the workgroups
        float* row = p.partial + (int64_t)blockIdx.x * (p.C1 * 10);
        if (v < 9) row[ch * 9 + v] = s; else row[p.C1 * 9 + ch] = s;
      } else if (v < 9) atomicAdd(p.dw1 + ch * 9 + v, s);
      else atomicAdd(p.db1 + ch, s);
    }
  }
}

// ------------------------------------------------------------------------------------------------ col2im (+ ReLU mask)
// dact1[b,t1,f1,c] = [act1 > 0] * sum_{kh,kw valid} dcol[(b,t2,f2), (kh*3+kw)*C1 + c],
//   t2 = (t1-kh)/2, f2 = (f1+1-kw)/2 (both exact)
template <class T> __global__ __launch_bounds__(256) void col2im_kernel(ConvArgs p) {
  const int CG = p.C1 / 8;
  const int cg = threadIdx.x % CG;
  const int pix_per_iter = blockDim.x / CG;
  const int64_t npix = (int64_t)p.B * p.T1 * p.F1;
  const T* dcol = reinterpret_cast<const T*>(p.dcol);
  const T* act = reinterpret_cast<const T*>(p.act1_in);
  T* out = reinterpret_cast<T*>(p.dact1_out);
  const int64_t ldc = 9 * (int64_t)p.C1;
  for (int64_t pix = (int64_t)blockIdx.x * pix_per_iter + threadIdx.x / CG; pix < npix; pix += (int64_t)gridDim.x * pix_per_iter) {
    const uint32_t pu = (uint32_t)pix, bt = pu / (uint32_t)p.F1;       // 32-bit: conv_check bounds the pixel count
    const int f1 = (int)(pu - bt * (uint32_t)p.F1), b = (int)(bt / (uint32_t)p.T1), t1 = (int)(bt - (uint32_t)b * (uint32_t)p.T1);
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      int tt = t1 - kh;
      if (tt < 0 || (tt & 1)) continue;
      int t2 = tt >> 1;
      if (t2 >= p.T2) continue;
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        int ff = f1 + 1 - kw;
        if (ff < 0 || (ff & 1)) continue;
        int f2 = ff >> 1;
        if (f2 >= p.F2) continue;
        int64_t m = ((int64_t)b * p.T2 + t2) * p.F2 + f2;
        float v[8];
        load_row<T, 8>(dcol + m * ldc + (kh * 3 + kw) * p.C1 + cg * 8, 8, true, v);
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[c] += v[c];
      }
    }
    float a[8];
    load_row<T, 8>(act + pix * p.C1 + cg * 8, 8, true, a);
#pragma unroll
    for (int c = 0; c < 8; ++c) acc[c] = a[c] > 0.f ? acc[c] : 0.f;
    store8<T>(out + pix * p.C1 + cg * 8, acc);
  }
}

// ------------------------------------------------------------------------------------------------ host
static int32_t conv_check(const otr_conv_desc_t* d, ConvArgs& a) {
  OTR_REQUIRE(d != nullptr, "conv: null descriptor");
  OTR_REQUIRE(d->B > 0 && d->T >= 7 && d->F >= 3, "conv: bad input shape B=%d T=%d F=%d", d->B, d->T, d->F);
  OTR_REQUIRE(d->T1 == (d->T - 3) / 2 + 1 && d->T2 == (d->T1 - 3) / 2 + 1, "conv: T1/T2 inconsistent with T");
  OTR_REQUIRE(d->F1 == (d->F - 1) / 2 + 1 && d->F2 == (d->F1 - 1) / 2 + 1, "conv: F1/F2 inconsistent with F");
  OTR_REQUIRE(d->C1 >= 8 && d->C1 % 8 == 0 && d->C1 <= 256 && 256 % (d->C1 / 8) == 0,
              "conv: C1=%d must be a multiple of 8 with C1/8 dividing 256", d->C1);
  OTR_REQUIRE(d->act_dtype == OTR_F32 || d->act_dtype == OTR_H16, "conv: bad act dtype");
  OTR_REQUIRE((int64_t)d->B * d->T1 * d->F1 * d->C1 < (1ll << 31), "conv: act1 too large for 32-bit pixel index");
  a.B = d->B; a.T = d->T; a.F = d->F; a.C1 = d->C1; a.C2 = d->C2;
  a.T1 = d->T1; a.F1 = d->F1; a.T2 = d->T2; a.F2 = d->F2;
  return 0;
}
static unsigned conv_grid(const ConvArgs& a) {
  int64_t npix = (int64_t)a.B * a.T1 * a.F1;
  int ppi = 256 / (a.C1 / 8);
  int64_t g = (npix + ppi - 1) / ppi;
  return (unsigned)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

extern "C" int32_t otr_conv1_fwd(const otr_conv_desc_t* d, const float* x, const float* w1, const float* b1,
                                 void* act1, void* stream) {
  ConvArgs a{};
  if (int32_t e = conv_check(d, a)) return e;
  OTR_REQUIRE(x && w1 && b1 && act1, "conv1_fwd: null pointer");
  a.x = x; a.w1 = w1; a.b1 = b1; a.act1 = act1;
  hipStream_t s = (hipStream_t)stream;
  if (d->act_dtype == OTR_F32) hipLaunchKernelGGL(conv1_fwd_kernel<float>, dim3(conv_grid(a)), dim3(256), 0, s, a);
  else hipLaunchKernelGGL(conv1_fwd_kernel<bf16_t>, dim3(conv_grid(a)), dim3(256), 0, s, a);
  return otr_check_launch("conv1_fwd");
}

constexpr int CONV1_WGRAD_PARTIAL_BLOCKS = 1024;
extern "C" int32_t otr_conv1_wgrad_partial_rows(void) { return CONV1_WGRAD_PARTIAL_BLOCKS; }
extern "C" int32_t otr_conv1_wgrad(const otr_conv_desc_t* d, const float* x, const void* dact1, float* dw1, float* db1,
                                   float* partial, void* stream) {
  ConvArgs a{};
  if (int32_t e = conv_check(d, a)) return e;
  OTR_REQUIRE(x && dact1 && (partial || (dw1 && db1)), "conv1_wgrad: null pointer");
  a.x = x; a.dact1_in = dact1; a.dw1 = dw1; a.db1 = db1; a.partial = partial;
  hipStream_t s = (hipStream_t)stream;
  unsigned g = conv_grid(a);
  // atomics: every block ends with 10*C1 atomics on the same addresses (1024 blocks spent ~75 % of the kernel there) -> 256
  // blocks; with `partial` every block writes its own row and the chip can be filled (4 waves per CU left the 80-accumulator
  // reduction latency-bound at 124 us)
  if (partial) g = CONV1_WGRAD_PARTIAL_BLOCKS; else if (g > 256) g = 256;
  if (d->act_dtype == OTR_F32) hipLaunchKernelGGL(conv1_wgrad_kernel<float>, dim3(g), dim3(256), 0, s, a);
  else hipLaunchKernelGGL(conv1_wgrad_kernel<bf16_t>, dim3(g), dim3(256), 0, s, a);
  return otr_check_launch("conv1_wgrad");
}

extern "C" int32_t otr_conv2_col2im(const otr_conv_desc_t* d, const void* dcol, const void* act1, void* dact1,
                                    void* stream) {
  ConvArgs a{};
  if (int32_t e = conv_check(d, a)) return e;
  OTR_REQUIRE(dcol && act1 && dact1, "conv2_col2im: null pointer");
  a.dcol = dcol; a.act1_in = act1; a.dact1_out = dact1;
  hipStream_t s = (hipStream_t)stream;
  if (d->act_dtype == OTR_F32) hipLaunchKernelGGL(col2im_kernel<float>, dim3(conv_grid(a)), dim3(256), 0, s, a);
  else hipLaunchKernelGGL(col2im_kernel<bf16_t>, dim3(conv_grid(a)), dim3(256), 0, s, a);
  return otr_check_launch("conv2_col2im");
}
