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
  // (four pixels per lane in flight were tried: 47.5 us against 48.7 -- the kernel is bound by its ~100 VALU instructions per
  //  pixel and channel group, not by the latency of the nine taps)
  for (int64_t pix = (int64_t)blockIdx.x * pix_per_iter + threadIdx.x / CG; pix < npix; pix += stride) {
    float in[9];
    gather(pix, in);
    emit(pix, in);
  }
}

// ------------------------------------------------------------------------------------------------ conv1 forward on the fp32 matrix pipe
// 64 output channels, 16-bit activations.  The stencil kernel above spends ~100 VALU instructions per (pixel, 8 channels) and is
// bound by them (46 us for 0.7 GFLOP).  As a GEMM the layer is out[pixel][c] = sum_tap patch[pixel][tap] w[c][tap] with 9 taps:
// v_mfma_f32_32x32x2_f32 -- fp32 products and sums, exactly the arithmetic of the stencil up to the order of the nine additions --
// takes taps in pairs: A = w (lane (channel, tap parity)), B = the patches (lane (pixel, tap parity)): FIVE loads per lane give 32
// pixels x 64 channels (10 MFMAs).  The accumulators start from the bias; ReLU and the 16-bit conversion run on them; a tile of 32
// consecutive pixels is one contiguous 4 KiB block of the channel-last output, written through LDS as whole lines.
constexpr int C1M_PITCH = 144;                   // bytes per pixel row of the LDS image (128 + 16: the 8-byte pieces of a tile spread over the banks)
__global__ __launch_bounds__(256) void conv1_fwd_mfma_kernel(ConvArgs p) {
  typedef __attribute__((ext_vector_type(16))) float f32x16_t;
  __shared__ __attribute__((aligned(16))) unsigned char smem[4 * 32 * C1M_PITCH];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, m = lane & 31, hi = lane >> 5;
  unsigned char* img = smem + wid * (32 * C1M_PITCH);
  // r06: blockIdx.y = the group of 64 output channels this workgroup produces (C1 = 64: one group; the Conformer's 256: four -- the
  // VALU stencil took 138 us there); the input patches are re-read per group (they are 1/C1 of the output's bytes)
  const int c0 = (int)blockIdx.y * 64;
  // A operands: step s covers taps 2s, 2s + 1; lane (channel 32 ct + m, tap 2s + hi); tap 9 does not exist
  float wa[2][5], bias[2][16];
#pragma unroll
  for (int ct = 0; ct < 2; ++ct) {
#pragma unroll
    for (int s5 = 0; s5 < 5; ++s5) wa[ct][s5] = (2 * s5 + hi < 9) ? p.w1[(c0 + 32 * ct + m) * 9 + 2 * s5 + hi] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) bias[ct][r] = p.b1[c0 + 32 * ct + 8 * (r >> 2) + 4 * hi + (r & 3)];
  }
  const int64_t npix = (int64_t)p.B * p.T1 * p.F1;
  const int64_t ntile = (npix + 31) / 32;
  uint16_t* out = reinterpret_cast<uint16_t*>(p.act1);
  for (int64_t tile = (int64_t)blockIdx.x * 4 + wid; tile < ntile; tile += (int64_t)gridDim.x * 4) {
    const int64_t pix = tile * 32 + m;
    const uint32_t pu = (uint32_t)(pix < npix ? pix : npix - 1), bt = pu / (uint32_t)p.F1;       // 32-bit: conv_check bounds the pixel count
    const int f1 = (int)(pu - bt * (uint32_t)p.F1), b = (int)(bt / (uint32_t)p.T1), t1 = (int)(bt - (uint32_t)b * (uint32_t)p.T1);
    const float* xin = p.x + ((int64_t)b * p.T + 2 * t1) * p.F;
    float xb[5];
#pragma unroll
    for (int s5 = 0; s5 < 5; ++s5) {
      const int tap = 2 * s5 + hi, kh = tap / 3, kw = tap - 3 * kh, f = 2 * f1 + kw - 1;
      const bool ok = tap < 9 && f >= 0 && f < p.F;
      const float v = xin[(ok ? kh : 0) * p.F + (ok ? f : 0)];
      xb[s5] = ok ? v : 0.f;
    }
    f32x16_t acc[2];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ct][r] = bias[ct][r];
#pragma unroll
      for (int s5 = 0; s5 < 5; ++s5) acc[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[ct][s5], xb[s5], acc[ct], 0, 0, 0);
    }
    // D: lane (pixel m, hi) holds channels 32 ct + 8 q + 4 hi + (0..3) for q = r >> 2: 8-byte pieces of the pixel's 128-byte row
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const uint2 v = make_uint2(pack2h(fmaxf(acc[ct][4 * q], 0.f), fmaxf(acc[ct][4 * q + 1], 0.f)),
                                   pack2h(fmaxf(acc[ct][4 * q + 2], 0.f), fmaxf(acc[ct][4 * q + 3], 0.f)));
        *reinterpret_cast<uint2*>(img + m * C1M_PITCH + (32 * ct + 8 * q + 4 * hi) * 2) = v;
      }
    __builtin_amdgcn_s_waitcnt(0xc07f);          // lgkmcnt(0): this wave's LDS writes (the image is wave-private: no barrier)
    const int64_t pix0 = tile * 32;
    const int nvalid = (int)(npix - pix0 < 32 ? npix - pix0 : 32);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int pr = 8 * i + (lane >> 3), piece = lane & 7;
      const uint4 v = *reinterpret_cast<const uint4*>(img + pr * C1M_PITCH + piece * 16);
      if (pr < nvalid) *reinterpret_cast<uint4*>(out + (pix0 + pr) * p.C1 + c0 + piece * 8) = v;
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);          // reads done before the next tile overwrites the image
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
  for (int64_t pix = (int64_t)blockIdx.x * pix_per_iter + threadIdx.x / CG; pix < npix; pix += stride) {
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
extern int g_otr_conv1_stencil;    // api.hip (otr_debug_set(17, 1)): the VALU stencil for every shape, for A/B runs
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
  else if (a.C1 % 64 == 0 && !g_otr_conv1_stencil && (uintptr_t)act1 % 16 == 0) {
    const int64_t ntile = ((int64_t)a.B * a.T1 * a.F1 + 31) / 32;
    const int64_t g = (ntile + 3) / 4;
    hipLaunchKernelGGL(conv1_fwd_mfma_kernel, dim3((unsigned)(g < 1 ? 1 : (g > 2048 ? 2048 : g)), (unsigned)(a.C1 / 64)), dim3(256), 0, s, a);
  } else hipLaunchKernelGGL(conv1_fwd_kernel<bf16_t>, dim3(conv_grid(a)), dim3(256), 0, s, a);
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

// ------------------------------------------------------------------------------------------------ conv2 input gradient, implicit
// dact1[b,t1,f1,:] = [act1 > 0] * sum over the taps (kh,kw) that reach it of  W[:,kh,kw,:]^T g2[b,t2,f2,:],
//   t2 = (t1-kh)/2, f2 = (f1+1-kw)/2 (both exact and in range).
// Stride 2 splits the output pixels into four parity classes (t1 & 1, f1 & 1) with 4 / 2 / 2 / 1 taps; inside a class every
// pixel sees the same taps, so the class is a plain GEMM  D[c1, pixel] = sum_tap A_tap[c1, c2] B_tap[c2, pixel]  with
//   A_tap = W[:,kh,kw,:]^T (at most 4 x 16 KB, built once per workgroup in LDS as MFMA A-fragments),
//   B_tap = rows of g2: lane (pixel, hi) reads 16 contiguous bytes of its pixel's channel row -- straight from global memory into
//           the B operand, no staging (a pixel row is read by one wave only).
// D comes out with 4 consecutive c1 per lane and register group, i.e. 8-byte pieces of the channel-last dact1 rows, where the
// ReLU mask of act1 is applied.  Nothing like the [pixels, 9*C1] column matrix of the explicit form is written or read
// (184 MB each way at the AISHELL shapes: the GEMM + col2im pair took 77 + 70 us).  Workgroups are persistent, belong to one
// class (their A fragments never change) and the classes get workgroups in proportion to pixels x taps.
extern unsigned long long* g_otr_trace;   // api.hip (otr_debug_trace)
int g_otr_conv2_dgrad_ablate = 0;         // tuning hook (otr_debug_set(10, v)), see Conv2DgArgs
int g_otr_conv2_dgrad_wide = 1;           // the sliced form for 256-channel outputs (otr_debug_set(30, 0) = column matrix + col2im)
struct Conv2DgArgs {
  const uint16_t* g2; const uint16_t* w2r; const uint16_t* act1; uint16_t* dact1;
  int B, T1, F1, T2, F2;
  int wg0[5];                       // class c = 2*(t1&1) + (f1&1) owns workgroups [wg0[c], wg0[c+1])
  int ablate;                       // tuning hook (otr_debug_set(10, v)): 1 = no mask loads / result stores, 2 = every operand load from one line
  int C1full, nslice;               // sliced form (conv2_dgrad_sliced_kernel): the channels of act1, and how many 64-channel slices they make
  unsigned long long* trace;        // tuning hook (otr_debug_trace): [workgroup][4] = 100 MHz real-time at start, after the A
                                    // fragments are built, at the end, and the class; or NULL
};

// one parity class (PT = t1 & 1, PF = f1 & 1): the tap count is a compile-time constant, so a tile is straight-line code --
// every load is unconditional (the last tile prefetches its own first tap again) and hipcc can count vmcnt exactly instead
// of draining the prefetch before the MFMAs that do not need it
// SL (sliced form, wide frontends): the workgroup owns the C1 = RT * 32 channels from c1_0 of act1's p.C1full
template <int RT, int KS, int PT, int PF, bool SL = false>
__device__ __forceinline__ void conv2_dgrad_class(const Conv2DgArgs& p, uint4* afrag, uint4* ebuf_all, int w, int nwg, int c1_0 = 0) {
  constexpr int C1 = RT * 32, C2 = KS * 16, NKW = PF ? 2 : 1, NT = (PT ? 1 : 2) * NKW;
  const int ldc = SL ? p.C1full : C1;
  // local tap tt = a * NKW + b2:  kh = PT ? 1 : 2a,  kw = PF ? 2 b2 : 1
  constexpr int NENT = NT * RT * KS * 64;
#pragma unroll
  for (int it = 0; it < (NENT + 511) / 512; ++it) {            // unrolled: the loads of all entries are in flight together
    const int e = it * 512 + (int)threadIdx.x;
    if (NENT % 512 != 0 && e >= NENT) break;
    const int ln = e & 63, f = e >> 6, ks = f % KS, rt = (f / KS) % RT, tt = f / (KS * RT);
    const int a = tt / NKW, b2 = tt - a * NKW;
    const int tap = (PT ? 1 : 2 * a) * 3 + (PF ? 2 * b2 : 1);
    const uint16_t* src = p.w2r + ((int64_t)(ks * 16 + (ln >> 5) * 8) * 9 + tap) * ldc + c1_0 + rt * 32 + (ln & 31);
    uint32_t v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = src[(int64_t)j * 9 * ldc];
    afrag[e] = make_uint4(v[0] | (v[1] << 16), v[2] | (v[3] << 16), v[4] | (v[5] << 16), v[6] | (v[7] << 16));
  }
  __syncthreads();
  if (p.trace && threadIdx.x == 0) p.trace[(int64_t)blockIdx.x * 4 + 1] = __builtin_amdgcn_s_memrealtime();
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, hi = lane >> 5, pl = lane & 31;
  const int nT = PT ? p.T1 / 2 : (p.T1 + 1) / 2, nF = PF ? p.F1 / 2 : (p.F1 + 1) / 2;
  const int Mc = p.B * nT * nF;
  const int ntile = (Mc + 255) / 256;
  if (w >= ntile) return;

  struct Pix { int b, i, j, live; };
  auto pix_of = [&](int tl) {
    const int m = tl * 256 + wid * 32 + pl;
    const uint32_t mc = (uint32_t)min(m, Mc - 1), bi = mc / (uint32_t)nF, b = bi / (uint32_t)nT;
    return Pix{(int)b, (int)(bi - b * (uint32_t)nT), (int)(mc - bi * (uint32_t)nF), m < Mc};
  };
  // address of lane (pixel, hi)'s 16-byte pieces of tap tt; false = the tap falls outside g2 for this pixel (operand = 0)
  auto src_of = [&](const Pix& px, int tt, const uint16_t*& src) {
    const int a = tt / NKW, b2 = tt - a * NKW;
    const int t2 = PT ? px.i : px.i - a, f2 = PF ? px.j + 1 - b2 : px.j;
    const int t2c = min(max(t2, 0), p.T2 - 1), f2c = min(max(f2, 0), p.F2 - 1);
    src = p.g2 + ((int64_t)((px.b * p.T2 + t2c) * p.F2 + f2c)) * C2 + hi * 8;
    return t2 >= 0 && t2 < p.T2 && f2 >= 0 && f2 < p.F2;
  };

  f32x16 acc[RT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[rt][r] = 0.f;
  Pix px = pix_of(w);
  uint4 cur[KS];
  const uint16_t* srcc;
  bool okc = src_of(px, 0, srcc);
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) cur[ks] = ld_global_b128(srcc + ks * 16);
  // epilogue layout: the accumulators hold 8-byte pieces of 32 different pixel rows per lane group, and a store of such
  // pieces costs the address path of the CU one cycle per LANE (every lane a different 128-byte line; so did the ReLU-mask
  // loads: 16 such instructions per tile were most of the kernel's first 116 us).  Each 16-channel slab goes through 1 KB
  // of LDS per wave instead and leaves as 16 bytes per lane, lanes 2i / 2i+1 on the same row.
  uint4* ebuf = ebuf_all + wid * 64;
  unsigned char* ebytes = reinterpret_cast<unsigned char*>(ebuf);
  const int erow = lane >> 1, ehalf = lane & 1;               // the pixel row / 8-channel half this lane stores
  auto ebase_of = [&](int tl, bool& live) {
    const int m = tl * 256 + wid * 32 + erow;
    live = m < Mc;
    const uint32_t mc = (uint32_t)min(m, Mc - 1), bi = mc / (uint32_t)nF, b = bi / (uint32_t)nT;
    const int i = (int)(bi - b * (uint32_t)nT), j = (int)(mc - bi * (uint32_t)nF);
    return ((int64_t)(((int)b * p.T1 + 2 * i + PT) * p.F1) + 2 * j + PF) * ldc + c1_0 + 8 * ehalf;
  };
  for (int tile = w; tile < ntile; tile += nwg) {
    const Pix npx = tile + nwg < ntile ? pix_of(tile + nwg) : px;
    bool elive;
    const int64_t obase = ebase_of(tile, elive);
    uint4 am[RT * 2];
    // the A fragments do not depend on the tile: without this opaque zero hipcc hoists all NT*RT*KS LDS reads out of the
    // tile loop (256 registers at the AISHELL shape, 455 spilled)
    int opq = 0;
    asm volatile("" : "+v"(opq));
    const uint4* af = afrag + opq + lane;
    uint4 ac[RT], an[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) ac[rt] = af[rt * KS * 64];
#pragma unroll
    for (int tt = 0; tt < NT; ++tt) {
      const uint16_t* srcn;
      const bool okn = tt + 1 < NT ? src_of(px, tt + 1, srcn) : src_of(npx, 0, srcn);
      if (p.ablate & 2) srcn = p.g2 + hi * 8;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        if (tt == 0 && ks == 0) {                              // ReLU mask of this tile's pixels (HBM every time): in flight under the whole tile
#pragma unroll
          for (int g = 0; g < RT * 2; ++g) am[g] = (p.ablate & 1) ? make_uint4(0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u) : ld_global_b128(p.act1 + obase + g * 16);
        }
        const uint4 bq = okc ? cur[ks] : make_uint4(0u, 0u, 0u, 0u);
        {                                                      // next step's A fragments: their LDS latency under this step's MFMAs
          const int nt = ks + 1 < KS ? tt : (tt + 1 < NT ? tt + 1 : 0), nk = ks + 1 < KS ? ks + 1 : 0;
#pragma unroll
          for (int rt = 0; rt < RT; ++rt) an[rt] = af[((nt * RT + rt) * KS + nk) * 64];
        }
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) mma32(acc[rt], ac[rt], bq);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) ac[rt] = an[rt];
        // the registers just consumed take the same pieces of the next tap (or of the next tile's first tap): the operand set
        // stays in flight under the MFMAs without a second buffer.  Four at a time = one 128-byte line of every pixel row:
        // issued one by one between the MFMAs, the four touches of a line were a whole tap apart and the 16 waves of a CU
        // (128 KB of rows in flight, 32 KB of L1) evicted it in between -- every line came from L2 four times (116 us)
        if ((ks & 3) == 3) {
#pragma unroll
          for (int k2 = ks - 3; k2 <= ks; ++k2) cur[k2] = ld_global_b128(srcn + k2 * 16);
        }
        __builtin_amdgcn_sched_barrier(0);                     // keep the loads here: hipcc otherwise sinks them to just before their use
      }
      okc = okn;
    }
#pragma unroll
    for (int g = 0; g < RT * 2; ++g) {                       // 16-channel slab g: c1 in [16g, 16g + 16)
      const int rt = g >> 1, q0 = 2 * (g & 1);
#pragma unroll
      for (int qq = 0; qq < 2; ++qq) {
        const int q = q0 + qq;
        otr_u32x2 w = {pack2h(acc[rt][4 * q], acc[rt][4 * q + 1]), pack2h(acc[rt][4 * q + 2], acc[rt][4 * q + 3])};
        *reinterpret_cast<otr_u32x2*>(ebytes + pl * 32 + qq * 16 + hi * 8) = w;
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[rt][4 * q + r] = 0.f;
      }
      asm volatile("" ::: "memory");                           // the pieces above are read back by other lanes of this wave
      const uint4 v = ebuf[lane];                              // (LDS operations of a wave complete in order: no barrier)
      asm volatile("" ::: "memory");
      const uint4 a = am[g];
      auto keep = [](uint32_t act, uint32_t val) {
        return ((int16_t)(act & 0xffffu) > 0 ? (val & 0xffffu) : 0u) | ((int16_t)(act >> 16) > 0 ? (val & 0xffff0000u) : 0u);
      };
      if (elive && !(p.ablate & 1)) st_global_b128(p.dact1 + obase + g * 16, make_uint4(keep(a.x, v.x), keep(a.y, v.y), keep(a.z, v.z), keep(a.w, v.w)));
    }
    px = npx;
  }
}

template <int RT, int KS> __global__ __launch_bounds__(512, 4) void conv2_dgrad_kernel(Conv2DgArgs p) {
  __shared__ uint4 afrag[4 * RT * KS * 64];
  __shared__ uint4 ebuf[8 * 64];                              // 1 KB per wave: epilogue transposition
  int cls = 0;
  while (cls < 3 && (int)blockIdx.x >= p.wg0[cls + 1]) ++cls;
  const int w = (int)blockIdx.x - p.wg0[cls], nwg = p.wg0[cls + 1] - p.wg0[cls];
  if (p.trace && threadIdx.x == 0) {
    p.trace[(int64_t)blockIdx.x * 4 + 0] = __builtin_amdgcn_s_memrealtime();
    p.trace[(int64_t)blockIdx.x * 4 + 3] = (unsigned long long)cls;
  }
  if (cls == 0) conv2_dgrad_class<RT, KS, 0, 0>(p, afrag, ebuf, w, nwg);
  else if (cls == 1) conv2_dgrad_class<RT, KS, 0, 1>(p, afrag, ebuf, w, nwg);
  else if (cls == 2) conv2_dgrad_class<RT, KS, 1, 0>(p, afrag, ebuf, w, nwg);
  else conv2_dgrad_class<RT, KS, 1, 1>(p, afrag, ebuf, w, nwg);
  __syncthreads();
  if (p.trace && threadIdx.x == 0) p.trace[(int64_t)blockIdx.x * 4 + 2] = __builtin_amdgcn_s_memrealtime();
}

// Wide frontends (the Conformer's 256 -> 256, conformer_baseline.yaml): the A fragments of ALL output channels do not fit LDS (4 taps x
// 256 x 256 x 2 B), those of a 64-channel slice do (128 KB) -- the grid is nslice copies of the plan, copy s owning channels 64 s .. + 63.
// The copies of one workgroup index share an XCD (the plan's size is a multiple of 8) and read the same g2 rows there.  One workgroup
// per CU (136 KB of LDS), 8 waves.  It replaces a [pixels, 9 C1] column matrix written by a GEMM and read back by col2im (696 MB each
// way at batch 32: 367 + 275 us).
template <int RT, int KS> __global__ __launch_bounds__(512, 2) void conv2_dgrad_sliced_kernel(Conv2DgArgs p) {
  __shared__ uint4 afrag[4 * RT * KS * 64];
  __shared__ uint4 ebuf[8 * 64];
  const int per = p.wg0[4], slice = (int)blockIdx.x / per, bx = (int)blockIdx.x - slice * per;
  int cls = 0;
  while (cls < 3 && bx >= p.wg0[cls + 1]) ++cls;
  const int w = bx - p.wg0[cls], nwg = p.wg0[cls + 1] - p.wg0[cls], c1_0 = slice * RT * 32;
  if (cls == 0) conv2_dgrad_class<RT, KS, 0, 0, true>(p, afrag, ebuf, w, nwg, c1_0);
  else if (cls == 1) conv2_dgrad_class<RT, KS, 0, 1, true>(p, afrag, ebuf, w, nwg, c1_0);
  else if (cls == 2) conv2_dgrad_class<RT, KS, 1, 0, true>(p, afrag, ebuf, w, nwg, c1_0);
  else conv2_dgrad_class<RT, KS, 1, 1, true>(p, afrag, ebuf, w, nwg, c1_0);
}

// (Tried and removed: the B operand staged through LDS -- rows fetched line by line, 8 lanes per 128-byte line, XOR-swizzled
//  chunks, 16 waves sharing the A fragments -- to spare the address path the one-lane-per-line loads above: 82 us against 78,
//  bit-identical results.  What bounds this kernel is the latency of a wave's serial tile chain (a tile is 16-64 MFMAs, its
//  ReLU-mask rows come from HBM every time), not the load instructions; profiles/r02_conv2_dgrad_bench.json.)
// The launch's split into parity classes (host only): wg0[c..c+1) = the workgroups of class c = 2*(t1&1) + (f1&1), tiles[c] =
// its 256-pixel tiles.  Shared by otr_conv2_dgrad and otr_debug_conv2_dgrad_plan (tests/test_cabi.py replays the kernel's pixel
// and tap arithmetic on it).
static bool conv2_dgrad_sliced(const otr_conv_desc_t* d) { return d->C2 == 256 && d->C1 % 64 == 0 && d->C1 >= 128 && d->C1 <= 256; }
static void conv2_dgrad_plan_g(const otr_conv_desc_t* d, int G, int fix, int* wg0, int* tiles);
static void conv2_dgrad_plan(const otr_conv_desc_t* d, int* wg0, int* tiles) {
  conv2_dgrad_plan_g(d, conv2_dgrad_sliced(d) ? 256 / (d->C1 / 64) / 8 * 8 : 512, 37, wg0, tiles);
}
// G workgroups over the four classes; a tile costs `fix` + 10 per tap (units of 0.1 tap)
static void conv2_dgrad_plan_g(const otr_conv_desc_t* d, int G, int fix, int* wg0, int* tiles) {
  // Workgroups per class: a tile is modelled as a fixed part (mask rows, the epilogue's LDS round trips, the header wait) plus
  // one part per tap, 3.7 : 1.  Every class gets one workgroup, the rest go one by one to the class whose workgroups
  // currently run longest (exact for this min-max problem).  (Splitting by pixels x taps instead gave the same 75 us at the
  // AISHELL shape: the launch is not bound by the balance between the classes -- profiles/r02_conv2_dgrad_pmc.txt.)
  const int TILE = 256;
  int n[4], taps[4];
  for (int c = 0; c < 4; ++c) {
    const int pt = c >> 1, pf = c & 1;
    const int64_t nT = pt ? d->T1 / 2 : (d->T1 + 1) / 2, nF = pf ? d->F1 / 2 : (d->F1 + 1) / 2;
    const int64_t Mc = (int64_t)d->B * nT * nF;
    tiles[c] = (int)((Mc + TILE - 1) / TILE);
    taps[c] = (pt ? 1 : 2) * (pf ? 2 : 1);
    n[c] = tiles[c] > 0 ? 1 : 0;
  }
  auto span = [&](int c) {                       // time of the class's longest workgroup, in units of 0.1 tap
    return n[c] > 0 ? (int64_t)((tiles[c] + n[c] - 1) / n[c]) * (fix + 10 * taps[c]) : 0;
  };
  for (int left = G - (n[0] + n[1] + n[2] + n[3]); left > 0; --left) {
    int best = -1;
    for (int c = 0; c < 4; ++c)
      if (n[c] > 0 && n[c] < tiles[c] && (best < 0 || span(c) > span(best))) best = c;
    if (best < 0) break;
    ++n[best];
  }
  wg0[0] = 0;
  for (int c = 0; c < 4; ++c) wg0[c + 1] = wg0[c] + n[c];
}
static bool conv2_dgrad_serves(const otr_conv_desc_t* d) {
  const bool big = d->C1 == 64 && d->C2 == 128, small = d->C1 == 32 && d->C2 == 64;
  return d->act_dtype == OTR_H16 && d->w_dtype == OTR_H16 && (big || small || (conv2_dgrad_sliced(d) && g_otr_conv2_dgrad_wide));
}
// out: {served (0 / 1), wg0[0..4], tiles[0..3]}
extern "C" int32_t otr_debug_conv2_dgrad_plan(const otr_conv_desc_t* d, int32_t* out) {
  ConvArgs chk{};
  if (int32_t e = conv_check(d, chk)) return e;
  OTR_REQUIRE(out != nullptr, "debug_conv2_dgrad_plan: null pointer");
  int wg0[5], tiles[4];
  conv2_dgrad_plan(d, wg0, tiles);
  out[0] = conv2_dgrad_serves(d) ? 1 : 0;
  for (int c = 0; c < 5; ++c) out[1 + c] = wg0[c];
  for (int c = 0; c < 4; ++c) out[6 + c] = tiles[c];
  return 0;
}

// 0 = launched, 1 = shape not served (the caller uses otr_conv2_dgrad_cols + otr_conv2_col2im), < 0 = bad argument
extern "C" int32_t otr_conv2_dgrad(const otr_conv_desc_t* d, const void* dact2, const void* w2r, const void* act1, void* dact1,
                                   void* stream) {
  ConvArgs chk{};
  if (int32_t e = conv_check(d, chk)) return e;
  OTR_REQUIRE(dact2 && w2r && act1 && dact1, "conv2_dgrad: null pointer");
  const bool big = d->C1 == 64 && d->C2 == 128;
  if (!conv2_dgrad_serves(d)) return 1;
  if (((uintptr_t)dact2 | (uintptr_t)act1 | (uintptr_t)dact1) % 16 != 0) return 1;
  OTR_REQUIRE((int64_t)d->B * d->T2 * d->F2 * d->C2 < (1ll << 31), "conv2_dgrad: act2 too large for 32-bit pixel index");
  Conv2DgArgs a{};
  a.g2 = (const uint16_t*)dact2; a.w2r = (const uint16_t*)w2r; a.act1 = (const uint16_t*)act1; a.dact1 = (uint16_t*)dact1;
  a.B = d->B; a.T1 = d->T1; a.F1 = d->F1; a.T2 = d->T2; a.F2 = d->F2;
  a.trace = g_otr_trace;
  a.ablate = g_otr_conv2_dgrad_ablate;
  int tiles[4];
  conv2_dgrad_plan(d, a.wg0, tiles);
  if (a.wg0[4] == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  if (conv2_dgrad_sliced(d)) {
    a.C1full = d->C1; a.nslice = d->C1 / 64;
    hipLaunchKernelGGL((conv2_dgrad_sliced_kernel<2, 16>), dim3((unsigned)(a.wg0[4] * a.nslice)), dim3(512), 0, s, a);
  } else if (big) hipLaunchKernelGGL((conv2_dgrad_kernel<2, 8>), dim3((unsigned)a.wg0[4]), dim3(512), 0, s, a);
  else hipLaunchKernelGGL((conv2_dgrad_kernel<1, 4>), dim3((unsigned)a.wg0[4]), dim3(512), 0, s, a);
  return otr_check_launch("conv2_dgrad");
}

// ------------------------------------------------------------------------------------------------ wide frontends (conv2wide.hip)
int64_t conv2wide_workspace_bytes();
int32_t conv2wide_dgrad(const void* g2, const void* w2r, const void* act1, void* dact1, int B, int T1, int F1, int T2, int F2, const int* wg0,
                        void* scratch, hipStream_t s);
int g_otr_conv2_wide = 1;                 // otr_debug_set(31, 0): the entry below answers "not served"
static bool conv2wide_serves(const otr_conv_desc_t* d, const void* a, const void* b, const void* c, const void* e, const void* scratch,
                             int64_t scratch_bytes) {
  return g_otr_conv2_wide && d->C1 == 256 && d->C2 == 256 && d->act_dtype == OTR_H16 && d->w_dtype == OTR_H16 && d->compute == OTR_H16 &&
         scratch && scratch_bytes >= conv2wide_workspace_bytes() && ((uintptr_t)a | (uintptr_t)b | (uintptr_t)c | (uintptr_t)e | (uintptr_t)scratch) % 16 == 0;
}
extern "C" int64_t otr_conv2_wide_scratch_bytes(void) { return conv2wide_workspace_bytes(); }
// 0 = launched, 1 = not served (otr_conv2_dgrad, or otr_conv2_dgrad_cols + otr_conv2_col2im), < 0 = bad argument
extern "C" int32_t otr_conv2_dgrad_wide(const otr_conv_desc_t* d, const void* dact2, const void* w2r, const void* act1, void* dact1,
                                        void* scratch, int64_t scratch_bytes, void* stream) {
  ConvArgs chk{};
  if (int32_t e = conv_check(d, chk)) return e;
  OTR_REQUIRE(dact2 && w2r && act1 && dact1, "conv2_dgrad_wide: null pointer");
  if (!conv2wide_serves(d, dact2, w2r, act1, dact1, scratch, scratch_bytes)) return 1;
  int wg0[5], tiles[4];
  conv2_dgrad_plan_g(d, 256, 10, wg0, tiles);
  return conv2wide_dgrad(dact2, w2r, act1, dact1, d->B, d->T1, d->F1, d->T2, d->F2, wg0, scratch, (hipStream_t)stream);
}
