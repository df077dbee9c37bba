// Conformer-only kernels (BASELINE.json configs[3]):
//  * residual_add        x + scale * dropout(branch)                    encoder/conformer.py:50-73
//  * head_bias_add       q + pos_bias_u | q + pos_bias_v                module/attention.py:241-245
//  * add2_strided        dq = d(q+u) + d(q+v) into the packed qkv gradient
//  * dwconv / BatchNorm(batch statistics) / swish of ConformerConvolutionModule, fwd + bwd
//                                                                        module/conformer.py:36-57
//  * row_mask            masked_fill_(~mask, 0)                          module/conformer.py:46,55
// All are bandwidth-bound [B*T, C] passes: a thread owns 4 consecutive channels of one frame, and the
// per-channel reductions (BatchNorm statistics, depthwise-conv weight gradients) are accumulated per
// thread over a strip of rows and merged with one fp32 atomic per (workgroup, channel).
#include "common.h"

#include <algorithm>


template <class T> __device__ __forceinline__ void ldc4(const T* p, float* o) { load_row<T, 4>(p, 4, true, o); }
template <class T> __device__ __forceinline__ void stc4(T* p, const float* o) {
  if constexpr (sizeof(T) == 4) *reinterpret_cast<float4*>(p) = make_float4(o[0], o[1], o[2], o[3]);
  else *reinterpret_cast<uint2*>(p) = make_uint2(pack2bf(o[0], o[1]), pack2bf(o[2], o[3]));
}
__device__ __forceinline__ float sigm(float x) { return 1.f / (1.f + expf(-x)); }

// ------------------------------------------------------------------------------------------------ residual add
template <class AT> __global__ void residual_add_fwd_kernel(const float* x, const AT* a, float* y, int64_t n4, float scale,
                                                            float p_drop, const uint64_t* seed, uint64_t off) {
  const bool drop = p_drop > 0.f;
  const uint64_t sd = drop ? *seed : 0;
  const uint32_t thr = drop ? (uint32_t)fminf(p_drop * 4294967296.f, 4294967295.f) : 0;
  const float keep = drop ? 1.f / (1.f - p_drop) : 1.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    float xv[4], av[4];
    ldc4<float>(x + i * 4, xv);
    ldc4<AT>(a + i * 4, av);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float m = drop ? (otr_rand32(sd, off + (uint64_t)(i * 4 + e)) >= thr ? keep : 0.f) : 1.f;
      xv[e] += scale * m * av[e];
    }
    stc4<float>(y + i * 4, xv);
  }
}
template <class AT> __global__ void residual_add_bwd_kernel(const float* dy, AT* da, int64_t n4, float scale, float p_drop,
                                                            const uint64_t* seed, uint64_t off) {
  const bool drop = p_drop > 0.f;
  const uint64_t sd = drop ? *seed : 0;
  const uint32_t thr = drop ? (uint32_t)fminf(p_drop * 4294967296.f, 4294967295.f) : 0;
  const float keep = drop ? 1.f / (1.f - p_drop) : 1.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    float g[4];
    ldc4<float>(dy + i * 4, g);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float m = drop ? (otr_rand32(sd, off + (uint64_t)(i * 4 + e)) >= thr ? keep : 0.f) : 1.f;
      g[e] *= scale * m;
    }
    stc4<AT>(da + i * 4, g);
  }
}
static unsigned ew_grid(int64_t n) { int64_t g = (n + 255) / 256; return (unsigned)(g < 1 ? 1 : (g > 4096 ? 4096 : g)); }

extern "C" int32_t otr_residual_add_fwd(const float* x, const void* a, int32_t a_dtype, float* y, int64_t n, float scale,
                                        float p_drop, const uint64_t* seed, uint64_t rng_offset, void* stream) {
  OTR_REQUIRE(x && a && y, "residual_add_fwd: null pointer");
  OTR_REQUIRE(n % 4 == 0 && n >= 0, "residual_add_fwd: n must be a multiple of 4");
  OTR_REQUIRE(p_drop == 0.f || seed, "residual_add_fwd: dropout needs a seed");
  if (n == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  if (a_dtype == OTR_F32) hipLaunchKernelGGL(residual_add_fwd_kernel<float>, dim3(ew_grid(n / 4)), dim3(256), 0, s, x, (const float*)a, y, n / 4, scale, p_drop, seed, rng_offset);
  else hipLaunchKernelGGL(residual_add_fwd_kernel<bf16_t>, dim3(ew_grid(n / 4)), dim3(256), 0, s, x, (const bf16_t*)a, y, n / 4, scale, p_drop, seed, rng_offset);
  return otr_check_launch("residual_add_fwd");
}
extern "C" int32_t otr_residual_add_bwd(const float* dy, void* da, int32_t a_dtype, int64_t n, float scale, float p_drop,
                                        const uint64_t* seed, uint64_t rng_offset, void* stream) {
  OTR_REQUIRE(dy && da, "residual_add_bwd: null pointer");
  OTR_REQUIRE(n % 4 == 0 && n >= 0, "residual_add_bwd: n must be a multiple of 4");
  if (n == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  if (a_dtype == OTR_F32) hipLaunchKernelGGL(residual_add_bwd_kernel<float>, dim3(ew_grid(n / 4)), dim3(256), 0, s, dy, (float*)da, n / 4, scale, p_drop, seed, rng_offset);
  else hipLaunchKernelGGL(residual_add_bwd_kernel<bf16_t>, dim3(ew_grid(n / 4)), dim3(256), 0, s, dy, (bf16_t*)da, n / 4, scale, p_drop, seed, rng_offset);
  return otr_check_launch("residual_add_bwd");
}

// ------------------------------------------------------------------------------------------------ stand-alone dropout
// y = dropout(x) = x * mask / (1 - p) with the library's counter RNG (mask regenerated, never stored): nn.Dropout on the
// projected attention context (module/attention.py:46), on the FFN hidden (module/ffn.py:40), behind the frontend's
// Conv2dLayers (frontend/conv.py:66) and at the end of the Conformer convolution module.  The backward pass is the same
// map applied to dy, so one kernel serves both.
template <class T> __global__ void dropout_kernel(const T* x, T* y, int64_t n4, float p_drop, const uint64_t* seed, uint64_t off) {
  const uint64_t sd = *seed;
  const uint32_t thr = (uint32_t)fminf(p_drop * 4294967296.f, 4294967295.f);
  const float keep = 1.f / (1.f - p_drop);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    float v[4];
    ldc4<T>(x + i * 4, v);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] *= otr_rand32(sd, off + (uint64_t)(i * 4 + e)) >= thr ? keep : 0.f;
    stc4<T>(y + i * 4, v);
  }
}
extern "C" int32_t otr_dropout(const void* x, void* y, int32_t dtype, int64_t n, float p_drop, const uint64_t* seed,
                               uint64_t rng_offset, void* stream) {
  OTR_REQUIRE(x && y && seed, "dropout: null pointer");
  OTR_REQUIRE(dtype == OTR_F32 || dtype == OTR_H16, "dropout: bad dtype");
  OTR_REQUIRE(n % 4 == 0 && n >= 0, "dropout: n must be a multiple of 4");
  OTR_REQUIRE(p_drop > 0.f && p_drop < 1.f, "dropout: p_drop=%f out of (0,1)", (double)p_drop);
  if (n == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == OTR_F32) hipLaunchKernelGGL(dropout_kernel<float>, dim3(ew_grid(n / 4)), dim3(256), 0, s, (const float*)x, (float*)y, n / 4, p_drop, seed, rng_offset);
  else hipLaunchKernelGGL(dropout_kernel<bf16_t>, dim3(ew_grid(n / 4)), dim3(256), 0, s, (const bf16_t*)x, (bf16_t*)y, n / 4, p_drop, seed, rng_offset);
  return otr_check_launch("dropout");
}

// ------------------------------------------------------------------------------------------------ q + u | q + v
template <class T> __global__ void head_bias_add_kernel(const T* q, int64_t ldq, const float* u, const float* v, T* out, int64_t M,
                                                        int d) {
  const int d4 = d / 4;
  const int64_t total = M * d4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t row = i / d4;
    int c = (int)(i - row * d4) * 4;
    float qv[4], uv[4], vv[4], o1[4], o2[4];
    ldc4<T>(q + row * ldq + c, qv);
    ldc4<float>(u + c, uv);
    ldc4<float>(v + c, vv);
#pragma unroll
    for (int e = 0; e < 4; ++e) { o1[e] = qv[e] + uv[e]; o2[e] = qv[e] + vv[e]; }
    stc4<T>(out + row * 2 * d + c, o1);
    stc4<T>(out + row * 2 * d + d + c, o2);
  }
}
extern "C" int32_t otr_head_bias_add(const void* q, int64_t ldq, const float* u, const float* v, void* out, int32_t dtype,
                                     int64_t M, int32_t d, void* stream) {
  OTR_REQUIRE(q && u && v && out, "head_bias_add: null pointer");
  OTR_REQUIRE(d % 4 == 0 && ldq % 4 == 0, "head_bias_add: d and ldq must be multiples of 4");
  if (M <= 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == OTR_F32) hipLaunchKernelGGL(head_bias_add_kernel<float>, dim3(ew_grid(M * d / 4)), dim3(256), 0, s, (const float*)q, ldq, u, v, (float*)out, M, d);
  else hipLaunchKernelGGL(head_bias_add_kernel<bf16_t>, dim3(ew_grid(M * d / 4)), dim3(256), 0, s, (const bf16_t*)q, ldq, u, v, (bf16_t*)out, M, d);
  return otr_check_launch("head_bias_add");
}

// out[r, :cols] = a[r, :cols] + b[r, :cols] with independent leading dimensions
template <class T> __global__ void add2_kernel(const T* a, int64_t lda, const T* b, int64_t ldb, T* out, int64_t ldo, int64_t M, int cols) {
  const int c4 = cols / 4;
  const int64_t total = M * c4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t row = i / c4;
    int c = (int)(i - row * c4) * 4;
    float x[4], y[4];
    ldc4<T>(a + row * lda + c, x);
    ldc4<T>(b + row * ldb + c, y);
#pragma unroll
    for (int e = 0; e < 4; ++e) x[e] += y[e];
    stc4<T>(out + row * ldo + c, x);
  }
}
extern "C" int32_t otr_add2_strided(const void* a, int64_t lda, const void* b, int64_t ldb, void* out, int64_t ldo,
                                    int32_t dtype, int64_t M, int32_t cols, void* stream) {
  OTR_REQUIRE(a && b && out, "add2_strided: null pointer");
  OTR_REQUIRE(cols % 4 == 0 && lda % 4 == 0 && ldb % 4 == 0 && ldo % 4 == 0, "add2_strided: sizes must be multiples of 4");
  if (M <= 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == OTR_F32) hipLaunchKernelGGL(add2_kernel<float>, dim3(ew_grid(M * cols / 4)), dim3(256), 0, s, (const float*)a, lda, (const float*)b, ldb, (float*)out, ldo, M, cols);
  else hipLaunchKernelGGL(add2_kernel<bf16_t>, dim3(ew_grid(M * cols / 4)), dim3(256), 0, s, (const bf16_t*)a, lda, (const bf16_t*)b, ldb, (bf16_t*)out, ldo, M, cols);
  return otr_check_launch("add2_strided");
}

// The same with the column sums of a and of b left as per-workgroup partials [blocks][2 cols] (r06: d(q+u) + d(q+v) of the relative-position
// attention is the sum the packed qkv gradient needs, and the column sums of the two addends are the gradients of pos_bias_u / pos_bias_v
// (module/attention.py:241-245): a separate column-sum pass re-read both, 147 MB per step).  A workgroup owns A2_RPB rows; thread
// (row phase, 4 columns) walks its rows, four loads of each operand in flight.
constexpr int A2_RPB = 32;
template <class T> __global__ __launch_bounds__(256) void add2_colsum_kernel(const T* a, int64_t lda, const T* b, int64_t ldb, T* out, int64_t ldo,
                                                                            int64_t M, int cols, float* partial, int NY) {
  extern __shared__ float a2_red[];                       // [NY][2 cols]
  const int c4 = cols / 4, cg = threadIdx.x % c4, ty = threadIdx.x / c4, c = cg * 4;
  const int64_t r0 = (int64_t)blockIdx.x * A2_RPB, r1 = min(M, r0 + A2_RPB);
  float sa[4] = {0.f, 0.f, 0.f, 0.f}, sb[4] = {0.f, 0.f, 0.f, 0.f};
  if (ty < NY) {
    for (int64_t base = r0 + ty; base < r1; base += 4 * NY) {
      float x[4][4], y[4][4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int64_t row = min(base + (int64_t)u * NY, r1 - 1);
        ldc4<T>(a + row * lda + c, x[u]);
        ldc4<T>(b + row * ldb + c, y[u]);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int64_t row = base + (int64_t)u * NY;
        if (row < r1) {
          float o[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) { sa[e] += x[u][e]; sb[e] += y[u][e]; o[e] = x[u][e] + y[u][e]; }
          stc4<T>(out + row * ldo + c, o);
        }
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) { a2_red[ty * 2 * cols + c + e] = sa[e]; a2_red[ty * 2 * cols + cols + c + e] = sb[e]; }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * cols; i += 256) {
    float v = 0.f;
    for (int y = 0; y < NY; ++y) v += a2_red[y * 2 * cols + i];
    partial[(int64_t)blockIdx.x * 2 * cols + i] = v;
  }
}
extern "C" int64_t otr_add2_colsum_partial_rows(int64_t M) { return (M + A2_RPB - 1) / A2_RPB; }
extern "C" int32_t otr_add2_strided_colsum(const void* a, int64_t lda, const void* b, int64_t ldb, void* out, int64_t ldo, int32_t dtype, int64_t M,
                                           int32_t cols, float* partial, void* stream) {
  OTR_REQUIRE(a && b && out && partial, "add2_strided_colsum: null pointer");
  OTR_REQUIRE(cols % 4 == 0 && cols >= 4 && cols <= 1024 && lda % 4 == 0 && ldb % 4 == 0 && ldo % 4 == 0, "add2_strided_colsum: sizes must be multiples of 4, cols <= 1024");
  if (M <= 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  const int NY = std::max(1, 256 / (cols / 4));
  const dim3 grid((unsigned)((M + A2_RPB - 1) / A2_RPB));
  const size_t lds = (size_t)NY * 2 * cols * sizeof(float);
  if (dtype == OTR_F32) hipLaunchKernelGGL(add2_colsum_kernel<float>, grid, dim3(256), lds, s, (const float*)a, lda, (const float*)b, ldb, (float*)out, ldo, M, cols, partial, NY);
  else hipLaunchKernelGGL(add2_colsum_kernel<bf16_t>, grid, dim3(256), lds, s, (const bf16_t*)a, lda, (const bf16_t*)b, ldb, (bf16_t*)out, ldo, M, cols, partial, NY);
  return otr_check_launch("add2_strided_colsum");
}

// ------------------------------------------------------------------------------------------------ row mask
template <class TI, class TO> __global__ void row_mask_kernel(const TI* x, const uint8_t* mask, TO* out, int64_t M, int C) {
  const int c4 = C / 4;
  const int64_t total = M * c4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t row = i / c4;
    float v[4];
    ldc4<TI>(x + i * 4, v);
    if (!mask[row]) { v[0] = 0.f; v[1] = 0.f; v[2] = 0.f; v[3] = 0.f; }
    stc4<TO>(out + i * 4, v);
  }
}
extern "C" int32_t otr_row_mask_cast(const void* x, int32_t x_dtype, const uint8_t* mask, void* out, int32_t out_dtype, int64_t M,
                                     int32_t C, void* stream) {
  OTR_REQUIRE(x && mask && out, "row_mask: null pointer");
  OTR_REQUIRE(C % 4 == 0, "row_mask: C must be a multiple of 4");
  OTR_REQUIRE((x_dtype == OTR_F32 || x_dtype == OTR_H16) && (out_dtype == OTR_F32 || out_dtype == OTR_H16), "row_mask: bad dtype");
  if (M <= 0) return 0;
  const dim3 g(ew_grid(M * C / 4));
  hipStream_t s = (hipStream_t)stream;
  if (x_dtype == OTR_F32 && out_dtype == OTR_F32) hipLaunchKernelGGL((row_mask_kernel<float, float>), g, dim3(256), 0, s, (const float*)x, mask, (float*)out, M, C);
  else if (x_dtype == OTR_F32) hipLaunchKernelGGL((row_mask_kernel<float, bf16_t>), g, dim3(256), 0, s, (const float*)x, mask, (bf16_t*)out, M, C);
  else if (out_dtype == OTR_F32) hipLaunchKernelGGL((row_mask_kernel<bf16_t, float>), g, dim3(256), 0, s, (const bf16_t*)x, mask, (float*)out, M, C);
  else hipLaunchKernelGGL((row_mask_kernel<bf16_t, bf16_t>), g, dim3(256), 0, s, (const bf16_t*)x, mask, (bf16_t*)out, M, C);
  return otr_check_launch("row_mask");
}
extern "C" int32_t otr_row_mask(const float* x, const uint8_t* mask, float* out, int64_t M, int32_t C, void* stream) {
  return otr_row_mask_cast(x, OTR_F32, mask, out, OTR_F32, M, C, stream);
}

// ------------------------------------------------------------------------------------------------ depthwise conv + BN stats
struct DwArgs {
  const void* g; const float* w; const float* b; float* y; float* stats;
  const float* dy; void* dg; float* dw; float* db;
  float* part;           // backward: per-workgroup sums [blocks][C*k | C] instead of atomics on dw / db (the caller column-sums them)
  float* spart;          // forward: per-workgroup BatchNorm sums [blocks][sum y (C) | sum y^2 (C)] instead of atomics on stats
  int B, T, C, k, pad;   // pad = taps left of the output position ((k-1)/2: 'same' conv; 0: look-ahead conv)
};

// Thread layout of the depthwise-conv kernels: 256 threads = DW_NY row phases x (C/4 <= 256/DW_NY) channel groups;
// a workgroup owns DW_RPB consecutive rows, thread (ty, cg) walks rows ty, ty+NY, ...  All loads are unconditional
// (time index clamped into the utterance, contribution masked); the per-channel sums are reduced through LDS so each
// workgroup issues one set of atomics.  (The first version: 16-row strips, 96 active threads, every tap load behind
// its own bounds branch, 498 x 2304 atomics: 271 us for the backward at C=384.)
// r05: a thread walks a SEGMENT of consecutive rows and keeps the taps' rows in registers (a sliding window over the flat row index:
// one load per row and operand instead of KT; neighbours across an utterance boundary are loaded and masked), and a workgroup owns 32
// rows: 249 workgroups at the bench batch instead of 125 walking 32 rows per thread behind 5-11 loads each (forward 30 -> , backward
// 46 -> us at 7968 x 384, rocprofv3).
constexpr int DW_RPB = 32;
// KT = compile-time tap count (3 / 5 / 7: the smallest >= k), so only real taps are loaded

// y[b,t,c] = bias[c] + sum_j w[c,j] * g[b, t + j - pad, c]   (zero padded in time, per utterance)
// stats[c] += sum y, stats[C + c] += sum y^2   over ALL B*T positions (the reference's BatchNorm1d sees padded frames)
template <class T, int KT> __global__ __launch_bounds__(256) void dwconv_fwd_kernel(DwArgs p, int NY) {
  extern __shared__ float dw_red[];                       // [NY][C][2]
  const int C4 = p.C / 4, pad = p.pad;
  const int64_t M = (int64_t)p.B * p.T;
  const int64_t r0 = (int64_t)blockIdx.x * DW_RPB, r1 = min(M, r0 + DW_RPB);
  const T* g = reinterpret_cast<const T*>(p.g);
  const int cg = threadIdx.x % C4, ty = threadIdx.x / C4;
  const bool active = ty < NY;
  const int c = cg * 4, seg = (DW_RPB + NY - 1) / NY;
  float w[4][KT], bias[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    if (p.b) bias[e] = p.b[c + e];
#pragma unroll
    for (int j = 0; j < KT; ++j) w[e][j] = (j < p.k) ? p.w[(c + e) * p.k + j] : 0.f;
  }
  float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
  const int64_t rs = r0 + (int64_t)ty * seg, re = min(r1, rs + seg);
  if (active && rs < re) {
    auto flat = [&](int64_t r) { return min(max(r, (int64_t)0), M - 1); };
    int t = (int)(rs % p.T);
    float gw[KT][4];                                      // gw[j] = g[row + j - pad] (flat rows; taps outside the utterance are masked)
#pragma unroll
    for (int j = 0; j < KT - 1; ++j) ldc4<T>(g + flat(rs + j - pad) * p.C + c, gw[j]);
    for (int64_t row = rs; row < re; ++row) {
      ldc4<T>(g + flat(row + KT - 1 - pad) * p.C + c, gw[KT - 1]);
      float acc[4] = {bias[0], bias[1], bias[2], bias[3]};
#pragma unroll
      for (int j = 0; j < KT; ++j) {
        const int tt = t + j - pad;
        const float m = (j < p.k && tt >= 0 && tt < p.T) ? 1.f : 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] = fmaf(w[e][j] * m, gw[j][e], acc[e]);
      }
      stc4<float>(p.y + row * p.C + c, acc);
#pragma unroll
      for (int e = 0; e < 4; ++e) { s1[e] += acc[e]; s2[e] += acc[e] * acc[e]; }
#pragma unroll
      for (int j = 0; j < KT - 1; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) gw[j][e] = gw[j + 1][e];
      if (++t == p.T) t = 0;
    }
  }
  if (p.stats || p.spart) {
    if (active) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { dw_red[(ty * p.C + c + e) * 2] = s1[e]; dw_red[(ty * p.C + c + e) * 2 + 1] = s2[e]; }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * p.C; i += 256) {
      const int ch = i >> 1, which = i & 1;
      float a = 0.f;
      for (int y = 0; y < NY; ++y) a += dw_red[(y * p.C + ch) * 2 + which];
      if (p.spart) p.spart[(int64_t)blockIdx.x * 2 * p.C + which * p.C + ch] = a;
      else atomicAdd(p.stats + which * p.C + ch, a);
    }
  }
}

// dg[b,t,c] = sum_j w[c,j] * dy[b, t - j + pad, c];  dw[c,j] += sum dy[b,t,c] g[b,t+j-pad,c];  db[c] += sum dy
template <class T, int KT> __global__ __launch_bounds__(256) void dwconv_bwd_kernel(DwArgs p, int NY) {
  extern __shared__ float dw_red[];                       // [NY][C][k+1]
  const int C4 = p.C / 4, pad = p.pad, K1 = p.k + 1;
  const int64_t M = (int64_t)p.B * p.T;
  const int64_t r0 = (int64_t)blockIdx.x * DW_RPB, r1 = min(M, r0 + DW_RPB);
  const T* g = reinterpret_cast<const T*>(p.g);
  T* dg = reinterpret_cast<T*>(p.dg);
  const int cg = threadIdx.x % C4, ty = threadIdx.x / C4;
  const bool active = ty < NY;
  const int c = cg * 4, seg = (DW_RPB + NY - 1) / NY;
  float w[4][KT], dw[4][KT], db[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int e = 0; e < 4; ++e)
#pragma unroll
    for (int j = 0; j < KT; ++j) { w[e][j] = (j < p.k) ? p.w[(c + e) * p.k + j] : 0.f; dw[e][j] = 0.f; }
  const int64_t rs = r0 + (int64_t)ty * seg, re = min(r1, rs + seg);
  if (active) {
    if (rs < re) {
      auto flat = [&](int64_t r) { return min(max(r, (int64_t)0), M - 1); };
      int t = (int)(rs % p.T);
      float gw[KT][4], yw[KT][4];                         // gw[j] = g[row + j - pad], yw[j] = dy[row - j + pad] (flat rows, masked by t)
#pragma unroll
      for (int j = 0; j < KT - 1; ++j) ldc4<T>(g + flat(rs + j - pad) * p.C + c, gw[j]);
#pragma unroll
      for (int j = 1; j < KT; ++j) ldc4<float>(p.dy + flat(rs - j + pad) * p.C + c, yw[j]);
      for (int64_t row = rs; row < re; ++row) {
        ldc4<T>(g + flat(row + KT - 1 - pad) * p.C + c, gw[KT - 1]);
        ldc4<float>(p.dy + flat(row + pad) * p.C + c, yw[0]);
        float acc[4] = {0.f, 0.f, 0.f, 0.f}, dyv[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) dyv[e] = 0.f;
#pragma unroll
        for (int j = 0; j < KT; ++j)
          if (j == pad) {                                  // dy[row] sits in the window (pad < KT)
#pragma unroll
            for (int e = 0; e < 4; ++e) dyv[e] = yw[j][e];
          }
#pragma unroll
        for (int e = 0; e < 4; ++e) db[e] += dyv[e];
#pragma unroll
        for (int j = 0; j < KT; ++j) {
          const int tg = t + j - pad, tyy = t - j + pad;
          const float mg = (j < p.k && tg >= 0 && tg < p.T) ? 1.f : 0.f;       // forward tap: y[t] used g[t + j - pad]
          const float my = (j < p.k && tyy >= 0 && tyy < p.T) ? 1.f : 0.f;     // dg[t] collects dy[t - j + pad] * w[j]
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            dw[e][j] = fmaf(dyv[e] * mg, gw[j][e], dw[e][j]);
            acc[e] = fmaf(w[e][j] * my, yw[j][e], acc[e]);
          }
        }
        stc4<T>(dg + row * p.C + c, acc);
#pragma unroll
        for (int j = 0; j < KT - 1; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e) gw[j][e] = gw[j + 1][e];
#pragma unroll
        for (int j = KT - 1; j > 0; --j)
#pragma unroll
          for (int e = 0; e < 4; ++e) yw[j][e] = yw[j - 1][e];
        if (++t == p.T) t = 0;
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
#pragma unroll
      for (int j = 0; j < KT; ++j)
        if (j < p.k) dw_red[(ty * p.C + c + e) * K1 + j] = dw[e][j];
      dw_red[(ty * p.C + c + e) * K1 + p.k] = db[e];
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < p.C * K1; i += 256) {
    float a = 0.f;
    for (int y = 0; y < NY; ++y) a += dw_red[y * p.C * K1 + i];
    const int ch = i / K1, j = i - ch * K1;
    if (p.part) p.part[(int64_t)blockIdx.x * (p.C * K1) + (j < p.k ? ch * p.k + j : p.C * p.k + ch)] = a;
    else if (j < p.k) atomicAdd(p.dw + ch * p.k + j, a);
    else if (p.db) atomicAdd(p.db + ch, a);
  }
}

static int32_t dw_check(int B, int T, int C, int k, int pad) {
  OTR_REQUIRE(B > 0 && T > 0 && C > 0 && C % 4 == 0 && C <= 1024, "dwconv: bad shape B=%d T=%d C=%d (C % 4 == 0, C <= 1024)", B, T, C);
  OTR_REQUIRE(k >= 1 && k <= 7, "dwconv: kernel size %d must be in 1..7", k);
  OTR_REQUIRE(pad >= 0 && pad < k, "dwconv: pad %d must be in [0, k)", pad);
  return 0;
}
extern "C" int64_t otr_dwconv_fwd_partial_rows(int64_t M) { return (M + DW_RPB - 1) / DW_RPB; }

static int32_t dwconv_fwd_launch(const void* g, int32_t dtype, const float* w, const float* bias, float* y, float* stats, float* spart,
                                 int32_t B, int32_t T, int32_t C, int32_t k, int32_t pad, void* stream);

extern "C" int32_t otr_dwconv_fwd(const void* g, int32_t dtype, const float* w, const float* bias, float* y, float* stats,
                                  int32_t B, int32_t T, int32_t C, int32_t k, int32_t pad, void* stream) {
  return dwconv_fwd_launch(g, dtype, w, bias, y, stats, nullptr, B, T, C, k, pad, stream);
}

extern "C" int32_t otr_dwconv_fwd_part(const void* g, int32_t dtype, const float* w, const float* bias, float* y, float* spart,
                                       int32_t B, int32_t T, int32_t C, int32_t k, int32_t pad, void* stream) {
  OTR_REQUIRE(spart, "dwconv_fwd_part: null partial buffer");
  return dwconv_fwd_launch(g, dtype, w, bias, y, nullptr, spart, B, T, C, k, pad, stream);
}

static int32_t dwconv_fwd_launch(const void* g, int32_t dtype, const float* w, const float* bias, float* y, float* stats, float* spart,
                                 int32_t B, int32_t T, int32_t C, int32_t k, int32_t pad, void* stream) {
  if (int32_t e = dw_check(B, T, C, k, pad)) return e;
  OTR_REQUIRE(g && w && y, "dwconv_fwd: null pointer");
  DwArgs p{}; p.g = g; p.w = w; p.b = bias; p.y = y; p.stats = stats; p.spart = spart; p.B = B; p.T = T; p.C = C; p.k = k; p.pad = pad;
  hipStream_t s = (hipStream_t)stream;
  if (stats) otr_zero_f32(stats, 2 * C, s);
  const int NY = 256 / (C / 4);
  const dim3 grid((unsigned)(((int64_t)B * T + DW_RPB - 1) / DW_RPB));
  const size_t lds = (size_t)NY * C * 2 * sizeof(float);
#define DW_LAUNCH(KERNEL, KT)                                                                              \
  {                                                                                                         \
    if (dtype == OTR_F32) hipLaunchKernelGGL((KERNEL<float, KT>), grid, dim3(256), lds, s, p, NY);          \
    else hipLaunchKernelGGL((KERNEL<bf16_t, KT>), grid, dim3(256), lds, s, p, NY);                          \
  }
  if (k <= 3) DW_LAUNCH(dwconv_fwd_kernel, 3) else if (k <= 5) DW_LAUNCH(dwconv_fwd_kernel, 5) else DW_LAUNCH(dwconv_fwd_kernel, 7)
  return otr_check_launch("dwconv_fwd");
}
extern "C" int64_t otr_dwconv_bwd_partial_rows(int64_t M) { return (M + DW_RPB - 1) / DW_RPB; }

static int32_t dwconv_bwd_launch(const float* dy, const void* g, int32_t dtype, const float* w, void* dg, float* dw, float* db, float* part,
                                 int32_t B, int32_t T, int32_t C, int32_t k, int32_t pad, void* stream);

extern "C" int32_t otr_dwconv_bwd_part(const float* dy, const void* g, int32_t dtype, const float* w, void* dg, float* part,
                                       int32_t B, int32_t T, int32_t C, int32_t k, int32_t pad, void* stream) {
  OTR_REQUIRE(part && (uintptr_t)part % 16 == 0, "dwconv_bwd_part: null / unaligned partial buffer");
  return dwconv_bwd_launch(dy, g, dtype, w, dg, nullptr, nullptr, part, B, T, C, k, pad, stream);
}

extern "C" int32_t otr_dwconv_bwd(const float* dy, const void* g, int32_t dtype, const float* w, void* dg, float* dw, float* db,
                                  int32_t B, int32_t T, int32_t C, int32_t k, int32_t pad, void* stream) {
  OTR_REQUIRE(dw, "dwconv_bwd: null pointer");
  return dwconv_bwd_launch(dy, g, dtype, w, dg, dw, db, nullptr, B, T, C, k, pad, stream);
}

static int32_t dwconv_bwd_launch(const float* dy, const void* g, int32_t dtype, const float* w, void* dg, float* dw, float* db, float* part,
                                 int32_t B, int32_t T, int32_t C, int32_t k, int32_t pad, void* stream) {
  if (int32_t e = dw_check(B, T, C, k, pad)) return e;
  OTR_REQUIRE(dy && g && w && dg, "dwconv_bwd: null pointer");
  DwArgs p{}; p.dy = dy; p.g = g; p.w = w; p.dg = dg; p.dw = dw; p.db = db; p.part = part; p.B = B; p.T = T; p.C = C; p.k = k; p.pad = pad;
  hipStream_t s = (hipStream_t)stream;
  const int NY = 256 / (C / 4);
  const dim3 grid((unsigned)(((int64_t)B * T + DW_RPB - 1) / DW_RPB));
  const size_t lds = (size_t)NY * C * (k + 1) * sizeof(float);
  if (k <= 3) DW_LAUNCH(dwconv_bwd_kernel, 3) else if (k <= 5) DW_LAUNCH(dwconv_bwd_kernel, 5) else DW_LAUNCH(dwconv_bwd_kernel, 7)
#undef DW_LAUNCH
  return otr_check_launch("dwconv_bwd");
}

// ------------------------------------------------------------------------------------------------ fused middle of the module's backward
// r06: BatchNorm backward (apply step) + depthwise-conv backward + GLU backward of ConformerConvolutionModule (module/conformer.py:36-57)
// in ONE launch.  dwconv_bwd_kernel's layout (a thread owns 4 channels and walks a segment of rows with the taps' rows in registers),
// with both ends opened up:
//   * the dy window is not loaded but COMPUTED from y (the saved conv output) and ds (the gradient behind the swish), exactly as
//     bn_swish_bwd_kernel<MODE 1> does, from the reduced sums `red` -- the [M, C] fp32 dy tensor is never written or read
//     (the KT - 1 halo rows of a segment are computed twice);
//   * the finished dg row goes straight through the GLU's backward with the row of h (value | gate) and leaves as the row of dh [M, 2C];
//     the column sums of dh (the bias gradient of pointwise_conv1) are left as per-workgroup partials like the dw / db sums.
// The three launches it replaces took 7.6 + 23.5 + 10.7 us per Conformer block at the bench batch.
struct CmArgs {
  const float* y; const void* ds; const float* saved; const float* gamma; const float* beta; const float* red;
  const void* g; const float* w; const void* h; const uint8_t* mask; void* dh;
  float* part;           // [blocks][C*k | C]: dw, db sums of the block's rows
  float* gpart;          // [blocks][2C]: column sums of dh
  float n;
  int training, B, T, C, k, pad;
};
__device__ __forceinline__ float cm_sigmoid_fast(float x) { return 1.f / (1.f + __expf(-x)); }     // (the GLU kernels' form, elementwise.hip)
template <class T, int KT> __global__ __launch_bounds__(256) void conv_mid_bwd_kernel(CmArgs p, int NY) {
  extern __shared__ float dw_red[];                       // [NY][C][k + 3]: dw taps, db, dh value sum, dh gate sum
  const int C4 = p.C / 4, pad = p.pad, K3 = p.k + 3;
  const int64_t M = (int64_t)p.B * p.T;
  const int64_t r0 = (int64_t)blockIdx.x * DW_RPB, r1 = min(M, r0 + DW_RPB);
  const T* g = reinterpret_cast<const T*>(p.g);
  const T* ds = reinterpret_cast<const T*>(p.ds);
  const T* h = reinterpret_cast<const T*>(p.h);
  T* dh = reinterpret_cast<T*>(p.dh);
  const int cg = threadIdx.x % C4, ty = threadIdx.x / C4;
  const bool active = ty < NY;
  const int c = cg * 4, seg = (DW_RPB + NY - 1) / NY;
  float w[4][KT], dw[4][KT], db[4] = {0.f, 0.f, 0.f, 0.f}, sa[4] = {0.f, 0.f, 0.f, 0.f}, sg[4] = {0.f, 0.f, 0.f, 0.f};
  float mean[4], rstd[4], gam[4], bet[4], a0[4], a1[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
#pragma unroll
    for (int j = 0; j < KT; ++j) { w[e][j] = (j < p.k) ? p.w[(c + e) * p.k + j] : 0.f; dw[e][j] = 0.f; }
    mean[e] = p.saved[c + e]; rstd[e] = p.saved[p.C + c + e]; gam[e] = p.gamma[c + e]; bet[e] = p.beta[c + e];
    a0[e] = p.red[c + e] / p.n; a1[e] = p.red[p.C + c + e] / p.n;
  }
  const int64_t rs = r0 + (int64_t)ty * seg, re = min(r1, rs + seg);
  if (active) {
    if (rs < re) {
      auto flat = [&](int64_t r) { return min(max(r, (int64_t)0), M - 1); };
      // dy of one (flat) row: bn_swish_bwd_kernel<MODE 1>
      auto dy_row = [&](int64_t r, float (&o)[4]) {
        float yv[4], dsv[4];
        ldc4<float>(p.y + r * p.C + c, yv);
        ldc4<T>(ds + r * p.C + c, dsv);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float xh = (yv[e] - mean[e]) * rstd[e], z = xh * gam[e] + bet[e], s = sigm(z);
          const float dz = dsv[e] * (s + z * s * (1.f - s));
          o[e] = p.training ? gam[e] * rstd[e] * (dz - a0[e] - xh * a1[e]) : gam[e] * rstd[e] * dz;
        }
      };
      int t = (int)(rs % p.T);
      float gw[KT][4], yw[KT][4];                         // gw[j] = g[row + j - pad], yw[j] = dy[row - j + pad] (flat rows, masked by t)
#pragma unroll
      for (int j = 0; j < KT - 1; ++j) ldc4<T>(g + flat(rs + j - pad) * p.C + c, gw[j]);
#pragma unroll
      for (int j = 1; j < KT; ++j) dy_row(flat(rs - j + pad), yw[j]);
      for (int64_t row = rs; row < re; ++row) {
        ldc4<T>(g + flat(row + KT - 1 - pad) * p.C + c, gw[KT - 1]);
        dy_row(flat(row + pad), yw[0]);
        float ha[4], hb[4];                                // this row of h: its loads fly under the window arithmetic
        ldc4<T>(h + row * 2 * p.C + c, ha);
        ldc4<T>(h + row * 2 * p.C + p.C + c, hb);
        const bool keep = !p.mask || p.mask[row];
        float acc[4] = {0.f, 0.f, 0.f, 0.f}, dyv[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) dyv[e] = 0.f;
#pragma unroll
        for (int j = 0; j < KT; ++j)
          if (j == pad) {                                  // dy[row] sits in the window (pad < KT)
#pragma unroll
            for (int e = 0; e < 4; ++e) dyv[e] = yw[j][e];
          }
#pragma unroll
        for (int e = 0; e < 4; ++e) db[e] += dyv[e];
#pragma unroll
        for (int j = 0; j < KT; ++j) {
          const int tg = t + j - pad, tyy = t - j + pad;
          const float mg = (j < p.k && tg >= 0 && tg < p.T) ? 1.f : 0.f;       // forward tap: y[t] used g[t + j - pad]
          const float my = (j < p.k && tyy >= 0 && tyy < p.T) ? 1.f : 0.f;     // dg[t] collects dy[t - j + pad] * w[j]
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            dw[e][j] = fmaf(dyv[e] * mg, gw[j][e], dw[e][j]);
            acc[e] = fmaf(w[e][j] * my, yw[j][e], acc[e]);
          }
        }
        // GLU backward of the row (glu_bwd_kernel): dg is rounded to the activation type first, as the stand-alone chain stored it
        float oa[4], og[4], dgr[4];
        if constexpr (sizeof(T) == 4) {
#pragma unroll
          for (int e = 0; e < 4; ++e) dgr[e] = acc[e];
        } else {
          const uint32_t w0 = pack2bf(acc[0], acc[1]), w1 = pack2bf(acc[2], acc[3]);
          dgr[0] = h2f_lo(w0); dgr[1] = h2f_hi(w0); dgr[2] = h2f_lo(w1); dgr[3] = h2f_hi(w1);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float sgm = cm_sigmoid_fast(hb[e]), dd = keep ? dgr[e] : 0.f;
          oa[e] = dd * sgm;
          og[e] = dd * ha[e] * sgm * (1.f - sgm);
          sa[e] += oa[e]; sg[e] += og[e];
        }
        stc4<T>(dh + row * 2 * p.C + c, oa);
        stc4<T>(dh + row * 2 * p.C + p.C + c, og);
#pragma unroll
        for (int j = 0; j < KT - 1; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e) gw[j][e] = gw[j + 1][e];
#pragma unroll
        for (int j = KT - 1; j > 0; --j)
#pragma unroll
          for (int e = 0; e < 4; ++e) yw[j][e] = yw[j - 1][e];
        if (++t == p.T) t = 0;
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
#pragma unroll
      for (int j = 0; j < KT; ++j)
        if (j < p.k) dw_red[(ty * p.C + c + e) * K3 + j] = dw[e][j];
      dw_red[(ty * p.C + c + e) * K3 + p.k] = db[e];
      dw_red[(ty * p.C + c + e) * K3 + p.k + 1] = sa[e];
      dw_red[(ty * p.C + c + e) * K3 + p.k + 2] = sg[e];
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < p.C * K3; i += 256) {
    float a = 0.f;
    for (int y = 0; y < NY; ++y) a += dw_red[y * p.C * K3 + i];
    const int ch = i / K3, j = i - ch * K3;
    if (j < p.k) p.part[(int64_t)blockIdx.x * (p.C * (p.k + 1)) + ch * p.k + j] = a;
    else if (j == p.k) p.part[(int64_t)blockIdx.x * (p.C * (p.k + 1)) + p.C * p.k + ch] = a;
    else p.gpart[(int64_t)blockIdx.x * 2 * p.C + (j - p.k - 1) * p.C + ch] = a;
  }
}

// ------------------------------------------------------------------------------------------------ BatchNorm1d + swish
// stats (training): [sum | sumsq] over N rows -> mean / biased var; running stats updated with momentum
// (unbiased var).  eval: running stats.  saved[c] = mean, saved[C + c] = rstd.
__global__ void bn_prepare_kernel(const float* stats, float* run_mean, float* run_var, float* saved, int C, float n, float eps,
                                  float momentum, int training) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float mean, var;
  if (training) {
    mean = stats[c] / n;
    var = fmaxf(stats[C + c] / n - mean * mean, 0.f);
    if (run_mean) {
      run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * mean;
      run_var[c] = (1.f - momentum) * run_var[c] + momentum * var * (n > 1.f ? n / (n - 1.f) : 1.f);
    }
  } else {
    mean = run_mean[c];
    var = run_var[c];
  }
  saved[c] = mean;
  saved[C + c] = rsqrtf(var + eps);
}

// the same from the depthwise convolution's per-workgroup sums spart [nblk][2C] (otr_dwconv_fwd_part: no zeroing launch, no atomics):
// block = 16 channels x 16 row lanes, four independent loads in flight per lane (6 blocks of 64 channels x 4 lanes walked 62 rows
// each, one dependent load after the other: 20 us)
__global__ __launch_bounds__(256) void bn_prepare_part_kernel(const float* spart, int nblk, float* run_mean, float* run_var, float* saved,
                                                             int C, float n, float eps, float momentum) {
  __shared__ float red[2][16][17];
  const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl, cc = min(c, C - 1);
  float s1 = 0.f, s2 = 0.f;
  for (int r0 = rl; r0 < nblk; r0 += 64) {
    float a[4], b[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int r = min(r0 + 16 * u, nblk - 1);
      a[u] = spart[(int64_t)r * 2 * C + cc];
      b[u] = spart[(int64_t)r * 2 * C + C + cc];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (r0 + 16 * u < nblk) { s1 += a[u]; s2 += b[u]; }
  }
  red[0][rl][cl] = s1; red[1][rl][cl] = s2;
  __syncthreads();
  if (rl == 0 && c < C) {
    s1 = 0.f; s2 = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) { s1 += red[0][q][cl]; s2 += red[1][q][cl]; }
    const float mean = s1 / n, var = fmaxf(s2 / n - mean * mean, 0.f);
    if (run_mean) {
      run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * mean;
      run_var[c] = (1.f - momentum) * run_var[c] + momentum * var * (n > 1.f ? n / (n - 1.f) : 1.f);
    }
    saved[c] = mean;
    saved[C + c] = rsqrtf(var + eps);
  }
}

template <class T> __global__ void bn_swish_fwd_kernel(const float* y, const float* saved, const float* gamma, const float* beta, T* out,
                                                       int64_t M, int C) {
  const int c4 = C / 4;
  const int64_t total = M * c4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int c = (int)(i % c4) * 4;
    float v[4], o[4];
    ldc4<float>(y + i * 4, v);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float z = (v[e] - saved[c + e]) * saved[C + c + e] * gamma[c + e] + beta[c + e];
      o[e] = z * sigm(z);
    }
    stc4<T>(out + i * 4, o);
  }
}

// MODE 0: partial[strip][c] = sum dz, partial[strip][C+c] = sum dz*xhat over the strip's rows (no atomics; bn_reduce_kernel
// sums the strips)      MODE 1: dy = gamma*rstd*(dz - red0/N - xhat*red1/N)
// Block = TX channel groups (4 channels each; TX = largest power of two dividing C/4) x 256/TX row lanes on a BN_RPB-row
// strip, four rows per lane in flight.  (First version: 16-row strips walked serially by 96 of 128 threads, then 2C atomics
// per workgroup onto the same 2C addresses: 61 us of which most was the atomics.)
constexpr int BN_RPB = 32;
template <class T, int MODE> __global__ __launch_bounds__(256) void bn_swish_bwd_kernel(const float* y, const T* ds, const float* saved,
                                                                                      const float* gamma, const float* beta, const float* red,
                                                                                      float* partial, float* dy, int64_t M, int C, float n,
                                                                                      int training, int TX) {
  __shared__ float sh[256][9];
  const int tx = threadIdx.x % TX, ty = threadIdx.x / TX, TY = 256 / TX;
  const int c = (blockIdx.x * TX + tx) * 4;
  const int64_t r0 = (int64_t)blockIdx.y * BN_RPB, r1 = min(M, r0 + BN_RPB);
  float mean[4], rstd[4], gam[4], bet[4], a0[4], a1[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    mean[e] = saved[c + e]; rstd[e] = saved[C + c + e]; gam[e] = gamma[c + e]; bet[e] = beta[c + e];
    a0[e] = 0.f; a1[e] = 0.f;
    if (MODE == 1) { a0[e] = red[c + e] / n; a1[e] = red[C + c + e] / n; }
  }
  constexpr int UNR = 4;
  for (int64_t base = r0 + ty; base < r1; base += (int64_t)TY * UNR) {
    float yv[UNR][4], dsv[UNR][4];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int64_t row = min(base + (int64_t)u * TY, r1 - 1);
      ldc4<float>(y + row * C + c, yv[u]);
      ldc4<T>(ds + row * C + c, dsv[u]);
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int64_t row = base + (int64_t)u * TY;
      if (row >= r1) break;
      float o[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float xh = (yv[u][e] - mean[e]) * rstd[e];
        float z = xh * gam[e] + bet[e];
        float sg = sigm(z);
        float dz = dsv[u][e] * (sg + z * sg * (1.f - sg));
        if (MODE == 0) { a0[e] += dz; a1[e] += dz * xh; }
        else o[e] = training ? gam[e] * rstd[e] * (dz - a0[e] - xh * a1[e]) : gam[e] * rstd[e] * dz;
      }
      if (MODE == 1) stc4<float>(dy + row * C + c, o);
    }
  }
  if (MODE == 0) {
    if (TY > 1) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { sh[threadIdx.x][e] = a0[e]; sh[threadIdx.x][4 + e] = a1[e]; }
      __syncthreads();
      if (ty == 0)
        for (int k = 1; k < TY; ++k) {
#pragma unroll
          for (int e = 0; e < 4; ++e) { a0[e] += sh[k * TX + tx][e]; a1[e] += sh[k * TX + tx][4 + e]; }
        }
    }
    if (ty == 0) {
      float* dst = partial + (int64_t)blockIdx.y * 2 * C;
#pragma unroll
      for (int e = 0; e < 4; ++e) { dst[c + e] = a0[e]; dst[C + c + e] = a1[e]; }
    }
  }
}
// red[j] = sum over strips of partial[strip][j] (j < 2C); optionally the same sums are added to the parameter gradients
// (dbeta += red[:C], dgamma += red[C:]) so the caller has no elementwise add to launch.  Block = 16 columns x 16 strip lanes
// (one thread per column walked its 249 strips as 62 dependent load steps: 21 us for 0.8 MB).
__global__ __launch_bounds__(256) void bn_reduce_kernel(const float* partial, int nstrip, int C, float* red, float* dgamma, float* dbeta) {
  __shared__ float sh[16][17];
  const int cx = threadIdx.x & 15, sy = threadIdx.x >> 4;
  const int j = blockIdx.x * 16 + cx;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (j < 2 * C) {
    int k = sy;
    for (; k + 48 < nstrip; k += 64) {
      s0 += partial[(int64_t)k * 2 * C + j]; s1 += partial[(int64_t)(k + 16) * 2 * C + j];
      s2 += partial[(int64_t)(k + 32) * 2 * C + j]; s3 += partial[(int64_t)(k + 48) * 2 * C + j];
    }
    for (; k < nstrip; k += 16) s0 += partial[(int64_t)k * 2 * C + j];
  }
  sh[sy][cx] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (sy == 0 && j < 2 * C) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += sh[k][cx];
    red[j] = t;
    if (j < C) { if (dbeta) dbeta[j] += t; }
    else if (dgamma) dgamma[j - C] += t;
  }
}

extern "C" int32_t otr_bn_swish_fwd(const float* y, const float* stats, const float* gamma, const float* beta,
                                    float* running_mean, float* running_var, float* saved, void* out, int32_t out_dtype,
                                    int64_t M, int32_t C, float eps, float momentum, int32_t training, void* stream) {
  OTR_REQUIRE(y && gamma && beta && saved && out, "bn_swish_fwd: null pointer");
  OTR_REQUIRE(C % 4 == 0 && M > 0, "bn_swish_fwd: bad shape");
  OTR_REQUIRE(training ? stats != nullptr : (running_mean && running_var), "bn_swish_fwd: missing statistics");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(bn_prepare_kernel, dim3((C + 127) / 128), dim3(128), 0, s, stats, running_mean, running_var, saved, C, (float)M,
                     eps, momentum, training);
  if (out_dtype == OTR_F32) hipLaunchKernelGGL(bn_swish_fwd_kernel<float>, dim3(ew_grid(M * C / 4)), dim3(256), 0, s, y, saved, gamma, beta, (float*)out, M, C);
  else hipLaunchKernelGGL(bn_swish_fwd_kernel<bf16_t>, dim3(ew_grid(M * C / 4)), dim3(256), 0, s, y, saved, gamma, beta, (bf16_t*)out, M, C);
  return otr_check_launch("bn_swish_fwd");
}

// training only: the batch statistics come as the depthwise convolution's per-workgroup sums (otr_dwconv_fwd_part)
extern "C" int32_t otr_bn_swish_fwd_part(const float* y, const float* spart, int32_t nblk, const float* gamma, const float* beta,
                                         float* running_mean, float* running_var, float* saved, void* out, int32_t out_dtype, int64_t M,
                                         int32_t C, float eps, float momentum, void* stream) {
  OTR_REQUIRE(y && spart && gamma && beta && saved && out && nblk > 0, "bn_swish_fwd_part: null pointer");
  OTR_REQUIRE(C % 4 == 0 && M > 0, "bn_swish_fwd_part: bad shape");
  OTR_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "bn_swish_fwd_part: running statistics come in pairs");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(bn_prepare_part_kernel, dim3((C + 15) / 16), dim3(256), 0, s, spart, nblk, running_mean, running_var, saved, C, (float)M, eps,
                     momentum);
  if (out_dtype == OTR_F32) hipLaunchKernelGGL(bn_swish_fwd_kernel<float>, dim3(ew_grid(M * C / 4)), dim3(256), 0, s, y, saved, gamma, beta, (float*)out, M, C);
  else hipLaunchKernelGGL(bn_swish_fwd_kernel<bf16_t>, dim3(ew_grid(M * C / 4)), dim3(256), 0, s, y, saved, gamma, beta, (bf16_t*)out, M, C);
  return otr_check_launch("bn_swish_fwd_part");
}

// red: f32 [2C]; on return red[c] = d beta, red[C + c] = d gamma (this call's sums; also the input of the second pass).
// partial: f32 [otr_bn_swish_bwd_partial_rows(M)][2C] scratch.  dgamma_acc / dbeta_acc (f32 [C], may be NULL): += the same sums.
// dy: f32 [M, C].
extern "C" int32_t otr_bn_swish_bwd_partial_rows(int64_t M) { return (int32_t)((M + BN_RPB - 1) / BN_RPB); }
extern "C" int32_t otr_bn_swish_bwd(const float* y, const void* ds, int32_t ds_dtype, const float* saved, const float* gamma,
                                    const float* beta, float* red, float* partial, float* dgamma_acc, float* dbeta_acc, float* dy,
                                    int64_t M, int32_t C, int32_t training, void* stream) {
  OTR_REQUIRE(y && ds && saved && gamma && beta && red && partial && dy, "bn_swish_bwd: null pointer");
  OTR_REQUIRE(C % 4 == 0 && M > 0, "bn_swish_bwd: bad shape");
  hipStream_t s = (hipStream_t)stream;
  const int C4 = C / 4;
  int TX = 1;
  while (TX < 256 && C4 % (2 * TX) == 0) TX *= 2;
  const int nstrip = (int)((M + BN_RPB - 1) / BN_RPB);
  const dim3 grid((unsigned)(C4 / TX), (unsigned)nstrip);
#define BN_BWD(T, MODE) hipLaunchKernelGGL((bn_swish_bwd_kernel<T, MODE>), grid, dim3(256), 0, s, y, (const T*)ds, saved, gamma, beta, red, partial, dy, M, C, (float)M, training, TX)
  if (ds_dtype == OTR_F32) BN_BWD(float, 0); else BN_BWD(bf16_t, 0);
  hipLaunchKernelGGL(bn_reduce_kernel, dim3((unsigned)((2 * C + 15) / 16)), dim3(256), 0, s, partial, nstrip, C, red, dgamma_acc, dbeta_acc);
  if (ds_dtype == OTR_F32) BN_BWD(float, 1); else BN_BWD(bf16_t, 1);
#undef BN_BWD
  return otr_check_launch("bn_swish_bwd");
}

// BatchNorm backward up to the reduced sums only (bn_swish_bwd_kernel<MODE 0> + bn_reduce_kernel): `red` [2C] for otr_conformer_conv_bwd_mid
extern "C" int32_t otr_bn_swish_bwd_sums(const float* y, const void* ds, int32_t ds_dtype, const float* saved, const float* gamma,
                                         const float* beta, float* red, float* partial, float* dgamma_acc, float* dbeta_acc, int64_t M,
                                         int32_t C, void* stream) {
  OTR_REQUIRE(y && ds && saved && gamma && beta && red && partial, "bn_swish_bwd_sums: null pointer");
  OTR_REQUIRE(C % 4 == 0 && M > 0, "bn_swish_bwd_sums: bad shape");
  hipStream_t s = (hipStream_t)stream;
  const int C4 = C / 4;
  int TX = 1;
  while (TX < 256 && C4 % (2 * TX) == 0) TX *= 2;
  const int nstrip = (int)((M + BN_RPB - 1) / BN_RPB);
  const dim3 grid((unsigned)(C4 / TX), (unsigned)nstrip);
  if (ds_dtype == OTR_F32)
    hipLaunchKernelGGL((bn_swish_bwd_kernel<float, 0>), grid, dim3(256), 0, s, y, (const float*)ds, saved, gamma, beta, red, partial, nullptr, M, C, (float)M, 1, TX);
  else
    hipLaunchKernelGGL((bn_swish_bwd_kernel<bf16_t, 0>), grid, dim3(256), 0, s, y, (const bf16_t*)ds, saved, gamma, beta, red, partial, nullptr, M, C, (float)M, 1, TX);
  hipLaunchKernelGGL(bn_reduce_kernel, dim3((unsigned)((2 * C + 15) / 16)), dim3(256), 0, s, partial, nstrip, C, red, dgamma_acc, dbeta_acc);
  return otr_check_launch("bn_swish_bwd_sums");
}

// The rest of ConformerConvolutionModule's backward between the two pointwise convolutions in one launch (conv_mid_bwd_kernel):
// dh [M, 2C] from y, ds, the reduced BatchNorm sums, g (the GLU output the depthwise conv read) and h (the GLU input).
// part [otr_dwconv_bwd_partial_rows(M)][C*k + C] and gpart [same rows][2C] receive per-workgroup sums (no atomics) of the depthwise
// conv's weight / bias gradients and of dh's columns.  All 16-bit tensors in `dtype`; k <= 7; row_mask [M] or NULL.
extern "C" int32_t otr_conformer_conv_bwd_mid(const float* y, const void* ds, const float* saved, const float* gamma, const float* beta,
                                              const float* red, const void* g, const float* w, const void* h, const uint8_t* row_mask,
                                              void* dh, float* part, float* gpart, int32_t dtype, int32_t training, int32_t B, int32_t T,
                                              int32_t C, int32_t k, int32_t pad, void* stream) {
  if (int32_t e = dw_check(B, T, C, k, pad)) return e;
  OTR_REQUIRE(y && ds && saved && gamma && beta && red && g && w && h && dh && part && gpart, "conformer_conv_bwd_mid: null pointer");
  OTR_REQUIRE(dtype == OTR_F32 || dtype == OTR_H16, "conformer_conv_bwd_mid: bad dtype");
  OTR_REQUIRE(C % 4 == 0 && C / 4 <= 256, "conformer_conv_bwd_mid: C = %d must be a multiple of 4, at most 1024", C);
  CmArgs p{};
  p.y = y; p.ds = ds; p.saved = saved; p.gamma = gamma; p.beta = beta; p.red = red; p.g = g; p.w = w; p.h = h; p.mask = row_mask; p.dh = dh;
  p.part = part; p.gpart = gpart; p.n = (float)((int64_t)B * T); p.training = training; p.B = B; p.T = T; p.C = C; p.k = k; p.pad = pad;
  hipStream_t s = (hipStream_t)stream;
  const int NY = 256 / (C / 4);
  const dim3 grid((unsigned)(((int64_t)B * T + DW_RPB - 1) / DW_RPB));
  const size_t lds = (size_t)NY * C * (k + 3) * sizeof(float);
  OTR_REQUIRE(lds <= 64 * 1024, "conformer_conv_bwd_mid: %zu bytes of LDS", lds);
#define CM_LAUNCH(KT)                                                                                        \
  {                                                                                                         \
    if (dtype == OTR_F32) hipLaunchKernelGGL((conv_mid_bwd_kernel<float, KT>), grid, dim3(256), lds, s, p, NY); \
    else hipLaunchKernelGGL((conv_mid_bwd_kernel<bf16_t, KT>), grid, dim3(256), lds, s, p, NY);              \
  }
  if (k <= 3) CM_LAUNCH(3) else if (k <= 5) CM_LAUNCH(5) else CM_LAUNCH(7)
#undef CM_LAUNCH
  return otr_check_launch("conformer_conv_bwd_mid");
}
