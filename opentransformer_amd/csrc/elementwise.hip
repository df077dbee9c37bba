// Bandwidth-bound glue kernels: GLU, positional encoding, embedding, column sums, scaling.
// All are vectorised 16 B per lane where the layout allows and sized as grid-stride loops.
#include "common.h"

static inline unsigned grid_for(int64_t work_items, int block = 256, int64_t cap = 2048 * 4) {
  int64_t g = (work_items + block - 1) / block;
  if (g < 1) g = 1;
  if (g > cap) g = cap;
  return (unsigned)g;
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + __expf(-x)); }

// ------------------------------------------------------------------------------------------------ GLU
// F.glu(h, -1): u = h[:, :F] * sigmoid(h[:, F:])     (module/ffn.py:18,40)
template <class T> __global__ void glu_fwd_kernel(const T* h, T* u, int64_t M, int64_t F, const uint8_t* row_mask) {
  constexpr int V = 16 / sizeof(T);
  const int64_t per_row = F / V, total = M * per_row;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t row = i / per_row, c = (i - row * per_row) * V;
    float a[V], g[V], o[V];
    load_row<T, V>(h + row * 2 * F + c, V, true, a);
    load_row<T, V>(h + row * 2 * F + F + c, V, true, g);
    const bool live = !row_mask || row_mask[row];
#pragma unroll
    for (int e = 0; e < V; ++e) o[e] = live ? a[e] * sigmoidf_(g[e]) : 0.f;
    if constexpr (sizeof(T) == 4) *reinterpret_cast<float4*>(u + row * F + c) = make_float4(o[0], o[1], o[2], o[3]);
    else *reinterpret_cast<uint4*>(u + row * F + c) = MMA<bf16_t>::pack(o);
  }
}

// dh[:, :F] = du * sig(g);  dh[:, F:] = du * a * sig(g) * (1 - sig(g));  optional bias-gradient partials
constexpr int GLU_RPB = 32;  // rows per block in the backward (each thread owns V columns)
// Block = TX column threads (16 bytes of a row each) x 256/TX row lanes; gridDim.x * TX covers F exactly (TX is the largest
// power of two dividing F / V, so narrow matrices -- the conformer's F = 384 gate -- still fill their 256 threads), blockIdx.y
// picks the GLU_RPB-row strip.  Four rows per lane are in flight (a serial 32-row walk per thread left the 120 MB pass of the
// conformer FFN latency-bound at 43 us).
template <class T> __global__ __launch_bounds__(256) void glu_bwd_kernel(const T* h, const T* du, T* dh, float* dbias, int64_t M, int64_t F,
                                                                         const uint8_t* row_mask, int has_sig, int TX) {
  constexpr int V = 16 / sizeof(T);
  __shared__ float red[256][2 * V + 1];
  const int tx = threadIdx.x % TX, ty = threadIdx.x / TX, TY = 256 / TX;
  const int64_t c = ((int64_t)blockIdx.x * TX + tx) * V;
  float sa[V], sg[V];
#pragma unroll
  for (int e = 0; e < V; ++e) { sa[e] = 0.f; sg[e] = 0.f; }
  const int64_t r0 = (int64_t)blockIdx.y * GLU_RPB, r1 = min(M, r0 + GLU_RPB);
  constexpr int UNR = 4;
  for (int64_t base = r0 + ty; base < r1; base += (int64_t)TY * UNR) {
    float a[UNR][V], g[UNR][V], d[UNR][V];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int64_t row = min(base + (int64_t)u * TY, r1 - 1);        // clamped duplicates are not stored or summed
      load_row<T, V>(h + row * 2 * F + c, V, true, a[u]);
      load_row<T, V>(h + row * 2 * F + F + c, V, true, g[u]);
      load_row<T, V>(du + row * F + c, V, true, d[u]);
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int64_t row = base + (int64_t)u * TY;
      if (row >= r1) break;
      const bool keep = !row_mask || row_mask[row];                   // masked_fill_(~mask, 0) after the GLU
      float oa[V], og[V];
#pragma unroll
      for (int e = 0; e < V; ++e) {
        float sgm = has_sig ? g[u][e] : sigmoidf_(g[u][e]);           // h[:, F:] may already hold sigmoid(gate) (otr_ffn_glu_fwd)
        float dd = keep ? d[u][e] : 0.f;
        oa[e] = dd * sgm;
        og[e] = dd * a[u][e] * sgm * (1.f - sgm);
        sa[e] += oa[e]; sg[e] += og[e];
      }
      if constexpr (sizeof(T) == 4) {
        *reinterpret_cast<float4*>(dh + row * 2 * F + c) = make_float4(oa[0], oa[1], oa[2], oa[3]);
        *reinterpret_cast<float4*>(dh + row * 2 * F + F + c) = make_float4(og[0], og[1], og[2], og[3]);
      } else {
        *reinterpret_cast<uint4*>(dh + row * 2 * F + c) = MMA<bf16_t>::pack(oa);
        *reinterpret_cast<uint4*>(dh + row * 2 * F + F + c) = MMA<bf16_t>::pack(og);
      }
    }
  }
  if (dbias) {  // per-row-block partial column sums [gridDim.y][2F] (no atomics); the caller column-sums them
    if (TY > 1) {
#pragma unroll
      for (int e = 0; e < V; ++e) { red[threadIdx.x][e] = sa[e]; red[threadIdx.x][V + e] = sg[e]; }
      __syncthreads();
      if (ty == 0) {
        for (int k = 1; k < TY; ++k) {
#pragma unroll
          for (int e = 0; e < V; ++e) { sa[e] += red[k * TX + tx][e]; sg[e] += red[k * TX + tx][V + e]; }
        }
      }
    }
    if (ty == 0) {
      float* dst = dbias + (int64_t)blockIdx.y * 2 * F;
#pragma unroll
      for (int e = 0; e < V; ++e) { dst[c + e] = sa[e]; dst[F + c + e] = sg[e]; }
    }
  }
}

extern "C" int32_t otr_glu_fwd(const void* h, void* u, int32_t dtype, int64_t M, int64_t F, const uint8_t* row_mask,
                               void* stream) {
  OTR_REQUIRE(h && u, "glu_fwd: null pointer");
  OTR_REQUIRE(dtype == OTR_F32 || dtype == OTR_H16, "glu_fwd: bad dtype");
  OTR_REQUIRE(F > 0 && F % 8 == 0 && M >= 0, "glu_fwd: F=%lld must be a positive multiple of 8", (long long)F);
  if (M == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == OTR_F32) hipLaunchKernelGGL(glu_fwd_kernel<float>, dim3(grid_for(M * F / 4)), dim3(256), 0, s, (const float*)h, (float*)u, M, F, row_mask);
  else hipLaunchKernelGGL(glu_fwd_kernel<bf16_t>, dim3(grid_for(M * F / 8)), dim3(256), 0, s, (const bf16_t*)h, (bf16_t*)u, M, F, row_mask);
  return otr_check_launch("glu_fwd");
}

extern "C" int32_t otr_glu_bwd(const void* h, const void* du, void* dh, float* dbias, int32_t dtype, int64_t M,
                               int64_t F, const uint8_t* row_mask, int32_t h_has_sigmoid, void* stream) {
  OTR_REQUIRE(h && du && dh, "glu_bwd: null pointer");
  OTR_REQUIRE(dtype == OTR_F32 || dtype == OTR_H16, "glu_bwd: bad dtype");
  OTR_REQUIRE(F > 0 && F % 8 == 0 && M >= 0, "glu_bwd: F=%lld must be a positive multiple of 8", (long long)F);
  if (M == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  int V = dtype == OTR_F32 ? 4 : 8;
  const int64_t cols = F / V;                    // 16-byte column groups per row
  int TX = 1;
  while (TX < 256 && cols % (2 * TX) == 0) TX *= 2;
  dim3 grid((unsigned)(cols / TX), (unsigned)((M + GLU_RPB - 1) / GLU_RPB));
  if (dtype == OTR_F32) hipLaunchKernelGGL(glu_bwd_kernel<float>, grid, dim3(256), 0, s, (const float*)h, (const float*)du, (float*)dh, dbias, M, F, row_mask, h_has_sigmoid, TX);
  else hipLaunchKernelGGL(glu_bwd_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)h, (const bf16_t*)du, (bf16_t*)dh, dbias, M, F, row_mask, h_has_sigmoid, TX);
  return otr_check_launch("glu_bwd");
}

// ------------------------------------------------------------------------------------------------ ReLU bwd
template <class T> __global__ void relu_bwd_kernel(const T* y, const T* g, T* out, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    ElemIO<T>::st(out + i, ElemIO<T>::ld(y + i) > 0.f ? ElemIO<T>::ld(g + i) : 0.f);
}
extern "C" int32_t otr_relu_bwd(const void* y, const void* g, void* out, int32_t dtype, int64_t n, void* stream) {
  OTR_REQUIRE(y && g && out, "relu_bwd: null pointer");
  OTR_REQUIRE(dtype == OTR_F32 || dtype == OTR_H16, "relu_bwd: bad dtype");
  if (n <= 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == OTR_F32) hipLaunchKernelGGL(relu_bwd_kernel<float>, dim3(grid_for(n)), dim3(256), 0, s, (const float*)y, (const float*)g, (float*)out, n);
  else hipLaunchKernelGGL(relu_bwd_kernel<bf16_t>, dim3(grid_for(n)), dim3(256), 0, s, (const bf16_t*)y, (const bf16_t*)g, (bf16_t*)out, n);
  return otr_check_launch("relu_bwd");
}

// ReLU backward of a [rows, cols] matrix with the bias gradient (column sums of the result) in the same pass: the conv2 layer of
// the frontend (frontend/conv.py:63-66) needs both, and the separate column-sum launch re-read the 41 MB it had just written.
// A thread owns 16 bytes of a row; blocks stride over the rows, four rows in flight per lane; every block leaves one row of
// per-block sums in `partial` [gridDim.x][cols] for the caller's (deferred, grouped) column sum -- no atomics.
constexpr int RBC_MAX_BLOCKS = 1024;
template <class T> __global__ __launch_bounds__(256) void relu_bwd_colsum_kernel(const T* __restrict__ y, const T* __restrict__ g,
                                                                                 T* __restrict__ out, float* __restrict__ partial,
                                                                                 int64_t rows, int cols) {
  constexpr int V = 16 / (int)sizeof(T);
  __shared__ float red[256][V + 1];
  const int tpr = cols / V;                    // threads per row (divides 256)
  const int cg = threadIdx.x % tpr, rpi = 256 / tpr;
  float s[V];
#pragma unroll
  for (int e = 0; e < V; ++e) s[e] = 0.f;
  const int64_t stride = (int64_t)gridDim.x * rpi;
  int64_t r = (int64_t)blockIdx.x * rpi + threadIdx.x / tpr;
  auto one = [&](const float* yv, const float* gv, int64_t row) {
    float o[V];
#pragma unroll
    for (int e = 0; e < V; ++e) { o[e] = yv[e] > 0.f ? gv[e] : 0.f; s[e] += o[e]; }
    T* dst = out + row * cols + cg * V;
    if constexpr (sizeof(T) == 4) *reinterpret_cast<float4*>(dst) = make_float4(o[0], o[1], o[2], o[3]);
    else *reinterpret_cast<uint4*>(dst) = MMA<bf16_t>::pack(o);
  };
  for (; r + 3 * stride < rows; r += 4 * stride) {
    float yv[4][V], gv[4][V];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      load_row<T, V>(y + (r + u * stride) * cols + cg * V, V, true, yv[u]);
      load_row<T, V>(g + (r + u * stride) * cols + cg * V, V, true, gv[u]);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) one(yv[u], gv[u], r + u * stride);
  }
  for (; r < rows; r += stride) {
    float yv[V], gv[V];
    load_row<T, V>(y + r * cols + cg * V, V, true, yv);
    load_row<T, V>(g + r * cols + cg * V, V, true, gv);
    one(yv, gv, r);
  }
#pragma unroll
  for (int e = 0; e < V; ++e) red[threadIdx.x][e] = s[e];
  __syncthreads();
  for (int c = threadIdx.x; c < cols; c += 256) {
    const int t0 = c / V, e = c % V;
    float a = 0.f;
    for (int k = 0; k < rpi; ++k) a += red[k * tpr + t0][e];
    partial[(int64_t)blockIdx.x * cols + c] = a;
  }
}
static inline int rbc_blocks(int64_t rows, int cols, int dtype) {
  const int V = dtype == OTR_F32 ? 4 : 8, rpi = 256 / (cols / V);
  const int64_t need = (rows + rpi - 1) / rpi;
  return (int)(need < 1 ? 1 : (need > RBC_MAX_BLOCKS ? RBC_MAX_BLOCKS : need));
}
static inline bool rbc_shape_ok(int64_t rows, int cols, int dtype) {
  const int V = dtype == OTR_F32 ? 4 : 8;
  return rows > 0 && cols > 0 && cols % V == 0 && cols / V <= 256 && 256 % (cols / V) == 0;
}
extern "C" int32_t otr_relu_bwd_colsum_partial_rows(int64_t rows, int32_t cols, int32_t dtype) {
  if ((dtype != OTR_F32 && dtype != OTR_H16) || !rbc_shape_ok(rows, cols, dtype)) return 0;     // 0: shape not served
  return rbc_blocks(rows, cols, dtype);
}
extern "C" int32_t otr_relu_bwd_colsum(const void* y, const void* g, void* out, float* partial, int32_t dtype, int64_t rows,
                                       int32_t cols, void* stream) {
  OTR_REQUIRE(y && g && out && partial, "relu_bwd_colsum: null pointer");
  OTR_REQUIRE(dtype == OTR_F32 || dtype == OTR_H16, "relu_bwd_colsum: bad dtype");
  OTR_REQUIRE(rbc_shape_ok(rows, cols, dtype), "relu_bwd_colsum: cols=%d must be a multiple of the 16-byte vector with cols/vector dividing 256", cols);
  OTR_REQUIRE(((uintptr_t)y | (uintptr_t)g | (uintptr_t)out) % 16 == 0, "relu_bwd_colsum: operands must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid((unsigned)rbc_blocks(rows, cols, dtype));
  if (dtype == OTR_F32) hipLaunchKernelGGL(relu_bwd_colsum_kernel<float>, grid, dim3(256), 0, s, (const float*)y, (const float*)g, (float*)out, partial, rows, cols);
  else hipLaunchKernelGGL(relu_bwd_colsum_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)y, (const bf16_t*)g, (bf16_t*)out, partial, rows, cols);
  return otr_check_launch("relu_bwd_colsum");
}

// ------------------------------------------------------------------------------------------------ FFN activations
// module/ffn.py:15-21 besides relu (GEMM epilogue) and glu (own kernels): gelu (erf form, F.gelu's default), tanh, swish.
// HBM-bound: 16 bytes per lane per access; the backward recomputes from the saved pre-activation.
enum { ACT_GELU = 1, ACT_TANH = 2, ACT_SWISH = 3 };
template <int KIND> __device__ __forceinline__ float act_f(float x) {
  if constexpr (KIND == ACT_GELU) return 0.5f * x * (1.f + erff(x * 0.70710678118654752f));
  else if constexpr (KIND == ACT_TANH) return tanhf(x);
  else return x / (1.f + __expf(-x));
}
template <int KIND> __device__ __forceinline__ float act_df(float x) {
  if constexpr (KIND == ACT_GELU) {
    return 0.5f * (1.f + erff(x * 0.70710678118654752f)) + x * 0.3989422804014327f * __expf(-0.5f * x * x);
  } else if constexpr (KIND == ACT_TANH) {
    const float t = tanhf(x);
    return 1.f - t * t;
  } else {
    const float sg = 1.f / (1.f + __expf(-x));
    return sg * (1.f + x * (1.f - sg));
  }
}
template <class T, int KIND, bool BWD> __global__ void act_kernel(const T* __restrict__ x, const T* __restrict__ dy, T* __restrict__ out, int64_t n) {
  constexpr int V = 16 / (int)sizeof(T);
  const int64_t nv = n / V;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (int64_t)gridDim.x * blockDim.x) {
    float xv[V], gv[V], ov[V];
    load_row<T, V>(x + i * V, V, true, xv);
    if constexpr (BWD) load_row<T, V>(dy + i * V, V, true, gv);
#pragma unroll
    for (int e = 0; e < V; ++e) ov[e] = BWD ? gv[e] * act_df<KIND>(xv[e]) : act_f<KIND>(xv[e]);
    if constexpr (sizeof(T) == 4) *reinterpret_cast<float4*>(out + i * V) = make_float4(ov[0], ov[1], ov[2], ov[3]);
    else *reinterpret_cast<uint4*>(out + i * V) = MMA<bf16_t>::pack(ov);
  }
  if (blockIdx.x == 0)   // tail (n not a multiple of the vector width)
    for (int64_t i = nv * V + threadIdx.x; i < n; i += blockDim.x) {
      const float xe = ElemIO<T>::ld(x + i);
      ElemIO<T>::st(out + i, BWD ? ElemIO<T>::ld(dy + i) * act_df<KIND>(xe) : act_f<KIND>(xe));
    }
}
template <class T, bool BWD> static void act_launch(int kind, const void* x, const void* dy, void* out, int64_t n, hipStream_t s) {
  const unsigned g = grid_for(n / (16 / (int)sizeof(T)) + 1);
#define OTR_ACT(K) hipLaunchKernelGGL((act_kernel<T, K, BWD>), dim3(g), dim3(256), 0, s, (const T*)x, (const T*)dy, (T*)out, n)
  if (kind == ACT_GELU) OTR_ACT(ACT_GELU);
  else if (kind == ACT_TANH) OTR_ACT(ACT_TANH);
  else OTR_ACT(ACT_SWISH);
#undef OTR_ACT
}
static int32_t act_check(const char* what, const void* x, const void* out, int32_t dtype, int64_t n, int32_t kind) {
  OTR_REQUIRE(x && out, "%s: null pointer", what);
  OTR_REQUIRE(dtype == OTR_F32 || dtype == OTR_H16, "%s: bad dtype", what);
  OTR_REQUIRE(kind >= ACT_GELU && kind <= ACT_SWISH, "%s: kind must be 1 (gelu), 2 (tanh) or 3 (swish)", what);
  OTR_REQUIRE(n >= 0 && (uintptr_t)x % 16 == 0 && (uintptr_t)out % 16 == 0, "%s: buffers must be 16-byte aligned", what);
  return 0;
}
extern "C" int32_t otr_act_fwd(const void* x, void* y, int32_t dtype, int64_t n, int32_t kind, void* stream) {
  if (int32_t e = act_check("act_fwd", x, y, dtype, n, kind)) return e;
  if (n == 0) return 0;
  if (dtype == OTR_F32) act_launch<float, false>(kind, x, nullptr, y, n, (hipStream_t)stream);
  else act_launch<bf16_t, false>(kind, x, nullptr, y, n, (hipStream_t)stream);
  return otr_check_launch("act_fwd");
}
extern "C" int32_t otr_act_bwd(const void* x, const void* dy, void* dx, int32_t dtype, int64_t n, int32_t kind, void* stream) {
  if (int32_t e = act_check("act_bwd", x, dx, dtype, n, kind)) return e;
  OTR_REQUIRE(dy && (uintptr_t)dy % 16 == 0, "act_bwd: dy must be a 16-byte aligned pointer");
  if (n == 0) return 0;
  if (dtype == OTR_F32) act_launch<float, true>(kind, x, dy, dx, n, (hipStream_t)stream);
  else act_launch<bf16_t, true>(kind, x, dy, dx, n, (hipStream_t)stream);
  return otr_check_launch("act_bwd");
}

// ------------------------------------------------------------------------------------------------ posenc / embedding
// PE[t, 2i] = sin(t * exp(-2i ln(1e4)/d)), PE[t, 2i+1] = cos(same)        (module/pos.py:30-42)
// (pe_value lives in common.h: the incremental decoder must produce the same bits)

// Four columns per thread (16-byte loads / stores; the table entry is still pe_value, element by element: the decoder's incremental
// step must produce the same bits).  Optionally the launch also leaves the encoder's key mask as bytes: mask_out[row] =
// mask_in[(row / T) * mask_bs + (row % T) * mask_ts] != 0 -- the frame mask after the two stride-2 convolutions is the strided view
// mask[:, 1::2][:, :t1][:, 1::2][:, :t2] of the batch's bool mask (frontend/conv.py:78-83), and casting it was a launch of its own
// (9 us in the AISHELL step) in front of the first attention kernel.
__global__ void posenc_kernel(const float* x, float* y, bf16_t* y_lp, int64_t rows, int T, int d, float scale, const uint8_t* mask_in,
                              int64_t mask_bs, int64_t mask_ts, uint8_t* mask_out) {
  const float nl = -logf(10000.f) / (float)d;
  const int d4 = d >> 2;
  const int64_t total = rows * d4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / d4;
    const int col = (int)(i - row * d4) * 4;
    const int t = (int)(row % T);
    const float4 xv = reinterpret_cast<const float4*>(x)[i];
    float4 v;
    v.x = xv.x * scale + pe_value(t, col, nl);
    v.y = xv.y * scale + pe_value(t, col + 1, nl);
    v.z = xv.z * scale + pe_value(t, col + 2, nl);
    v.w = xv.w * scale + pe_value(t, col + 3, nl);
    reinterpret_cast<float4*>(y)[i] = v;
    if (y_lp) reinterpret_cast<uint2*>(y_lp)[i] = make_uint2(pack2bf(v.x, v.y), pack2bf(v.z, v.w));
    if (mask_out && col == 0) mask_out[row] = mask_in[(row / T) * mask_bs + (int64_t)t * mask_ts] != 0 ? 1 : 0;
  }
}
// any d, any alignment: one element per thread (the four-column kernel above must not be the only form -- a model with
// d_model % 4 != 0, or a contiguous view at an odd storage offset, ran before round 5 and runs again)
__global__ void posenc_scalar_kernel(const float* x, float* y, bf16_t* y_lp, int64_t rows, int T, int d, float scale, const uint8_t* mask_in,
                                     int64_t mask_bs, int64_t mask_ts, uint8_t* mask_out) {
  const float nl = -logf(10000.f) / (float)d;
  const int64_t total = rows * d;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / d;
    const int col = (int)(i - row * d);
    const int t = (int)(row % T);
    const float v = x[i] * scale + pe_value(t, col, nl);
    y[i] = v;
    if (y_lp) y_lp[i] = f2bf(v);
    if (mask_out && col == 0) mask_out[row] = mask_in[(row / T) * mask_bs + (int64_t)t * mask_ts] != 0 ? 1 : 0;
  }
}
extern "C" int32_t otr_posenc_mask_fwd(const float* x, float* y, void* y_bf16, int64_t rows, int32_t T, int32_t d, float scale,
                                       const uint8_t* mask_in, int64_t mask_bs, int64_t mask_ts, uint8_t* mask_out, void* stream) {
  OTR_REQUIRE(x && y, "posenc_fwd: null pointer");
  OTR_REQUIRE(T > 0 && d > 0 && rows >= 0, "posenc_fwd: bad shape");
  OTR_REQUIRE((mask_in == nullptr) == (mask_out == nullptr), "posenc_mask_fwd: mask_in and mask_out go together");
  if (rows == 0) return 0;
  const bool vec = d % 4 == 0 && ((uintptr_t)x | (uintptr_t)y) % 16 == 0 && (uintptr_t)y_bf16 % 8 == 0;
  if (vec)
    hipLaunchKernelGGL(posenc_kernel, dim3(grid_for(rows * (d / 4))), dim3(256), 0, (hipStream_t)stream, x, y, (bf16_t*)y_bf16, rows, T, d, scale,
                       mask_in, mask_bs, mask_ts, mask_out);
  else
    hipLaunchKernelGGL(posenc_scalar_kernel, dim3(grid_for(rows * (int64_t)d)), dim3(256), 0, (hipStream_t)stream, x, y, (bf16_t*)y_bf16, rows, T,
                       d, scale, mask_in, mask_bs, mask_ts, mask_out);
  return otr_check_launch("posenc_fwd");
}
extern "C" int32_t otr_posenc_fwd(const float* x, float* y, void* y_bf16, int64_t rows, int32_t T, int32_t d,
                                  float scale, void* stream) {
  return otr_posenc_mask_fwd(x, y, y_bf16, rows, T, d, scale, nullptr, 0, 0, nullptr, stream);
}

// tokens are addressed as tok[(row / L) * ldt + row % L]: a [B, L] view with row stride ldt (truth[:, :-1] of a [B, L + 1] matrix,
// model/speech2text.py:53) needs no copy
__global__ void embed_posenc_kernel(const int64_t* tok, int64_t ldt, const float* E, float* y, bf16_t* y_lp, int64_t rows, int L, int d,
                                    int vocab, float scale) {
  const float nl = -logf(10000.f) / (float)d;
  const int64_t total = rows * d;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t row = i / d;
    int col = (int)(i - row * d);
    int64_t t = tok[(row / L) * ldt + (row % L)];
    float e = (t >= 0 && t < vocab) ? E[t * d + col] : 0.f;
    float v = e * scale + pe_value((int)(row % L), col, nl);
    y[i] = v;
    if (y_lp) y_lp[i] = f2bf(v);
  }
}
extern "C" int32_t otr_embed_posenc_fwd_ld(const int64_t* tok, int64_t ld_tok, const float* E, float* y, void* y_bf16, int64_t rows,
                                           int32_t L, int32_t d, int32_t vocab, float scale, void* stream) {
  OTR_REQUIRE(tok && E && y, "embed_posenc_fwd: null pointer");
  OTR_REQUIRE(L > 0 && d > 0 && vocab > 0 && rows >= 0 && ld_tok >= L, "embed_posenc_fwd: bad shape");
  if (rows == 0) return 0;
  hipLaunchKernelGGL(embed_posenc_kernel, dim3(grid_for(rows * d)), dim3(256), 0, (hipStream_t)stream, tok, ld_tok, E, y, (bf16_t*)y_bf16, rows, L, d,
                     vocab, scale);
  return otr_check_launch("embed_posenc_fwd");
}
extern "C" int32_t otr_embed_posenc_fwd(const int64_t* tok, const float* E, float* y, void* y_bf16, int64_t rows, int32_t L,
                                        int32_t d, int32_t vocab, float scale, void* stream) {
  return otr_embed_posenc_fwd_ld(tok, L, E, y, y_bf16, rows, L, d, vocab, scale, stream);
}

// dE[tok[r],:] += scale * (dy[r,:] + sum_s slabs[s][r,:]): `slabs` (16-bit partial sums [nslab][rows][d], the fused decoder stack's
// last launch leaves its input gradient that way: csrc/declayer.hip) may be NULL / nslab 0
__global__ void embed_bwd_kernel(const int64_t* tok, int64_t ldt, int L, const float* dy, const bf16_t* slabs, int nslab, float* dE, int64_t rows,
                                 int d, int vocab, float scale) {
  const int64_t total = rows * d;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t row = i / d;
    int col = (int)(i - row * d);
    int64_t t = tok[(row / L) * ldt + (row % L)];
    if (t >= 0 && t < vocab) {
      float g = dy ? dy[i] : 0.f;
      for (int s = 0; s < nslab; ++s) g += bf2f(slabs[(int64_t)s * total + i]);
      atomicAdd(dE + t * d + col, g * scale);
    }
  }
}
extern "C" int32_t otr_embed_bwd_ld(const int64_t* tok, int64_t ld_tok, int32_t L, const float* dy, const void* slabs, int32_t nslab, float* dE,
                                    int64_t rows, int32_t d, int32_t vocab, float scale, void* stream) {
  OTR_REQUIRE(tok && dE && (dy || (slabs && nslab > 0)), "embed_bwd: null pointer");
  OTR_REQUIRE(L > 0 && ld_tok >= L && nslab >= 0 && (nslab == 0 || slabs), "embed_bwd: bad shape");
  if (rows <= 0) return 0;
  hipLaunchKernelGGL(embed_bwd_kernel, dim3(grid_for(rows * d)), dim3(256), 0, (hipStream_t)stream, tok, ld_tok, L, dy, (const bf16_t*)slabs, nslab, dE,
                     rows, d, vocab, scale);
  return otr_check_launch("embed_bwd");
}
extern "C" int32_t otr_embed_bwd(const int64_t* tok, const float* dy, float* dE, int64_t rows, int32_t d, int32_t vocab,
                                 float scale, void* stream) {
  OTR_REQUIRE(tok && dy && dE, "embed_bwd: null pointer");
  if (rows <= 0) return 0;
  OTR_REQUIRE(rows <= 0x7fffffff, "embed_bwd: too many rows");
  return otr_embed_bwd_ld(tok, rows, (int32_t)rows, dy, nullptr, 0, dE, rows, d, vocab, scale, stream);
}

// ------------------------------------------------------------------------------------------------ cast
__global__ void cast_bf16_kernel(const float* src, bf16_t* dst, int64_t n) {
  const int64_t n4 = n / 4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    float4 v = reinterpret_cast<const float4*>(src)[i];
    reinterpret_cast<uint2*>(dst)[i] = make_uint2(pack2bf(v.x, v.y), pack2bf(v.z, v.w));
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) dst[n4 * 4 + threadIdx.x] = f2bf(src[n4 * 4 + threadIdx.x]);
}
extern "C" int32_t otr_cast_f32_to_bf16(const float* src, void* dst, int64_t n, void* stream) {
  OTR_REQUIRE(src && dst, "cast_f32_to_bf16: null pointer");
  OTR_REQUIRE((uintptr_t)src % 16 == 0 && (uintptr_t)dst % 8 == 0, "cast_f32_to_bf16: unaligned buffers");
  if (n <= 0) return 0;
  hipLaunchKernelGGL(cast_bf16_kernel, dim3(grid_for(n / 4 + 1)), dim3(256), 0, (hipStream_t)stream, src, (bf16_t*)dst, n);
  return otr_check_launch("cast_f32_to_bf16");
}

// ------------------------------------------------------------------------------------------------ scale
__global__ void scale_kernel(const float* x, float* y, int64_t n, const float* s_dev, float s_host) {
  const float s = (s_dev ? *s_dev : 1.f) * s_host;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) y[i] = x[i] * s;
}
extern "C" int32_t otr_scale(const float* x, float* y, int64_t n, const float* s_dev, float s_host, void* stream) {
  OTR_REQUIRE(x && y, "scale: null pointer");
  if (n <= 0) return 0;
  hipLaunchKernelGGL(scale_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, y, n, s_dev, s_host);
  return otr_check_launch("scale");
}

// y16 = (16-bit) (x * s): the gradient of a positional encoding's input (module/pos.py:44-57: dx = sqrt(d) dy) leaves as the 16-bit
// GEMM operand its only consumers -- the three GEMMs of the Linear in front of it -- read, instead of as fp32 (16 bytes per lane)
__global__ void scale_cast_kernel(const float* x, bf16_t* y, int64_t n, float s) {
  const int64_t n4 = n >> 2;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const float4 v = reinterpret_cast<const float4*>(x)[i];
    reinterpret_cast<uint2*>(y)[i] = make_uint2(pack2bf(v.x * s, v.y * s), pack2bf(v.z * s, v.w * s));
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) y[4 * n4 + threadIdx.x] = f2bf(x[4 * n4 + threadIdx.x] * s);
}
extern "C" int32_t otr_scale_cast(const float* x, void* y16, int64_t n, float s, void* stream) {
  OTR_REQUIRE(x && y16, "scale_cast: null pointer");
  OTR_REQUIRE((uintptr_t)x % 16 == 0 && (uintptr_t)y16 % 8 == 0, "scale_cast: unaligned buffers");
  if (n <= 0) return 0;
  hipLaunchKernelGGL(scale_cast_kernel, dim3(grid_for(n / 4 + 1)), dim3(256), 0, (hipStream_t)stream, x, (bf16_t*)y16, n, s);
  return otr_check_launch("scale_cast");
}

// ------------------------------------------------------------------------------------------------ start of a training step
// The two launches every step begins with -- zero the flat gradient buffer (146 MB at the AISHELL model: 19.6 us), advance the
// dropout seed (an 8-byte add: 5 us of launch) -- as one: thread 0 of workgroup 0 bumps the counter, everybody clears.  16-byte
// stores; n = floats, the buffer 16-byte aligned; counter may be NULL.
__global__ __launch_bounds__(256) void zero_tick_kernel(float* buf, int64_t n, int64_t* counter, int64_t inc) {
  if (counter && blockIdx.x == 0 && threadIdx.x == 0) counter[0] += inc;
  const int64_t n4 = n >> 2;
  const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) reinterpret_cast<float4*>(buf)[i] = z;
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) buf[4 * n4 + threadIdx.x] = 0.f;
}
extern "C" int32_t otr_zero_tick(float* buf, int64_t n, int64_t* counter, int64_t inc, void* stream) {
  OTR_REQUIRE(n >= 0 && (buf || n == 0), "zero_tick: bad buffer");
  OTR_REQUIRE((uintptr_t)buf % 16 == 0 && (uintptr_t)counter % 8 == 0, "zero_tick: buffer must be 16-byte, counter 8-byte aligned");
  if (n == 0 && !counter) return 0;
  // one 16-byte store per thread (the fill of 146 MB: 19.6 us that way, 23.8 us as 4096 grid-stride workgroups)
  const int64_t g = ((n >> 2) + 255) / 256;
  OTR_REQUIRE(g < (1ll << 31), "zero_tick: buffer too large");
  hipLaunchKernelGGL(zero_tick_kernel, dim3((unsigned)(g < 1 ? 1 : g)), dim3(256), 0, (hipStream_t)stream, buf, n, counter, inc);
  return otr_check_launch("zero_tick");
}

// ------------------------------------------------------------------------------------------------ touch
// Read a range once, consume nothing: the lines are then in the memory-side cache (256 MB, shared by the XCDs, not flushed between
// launches) for the launches that follow.  The fused decoder uses it for its packed weights: 64-120 workgroups per launch stream
// 1/4 .. 1/8 of a layer's weights each and wait for every cold line (csrc/declayer.hip; profiles/r04_dec_trace.txt is 6 us per FFN
// launch faster on warm weights than the same launches inside the step).  One dword per 64 bytes.
__global__ __launch_bounds__(256) void touch_kernel(const unsigned char* p, int64_t lines) {
  uint32_t acc = 0;
  for (int64_t l = (int64_t)blockIdx.x * 256 + threadIdx.x; l < lines; l += (int64_t)gridDim.x * 256)
    acc |= *reinterpret_cast<const volatile uint32_t*>(p + l * 64);
  if (acc == 0x9e3779b9u && lines < 0) const_cast<unsigned char*>(p)[0] = 0;      // never true: keeps the loads
}
extern "C" int32_t otr_touch(const void* p, int64_t bytes, void* stream) {
  OTR_REQUIRE(bytes >= 0 && (p || bytes == 0), "touch: bad range");
  const int64_t lines = bytes / 64;
  if (lines == 0) return 0;
  const int64_t g = (lines + 255) / 256;
  hipLaunchKernelGGL(touch_kernel, dim3((unsigned)(g > 1024 ? 1024 : g)), dim3(256), 0, (hipStream_t)stream, (const unsigned char*)p, lines);
  return otr_check_launch("touch");
}

// ------------------------------------------------------------------------------------------------ regrouping add
// dst[r, c, f] += src[r, f, c] (fp32), optionally leaving src ZERO: a weight gradient that a kernel produced in ITS column order
// (the frontend Linear's f*C+c columns, conv2's channel-last taps) lands in the parameter's layout.  With `clear` the staging image is
// zero again when the launch ends -- the next backward pass, whenever and however it is issued (eager, or a replay of a graph that
// was captured without the gradient clear), starts from zeros without anybody on the host having to know (r06: a captured forward +
// backward replayed twice left 3 x the frontend Linear's gradient; the host-side "dirty" flag had been baked into the graph).
// One 32 x 32 tile of (f, c) per workgroup and row: both sides move as 128-byte row pieces.
__global__ __launch_bounds__(256) void regroup_add_kernel(float* __restrict__ dst, float* __restrict__ src, int C, int F, int clear) {
  __shared__ float tile[32][33];
  const int r = blockIdx.z, f0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;          // 8 rows of 32 per pass
  float* s = src + (int64_t)r * F * C;
  float* d = dst + (int64_t)r * C * F;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int f = f0 + ty + 8 * i, c = c0 + tx;
    if (f < F && c < C) {
      tile[ty + 8 * i][tx] = s[(int64_t)f * C + c];
      if (clear) s[(int64_t)f * C + c] = 0.f;
    }
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = c0 + ty + 8 * i, f = f0 + tx;
    if (f < F && c < C) d[(int64_t)c * F + f] += tile[tx][ty + 8 * i];
  }
}
extern "C" int32_t otr_regroup_add(float* dst, float* src, int64_t rows, int32_t C, int32_t F, int32_t clear_src, void* stream) {
  OTR_REQUIRE(dst && src, "regroup_add: null pointer");
  OTR_REQUIRE(rows >= 0 && rows < 65536 && C > 0 && F > 0, "regroup_add: bad shape");
  if (rows == 0) return 0;
  hipLaunchKernelGGL(regroup_add_kernel, dim3((unsigned)((C + 31) / 32), (unsigned)((F + 31) / 32), (unsigned)rows), dim3(256), 0, (hipStream_t)stream,
                     dst, src, C, F, clear_src);
  return otr_check_launch("regroup_add");
}

// ------------------------------------------------------------------------------------------------ column sums
// out[n] (+)= sum_m a[m, n].  block (64 x 4): x -> 4 consecutive columns per lane, y -> row lanes.
constexpr int CS_RPB = 128;
template <class T> __global__ void colsum_kernel(const T* a, int64_t M, int64_t N, int64_t lda, float* out) {
  __shared__ float red[4][64][4];
  const int tx = threadIdx.x, ty = threadIdx.y;
  const int64_t c = ((int64_t)blockIdx.x * 64 + tx) * 4;
  float s[4] = {0.f, 0.f, 0.f, 0.f};
  if (c < N) {
    const int64_t r0 = (int64_t)blockIdx.y * CS_RPB, r1 = min(M, r0 + CS_RPB);
    const bool full = c + 4 <= N && (lda % 4 == 0);
    for (int64_t r = r0 + ty; r < r1; r += 4) {
      float v[4];
      load_row<T, 4>(a + r * lda + c, (int)min((int64_t)4, N - c), full, v);
#pragma unroll
      for (int e = 0; e < 4; ++e) s[e] += v[e];
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) red[ty][tx][e] = s[e];
  __syncthreads();
  if (ty == 0 && c < N) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (c + e < N) atomicAdd(out + c + e, red[0][tx][e] + red[1][tx][e] + red[2][tx][e] + red[3][tx][e]);
  }
}
extern "C" int32_t otr_colsum(const void* a, int32_t dtype, int64_t M, int64_t N, int64_t lda, float* out,
                              int32_t accumulate, void* stream) {
  OTR_REQUIRE(a && out, "colsum: null pointer");
  OTR_REQUIRE(dtype == OTR_F32 || dtype == OTR_H16, "colsum: bad dtype");
  OTR_REQUIRE(N > 0 && M >= 0 && lda >= N, "colsum: bad shape");
  OTR_REQUIRE((uintptr_t)a % 16 == 0, "colsum: input must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  if (!accumulate) otr_zero_f32(out, N, s);
  if (M == 0) return 0;
  dim3 grid((unsigned)((N + 255) / 256), (unsigned)((M + CS_RPB - 1) / CS_RPB)), block(64, 4);
  if (dtype == OTR_F32) hipLaunchKernelGGL(colsum_kernel<float>, grid, block, 0, s, (const float*)a, M, N, lda, out);
  else hipLaunchKernelGGL(colsum_kernel<bf16_t>, grid, block, 0, s, (const bf16_t*)a, M, N, lda, out);
  return otr_check_launch("colsum");
}

// many column sums in one launch (all bias gradients of a backward pass): out_i[N_i] += colsum(a_i[M_i, N_i])
constexpr int CSG_MAX = 64;
struct ColsumGroup {
  int n;
  int first[CSG_MAX + 1];
  const void* a[CSG_MAX];
  float* out[CSG_MAX];
  int M[CSG_MAX], N[CSG_MAX], lda[CSG_MAX], rblocks[CSG_MAX];
  int tpr[CSG_MAX];   // threads per row: a thread owns 4 columns, a block min(N, 256) columns -> 256 / tpr rows per pass
  int rpb[CSG_MAX];   // rows per block: >= 128, and few enough blocks per column (<= ~192) that their atomics do not pile up
};
// EPT = 4 columns per thread for both element types: the matrices of a backward pass are mostly [M, 256], which 8 columns
// per thread (16-byte loads of 16-bit rows) would cover with half a wave -- measured 169 -> 300 us per training step.
// Narrow matrices (the conv2 bias gradient is [151392, 32]) fold their rows into the idle lanes: 8 threads per row, 32 rows
// per pass (with 64 threads per row 7/8 of every wave idled: 128 us of a training step for 10 MB).
template <class T> __global__ void colsum_grouped_kernel(ColsumGroup g) {
  constexpr int EPT = 4, UN = 8;
  __shared__ float red[256][EPT];
  const int b = (int)blockIdx.x;
  int i = 0;
  for (int j = 1; j < g.n; ++j) i = (g.first[j] <= b) ? j : i;
  const T* a = reinterpret_cast<const T*>(g.a[i]);
  float* out = g.out[i];
  const int64_t M = g.M[i], N = g.N[i], lda = g.lda[i];
  const int tpr = g.tpr[i], nrl = 256 / tpr;                 // row lanes of the block
  const int rpb = g.rpb[i];
  const int lb = b - g.first[i], by = lb % g.rblocks[i], bx = lb / g.rblocks[i];
  const int t = (int)(threadIdx.y * 64 + threadIdx.x);
  const int tx = t % tpr, ty = t / tpr;
  const int64_t c = ((int64_t)bx * tpr + tx) * EPT;
  float s[EPT];
#pragma unroll
  for (int e = 0; e < EPT; ++e) s[e] = 0.f;
  if (c < N && ty < nrl) {
    const int64_t r0 = (int64_t)by * rpb, r1 = min(M, r0 + rpb);
    const bool full = c + EPT <= N && (lda % EPT == 0);
    int64_t r = r0 + ty;
    if (full) {   // UN rows in flight per thread (rows clamped, tail rows weighted 0): the serial loop was latency bound
      for (; r < r1; r += (int64_t)UN * nrl) {
        float v[UN][EPT];
#pragma unroll
        for (int u = 0; u < UN; ++u) load_row<T, EPT>(a + min(r + (int64_t)nrl * u, M - 1) * lda + c, EPT, true, v[u]);
#pragma unroll
        for (int u = 0; u < UN; ++u) {
          const float w = (r + (int64_t)nrl * u < r1) ? 1.f : 0.f;
#pragma unroll
          for (int e = 0; e < EPT; ++e) s[e] += w * v[u][e];
        }
      }
    } else {
      for (; r < r1; r += nrl) {
        float v[EPT];
        load_row<T, EPT>(a + r * lda + c, (int)min((int64_t)EPT, N - c), false, v);
#pragma unroll
        for (int e = 0; e < EPT; ++e) s[e] += v[e];
      }
    }
  }
#pragma unroll
  for (int e = 0; e < EPT; ++e) red[t][e] = s[e];
  __syncthreads();
  if (ty == 0 && c < N) {
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      float v = 0.f;
      for (int q = 0; q < nrl; ++q) v += red[q * tpr + tx][e];
      if (c + e < N) atomicAdd(out + c + e, v);
    }
  }
}
extern "C" int32_t otr_colsum_grouped(const otr_colsum_item_t* items, int32_t n, void* stream) {
  OTR_REQUIRE(n >= 0 && (items || n == 0), "colsum_grouped: null items");
  hipStream_t s = (hipStream_t)stream;
  for (int pass = 0; pass < 2; ++pass) {
    const int dt = pass == 0 ? OTR_F32 : OTR_H16;
    ColsumGroup g{};
    int blocks = 0;
    auto flush = [&]() -> int32_t {
      if (g.n == 0) return 0;
      g.first[g.n] = blocks;
      if (dt == OTR_F32) hipLaunchKernelGGL(colsum_grouped_kernel<float>, dim3((unsigned)blocks), dim3(64, 4), 0, s, g);
      else hipLaunchKernelGGL(colsum_grouped_kernel<bf16_t>, dim3((unsigned)blocks), dim3(64, 4), 0, s, g);
      g.n = 0;
      blocks = 0;
      return otr_check_launch("colsum_grouped");
    };
    for (int i = 0; i < n; ++i) {
      const otr_colsum_item_t& it = items[i];
      OTR_REQUIRE(it.a && it.out, "colsum_grouped: item %d has a null pointer", i);
      OTR_REQUIRE(it.dtype == OTR_F32 || it.dtype == OTR_H16, "colsum_grouped: item %d has a bad dtype", i);
      OTR_REQUIRE(it.N > 0 && it.M >= 0 && it.lda >= it.N && it.lda < (1ll << 31) && it.M < (1ll << 31),
                  "colsum_grouped: item %d has a bad shape", i);
      OTR_REQUIRE((uintptr_t)it.a % 16 == 0, "colsum_grouped: item %d input must be 16-byte aligned", i);
      if (it.dtype != dt || it.M == 0) continue;
      int tpr = 64;                                           // threads per row: the power of two >= N / 4, at most 64
      while (tpr > 8 && (int64_t)(tpr / 2) * 4 >= it.N) tpr /= 2;
      // every block ends with one atomic per column: 1245 blocks of a [159360, 128] matrix queued 20 k atomics on each of its 8
      // cache lines -- 100 of that launch's 127 us.  At most ~192 row blocks per matrix.
      const int nrl = 256 / tpr, pass_rows = nrl * 8 > 128 ? nrl * 8 : 128;
      int rpb = pass_rows;
      if ((it.M + rpb - 1) / rpb > 192) rpb = (int)(((it.M + 191) / 192 + pass_rows - 1) / pass_rows) * pass_rows;
      const int cpb = tpr * 4;                                // columns per block
      const int rb = (int)((it.M + rpb - 1) / rpb), cb = (int)((it.N + cpb - 1) / cpb);
      const int k = g.n++;
      g.first[k] = blocks;
      g.a[k] = it.a; g.out[k] = it.out; g.M[k] = (int)it.M; g.N[k] = (int)it.N; g.lda[k] = (int)it.lda; g.rblocks[k] = rb; g.tpr[k] = tpr; g.rpb[k] = rpb;
      blocks += rb * cb;
      if (g.n == CSG_MAX)
        if (int32_t e = flush()) return e;
    }
    if (int32_t e = flush()) return e;
  }
  return 0;
}

// ------------------------------------------------------------------------------------------------ batched transpose
// One launch transposes every 2-D weight shadow: block -> (matrix, 64x64 tile) by binary search in the tile prefix.
template <class T>
__global__ __launch_bounds__(256) void transpose_batched_kernel(const T* __restrict__ src, T* __restrict__ dst,
                                                               const int64_t* __restrict__ table, int n_mats) {
  __shared__ T tile[64][65];
  const int64_t b = blockIdx.x;
  int lo = 0, hi = n_mats - 1;
  while (lo < hi) {                                   // last matrix whose first tile <= b
    int mid = (lo + hi + 1) >> 1;
    if (table[mid * 4 + 3] <= b) lo = mid; else hi = mid - 1;
  }
  const int64_t off = table[lo * 4], rows = table[lo * 4 + 1], cols = table[lo * 4 + 2];
  const int64_t t = b - table[lo * 4 + 3];
  const int64_t tc = (cols + 63) >> 6;
  const int64_t r0 = (t / tc) << 6, c0 = (t % tc) << 6;
  const T* s = src + off;
  T* d = dst + off;
  if constexpr (sizeof(T) == 2) {
    // full 64x64 tiles of 16-bit matrices with 16-byte aligned rows: 16-byte loads and stores (the element-wise form ran at
    // 1.9 TB/s on the 73 MB of weight shadows refreshed after every optimizer step)
    const bool fast = r0 + 64 <= rows && c0 + 64 <= cols && cols % 8 == 0 && rows % 8 == 0 && off % 8 == 0;
    if (fast) {
      const int tid = threadIdx.x;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int id = tid + 256 * u, r = id >> 3, ch = id & 7;
        const uint4 v = *reinterpret_cast<const uint4*>(s + (r0 + r) * cols + c0 + ch * 8);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) tile[r][ch * 8 + e] = (T)((w[e >> 1] >> ((e & 1) * 16)) & 0xffffu);
      }
      __syncthreads();
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int id = tid + 256 * u, c = id >> 3, ch = id & 7;      // output row c (a source column), 8 source rows per chunk
        uint32_t w[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) w[e] = (uint32_t)tile[ch * 8 + 2 * e][c] | ((uint32_t)tile[ch * 8 + 2 * e + 1][c] << 16);
        *reinterpret_cast<uint4*>(d + (c0 + c) * rows + r0 + ch * 8) = make_uint4(w[0], w[1], w[2], w[3]);
      }
      return;
    }
  }
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int r = ty; r < 64; r += 4)
    if (r0 + r < rows && c0 + tx < cols) tile[r][tx] = s[(r0 + r) * cols + c0 + tx];
  __syncthreads();
  for (int c = ty; c < 64; c += 4)
    if (c0 + c < cols && r0 + tx < rows) d[(c0 + c) * rows + r0 + tx] = tile[tx][c];
}

extern "C" int32_t otr_transpose_batched(const void* src, void* dst, const int64_t* table, int32_t n_mats,
                                         int64_t total_tiles, int32_t elem_bytes, void* stream) {
  OTR_REQUIRE(src && dst && table, "transpose_batched: null pointer");
  OTR_REQUIRE(src != dst, "transpose_batched: in-place transposition is not supported");
  OTR_REQUIRE(elem_bytes == 2 || elem_bytes == 4, "transpose_batched: elem_bytes must be 2 or 4");
  OTR_REQUIRE(n_mats >= 0 && total_tiles >= 0 && total_tiles < (1ll << 31), "transpose_batched: bad sizes");
  if (n_mats == 0 || total_tiles == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  if (elem_bytes == 2)
    hipLaunchKernelGGL(transpose_batched_kernel<uint16_t>, dim3((unsigned)total_tiles), dim3(256), 0, s,
                       (const uint16_t*)src, (uint16_t*)dst, table, n_mats);
  else
    hipLaunchKernelGGL(transpose_batched_kernel<uint32_t>, dim3((unsigned)total_tiles), dim3(256), 0, s,
                       (const uint32_t*)src, (uint32_t*)dst, table, n_mats);
  return otr_check_launch("transpose_batched");
}

// ------------------------------------------------------------------------------------------------ SpecAugment masks
// x[b, t, f] = 0 wherever (t, f) falls into one of the NR rectangles of utterance b.  ranges: int32 [B, NR, 4] =
// {t0, t1, f0, f1} half-open (a frequency mask covers all t, a time mask all f).  data/augment.py:9-41 draws the
// rectangles on the host with numpy / random; drawing them there with the same calls keeps the masks bit-identical.
__global__ void spec_mask_kernel(float* x, const int32_t* ranges, int NR, int T, int F) {
  const int b = blockIdx.y;
  const int32_t* r = ranges + (int64_t)b * NR * 4;
  float* xb = x + (int64_t)b * T * F;
  const int64_t total = (int64_t)T * F;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int t = (int)(i / F), f = (int)(i - (int64_t)t * F);
    bool hit = false;
    for (int k = 0; k < NR; ++k) hit = hit || (t >= r[4 * k] && t < r[4 * k + 1] && f >= r[4 * k + 2] && f < r[4 * k + 3]);
    if (hit) xb[i] = 0.f;
  }
}
extern "C" int32_t otr_spec_mask(float* x, const int32_t* ranges, int32_t B, int32_t NR, int32_t T, int32_t F, void* stream) {
  OTR_REQUIRE(x && ranges, "spec_mask: null pointer");
  OTR_REQUIRE(B >= 0 && NR >= 0 && NR <= 64 && T > 0 && F > 0, "spec_mask: bad shape B=%d NR=%d T=%d F=%d", B, NR, T, F);
  if (B == 0 || NR == 0) return 0;
  unsigned gx = (unsigned)(((int64_t)T * F + 255) / 256);
  if (gx > 512) gx = 512;
  hipLaunchKernelGGL(spec_mask_kernel, dim3(gx, (unsigned)B), dim3(256), 0, (hipStream_t)stream, x, ranges, NR, T, F);
  return otr_check_launch("spec_mask");
}
