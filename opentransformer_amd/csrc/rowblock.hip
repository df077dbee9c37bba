// Row-block projections of the attention sub-layers (module/attention.py:62-75 qvk_proj / output_proj, :120-140 q_proj /
// output_proj; encoder/transformer.py:47-56, decoder/transformer.py:58-80 residual + LayerNorm around them):
//
//   otr_rb_linear     :  out = x16 . W^T (+ bias) (+ skip)                       the q|k|v projection and its input gradient
//   otr_proj_ln_fwd   :  y = LayerNorm(x + dropout(c16 . W^T + b))               output projection + residual + LayerNorm
//   otr_ln_bwd_proj   :  LayerNorm backward (dx, da, affine / bias partials) + dc = da . W   in ONE launch
//
// d_model = 256 makes these GEMMs skinny (N, K in {256, 768}): on the tile GEMM a launch is 126 workgroups of 8 k-steps,
// 12-25 us for 1-3 GFLOP, plus a split-K reduce and a separate add+LayerNorm launch per sub-layer.  Here, as in
// ffn_fused.hip, a workgroup owns 32 rows of the residual stream (249 workgroups at B = 32 x 249 frames), its 4 waves own
// different 32-wide groups of output columns, weights are pre-packed fragment-major (otr_pack_frags, perm 0) and stream
// L2 -> VGPR through a register ring with no barrier in the loop, activations sit in LDS as MFMA B operands, and the
// epilogue (bias / dropout / residual / LayerNorm, or the 16-bit store) runs on whole rows.  Bound: the packed weight
// (128 KB for 256 x 256, 384 KB for 768 x 256) into every CU at ~22 B/clk: 3 / 8 us.
#include "common.h"
#include "ln_pro.h"

namespace {

constexpr int RB = 32;        // rows per workgroup
constexpr int RB_PD = 16;     // weight fragments in flight per wave (1 KiB each; 32 measured no faster)

// 32 rows x K 16-bit activations -> LDS as 16-byte chunks, chunk index XOR (row & 15): the B-operand read of lane
// (m = lane&31, hi) -- chunk (2*ks + hi) of row m -- is then bank-conflict free for ds_read_b128 (as ffn_fused.hip)
template <int K, int NTHR = 256> __device__ __forceinline__ void rb_stage_rows(uint4* dst, const uint16_t* src, int64_t ld, int row0, int M, int tid) {
  constexpr int CPR = K / 8, PER = RB * CPR / NTHR;       // chunks per thread: 4 (K = 256) / 12 (K = 768), all loads before any store
  uint4 v[PER];
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const int i = tid + j * NTHR, r = i / CPR, ch = i % CPR;
    v[j] = ld_global_b128(src + (int64_t)min(row0 + r, M - 1) * ld + ch * 8);
  }
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const int i = tid + j * NTHR, r = i / CPR, ch = i % CPR;
    dst[r * CPR + (ch ^ (r & 15))] = v[j];
  }
}
template <int K> __device__ __forceinline__ uint4 rb_frag_b(const uint4* rows, int m, int hi, int ks) {
  return rows[m * (K / 8) + ((2 * ks + hi) ^ (m & 15))];
}

// acc[i] (i < TPW) += W[tile (t0 + i)] . x^T: the wave walks the contraction outermost (one LDS operand per k-step, used by
// all its TPW tiles) and streams fragment (tile, ks) = pw[(tile * NKS + ks) * 64 + lane] through a RB_PD-deep ring.
// rb_fill starts the stream (call it before the activations are staged: the first fragments travel meanwhile).
template <int K, int TPW> struct RbStream {
  static constexpr int NKS = K / 16, STEPS = NKS * TPW;
  static_assert(STEPS % RB_PD == 0, "ring slots are compile-time constants");
  const uint4* P;
  int rot;                                     // workgroups walk the contraction from different starting points: 31 workgroups
  uint4 ring[RB_PD];                           // of an XCD asking its L2 for the same line at the same time serialise on one channel
  __device__ __forceinline__ int ks_of(int s) const {
    const int k = s / TPW + rot;
    return k >= NKS ? k - NKS : k;
  }
  __device__ __forceinline__ const uint4* fptr(int s) const { return P + (int64_t)((s % TPW) * NKS + ks_of(s)) * 64; }
  __device__ __forceinline__ void fill(const uint4* pw, int t0, int lane) {
    rot = (int)((blockIdx.x >> 3) % (unsigned)NKS);
    P = pw + (int64_t)t0 * NKS * 64 + lane;
#pragma unroll
    for (int s = 0; s < RB_PD; ++s) ring[s] = ld_global_b128(fptr(s));
    __builtin_amdgcn_sched_barrier(0);
  }
  __device__ __forceinline__ void run(f32x16 (&acc)[TPW], const uint4* xs, int lane) {
    const int m = lane & 31, hi = lane >> 5;
    uint4 xb;
#pragma clang loop unroll(full)
    for (int s = 0; s < STEPS; ++s) {
      if (s % TPW == 0) xb = rb_frag_b<K>(xs, m, hi, ks_of(s));
      const uint4 w = ring[s % RB_PD];
      mma32(acc[s % TPW], w, xb);
      if (s + RB_PD < STEPS) ring[s % RB_PD] = ld_global_b128(fptr(s + RB_PD));
      __builtin_amdgcn_sched_barrier(0);      // keep the loads RB_PD steps ahead (the scheduler otherwise sinks them next to their use)
    }
  }
};

// accumulator tile (out columns 8q + 4hi + (r&3), q = r>>2, of row m = lane&31) -> red[m][col0 + ...] (row stride RS floats)
template <int RS> __device__ __forceinline__ void rb_put_tile(float* red, const f32x16& a, int col0, int lane) {
  const int m = lane & 31, hi = lane >> 5;
#pragma unroll
  for (int q = 0; q < 4; ++q)
    *reinterpret_cast<float4*>(red + m * RS + col0 + 8 * q + 4 * hi) = make_float4(a[4 * q], a[4 * q + 1], a[4 * q + 2], a[4 * q + 3]);
}

// ------------------------------------------------------------------------------------------------ a touch for the launch that follows
// Inside the step every launch finds its operands in HBM only.  Where that hurts the NEXT launch (scattered or latency-critical first
// reads: the FFN backward launch's packs, the attention backward launch's saved q|k|v and context) the launch before it touches
// those ranges -- one dword per 64 bytes, issued behind its main loop, never consumed -- so the lines sit in the memory-side cache
// in time.  The destination registers stay OWNED until the closing wait: hipcc does not know the asm is a load (a register it
// believed dead was reused and the returning load overwrote a live value: 1 % wrong dx in the first build).
struct RbTouch { const unsigned char* p[2]; int64_t lines[2]; };
constexpr int RB_TOUCH_PER = 10;                                      // loads per thread at most (grid x threads x 6 x 64 B >= 16 MB + 3 MB here)
__device__ __forceinline__ void rb_touch_issue(const RbTouch& t, uint32_t (&sink)[RB_TOUCH_PER], int tid, int nthr) {
  if (!t.p[0]) return;
  const int64_t total = t.lines[0] + t.lines[1];
  const int64_t per = min((total + gridDim.x - 1) / gridDim.x, (int64_t)RB_TOUCH_PER * nthr), l0 = (int64_t)blockIdx.x * per;
#pragma unroll
  for (int k = 0; k < RB_TOUCH_PER; ++k) {
    const int64_t l = l0 + tid + (int64_t)k * nthr;
    if (l < l0 + per && l < total) {
      const unsigned char* a = l < t.lines[0] ? t.p[0] + l * 64 : t.p[1] + (l - t.lines[0]) * 64;
      asm volatile("global_load_dword %0, %1, off" : "+v"(sink[k]) : "v"(a) : "memory");
    }
  }
}
__device__ __forceinline__ void rb_touch_wait(uint32_t (&sink)[RB_TOUCH_PER]) {
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(sink[0]), "+v"(sink[1]), "+v"(sink[2]), "+v"(sink[3]), "+v"(sink[4]), "+v"(sink[5]), "+v"(sink[6]), "+v"(sink[7]),
               "+v"(sink[8]), "+v"(sink[9])::"memory");
}

// ------------------------------------------------------------------------------------------------ plain projection
struct RbLinArgs {
  const uint16_t* x16;     // [M, K] 16-bit rows, row stride ldx
  const uint4* pw;         // W packed: rows = N outputs, contraction = K, perm 0
  const float* bias;       // [N] or NULL
  const float* skip;       // f32 [M, N] (row stride lds) or NULL: out = skip + ...
  void* out;               // [M, N] f32 or 16-bit, row stride ldo
  int64_t ldx, lds, ldo;
  int M, N, out_h16;
};

// NSPLIT workgroups share a row block, each takes N / NSPLIT output columns (blockIdx.y): with two resident workgroups a CU has 8
// waves in flight and ingests ~33 B/clk instead of ~22 for the same weight bytes
template <int K, int TPW, int NSPLIT = 1>
__global__ __launch_bounds__(256, NSPLIT) void rb_linear_kernel(RbLinArgs p) {
  constexpr int N = 128 * TPW, RS = N + 4;
  const int cbase = (int)blockIdx.y * N;                         // first output column of this workgroup
  __shared__ __attribute__((aligned(16))) unsigned char smem[RB * K * 2 + RB * RS * 4];
  uint4* xs = reinterpret_cast<uint4*>(smem);
  float* red = reinterpret_cast<float*>(smem + RB * K * 2);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int row0 = blockIdx.x * RB;
  RbStream<K, TPW> ws;
  ws.fill(p.pw, (int)blockIdx.y * 4 * TPW + wid * TPW, lane);
  rb_stage_rows<K>(xs, p.x16, p.ldx, row0, p.M, tid);
  __syncthreads();
  f32x16 acc[TPW];
#pragma unroll
  for (int i = 0; i < TPW; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  ws.run(acc, xs, lane);
#pragma unroll
  for (int i = 0; i < TPW; ++i) rb_put_tile<RS>(red, acc[i], (wid * TPW + i) * 32, lane);
  __syncthreads();
  // whole rows: wave w owns rows 8w .. 8w+7, a lane 4 consecutive columns per pass of 256
#pragma unroll
  for (int rr = 0; rr < 8; ++rr) {
    const int r = wid * 8 + rr;
    const int64_t row = (int64_t)row0 + r;
    if (row >= p.M) break;                                          // wave-uniform
#pragma unroll
    for (int c0 = 0; c0 < N; c0 += 256) {
      const int lc = c0 + lane * 4;
      if (N % 256 != 0 && lc >= N) break;
      const int col = cbase + lc;
      float4 v = *reinterpret_cast<const float4*>(red + r * RS + lc);
      if (p.bias) {
        const float4 b = *reinterpret_cast<const float4*>(p.bias + col);
        v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
      }
      if (p.skip) {
        const float4 s = *reinterpret_cast<const float4*>(p.skip + row * p.lds + col);
        v.x += s.x; v.y += s.y; v.z += s.z; v.w += s.w;
      }
      if (p.out_h16) *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(p.out) + row * p.ldo + col) = make_uint2(pack2h(v.x, v.y), pack2h(v.z, v.w));
      else *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.out) + row * p.ldo + col) = v;
    }
  }
}

// The q|k|v projection behind a split FFN in SLAB mode (ffn3.hip): the FFN launch leaves four 16-bit partial slabs instead of
// exchanging them, and this launch finishes  y = LayerNorm(xres + dropout(sum of the slabs + b_2))  for its 32 rows in its prologue
// (ln_pro.h; both column-half workgroups of a row block do, the first one writes y / y16 / z / mean / rstd), then projects.
struct RbLinLnArgs {
  DlLn ln;
  const uint4* pw; const float* bias;
  void* out; int64_t ldo;
  int M, out_h16;
};
// NW = 8, NSPLIT = 1: one 8-wave workgroup per row block takes all 768 columns -- the prologue's ~100 KiB (residual rows + four slabs)
// travel once per row block instead of once per column half (per-CU ingest is what bounds these launches).
template <int TPW, int NSPLIT, int NW>
__global__ __launch_bounds__(NW * 64, NSPLIT) void rb_linear_ln_kernel(RbLinLnArgs p) {
  constexpr int K = 256, N = 32 * NW * TPW, RS = N + 4, RPW = RB / NW;
  const int cbase = (int)blockIdx.y * N;
  __shared__ __attribute__((aligned(16))) unsigned char smem[RB * K * 2 + RB * RS * 4];
  uint4* xs = reinterpret_cast<uint4*>(smem);
  float* red = reinterpret_cast<float*>(smem + RB * K * 2);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int row0 = blockIdx.x * RB;
  const int nrows = min(RB, p.M - row0);
  DlPro<NW, 4> pro;                                                // its loads first (vmcnt retires in order), then the weight stream
  pro.issue(p.ln, row0, nrows, tid);
  RbStream<K, TPW> ws;
  ws.fill(p.pw, (int)blockIdx.y * NW * TPW + wid * TPW, lane);
  pro.finish(p.ln, nrows, blockIdx.y == 0, [&](int r, int ch, const uint4& v) { xs[r * (K / 8) + (ch ^ (r & 15))] = v; }, tid);
  __syncthreads();
  f32x16 acc[TPW];
#pragma unroll
  for (int i = 0; i < TPW; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  ws.run(acc, xs, lane);
#pragma unroll
  for (int i = 0; i < TPW; ++i) rb_put_tile<RS>(red, acc[i], (wid * TPW + i) * 32, lane);
  __syncthreads();
#pragma unroll
  for (int rr = 0; rr < RPW; ++rr) {
    const int r = wid * RPW + rr;
    const int64_t row = (int64_t)row0 + r;
    if (row >= p.M) break;                                          // wave-uniform
#pragma unroll
    for (int c0 = 0; c0 < N; c0 += 256) {
      const int lc = c0 + lane * 4;
      if (N % 256 != 0 && lc >= N) break;
      const int col = cbase + lc;
      float4 v = *reinterpret_cast<const float4*>(red + r * RS + lc);
      if (p.bias) {
        const float4 b = *reinterpret_cast<const float4*>(p.bias + col);
        v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
      }
      if (p.out_h16) *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(p.out) + row * p.ldo + col) = make_uint2(pack2h(v.x, v.y), pack2h(v.z, v.w));
      else *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.out) + row * p.ldo + col) = v;
    }
  }
}

// ------------------------------------------------------------------------------------------------ projection + residual + LayerNorm
struct ProjLnArgs {
  const float* x;          // residual stream [M, 256] f32
  const uint16_t* c16;     // branch input [M, 256] 16-bit (attention context), row stride ldc
  const uint4* pw;         // W packed (rows = 256 outputs, contraction = 256)
  const float* bias; const float* gamma; const float* beta; const uint64_t* seed;
  float* y; uint16_t* y16; float* z; float* mean; float* rstd;
  int64_t ldc;
  int M;
  float eps, p_drop;
  uint64_t rng_offset;
  RbTouch touch;           // see RbTouch: the packs of the FFN launch that follows (otr_touch_hint)
};

// NW = 4 or 8 waves: with 8 (two per SIMD) the CU has twice the loads in flight and ingests the 128 KB of packed weights at ~33
// instead of ~22 B/clk; a wave then owns one column tile and four rows of the epilogue
template <int NW>
__global__ __launch_bounds__(64 * NW, NW / 4) void proj_ln_fwd_kernel(ProjLnArgs p) {
  constexpr int D = 256, RS = D + 4, TPW = 8 / NW, RPW = RB / NW;
  __shared__ __attribute__((aligned(16))) unsigned char smem[RB * D * 2 + RB * RS * 4];
  uint4* xs = reinterpret_cast<uint4*>(smem);
  float* red = reinterpret_cast<float*>(smem + RB * D * 2);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int row0 = blockIdx.x * RB;
  RbStream<D, TPW> ws;
  ws.fill(p.pw, wid * TPW, lane);
  rb_stage_rows<D, 64 * NW>(xs, p.c16, p.ldc, row0, p.M, tid);
  __syncthreads();
  f32x16 acc[TPW];
#pragma unroll
  for (int i = 0; i < TPW; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  // the residual rows of this wave travel while the GEMM runs
  const int col = lane * 4;
  float4 xr[RPW];
#pragma unroll
  for (int i = 0; i < RPW; ++i) {
    const int64_t row = min((int64_t)row0 + wid * RPW + i, (int64_t)p.M - 1);
    xr[i] = *reinterpret_cast<const float4*>(p.x + row * D + col);
  }
  const bool drop = p.p_drop > 0.f;
  const uint64_t seed = drop ? *p.seed : 0;
  const uint32_t thr = drop ? (uint32_t)fminf(p.p_drop * 4294967296.f, 4294967295.f) : 0;
  const float inv_keep = drop ? 1.f / (1.f - p.p_drop) : 1.f;
  float4 bb = make_float4(0.f, 0.f, 0.f, 0.f);
  if (p.bias) bb = *reinterpret_cast<const float4*>(p.bias + col);
  const float4 gm = *reinterpret_cast<const float4*>(p.gamma + col);
  const float4 bt = *reinterpret_cast<const float4*>(p.beta + col);
  ws.run(acc, xs, lane);
  uint32_t sink[RB_TOUCH_PER] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
  rb_touch_issue(p.touch, sink, tid, 64 * NW);
#pragma unroll
  for (int i = 0; i < TPW; ++i) rb_put_tile<RS>(red, acc[i], (wid * TPW + i) * 32, lane);
  __syncthreads();
  // the 8 rows of a wave are normalised TOGETHER: their 2 x 6 butterfly steps are independent, so the cross-lane latency
  // is paid 12 times per wave instead of 96 (one row after the other it was ~5 us of this kernel)
  float v[RPW][4], sm[RPW], qq[RPW];
#pragma unroll
  for (int i = 0; i < RPW; ++i) {
    const int r = wid * RPW + i;
    const int64_t row = min((int64_t)row0 + r, (int64_t)p.M - 1);
    const float4 t = *reinterpret_cast<const float4*>(red + r * RS + col);
    const float bv[4] = {t.x + bb.x, t.y + bb.y, t.z + bb.z, t.w + bb.w};
    const float xv[4] = {xr[i].x, xr[i].y, xr[i].z, xr[i].w};
    sm[i] = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float sc = 1.f;
      if (drop) sc = otr_rand32(seed, p.rng_offset + (uint64_t)(row * D + col + e)) >= thr ? inv_keep : 0.f;
      v[i][e] = xv[e] + bv[e] * sc;
      sm[i] += v[i][e];
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1)
#pragma unroll
    for (int i = 0; i < RPW; ++i) sm[i] += __shfl_xor(sm[i], o);
#pragma unroll
  for (int i = 0; i < RPW; ++i) {
    sm[i] *= (1.f / D);
    qq[i] = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) { const float d = v[i][e] - sm[i]; qq[i] += d * d; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1)
#pragma unroll
    for (int i = 0; i < RPW; ++i) qq[i] += __shfl_xor(qq[i], o);
  const float g4[4] = {gm.x, gm.y, gm.z, gm.w}, b4[4] = {bt.x, bt.y, bt.z, bt.w};
#pragma unroll
  for (int i = 0; i < RPW; ++i) {
    const int64_t row = (int64_t)row0 + wid * RPW + i;
    if (row >= p.M) break;                                          // wave-uniform
    const float mean = sm[i], rstd = rsqrtf(qq[i] * (1.f / D) + p.eps);
    if (p.z) *reinterpret_cast<float4*>(p.z + row * D + col) = make_float4(v[i][0], v[i][1], v[i][2], v[i][3]);
    float o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = (v[i][e] - mean) * rstd * g4[e] + b4[e];
    *reinterpret_cast<float4*>(p.y + row * D + col) = make_float4(o[0], o[1], o[2], o[3]);
    if (p.y16) *reinterpret_cast<uint2*>(p.y16 + row * D + col) = make_uint2(pack2h(o[0], o[1]), pack2h(o[2], o[3]));
    if (lane == 0) { p.mean[row] = mean; p.rstd[row] = rstd; }
  }
  rb_touch_wait(sink);
}

// ------------------------------------------------------------------------------------------------ LayerNorm backward + input gradient of the projection
struct LnBwdProjArgs {
  const uint16_t* slabs;   // NULL, or [nslab][M][256] 16-bit: shares of the gradient ON TOP of dy (the split FFN's backward launch in slab
  int nslab;               // mode leaves its four input-gradient shares; dy is then the skip-path part)
  const float* dy;         // [M, 256] f32: gradient of the LayerNorm output
  const float* z;          // saved pre-norm sum x + dropout(branch)
  const float* mean; const float* rstd; const float* gamma; const uint64_t* seed;
  const uint4* pwt;        // W^T packed: rows = 256 inputs of the projection, contraction = its 256 outputs
  float* dx;               // [M, 256] f32: gradient of the residual input (= d z)
  uint16_t* da16;          // [M, 256] 16-bit: gradient of the projection output (dropout applied) -- the weight-gradient operand
  uint16_t* dc16;          // [M, 256] 16-bit, row stride ldc: gradient of the projection input (attention context)
  float* partial;          // [gridDim.x][3][256]: this workgroup's sums of dgamma | dbeta | da (bias gradient)
  int64_t ldc;
  int M;
  float p_drop;
  uint64_t rng_offset;
  RbTouch touch;           // see RbTouch: the attention backward launch's saved q|k|v and context (otr_touch_hint)
};

template <int NW>     // 4 or 8 waves (proj_ln_fwd_kernel)
__global__ __launch_bounds__(64 * NW, NW / 4) void ln_bwd_proj_kernel(LnBwdProjArgs p) {
  constexpr int D = 256, RS = D + 4, CPR = D / 8, TPW = 8 / NW, RPW = RB / NW;
  __shared__ __attribute__((aligned(16))) unsigned char smem[RB * D * 2 + RB * RS * 4];
  uint4* xs = reinterpret_cast<uint4*>(smem);
  float* red = reinterpret_cast<float*>(smem + RB * D * 2);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int row0 = blockIdx.x * RB;
  const bool drop = p.p_drop > 0.f;
  const uint64_t seed = drop ? *p.seed : 0;
  const uint32_t thr = drop ? (uint32_t)fminf(p.p_drop * 4294967296.f, 4294967295.f) : 0;
  const float inv_keep = drop ? 1.f / (1.f - p.p_drop) : 1.f;
  const int col = lane * 4;
  RbStream<D, TPW> ws;
  ws.fill(p.pwt, wid * TPW, lane);                                   // W^T starts to travel under the LayerNorm arithmetic
  const float4 gm4 = *reinterpret_cast<const float4*>(p.gamma + col);
  const float gam[4] = {gm4.x, gm4.y, gm4.z, gm4.w};
  // ---- LayerNorm backward on this wave's 8 rows (all loads first)
  float4 dyv[RPW], zv[RPW];
  float mean[RPW], rstd[RPW];
#pragma unroll
  for (int i = 0; i < RPW; ++i) {
    const int64_t row = min((int64_t)row0 + wid * RPW + i, (int64_t)p.M - 1);
    dyv[i] = *reinterpret_cast<const float4*>(p.dy + row * D + col);
    zv[i] = *reinterpret_cast<const float4*>(p.z + row * D + col);
    mean[i] = p.mean[row]; rstd[i] = p.rstd[row];
  }
  if (p.nslab > 0) {
    uint2 sl[4][RPW];
#pragma unroll
    for (int s_ = 0; s_ < 4; ++s_)
#pragma unroll
      for (int i = 0; i < RPW; ++i) {
        const int64_t row = min((int64_t)row0 + wid * RPW + i, (int64_t)p.M - 1);
        sl[s_][i] = *reinterpret_cast<const uint2*>(p.slabs + ((int64_t)min(s_, p.nslab - 1) * p.M + row) * D + col);
      }
#pragma unroll
    for (int s_ = 0; s_ < 4; ++s_) {
      const float live = s_ < p.nslab ? 1.f : 0.f;
#pragma unroll
      for (int i = 0; i < RPW; ++i) {
        dyv[i].x += live * h2f_lo(sl[s_][i].x); dyv[i].y += live * h2f_hi(sl[s_][i].x);
        dyv[i].z += live * h2f_lo(sl[s_][i].y); dyv[i].w += live * h2f_hi(sl[s_][i].y);
      }
    }
  }
  float dg[4] = {0.f, 0.f, 0.f, 0.f}, db[4] = {0.f, 0.f, 0.f, 0.f}, dab[4] = {0.f, 0.f, 0.f, 0.f};
  float d4[RPW][4], z4[RPW][4], s1[RPW], s2[RPW];
#pragma unroll
  for (int i = 0; i < RPW; ++i) {
    const float live = (int64_t)row0 + wid * RPW + i < p.M ? 1.f : 0.f;
    const float dd[4] = {dyv[i].x * live, dyv[i].y * live, dyv[i].z * live, dyv[i].w * live};
    const float zz[4] = {zv[i].x, zv[i].y, zv[i].z, zv[i].w};
    s1[i] = 0.f; s2[i] = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      d4[i][e] = dd[e];
      z4[i][e] = (zz[e] - mean[i]) * rstd[i];
      const float g = dd[e] * gam[e];
      s1[i] += g; s2[i] += g * z4[i][e];
      dg[e] += dd[e] * z4[i][e];
      db[e] += dd[e];
    }
  }
  // the rows' reductions run together (16 independent butterflies: the cross-lane latency is paid 6 times, not 96)
#pragma unroll
  for (int o = 32; o > 0; o >>= 1)
#pragma unroll
    for (int i = 0; i < RPW; ++i) { s1[i] += __shfl_xor(s1[i], o); s2[i] += __shfl_xor(s2[i], o); }
#pragma unroll
  for (int i = 0; i < RPW; ++i) {
    const int r = wid * RPW + i;
    const int64_t row = (int64_t)row0 + r;
    const float m1 = s1[i] * (1.f / D), m2 = s2[i] * (1.f / D);
    float dz[4], da[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      dz[e] = rstd[i] * (d4[i][e] * gam[e] - m1 - z4[i][e] * m2);
      const float sc = drop ? (otr_rand32(seed, p.rng_offset + (uint64_t)(row * D + col + e)) >= thr ? inv_keep : 0.f) : 1.f;
      da[e] = dz[e] * sc;
      dab[e] += da[e];
    }
    const uint2 h = make_uint2(pack2h(da[0], da[1]), pack2h(da[2], da[3]));
    // B-operand image: 8 bytes = half of chunk (lane >> 1) of row r
    reinterpret_cast<uint2*>(xs + r * CPR + ((lane >> 1) ^ (r & 15)))[lane & 1] = h;
    if (row < p.M) {
      if (p.dx) *reinterpret_cast<float4*>(p.dx + row * D + col) = make_float4(dz[0], dz[1], dz[2], dz[3]);
      if (p.da16) *reinterpret_cast<uint2*>(p.da16 + row * D + col) = h;
    }
  }
  // ---- the workgroup's affine / bias partial sums (the grouped column-sum launch adds the workgroups up)
  float* part = reinterpret_cast<float*>(red);                      // [3][4 waves][256]
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    part[(0 * NW + wid) * D + col + e] = dg[e];
    part[(1 * NW + wid) * D + col + e] = db[e];
    part[(2 * NW + wid) * D + col + e] = dab[e];
  }
  __syncthreads();
  if (p.partial) {
    float* prow = p.partial + (int64_t)blockIdx.x * 3 * D;
    for (int c = tid; c < 3 * D; c += 64 * NW) {
      const int k = c >> 8, cc = c & 255;
      float t_ = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) t_ += part[(k * NW + w) * D + cc];
      prow[c] = t_;
    }
  }
  __syncthreads();                                                  // `red` is reused for the output tiles
  // ---- dc = da . W: output column = input feature of the projection
  f32x16 acc[TPW];
#pragma unroll
  for (int i = 0; i < TPW; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  ws.run(acc, xs, lane);
  uint32_t sink[RB_TOUCH_PER] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
  rb_touch_issue(p.touch, sink, tid, 64 * NW);                       // nothing below loads anything
#pragma unroll
  for (int i = 0; i < TPW; ++i) rb_put_tile<RS>(red, acc[i], (wid * TPW + i) * 32, lane);
  __syncthreads();
#pragma unroll
  for (int i = 0; i < RPW; ++i) {
    const int r = wid * RPW + i;
    const int64_t row = (int64_t)row0 + r;
    if (row >= p.M) break;
    const float4 v = *reinterpret_cast<const float4*>(red + r * RS + col);
    *reinterpret_cast<uint2*>(p.dc16 + row * p.ldc + col) = make_uint2(pack2h(v.x, v.y), pack2h(v.z, v.w));
  }
  rb_touch_wait(sink);
}

// ------------------------------------------------------------------------------------------------ input gradient of a projection + the LayerNorm backward it feeds
// dy = skip + g16 . W is the gradient of a LayerNorm OUTPUT y = LN(z), z = x + dropout(a) (the q|k|v projection of layer l reads
// the output of layer l-1's FFN sub-layer: encoder/transformer.py:47-63): the rows are complete in this workgroup, so the
// LayerNorm backward runs in the epilogue instead of a launch of its own that would read dy back -- dy itself is never stored.
// Outputs as otr_ln_bwd_proj: dx = d z (the skip-connection gradient of that sub-layer), da16 = the dropout-masked branch gradient,
// partial = this workgroup's sums of dgamma | dbeta | da.
struct RbLinLnBwdArgs {
  const uint16_t* g16;     // [M, K] 16-bit rows (row stride ldg): gradient of the projection's output
  const uint4* pw;         // W^T packed: rows = 256 inputs of the projection, contraction = K
  const float* skip;       // f32 [M, 256] (row stride lds) or NULL
  const float* z; const float* mean; const float* rstd; const float* gamma; const uint64_t* seed;
  float* dx; uint16_t* da16; float* partial;
  int64_t ldg, lds;
  int M;
  float p_drop;
  uint64_t rng_offset;
  // What the NEXT launch streams first and would otherwise find nowhere but in HBM (the split FFN's backward launch: its two
  // input-gradient packs, 3 MB, the second read in 8 KiB runs 256 KiB apart): see RbTouch (tools/ffn3_prefetch_probe.py: 50.8 -> 47.0
  // us for the backward launch on cold operands).
  RbTouch touch;
};

template <int K, int NW>
__global__ __launch_bounds__(64 * NW, NW / 4) void rb_linear_ln_bwd_kernel(RbLinLnBwdArgs p) {
  constexpr int D = 256, RS = D + 4, TPW = 8 / NW, RPW = RB / NW;
  __shared__ __attribute__((aligned(16))) unsigned char smem[RB * K * 2 + RB * RS * 4];
  uint4* xs = reinterpret_cast<uint4*>(smem);
  float* red = reinterpret_cast<float*>(smem + RB * K * 2);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int row0 = blockIdx.x * RB;
  const int col = lane * 4;
  RbStream<K, TPW> ws;
  ws.fill(p.pw, wid * TPW, lane);
  rb_stage_rows<K, 64 * NW>(xs, p.g16, p.ldg, row0, p.M, tid);
  // everything of the LayerNorm backward that does not depend on the product travels under the GEMM
  const bool drop = p.p_drop > 0.f;
  const uint64_t seed = drop ? *p.seed : 0;
  const uint32_t thr = drop ? (uint32_t)fminf(p.p_drop * 4294967296.f, 4294967295.f) : 0;
  const float inv_keep = drop ? 1.f / (1.f - p.p_drop) : 1.f;
  const float4 gm4 = *reinterpret_cast<const float4*>(p.gamma + col);
  const float gam[4] = {gm4.x, gm4.y, gm4.z, gm4.w};
  float4 skv[RPW], zv[RPW];
  float mean[RPW], rstd[RPW];
#pragma unroll
  for (int i = 0; i < RPW; ++i) {
    const int64_t row = min((int64_t)row0 + wid * RPW + i, (int64_t)p.M - 1);
    skv[i] = p.skip ? *reinterpret_cast<const float4*>(p.skip + row * p.lds + col) : make_float4(0.f, 0.f, 0.f, 0.f);
    zv[i] = *reinterpret_cast<const float4*>(p.z + row * D + col);
    mean[i] = p.mean[row]; rstd[i] = p.rstd[row];
  }
  __syncthreads();
  f32x16 acc[TPW];
#pragma unroll
  for (int i = 0; i < TPW; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  ws.run(acc, xs, lane);
  uint32_t sink[RB_TOUCH_PER] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
  rb_touch_issue(p.touch, sink, tid, 64 * NW);                       // nothing below loads anything
#pragma unroll
  for (int i = 0; i < TPW; ++i) rb_put_tile<RS>(red, acc[i], (wid * TPW + i) * 32, lane);
  __syncthreads();
  // ---- LayerNorm backward on this wave's 8 rows (ln_bwd_proj_kernel), dy = the product + skip
  float dg[4] = {0.f, 0.f, 0.f, 0.f}, db[4] = {0.f, 0.f, 0.f, 0.f}, dab[4] = {0.f, 0.f, 0.f, 0.f};
  float d4[RPW][4], z4[RPW][4], s1[RPW], s2[RPW];
#pragma unroll
  for (int i = 0; i < RPW; ++i) {
    const int r = wid * RPW + i;
    const float live = (int64_t)row0 + r < p.M ? 1.f : 0.f;
    const float4 v = *reinterpret_cast<const float4*>(red + r * RS + col);
    const float dd[4] = {(v.x + skv[i].x) * live, (v.y + skv[i].y) * live, (v.z + skv[i].z) * live, (v.w + skv[i].w) * live};
    const float zz[4] = {zv[i].x, zv[i].y, zv[i].z, zv[i].w};
    s1[i] = 0.f; s2[i] = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      d4[i][e] = dd[e];
      z4[i][e] = (zz[e] - mean[i]) * rstd[i];
      const float g = dd[e] * gam[e];
      s1[i] += g; s2[i] += g * z4[i][e];
      dg[e] += dd[e] * z4[i][e];
      db[e] += dd[e];
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1)
#pragma unroll
    for (int i = 0; i < RPW; ++i) { s1[i] += __shfl_xor(s1[i], o); s2[i] += __shfl_xor(s2[i], o); }
#pragma unroll
  for (int i = 0; i < RPW; ++i) {
    const int64_t row = (int64_t)row0 + wid * RPW + i;
    const float m1 = s1[i] * (1.f / D), m2 = s2[i] * (1.f / D);
    float dz[4], da[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      dz[e] = rstd[i] * (d4[i][e] * gam[e] - m1 - z4[i][e] * m2);
      const float sc = drop ? (otr_rand32(seed, p.rng_offset + (uint64_t)(row * D + col + e)) >= thr ? inv_keep : 0.f) : 1.f;
      da[e] = dz[e] * sc;
      dab[e] += da[e];
    }
    if (row < p.M) {
      *reinterpret_cast<float4*>(p.dx + row * D + col) = make_float4(dz[0], dz[1], dz[2], dz[3]);
      *reinterpret_cast<uint2*>(p.da16 + row * D + col) = make_uint2(pack2h(da[0], da[1]), pack2h(da[2], da[3]));
    }
  }
  __syncthreads();                                                  // every wave has read its rows: `red` becomes the partial sums
  float* part = red;                                                // [3][4 waves][256]
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    part[(0 * NW + wid) * D + col + e] = dg[e];
    part[(1 * NW + wid) * D + col + e] = db[e];
    part[(2 * NW + wid) * D + col + e] = dab[e];
  }
  __syncthreads();
  float* prow = p.partial + (int64_t)blockIdx.x * 3 * D;
  for (int c = tid; c < 3 * D; c += 64 * NW) {
    const int k = c >> 8, cc = c & 255;
    float t_ = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) t_ += part[(k * NW + w) * D + cc];
    prow[c] = t_;
  }
  rb_touch_wait(sink);
}

}  // namespace

// ------------------------------------------------------------------------------------------------ C ABI
extern int g_otr_rb_waves8;        // api.hip (otr_debug_set(19, v)): 8-wave workgroups for the 256-column row-block kernels
extern int g_otr_rb_nsplit;        // api.hip (otr_debug_set(18, v)): the 768-column projection on two workgroups per row block
extern "C" int32_t otr_rb_linear(const void* x16, int64_t ldx, const void* w_pack, const float* bias, const float* skip, int64_t lds,
                                 void* out, int32_t out_dtype, int64_t ldo, int64_t M, int32_t N, int32_t K, void* stream) {
  OTR_REQUIRE(x16 && w_pack && out, "rb_linear: null pointer");
  OTR_REQUIRE(M >= 0 && M < (1ll << 31), "rb_linear: bad M");
  OTR_REQUIRE((K == 256 && (N == 256 || N == 768)) || (K == 768 && N == 256), "rb_linear: built for (N, K) in {(256,256), (768,256), (256,768)}, got (%d, %d)", N, K);
  OTR_REQUIRE(out_dtype == OTR_F32 || out_dtype == OTR_H16, "rb_linear: bad out dtype");
  OTR_REQUIRE(ldx >= K && ldx % 8 == 0 && (uintptr_t)x16 % 16 == 0, "rb_linear: x rows must be 16-byte aligned");
  OTR_REQUIRE(ldo >= N && ldo % 4 == 0 && (uintptr_t)out % 16 == 0 && (uintptr_t)w_pack % 16 == 0, "rb_linear: out / pack alignment");
  OTR_REQUIRE(!skip || (lds >= N && lds % 4 == 0 && (uintptr_t)skip % 16 == 0), "rb_linear: skip alignment");
  OTR_REQUIRE(!bias || (uintptr_t)bias % 16 == 0, "rb_linear: bias alignment");
  if (M == 0) return 0;
  RbLinArgs p{};
  p.x16 = reinterpret_cast<const uint16_t*>(x16); p.pw = reinterpret_cast<const uint4*>(w_pack); p.bias = bias; p.skip = skip; p.out = out;
  p.ldx = ldx; p.lds = lds; p.ldo = ldo; p.M = (int)M; p.N = N; p.out_h16 = out_dtype == OTR_H16;
  const dim3 grid((unsigned)((M + RB - 1) / RB));
  hipStream_t s = (hipStream_t)stream;
  if (K == 256 && N == 256) hipLaunchKernelGGL((rb_linear_kernel<256, 2>), grid, dim3(256), 0, s, p);
  else if (K == 256 && g_otr_rb_nsplit) hipLaunchKernelGGL((rb_linear_kernel<256, 3, 2>), dim3(grid.x, 2), dim3(256), 0, s, p);
  else if (K == 256) hipLaunchKernelGGL((rb_linear_kernel<256, 6>), grid, dim3(256), 0, s, p);
  else hipLaunchKernelGGL((rb_linear_kernel<768, 2>), grid, dim3(256), 0, s, p);
  return otr_check_launch("rb_linear");
}

extern "C" int32_t otr_rb_linear_ln(const otr_dec_ln_t* ln, const void* w_pack, const float* bias, void* out, int32_t out_dtype, int64_t ldo,
                                    int64_t M, int32_t N, int32_t K, void* stream) {
  OTR_REQUIRE(ln && w_pack && out, "rb_linear_ln: null pointer");
  OTR_REQUIRE(M > 0 && M < (1ll << 31) && N == 768 && K == 256, "rb_linear_ln: built for the q|k|v projection (N, K) = (768, 256), got (%d, %d)", N, K);
  OTR_REQUIRE(ln->nslab > 0 && ln->nslab <= 4 && ln->xres && ln->slabs && ln->gamma && ln->beta, "rb_linear_ln: bad LayerNorm descriptor");
  OTR_REQUIRE(ln->p_drop >= 0.f && ln->p_drop < 1.f && (ln->p_drop == 0.f || ln->seed), "rb_linear_ln: bad dropout arguments");
  OTR_REQUIRE(out_dtype == OTR_F32 || out_dtype == OTR_H16, "rb_linear_ln: bad out dtype");
  OTR_REQUIRE(ldo >= N && ldo % 4 == 0 && ((uintptr_t)out | (uintptr_t)w_pack | (uintptr_t)bias | (uintptr_t)ln->xres | (uintptr_t)ln->slabs |
                                           (uintptr_t)ln->bias | (uintptr_t)ln->gamma | (uintptr_t)ln->beta | (uintptr_t)ln->y | (uintptr_t)ln->y16 |
                                           (uintptr_t)ln->z) % 16 == 0, "rb_linear_ln: alignment");
  RbLinLnArgs p{};
  p.ln.xres = ln->xres; p.ln.x16 = nullptr; p.ln.slabs = ln->slabs; p.ln.nslab = ln->nslab; p.ln.bias = ln->bias; p.ln.gamma = ln->gamma;
  p.ln.beta = ln->beta; p.ln.seed = ln->seed; p.ln.p_drop = ln->p_drop; p.ln.eps = ln->eps; p.ln.rng_offset = ln->rng_offset;
  p.ln.y = ln->y; p.ln.y16 = (uint16_t*)ln->y16; p.ln.z = ln->z; p.ln.mean = ln->mean; p.ln.rstd = ln->rstd; p.ln.R = M;
  p.pw = reinterpret_cast<const uint4*>(w_pack); p.bias = bias; p.out = out; p.ldo = ldo; p.M = (int)M; p.out_h16 = out_dtype == OTR_H16;
  if (g_otr_rb_waves8) hipLaunchKernelGGL((rb_linear_ln_kernel<3, 1, 8>), dim3((unsigned)((M + RB - 1) / RB), 1), dim3(512), 0, (hipStream_t)stream, p);
  else hipLaunchKernelGGL((rb_linear_ln_kernel<3, 2, 4>), dim3((unsigned)((M + RB - 1) / RB), 2), dim3(256), 0, (hipStream_t)stream, p);
  return otr_check_launch("rb_linear_ln");
}

// otr_touch_hint: ranges the NEXT otr_proj_ln_fwd / otr_ln_bwd_proj / otr_ln_bwd_proj_slabs call of this thread passes to its kernel (RbTouch), then forgotten
static thread_local RbTouch g_touch_hint{};
extern "C" int32_t otr_touch_hint(const void* p0, int64_t bytes0, const void* p1, int64_t bytes1) {
  OTR_REQUIRE(bytes0 >= 0 && bytes1 >= 0 && (p0 || bytes0 == 0) && (p1 || bytes1 == 0), "touch_hint: bad range");
  g_touch_hint = RbTouch{};
  int n = 0;
  if (bytes0 >= 64) { g_touch_hint.p[n] = reinterpret_cast<const unsigned char*>(p0); g_touch_hint.lines[n] = bytes0 / 64; ++n; }
  if (bytes1 >= 64) { g_touch_hint.p[n] = reinterpret_cast<const unsigned char*>(p1); g_touch_hint.lines[n] = bytes1 / 64; ++n; }
  return 0;
}
static RbTouch take_touch_hint() { RbTouch t = g_touch_hint; g_touch_hint = RbTouch{}; return t; }

extern "C" int32_t otr_proj_ln_fwd(const float* x, const void* c16, int64_t ldc, const void* w_pack, const float* bias, const float* gamma,
                                   const float* beta, const uint64_t* seed, float* y, void* y16, float* z, float* mean, float* rstd,
                                   int64_t M, int32_t d_model, float eps, float p_drop, uint64_t rng_offset, void* stream) {
  OTR_REQUIRE(x && c16 && w_pack && gamma && beta && y && mean && rstd, "proj_ln_fwd: null pointer");
  OTR_REQUIRE(d_model == 256, "proj_ln_fwd: built for d_model = 256 (got %d)", d_model);
  OTR_REQUIRE(M >= 0 && M < (1ll << 31) && ldc >= 256 && ldc % 8 == 0 && (uintptr_t)c16 % 16 == 0, "proj_ln_fwd: bad shape / alignment");
  OTR_REQUIRE(p_drop >= 0.f && p_drop < 1.f && (p_drop == 0.f || seed), "proj_ln_fwd: bad dropout");
  if (M == 0) return 0;
  ProjLnArgs p{};
  p.x = x; p.c16 = reinterpret_cast<const uint16_t*>(c16); p.pw = reinterpret_cast<const uint4*>(w_pack); p.bias = bias; p.gamma = gamma;
  p.beta = beta; p.seed = seed; p.y = y; p.y16 = reinterpret_cast<uint16_t*>(y16); p.z = z; p.mean = mean; p.rstd = rstd;
  p.ldc = ldc; p.M = (int)M; p.eps = eps; p.p_drop = p_drop; p.rng_offset = rng_offset;
  p.touch = take_touch_hint();
  if (g_otr_rb_waves8) hipLaunchKernelGGL(proj_ln_fwd_kernel<8>, dim3((unsigned)((M + RB - 1) / RB)), dim3(512), 0, (hipStream_t)stream, p);
  else hipLaunchKernelGGL(proj_ln_fwd_kernel<4>, dim3((unsigned)((M + RB - 1) / RB)), dim3(256), 0, (hipStream_t)stream, p);
  return otr_check_launch("proj_ln_fwd");
}

extern "C" int64_t otr_ln_bwd_proj_partial_rows(int64_t M) { return (M + RB - 1) / RB; }


extern "C" int32_t otr_ln_bwd_proj(const float* dy, const float* z, const float* mean, const float* rstd, const float* gamma,
                                   const uint64_t* seed, const void* wt_pack, float* dx, void* da16, void* dc16, int64_t ldc,
                                   float* partial, int64_t M, int32_t d_model, float p_drop, uint64_t rng_offset, void* stream) {
  OTR_REQUIRE(dy && z && mean && rstd && gamma && wt_pack && dc16, "ln_bwd_proj: null pointer");
  OTR_REQUIRE(d_model == 256, "ln_bwd_proj: built for d_model = 256 (got %d)", d_model);
  OTR_REQUIRE(M >= 0 && M < (1ll << 31) && ldc >= 256 && ldc % 4 == 0 && (uintptr_t)dc16 % 8 == 0, "ln_bwd_proj: bad shape / alignment");
  OTR_REQUIRE(p_drop >= 0.f && p_drop < 1.f && (p_drop == 0.f || seed), "ln_bwd_proj: bad dropout");
  if (M == 0) return 0;
  LnBwdProjArgs p{};
  p.dy = dy; p.z = z; p.mean = mean; p.rstd = rstd; p.gamma = gamma; p.seed = seed; p.pwt = reinterpret_cast<const uint4*>(wt_pack);
  p.dx = dx; p.da16 = reinterpret_cast<uint16_t*>(da16); p.dc16 = reinterpret_cast<uint16_t*>(dc16); p.partial = partial;
  p.ldc = ldc; p.M = (int)M; p.p_drop = p_drop; p.rng_offset = rng_offset;
  p.touch = take_touch_hint();
  if (g_otr_rb_waves8) hipLaunchKernelGGL(ln_bwd_proj_kernel<8>, dim3((unsigned)((M + RB - 1) / RB)), dim3(512), 0, (hipStream_t)stream, p);
  else hipLaunchKernelGGL(ln_bwd_proj_kernel<4>, dim3((unsigned)((M + RB - 1) / RB)), dim3(256), 0, (hipStream_t)stream, p);
  return otr_check_launch("ln_bwd_proj");
}

extern "C" int32_t otr_ln_bwd_proj_slabs(const float* dskip, const void* slabs, int32_t nslab, const float* z, const float* mean, const float* rstd,
                                         const float* gamma, const uint64_t* seed, const void* wt_pack, float* dx, void* da16, void* dc16,
                                         int64_t ldc, float* partial, int64_t M, int32_t d_model, float p_drop, uint64_t rng_offset,
                                         void* stream) {
  OTR_REQUIRE(dskip && slabs && z && mean && rstd && gamma && wt_pack && dc16, "ln_bwd_proj_slabs: null pointer");
  OTR_REQUIRE(nslab > 0 && nslab <= 4 && (uintptr_t)slabs % 8 == 0, "ln_bwd_proj_slabs: 1 .. 4 slabs");
  OTR_REQUIRE(d_model == 256, "ln_bwd_proj_slabs: built for d_model = 256 (got %d)", d_model);
  OTR_REQUIRE(M >= 0 && M < (1ll << 31) && ldc >= 256 && ldc % 4 == 0 && (uintptr_t)dc16 % 8 == 0, "ln_bwd_proj_slabs: bad shape / alignment");
  OTR_REQUIRE(p_drop >= 0.f && p_drop < 1.f && (p_drop == 0.f || seed), "ln_bwd_proj_slabs: bad dropout");
  if (M == 0) return 0;
  LnBwdProjArgs p{};
  p.slabs = reinterpret_cast<const uint16_t*>(slabs); p.nslab = nslab;
  p.dy = dskip; p.z = z; p.mean = mean; p.rstd = rstd; p.gamma = gamma; p.seed = seed; p.pwt = reinterpret_cast<const uint4*>(wt_pack);
  p.dx = dx; p.da16 = reinterpret_cast<uint16_t*>(da16); p.dc16 = reinterpret_cast<uint16_t*>(dc16); p.partial = partial;
  p.ldc = ldc; p.M = (int)M; p.p_drop = p_drop; p.rng_offset = rng_offset;
  p.touch = take_touch_hint();
  if (g_otr_rb_waves8) hipLaunchKernelGGL(ln_bwd_proj_kernel<8>, dim3((unsigned)((M + RB - 1) / RB)), dim3(512), 0, (hipStream_t)stream, p);
  else hipLaunchKernelGGL(ln_bwd_proj_kernel<4>, dim3((unsigned)((M + RB - 1) / RB)), dim3(256), 0, (hipStream_t)stream, p);
  return otr_check_launch("ln_bwd_proj_slabs");
}

extern "C" int32_t otr_rb_linear_ln_bwd_pf(const void* g16, int64_t ldg, const void* wt_pack, const float* skip, int64_t lds, const float* z,
                                           const float* mean, const float* rstd, const float* gamma, const uint64_t* seed, float p_drop,
                                           uint64_t rng_offset, float* dx, void* da16, float* partial, int64_t M, int32_t N, int32_t K,
                                           const void* prefetch, int64_t prefetch_bytes, void* stream);
extern "C" int32_t otr_rb_linear_ln_bwd(const void* g16, int64_t ldg, const void* wt_pack, const float* skip, int64_t lds, const float* z,
                                        const float* mean, const float* rstd, const float* gamma, const uint64_t* seed, float p_drop,
                                        uint64_t rng_offset, float* dx, void* da16, float* partial, int64_t M, int32_t N, int32_t K,
                                        void* stream) {
  return otr_rb_linear_ln_bwd_pf(g16, ldg, wt_pack, skip, lds, z, mean, rstd, gamma, seed, p_drop, rng_offset, dx, da16, partial, M, N, K, nullptr,
                                 0, stream);
}

extern "C" int32_t otr_rb_linear_ln_bwd_pf(const void* g16, int64_t ldg, const void* wt_pack, const float* skip, int64_t lds, const float* z,
                                           const float* mean, const float* rstd, const float* gamma, const uint64_t* seed, float p_drop,
                                           uint64_t rng_offset, float* dx, void* da16, float* partial, int64_t M, int32_t N, int32_t K,
                                           const void* prefetch, int64_t prefetch_bytes, void* stream) {
  OTR_REQUIRE(prefetch_bytes >= 0 && (prefetch || prefetch_bytes == 0), "rb_linear_ln_bwd: bad prefetch range");
  OTR_REQUIRE(g16 && wt_pack && z && mean && rstd && gamma && dx && da16 && partial, "rb_linear_ln_bwd: null pointer");
  OTR_REQUIRE(N == 256 && (K == 256 || K == 768), "rb_linear_ln_bwd: built for N = 256, K in {256, 768} (got %d, %d)", N, K);
  OTR_REQUIRE(M >= 0 && M < (1ll << 31) && ldg >= K && ldg % 8 == 0 && (uintptr_t)g16 % 16 == 0 && (uintptr_t)wt_pack % 16 == 0,
              "rb_linear_ln_bwd: bad shape / alignment");
  OTR_REQUIRE(!skip || (lds >= N && lds % 4 == 0 && (uintptr_t)skip % 16 == 0), "rb_linear_ln_bwd: skip alignment");
  OTR_REQUIRE(p_drop >= 0.f && p_drop < 1.f && (p_drop == 0.f || seed), "rb_linear_ln_bwd: bad dropout");
  if (M == 0) return 0;
  RbLinLnBwdArgs p{};
  p.g16 = reinterpret_cast<const uint16_t*>(g16); p.pw = reinterpret_cast<const uint4*>(wt_pack); p.skip = skip; p.z = z; p.mean = mean;
  p.rstd = rstd; p.gamma = gamma; p.seed = seed; p.dx = dx; p.da16 = reinterpret_cast<uint16_t*>(da16); p.partial = partial;
  p.ldg = ldg; p.lds = lds; p.M = (int)M; p.p_drop = p_drop; p.rng_offset = rng_offset;
  if (prefetch_bytes >= 64) { p.touch.p[0] = reinterpret_cast<const unsigned char*>(prefetch); p.touch.lines[0] = prefetch_bytes / 64; }
  {                                        // a pending otr_touch_hint rides along as the second range
    const RbTouch h = take_touch_hint();
    if (h.p[0]) {
      const int k = p.touch.p[0] ? 1 : 0;
      p.touch.p[k] = h.p[0]; p.touch.lines[k] = h.lines[0];
    }
  }
  const dim3 grid((unsigned)((M + RB - 1) / RB));
  hipStream_t s = (hipStream_t)stream;
  if (g_otr_rb_waves8) {
    if (K == 768) hipLaunchKernelGGL((rb_linear_ln_bwd_kernel<768, 8>), grid, dim3(512), 0, s, p);
    else hipLaunchKernelGGL((rb_linear_ln_bwd_kernel<256, 8>), grid, dim3(512), 0, s, p);
  } else {
    if (K == 768) hipLaunchKernelGGL((rb_linear_ln_bwd_kernel<768, 4>), grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL((rb_linear_ln_bwd_kernel<256, 4>), grid, dim3(256), 0, s, p);
  }
  return otr_check_launch("rb_linear_ln_bwd");
}
