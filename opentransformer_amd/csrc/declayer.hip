// Fused decoder layer for the handful of rows a Transformer decoder sees in training (decoder/transformer.py:47-90,161-183: B x L
// = 32 x 15 = 480 rows at the AISHELL batch) and in the re-forward decode loop.
//
// The unfused path spends ~10 launches per layer forward (q|k|v GEMM, attention, output projection + LayerNorm, q GEMM,
// cross-attention, output projection + LayerNorm, w_1 + GLU, w_2 split-K, reduce, add + LayerNorm), each a 5-9 us chain of dependent
// memory round trips whatever its work: 0.42 ms per step for 3.7 % of the FLOPs.  Everything in a decoder layer is independent
// per UTTERANCE except the weights, so the layer is cut along (utterance group, head) / (row block, hidden slice) instead of along
// operators -- three launches per layer, every workgroup streaming a small share of the weights:
//
//   dec_self_fwd   grid (groups, heads):   LN of the previous sub-layer (sum of its partial slabs + bias + dropout + residual)
//                                          -> q|k|v of ONE head (192 of the 768 columns) -> causal self-attention of that head
//                                          -> the head's share of the output projection (64 of its 256 contraction indices)
//                                          -> partial slab [head][rows][256]
//   dec_cross_fwd  grid (groups, heads):   LN1 (sum of the four head slabs ...) -> q of one head -> cross-attention of that head over
//                                          the utterance's encoder keys / values (flash style, 8 waves take key tiles in turn)
//                                          -> the head's share of the output projection -> partial slab
//   dec_ffn_fwd    grid (row blocks, S):   LN2 (...) -> w_1 + GLU + w_2 on 1/S of the hidden units (the 32-row kernel of
//                                          ffn_fused.hip, weights L2 -> VGPR) -> partial slab [slice][rows][256]
//   dec_ln         the closing LayerNorm of the last layer (sum of the S slabs ...)
//
// A "group" is 32 / L whole utterances (L decoder rows each: one 32-row MFMA operand tile); the partial sums of a sub-layer are
// never added up by a launch of their own: the NEXT launch's prologue reads the slabs of its 32 rows (4 or S x 32 KiB), adds bias,
// dropout mask and residual, normalises, and (the head-0 / slice-0 workgroup) writes y, its 16-bit twin and the LayerNorm's saved
// z / mean / rstd for the backward pass.  Weights are the fragment-major packs the row-block kernels use (otr_pack_frags):
// a head's columns are whole 32-row tiles of the q|k|v pack, and its share of the output projection is 4 of the 16 contraction
// steps of that pack.  d_model = 256, 4 heads of 64, 16-bit operands.
#include "ffn_frag.h"
#include "ln_pro.h"

extern int g_otr_dec_group;     // api.hip (otr_debug_set(23, v)): cap on the utterances per attention workgroup, 0 = dl_group_size's choice

namespace {

constexpr int DL_H = 4, DL_DK = 64;        // DL_D = 256, DL_RB = 32: ln_pro.h
constexpr int DL_YS = DL_D * 2 + 16;      // bytes per row of the [32][256] 16-bit activation image (MFMA B operands)
constexpr int DL_HS = DL_DK * 2 + 16;     // bytes per row of a [32][64] 16-bit head image (q, k, context)
constexpr int DL_VT = DL_RB * 2 + 8;      // bytes per row of a transposed [64][32] value image
constexpr int DL_RS = DL_D + 4;           // floats per row of the fp32 output staging tile
// tuning hook (otr_debug_trace): thread 0 of every workgroup stamps the shader clock into trace[16384 + (kernel id * 256 + workgroup) * 16 + k]
// (the first 16384 entries are the GEMM kernels' region of the same buffer)
#define DL_STAMP(KID, K) do { if (p.trace && threadIdx.x == 0) p.trace[16384 + ((KID) * 256 + ((int)blockIdx.x & 255)) * 16 + (K)] = __builtin_amdgcn_s_memtime(); } while (0)

__device__ __forceinline__ uint4 dl_frag(const unsigned char* img, int stride, int m, int hi, int ks) {
  return *reinterpret_cast<const uint4*>(img + m * stride + (2 * ks + hi) * 16);
}

// acc[i] += W[tile t0 + i tstep][contraction steps ks0 .. ks0 + NK) . (the 32 rows of `img`)^T; fragment (tile, ks) of the pack sits at
// ((tile nks + ks) 64 + lane) uint4.  fill() starts the stream (the first PD fragments travel under whatever the caller does
// next), run() needs the image complete.
template <int TPW, int NK, int PD> struct DlStream {
  static constexpr int STEPS = TPW * NK;
  static_assert(STEPS % PD == 0 && PD <= STEPS, "ring slots are compile-time constants");
  const uint4* P;
  int nks, tstep;
  uint4 ring[PD];
  __device__ __forceinline__ const uint4* fptr(int s) const { return P + (int64_t)((s % TPW) * tstep * nks + (s / TPW)) * 64; }
  __device__ __forceinline__ void fill(const uint4* pack, int nks_total, int t0, int tstep_, int ks0, int lane) {
    nks = nks_total; tstep = tstep_;
    P = pack + ((int64_t)t0 * nks_total + ks0) * 64 + lane;
#pragma unroll
    for (int s = 0; s < PD; ++s) ring[s] = ld_global_b128(fptr(s));
    __builtin_amdgcn_sched_barrier(0);
  }
  __device__ __forceinline__ void run(f32x16 (&acc)[TPW], const unsigned char* img, int stride, int xk0, int lane) {
    const int m = lane & 31, hi = lane >> 5;
    uint4 xb;
#pragma clang loop unroll(full)
    for (int s = 0; s < STEPS; ++s) {
      if (s % TPW == 0) xb = dl_frag(img, stride, m, hi, xk0 + s / TPW);
      const uint4 w = ring[s % PD];
      mma32(acc[s % TPW], w, xb);
      if (s + PD < STEPS) ring[s % PD] = ld_global_b128(fptr(s + PD));
      __builtin_amdgcn_sched_barrier(0);
    }
  }
};

template <int N> __device__ __forceinline__ void dl_zero(f32x16 (&acc)[N]) {
#pragma unroll
  for (int i = 0; i < N; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
}

// accumulator tile (columns col0 + 8q + 4hi + (r&3) of row m) -> red[m][..] (row stride DL_RS floats)
__device__ __forceinline__ void dl_put_tile(float* red, const f32x16& a, int col0, int lane) {
  const int m = lane & 31, hi = lane >> 5;
#pragma unroll
  for (int q = 0; q < 4; ++q)
    *reinterpret_cast<float4*>(red + m * DL_RS + col0 + 8 * q + 4 * hi) = make_float4(a[4 * q], a[4 * q + 1], a[4 * q + 2], a[4 * q + 3]);
}

// red[32][256] -> slab rows, rounded to 16 bits (whole 512-byte rows per wave instruction).  Every slab of this file is 16-bit: a
// launch's prologue is bound by what its CU can ingest (~22 B/clk: the slabs + residual rows of 32 rows were 160 KiB of a 256 KiB
// prologue, 20 k cycles of a 30 k cycle launch by clock stamps); the shares are rounded once, like every other branch of this model
// that feeds an add + LayerNorm, and added up in fp32 by the consumer.
template <int NW>
__device__ __forceinline__ void dl_store_slab(uint16_t* slab, const float* red, int64_t row0, int nrows, int tid) {
  constexpr int RPW = DL_RB / NW;
  const int lane = tid & 63, wid = tid >> 6, col = lane * 4;
#pragma unroll
  for (int i = 0; i < RPW; ++i) {
    const int r = wid * RPW + i;
    const float4 v = *reinterpret_cast<const float4*>(red + r * DL_RS + col);
    if (r < nrows) *reinterpret_cast<uint2*>(slab + (row0 + r) * DL_D + col) = make_uint2(pack2h(v.x, v.y), pack2h(v.z, v.w));
  }
}

struct DlGeom { int B, L, G; };    // G = 32 / L utterances per group
__device__ __forceinline__ void dl_group(const DlGeom& g, int gi, int& u0, int64_t& row0, int& nrows) {
  u0 = gi * g.G;
  row0 = (int64_t)u0 * g.L;
  nrows = min(g.G, g.B - u0) * g.L;
}
// Workgroup id -> (row block or group, part), part = head or hidden slice, nparts of them.  The nparts workgroups of one row block
// read the SAME slabs / residual rows in their prologue (every one of them finishes the LayerNorm): with ids that are equal modulo
// 8 they sit on one XCD (consecutive ids go to consecutive XCDs: observed placement, used for locality only) and the rows cross the
// fabric once instead of nparts times.  Launch 8 * nparts * ceil(nblocks / 8) workgroups; returns false for the padding ones.
__host__ __device__ __forceinline__ bool dl_block_of(int id, int nparts, int nblocks, int& blk, int& part) {
  const int xcd = id & 7, k = id >> 3;
  part = k % nparts;
  blk = (k / nparts) * 8 + xcd;
  return blk < nblocks;
}
static inline unsigned dl_grid(int nparts, int nblocks) { return (unsigned)(8 * nparts * ((nblocks + 7) / 8)); }

// ------------------------------------------------------------------------------------------------ self-attention launch
struct DlSelfArgs {
  unsigned long long* trace;
  DlLn ln;
  DlGeom g;
  const uint4* wqkv; const float* bqkv;      // forward pack of qvk_proj.weight [768, 256] (q | k | v rows), bias [768]
  const uint4* wo;                           // forward pack of output_proj.weight [256, 256]
  uint16_t* qkv16;                           // [R, 768] out (saved for the backward pass)
  uint16_t* ctx16;                           // [R, 256] out: merged-head attention context (operand of the output projection)
  float* lse;                                // [B, H, L] out
  uint16_t* slabs;                           // [H][R][256] 16-bit out: this head's share of context . W_o^T
  float scale;
};

__global__ __launch_bounds__(256, 1) void dec_self_fwd_kernel(DlSelfArgs p) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[DL_RB * DL_YS + 3 * DL_RB * DL_HS + DL_DK * DL_VT + DL_RB * DL_RS * 4];
  unsigned char* ys = smem;
  unsigned char* qs = ys + DL_RB * DL_YS;
  unsigned char* ks_ = qs + DL_RB * DL_HS;
  unsigned char* cs = ks_ + DL_RB * DL_HS;
  unsigned char* vt = cs + DL_RB * DL_HS;
  float* red = reinterpret_cast<float*>(vt + DL_DK * DL_VT);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 31, hi = lane >> 5;
  int gi, h;
  if (!dl_block_of((int)blockIdx.x, DL_H, (p.g.B + p.g.G - 1) / p.g.G, gi, h)) return;
  int u0, nrows;
  int64_t row0;
  dl_group(p.g, gi, u0, row0, nrows);
  DL_STAMP(0, 0);
  DlPro<4, 8> pro;                                                     // its loads go out first: vmcnt retires in order
  pro.issue(p.ln, row0, nrows, tid);
  DlStream<2, 16, 32> sq;                                             // all 32 fragments of a wave in flight: one round trip
  float4 bq4[2][4];
  if (wid < 3) {
    sq.fill(p.wqkv, 16, wid * 8 + 2 * h, 1, 0, lane);                  // wave 0: q tiles of this head, 1: k, 2: v
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) bq4[t][q] = *reinterpret_cast<const float4*>(p.bqkv + wid * DL_D + DL_DK * h + 32 * t + 8 * q + 4 * (lane >> 5));
  }
#ifdef DL_PROBE      // tuning build only (make DEFS=-DDL_PROBE): where the prologue's cycles go -- loads issued / everything landed / LayerNorm done
  DL_STAMP(0, 8);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  DL_STAMP(0, 9);
#endif
  pro.finish(p.ln, nrows, h == 0, [&](int r, int ch, const uint4& v) { *reinterpret_cast<uint4*>(ys + r * DL_YS + ch * 16) = v; }, tid);
#ifdef DL_PROBE
  DL_STAMP(0, 10);
#endif
  __syncthreads();
  DL_STAMP(0, 1);
  if (wid < 3) {
    f32x16 acc[2];
    dl_zero(acc);
    sq.run(acc, ys, DL_YS, 0, lane);
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c = 32 * t + 8 * q + 4 * hi;                       // column of the head
        const float4 b = bq4[t][q];
        const float v0 = acc[t][4 * q] + b.x, v1 = acc[t][4 * q + 1] + b.y, v2 = acc[t][4 * q + 2] + b.z, v3 = acc[t][4 * q + 3] + b.w;
        const uint2 pk = make_uint2(pack2h(v0, v1), pack2h(v2, v3));
        if (wid == 0) *reinterpret_cast<uint2*>(qs + m * DL_HS + c * 2) = pk;
        else if (wid == 1) *reinterpret_cast<uint2*>(ks_ + m * DL_HS + c * 2) = pk;
        else {
          uint16_t* col = reinterpret_cast<uint16_t*>(vt + c * DL_VT) + m;
          col[0] = (uint16_t)(pk.x & 0xffffu); col[DL_VT / 2] = (uint16_t)(pk.x >> 16);
          col[2 * (DL_VT / 2)] = (uint16_t)(pk.y & 0xffffu); col[3 * (DL_VT / 2)] = (uint16_t)(pk.y >> 16);
        }
        if (m < nrows) *reinterpret_cast<uint2*>(p.qkv16 + (row0 + m) * (3 * DL_D) + wid * DL_D + DL_DK * h + c) = pk;
      }
  }
  DlStream<2, 4, 8> so;
  so.fill(p.wo, 16, 2 * wid, 1, 4 * h, lane);                         // this head's 4 contraction steps of the output projection
  __syncthreads();
  DL_STAMP(0, 2);
  if (wid == 0) {
    // S^T = K Q^T for the 32 rows of the group: rows = keys j, columns = queries i; lane (i, hi) holds j = 8q + 4hi + (r & 3)
    f32x16 st[1];
    dl_zero(st);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) mma32(st[0], dl_frag(ks_, DL_HS, m, hi, ks), dl_frag(qs, DL_HS, m, hi, ks));
    const int i = m, ui = i / p.g.L;
    float s[16], mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int j = 8 * (r >> 2) + 4 * hi + (r & 3);
      const bool ok = j <= i && i < nrows && j >= ui * p.g.L;        // causal inside the utterance (decoder/utils.py:7-11)
      s[r] = ok ? st[0][r] * p.scale : -INFINITY;
      mx = fmaxf(mx, s[r]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float mref = mx == -INFINITY ? 0.f : mx;
    float sum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s[r] = __expf(s[r] - mref); sum += s[r]; }
    sum += __shfl_xor(sum, 32);
    const float inv = sum > 0.f ? 1.f / sum : 0.f;
    if (hi == 0 && i < nrows) p.lse[((int64_t)(u0 + ui) * DL_H + h) * p.g.L + (i - ui * p.g.L)] = mref + __logf(sum);
    uint4 pb[2];
#pragma unroll
    for (int k2 = 0; k2 < 2; ++k2)
      pb[k2] = make_uint4(pack2h(s[8 * k2] * inv, s[8 * k2 + 1] * inv), pack2h(s[8 * k2 + 2] * inv, s[8 * k2 + 3] * inv),
                          pack2h(s[8 * k2 + 4] * inv, s[8 * k2 + 5] * inv), pack2h(s[8 * k2 + 6] * inv, s[8 * k2 + 7] * inv));
    // O^T = V^T P^T: contraction slot (hi, e) of step k2 is key 16 k2 + 4 hi + e (e < 4) / 16 k2 + 8 + 4 hi + e - 4
    f32x16 ot[2];
    dl_zero(ot);
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int k2 = 0; k2 < 2; ++k2) {
        const unsigned char* vr = vt + (32 * ct + m) * DL_VT + (16 * k2 + 4 * hi) * 2;
        const uint2 lo = *reinterpret_cast<const uint2*>(vr), up = *reinterpret_cast<const uint2*>(vr + 16);
        mma32(ot[ct], make_uint4(lo.x, lo.y, up.x, up.y), pb[k2]);
      }
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c = 32 * ct + 8 * q + 4 * hi;
        const uint2 pk = make_uint2(pack2h(ot[ct][4 * q], ot[ct][4 * q + 1]), pack2h(ot[ct][4 * q + 2], ot[ct][4 * q + 3]));
        *reinterpret_cast<uint2*>(cs + i * DL_HS + c * 2) = pk;
        if (i < nrows) *reinterpret_cast<uint2*>(p.ctx16 + (row0 + i) * DL_D + DL_DK * h + c) = pk;
      }
  }
  __syncthreads();
  DL_STAMP(0, 3);
  f32x16 acc[2];
  dl_zero(acc);
  so.run(acc, cs, DL_HS, 0, lane);
  dl_put_tile(red, acc[0], (2 * wid) * 32, lane);
  dl_put_tile(red, acc[1], (2 * wid + 1) * 32, lane);
  __syncthreads();
  DL_STAMP(0, 4);
  dl_store_slab<4>(p.slabs + (int64_t)h * p.ln.R * DL_D, red, row0, nrows, tid);
  DL_STAMP(0, 5);
}

// ------------------------------------------------------------------------------------------------ self-attention launch, cached decoding
// One beam-search step with a KV cache (recognize.py CachedBeamState) sees ONE new position per hypothesis: R = batch x beam rows
// (80 at C5), every row its own attention problem over the cached positions of its ancestors (decode.hip).  The step spent four
// launches on the sub-layer -- closing LayerNorm of the layer below, q|k|v GEMM, cached attention, output projection + LayerNorm: 23 us
// of dependent 5-7 us launches for 0.1 GFLOP.  Here it is one launch cut along (block of 8 rows, head), 512 threads:
//   LN of the layer below in the prologue (wave w = row w) -> q | k | v of the head, one 32-column tile per wave (waves 0-5) ->
//   k, v appended to the caches -> wave w = row w: scores against the ancestors' cached keys, softmax, context, all on (position,
//   16-byte piece) pairs whose loads were issued before the projection -> the head's share of the output projection, one tile per
//   wave -> slab [head][rows][256].
// The NEXT launch (dec_cross_fwd for a decoder layer, dec_ffn_fwd for an LM layer) finishes the sub-layer's add + LayerNorm in its
// prologue, as in training.  The 32-row MFMA tile holds 8 live rows: the MFMA work is noise, the launch is a chain of round trips.
constexpr int DL_SB = 8;                     // rows per workgroup = waves per workgroup
struct DlStepArgs {
  unsigned long long* trace;
  DlLn ln;
  const uint4* wqkv; const float* bqkv;      // forward pack of qvk_proj.weight [768, 256], bias [768]
  const uint4* wo;                           // forward pack of output_proj.weight [256, 256]
  uint16_t* kc; uint16_t* vc;                // [R, maxlen, 256] write-once caches
  const int32_t* anc;                        // [R, maxlen]: anc[r][j] = the row whose cache holds position j of hypothesis r's prefix
  const int32_t* pos;                        // device scalar: the new position p (keys 0..p)
  int maxlen;
  uint16_t* slabs;                           // [H][R][256] 16-bit out
  float scale;
};

// The LayerNorm prologue for ONE row per wave (lane = 4 columns, 8-byte loads): DlPro finishes 32-row tiles, 3/4 of them copies of
// the last live row when a block has 8.  Up to 16 slabs in flight (the FFN launch of a decode step cuts the hidden units 16 ways).
// No dropout (decoding).
struct DlProRow {
  static constexpr int NS = 16;
  float4 x, bb, gm, bt;
  uint2 t[NS];
  uint2 px;
  __device__ __forceinline__ void issue(const DlLn& p, int64_t row, int lane) {
    const int col = lane * 4;
    if (p.nslab == 0) { px = *reinterpret_cast<const uint2*>(p.x16 + row * DL_D + col); return; }
    bb = p.bias ? *reinterpret_cast<const float4*>(p.bias + col) : make_float4(0.f, 0.f, 0.f, 0.f);
    gm = *reinterpret_cast<const float4*>(p.gamma + col);
    bt = *reinterpret_cast<const float4*>(p.beta + col);
    x = *reinterpret_cast<const float4*>(p.xres + row * DL_D + col);
#pragma unroll
    for (int s = 0; s < NS; ++s)
      if (s < p.nslab) t[s] = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(p.slabs) + ((int64_t)s * p.R + row) * DL_D + col);
  }
  // leaves the row's 16-bit image at img (512 bytes); write: also y / y16 of the row
  __device__ __forceinline__ void finish(const DlLn& p, int64_t row, bool write, unsigned char* img, int lane) {
    const int col = lane * 4;
    if (p.nslab == 0) { *reinterpret_cast<uint2*>(img + col * 2) = px; return; }
    float v[4] = {bb.x, bb.y, bb.z, bb.w};
#pragma unroll
    for (int s = 0; s < NS; ++s)
      if (s < p.nslab) { v[0] += h2f_lo(t[s].x); v[1] += h2f_hi(t[s].x); v[2] += h2f_lo(t[s].y); v[3] += h2f_hi(t[s].y); }
    v[0] += x.x; v[1] += x.y; v[2] += x.z; v[3] += x.w;
    const float mean = wave_sum(v[0] + v[1] + v[2] + v[3]) * (1.f / DL_D);
    float qq = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) { const float d = v[e] - mean; qq += d * d; }
    const float rstd = rsqrtf(wave_sum(qq) * (1.f / DL_D) + p.eps);
    const float o0 = (v[0] - mean) * rstd * gm.x + bt.x, o1 = (v[1] - mean) * rstd * gm.y + bt.y;
    const float o2 = (v[2] - mean) * rstd * gm.z + bt.z, o3 = (v[3] - mean) * rstd * gm.w + bt.w;
    const uint2 h = make_uint2(pack2h(o0, o1), pack2h(o2, o3));
    *reinterpret_cast<uint2*>(img + col * 2) = h;
    if (write) {
      if (p.y) *reinterpret_cast<float4*>(p.y + row * DL_D + col) = make_float4(o0, o1, o2, o3);
      if (p.y16) *reinterpret_cast<uint2*>(p.y16 + row * DL_D + col) = h;
    }
  }
};

// (body + two kernels: alone, and as a PAIR -- r06: the decoder's and the language model's layers of one beam-search step are
// independent chains of the SAME launches; a pair launch runs both problems' workgroups side by side (block ids 0 .. n0-1 = problem a,
// the rest = problem b), so the LM chain rides in the decoder's launches instead of a forked stream: a branch anywhere in a hipGraph takes
// the whole graph off the runtime's fast path, profiles/r06_boundary_probe.txt)
__device__ __forceinline__ void dec_self_step_body(const DlStepArgs& p, const int bid) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[DL_RB * DL_YS + 4 * DL_RB * DL_HS + DL_RB * DL_RS * 4];
  unsigned char* ys = smem;
  unsigned char* qs = ys + DL_RB * DL_YS;
  unsigned char* ks_ = qs + DL_RB * DL_HS;
  unsigned char* vs = ks_ + DL_RB * DL_HS;
  unsigned char* cs = vs + DL_RB * DL_HS;
  float* red = reinterpret_cast<float*>(cs + DL_RB * DL_HS);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 31, hi = lane >> 5;
  int rb, h;
  if (!dl_block_of(bid, DL_H, (int)((p.ln.R + DL_SB - 1) / DL_SB), rb, h)) return;
  const int64_t row0 = (int64_t)rb * DL_SB;
  const int nrows = (int)min((int64_t)DL_SB, p.ln.R - row0);
  const bool live = wid < nrows;                                       // wave w = row w of the block, here and in the attention below
  const int64_t r = row0 + min(wid, nrows - 1);
  // Attention works on (position, 16-byte piece) pairs: lane (g, ch) = (lane >> 3, lane & 7) holds piece ch of positions g + 8t,
  // t < 8, of a 64-position chunk -- a wave instruction touches 8 cache lines (a lane per position touched 64, and the launch
  // grew from 7 to 17 us over 60 positions).  The ancestors of chunk 0 are the first loads of the launch.
  const int g = lane >> 3, ch = lane & 7;
  int rowt[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) rowt[t] = p.anc[r * p.maxlen + min(g + 8 * t, p.maxlen - 1)];
  DlProRow pro;
  pro.issue(p.ln, r, lane);
  const int pp = *p.pos;
  const int part = wid >> 1, half = wid & 1;                           // waves 0-5: (q | k | v, 32-column half of the head)
  DlStream<1, 16, 16> sq;
  float4 bq4[4];
  if (wid < 6) {
    sq.fill(p.wqkv, 16, part * 8 + 2 * h + half, 1, 0, lane);
#pragma unroll
    for (int q = 0; q < 4; ++q) bq4[q] = *reinterpret_cast<const float4*>(p.bqkv + part * DL_D + DL_DK * h + 32 * half + 8 * q + 4 * hi);
  }
  pro.finish(p.ln, r, h == 0 && live, ys + wid * DL_YS, lane);
  uint4 kk[8], vv[8];
  auto fetch = [&](int c0) {                                           // cached keys / values of a chunk: every load in flight at once
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int j = c0 + g + 8 * t;
      kk[t] = vv[t] = make_uint4(0u, 0u, 0u, 0u);
      if (live && j < pp) {
        const int64_t o = ((int64_t)rowt[t] * p.maxlen + j) * DL_D + DL_DK * h + 8 * ch;
        kk[t] = ld_global_b128(p.kc + o);
        vv[t] = ld_global_b128(p.vc + o);
      }
    }
  };
  fetch(0);                                                            // travels under the q | k | v projection
  __syncthreads();
  if (wid < 6) {
    f32x16 acc[1];
    dl_zero(acc);
    sq.run(acc, ys, DL_YS, 0, lane);
    unsigned char* img = part == 0 ? qs : part == 1 ? ks_ : vs;
    uint16_t* cache = part == 1 ? p.kc : p.vc;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int c = 32 * half + 8 * q + 4 * hi;                        // column of the head
      const float4 b = bq4[q];
      const uint2 pk = make_uint2(pack2h(acc[0][4 * q] + b.x, acc[0][4 * q + 1] + b.y), pack2h(acc[0][4 * q + 2] + b.z, acc[0][4 * q + 3] + b.w));
      *reinterpret_cast<uint2*>(img + m * DL_HS + c * 2) = pk;
      if (part > 0 && m < nrows) *reinterpret_cast<uint2*>(cache + ((row0 + m) * p.maxlen + pp) * DL_D + DL_DK * h + c) = pk;
    }
  }
  DlStream<1, 4, 4> so;
  so.fill(p.wo, 16, wid, 1, 4 * h, lane);                              // this head's 4 contraction steps, output columns 32 wid ..
  __syncthreads();
  if (live) {
    const uint4 qv = *reinterpret_cast<const uint4*>(qs + wid * DL_HS + 16 * ch);
    const uint4 knew = *reinterpret_cast<const uint4*>(ks_ + wid * DL_HS + 16 * ch);   // the new position: from LDS, not back through the cache
    const uint4 vnew = *reinterpret_cast<const uint4*>(vs + wid * DL_HS + 16 * ch);
    const uint32_t qw[4] = {qv.x, qv.y, qv.z, qv.w};
    float mrun = -INFINITY, l = 0.f, acc8[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc8[e] = 0.f;
    for (int c0 = 0;;) {
      float sc[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        if (c0 + g + 8 * t == pp) { kk[t] = knew; vv[t] = vnew; }
        const uint32_t w[4] = {kk[t].x, kk[t].y, kk[t].z, kk[t].w};
        float a = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) { a += h2f_lo(qw[e]) * h2f_lo(w[e]); a += h2f_hi(qw[e]) * h2f_hi(w[e]); }
        sc[t] = a;
      }
#pragma unroll
      for (int o = 1; o < 8; o <<= 1)
#pragma unroll
        for (int t = 0; t < 8; ++t) sc[t] += __shfl_xor(sc[t], o);     // the 8 pieces of a position
      float mx = -INFINITY;
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        sc[t] = (c0 + g + 8 * t <= pp) ? sc[t] * p.scale : -INFINITY;
        mx = fmaxf(mx, sc[t]);
      }
#pragma unroll
      for (int o = 8; o < 64; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
      const float mn = fmaxf(mrun, mx);
      const float corr = expf(mrun - mn);                              // -inf on the first chunk -> 0
      float ls = 0.f;
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        sc[t] = (c0 + g + 8 * t <= pp) ? expf(sc[t] - mn) : 0.f;
        ls += sc[t];
      }
#pragma unroll
      for (int o = 8; o < 64; o <<= 1) ls += __shfl_xor(ls, o);
      l = l * corr + ls;
      mrun = mn;
#pragma unroll
      for (int e = 0; e < 8; ++e) acc8[e] *= corr;
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const uint32_t w[4] = {vv[t].x, vv[t].y, vv[t].z, vv[t].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) { acc8[2 * e] += sc[t] * h2f_lo(w[e]); acc8[2 * e + 1] += sc[t] * h2f_hi(w[e]); }
      }
      c0 += 64;
      if (c0 > pp) break;
#pragma unroll
      for (int t = 0; t < 8; ++t) rowt[t] = p.anc[r * p.maxlen + min(c0 + g + 8 * t, p.maxlen - 1)];
      fetch(c0);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      acc8[e] += __shfl_xor(acc8[e], 8);
      acc8[e] += __shfl_xor(acc8[e], 16);
      acc8[e] += __shfl_xor(acc8[e], 32);
    }
    if (lane < 8) {
      const float inv = 1.f / l;
      *reinterpret_cast<uint4*>(cs + wid * DL_HS + 16 * lane) =
          make_uint4(pack2h(acc8[0] * inv, acc8[1] * inv), pack2h(acc8[2] * inv, acc8[3] * inv), pack2h(acc8[4] * inv, acc8[5] * inv),
                     pack2h(acc8[6] * inv, acc8[7] * inv));
    }
  }
  __syncthreads();
  f32x16 acc[1];
  dl_zero(acc);
  so.run(acc, cs, DL_HS, 0, lane);
  dl_put_tile(red, acc[0], wid * 32, lane);
  __syncthreads();
  dl_store_slab<8>(p.slabs + (int64_t)h * p.ln.R * DL_D, red, row0, nrows, tid);
}

__global__ __launch_bounds__(512, 1) void dec_self_step_kernel(DlStepArgs p) { dec_self_step_body(p, (int)blockIdx.x); }
struct DlStepPair { DlStepArgs a, b; int n0; };
__global__ __launch_bounds__(512, 1) void dec_self_step_pair_kernel(DlStepPair pp) {
  if ((int)blockIdx.x < pp.n0) dec_self_step_body(pp.a, (int)blockIdx.x);
  else dec_self_step_body(pp.b, (int)blockIdx.x - pp.n0);
}

// ------------------------------------------------------------------------------------------------ cross-attention launch
struct DlCrossArgs {
  unsigned long long* trace;
  DlLn ln;
  DlGeom g;
  const uint4* wq; const float* bq;          // forward pack of q_proj.weight [256, 256], bias
  const uint4* wo;                           // forward pack of output_proj.weight
  const uint16_t* kv;                        // keys / values of the encoder memory: element (b, t, c) at kv[b kv_bs + t kv_ts + c]
  int64_t kv_bs, kv_ts;
  int koff, voff;                            // first column of this layer's keys / values (head h adds 64 h)
  const uint8_t* kmask;                      // [B, Tk] 1 = valid key, or NULL
  int Tk;
  uint16_t* q16; uint16_t* ctx16;            // [R, 256] out
  float* lse;                                // [B, H, L] out
  uint16_t* slabs;                           // [H][R][256] 16-bit out
  float scale;
};

struct DlKvTile { uint4 k[4]; uint4 v[4]; uint32_t valid; };

__device__ __forceinline__ void dl_load_kv(DlKvTile& t, const DlCrossArgs& p, int b, int j0, int h, int lane) {
  const int m = lane & 31, hi = lane >> 5;
  const uint16_t* base = p.kv + (int64_t)b * p.kv_bs;
  const uint16_t* kr = base + (int64_t)min(j0 + m, p.Tk - 1) * p.kv_ts + p.koff + DL_DK * h + 8 * hi;
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) t.k[ks] = ld_global_b128(kr + 16 * ks);          // MFMA A operand: key row m, contraction 16 ks + 8 hi ..
#pragma unroll
  for (int q = 0; q < 4; ++q)                                                      // value rows 8 q + lane / 8, 16-byte piece lane % 8
    t.v[q] = ld_global_b128(base + (int64_t)min(j0 + 8 * q + (lane >> 3), p.Tk - 1) * p.kv_ts + p.voff + DL_DK * h + 8 * (lane & 7));
  uint32_t ok = 0;
  if (lane < 32 && j0 + lane < p.Tk) ok = p.kmask ? p.kmask[(int64_t)b * p.Tk + j0 + lane] : 1u;
  t.valid = (uint32_t)__ballot(ok != 0);
}

__global__ __launch_bounds__(512, 1) void dec_cross_fwd_kernel(DlCrossArgs p) {
  constexpr int OS = DL_DK + 4;               // floats per row of a wave's partial context
  __shared__ __attribute__((aligned(16))) unsigned char smem[DL_RB * DL_YS + 2 * DL_RB * DL_HS + 8 * DL_DK * DL_VT + 8 * DL_RB * OS * 4 +
                                                              2 * 8 * DL_RB * 4];
  static_assert(8 * DL_RB * OS >= DL_RB * DL_RS, "the output staging tile reuses the waves' partial contexts");
  unsigned char* ys = smem;
  unsigned char* qs = ys + DL_RB * DL_YS;
  unsigned char* cs = qs + DL_RB * DL_HS;
  unsigned char* vtw = cs + DL_RB * DL_HS;                            // per wave: transposed value tile [64][32]
  float* ob = reinterpret_cast<float*>(vtw + 8 * DL_DK * DL_VT);       // [8 waves][32][OS]
  float* mb = ob + 8 * DL_RB * OS;                                     // [8][32] running maxima
  float* lb = mb + 8 * DL_RB;                                          // [8][32] running sums
  float* red = ob;                                                     // after the merge
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 31, hi = lane >> 5;
  int gi, h;
  if (!dl_block_of((int)blockIdx.x, DL_H, (p.g.B + p.g.G - 1) / p.g.G, gi, h)) return;
  int u0, nrows;
  int64_t row0;
  dl_group(p.g, gi, u0, row0, nrows);
  const int nutt = nrows / p.g.L, ntile = (p.Tk + 31) >> 5, nit = nutt * ntile;
  DL_STAMP(1, 0);
  DlPro<8, 4> pro;
  pro.issue(p.ln, row0, nrows, tid);
  // the first key / value tile of this wave travels under the prologue and the q projection (it does not depend on them)
  DlKvTile cur;
  {
    const int it = min(wid, nit - 1);
    dl_load_kv(cur, p, u0 + it / ntile, (it % ntile) * 32, h, lane);
  }
  DlStream<1, 16, 16> sq;
  if (wid < 2) sq.fill(p.wq, 16, 2 * h + wid, 1, 0, lane);
  DlStream<1, 4, 4> so;
  so.fill(p.wo, 16, wid, 1, 4 * h, lane);
  pro.finish(p.ln, nrows, h == 0, [&](int r, int ch, const uint4& v) { *reinterpret_cast<uint4*>(ys + r * DL_YS + ch * 16) = v; }, tid);
  DlKvTile alt;
  if (wid + 8 < nit) {                                                // this wave's second tile lands under the q projection; one utterance of
    const int it = wid + 8;                                           // <= 256 frames per workgroup (the shipped shapes) has none: 8 KiB per wave
    dl_load_kv(alt, p, u0 + it / ntile, (it % ntile) * 32, h, lane);  // that the CU's ingest does not spend
  } else {
    alt = DlKvTile{};
  }
  __syncthreads();
  DL_STAMP(1, 1);
  if (wid < 2) {
    f32x16 acc[1];
    dl_zero(acc);
    sq.run(acc, ys, DL_YS, 0, lane);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int c = 32 * wid + 8 * q + 4 * hi;
      const float4 b = *reinterpret_cast<const float4*>(p.bq + DL_DK * h + c);
      const uint2 pk = make_uint2(pack2h(acc[0][4 * q] + b.x, acc[0][4 * q + 1] + b.y), pack2h(acc[0][4 * q + 2] + b.z, acc[0][4 * q + 3] + b.w));
      *reinterpret_cast<uint2*>(qs + m * DL_HS + c * 2) = pk;
      if (m < nrows) *reinterpret_cast<uint2*>(p.q16 + (row0 + m) * DL_D + DL_DK * h + c) = pk;
    }
  }
  __syncthreads();
  DL_STAMP(1, 2);
  // ---- flash attention: this wave takes (utterance, key tile) pairs wid, wid + 8, ...; the queries are the group's 32 rows, of which
  // only the rows of that utterance take part (the others see -inf scores and keep their state).  Two tile register sets take
  // turns (no copies: a copy of registers that a load still has to fill waits for the load, which serialised the loop)
  uint4 qf[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) qf[ks] = dl_frag(qs, DL_HS, m, hi, ks);
  const int i = m, ui = i / p.g.L;
  float mrun = -INFINITY, lrun = 0.f;
  f32x16 o[2];
  dl_zero(o);
  unsigned char* vt = vtw + wid * (DL_DK * DL_VT);
  auto load_tile = [&](DlKvTile& t, int it) {
    const int itc = min(it, nit - 1);                                   // past the end: a valid tile, loaded and never used
    dl_load_kv(t, p, u0 + itc / ntile, (itc % ntile) * 32, h, lane);
  };
  auto consume = [&](const DlKvTile& t, int it) {
    const int u = it / ntile;
    // values -> transposed image of this wave (rows = the head's 64 columns, 32 keys each)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int j = 8 * q + (lane >> 3), c0 = 8 * (lane & 7);
      const uint32_t w[4] = {t.v[q].x, t.v[q].y, t.v[q].z, t.v[q].w};
#pragma unroll
      for (int e = 0; e < 8; ++e)
        *reinterpret_cast<uint16_t*>(vt + (c0 + e) * DL_VT + j * 2) = (uint16_t)((e & 1) ? (w[e >> 1] >> 16) : (w[e >> 1] & 0xffffu));
    }
    f32x16 st[1];
    dl_zero(st);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) mma32(st[0], t.k[ks], qf[ks]);
    float s[16], tmax = -INFINITY;
    const bool mine = ui == u && i < nrows;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int j = 8 * (r >> 2) + 4 * hi + (r & 3);
      const bool ok = mine && ((t.valid >> j) & 1u);
      s[r] = ok ? st[0][r] * p.scale : -INFINITY;
      tmax = fmaxf(tmax, s[r]);
    }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
    const float mnew = fmaxf(mrun, tmax), mref = mnew == -INFINITY ? 0.f : mnew;
    const float alpha = __expf(mrun - mref);
    float psum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s[r] = __expf(s[r] - mref); psum += s[r]; }
    psum += __shfl_xor(psum, 32);
    lrun = lrun * alpha + psum;
    mrun = mnew;
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[ct][r] *= alpha;
    uint4 pb[2];
#pragma unroll
    for (int k2 = 0; k2 < 2; ++k2)
      pb[k2] = make_uint4(pack2h(s[8 * k2], s[8 * k2 + 1]), pack2h(s[8 * k2 + 2], s[8 * k2 + 3]),
                          pack2h(s[8 * k2 + 4], s[8 * k2 + 5]), pack2h(s[8 * k2 + 6], s[8 * k2 + 7]));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                // this wave's own LDS writes of the value tile (in order, same wave)
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int k2 = 0; k2 < 2; ++k2) {
        const unsigned char* vr = vt + (32 * ct + m) * DL_VT + (16 * k2 + 4 * hi) * 2;
        const uint2 lo = *reinterpret_cast<const uint2*>(vr), up = *reinterpret_cast<const uint2*>(vr + 16);
        mma32(o[ct], make_uint4(lo.x, lo.y, up.x, up.y), pb[k2]);
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                // the tile is read before the next round overwrites it
  };
  for (int it = wid; it < nit; it += 16) {
    consume(cur, it);
    if (it + 8 < nit) {
      load_tile(cur, it + 16);
      consume(alt, it + 8);
      load_tile(alt, it + 24);
    }
  }
  DL_STAMP(1, 3);
  // ---- the eight waves' partial results meet in LDS
  if (hi == 0) { mb[wid * DL_RB + i] = mrun; lb[wid * DL_RB + i] = lrun; }
#pragma unroll
  for (int ct = 0; ct < 2; ++ct)
#pragma unroll
    for (int q = 0; q < 4; ++q)
      *reinterpret_cast<float4*>(ob + (wid * DL_RB + i) * OS + 32 * ct + 8 * q + 4 * hi) =
          make_float4(o[ct][4 * q], o[ct][4 * q + 1], o[ct][4 * q + 2], o[ct][4 * q + 3]);
  __syncthreads();
  DL_STAMP(1, 4);
  {
    const int r = tid >> 4, c = (tid & 15) * 4;                        // 512 threads: row r, four columns of the head
    float M = -INFINITY;
#pragma unroll
    for (int w = 0; w < 8; ++w) M = fmaxf(M, mb[w * DL_RB + r]);
    const float mref = M == -INFINITY ? 0.f : M;
    float den = 0.f, acc4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int w = 0; w < 8; ++w) {
      const float f = __expf(mb[w * DL_RB + r] - mref);
      den += f * lb[w * DL_RB + r];
      const float4 t = *reinterpret_cast<const float4*>(ob + (w * DL_RB + r) * OS + c);
      acc4[0] += f * t.x; acc4[1] += f * t.y; acc4[2] += f * t.z; acc4[3] += f * t.w;
    }
    const float inv = den > 0.f ? 1.f / den : 0.f;
    const uint2 pk = make_uint2(pack2h(acc4[0] * inv, acc4[1] * inv), pack2h(acc4[2] * inv, acc4[3] * inv));
    *reinterpret_cast<uint2*>(cs + r * DL_HS + c * 2) = pk;
    if (r < nrows) {
      *reinterpret_cast<uint2*>(p.ctx16 + (row0 + r) * DL_D + DL_DK * h + c) = pk;
      if ((tid & 15) == 0) {
        const int ur = r / p.g.L;
        p.lse[((int64_t)(u0 + ur) * DL_H + h) * p.g.L + (r - ur * p.g.L)] = mref + __logf(den);
      }
    }
  }
  __syncthreads();
  DL_STAMP(1, 5);
  f32x16 acc[1];
  dl_zero(acc);
  so.run(acc, cs, DL_HS, 0, lane);
  dl_put_tile(red, acc[0], wid * 32, lane);
  __syncthreads();
  dl_store_slab<8>(p.slabs + (int64_t)h * p.ln.R * DL_D, red, row0, nrows, tid);
  DL_STAMP(1, 6);
}

// ------------------------------------------------------------------------------------------------ FFN launch
struct DlFfnArgs {
  unsigned long long* trace;
  DlLn ln;
  const uint4* p1; const float* b1; const uint4* p2;   // packs of w_1 [2F, 256] (perm 0) and w_2 [256, F] (perm 1), as otr_ffn_ln_fwd
  uint16_t* slabs;                                      // [S][R][256] 16-bit out: w_2 glu(w_1 y + b_1) over this slice's hidden units (no b_2)
  uint4* hsave;                                         // NULL, or [row blocks][F / 32 chunks][4][64 lanes] x 16 B out: (value + bias, sigmoid(gate)) of
                                                        // every hidden unit in accumulator order (pieces 0, 1 = values, 2, 3 = sigmoids): what the
                                                        // backward launch reads instead of recomputing the hidden (1/3 less weight traffic there)
  int F, S;
};

// red[4 waves][32][256] fp32 partial tiles -> one 16-bit slab (the partial sums of a slice are rounded once, like every other
// branch of this model that feeds an add + LayerNorm; the consumer adds the S slabs in fp32)
__device__ __forceinline__ void dl_store_slab16(uint16_t* slab, const float* red, int64_t row0, int nrows, int wid, int lane) {
  const int col = lane * 4;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int r = wid * 8 + i;
    float4 v = *reinterpret_cast<const float4*>(red + r * DL_RS + col);
#pragma unroll
    for (int w = 1; w < 4; ++w) {
      const float4 t = *reinterpret_cast<const float4*>(red + (w * DL_RB + r) * DL_RS + col);
      v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
    }
    if (r < nrows) *reinterpret_cast<uint2*>(slab + (row0 + r) * DL_D + col) = make_uint2(pack2h(v.x, v.y), pack2h(v.z, v.w));
  }
}

__device__ __forceinline__ void dec_ffn_fwd_body(const DlFfnArgs& p, const int bid) {
  constexpr int D = DL_D, NKS = D / 16, NT = D / 32, STEPS = 2 * NKS + 2 * NT, PD = 24;
  static_assert(STEPS % PD == 0, "ring slots must be compile-time constants");
  __shared__ __attribute__((aligned(16))) unsigned char smem[DL_RB * DL_YS + 4 * DL_RB * DL_RS * 4];
  unsigned char* ys = smem;
  float* red = reinterpret_cast<float*>(smem + DL_RB * DL_YS);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 31, hi = lane >> 5;
  int rb, sl;
  if (!dl_block_of(bid, p.S, (int)((p.ln.R + DL_RB - 1) / DL_RB), rb, sl)) return;
  const int64_t row0 = (int64_t)rb * DL_RB;
  const int nrows = (int)min((int64_t)DL_RB, p.ln.R - row0);
  const int nchunk = p.F / 32, cps = nchunk / p.S, nit = cps / 4;     // chunks (32 hidden units) of the layer / of a slice / of a wave
  auto chunk_of = [&](int it) { return sl * cps + 4 * it + wid; };
  const uint4* P1 = p.p1 + lane;
  const uint4* P2 = p.p2 + lane;
  auto fptr = [&](int c, int s) -> const uint4* {
    if (s < 2 * NKS) return P1 + (int64_t)(((s & 1) ? nchunk + c : c) * NKS + (s >> 1)) * 64;
    const int t = s - 2 * NKS;
    return P2 + (int64_t)((t >> 1) * (2 * nchunk) + 2 * c + (t & 1)) * 64;
  };
  DL_STAMP(2, 0);
  DlPro<4, 4> pro;
  pro.issue(p.ln, row0, nrows, tid);
  uint4 ring[PD];
  int c = chunk_of(0);
#pragma unroll
  for (int s = 0; s < PD; ++s) ring[s] = ld_global_b128(fptr(c, s));
  __builtin_amdgcn_sched_barrier(0);
  pro.finish(p.ln, nrows, sl == 0, [&](int r, int ch, const uint4& v) { *reinterpret_cast<uint4*>(ys + r * DL_YS + ch * 16) = v; }, tid);
  __syncthreads();
  DL_STAMP(2, 1);
  f32x16 yacc[NT];
  dl_zero(yacc);
  for (int it = 0; it < nit; ++it) {
    const int cn = chunk_of(min(it + 1, nit - 1));
    float4 bv[4], bg[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      bv[q] = *reinterpret_cast<const float4*>(p.b1 + c * 32 + 8 * q + 4 * hi);
      bg[q] = *reinterpret_cast<const float4*>(p.b1 + p.F + c * 32 + 8 * q + 4 * hi);
    }
    f32x16 av, ag;
#pragma unroll
    for (int r = 0; r < 16; ++r) { av[r] = 0.f; ag[r] = 0.f; }
    uint4 xb, uf0, uf1;
#pragma clang loop unroll(full)
    for (int s = 0; s < STEPS; ++s) {
      const uint4 w = ring[s % PD];
      if (s < 2 * NKS) {
        if ((s & 1) == 0) xb = dl_frag(ys, DL_YS, m, hi, s >> 1);
        if (s & 1) mma32(ag, w, xb); else mma32(av, w, xb);
      } else {
        const int t = s - 2 * NKS;
        mma32(yacc[t >> 1], w, (t & 1) ? uf1 : uf0);
      }
      ring[s % PD] = ld_global_b128(s + PD < STEPS ? fptr(c, s + PD) : fptr(cn, s + PD - STEPS));
      if (s == 2 * NKS - 1) {
        float u[16], a[16], sg[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          a[r] = av[r] + reinterpret_cast<const float*>(&bv[r >> 2])[r & 3];
          sg[r] = fast_sigmoid(ag[r] + reinterpret_cast<const float*>(&bg[r >> 2])[r & 3]);
          u[r] = a[r] * sg[r];
        }
        tile_to_frags(u, uf0, uf1);
        if (p.hsave) {
          uint4 a0, a1, s0, s1;
          tile_to_frags(a, a0, a1);
          tile_to_frags(sg, s0, s1);
          uint4* hs = p.hsave + (((int64_t)rb * nchunk + c) * 4) * 64 + lane;
          st_global_b128(hs, a0); st_global_b128(hs + 64, a1); st_global_b128(hs + 128, s0); st_global_b128(hs + 192, s1);
        }
      }
    }
    c = cn;
  }
  DL_STAMP(2, 2);
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) dl_put_tile(red + wid * DL_RB * DL_RS, yacc[nt], nt * 32, lane);
  __syncthreads();
  DL_STAMP(2, 3);
  dl_store_slab16(p.slabs + (int64_t)sl * p.ln.R * D, red, row0, nrows, wid, lane);
}

__global__ __launch_bounds__(256, 1) void dec_ffn_fwd_kernel(DlFfnArgs p) { dec_ffn_fwd_body(p, (int)blockIdx.x); }
struct DlFfnPair { DlFfnArgs a, b; int n0; };
__global__ __launch_bounds__(256, 1) void dec_ffn_fwd_pair_kernel(DlFfnPair pp) {
  if ((int)blockIdx.x < pp.n0) dec_ffn_fwd_body(pp.a, (int)blockIdx.x);
  else dec_ffn_fwd_body(pp.b, (int)blockIdx.x - pp.n0);
}

// ------------------------------------------------------------------------------------------------ closing LayerNorm
struct DlLnArgs { DlLn ln; };
__device__ __forceinline__ void dec_ln_body(const DlLnArgs& p, const int bid) {
  // 8 rows per workgroup (2 per wave): 60 workgroups at 480 rows instead of 15, each reading S x 8 KiB of slabs
  constexpr int RPB = 8;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, col = lane * 4;
  const int64_t row0 = (int64_t)bid * RPB;
  const int nrows = (int)min((int64_t)RPB, p.ln.R - row0);
  const DlLn& q = p.ln;
  const bool drop = q.p_drop > 0.f;
  const uint64_t seed = drop ? *q.seed : 0;
  const uint32_t thr = drop ? (uint32_t)fminf(q.p_drop * 4294967296.f, 4294967295.f) : 0;
  const float inv_keep = drop ? 1.f / (1.f - q.p_drop) : 1.f;
  float4 bb = make_float4(0.f, 0.f, 0.f, 0.f);
  if (q.bias) bb = *reinterpret_cast<const float4*>(q.bias + col);
  const float4 gm = *reinterpret_cast<const float4*>(q.gamma + col);
  const float4 bt = *reinterpret_cast<const float4*>(q.beta + col);
  const float g4[4] = {gm.x, gm.y, gm.z, gm.w}, b4[4] = {bt.x, bt.y, bt.z, bt.w};
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = wid * 2 + i;
    if (r >= nrows) break;                                            // wave-uniform
    const int64_t row = row0 + r;
    const float4 xr = *reinterpret_cast<const float4*>(q.xres + row * DL_D + col);
    float a[4] = {bb.x, bb.y, bb.z, bb.w};
    for (int s0 = 0; s0 < q.nslab; s0 += 4) {                         // four slabs in flight
      uint2 t[4];                                                      // the FFN launches' slabs are 16-bit
#pragma unroll
      for (int s = 0; s < 4; ++s)
        t[s] = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(q.slabs) + ((int64_t)min(s0 + s, q.nslab - 1) * q.R + row) * DL_D + col);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const float live = s0 + s < q.nslab ? 1.f : 0.f;
        a[0] += live * h2f_lo(t[s].x); a[1] += live * h2f_hi(t[s].x); a[2] += live * h2f_lo(t[s].y); a[3] += live * h2f_hi(t[s].y);
      }
    }
    const float xv[4] = {xr.x, xr.y, xr.z, xr.w};
    float v[4], sm = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float sc = 1.f;
      if (drop) sc = otr_rand32(seed, q.rng_offset + (uint64_t)(row * DL_D + col + e)) >= thr ? inv_keep : 0.f;
      v[e] = xv[e] + a[e] * sc;
      sm += v[e];
    }
    sm = wave_sum(sm) * (1.f / DL_D);
    float qq = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) { const float d = v[e] - sm; qq += d * d; }
    qq = wave_sum(qq);
    const float rstd = rsqrtf(qq * (1.f / DL_D) + q.eps);
    float o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = (v[e] - sm) * rstd * g4[e] + b4[e];
    if (q.z) *reinterpret_cast<float4*>(q.z + row * DL_D + col) = make_float4(v[0], v[1], v[2], v[3]);
    if (q.y) *reinterpret_cast<float4*>(q.y + row * DL_D + col) = make_float4(o[0], o[1], o[2], o[3]);
    if (q.y16) *reinterpret_cast<uint2*>(q.y16 + row * DL_D + col) = make_uint2(pack2h(o[0], o[1]), pack2h(o[2], o[3]));
    if (lane == 0) {
      if (q.mean) q.mean[row] = sm;
      if (q.rstd) q.rstd[row] = rstd;
    }
  }
}
__global__ __launch_bounds__(256) void dec_ln_kernel(DlLnArgs p) { dec_ln_body(p, (int)blockIdx.x); }
struct DlLnPair { DlLnArgs a, b; int n0; };
__global__ __launch_bounds__(256) void dec_ln_pair_kernel(DlLnPair pp) {
  if ((int)blockIdx.x < pp.n0) dec_ln_body(pp.a, (int)blockIdx.x);
  else dec_ln_body(pp.b, (int)blockIdx.x - pp.n0);
}

// ================================================================================================ backward
// The same cut, mirrored.  The gradient of a sub-layer's LayerNorm OUTPUT arrives as a skip part (d z of the sub-layer above, f32
// [R,256]) plus partial slabs (the input-gradient shares of the heads / hidden slices above); every backward launch finishes it
// and runs that LayerNorm's backward in its prologue:
//   dec_ffn_bwd    grid (row blocks, S):  LN3 backward -> da (the FFN output's gradient) -> the 32-row FFN backward of ffn_fused.hip
//                                         (hidden recomputed) on 1/S of the hidden units: dh, u for the weight gradients, the slice's
//                                         share of dx -> slab
//   dec_cross_bwd  grid (groups, heads):  LN2 backward -> d context of one head (da . W_o restricted to its 64 columns) -> attention
//                                         backward of that head (dq; dk, dv written into the shared key / value gradient) -> the
//                                         head's share of dq . W_q -> slab
//   dec_self_bwd   grid (groups, heads):  LN1 backward -> d context -> causal self-attention backward (dq, dk, dv of the head) -> the
//                                         head's share of dqkv . W_qkv -> slab
//   dec_sum        dx = skip + sum of slabs (the gradient that leaves the stack towards the embedding)
// The workgroup with head / slice 0 writes what is per row: d z (the skip part for the launch below), the 16-bit branch gradient (the
// weight-gradient operand of the output projection / w_2) and its sums of dgamma | dbeta | d bias.
struct DlLnB {
  const float* dskip;      // [R,256] or NULL: skip-path part of the gradient of the LayerNorm output
  const void* slabs;       // [nslab][R][256]: the rest of it, in shares (nslab may be 0); fp32 or 16-bit, fixed per launch type
  int nslab;
  const float* z; const float* mean; const float* rstd; const float* gamma; const uint64_t* seed;
  float p_drop;
  uint64_t rng_offset;
  float* dz;               // [R,256] out: gradient of the pre-norm sum = of the residual input (and, masked, of the branch)
  uint16_t* da16;          // [R,256] out: dropout-masked branch gradient, 16-bit
  float* partial;          // [row blocks][3][256] out: sums over the block's rows of dgamma | dbeta | d branch
  int64_t R;
};

// LayerNorm backward of rows row0 .. (clamped to nrows), in two steps like DlPro: issue() starts every global load, finish() leaves
// the 16-bit branch gradient in `img` ([32][DL_YS]) as MFMA B operands.  stage: [3][NW][256] floats of LDS scratch; finish() contains
// two __syncthreads when `write` (block-uniform).  NS = slabs in flight.
template <int NW, int NS> struct DlProB {
  static constexpr int RPW = DL_RB / NW, NP = RPW / 2, D = DL_D;
  int64_t rows[NP];
  float4 z0[NP], z1[NP], k0[NP], k1[NP];
  float mean[NP], rstd[NP];
  otr_u32x4 t[NS][NP];
  float4 gm0, gm1;
  uint64_t seed;
  __device__ __forceinline__ void issue(const DlLnB& p, int64_t row0, int nrows, int tid) {
    const int lane = tid & 63, wid = tid >> 6, hw = lane >> 5, col = (lane & 31) * 8;
    gm0 = *reinterpret_cast<const float4*>(p.gamma + col); gm1 = *reinterpret_cast<const float4*>(p.gamma + col + 4);
    seed = p.p_drop > 0.f ? *p.seed : 0;
#pragma unroll
    for (int k = 0; k < NP; ++k) {
      rows[k] = row0 + min(wid * RPW + 2 * k + hw, nrows - 1);
      z0[k] = *reinterpret_cast<const float4*>(p.z + rows[k] * D + col);
      z1[k] = *reinterpret_cast<const float4*>(p.z + rows[k] * D + col + 4);
      mean[k] = p.mean[rows[k]]; rstd[k] = p.rstd[rows[k]];
      k0[k] = k1[k] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (p.dskip) { k0[k] = *reinterpret_cast<const float4*>(p.dskip + rows[k] * D + col); k1[k] = *reinterpret_cast<const float4*>(p.dskip + rows[k] * D + col + 4); }
    }
#pragma unroll
    for (int s = 0; s < NS; ++s)
      if (s < p.nslab) {
#pragma unroll
        for (int k = 0; k < NP; ++k) {
          const uint4 q = ld_global_b128(reinterpret_cast<const uint16_t*>(p.slabs) + ((int64_t)s * p.R + rows[k]) * D + col);
          t[s][k] = otr_u32x4{q.x, q.y, q.z, q.w};
        }
      }
  }
  __device__ __forceinline__ void finish(const DlLnB& p, int nrows, int block, bool write, unsigned char* img, float* stage, int tid) {
    const int lane = tid & 63, wid = tid >> 6, hw = lane >> 5, col = (lane & 31) * 8;
    const bool drop = p.p_drop > 0.f;
    const uint32_t thr = drop ? (uint32_t)fminf(p.p_drop * 4294967296.f, 4294967295.f) : 0;
    const float inv_keep = drop ? 1.f / (1.f - p.p_drop) : 1.f;
    const float gam[8] = {gm0.x, gm0.y, gm0.z, gm0.w, gm1.x, gm1.y, gm1.z, gm1.w};
    float d8[NP][8];
#pragma unroll
    for (int k = 0; k < NP; ++k) {
      d8[k][0] = k0[k].x; d8[k][1] = k0[k].y; d8[k][2] = k0[k].z; d8[k][3] = k0[k].w; d8[k][4] = k1[k].x; d8[k][5] = k1[k].y; d8[k][6] = k1[k].z; d8[k][7] = k1[k].w;
#pragma unroll
      for (int s = 0; s < NS; ++s)
        if (s < p.nslab) {
          const uint32_t w[4] = {t[s][k].x, t[s][k].y, t[s][k].z, t[s][k].w};
#pragma unroll
          for (int e = 0; e < 4; ++e) { d8[k][2 * e] += h2f_lo(w[e]); d8[k][2 * e + 1] += h2f_hi(w[e]); }
        }
      for (int s = NS; s < p.nslab; ++s) {
        const uint4 q = ld_global_b128(reinterpret_cast<const uint16_t*>(p.slabs) + ((int64_t)s * p.R + rows[k]) * D + col);
        const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) { d8[k][2 * e] += h2f_lo(w[e]); d8[k][2 * e + 1] += h2f_hi(w[e]); }
      }
    }
    float dg[8], db[8], dab[8], z8[NP][8], s1[NP], s2[NP];
#pragma unroll
    for (int e = 0; e < 8; ++e) { dg[e] = 0.f; db[e] = 0.f; dab[e] = 0.f; }
#pragma unroll
    for (int k = 0; k < NP; ++k) {
      const float live = wid * RPW + 2 * k + hw < nrows ? 1.f : 0.f;
      const float zz[8] = {z0[k].x, z0[k].y, z0[k].z, z0[k].w, z1[k].x, z1[k].y, z1[k].z, z1[k].w};
      s1[k] = 0.f; s2[k] = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        d8[k][e] *= live;
        z8[k][e] = (zz[e] - mean[k]) * rstd[k];
        const float g = d8[k][e] * gam[e];
        s1[k] += g; s2[k] += g * z8[k][e];
        dg[e] += d8[k][e] * z8[k][e];
        db[e] += d8[k][e];
      }
    }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1)
#pragma unroll
      for (int k = 0; k < NP; ++k) { s1[k] += __shfl_xor(s1[k], o); s2[k] += __shfl_xor(s2[k], o); }
#pragma unroll
    for (int k = 0; k < NP; ++k) {
      const int r = wid * RPW + 2 * k + hw;
      const float m1 = s1[k] * (1.f / D), m2 = s2[k] * (1.f / D);
      float dz[8], da[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        dz[e] = rstd[k] * (d8[k][e] * gam[e] - m1 - z8[k][e] * m2);
        const float sc = drop ? (otr_rand32(seed, p.rng_offset + (uint64_t)(rows[k] * D + col + e)) >= thr ? inv_keep : 0.f) : 1.f;
        da[e] = dz[e] * sc;
        dab[e] += da[e];
      }
      const uint4 h = make_uint4(pack2h(da[0], da[1]), pack2h(da[2], da[3]), pack2h(da[4], da[5]), pack2h(da[6], da[7]));
      *reinterpret_cast<uint4*>(img + r * DL_YS + col * 2) = h;
      if (write && r < nrows) {
        if (p.dz) {
          *reinterpret_cast<float4*>(p.dz + rows[k] * D + col) = make_float4(dz[0], dz[1], dz[2], dz[3]);
          *reinterpret_cast<float4*>(p.dz + rows[k] * D + col + 4) = make_float4(dz[4], dz[5], dz[6], dz[7]);
        }
        if (p.da16) *reinterpret_cast<uint4*>(p.da16 + rows[k] * D + col) = h;
      }
    }
    if (write && p.partial) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {                     // the two half-waves hold different rows of the same columns
        dg[e] += __shfl_xor(dg[e], 32); db[e] += __shfl_xor(db[e], 32); dab[e] += __shfl_xor(dab[e], 32);
      }
      if (hw == 0) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          stage[(0 * NW + wid) * D + col + e] = dg[e];
          stage[(1 * NW + wid) * D + col + e] = db[e];
          stage[(2 * NW + wid) * D + col + e] = dab[e];
        }
      }
      __syncthreads();
      float* prow = p.partial + (int64_t)block * 3 * D;
      for (int c = tid; c < 3 * D; c += 64 * NW) {
        const int k = c >> 8, cc = c & 255;
        float t_ = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) t_ += stage[(k * NW + w) * D + cc];
        prow[c] = t_;
      }
      __syncthreads();
    }
  }
};

// ------------------------------------------------------------------------------------------------ FFN backward launch
struct DlFfnBwdArgs {
  unsigned long long* trace;
  DlLnB ln;
  const uint4* hsave;      // (value + bias, sigmoid) tiles the forward launch left (DlFfnArgs::hsave)
  const uint4* p3; const uint4* p4;   // packs of w_2^T (rows = F hidden units, perm 0) and w_1^T (rows = 256, contraction 2F, perm 1), as otr_ffn_bwd
  uint16_t* dh;            // [R, 2F] out
  uint16_t* u;             // [R, F] out
  float* bpart;            // [row blocks][2F] out: column sums of dh over the block's rows
  uint16_t* slabs;         // [S][R][256] 16-bit out: dh[slice] . w_1[slice]
  int F, S;
};

__global__ __launch_bounds__(256, 1) void dec_ffn_bwd_kernel(DlFfnBwdArgs p) {
  constexpr int D = DL_D, NKS = D / 16, NT = D / 32;
  constexpr int S2 = NKS, STEPS = S2 + 4 * NT, PD = 24;            // 16 + 32 weight fragments per chunk
  static_assert(STEPS % PD == 0, "ring slots must be compile-time constants");
  __shared__ __attribute__((aligned(16))) unsigned char smem[4 * DL_RB * DL_RS * 4];   // operand image first, partial sums after
  unsigned char* ds = smem;
  float* stage = reinterpret_cast<float*>(smem + DL_RB * DL_YS);
  float* red = reinterpret_cast<float*>(smem);
  static_assert(DL_RB * DL_YS + 3 * 4 * DL_D * 4 <= 4 * DL_RB * DL_RS * 4, "LDS layout");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 31, hi = lane >> 5;
  int rb, sl;
  if (!dl_block_of((int)blockIdx.x, p.S, (int)((p.ln.R + DL_RB - 1) / DL_RB), rb, sl)) return;
  const int64_t row0 = (int64_t)rb * DL_RB;
  const int nrows = (int)min((int64_t)DL_RB, p.ln.R - row0);
  const int nchunk = p.F / 32, cps = nchunk / p.S, nit = cps / 4;
  auto chunk_of = [&](int it) { return sl * cps + 4 * it + wid; };
  const uint4* P3 = p.p3 + lane;
  const uint4* P4 = p.p4 + lane;
  auto fptr = [&](int c, int s) -> const uint4* {
    if (s < S2) return P3 + (int64_t)(c * NKS + s) * 64;
    const int t = s - S2, j4 = t & 3;
    const int ksf = (j4 < 2) ? 2 * c + j4 : 2 * nchunk + 2 * c + (j4 - 2);
    return P4 + (int64_t)((t >> 2) * (4 * nchunk) + ksf) * 64;
  };
  DL_STAMP(3, 0);
  DlProB<4, 4> pro;
  pro.issue(p.ln, row0, nrows, tid);
  uint4 ring[PD];
  int c = chunk_of(0);
#pragma unroll
  for (int s = 0; s < PD; ++s) ring[s] = ld_global_b128(fptr(c, s));
  __builtin_amdgcn_sched_barrier(0);
  pro.finish(p.ln, nrows, rb, sl == 0, ds, stage, tid);
  __syncthreads();
  DL_STAMP(3, 1);
  f32x16 xacc[NT];
  dl_zero(xacc);
  const bool live = m < nrows;
  const int64_t crow = row0 + min(m, nrows - 1);
  for (int it = 0; it < nit; ++it) {
    const int cn = chunk_of(min(it + 1, nit - 1));
    const uint4* hs = p.hsave + (((int64_t)rb * nchunk + c) * 4) * 64 + lane;
    const uint4 a0 = ld_global_b128(hs), a1 = ld_global_b128(hs + 64), g0 = ld_global_b128(hs + 128), g1 = ld_global_b128(hs + 192);
    f32x16 du;
#pragma unroll
    for (int r = 0; r < 16; ++r) du[r] = 0.f;
    uint4 ob, hf[4];
#pragma clang loop unroll(full)
    for (int s = 0; s < STEPS; ++s) {
      const uint4 w = ring[s % PD];
      if (s < S2) {
        ob = dl_frag(ds, DL_YS, m, hi, s);
        mma32(du, w, ob);
      } else {
        const int t = s - S2;
        mma32(xacc[t >> 2], w, hf[t & 3]);
      }
      ring[s % PD] = ld_global_b128(s + PD < STEPS ? fptr(c, s + PD) : fptr(cn, s + PD - STEPS));
      if (s == S2 - 1) {
        // GLU backward on the accumulators against the SAVED (value, sigmoid): u = a sg; d a = du sg; d gate = du a sg (1 - sg)
        const uint32_t aw[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w}, gw[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
        float uu[16], da_[16], dg_[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float a = (r & 1) ? h2f_hi(aw[r >> 1]) : h2f_lo(aw[r >> 1]);
          const float sg = (r & 1) ? h2f_hi(gw[r >> 1]) : h2f_lo(gw[r >> 1]);
          uu[r] = a * sg;
          da_[r] = du[r] * sg;
          dg_[r] = du[r] * uu[r] * (1.f - sg);
        }
        uint4 u0, u1;
        tile_to_frags(uu, u0, u1);
        tile_to_frags(da_, hf[0], hf[1]);
        tile_to_frags(dg_, hf[2], hf[3]);
        store_tile_row(p.u + crow * p.F + c * 32, u0, u1, hi, live);
        store_tile_row(p.dh + crow * (2 * (int64_t)p.F) + c * 32, hf[0], hf[1], hi, live);
        store_tile_row(p.dh + crow * (2 * (int64_t)p.F) + p.F + c * 32, hf[2], hf[3], hi, live);
        float* bp = p.bpart + (int64_t)rb * (2 * p.F) + c * 32;
        tile_colsum_store(da_, bp, lane, hi, live);
        tile_colsum_store(dg_, bp + p.F, lane, hi, live);
      }
    }
    c = cn;
  }
  DL_STAMP(3, 2);
  __syncthreads();                                          // every wave is done with the operand image
  DL_STAMP(3, 3);
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) dl_put_tile(red + wid * DL_RB * DL_RS, xacc[nt], nt * 32, lane);
  __syncthreads();
  dl_store_slab16(p.slabs + (int64_t)sl * p.ln.R * D, red, row0, nrows, wid, lane);
}

// ------------------------------------------------------------------------------------------------ attention backward, shared pieces
// d context of one head from the 16-bit branch gradient image: waves 0 and 1 take one 32-column tile each of da . W_o (the input
// gradient pack: rows = the projection's inputs).  Leaves dO row-major (`dos`), transposed (`dot`) and, with the forward's context
// O read from memory, the per-row partial of delta = rowsum(dO . O) in dpart[tile][row].
__device__ __forceinline__ void dl_dctx(DlStream<1, 16, 16>& sdo, const unsigned char* img, unsigned char* dos, unsigned char* dot,
                                        float* dpart, const uint16_t* ctx16, int64_t row0, int nrows, int h, int wid, int lane) {
  const int m = lane & 31, hi = lane >> 5;
  f32x16 acc[1];
  dl_zero(acc);
  sdo.run(acc, img, DL_YS, 0, lane);
  float dsum = 0.f;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int c = 32 * wid + 8 * q + 4 * hi;
    const uint2 o2 = *reinterpret_cast<const uint2*>(ctx16 + (row0 + min(m, nrows - 1)) * DL_D + DL_DK * h + c);
    const float v0 = acc[0][4 * q], v1 = acc[0][4 * q + 1], v2 = acc[0][4 * q + 2], v3 = acc[0][4 * q + 3];
    const uint2 pk = make_uint2(pack2h(v0, v1), pack2h(v2, v3));
    // delta from the ROUNDED gradient (what the MFMAs below multiply), so that sum_j dS_ij = 0 holds in the arithmetic actually done
    dsum += h2f_lo(pk.x) * h2f_lo(o2.x) + h2f_hi(pk.x) * h2f_hi(o2.x) + h2f_lo(pk.y) * h2f_lo(o2.y) + h2f_hi(pk.y) * h2f_hi(o2.y);
    *reinterpret_cast<uint2*>(dos + m * DL_HS + c * 2) = pk;
    uint16_t* colp = reinterpret_cast<uint16_t*>(dot + c * DL_VT) + m;
    colp[0] = (uint16_t)(pk.x & 0xffffu); colp[DL_VT / 2] = (uint16_t)(pk.x >> 16);
    colp[2 * (DL_VT / 2)] = (uint16_t)(pk.y & 0xffffu); colp[3 * (DL_VT / 2)] = (uint16_t)(pk.y >> 16);
  }
  dsum += __shfl_xor(dsum, 32);
  if (hi == 0) dpart[wid * DL_RB + m] = dsum;
}

// two 8-byte pieces of row `rowp` of a transposed image: contraction slots of step k2 in accumulator order
__device__ __forceinline__ uint4 dl_tfrag(const unsigned char* timg, int row, int hi, int k2) {
  const unsigned char* vr = timg + row * DL_VT + (16 * k2 + 4 * hi) * 2;
  const uint2 lo = *reinterpret_cast<const uint2*>(vr), up = *reinterpret_cast<const uint2*>(vr + 16);
  return make_uint4(lo.x, lo.y, up.x, up.y);
}
__device__ __forceinline__ uint4 dl_pack8(const float* v) {
  return make_uint4(pack2h(v[0], v[1]), pack2h(v[2], v[3]), pack2h(v[4], v[5]), pack2h(v[6], v[7]));
}

// ------------------------------------------------------------------------------------------------ cross-attention backward launch
struct DlCrossBwdArgs {
  unsigned long long* trace;
  DlLnB ln;
  DlGeom g;
  const uint4* wo_t;       // input-gradient pack of output_proj.weight (rows = its 256 inputs)
  const uint4* wq_t;       // input-gradient pack of q_proj.weight
  const uint16_t* q16; const uint16_t* ctx16; const float* lse;
  const uint16_t* kv; uint16_t* dkv;
  int64_t kv_bs, kv_ts;
  int koff, voff;
  const uint8_t* kmask;
  int Tk;
  uint16_t* dq16;          // [R,256] out: gradient of the projected queries (weight-gradient operand of q_proj)
  uint16_t* slabs;         // [H][R][256] 16-bit out: dq_h . W_q[head rows]
  float scale;
};

struct DlKvRows { uint4 k[4]; uint4 v[4]; uint32_t valid; };
__device__ __forceinline__ void dl_load_kv_rows(DlKvRows& t, const uint16_t* kv, int64_t kv_bs, int64_t kv_ts, int koff, int voff,
                                                const uint8_t* kmask, int Tk, int b, int j0, int h, int lane) {
  const int m = lane & 31, hi = lane >> 5;
  const uint16_t* rowp = kv + (int64_t)b * kv_bs + (int64_t)min(j0 + m, Tk - 1) * kv_ts + DL_DK * h + 8 * hi;
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) { t.k[ks] = ld_global_b128(rowp + koff + 16 * ks); t.v[ks] = ld_global_b128(rowp + voff + 16 * ks); }
  uint32_t ok = 0;
  if (lane < 32 && j0 + lane < Tk) ok = kmask ? kmask[(int64_t)b * Tk + j0 + lane] : 1u;
  t.valid = (uint32_t)__ballot(ok != 0);
}

// 8 waves: wave w takes the (utterance, key tile) pairs slot, slot + 4, ... with slot = w & 3, and ONE of the two orientations of the
// score tile (w >> 2): 0 = lane per query (-> dq, summed over the tiles), 1 = lane per key (-> dk, dv of the tile, complete).  A wave
// is a chain of dependent MFMA / LDS / exp steps with nothing to overlap them (10 k cycles per tile for both orientations in one
// wave, clock stamps): two waves per tile halve it.
__global__ __launch_bounds__(512, 1) void dec_cross_bwd_kernel(DlCrossBwdArgs p) {
  constexpr int OS = DL_DK;
  constexpr int SM_IMG = DL_RB * DL_YS, SM_H = DL_RB * DL_HS, SM_T = DL_DK * DL_VT;
  __shared__ __attribute__((aligned(16))) unsigned char smem[SM_IMG + 3 * SM_H + 2 * SM_T + 4 * SM_T + 8 * SM_H + 4 * DL_RB * 4 + DL_RB * DL_RS * 4];
  unsigned char* ys = smem;                       // branch gradient image (LayerNorm backward output)
  unsigned char* dos = ys + SM_IMG;               // dO [32][64]
  unsigned char* qs = dos + SM_H;                 // Q  [32][64]
  unsigned char* dqs = qs + SM_H;                 // dq [32][64]
  unsigned char* dot = dqs + SM_H;                // dO^T [64][32]
  unsigned char* qt = dot + SM_T;                 // Q^T
  unsigned char* ktw = qt + SM_T;                 // per orientation-0 wave: K^T of the current key tile
  unsigned char* ogw = ktw + 4 * SM_T;            // per orientation-1 wave: dk | dv tiles staged for whole-row stores
  float* lses = reinterpret_cast<float*>(ogw + 8 * SM_H);   // [32]
  float* dels = lses + DL_RB;                               // [32]
  float* dpart = dels + DL_RB;                              // [2][32]
  float* red = dpart + 2 * DL_RB;                           // [32][DL_RS]: LayerNorm staging, then the waves' dq partials, then the output tile
  static_assert(3 * 8 * DL_D <= DL_RB * DL_RS && 4 * DL_RB * OS <= DL_RB * DL_RS, "LDS layout");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int slot = wid & 3, orient = wid >> 2;
  const int m = lane & 31, hi = lane >> 5;
  int gi, h;
  if (!dl_block_of((int)blockIdx.x, DL_H, (p.g.B + p.g.G - 1) / p.g.G, gi, h)) return;
  int u0, nrows;
  int64_t row0;
  dl_group(p.g, gi, u0, row0, nrows);
  const int nutt = nrows / p.g.L, ntile = (p.Tk + 31) >> 5, nit = nutt * ntile;
  DL_STAMP(4, 0);
  DlProB<8, 8> pro;
  pro.issue(p.ln, row0, nrows, tid);
  auto load_tile = [&](DlKvRows& t, int it) {
    const int itc = min(it, nit - 1);                                   // past the end: a valid tile, loaded and never used
    dl_load_kv_rows(t, p.kv, p.kv_bs, p.kv_ts, p.koff, p.voff, p.kmask, p.Tk, u0 + itc / ntile, (itc % ntile) * 32, h, lane);
  };
  DlStream<1, 16, 16> sdo;
  if (wid < 2) sdo.fill(p.wo_t, 16, 2 * h + wid, 1, 0, lane);
  pro.finish(p.ln, nrows, gi, h == 0, ys, red, tid);
  DlKvRows cur, alt;                                                    // issued here (the prologue's registers are free again): they land
  load_tile(cur, slot);                                                 // under the d-context GEMM
  load_tile(alt, slot + 4);
  __syncthreads();
  DL_STAMP(4, 1);
  if (wid < 2) {
    dl_dctx(sdo, ys, dos, dot, dpart, p.ctx16, row0, nrows, h, wid, lane);
  } else if (wid == 2) {
    // the head's queries: row-major and transposed images; lane -> (row lane / 2, 64-byte half)
    const int r = lane >> 1, half = lane & 1;
    const uint16_t* src = p.q16 + (row0 + min(r, nrows - 1)) * DL_D + DL_DK * h + 32 * half;
    uint4 v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = ld_global_b128(src + 8 * j);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      *reinterpret_cast<uint4*>(qs + r * DL_HS + (32 * half + 8 * j) * 2) = v[j];
      const uint32_t w[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
#pragma unroll
      for (int e = 0; e < 8; ++e)
        *reinterpret_cast<uint16_t*>(qt + (32 * half + 8 * j + e) * DL_VT + r * 2) = (uint16_t)((e & 1) ? (w[e >> 1] >> 16) : (w[e >> 1] & 0xffffu));
    }
  } else if (wid == 3) {
    if (lane < DL_RB) {
      const int r = min(lane, nrows - 1), ur = r / p.g.L;
      lses[lane] = p.lse[((int64_t)(u0 + ur) * DL_H + h) * p.g.L + (r - ur * p.g.L)];
    }
  }
  __syncthreads();
  DL_STAMP(4, 2);
  if (tid < DL_RB) dels[tid] = dpart[tid] + dpart[DL_RB + tid];
  __syncthreads();
  // (the Q / dO operand fragments are re-read from LDS for every tile: held in registers across the loops they pushed the wave
  // past its 256 registers, and the spills' scratch round trips cost more than 8 LDS reads)
  float* ob = red;                                                     // [4][32][OS]: the orientation-0 waves' dq partials
  if (orient == 0) {
    // ---- lane = query i, registers = keys j -> dq
    f32x16 dq[2];
    dl_zero(dq);
    const float my_lse = lses[m], my_del = dels[m];
    const int my_u = m / p.g.L;
    unsigned char* kt = ktw + slot * SM_T;
    auto consume = [&](const DlKvRows& t, int it) {
      const int u = it / ntile;
      // K^T image of this tile (this wave's own): lane (j, hi) holds K[j][16 ks + 8 hi + e]
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const uint32_t w[4] = {t.k[ks].x, t.k[ks].y, t.k[ks].z, t.k[ks].w};
#pragma unroll
        for (int e = 0; e < 8; ++e)
          *reinterpret_cast<uint16_t*>(kt + (16 * ks + 8 * hi + e) * DL_VT + m * 2) = (uint16_t)((e & 1) ? (w[e >> 1] >> 16) : (w[e >> 1] & 0xffffu));
      }
      f32x16 st[1], dp[1];
      dl_zero(st); dl_zero(dp);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) { mma32(st[0], t.k[ks], dl_frag(qs, DL_HS, m, hi, ks)); mma32(dp[0], t.v[ks], dl_frag(dos, DL_HS, m, hi, ks)); }
      const bool mine = my_u == u && m < nrows;
      float dsv[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int j = 8 * (r >> 2) + 4 * hi + (r & 3);
        const bool ok = mine && ((t.valid >> j) & 1u);
        const float pr = ok ? __expf(st[0][r] * p.scale - my_lse) : 0.f;
        dsv[r] = pr * (dp[0][r] - my_del);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int k2 = 0; k2 < 2; ++k2) {
        const uint4 pb = dl_pack8(dsv + 8 * k2);
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) mma32(dq[ct], dl_tfrag(kt, 32 * ct + m, hi, k2), pb);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };
    for (int it = slot; it < nit; it += 8) {
      consume(cur, it);
      if (it + 4 < nit) {
        load_tile(cur, it + 8);
        consume(alt, it + 4);
        load_tile(alt, it + 12);
      }
    }
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *reinterpret_cast<float4*>(ob + (slot * DL_RB + m) * OS + 32 * ct + 8 * q + 4 * hi) =
            make_float4(dq[ct][4 * q], dq[ct][4 * q + 1], dq[ct][4 * q + 2], dq[ct][4 * q + 3]);
  } else {
    // ---- lane = key j, registers = queries i -> dk, dv of this tile (complete: one wave owns (utterance, tile))
    unsigned char* og = ogw + slot * (2 * SM_H);                       // this wave's dk | dv tiles on their way out ([32][64] 16-bit each)
    auto consume = [&](const DlKvRows& t, int it) {
      const int u = it / ntile, b = u0 + u, j0 = (it % ntile) * 32, ulo = u * p.g.L;      // the utterance's rows: ulo .. ulo + L - 1
      f32x16 st[1], dp[1];
      dl_zero(st); dl_zero(dp);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) { mma32(st[0], dl_frag(qs, DL_HS, m, hi, ks), t.k[ks]); mma32(dp[0], dl_frag(dos, DL_HS, m, hi, ks), t.v[ks]); }
      const bool keyok = (t.valid >> m) & 1u;
      float pv[16], dsv[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int i = 8 * (r >> 2) + 4 * hi + (r & 3);
        const bool ok = keyok && i < nrows && i >= ulo && i < ulo + p.g.L;
        pv[r] = ok ? __expf(st[0][r] * p.scale - lses[i]) : 0.f;
        dsv[r] = pv[r] * (dp[0][r] - dels[i]) * p.scale;
      }
      // out through this wave's LDS tile: the accumulator layout gives every lane 8-byte pieces of 32 different rows (32 partial
      // lines per store instruction); staged, an instruction writes 8 whole 128-byte rows.  dv first, then dk: one pair of
      // accumulator tiles live at a time (the wave has 256 registers)
      {
        const uint4 pb0 = dl_pack8(pv), pb1 = dl_pack8(pv + 8);
        f32x16 dv[2];
        dl_zero(dv);
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) { mma32(dv[ct], dl_tfrag(dot, 32 * ct + m, hi, 0), pb0); mma32(dv[ct], dl_tfrag(dot, 32 * ct + m, hi, 1), pb1); }
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
          for (int q = 0; q < 4; ++q)
            *reinterpret_cast<uint2*>(og + SM_H + m * DL_HS + (32 * ct + 8 * q + 4 * hi) * 2) =
                make_uint2(pack2h(dv[ct][4 * q], dv[ct][4 * q + 1]), pack2h(dv[ct][4 * q + 2], dv[ct][4 * q + 3]));
      }
      __builtin_amdgcn_sched_barrier(0);
      {
        const uint4 sb0 = dl_pack8(dsv), sb1 = dl_pack8(dsv + 8);
        f32x16 dk[2];
        dl_zero(dk);
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) { mma32(dk[ct], dl_tfrag(qt, 32 * ct + m, hi, 0), sb0); mma32(dk[ct], dl_tfrag(qt, 32 * ct + m, hi, 1), sb1); }
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
          for (int q = 0; q < 4; ++q)
            *reinterpret_cast<uint2*>(og + m * DL_HS + (32 * ct + 8 * q + 4 * hi) * 2) =
                make_uint2(pack2h(dk[ct][4 * q], dk[ct][4 * q + 1]), pack2h(dk[ct][4 * q + 2], dk[ct][4 * q + 3]));
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int j = 8 * q + (lane >> 3), ch = lane & 7;
        if (j0 + j < p.Tk) {
          uint16_t* orow = p.dkv + (int64_t)b * p.kv_bs + (int64_t)(j0 + j) * p.kv_ts + DL_DK * h + 8 * ch;
          st_global_b128(orow + p.koff, *reinterpret_cast<const uint4*>(og + j * DL_HS + ch * 16));
          st_global_b128(orow + p.voff, *reinterpret_cast<const uint4*>(og + SM_H + j * DL_HS + ch * 16));
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };
    for (int it = slot; it < nit; it += 8) {
      consume(cur, it);
      if (it + 4 < nit) {
        load_tile(cur, it + 8);
        consume(alt, it + 4);
        load_tile(alt, it + 12);
      }
    }
  }
  DL_STAMP(4, 3);
  // ---- the four orientation-0 waves' dq partials: sum -> 16-bit image + memory
  DlStream<1, 4, 4> sdy;                                                // (filled here: inside the loops its 16 registers spilled)
  sdy.fill(p.wq_t, 16, wid, 1, 4 * h, lane);
  __syncthreads();
  {
    const int r = tid >> 4, c = (tid & 15) * 4;                        // 512 threads: row r, four columns of the head
    float a4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const float4 t = *reinterpret_cast<const float4*>(ob + (w * DL_RB + r) * OS + c);
      a4[0] += t.x; a4[1] += t.y; a4[2] += t.z; a4[3] += t.w;
    }
    const uint2 pk = make_uint2(pack2h(a4[0] * p.scale, a4[1] * p.scale), pack2h(a4[2] * p.scale, a4[3] * p.scale));
    *reinterpret_cast<uint2*>(dqs + r * DL_HS + c * 2) = pk;
    if (r < nrows) *reinterpret_cast<uint2*>(p.dq16 + (row0 + r) * DL_D + DL_DK * h + c) = pk;
  }
  __syncthreads();
  DL_STAMP(4, 4);
  f32x16 acc[1];
  dl_zero(acc);
  sdy.run(acc, dqs, DL_HS, 0, lane);
  dl_put_tile(red, acc[0], wid * 32, lane);
  __syncthreads();
  dl_store_slab<8>(p.slabs + (int64_t)h * p.ln.R * DL_D, red, row0, nrows, tid);
  DL_STAMP(4, 5);
}

// ------------------------------------------------------------------------------------------------ self-attention backward launch
struct DlSelfBwdArgs {
  unsigned long long* trace;
  DlLnB ln;
  DlGeom g;
  const uint4* wo_t;       // input-gradient pack of output_proj.weight
  const uint4* wqkv_t;     // input-gradient pack of qvk_proj.weight: rows = 256 inputs, contraction = 768 (q | k | v)
  const uint16_t* qkv16; const uint16_t* ctx16; const float* lse;
  uint16_t* dqkv16;        // [R,768] out
  uint16_t* slabs;         // [H][R][256] 16-bit out: dqkv_h . W_qkv[head rows]
  float scale;
};

__global__ __launch_bounds__(256, 1) void dec_self_bwd_kernel(DlSelfBwdArgs p) {
  constexpr int SM_IMG = DL_RB * DL_YS, SM_H = DL_RB * DL_HS, SM_T = DL_DK * DL_VT, GS = 3 * DL_DK * 2 + 16;   // GS: bytes per row of the dqkv image
  __shared__ __attribute__((aligned(16))) unsigned char smem[SM_IMG + 4 * SM_H + 3 * SM_T + DL_RB * GS + 4 * DL_RB * 4 + DL_RB * DL_RS * 4];
  unsigned char* ys = smem;
  unsigned char* qs = ys + SM_IMG;
  unsigned char* ks_ = qs + SM_H;
  unsigned char* vs = ks_ + SM_H;
  unsigned char* dos = vs + SM_H;
  unsigned char* qt = dos + SM_H;
  unsigned char* kt = qt + SM_T;
  unsigned char* dot = kt + SM_T;
  unsigned char* gs = dot + SM_T;                 // dq | dk | dv of the head, [32][192] 16-bit
  float* lses = reinterpret_cast<float*>(gs + DL_RB * GS);
  float* dels = lses + DL_RB;
  float* dpart = dels + DL_RB;
  float* red = dpart + 2 * DL_RB;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 31, hi = lane >> 5;
  int gi, h;
  if (!dl_block_of((int)blockIdx.x, DL_H, (p.g.B + p.g.G - 1) / p.g.G, gi, h)) return;
  int u0, nrows;
  int64_t row0;
  dl_group(p.g, gi, u0, row0, nrows);
  DL_STAMP(5, 0);
  DlProB<4, 4> pro;
  pro.issue(p.ln, row0, nrows, tid);
  DlStream<1, 16, 16> sdo;
  if (wid < 2) sdo.fill(p.wo_t, 16, 2 * h + wid, 1, 0, lane);
  DlStream<2, 4, 8> sg[3];                        // the head's q, k, v columns = contraction steps 4h.., 16 + 4h.., 32 + 4h.. of the 48
#pragma unroll
  for (int j = 0; j < 3; ++j) sg[j].fill(p.wqkv_t, 48, 2 * wid, 1, 16 * j + 4 * h, lane);
  pro.finish(p.ln, nrows, gi, h == 0, ys, red, tid);
  // the head's q, k, v rows: row-major images (all three), transposed images of q and k
  {
    const int part = tid >> 6;                    // wave 0: q, 1: k, 2: v, 3: the softmax statistics
    if (part < 3) {
      const int r = lane >> 1, half = lane & 1;
      const uint16_t* src = p.qkv16 + (row0 + min(r, nrows - 1)) * (3 * DL_D) + part * DL_D + DL_DK * h + 32 * half;
      unsigned char* img = part == 0 ? qs : part == 1 ? ks_ : vs;
      unsigned char* timg = part == 0 ? qt : kt;
      uint4 v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = ld_global_b128(src + 8 * j);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        *reinterpret_cast<uint4*>(img + r * DL_HS + (32 * half + 8 * j) * 2) = v[j];
        if (part < 2) {
          const uint32_t w[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
#pragma unroll
          for (int e = 0; e < 8; ++e)
            *reinterpret_cast<uint16_t*>(timg + (32 * half + 8 * j + e) * DL_VT + r * 2) = (uint16_t)((e & 1) ? (w[e >> 1] >> 16) : (w[e >> 1] & 0xffffu));
        }
      }
    } else if (lane < DL_RB) {
      const int r = min(lane, nrows - 1), ur = r / p.g.L;
      lses[lane] = p.lse[((int64_t)(u0 + ur) * DL_H + h) * p.g.L + (r - ur * p.g.L)];
    }
  }
  __syncthreads();
  DL_STAMP(5, 1);
  if (wid < 2) dl_dctx(sdo, ys, dos, dot, dpart, p.ctx16, row0, nrows, h, wid, lane);
  __syncthreads();
  DL_STAMP(5, 2);
  if (tid < DL_RB) dels[tid] = dpart[tid] + dpart[DL_RB + tid];
  __syncthreads();
  if (wid == 0) {
    // orientation 1: lane = query i, registers = keys j -> dq
    f32x16 st[1], dp[1];
    dl_zero(st); dl_zero(dp);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const uint4 qb = dl_frag(qs, DL_HS, m, hi, ks), ob_ = dl_frag(dos, DL_HS, m, hi, ks);
      mma32(st[0], dl_frag(ks_, DL_HS, m, hi, ks), qb);
      mma32(dp[0], dl_frag(vs, DL_HS, m, hi, ks), ob_);
    }
    const int i = m, ui = i / p.g.L;
    const float my_lse = lses[i], my_del = dels[i];
    float dsv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int j = 8 * (r >> 2) + 4 * hi + (r & 3);
      const bool ok = j <= i && i < nrows && j >= ui * p.g.L;
      const float pr = ok ? __expf(st[0][r] * p.scale - my_lse) : 0.f;
      dsv[r] = pr * (dp[0][r] - my_del) * p.scale;
    }
    f32x16 dq[2];
    dl_zero(dq);
#pragma unroll
    for (int k2 = 0; k2 < 2; ++k2) {
      const uint4 pb = dl_pack8(dsv + 8 * k2);
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) mma32(dq[ct], dl_tfrag(kt, 32 * ct + m, hi, k2), pb);
    }
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c = 32 * ct + 8 * q + 4 * hi;
        const uint2 pk = make_uint2(pack2h(dq[ct][4 * q], dq[ct][4 * q + 1]), pack2h(dq[ct][4 * q + 2], dq[ct][4 * q + 3]));
        *reinterpret_cast<uint2*>(gs + i * GS + c * 2) = pk;
        if (i < nrows) *reinterpret_cast<uint2*>(p.dqkv16 + (row0 + i) * (3 * DL_D) + DL_DK * h + c) = pk;
      }
  } else if (wid == 1) {
    // orientation 2: lane = key j, registers = queries i -> dk, dv
    f32x16 st[1], dp[1];
    dl_zero(st); dl_zero(dp);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const uint4 kb = dl_frag(ks_, DL_HS, m, hi, ks), vb = dl_frag(vs, DL_HS, m, hi, ks);
      mma32(st[0], dl_frag(qs, DL_HS, m, hi, ks), kb);
      mma32(dp[0], dl_frag(dos, DL_HS, m, hi, ks), vb);
    }
    const int j = m, uj = j / p.g.L;
    float pv[16], dsv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = 8 * (r >> 2) + 4 * hi + (r & 3);
      const bool ok = j <= i && i < nrows && i < (uj + 1) * p.g.L;
      pv[r] = ok ? __expf(st[0][r] * p.scale - lses[i]) : 0.f;
      dsv[r] = pv[r] * (dp[0][r] - dels[i]) * p.scale;
    }
    f32x16 dv[2], dk[2];
    dl_zero(dv); dl_zero(dk);
#pragma unroll
    for (int k2 = 0; k2 < 2; ++k2) {
      const uint4 pb = dl_pack8(pv + 8 * k2), sb = dl_pack8(dsv + 8 * k2);
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        mma32(dv[ct], dl_tfrag(dot, 32 * ct + m, hi, k2), pb);
        mma32(dk[ct], dl_tfrag(qt, 32 * ct + m, hi, k2), sb);
      }
    }
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c = 32 * ct + 8 * q + 4 * hi;
        const uint2 pk_k = make_uint2(pack2h(dk[ct][4 * q], dk[ct][4 * q + 1]), pack2h(dk[ct][4 * q + 2], dk[ct][4 * q + 3]));
        const uint2 pk_v = make_uint2(pack2h(dv[ct][4 * q], dv[ct][4 * q + 1]), pack2h(dv[ct][4 * q + 2], dv[ct][4 * q + 3]));
        *reinterpret_cast<uint2*>(gs + j * GS + (DL_DK + c) * 2) = pk_k;
        *reinterpret_cast<uint2*>(gs + j * GS + (2 * DL_DK + c) * 2) = pk_v;
        if (j < nrows) {
          *reinterpret_cast<uint2*>(p.dqkv16 + (row0 + j) * (3 * DL_D) + DL_D + DL_DK * h + c) = pk_k;
          *reinterpret_cast<uint2*>(p.dqkv16 + (row0 + j) * (3 * DL_D) + 2 * DL_D + DL_DK * h + c) = pk_v;
        }
      }
  }
  __syncthreads();
  DL_STAMP(5, 3);
  f32x16 acc[2];
  dl_zero(acc);
#pragma unroll
  for (int j = 0; j < 3; ++j) sg[j].run(acc, gs, GS, 4 * j, lane);
  dl_put_tile(red, acc[0], (2 * wid) * 32, lane);
  dl_put_tile(red, acc[1], (2 * wid + 1) * 32, lane);
  __syncthreads();
  dl_store_slab<4>(p.slabs + (int64_t)h * p.ln.R * DL_D, red, row0, nrows, tid);
  DL_STAMP(5, 4);
}

// dx = skip + sum of slabs
__global__ __launch_bounds__(256) void dec_sum_kernel(const float* skip, const uint16_t* slabs, int nslab, int64_t R, float* out) {
  const int64_t n4 = R * (DL_D / 4);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    float4 a = skip ? reinterpret_cast<const float4*>(skip)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    for (int s = 0; s < nslab; ++s) {
      const uint2 t = reinterpret_cast<const uint2*>(slabs + (int64_t)s * R * DL_D)[i];
      a.x += h2f_lo(t.x); a.y += h2f_hi(t.x); a.z += h2f_lo(t.y); a.w += h2f_hi(t.y);
    }
    reinterpret_cast<float4*>(out)[i] = a;
  }
}

int32_t dl_check_lnb(const char* who, const otr_dec_lnb_t* ln, int64_t R, DlLnB& o) {
  OTR_REQUIRE(ln != nullptr, "%s: null LayerNorm descriptor", who);
  OTR_REQUIRE(ln->nslab >= 0 && ln->nslab <= 64 && (ln->nslab == 0 || ln->slabs) && (ln->nslab > 0 || ln->dskip), "%s: bad gradient inputs", who);
  OTR_REQUIRE(ln->z && ln->mean && ln->rstd && ln->gamma, "%s: null LayerNorm input", who);
  OTR_REQUIRE(ln->p_drop >= 0.f && ln->p_drop < 1.f && (ln->p_drop == 0.f || ln->seed), "%s: bad dropout arguments", who);
  OTR_REQUIRE(((uintptr_t)ln->dskip | (uintptr_t)ln->slabs | (uintptr_t)ln->z | (uintptr_t)ln->gamma | (uintptr_t)ln->dz | (uintptr_t)ln->da16 |
               (uintptr_t)ln->partial) % 16 == 0, "%s: LayerNorm buffers must be 16-byte aligned", who);
  o.dskip = ln->dskip; o.slabs = ln->slabs; o.nslab = ln->nslab; o.z = ln->z; o.mean = ln->mean; o.rstd = ln->rstd; o.gamma = ln->gamma;
  o.seed = ln->seed; o.p_drop = ln->p_drop; o.rng_offset = ln->rng_offset; o.dz = ln->dz; o.da16 = (uint16_t*)ln->da16; o.partial = ln->partial;
  o.R = R;
  return 0;
}

int32_t dl_check_ln(const char* who, const otr_dec_ln_t* ln, int64_t R, DlLn& o) {
  OTR_REQUIRE(ln != nullptr, "%s: null LayerNorm descriptor", who);
  OTR_REQUIRE(ln->nslab >= 0 && ln->nslab <= 64, "%s: bad slab count %d", who, ln->nslab);
  if (ln->nslab == 0) {
    OTR_REQUIRE(ln->x16 != nullptr, "%s: nslab == 0 needs the rows' 16-bit twin", who);
  } else {
    OTR_REQUIRE(ln->xres && ln->slabs && ln->gamma && ln->beta, "%s: null LayerNorm input", who);
    OTR_REQUIRE(ln->p_drop >= 0.f && ln->p_drop < 1.f && (ln->p_drop == 0.f || ln->seed), "%s: bad dropout arguments", who);
  }
  OTR_REQUIRE(((uintptr_t)ln->xres | (uintptr_t)ln->x16 | (uintptr_t)ln->slabs | (uintptr_t)ln->bias | (uintptr_t)ln->gamma | (uintptr_t)ln->beta |
               (uintptr_t)ln->y | (uintptr_t)ln->y16 | (uintptr_t)ln->z) % 16 == 0, "%s: LayerNorm buffers must be 16-byte aligned", who);
  o.xres = ln->xres; o.x16 = (const uint16_t*)ln->x16; o.slabs = ln->slabs; o.nslab = ln->nslab; o.bias = ln->bias; o.gamma = ln->gamma;
  o.beta = ln->beta; o.seed = ln->seed; o.p_drop = ln->p_drop; o.eps = ln->eps; o.rng_offset = ln->rng_offset;
  o.y = ln->y; o.y16 = (uint16_t*)ln->y16; o.z = ln->z; o.mean = ln->mean; o.rstd = ln->rstd; o.R = R;
  return 0;
}

// Utterances per (group, head) workgroup of the attention launches.  A 32-row tile holds 32 / L whole utterances, but the launch is
// a grid of (groups x 4 heads) latency chains on 256 CUs: at the AISHELL batch (B = 32, L = 15) full tiles make 64 workgroups, each
// walking 16 (utterance, key tile) pairs in its flash loops; ONE utterance per group makes 128 with 8 pairs each -- the padding rows of
// the tile cost MFMA work nobody waits for (mfma_busy 0.01), the loops are what the launch waits for.  So: the largest group that
// still leaves >= 256 workgroups, else one utterance per group.  otr_debug_set(23, v) caps it by hand (v >= 32 / L = full tiles, the
// round-4 geometry).  The kernels never assumed full groups (the last group of a batch is ragged anyway).
static inline int dl_group_size(int B, int L) {
  const int gmax = DL_RB / L;
  if (g_otr_dec_group > 0) return g_otr_dec_group < gmax ? g_otr_dec_group : gmax;
  int G = gmax;
  while (G > 1 && ((B + G - 1) / G) * DL_H < 256) --G;
  return G;
}

int32_t dl_check_geom(const char* who, int32_t B, int32_t L, DlGeom& g) {
  OTR_REQUIRE(B > 0 && L > 0 && L <= DL_RB, "%s: %d utterances x %d decoder rows: a group is 32 / L whole utterances, L <= 32", who, B, L);
  g.B = B; g.L = L; g.G = dl_group_size(B, L);
  return 0;
}

}  // namespace
extern unsigned long long* g_otr_trace;   // api.hip (otr_debug_trace)

extern "C" int32_t otr_dec_group_size(int32_t B, int32_t L) {
  if (B <= 0 || L <= 0 || L > DL_RB) return 0;
  return dl_group_size(B, L);
}

extern "C" int32_t otr_dec_self_fwd(const otr_dec_ln_t* ln, int32_t B, int32_t L, const void* wqkv_pack, const float* bqkv,
                                    const void* wo_pack, void* qkv16, void* ctx16, float* lse, void* slabs, void* stream) {
  DlSelfArgs a{};
  if (int32_t e = dl_check_geom("dec_self_fwd", B, L, a.g)) return e;
  if (int32_t e = dl_check_ln("dec_self_fwd", ln, (int64_t)B * L, a.ln)) return e;
  OTR_REQUIRE(wqkv_pack && bqkv && wo_pack && qkv16 && ctx16 && lse && slabs, "dec_self_fwd: null pointer");
  OTR_REQUIRE(((uintptr_t)wqkv_pack | (uintptr_t)bqkv | (uintptr_t)wo_pack | (uintptr_t)qkv16 | (uintptr_t)ctx16 | (uintptr_t)slabs) % 16 == 0,
              "dec_self_fwd: buffers must be 16-byte aligned");
  a.wqkv = (const uint4*)wqkv_pack; a.bqkv = bqkv; a.wo = (const uint4*)wo_pack; a.qkv16 = (uint16_t*)qkv16; a.ctx16 = (uint16_t*)ctx16;
  a.lse = lse; a.slabs = (uint16_t*)slabs; a.scale = 0.125f;          // 1 / sqrt(64)
  a.trace = g_otr_trace;
  hipLaunchKernelGGL(dec_self_fwd_kernel, dim3(dl_grid(DL_H, (B + a.g.G - 1) / a.g.G)), dim3(256), 0, (hipStream_t)stream, a);
  return otr_check_launch("dec_self_fwd");
}

static int32_t dl_fill_self_step(DlStepArgs& a, const otr_dec_ln_t* ln, int64_t R, const void* wqkv_pack, const float* bqkv, const void* wo_pack,
                                 void* kcache, void* vcache, const int32_t* anc, const int32_t* pos, int32_t maxlen, void* slabs) {
  OTR_REQUIRE(R > 0 && R < (1ll << 31) / 4, "dec_self_step: bad row count");
  if (int32_t e = dl_check_ln("dec_self_step", ln, R, a.ln)) return e;
  OTR_REQUIRE(wqkv_pack && bqkv && wo_pack && kcache && vcache && anc && pos && slabs, "dec_self_step: null pointer");
  OTR_REQUIRE(maxlen > 0, "dec_self_step: bad cache length %d", maxlen);
  OTR_REQUIRE(ln->nslab <= 16 && (ln->nslab == 0 || ln->p_drop == 0.f), "dec_self_step: at most 16 slabs, no dropout (nslab=%d p_drop=%g)",
              ln->nslab, (double)ln->p_drop);
  OTR_REQUIRE(((uintptr_t)wqkv_pack | (uintptr_t)bqkv | (uintptr_t)wo_pack | (uintptr_t)kcache | (uintptr_t)vcache | (uintptr_t)slabs) % 16 == 0,
              "dec_self_step: buffers must be 16-byte aligned");
  a.wqkv = (const uint4*)wqkv_pack; a.bqkv = bqkv; a.wo = (const uint4*)wo_pack; a.kc = (uint16_t*)kcache; a.vc = (uint16_t*)vcache;
  a.anc = anc; a.pos = pos; a.maxlen = maxlen; a.slabs = (uint16_t*)slabs; a.scale = 0.125f; a.trace = g_otr_trace;
  return 0;
}
extern "C" int32_t otr_dec_self_step(const otr_dec_ln_t* ln, int64_t R, const void* wqkv_pack, const float* bqkv, const void* wo_pack,
                                     void* kcache, void* vcache, const int32_t* anc, const int32_t* pos, int32_t maxlen, void* slabs,
                                     void* stream) {
  DlStepArgs a{};
  if (int32_t e = dl_fill_self_step(a, ln, R, wqkv_pack, bqkv, wo_pack, kcache, vcache, anc, pos, maxlen, slabs)) return e;
  hipLaunchKernelGGL(dec_self_step_kernel, dim3(dl_grid(DL_H, (int)((R + DL_SB - 1) / DL_SB))), dim3(512), 0, (hipStream_t)stream, a);
  return otr_check_launch("dec_self_step");
}
extern "C" int32_t otr_dec_self_step_pair(const otr_dec_self_step_t* x, const otr_dec_self_step_t* y, void* stream) {
  OTR_REQUIRE(x && y, "dec_self_step_pair: null descriptor");
  DlStepPair pp{};
  if (int32_t e = dl_fill_self_step(pp.a, &x->ln, x->R, x->wqkv_pack, x->bqkv, x->wo_pack, x->kcache, x->vcache, x->anc, x->pos, x->maxlen, x->slabs)) return e;
  if (int32_t e = dl_fill_self_step(pp.b, &y->ln, y->R, y->wqkv_pack, y->bqkv, y->wo_pack, y->kcache, y->vcache, y->anc, y->pos, y->maxlen, y->slabs)) return e;
  pp.n0 = (int)dl_grid(DL_H, (int)((x->R + DL_SB - 1) / DL_SB));
  const unsigned n1 = dl_grid(DL_H, (int)((y->R + DL_SB - 1) / DL_SB));
  hipLaunchKernelGGL(dec_self_step_pair_kernel, dim3((unsigned)pp.n0 + n1), dim3(512), 0, (hipStream_t)stream, pp);
  return otr_check_launch("dec_self_step_pair");
}

extern "C" int32_t otr_dec_cross_fwd(const otr_dec_ln_t* ln, int32_t B, int32_t L, const void* wq_pack, const float* bq, const void* wo_pack,
                                     const void* kv, int64_t kv_bs, int64_t kv_ts, int32_t koff, int32_t voff, const uint8_t* key_mask,
                                     int32_t Tk, void* q16, void* ctx16, float* lse, void* slabs, void* stream) {
  DlCrossArgs a{};
  if (int32_t e = dl_check_geom("dec_cross_fwd", B, L, a.g)) return e;
  if (int32_t e = dl_check_ln("dec_cross_fwd", ln, (int64_t)B * L, a.ln)) return e;
  OTR_REQUIRE(wq_pack && bq && wo_pack && kv && q16 && ctx16 && lse && slabs, "dec_cross_fwd: null pointer");
  OTR_REQUIRE(Tk > 0 && kv_ts % 8 == 0 && kv_bs % 8 == 0 && koff % 8 == 0 && voff % 8 == 0, "dec_cross_fwd: key / value rows must be 16-byte aligned");
  OTR_REQUIRE(((uintptr_t)wq_pack | (uintptr_t)bq | (uintptr_t)wo_pack | (uintptr_t)kv | (uintptr_t)q16 | (uintptr_t)ctx16 | (uintptr_t)slabs) % 16 == 0,
              "dec_cross_fwd: buffers must be 16-byte aligned");
  a.wq = (const uint4*)wq_pack; a.bq = bq; a.wo = (const uint4*)wo_pack; a.kv = (const uint16_t*)kv; a.kv_bs = kv_bs; a.kv_ts = kv_ts;
  a.koff = koff; a.voff = voff; a.kmask = key_mask; a.Tk = Tk; a.q16 = (uint16_t*)q16; a.ctx16 = (uint16_t*)ctx16; a.lse = lse; a.slabs = (uint16_t*)slabs;
  a.scale = 0.125f; a.trace = g_otr_trace;
  hipLaunchKernelGGL(dec_cross_fwd_kernel, dim3(dl_grid(DL_H, (B + a.g.G - 1) / a.g.G)), dim3(512), 0, (hipStream_t)stream, a);
  return otr_check_launch("dec_cross_fwd");
}

extern "C" int64_t otr_dec_ffn_hsave_bytes(int64_t R, int32_t F) { return R > 0 && F > 0 ? ((R + DL_RB - 1) / DL_RB) * (int64_t)(F / 32) * 4096 : 0; }

static int32_t dl_fill_ffn_fwd(DlFfnArgs& a, const otr_dec_ln_t* ln, int64_t R, const void* w1_pack, const float* b1, const void* w2_pack, int32_t F,
                               int32_t S, void* slabs, void* hsave) {
  OTR_REQUIRE(R > 0 && R < (1ll << 31), "dec_ffn_fwd: bad row count");
  if (int32_t e = dl_check_ln("dec_ffn_fwd", ln, R, a.ln)) return e;
  OTR_REQUIRE(w1_pack && b1 && w2_pack && slabs, "dec_ffn_fwd: null pointer");
  OTR_REQUIRE(F > 0 && S > 0 && F % (128 * S) == 0, "dec_ffn_fwd: d_ff = %d does not split into %d slices of whole 128-unit wave rounds", F, S);
  OTR_REQUIRE(((uintptr_t)w1_pack | (uintptr_t)b1 | (uintptr_t)w2_pack | (uintptr_t)slabs | (uintptr_t)hsave) % 16 == 0, "dec_ffn_fwd: buffers must be 16-byte aligned");
  a.p1 = (const uint4*)w1_pack; a.b1 = b1; a.p2 = (const uint4*)w2_pack; a.slabs = (uint16_t*)slabs; a.hsave = (uint4*)hsave; a.F = F; a.S = S; a.trace = g_otr_trace;
  return 0;
}
extern "C" int32_t otr_dec_ffn_fwd(const otr_dec_ln_t* ln, int64_t R, const void* w1_pack, const float* b1, const void* w2_pack, int32_t F,
                                   int32_t S, void* slabs, void* hsave, void* stream) {
  DlFfnArgs a{};
  if (int32_t e = dl_fill_ffn_fwd(a, ln, R, w1_pack, b1, w2_pack, F, S, slabs, hsave)) return e;
  hipLaunchKernelGGL(dec_ffn_fwd_kernel, dim3(dl_grid(S, (int)((R + DL_RB - 1) / DL_RB))), dim3(256), 0, (hipStream_t)stream, a);
  return otr_check_launch("dec_ffn_fwd");
}
extern "C" int32_t otr_dec_ffn_fwd_pair(const otr_dec_ffn_fwd_t* x, const otr_dec_ffn_fwd_t* y, void* stream) {
  OTR_REQUIRE(x && y, "dec_ffn_fwd_pair: null descriptor");
  DlFfnPair pp{};
  if (int32_t e = dl_fill_ffn_fwd(pp.a, &x->ln, x->R, x->w1_pack, x->b1, x->w2_pack, x->F, x->S, x->slabs, x->hsave)) return e;
  if (int32_t e = dl_fill_ffn_fwd(pp.b, &y->ln, y->R, y->w1_pack, y->b1, y->w2_pack, y->F, y->S, y->slabs, y->hsave)) return e;
  pp.n0 = (int)dl_grid(x->S, (int)((x->R + DL_RB - 1) / DL_RB));
  const unsigned n1 = dl_grid(y->S, (int)((y->R + DL_RB - 1) / DL_RB));
  hipLaunchKernelGGL(dec_ffn_fwd_pair_kernel, dim3((unsigned)pp.n0 + n1), dim3(256), 0, (hipStream_t)stream, pp);
  return otr_check_launch("dec_ffn_fwd_pair");
}

extern "C" int32_t otr_dec_ln(const otr_dec_ln_t* ln, int64_t R, void* stream) {
  DlLnArgs a{};
  OTR_REQUIRE(R > 0 && R < (1ll << 31), "dec_ln: bad row count");
  if (int32_t e = dl_check_ln("dec_ln", ln, R, a.ln)) return e;
  OTR_REQUIRE(ln->nslab > 0, "dec_ln: nothing to normalise");
  hipLaunchKernelGGL(dec_ln_kernel, dim3((unsigned)((R + 7) / 8)), dim3(256), 0, (hipStream_t)stream, a);
  return otr_check_launch("dec_ln");
}

extern "C" int32_t otr_dec_ln_pair(const otr_dec_ln_t* lx, int64_t Rx, const otr_dec_ln_t* ly, int64_t Ry, void* stream) {
  DlLnPair pp{};
  OTR_REQUIRE(Rx > 0 && Rx < (1ll << 31) && Ry > 0 && Ry < (1ll << 31), "dec_ln_pair: bad row count");
  if (int32_t e = dl_check_ln("dec_ln_pair", lx, Rx, pp.a.ln)) return e;
  if (int32_t e = dl_check_ln("dec_ln_pair", ly, Ry, pp.b.ln)) return e;
  OTR_REQUIRE(lx->nslab > 0 && ly->nslab > 0, "dec_ln_pair: nothing to normalise");
  pp.n0 = (int)((Rx + 7) / 8);
  hipLaunchKernelGGL(dec_ln_pair_kernel, dim3((unsigned)(pp.n0 + (int)((Ry + 7) / 8))), dim3(256), 0, (hipStream_t)stream, pp);
  return otr_check_launch("dec_ln_pair");
}

extern "C" int32_t otr_dec_ffn_bwd(const otr_dec_lnb_t* ln, int64_t R, const void* hsave, const void* w2t_pack, const void* w1t_pack,
                                   int32_t F, int32_t S, void* dh, void* u, float* db1_part, void* slabs, void* stream) {
  DlFfnBwdArgs a{};
  OTR_REQUIRE(R > 0 && R < (1ll << 31), "dec_ffn_bwd: bad row count");
  if (int32_t e = dl_check_lnb("dec_ffn_bwd", ln, R, a.ln)) return e;
  OTR_REQUIRE(hsave && w2t_pack && w1t_pack && dh && u && db1_part && slabs, "dec_ffn_bwd: null pointer");
  OTR_REQUIRE(F > 0 && S > 0 && F % (128 * S) == 0, "dec_ffn_bwd: d_ff = %d does not split into %d slices of whole 128-unit wave rounds", F, S);
  OTR_REQUIRE(((uintptr_t)hsave | (uintptr_t)w2t_pack | (uintptr_t)w1t_pack | (uintptr_t)dh | (uintptr_t)u |
               (uintptr_t)db1_part | (uintptr_t)slabs) % 16 == 0, "dec_ffn_bwd: buffers must be 16-byte aligned");
  a.hsave = (const uint4*)hsave; a.p3 = (const uint4*)w2t_pack; a.p4 = (const uint4*)w1t_pack;
  a.dh = (uint16_t*)dh; a.u = (uint16_t*)u; a.bpart = db1_part; a.slabs = (uint16_t*)slabs; a.F = F; a.S = S; a.trace = g_otr_trace;
  hipLaunchKernelGGL(dec_ffn_bwd_kernel, dim3(dl_grid(S, (int)((R + DL_RB - 1) / DL_RB))), dim3(256), 0, (hipStream_t)stream, a);
  return otr_check_launch("dec_ffn_bwd");
}

extern "C" int32_t otr_dec_cross_bwd(const otr_dec_lnb_t* ln, int32_t B, int32_t L, const void* wo_dgrad_pack, const void* wq_dgrad_pack,
                                     const void* q16, const void* ctx16, const float* lse, const void* kv, void* dkv, int64_t kv_bs,
                                     int64_t kv_ts, int32_t koff, int32_t voff, const uint8_t* key_mask, int32_t Tk, void* dq16, void* slabs,
                                     void* stream) {
  DlCrossBwdArgs a{};
  if (int32_t e = dl_check_geom("dec_cross_bwd", B, L, a.g)) return e;
  if (int32_t e = dl_check_lnb("dec_cross_bwd", ln, (int64_t)B * L, a.ln)) return e;
  OTR_REQUIRE(wo_dgrad_pack && wq_dgrad_pack && q16 && ctx16 && lse && kv && dkv && dq16 && slabs, "dec_cross_bwd: null pointer");
  OTR_REQUIRE(Tk > 0 && kv_ts % 8 == 0 && kv_bs % 8 == 0 && koff % 8 == 0 && voff % 8 == 0, "dec_cross_bwd: key / value rows must be 16-byte aligned");
  OTR_REQUIRE(((uintptr_t)wo_dgrad_pack | (uintptr_t)wq_dgrad_pack | (uintptr_t)q16 | (uintptr_t)ctx16 | (uintptr_t)kv | (uintptr_t)dkv |
               (uintptr_t)dq16 | (uintptr_t)slabs) % 16 == 0, "dec_cross_bwd: buffers must be 16-byte aligned");
  a.wo_t = (const uint4*)wo_dgrad_pack; a.wq_t = (const uint4*)wq_dgrad_pack; a.q16 = (const uint16_t*)q16; a.ctx16 = (const uint16_t*)ctx16;
  a.lse = lse; a.kv = (const uint16_t*)kv; a.dkv = (uint16_t*)dkv; a.kv_bs = kv_bs; a.kv_ts = kv_ts; a.koff = koff; a.voff = voff;
  a.kmask = key_mask; a.Tk = Tk; a.dq16 = (uint16_t*)dq16; a.slabs = (uint16_t*)slabs; a.scale = 0.125f; a.trace = g_otr_trace;
  hipLaunchKernelGGL(dec_cross_bwd_kernel, dim3(dl_grid(DL_H, (B + a.g.G - 1) / a.g.G)), dim3(512), 0, (hipStream_t)stream, a);
  return otr_check_launch("dec_cross_bwd");
}

extern "C" int32_t otr_dec_self_bwd(const otr_dec_lnb_t* ln, int32_t B, int32_t L, const void* wo_dgrad_pack, const void* wqkv_dgrad_pack,
                                    const void* qkv16, const void* ctx16, const float* lse, void* dqkv16, void* slabs, void* stream) {
  DlSelfBwdArgs a{};
  if (int32_t e = dl_check_geom("dec_self_bwd", B, L, a.g)) return e;
  if (int32_t e = dl_check_lnb("dec_self_bwd", ln, (int64_t)B * L, a.ln)) return e;
  OTR_REQUIRE(wo_dgrad_pack && wqkv_dgrad_pack && qkv16 && ctx16 && lse && dqkv16 && slabs, "dec_self_bwd: null pointer");
  OTR_REQUIRE(((uintptr_t)wo_dgrad_pack | (uintptr_t)wqkv_dgrad_pack | (uintptr_t)qkv16 | (uintptr_t)ctx16 | (uintptr_t)dqkv16 | (uintptr_t)slabs) % 16 == 0,
              "dec_self_bwd: buffers must be 16-byte aligned");
  a.wo_t = (const uint4*)wo_dgrad_pack; a.wqkv_t = (const uint4*)wqkv_dgrad_pack; a.qkv16 = (const uint16_t*)qkv16; a.ctx16 = (const uint16_t*)ctx16;
  a.lse = lse; a.dqkv16 = (uint16_t*)dqkv16; a.slabs = (uint16_t*)slabs; a.scale = 0.125f; a.trace = g_otr_trace;
  hipLaunchKernelGGL(dec_self_bwd_kernel, dim3(dl_grid(DL_H, (B + a.g.G - 1) / a.g.G)), dim3(256), 0, (hipStream_t)stream, a);
  return otr_check_launch("dec_self_bwd");
}

extern "C" int32_t otr_dec_sum(const float* skip, const void* slabs, int32_t nslab, int64_t R, float* out, void* stream) {
  OTR_REQUIRE(out && R > 0 && nslab >= 0 && (nslab == 0 || slabs), "dec_sum: bad arguments");
  OTR_REQUIRE(((uintptr_t)skip | (uintptr_t)slabs | (uintptr_t)out) % 16 == 0, "dec_sum: buffers must be 16-byte aligned");
  const int64_t n4 = R * (DL_D / 4);
  hipLaunchKernelGGL(dec_sum_kernel, dim3((unsigned)((n4 + 255) / 256 > 1024 ? 1024 : (n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, skip,
                     (const uint16_t*)slabs, nslab, R, out);
  return otr_check_launch("dec_sum");
}
