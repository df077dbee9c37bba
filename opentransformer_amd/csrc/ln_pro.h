// The LayerNorm a launch finishes in its PROLOGUE from the partial sums ("slabs") the launch before it left: shared by the fused
// decoder launches (declayer.hip) and the encoder's q|k|v projection behind a split FFN in slab mode (rowblock.hip, ffn3.hip).
#pragma once
#include <type_traits>

#include "common.h"

constexpr int DL_D = 256, DL_RB = 32;

struct DlLn {              // the LayerNorm a launch finishes in its prologue: y = LN(xres + dropout(sum_s slabs[s] + bias))
  const float* xres;       // [R, 256] residual input
  const uint16_t* x16;     // nslab == 0: there is nothing to finish, the rows pass through (their 16-bit twin is given)
  const void* slabs;       // [nslab][R][256] partial sums of the branch: fp32 (attention head shares) or 16-bit (FFN slices), fixed per launch type
  int nslab;
  const float* bias; const float* gamma; const float* beta; const uint64_t* seed;
  float p_drop, eps;
  uint64_t rng_offset;
  float* y; uint16_t* y16; float* z; float* mean; float* rstd;      // outputs (written by ONE workgroup per row block)
  int64_t R;
};

// The LayerNorm prologue of a forward launch, in two steps.  issue() starts EVERY global load it needs (the residual rows and all
// slabs of the workgroup's 32 rows) before the caller starts its prefetches (weights, keys / values): vmcnt retires in order, and
// with the prefetches issued first the LayerNorm waited for ~200 KiB it did not need (7-10 us of a 17-21 us launch, clock stamps).
// finish() normalises and leaves the 16-bit rows in `img` ([32][DL_YS bytes]); wave w owns rows RPW w .., lane 4 consecutive columns;
// the rows of a wave are normalised together (independent butterflies).  Every slab is 16-bit.
template <int NW, int NS> struct DlPro {
  // NS = slabs in flight (4: the head shares, 8: the FFN slices).  Lane (hw, l32) owns columns 8 l32 .. + 7 of row 2k + hw of its
  // wave's rows: 16-byte loads, two whole 512-byte slab rows per instruction -- the prologue's time went with the NUMBER of load
  // instructions (63 outstanding per wave x their size is all that hides the ~2 us a row written by the previous launch is away):
  // 8-byte pieces of one row per instruction took 20 k cycles for 8 slabs, clock stamps.
  static constexpr int RPW = DL_RB / NW, NP = RPW / 2, D = DL_D, PT = DL_RB * 32 / (NW * 64);
  int64_t rows[NP];
  float4 x0[NP], x1[NP];
  otr_u32x4 t[NS][NP];
  uint4 px[PT];
  float4 bb0, bb1, gm0, gm1, bt0, bt1;
  uint64_t seed;
  __device__ __forceinline__ void issue(const DlLn& p, int64_t row0, int nrows, int tid) {
    const int lane = tid & 63, wid = tid >> 6, hw = lane >> 5, col = (lane & 31) * 8;
    if (p.nslab == 0) {
#pragma unroll
      for (int k = 0; k < PT; ++k) {
        const int i = tid + k * NW * 64, r = i >> 5, ch = i & 31;
        px[k] = ld_global_b128(p.x16 + (row0 + min(r, nrows - 1)) * D + ch * 8);
      }
      return;
    }
    // EVERY load of the prologue goes out here, the small ones included: one issued later queues behind the caller's prefetches
    bb0 = bb1 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.bias) { bb0 = *reinterpret_cast<const float4*>(p.bias + col); bb1 = *reinterpret_cast<const float4*>(p.bias + col + 4); }
    gm0 = *reinterpret_cast<const float4*>(p.gamma + col); gm1 = *reinterpret_cast<const float4*>(p.gamma + col + 4);
    bt0 = *reinterpret_cast<const float4*>(p.beta + col); bt1 = *reinterpret_cast<const float4*>(p.beta + col + 4);
    seed = p.p_drop > 0.f ? *p.seed : 0;
#pragma unroll
    for (int k = 0; k < NP; ++k) {
      rows[k] = row0 + min(wid * RPW + 2 * k + hw, nrows - 1);
      x0[k] = *reinterpret_cast<const float4*>(p.xres + rows[k] * D + col);
      x1[k] = *reinterpret_cast<const float4*>(p.xres + rows[k] * D + col + 4);
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
  // put(r, ch, v): row r (0..31) of the 16-bit image, 16-byte chunk ch (0..31) <- v
  template <class PUT>
  __device__ __forceinline__ void finish(const DlLn& p, int nrows, bool write, PUT put, int tid) {
    const int lane = tid & 63, wid = tid >> 6, hw = lane >> 5, col = (lane & 31) * 8;
    if (p.nslab == 0) {
#pragma unroll
      for (int k = 0; k < PT; ++k) {
        const int i = tid + k * NW * 64, r = i >> 5, ch = i & 31;
        put(r, ch, px[k]);
      }
      return;
    }
    const bool drop = p.p_drop > 0.f;
    const uint32_t thr = drop ? (uint32_t)fminf(p.p_drop * 4294967296.f, 4294967295.f) : 0;
    const float inv_keep = drop ? 1.f / (1.f - p.p_drop) : 1.f;
    const float bb[8] = {bb0.x, bb0.y, bb0.z, bb0.w, bb1.x, bb1.y, bb1.z, bb1.w};
    const float g8[8] = {gm0.x, gm0.y, gm0.z, gm0.w, gm1.x, gm1.y, gm1.z, gm1.w};
    const float b8[8] = {bt0.x, bt0.y, bt0.z, bt0.w, bt1.x, bt1.y, bt1.z, bt1.w};
    float v[NP][8], sm[NP], qq[NP];
#pragma unroll
    for (int k = 0; k < NP; ++k) {
      float a[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) a[e] = bb[e];
#pragma unroll
      for (int s = 0; s < NS; ++s)
        if (s < p.nslab) {
          const uint32_t w[4] = {t[s][k].x, t[s][k].y, t[s][k].z, t[s][k].w};
#pragma unroll
          for (int e = 0; e < 4; ++e) { a[2 * e] += h2f_lo(w[e]); a[2 * e + 1] += h2f_hi(w[e]); }
        }
      for (int s = NS; s < p.nslab; ++s) {               // more slabs than fit in flight (not the shipped shapes): one after the other
        const uint4 q = ld_global_b128(reinterpret_cast<const uint16_t*>(p.slabs) + ((int64_t)s * p.R + rows[k]) * D + col);
        const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) { a[2 * e] += h2f_lo(w[e]); a[2 * e + 1] += h2f_hi(w[e]); }
      }
      const float xv[8] = {x0[k].x, x0[k].y, x0[k].z, x0[k].w, x1[k].x, x1[k].y, x1[k].z, x1[k].w};
      sm[k] = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float sc = 1.f;
        if (drop) sc = otr_rand32(seed, p.rng_offset + (uint64_t)(rows[k] * D + col + e)) >= thr ? inv_keep : 0.f;
        v[k][e] = xv[e] + a[e] * sc;
        sm[k] += v[k][e];
      }
    }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1)
#pragma unroll
      for (int k = 0; k < NP; ++k) sm[k] += __shfl_xor(sm[k], o);
#pragma unroll
    for (int k = 0; k < NP; ++k) {
      sm[k] *= (1.f / D);
      qq[k] = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float d = v[k][e] - sm[k]; qq[k] += d * d; }
    }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1)
#pragma unroll
      for (int k = 0; k < NP; ++k) qq[k] += __shfl_xor(qq[k], o);
#pragma unroll
    for (int k = 0; k < NP; ++k) {
      const int r = wid * RPW + 2 * k + hw;
      const int64_t row = rows[k];
      const float mean = sm[k], rstd = rsqrtf(qq[k] * (1.f / D) + p.eps);
      float o[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (v[k][e] - mean) * rstd * g8[e] + b8[e];
      const uint4 h = make_uint4(pack2h(o[0], o[1]), pack2h(o[2], o[3]), pack2h(o[4], o[5]), pack2h(o[6], o[7]));
      put(r, col >> 3, h);
      if (write && r < nrows) {
        if (p.z) {
          *reinterpret_cast<float4*>(p.z + row * D + col) = make_float4(v[k][0], v[k][1], v[k][2], v[k][3]);
          *reinterpret_cast<float4*>(p.z + row * D + col + 4) = make_float4(v[k][4], v[k][5], v[k][6], v[k][7]);
        }
        if (p.y) {
          *reinterpret_cast<float4*>(p.y + row * D + col) = make_float4(o[0], o[1], o[2], o[3]);
          *reinterpret_cast<float4*>(p.y + row * D + col + 4) = make_float4(o[4], o[5], o[6], o[7]);
        }
        if (p.y16) *reinterpret_cast<uint4*>(p.y16 + row * D + col) = h;
        if ((lane & 31) == 0) {
          if (p.mean) p.mean[row] = mean;
          if (p.rstd) p.rstd[row] = rstd;
        }
      }
    }
  }
};

