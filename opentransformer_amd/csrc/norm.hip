// y = LayerNorm(x + dropout(a)) forward/backward (post-norm residual blocks of
// encoder/transformer.py:54-56,61-63 and decoder/transformer.py:66-68,76-78,84-86).
// HBM-bound: one wave per row, float4 lanes, fp32 statistics via 64-lane shuffles; the dropout mask
// is a counter RNG regenerated in backward (never stored).
#include "common.h"

struct LnArgs {
  const float* x; const void* a; const float* gamma; const float* beta; const uint64_t* seed;
  float* y; float* z; float* mean; float* rstd; bf16_t* y_lp;
  const float* dy; float* dx; void* da; float* dgamma; float* dbeta; const float* zin; float* da_colsum; float* partial;
  int64_t M; int d;
  float eps, p_drop;
  uint64_t rng_offset;
  const float* skip;                                      // backward: dx = skip + LayerNorm input gradient (pre-norm residual)
  float a_scale;                                          // the branch enters as a_scale * dropout(a) (encoder/conformer.py:56: 0.5 * ffn)
  // a SECOND LayerNorm on the first one's output, y2 = LN2(LN1(z)) (encoder/conformer.py:87-89: post_ffn_norm, then final_norm), r05:
  // forward writes y2 (and its twin) instead of y1 plus mean2 / rstd2; backward takes d y2 and recomputes y1 from z / mean / rstd
  const float* gamma2; const float* beta2; float* mean2; float* rstd2;
  // ... and a THIRD on the second one's output, y3 = LN3(y2) (r06: final_norm of a Conformer block is followed by the NEXT block's
  // macaron_ffn_norm, encoder/conformer.py:50,89): forward writes y3's 16-bit twin (it only feeds a Linear) [+ y3] beside y2; backward takes
  // d y3 as well and adds its LayerNorm-3 input gradient to d y2 before the chain above runs
  const float* gamma3; const float* beta3; float* mean3; float* rstd3; float* y3; bf16_t* y3_lp; const void* dy3; int dy3_h16;   // d y3: f32, or the 16-bit type (its consumer was a 16-bit reader: the Linear's input gradient comes back in that type)
  int dy_h16;                                             // backward: dy is 16-bit (otr_ln_desc_t.dy_dtype)
  const uint8_t* amask;                                   // [M] or NULL: rows with 0 take no branch (a row := 0; da row := 0): module/conformer.py:109
};

constexpr int LN_MAXV = 4;  // float4 per lane -> d <= 1024

template <class AT> __device__ __forceinline__ void ld4(const AT* p, float* o) {
  if constexpr (sizeof(AT) == 4) {
    float4 v = *reinterpret_cast<const float4*>(p);
    o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
  } else {
    uint2 v = *reinterpret_cast<const uint2*>(p);
    o[0] = h2f_lo(v.x); o[1] = h2f_hi(v.x);
    o[2] = h2f_lo(v.y); o[3] = h2f_hi(v.y);
  }
}
template <class AT> __device__ __forceinline__ void st4(AT* p, const float* o) {
  if constexpr (sizeof(AT) == 4) *reinterpret_cast<float4*>(p) = make_float4(o[0], o[1], o[2], o[3]);
  else *reinterpret_cast<uint2*>(p) = make_uint2(pack2bf(o[0], o[1]), pack2bf(o[2], o[3]));
}

__device__ __forceinline__ float drop_scale(uint64_t seed, uint64_t idx, uint32_t thr, float inv_keep) {
  return otr_rand32(seed, idx) >= thr ? inv_keep : 0.f;
}

template <class AT, bool HAS_A, bool LN2 = false> __global__ __launch_bounds__(256) void add_ln_fwd_kernel(LnArgs p) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int64_t row = (int64_t)blockIdx.x * 4 + wid;
  if (row >= p.M) return;
  const int d = p.d;
  const bool drop = HAS_A && p.p_drop > 0.f;
  const uint64_t seed = drop ? *p.seed : 0;
  const uint32_t thr = drop ? (uint32_t)fminf(p.p_drop * 4294967296.f, 4294967295.f) : 0;
  const float inv_keep = (drop ? 1.f / (1.f - p.p_drop) : 1.f) * p.a_scale;
  float v[LN_MAXV][4];
  float s = 0.f;
  const float am = (HAS_A && p.amask) ? (p.amask[row] ? 1.f : 0.f) : 1.f;
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    int col = (i * 64 + lane) * 4;
    if (col < d) {
      ld4<float>(p.x + row * d + col, v[i]);
      if constexpr (HAS_A) {
        float a[4];
        ld4<AT>(reinterpret_cast<const AT*>(p.a) + row * d + col, a);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float sc = drop ? drop_scale(seed, p.rng_offset + (uint64_t)(row * d + col + e), thr, inv_keep) : inv_keep;
          v[i][e] += a[e] * sc * am;
        }
      }
      if (p.z) st4<float>(p.z + row * d + col, v[i]);
      s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
    }
  }
  const float mean = wave_sum(s) / d;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    int col = (i * 64 + lane) * 4;
    if (col < d) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { float t = v[i][e] - mean; q += t * t; }
    }
  }
  const float rstd = rsqrtf(wave_sum(q) / d + p.eps);
  if constexpr (LN2) {
    float s2 = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
      int col = (i * 64 + lane) * 4;
      if (col < d) {
        float g[4], bta[4];
        ld4<float>(p.gamma + col, g);
        ld4<float>(p.beta + col, bta);
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[i][e] = (v[i][e] - mean) * rstd * g[e] + bta[e]; s2 += v[i][e]; }   // y1, in place
      }
    }
    const float mean2 = wave_sum(s2) / d;
    float q2 = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
      int col = (i * 64 + lane) * 4;
      if (col < d) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { float t = v[i][e] - mean2; q2 += t * t; }
      }
    }
    const float rstd2 = rsqrtf(wave_sum(q2) / d + p.eps);
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
      int col = (i * 64 + lane) * 4;
      if (col < d) {
        float g[4], bta[4], o[4];
        ld4<float>(p.gamma2 + col, g);
        ld4<float>(p.beta2 + col, bta);
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (v[i][e] - mean2) * rstd2 * g[e] + bta[e];
        st4<float>(p.y + row * d + col, o);
        if (p.y_lp) st4<bf16_t>(p.y_lp + row * d + col, o);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[i][e] = o[e];                    // y2, kept for the third LayerNorm
      }
    }
    if (lane == 0) { p.mean[row] = mean; p.rstd[row] = rstd; p.mean2[row] = mean2; p.rstd2[row] = rstd2; }
    if (p.gamma3) {                                                    // (uniform) y3 = LN3(y2)
      float s3 = 0.f;
#pragma unroll
      for (int i = 0; i < LN_MAXV; ++i)
        if ((i * 64 + lane) * 4 < d) s3 += v[i][0] + v[i][1] + v[i][2] + v[i][3];
      const float mean3 = wave_sum(s3) / d;
      float q3 = 0.f;
#pragma unroll
      for (int i = 0; i < LN_MAXV; ++i)
        if ((i * 64 + lane) * 4 < d) {
#pragma unroll
          for (int e = 0; e < 4; ++e) { float t = v[i][e] - mean3; q3 += t * t; }
        }
      const float rstd3 = rsqrtf(wave_sum(q3) / d + p.eps);
#pragma unroll
      for (int i = 0; i < LN_MAXV; ++i) {
        int col = (i * 64 + lane) * 4;
        if (col < d) {
          float g[4], bta[4], o[4];
          ld4<float>(p.gamma3 + col, g);
          ld4<float>(p.beta3 + col, bta);
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = (v[i][e] - mean3) * rstd3 * g[e] + bta[e];
          if (p.y3) st4<float>(p.y3 + row * d + col, o);
          if (p.y3_lp) st4<bf16_t>(p.y3_lp + row * d + col, o);
        }
      }
      if (lane == 0) { p.mean3[row] = mean3; p.rstd3[row] = rstd3; }
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    int col = (i * 64 + lane) * 4;
    if (col < d) {
      float g[4], bta[4], o[4];
      ld4<float>(p.gamma + col, g);
      ld4<float>(p.beta + col, bta);
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (v[i][e] - mean) * rstd * g[e] + bta[e];
      if (p.y) st4<float>(p.y + row * d + col, o);
      if (p.y_lp) st4<bf16_t>(p.y_lp + row * d + col, o);
    }
  }
  if (lane == 0) { p.mean[row] = mean; p.rstd[row] = rstd; }
}

constexpr int LN_BWD_ROWS = 4;  // (rounds 1-5: rows per wave of a 4-wave workgroup; the workgroup's row count below is derived from it)

// NV = ceil(d / 256): float4 slots per lane actually used (d = 256 -> 1).  The loads of a row (dy, z, mean, rstd -- and d y3 in the
// three-LayerNorm form) are issued back to back before any arithmetic (rows clamped, tails masked).
// r06: a workgroup is 8 waves x ONE row (it was 4 waves x 4 rows).  The kernel is an HBM stream whose only latency hiding is the number of
// waves in flight, and the row sets live in registers: at 4 rows per wave the three-LayerNorm form held 256 registers -- one wave per
// SIMD, 38 us for 54 MB -- and the single form 142 (17 us for 48 MB).  2 rows: 190 / ~100 registers, Conformer step 9.82 -> 9.51 ms;
// 1 row: 9.46 ms, C2 (one launch of this kernel per step) unchanged (the per-workgroup partial sums of the affine gradients double in number: 996 rows at M = 7968).
template <class AT, bool HAS_A, int NV, bool LN2 = false> __global__ __launch_bounds__(512) void add_ln_bwd_kernel(LnArgs p) {
  constexpr int ROWS = 1, NWV = 8, NTH = 64 * NWV;
  __shared__ float red[2][NWV][NV * 256];  // [gamma|beta][wave][column]
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int d = p.d;
  const bool drop = HAS_A && p.p_drop > 0.f;
  const uint64_t seed = drop ? *p.seed : 0;
  const uint32_t thr = drop ? (uint32_t)fminf(p.p_drop * 4294967296.f, 4294967295.f) : 0;
  const float inv_keep = (drop ? 1.f / (1.f - p.p_drop) : 1.f) * p.a_scale;
  const bool want_ab = HAS_A && p.da_colsum != nullptr;
  float gam[NV][4], dg[NV][4], db[NV][4], dab[NV][4];
  int colv[NV];
  float cmask[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int col = (i * 64 + lane) * 4;
    cmask[i] = col < d ? 1.f : 0.f;
    colv[i] = min(col, d - 4);              // clamped: loads stay unconditional, contributions are masked
#pragma unroll
    for (int e = 0; e < 4; ++e) { dg[i][e] = 0.f; db[i][e] = 0.f; dab[i][e] = 0.f; }
    ld4<float>(p.gamma + colv[i], gam[i]);
  }
  // LN2: p.dy is d y2; gam2 / bet1 and the second LayerNorm's saved statistics turn it into d y1 below, its affine sums go to
  // partial[...][3d .. 5d) (dgamma2 | dbeta2)
  float gam2[NV][4], bet1[NV][4], dg2[NV][4], db2[NV][4];
  float gam3[NV][4], bet2[NV][4], dg3[NV][4], db3[NV][4];
  const bool three = LN2 && p.dy3 != nullptr;                      // (uniform) d y3 arrives too: LayerNorm-3 backward in front of the chain
  if constexpr (LN2) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      ld4<float>(p.gamma2 + colv[i], gam2[i]);
      ld4<float>(p.beta + colv[i], bet1[i]);
#pragma unroll
      for (int e = 0; e < 4; ++e) { dg2[i][e] = 0.f; db2[i][e] = 0.f; dg3[i][e] = 0.f; db3[i][e] = 0.f; gam3[i][e] = 0.f; bet2[i][e] = 0.f; }
      if (three) { ld4<float>(p.gamma3 + colv[i], gam3[i]); ld4<float>(p.beta2 + colv[i], bet2[i]); }
    }
  }
  const int64_t row0 = ((int64_t)blockIdx.x * NWV + wid) * ROWS;
  float dyv[ROWS][NV][4], zh[ROWS][NV][4], mean[ROWS], rstd[ROWS];
  float mean2[ROWS], rstd2[ROWS], mean3[ROWS], rstd3[ROWS];
  float dy3v[LN2 ? ROWS : 1][NV][4];
#pragma unroll
  for (int rr = 0; rr < ROWS; ++rr) {
    const int64_t row = min(row0 + rr, p.M - 1);
    mean[rr] = p.mean[row];
    rstd[rr] = p.rstd[row];
    if constexpr (LN2) {
      mean2[rr] = p.mean2[row]; rstd2[rr] = p.rstd2[row];
      mean3[rr] = three ? p.mean3[row] : 0.f; rstd3[rr] = three ? p.rstd3[row] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      if (p.dy_h16) ld4<bf16_t>(reinterpret_cast<const bf16_t*>(p.dy) + row * d + colv[i], dyv[rr][i]);
      else ld4<float>(p.dy + row * d + colv[i], dyv[rr][i]);
      ld4<float>(p.zin + row * d + colv[i], zh[rr][i]);
      if constexpr (LN2) {
        if (three) {
          if (p.dy3_h16) ld4<bf16_t>(reinterpret_cast<const bf16_t*>(p.dy3) + row * d + colv[i], dy3v[rr][i]);
          else ld4<float>(reinterpret_cast<const float*>(p.dy3) + row * d + colv[i], dy3v[rr][i]);
        }
      }
    }
  }
  if constexpr (LN2) {
    // d y2 -> d y1 through the second LayerNorm: y1 = zhat gamma + beta (recomputed), yhat = (y1 - mean2) rstd2
#pragma unroll
    for (int rr = 0; rr < ROWS; ++rr) {
      const float rmask = row0 + rr < p.M ? 1.f : 0.f;
      float yh[NV][4], t1 = 0.f, t2 = 0.f;
      if (three) {
        // d y3 -> its share of d y2 through the third LayerNorm: y2 = yhat2 gamma2 + beta2 (recomputed), y2hat = (y2 - mean3) rstd3
        float y2h[NV][4], u1 = 0.f, u2 = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
          const float w = rmask * cmask[i];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float y1 = (zh[rr][i][e] - mean[rr]) * rstd[rr] * gam[i][e] + bet1[i][e];
            const float y2 = (y1 - mean2[rr]) * rstd2[rr] * gam2[i][e] + bet2[i][e];
            y2h[i][e] = (y2 - mean3[rr]) * rstd3[rr];
            const float d3 = dy3v[rr][i][e] * w;
            const float g = d3 * gam3[i][e];
            u1 += g; u2 += g * y2h[i][e];
            dg3[i][e] += d3 * y2h[i][e];
            db3[i][e] += d3;
            dy3v[rr][i][e] = g;
          }
        }
        u1 = wave_sum(u1) / d;
        u2 = wave_sum(u2) / d;
#pragma unroll
        for (int i = 0; i < NV; ++i)
#pragma unroll
          for (int e = 0; e < 4; ++e) dyv[rr][i][e] += rstd3[rr] * (dy3v[rr][i][e] - u1 - y2h[i][e] * u2);     // d y2 (both shares)
      }
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const float w = rmask * cmask[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float y1 = (zh[rr][i][e] - mean[rr]) * rstd[rr] * gam[i][e] + bet1[i][e];
          yh[i][e] = (y1 - mean2[rr]) * rstd2[rr];
          const float dy2 = dyv[rr][i][e] * w;
          const float g = dy2 * gam2[i][e];
          t1 += g; t2 += g * yh[i][e];
          dg2[i][e] += dy2 * yh[i][e];
          db2[i][e] += dy2;
          dyv[rr][i][e] = g;                                 // d y2 . gamma2, finished below
        }
      }
      t1 = wave_sum(t1) / d;
      t2 = wave_sum(t2) / d;
#pragma unroll
      for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) dyv[rr][i][e] = rstd2[rr] * (dyv[rr][i][e] - t1 - yh[i][e] * t2);   // d y1
    }
  }
#pragma unroll
  for (int rr = 0; rr < ROWS; ++rr) {
    const int64_t row = row0 + rr;
    const float rmask = row < p.M ? 1.f : 0.f;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const float w = rmask * cmask[i];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        dyv[rr][i][e] *= w;
        zh[rr][i][e] = (zh[rr][i][e] - mean[rr]) * rstd[rr];
        float g = dyv[rr][i][e] * gam[i][e];
        s1 += g; s2 += g * zh[rr][i][e];
        dg[i][e] += dyv[rr][i][e] * zh[rr][i][e];
        db[i][e] += dyv[rr][i][e];
      }
    }
    s1 = wave_sum(s1) / d;
    s2 = wave_sum(s2) / d;
    if (row < p.M) {
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int col = (i * 64 + lane) * 4;
        if (col < d) {
          float dz[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) dz[e] = rstd[rr] * (dyv[rr][i][e] * gam[i][e] - s1 - zh[rr][i][e] * s2);
          if (p.skip) {                                   // the gradient that reaches z from the residual stream: part of dz for dx AND da
            float sk[4];
            ld4<float>(p.skip + row * d + col, sk);
#pragma unroll
            for (int e = 0; e < 4; ++e) dz[e] += sk[e];
          }
          st4<float>(p.dx + row * d + col, dz);
          if constexpr (HAS_A) {
            if (p.da) {
              const float am = p.amask ? (p.amask[row] ? 1.f : 0.f) : 1.f;
              float o[4];
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                float sc = drop ? drop_scale(seed, p.rng_offset + (uint64_t)(row * d + col + e), thr, inv_keep) : inv_keep;
                o[e] = dz[e] * sc * am;
                dab[i][e] += o[e];
              }
              st4<AT>(reinterpret_cast<AT*>(p.da) + row * d + col, o);
            }
          }
        }
      }
    }
  }
  // block reduction of the affine gradients, then one atomic per column per block
#pragma unroll
  for (int i = 0; i < NV; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      red[0][wid][(i * 64 + lane) * 4 + e] = dg[i][e];
      red[1][wid][(i * 64 + lane) * 4 + e] = db[i][e];
    }
  __syncthreads();
  // partial != NULL: this workgroup's sums go to partial[blockIdx.x][0|1|2][d] (dgamma | dbeta | da column sums) and the
  // caller column-sums the blocks (with everything else, in the grouped launch at the end of backward): no atomics --
  // they were 2.7 of this kernel's 14 us -- and a deterministic result
  auto rsum = [&](int which, int c) {
    float a = 0.f;
#pragma unroll
    for (int w = 0; w < NWV; ++w) a += red[which][w][c];
    return a;
  };
  float* prow = p.partial ? p.partial + (int64_t)blockIdx.x * (LN2 ? (three ? 7 : 5) : 3) * d : nullptr;
  for (int c = threadIdx.x; c < d; c += NTH) {
    float a = rsum(0, c);
    float b = rsum(1, c);
    if (prow) { prow[c] = a; prow[d + c] = b; }
    else { atomicAdd(p.dgamma + c, a); atomicAdd(p.dbeta + c, b); }
  }
  if constexpr (LN2) {                                     // dgamma2 | dbeta2 -> partial columns [3d, 5d) (partial is required here)
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        red[0][wid][(i * 64 + lane) * 4 + e] = dg2[i][e];
        red[1][wid][(i * 64 + lane) * 4 + e] = db2[i][e];
      }
    __syncthreads();
    for (int c = threadIdx.x; c < d; c += NTH) {
      prow[3 * d + c] = rsum(0, c);
      prow[4 * d + c] = rsum(1, c);
    }
    if (three) {                                           // dgamma3 | dbeta3 -> partial columns [5d, 7d)
      __syncthreads();
#pragma unroll
      for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          red[0][wid][(i * 64 + lane) * 4 + e] = dg3[i][e];
          red[1][wid][(i * 64 + lane) * 4 + e] = db3[i][e];
        }
      __syncthreads();
      for (int c = threadIdx.x; c < d; c += NTH) {
        prow[5 * d + c] = rsum(0, c);
        prow[6 * d + c] = rsum(1, c);
      }
    }
  }
  if constexpr (HAS_A) {
    // column sums of da = the bias gradient of the Linear that produced the branch (saves its colsum launch)
    if (want_ab || prow) {
      __syncthreads();
#pragma unroll
      for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) red[0][wid][(i * 64 + lane) * 4 + e] = dab[i][e];
      __syncthreads();
      for (int c = threadIdx.x; c < d; c += NTH) {
        const float v = rsum(0, c);
        if (prow) prow[2 * d + c] = v;
        else atomicAdd(p.da_colsum + c, v);
      }
    }
  }
}

static int32_t ln_check(const otr_ln_desc_t* d) {
  OTR_REQUIRE(d != nullptr, "layernorm: null descriptor");
  OTR_REQUIRE(d->M >= 0 && d->d > 0 && d->d % 4 == 0 && d->d <= LN_MAXV * 256,
              "layernorm: d=%d must be a multiple of 4 and <= %d", d->d, LN_MAXV * 256);
  OTR_REQUIRE(d->p_drop >= 0.f && d->p_drop < 1.f, "layernorm: p_drop=%f out of [0,1)", (double)d->p_drop);
  OTR_REQUIRE(d->a_dtype == OTR_F32 || d->a_dtype == OTR_H16, "layernorm: bad a_dtype");
  return 0;
}

extern "C" int32_t otr_add_layernorm_fwd(const otr_ln_desc_t* d, const float* x, const void* a, const float* gamma,
                                         const float* beta, const uint64_t* seed, float* y, void* y_bf16, float* z,
                                         float* mean, float* rstd, void* stream) {
  if (int32_t e = ln_check(d)) return e;
  OTR_REQUIRE(x && gamma && beta && (y || y_bf16) && mean && rstd, "add_layernorm_fwd: null pointer");
  OTR_REQUIRE(d->p_drop == 0.f || (a && seed), "add_layernorm_fwd: dropout needs a and seed");
  if (d->M == 0) return 0;
  LnArgs p{};
  p.x = x; p.a = a; p.gamma = gamma; p.beta = beta; p.seed = seed; p.y = y; p.z = z; p.mean = mean; p.rstd = rstd; p.y_lp = (bf16_t*)y_bf16;
  p.M = d->M; p.d = d->d; p.eps = d->eps; p.p_drop = d->p_drop; p.rng_offset = d->rng_offset;
  p.a_scale = d->a_scale == 0.f ? 1.f : d->a_scale; p.amask = d->a_row_mask;
  dim3 grid((unsigned)((d->M + 3) / 4));
  hipStream_t s = (hipStream_t)stream;
  if (!a) hipLaunchKernelGGL((add_ln_fwd_kernel<float, false>), grid, dim3(256), 0, s, p);
  else if (d->a_dtype == OTR_F32) hipLaunchKernelGGL((add_ln_fwd_kernel<float, true>), grid, dim3(256), 0, s, p);
  else hipLaunchKernelGGL((add_ln_fwd_kernel<bf16_t, true>), grid, dim3(256), 0, s, p);
  return otr_check_launch("add_layernorm_fwd");
}

extern "C" int32_t otr_add_layernorm_bwd(const otr_ln_desc_t* d, const float* dy, const float* z, const float* mean,
                                         const float* rstd, const float* gamma, const uint64_t* seed, float* dx,
                                         void* da, float* dgamma, float* dbeta, float* da_colsum, float* partial,
                                         void* stream) {
  return otr_add_layernorm_bwd_skip(d, dy, z, mean, rstd, gamma, seed, nullptr, dx, da, dgamma, dbeta, da_colsum, partial, stream);
}
extern "C" int32_t otr_add_layernorm_bwd_skip(const otr_ln_desc_t* d, const float* dy, const float* z, const float* mean,
                                              const float* rstd, const float* gamma, const uint64_t* seed, const float* skip,
                                              float* dx, void* da, float* dgamma, float* dbeta, float* da_colsum, float* partial,
                                              void* stream) {
  if (int32_t e = ln_check(d)) return e;
  OTR_REQUIRE(dy && z && mean && rstd && gamma && dx && (partial || (dgamma && dbeta)), "add_layernorm_bwd: null pointer");
  OTR_REQUIRE(d->p_drop == 0.f || seed, "add_layernorm_bwd: dropout needs seed");
  if (d->M == 0) return 0;
  LnArgs p{};
  p.dy = dy; p.zin = z; p.mean = const_cast<float*>(mean); p.rstd = const_cast<float*>(rstd); p.gamma = gamma;
  OTR_REQUIRE(d->dy_dtype == OTR_F32 || d->dy_dtype == OTR_H16, "add_layernorm_bwd: bad dy_dtype %d", d->dy_dtype);
  p.dy_h16 = d->dy_dtype == OTR_H16;
  p.seed = seed; p.skip = skip; p.dx = dx; p.da = da; p.dgamma = dgamma; p.dbeta = dbeta; p.da_colsum = da ? da_colsum : nullptr; p.partial = partial;
  p.M = d->M; p.d = d->d; p.eps = d->eps; p.p_drop = d->p_drop; p.rng_offset = d->rng_offset;
  p.a_scale = d->a_scale == 0.f ? 1.f : d->a_scale; p.amask = d->a_row_mask;
  dim3 grid((unsigned)((d->M + 2 * LN_BWD_ROWS - 1) / (2 * LN_BWD_ROWS)));
  hipStream_t s = (hipStream_t)stream;
#define LN_BWD_LAUNCH(NV)                                                                                   \
  {                                                                                                         \
    if (!da) hipLaunchKernelGGL((add_ln_bwd_kernel<float, false, NV>), grid, dim3(512), 0, s, p);           \
    else if (d->a_dtype == OTR_F32) hipLaunchKernelGGL((add_ln_bwd_kernel<float, true, NV>), grid, dim3(512), 0, s, p); \
    else hipLaunchKernelGGL((add_ln_bwd_kernel<bf16_t, true, NV>), grid, dim3(512), 0, s, p);               \
  }
  const int nv = (d->d + 255) / 256;
  if (nv == 1) LN_BWD_LAUNCH(1) else if (nv == 2) LN_BWD_LAUNCH(2) else if (nv == 3) LN_BWD_LAUNCH(3) else LN_BWD_LAUNCH(4)
#undef LN_BWD_LAUNCH
  return otr_check_launch("add_layernorm_bwd");
}

// y2 = LN2(LN1(x + a_scale dropout(a))): two LayerNorms back to back in one launch each way (encoder/conformer.py:87-89)
extern "C" int32_t otr_add_layernorm2_fwd(const otr_ln_desc_t* d, const float* x, const void* a, const float* gamma, const float* beta,
                                          const float* gamma2, const float* beta2, const uint64_t* seed, float* y2, void* y2_bf16,
                                          float* z, float* mean, float* rstd, float* mean2, float* rstd2, void* stream) {
  if (int32_t e = ln_check(d)) return e;
  OTR_REQUIRE(x && gamma && beta && gamma2 && beta2 && y2 && mean && rstd && mean2 && rstd2, "add_layernorm2_fwd: null pointer");
  OTR_REQUIRE(d->p_drop == 0.f || (a && seed), "add_layernorm2_fwd: dropout needs a and seed");
  if (d->M == 0) return 0;
  LnArgs p{};
  p.x = x; p.a = a; p.gamma = gamma; p.beta = beta; p.gamma2 = gamma2; p.beta2 = beta2; p.seed = seed; p.y = y2; p.z = z;
  p.mean = mean; p.rstd = rstd; p.mean2 = mean2; p.rstd2 = rstd2; p.y_lp = (bf16_t*)y2_bf16;
  p.M = d->M; p.d = d->d; p.eps = d->eps; p.p_drop = d->p_drop; p.rng_offset = d->rng_offset;
  p.a_scale = d->a_scale == 0.f ? 1.f : d->a_scale; p.amask = d->a_row_mask;
  dim3 grid((unsigned)((d->M + 3) / 4));
  hipStream_t s = (hipStream_t)stream;
  if (!a) hipLaunchKernelGGL((add_ln_fwd_kernel<float, false, true>), grid, dim3(256), 0, s, p);
  else if (d->a_dtype == OTR_F32) hipLaunchKernelGGL((add_ln_fwd_kernel<float, true, true>), grid, dim3(256), 0, s, p);
  else hipLaunchKernelGGL((add_ln_fwd_kernel<bf16_t, true, true>), grid, dim3(256), 0, s, p);
  return otr_check_launch("add_layernorm2_fwd");
}

// dy2 -> dx (= skip + d z), da; partial f32 [otr_add_layernorm_bwd_partial_rows(M)][5 d]: dgamma | dbeta | da column sums | dgamma2 | dbeta2
extern "C" int32_t otr_add_layernorm2_bwd(const otr_ln_desc_t* d, const float* dy2, const float* z, const float* mean, const float* rstd,
                                          const float* gamma, const float* beta, const float* mean2, const float* rstd2,
                                          const float* gamma2, const uint64_t* seed, const float* skip, float* dx, void* da,
                                          float* partial, void* stream) {
  if (int32_t e = ln_check(d)) return e;
  OTR_REQUIRE(dy2 && z && mean && rstd && gamma && beta && mean2 && rstd2 && gamma2 && dx && partial, "add_layernorm2_bwd: null pointer");
  OTR_REQUIRE(d->p_drop == 0.f || seed, "add_layernorm2_bwd: dropout needs seed");
  if (d->M == 0) return 0;
  LnArgs p{};
  p.dy = dy2; p.zin = z; p.mean = const_cast<float*>(mean); p.rstd = const_cast<float*>(rstd); p.gamma = gamma; p.beta = beta;
  p.mean2 = const_cast<float*>(mean2); p.rstd2 = const_cast<float*>(rstd2); p.gamma2 = gamma2;
  p.seed = seed; p.skip = skip; p.dx = dx; p.da = da; p.partial = partial;
  p.M = d->M; p.d = d->d; p.eps = d->eps; p.p_drop = d->p_drop; p.rng_offset = d->rng_offset;
  p.a_scale = d->a_scale == 0.f ? 1.f : d->a_scale; p.amask = d->a_row_mask;
  dim3 grid((unsigned)((d->M + 2 * LN_BWD_ROWS - 1) / (2 * LN_BWD_ROWS)));
  hipStream_t s = (hipStream_t)stream;
#define LN2_BWD_LAUNCH(NV)                                                                                        \
  {                                                                                                               \
    if (!da) hipLaunchKernelGGL((add_ln_bwd_kernel<float, false, NV, true>), grid, dim3(512), 0, s, p);           \
    else if (d->a_dtype == OTR_F32) hipLaunchKernelGGL((add_ln_bwd_kernel<float, true, NV, true>), grid, dim3(512), 0, s, p); \
    else hipLaunchKernelGGL((add_ln_bwd_kernel<bf16_t, true, NV, true>), grid, dim3(512), 0, s, p);               \
  }
  const int nv = (d->d + 255) / 256;
  if (nv == 1) LN2_BWD_LAUNCH(1) else if (nv == 2) LN2_BWD_LAUNCH(2) else if (nv == 3) LN2_BWD_LAUNCH(3) else LN2_BWD_LAUNCH(4)
#undef LN2_BWD_LAUNCH
  return otr_check_launch("add_layernorm2_bwd");
}

// y2 = LN2(LN1(x + a_scale dropout(a))) and y3 = LN3(y2): the two closing LayerNorms of a Conformer block and the macaron LayerNorm at the head
// of the NEXT block (encoder/conformer.py:87-89, :50) in one launch each way
extern "C" int32_t otr_add_layernorm3_fwd(const otr_ln_desc_t* d, const float* x, const void* a, const float* gamma, const float* beta,
                                          const float* gamma2, const float* beta2, const float* gamma3, const float* beta3, const uint64_t* seed,
                                          float* y2, float* y3, void* y3_bf16, float* z, float* mean, float* rstd, float* mean2, float* rstd2,
                                          float* mean3, float* rstd3, void* stream) {
  if (int32_t e = ln_check(d)) return e;
  OTR_REQUIRE(x && gamma && beta && gamma2 && beta2 && gamma3 && beta3 && y2 && (y3 || y3_bf16) && mean && rstd && mean2 && rstd2 && mean3 && rstd3,
              "add_layernorm3_fwd: null pointer");
  OTR_REQUIRE(d->p_drop == 0.f || (a && seed), "add_layernorm3_fwd: dropout needs a and seed");
  if (d->M == 0) return 0;
  LnArgs p{};
  p.x = x; p.a = a; p.gamma = gamma; p.beta = beta; p.gamma2 = gamma2; p.beta2 = beta2; p.gamma3 = gamma3; p.beta3 = beta3; p.seed = seed;
  p.y = y2; p.z = z; p.y3 = y3; p.y3_lp = (bf16_t*)y3_bf16;
  p.mean = mean; p.rstd = rstd; p.mean2 = mean2; p.rstd2 = rstd2; p.mean3 = mean3; p.rstd3 = rstd3;
  p.M = d->M; p.d = d->d; p.eps = d->eps; p.p_drop = d->p_drop; p.rng_offset = d->rng_offset;
  p.a_scale = d->a_scale == 0.f ? 1.f : d->a_scale; p.amask = d->a_row_mask;
  dim3 grid((unsigned)((d->M + 3) / 4));
  hipStream_t s = (hipStream_t)stream;
  if (!a) hipLaunchKernelGGL((add_ln_fwd_kernel<float, false, true>), grid, dim3(256), 0, s, p);
  else if (d->a_dtype == OTR_F32) hipLaunchKernelGGL((add_ln_fwd_kernel<float, true, true>), grid, dim3(256), 0, s, p);
  else hipLaunchKernelGGL((add_ln_fwd_kernel<bf16_t, true, true>), grid, dim3(256), 0, s, p);
  return otr_check_launch("add_layernorm3_fwd");
}

// d y2 (may be NULL: zeros) and d y3 -> dx (= skip + d z), da; partial f32 [otr_add_layernorm_bwd_partial_rows(M)][7 d]:
// dgamma | dbeta | da column sums | dgamma2 | dbeta2 | dgamma3 | dbeta3
extern "C" int32_t otr_add_layernorm3_bwd(const otr_ln_desc_t* d, const float* dy2, const void* dy3, int32_t dy3_dtype, const float* z, const float* mean,
                                          const float* rstd, const float* gamma, const float* beta, const float* mean2, const float* rstd2,
                                          const float* gamma2, const float* beta2, const float* mean3, const float* rstd3, const float* gamma3,
                                          const uint64_t* seed, const float* skip, float* dx, void* da, float* partial, void* stream) {
  if (int32_t e = ln_check(d)) return e;
  OTR_REQUIRE(dy2 && dy3 && z && mean && rstd && gamma && beta && mean2 && rstd2 && gamma2 && beta2 && mean3 && rstd3 && gamma3 && dx && partial,
              "add_layernorm3_bwd: null pointer");
  OTR_REQUIRE(d->p_drop == 0.f || seed, "add_layernorm3_bwd: dropout needs seed");
  if (d->M == 0) return 0;
  LnArgs p{};
  OTR_REQUIRE(dy3_dtype == OTR_F32 || dy3_dtype == OTR_H16, "add_layernorm3_bwd: bad dy3 dtype");
  p.dy = dy2; p.dy3 = dy3; p.dy3_h16 = dy3_dtype == OTR_H16; p.zin = z; p.mean = const_cast<float*>(mean); p.rstd = const_cast<float*>(rstd); p.gamma = gamma; p.beta = beta;
  p.mean2 = const_cast<float*>(mean2); p.rstd2 = const_cast<float*>(rstd2); p.gamma2 = gamma2; p.beta2 = beta2;
  p.mean3 = const_cast<float*>(mean3); p.rstd3 = const_cast<float*>(rstd3); p.gamma3 = gamma3;
  p.seed = seed; p.skip = skip; p.dx = dx; p.da = da; p.partial = partial;
  p.M = d->M; p.d = d->d; p.eps = d->eps; p.p_drop = d->p_drop; p.rng_offset = d->rng_offset;
  p.a_scale = d->a_scale == 0.f ? 1.f : d->a_scale; p.amask = d->a_row_mask;
  dim3 grid((unsigned)((d->M + 2 * LN_BWD_ROWS - 1) / (2 * LN_BWD_ROWS)));
  hipStream_t s = (hipStream_t)stream;
#define LN3_BWD_LAUNCH(NV)                                                                                        \
  {                                                                                                               \
    if (!da) hipLaunchKernelGGL((add_ln_bwd_kernel<float, false, NV, true>), grid, dim3(512), 0, s, p);           \
    else if (d->a_dtype == OTR_F32) hipLaunchKernelGGL((add_ln_bwd_kernel<float, true, NV, true>), grid, dim3(512), 0, s, p); \
    else hipLaunchKernelGGL((add_ln_bwd_kernel<bf16_t, true, NV, true>), grid, dim3(512), 0, s, p);               \
  }
  const int nv = (d->d + 255) / 256;
  if (nv == 1) LN3_BWD_LAUNCH(1) else if (nv == 2) LN3_BWD_LAUNCH(2) else if (nv == 3) LN3_BWD_LAUNCH(3) else LN3_BWD_LAUNCH(4)
#undef LN3_BWD_LAUNCH
  return otr_check_launch("add_layernorm3_bwd");
}

// rows of the `partial` buffer of otr_add_layernorm_bwd for M input rows
extern "C" int64_t otr_add_layernorm_bwd_partial_rows(int64_t M) { return (M + 2 * LN_BWD_ROWS - 1) / (2 * LN_BWD_ROWS); }
