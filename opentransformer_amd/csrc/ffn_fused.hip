// Row-block fused FFN sub-layer for the post-norm Transformer layers (encoder/transformer.py:58-63,
// decoder/transformer.py:82-86, module/ffn.py:38-41 with activation 'glu'):
//
//   forward :  y  = LayerNorm(x + dropout(w_2(glu(w_1 x + b_1)) + b_2))                      ONE launch
//   backward:  dh = GLU'(w_1 x + b_1) * (dy . w_2),  u = glu(w_1 x + b_1),  dx = skip + dh . w_1   ONE launch
//
// The 4096-wide FFN hidden never reaches HBM in the forward pass and is RECOMPUTED in the backward pass (SURVEY.md K7 +
// K8): forward traffic per layer is x (fp32 + 16-bit twin) in, y out; the backward pass writes only what the
// weight-gradient GEMMs consume (dh, u).
//
// Design (d_model = 256 makes every GEMM of this model skinny, so the classic big-tile GEMM is the wrong tool):
//  * a workgroup owns RB = 32 rows of the residual stream (249 workgroups for B=32 x 249 frames: one per CU);
//  * the hidden dimension is cut into chunks of 32 units; the 4 waves of a workgroup own DIFFERENT chunks and never
//    synchronise inside the chunk loop: each wave computes h^T[32 hidden, 32 rows] with v_mfma_f32_32x32x16 (A = weight
//    fragment, B = activation fragment), applies the GLU on the accumulator registers, and feeds the result straight
//    back as the B operand of the next GEMM -- the accumulator layout (lane = row m, registers = hidden units) IS an
//    operand layout once the contraction index is permuted, and the permutation is applied to the WEIGHTS when they
//    are packed (otr_pack_frags, perm = 1);
//  * weights are pre-packed "fragment-major" (1 KiB = one MFMA A operand = 64 lanes x 16 B, in consumption order), so
//    the weight stream is L2 -> VGPR with fully coalesced 16-byte loads through a PD-deep register ring: no LDS, no
//    barrier in the main loop.  Each weight fragment is used exactly once per wave, so LDS staging would buy nothing;
//  * the waves' partial outputs (sums over their chunks) meet in LDS once, at the end, where the bias / dropout /
//    residual / LayerNorm epilogue (forward) or the skip-connection add (backward) runs on whole rows.
// Bound: the 3 MB (forward) / 5 MB (backward) of packed weights stream from L2 into every CU: 64 B/clk/CU ->
// ~20 us / ~33 us per layer at B=32, i.e. ~50 % of the MFMA rate; HBM traffic is a few MB.
#include "ffn_frag.h"

// ------------------------------------------------------------------------------------------------ weight packing
// Fragment (rt, ks) of a logical matrix A[r][c] (r = free index, c = contraction index), element (r, c) at
// src[off + r*rs + c*cs]:  1 KiB at dst[dst_off + (rt*(cols/16) + ks)*512 + lane*8 + j]  holding
//   A[rt*32 + (lane&31)][ks*16 + kmap(lane>>5, j)],   kmap(hi, j) = hi*8 + j                       (perm = 0)
//                                                      kmap(hi, j) = 4*hi + j (j<4), 8 + 4*hi + j-4  (perm = 1)
// perm = 1 matches an accumulator tile that is fed back as the B operand (its lane holds contraction indices
// {4hi..4hi+3, 8+4hi..8+4hi+3} of each group of 16).  table (device, int64 [n][8]):
//   {src_off, rs, cs, rows, cols, perm, dst_off, first_block}; a block packs 4 fragments.
__global__ __launch_bounds__(256) void pack_frags_kernel(const uint16_t* __restrict__ src, uint16_t* __restrict__ dst,
                                                        const int64_t* __restrict__ table, int n) {
  const int64_t b = blockIdx.x;
  int lo = 0, hi_ = n - 1;
  while (lo < hi_) {
    const int mid = (lo + hi_ + 1) >> 1;
    if (table[mid * 8 + 7] <= b) lo = mid; else hi_ = mid - 1;
  }
  const int64_t* t = table + lo * 8;
  const int64_t src_off = t[0], rs = t[1], cs = t[2], rows = t[3], cols = t[4], perm = t[5], dst_off = t[6];
  const int64_t nks = cols >> 4, nfrag = (rows >> 5) * nks;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int64_t frag = (b - t[7]) * 4 + wid;
  if (frag >= nfrag) return;
  const int64_t rt = frag / nks, ks = frag - rt * nks;
  const int64_t r = rt * 32 + (lane & 31);
  const int hi = lane >> 5;
  const uint16_t* s = src + src_off + r * rs;
  uint4 q;
  if (cs == 1 && perm == 0 && ((src_off + r * rs) % 8 == 0)) {        // contraction index contiguous: one 16-byte load
    q = *reinterpret_cast<const uint4*>(s + ks * 16 + hi * 8);
  } else if (cs == 1 && ((src_off + r * rs) % 4 == 0)) {                // perm = 1 on a contiguous source: two 8-byte loads
    const uint2 a = *reinterpret_cast<const uint2*>(s + ks * 16 + 4 * hi);
    const uint2 b = *reinterpret_cast<const uint2*>(s + ks * 16 + 8 + 4 * hi);
    q = make_uint4(a.x, a.y, b.x, b.y);
  } else {
    uint16_t v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int kk = perm ? (j < 4 ? 4 * hi + j : 8 + 4 * hi + (j - 4)) : hi * 8 + j;
      v[j] = s[(ks * 16 + kk) * cs];
    }
    q.x = v[0] | ((uint32_t)v[1] << 16); q.y = v[2] | ((uint32_t)v[3] << 16);
    q.z = v[4] | ((uint32_t)v[5] << 16); q.w = v[6] | ((uint32_t)v[7] << 16);
  }
  *reinterpret_cast<uint4*>(dst + dst_off + frag * 512 + lane * 8) = q;
}

extern "C" int32_t otr_pack_frags(const void* src, void* dst, const int64_t* table, int32_t n_items, int64_t total_blocks,
                                  void* stream) {
  OTR_REQUIRE(src && dst && table, "pack_frags: null pointer");
  OTR_REQUIRE(n_items >= 0 && total_blocks >= 0 && total_blocks < (1ll << 31), "pack_frags: bad sizes");
  if (n_items == 0 || total_blocks == 0) return 0;
  hipLaunchKernelGGL(pack_frags_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream,
                     (const uint16_t*)src, (uint16_t*)dst, table, n_items);
  return otr_check_launch("pack_frags");
}

// ------------------------------------------------------------------------------------------------ forward
struct FfnFwdArgs {
  const float* x;          // residual stream [M, D] f32
  const uint16_t* x16;     // its 16-bit twin [M, D] (the GEMM operand form)
  const uint4* p1;         // w_1 packed: rows = 2F hidden pre-activations (value rows then gate rows), contraction = D, perm 0
  const float* b1;         // [2F]
  const uint4* p2;         // w_2 packed: rows = D outputs, contraction = F hidden units, perm 1
  const float* b2;         // [D]
  const float* gamma; const float* beta; const uint64_t* seed;
  float* y; uint16_t* y16; float* z; float* mean; float* rstd;
  int M, F;
  float eps, p_drop;
  uint64_t rng_offset;
};

// bias + dropout + residual + LayerNorm on whole rows (shared by the forward kernels): red[slot][row][YP] holds the four
// waves' partial outputs
template <int D, int RPW, int YP>
__device__ __forceinline__ void ffn_fwd_row_epilogue(const FfnFwdArgs& p, const float* red, int row0, int wid, int lane) {
  // wave w owns rows RPW*w .. +RPW-1, lane owns columns 4*lane..+3
  const bool drop = p.p_drop > 0.f;
  const uint64_t seed = drop ? *p.seed : 0;
  const uint32_t thr = drop ? (uint32_t)fminf(p.p_drop * 4294967296.f, 4294967295.f) : 0;
  const float inv_keep = drop ? 1.f / (1.f - p.p_drop) : 1.f;
  const int col = lane * 4;
  const float4 b2 = *reinterpret_cast<const float4*>(p.b2 + col);
  const float4 gm = *reinterpret_cast<const float4*>(p.gamma + col);
  const float4 bt = *reinterpret_cast<const float4*>(p.beta + col);
  float4 xr[RPW];
#pragma unroll
  for (int i = 0; i < RPW; ++i) {
    const int64_t row = min((int64_t)row0 + wid * RPW + i, (int64_t)p.M - 1);
    xr[i] = *reinterpret_cast<const float4*>(p.x + row * D + col);
  }
  // the wave's rows are normalised TOGETHER: their butterfly steps are independent, so the cross-lane latency is paid
  // 12 times per wave instead of 12 x RPW (rowblock.hip: one row after the other it was ~5 us of a launch)
  float v[RPW][4], sm[RPW], qq[RPW];
#pragma unroll
  for (int i = 0; i < RPW; ++i) {
    const int r = wid * RPW + i;
    const int64_t row = min((int64_t)row0 + r, (int64_t)p.M - 1);
    float t4[4] = {b2.x, b2.y, b2.z, b2.w};
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const float4 t = *reinterpret_cast<const float4*>(red + (w * FF_RB + r) * YP + col);
      t4[0] += t.x; t4[1] += t.y; t4[2] += t.z; t4[3] += t.w;
    }
    const float xv[4] = {xr[i].x, xr[i].y, xr[i].z, xr[i].w};
    sm[i] = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float sc = 1.f;
      if (drop) sc = otr_rand32(seed, p.rng_offset + (uint64_t)(row * D + col + e)) >= thr ? inv_keep : 0.f;
      v[i][e] = xv[e] + t4[e] * sc;
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
    for (int e = 0; e < 4; ++e) { const float t = v[i][e] - sm[i]; qq[i] += t * t; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1)
#pragma unroll
    for (int i = 0; i < RPW; ++i) qq[i] += __shfl_xor(qq[i], o);
  const float g4[4] = {gm.x, gm.y, gm.z, gm.w}, b4[4] = {bt.x, bt.y, bt.z, bt.w};
#pragma unroll
  for (int i = 0; i < RPW; ++i) {
    const int64_t row = (int64_t)row0 + wid * RPW + i;
    if (row >= p.M) break;                                  // wave-uniform
    const float mean = sm[i], rstd = rsqrtf(qq[i] * (1.f / D) + p.eps);
    if (p.z) *reinterpret_cast<float4*>(p.z + row * D + col) = make_float4(v[i][0], v[i][1], v[i][2], v[i][3]);
    float o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = (v[i][e] - mean) * rstd * g4[e] + b4[e];
    *reinterpret_cast<float4*>(p.y + row * D + col) = make_float4(o[0], o[1], o[2], o[3]);
    if (p.y16) *reinterpret_cast<uint2*>(p.y16 + row * D + col) = make_uint2(pack2h(o[0], o[1]), pack2h(o[2], o[3]));
    if (lane == 0) { p.mean[row] = mean; p.rstd[row] = rstd; }
  }
}

// NW = waves per workgroup.  A pure streaming kernel ingests ~22 B/clk per CU with 4 waves and ~33 B/clk with 8
// (tools/ubench/l2stream.hip), and this kernel sits on the 4-wave figure -- but its 8-wave form (two waves per SIMD, <= 256
// registers each, a 12-deep ring per wave) measured 85 us against 61 us: the shallower rings and the shared matrix pipe cost
// more than the extra load issue slots bring.  NW = 4 is the production form; NW = 8 stays selectable (otr_debug_set(5, 8)).
template <int D, int NW, int PD>
__global__ __launch_bounds__(NW * 64, NW / 4) void ffn_ln_fwd_kernel(FfnFwdArgs p) {
  static_assert(D == 256, "the LayerNorm epilogue maps one float4 per lane: d_model = 256");
  constexpr int NKS = D / 16, NT = D / 32, YP = D + 4;
  constexpr int STEPS = 2 * NKS + 2 * NT;          // weight fragments (= MFMAs) per chunk: 32 + 16
  constexpr int NTHR = NW * 64, RPW = FF_RB / NW;  // PD = fragments in flight per wave (1 KiB each)
  static_assert(NW == 4 || NW == 8, "4 or 8 waves");
  static_assert(STEPS % PD == 0, "ring slots must be compile-time constants");
  __shared__ __attribute__((aligned(16))) unsigned char smem[FF_RB * D * 2 + 4 * FF_RB * YP * 4];
  uint4* xs = reinterpret_cast<uint4*>(smem);
  float* red = reinterpret_cast<float*>(smem + FF_RB * D * 2);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 31, hi = lane >> 5;
  const int row0 = blockIdx.x * FF_RB;
  stage_rows<D, NTHR>(xs, p.x16, row0, p.M, tid);
  __syncthreads();

  const int nchunk = p.F / 32, npair = nchunk / (2 * NW), nit = nchunk / NW;
  const int rot = (int)(blockIdx.x % (unsigned)npair);      // workgroups walk the weights from different starting points
  auto chunk_of = [&](int it) {
    int j = (it >> 1) + rot;
    if (j >= npair) j -= npair;
    return 2 * NW * j + 2 * wid + (it & 1);                  // a wave's consecutive chunks are adjacent
  };
  const uint4* P1 = p.p1 + lane;
  const uint4* P2 = p.p2 + lane;
  auto fptr = [&](int c, int s) -> const uint4* {
    if (s < 2 * NKS) return P1 + (int64_t)(((s & 1) ? nchunk + c : c) * NKS + (s >> 1)) * 64;
    const int t = s - 2 * NKS;
    return P2 + (int64_t)((t >> 1) * (2 * nchunk) + 2 * c + (t & 1)) * 64;
  };

  f32x16 yacc[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) yacc[i][r] = 0.f;

  uint4 ring[PD];
  int c = chunk_of(0);
#pragma unroll
  for (int s = 0; s < PD; ++s) ring[s] = ld_global_b128(fptr(c, s));

  for (int it = 0; it < nit; ++it) {
    const int cn = chunk_of(min(it + 1, nit - 1));          // last round: re-loads its own chunk (valid, unused)
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
        if ((s & 1) == 0) xb = frag_b<D>(xs, m, hi, s >> 1);
        if (s & 1) mma32(ag, w, xb); else mma32(av, w, xb);
      } else {
        const int t = s - 2 * NKS;
        mma32(yacc[t >> 1], w, (t & 1) ? uf1 : uf0);
      }
      ring[s % PD] = ld_global_b128(s + PD < STEPS ? fptr(c, s + PD) : fptr(cn, s + PD - STEPS));
      if (s == 2 * NKS - 1) {                               // GLU on the accumulators: u = (a + b_a) * sigmoid(g + b_g)
        float u[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float a = av[r] + reinterpret_cast<const float*>(&bv[r >> 2])[r & 3];
          const float g = ag[r] + reinterpret_cast<const float*>(&bg[r >> 2])[r & 3];
          u[r] = a * fast_sigmoid(g);
        }
        tile_to_frags(u, uf0, uf1);
      }
    }
    c = cn;
  }

  // the waves' partial y^T tiles meet in LDS: red[slot][m][n], n = nt*32 + 8q + 4hi + (r&3); four slots -- with 8 waves
  // the upper four first hand their tiles to the lower four, which fold them into their accumulators
  auto put = [&](int slot) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *reinterpret_cast<float4*>(red + (slot * FF_RB + m) * YP + nt * 32 + 8 * q + 4 * hi) =
            make_float4(yacc[nt][4 * q], yacc[nt][4 * q + 1], yacc[nt][4 * q + 2], yacc[nt][4 * q + 3]);
  };
  if constexpr (NW == 8) {
    if (wid >= 4) put(wid - 4);
    __syncthreads();
    if (wid < 4) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 t = *reinterpret_cast<const float4*>(red + (wid * FF_RB + m) * YP + nt * 32 + 8 * q + 4 * hi);
          yacc[nt][4 * q] += t.x; yacc[nt][4 * q + 1] += t.y; yacc[nt][4 * q + 2] += t.z; yacc[nt][4 * q + 3] += t.w;
        }
    }
    __syncthreads();
  }
  if (wid < 4) put(wid);
  __syncthreads();

  ffn_fwd_row_epilogue<D, RPW, YP>(p, red, row0, wid, lane);
}

// ------------------------------------------------------------------------------------------------ forward, weights through LDS-DMA
// Same structure as ffn_ln_fwd_kernel (32 rows per workgroup, a wave owns hidden chunks, no barrier in the loop), but the
// weight fragments travel global -> LDS by direct-to-LDS DMA into a wave-PRIVATE ring (PD x 1 KiB) and reach the MFMA by one
// ds_read_b128: the DMA path ingests ~33 B/clk per CU where global -> VGPR loads stop at ~22 (the v2 kernels
// measured 10.7 us for 0.75 MB).  The fragment address is wave-uniform (SGPR base + lane * 16), so a DMA costs scalar
// instructions only; the wave waits for its own DMAs with counted vmcnt (nothing else in the loop is a vector-memory
// operation: the biases are staged in LDS up front).  The rings live where the partial outputs meet after the loop.
template <int N> __device__ __forceinline__ void ffn_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int D, int PD>
__global__ __launch_bounds__(256, 1) void ffn_ln_fwd_dma_kernel(FfnFwdArgs p) {
  static_assert(D == 256, "the LayerNorm epilogue maps one float4 per lane: d_model = 256");
  constexpr int NKS = D / 16, NT = D / 32, YP = D + 4, NW = 4;
  constexpr int STEPS = 2 * NKS + 2 * NT;          // weight fragments (= MFMAs) per chunk: 32 + 16
  constexpr int RPW = FF_RB / NW;
  static_assert(STEPS % PD == 0, "ring slots must be compile-time constants");
  static_assert(NW * PD * 1024 + 16384 <= 4 * FF_RB * YP * 4, "rings + staged biases fit where the partial outputs meet");
  __shared__ __attribute__((aligned(1024))) unsigned char smem[4 * FF_RB * YP * 4 + FF_RB * D * 2];
  float* red = reinterpret_cast<float*>(smem);                       // after the loop; before it: rings | biases
  uint4* xs = reinterpret_cast<uint4*>(smem + 4 * FF_RB * YP * 4);
  float* bias_s = reinterpret_cast<float*>(smem + NW * PD * 1024);   // b_1 [2F] (F <= 2048)

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 31, hi = lane >> 5;
  const int row0 = blockIdx.x * FF_RB;
  stage_rows<D, 256>(xs, p.x16, row0, p.M, tid);
  for (int i = tid * 4; i < 2 * p.F; i += 1024) *reinterpret_cast<float4*>(bias_s + i) = *reinterpret_cast<const float4*>(p.b1 + i);
  __syncthreads();

  const int nchunk = p.F / 32, npair = nchunk / (2 * NW), nit = nchunk / NW;
  const int rot = (int)(blockIdx.x % (unsigned)npair);      // workgroups walk the weights from different starting points
  auto chunk_of = [&](int it) {
    int j = (it >> 1) + rot;
    if (j >= npair) j -= npair;
    return 2 * NW * j + 2 * wid + (it & 1);                  // a wave's consecutive chunks are adjacent
  };
  const unsigned char* P1 = reinterpret_cast<const unsigned char*>(p.p1);
  const unsigned char* P2 = reinterpret_cast<const unsigned char*>(p.p2);
  auto fptr = [&](int c, int s) -> const unsigned char* {   // wave-uniform address of fragment s of chunk c
    if (s < 2 * NKS) return P1 + ((int64_t)(((s & 1) ? nchunk + c : c) * NKS + (s >> 1)) << 10);
    const int t = s - 2 * NKS;
    return P2 + ((int64_t)((t >> 1) * (2 * nchunk) + 2 * c + (t & 1)) << 10);
  };
  const uint32_t ring0 = (uint32_t)(uintptr_t)(ffn_lds_byte*)smem + (uint32_t)wid * (PD * 1024);
  const uint32_t lane_off = (uint32_t)lane * 16u;
  const uint4* ring = reinterpret_cast<const uint4*>(smem + wid * (PD * 1024)) + lane;

  f32x16 yacc[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) yacc[i][r] = 0.f;

  int c = chunk_of(0);
#pragma unroll
  for (int s = 0; s < PD; ++s) ffn_dma(fptr(c, s), lane_off, ring0 + s * 1024);
  ffn_wait_vm<PD - 1>();
  uint4 w = ring[0];

  for (int it = 0; it < nit; ++it) {
    const int cn = chunk_of(min(it + 1, nit - 1));          // last round: re-loads its own chunk (valid, unused)
    float4 bv[4], bg[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      bv[q] = *reinterpret_cast<const float4*>(bias_s + c * 32 + 8 * q + 4 * hi);
      bg[q] = *reinterpret_cast<const float4*>(bias_s + p.F + c * 32 + 8 * q + 4 * hi);
    }
    f32x16 av, ag;
#pragma unroll
    for (int r = 0; r < 16; ++r) { av[r] = 0.f; ag[r] = 0.f; }
    uint4 xb, uf0, uf1;
#pragma clang loop unroll(full)
    for (int s = 0; s < STEPS; ++s) {
      // fragment s is in `w`; fragment s+1 has landed once at most PD-2 DMAs are outstanding (PD were in flight)
      ffn_wait_vm<PD - 2>();
      const uint4 wn = ring[((s + 1) % PD) * 64];
      if (s < 2 * NKS) {
        if ((s & 1) == 0) xb = frag_b<D>(xs, m, hi, s >> 1);
        if (s & 1) mma32(ag, w, xb); else mma32(av, w, xb);
      } else {
        const int t = s - 2 * NKS;
        mma32(yacc[t >> 1], w, (t & 1) ? uf1 : uf0);
      }
      // slot s % PD is free (its fragment sits in `w`, read one step ago): fetch the fragment PD steps ahead into it
      ffn_dma(s + PD < STEPS ? fptr(c, s + PD) : fptr(cn, s + PD - STEPS), lane_off, ring0 + (s % PD) * 1024);
      w = wn;
      if (s == 2 * NKS - 1) {                               // GLU on the accumulators: u = (a + b_a) * sigmoid(g + b_g)
        float u[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float a = av[r] + reinterpret_cast<const float*>(&bv[r >> 2])[r & 3];
          const float g = ag[r] + reinterpret_cast<const float*>(&bg[r >> 2])[r & 3];
          u[r] = a * fast_sigmoid(g);
        }
        tile_to_frags(u, uf0, uf1);
      }
    }
    c = cn;
  }
  ffn_wait_vm<0>();                                          // the trailing (unused) DMAs must land before the rings become `red`
  __syncthreads();

  auto put = [&](int slot) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *reinterpret_cast<float4*>(red + (slot * FF_RB + m) * YP + nt * 32 + 8 * q + 4 * hi) =
            make_float4(yacc[nt][4 * q], yacc[nt][4 * q + 1], yacc[nt][4 * q + 2], yacc[nt][4 * q + 3]);
  };
  put(wid);
  __syncthreads();
  ffn_fwd_row_epilogue<D, RPW, YP>(p, red, row0, wid, lane);
}

// ------------------------------------------------------------------------------------------------ backward
struct FfnBwdArgs {
  const uint16_t* x16;     // FFN input [M, D] (16-bit twin of the residual stream)
  const uint16_t* dy16;    // gradient of the FFN output [M, D] (branch gradient of the LayerNorm backward)
  const uint4* p1;         // w_1 packed as in the forward pass (recompute of the pre-activations)
  const float* b1;
  const uint4* p3;         // w_2^T packed: rows = F hidden units, contraction = D, perm 0       (du = dy . w_2)
  const uint4* p4;         // w_1^T packed: rows = D, contraction = 2F (value then gate), perm 1  (dx = dh . w_1)
  uint16_t* dh;            // [M, 2F] out: gradient of the pre-activations (operand of the w_1 weight gradient)
  uint16_t* u;             // [M, F] out: glu output (operand of the w_2 weight gradient)
  float* bpart;            // [gridDim.x][2F] out: column sums of dh over this workgroup's rows (the w_1 bias gradient, partial)
  const float* skip;       // [M, D] f32 or NULL, added to dx (the skip-connection gradient of y = LN(x + f(x)))
  float* dx;               // [M, D] f32 out (may alias skip)
  int M, F;
  int ablate;              // tuning hook (otr_debug_set(4, v)): bit 2 = no bias-gradient reduction; 0 in production
};

template <int D>
__global__ __launch_bounds__(256, 1) void ffn_bwd_kernel(FfnBwdArgs p) {
  static_assert(D == 256, "row epilogue maps one float4 per lane: d_model = 256");
  constexpr int NKS = D / 16, NT = D / 32, YP = D + 4;
  constexpr int S1 = 2 * NKS, S2 = S1 + NKS, STEPS = S2 + 4 * NT;   // 32 + 16 + 32 weight fragments per chunk
  constexpr int PD = 20;
  static_assert(STEPS % PD == 0, "ring slots must be compile-time constants");
  __shared__ __attribute__((aligned(16))) unsigned char smem[4 * FF_RB * YP * 4];   // operands first, partial sums after
  uint4* xs = reinterpret_cast<uint4*>(smem);
  uint4* ds = reinterpret_cast<uint4*>(smem + FF_RB * D * 2);
  float* red = reinterpret_cast<float*>(smem);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 31, hi = lane >> 5;
  const int row0 = blockIdx.x * FF_RB;
  stage_rows<D>(xs, p.x16, row0, p.M, tid);
  stage_rows<D>(ds, p.dy16, row0, p.M, tid);
  __syncthreads();

  const int nchunk = p.F / 32, npair = nchunk / 8, nit = nchunk / 4;
  const int rot = (int)(blockIdx.x % (unsigned)npair);
  auto chunk_of = [&](int it) {
    int j = (it >> 1) + rot;
    if (j >= npair) j -= npair;
    return 8 * j + 2 * wid + (it & 1);
  };
  const uint4* P1 = p.p1 + lane;
  const uint4* P3 = p.p3 + lane;
  const uint4* P4 = p.p4 + lane;
  auto fptr = [&](int c, int s) -> const uint4* {
    if (s < S1) return P1 + (int64_t)(((s & 1) ? nchunk + c : c) * NKS + (s >> 1)) * 64;
    if (s < S2) return P3 + (int64_t)(c * NKS + (s - S1)) * 64;
    const int t = s - S2, j4 = t & 3;
    const int ksf = (j4 < 2) ? 2 * c + j4 : 2 * nchunk + 2 * c + (j4 - 2);
    return P4 + (int64_t)((t >> 2) * (4 * nchunk) + ksf) * 64;
  };

  f32x16 xacc[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) xacc[i][r] = 0.f;

  uint4 ring[PD];
  int c = chunk_of(0);
#pragma unroll
  for (int s = 0; s < PD; ++s) ring[s] = ld_global_b128(fptr(c, s));
  const int64_t grow = (int64_t)row0 + m;
  const bool live = grow < p.M;
  const int64_t crow = min(grow, (int64_t)p.M - 1);

  for (int it = 0; it < nit; ++it) {
    const int cn = chunk_of(min(it + 1, nit - 1));
    float4 bv[4], bg[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      bv[q] = *reinterpret_cast<const float4*>(p.b1 + c * 32 + 8 * q + 4 * hi);
      bg[q] = *reinterpret_cast<const float4*>(p.b1 + p.F + c * 32 + 8 * q + 4 * hi);
    }
    f32x16 av, ag, du;
#pragma unroll
    for (int r = 0; r < 16; ++r) { av[r] = 0.f; ag[r] = 0.f; du[r] = 0.f; }
    uint4 ob, hf[4];
#pragma clang loop unroll(full)
    for (int s = 0; s < STEPS; ++s) {
      const uint4 w = ring[s % PD];
      if (s < S1) {
        if ((s & 1) == 0) ob = frag_b<D>(xs, m, hi, s >> 1);
        if (s & 1) mma32(ag, w, ob); else mma32(av, w, ob);
      } else if (s < S2) {
        ob = frag_b<D>(ds, m, hi, s - S1);
        mma32(du, w, ob);
      } else {
        const int t = s - S2;
        mma32(xacc[t >> 2], w, hf[t & 3]);
      }
      ring[s % PD] = ld_global_b128(s + PD < STEPS ? fptr(c, s + PD) : fptr(cn, s + PD - STEPS));
      if (s == S2 - 1) {
        // GLU backward on the accumulators: a = value, sg = sigmoid(gate);  u = a*sg;  d a = du*sg;  d gate = du*a*sg*(1-sg)
        float uu[16], da_[16], dg_[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float a = av[r] + reinterpret_cast<const float*>(&bv[r >> 2])[r & 3];
          const float sg = fast_sigmoid(ag[r] + reinterpret_cast<const float*>(&bg[r >> 2])[r & 3]);
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
        if (!(p.ablate & 4)) {
          float* bp = p.bpart + (int64_t)blockIdx.x * (2 * p.F) + c * 32;    // this wave owns chunk c of the block's row
          tile_colsum_store(da_, bp, lane, hi, live);
          tile_colsum_store(dg_, bp + p.F, lane, hi, live);
        }
      }
    }
    c = cn;
  }

  __syncthreads();                                          // every wave is done with the staged operands
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int q = 0; q < 4; ++q)
      *reinterpret_cast<float4*>(red + (wid * FF_RB + m) * YP + nt * 32 + 8 * q + 4 * hi) =
          make_float4(xacc[nt][4 * q], xacc[nt][4 * q + 1], xacc[nt][4 * q + 2], xacc[nt][4 * q + 3]);
  __syncthreads();
  const int col = lane * 4;
  float4 sk[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int64_t row = min((int64_t)row0 + wid * 8 + i, (int64_t)p.M - 1);
    sk[i] = p.skip ? *reinterpret_cast<const float4*>(p.skip + row * D + col) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int r = wid * 8 + i;
    const int64_t row = (int64_t)row0 + r;
    float4 v = sk[i];
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const float4 t = *reinterpret_cast<const float4*>(red + (w * FF_RB + r) * YP + col);
      v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
    }
    if (row < p.M) *reinterpret_cast<float4*>(p.dx + row * D + col) = v;
  }
}

// ------------------------------------------------------------------------------------------------ C ABI
extern int g_otr_ffn2_ablate;
int32_t ffn3_takes(int32_t F, int32_t S);                // ffn3.hip
int32_t ffn3_fwd_launch(const void* x16, const void* w1_pack, const float* b1, const void* w2_pack, float* slabs, int32_t S, int64_t M,
                        int32_t F, hipStream_t stream);
int64_t ffn3_scratch_bytes(int64_t M);
int64_t ffn3_sync_ints(int64_t M);
int64_t ffn3_hsave_bytes(int64_t M, int32_t F);
int64_t ffn3_padded_rows(int64_t M);
int32_t ffn3_ln_fwd_launch(const float* x, const void* x16, const void* w1_pack, const float* b1, const void* w2_pack, const float* b2,
                           const float* gamma, const float* beta, const uint64_t* seed, float p_drop, uint64_t rng_offset, float eps,
                           float* y, void* y16, float* z, float* mean, float* rstd, void* hsave, void* usave, float* scratch,
                           int32_t* sync, int64_t M, int32_t F, hipStream_t stream);
int32_t ffn3_bwd_launch(const void* dy16, const void* hsave, const void* w2t_pack, const void* w1t_pack, void* dh, const float* skip,
                        float* dx, float* scratch, int32_t* sync, int64_t M, int32_t F, hipStream_t stream);
extern int g_otr_ffn_waves;   // tuning hook (otr_debug_set(5, v)): 8 = the 8-wave form of the forward kernel, anything else = 4 waves
static int32_t ffn_shape_check(const char* who, int64_t M, int32_t F, int32_t d_model) {
  OTR_REQUIRE(M >= 0 && M < (1ll << 31), "%s: bad M", who);
  OTR_REQUIRE(d_model == 256, "%s: built for d_model = 256 (got %d); use the unfused path", who, d_model);
  OTR_REQUIRE(F > 0 && F % 256 == 0, "%s: d_ff = %d must be a multiple of 256", who, F);
  return 0;
}

extern "C" int32_t otr_ffn_ln_fwd(const float* x, const void* x16, const void* w1_pack, const float* b1, const void* w2_pack,
                                  const float* b2, const float* gamma, const float* beta, const uint64_t* seed, float p_drop,
                                  uint64_t rng_offset, float eps, float* y, void* y16, float* z, float* mean, float* rstd,
                                  int64_t M, int32_t F, int32_t d_model, void* stream) {
  if (int32_t e = ffn_shape_check("ffn_ln_fwd", M, F, d_model)) return e;
  OTR_REQUIRE(x && x16 && w1_pack && b1 && w2_pack && b2 && gamma && beta && y && mean && rstd, "ffn_ln_fwd: null pointer");
  OTR_REQUIRE(p_drop >= 0.f && p_drop < 1.f && (p_drop == 0.f || seed), "ffn_ln_fwd: bad dropout arguments");
  OTR_REQUIRE(((uintptr_t)x16 | (uintptr_t)w1_pack | (uintptr_t)w2_pack | (uintptr_t)x | (uintptr_t)y | (uintptr_t)b1 | (uintptr_t)b2) % 16 == 0,
              "ffn_ln_fwd: buffers must be 16-byte aligned");
  if (M == 0) return 0;
  FfnFwdArgs p{};
  p.x = x; p.x16 = (const uint16_t*)x16; p.p1 = (const uint4*)w1_pack; p.b1 = b1; p.p2 = (const uint4*)w2_pack; p.b2 = b2;
  p.gamma = gamma; p.beta = beta; p.seed = seed; p.y = y; p.y16 = (uint16_t*)y16; p.z = z; p.mean = mean; p.rstd = rstd;
  p.M = (int)M; p.F = F; p.eps = eps; p.p_drop = p_drop; p.rng_offset = rng_offset;
  const unsigned nblk = (unsigned)((M + FF_RB - 1) / FF_RB);
  if (g_otr_ffn_waves == 16 && F <= 2048)       // weights through LDS-DMA rings (otr_debug_set(5, 16))
    hipLaunchKernelGGL((ffn_ln_fwd_dma_kernel<256, 24>), dim3(nblk), dim3(256), 0, (hipStream_t)stream, p);
  else if (g_otr_ffn_waves != 8 || F % 512 != 0)
    hipLaunchKernelGGL((ffn_ln_fwd_kernel<256, 4, 24>), dim3(nblk), dim3(256), 0, (hipStream_t)stream, p);
  else
    hipLaunchKernelGGL((ffn_ln_fwd_kernel<256, 8, 12>), dim3(nblk), dim3(512), 0, (hipStream_t)stream, p);
  return otr_check_launch("ffn_ln_fwd");
}

extern "C" int64_t otr_ffn_split_scratch_bytes(int64_t M) { return M > 0 ? ffn3_scratch_bytes(M) : 0; }
extern "C" int64_t otr_ffn_split_sync_ints(int64_t M) { return M > 0 ? ffn3_sync_ints(M) : 0; }

extern "C" int64_t otr_ffn_split_hsave_bytes(int64_t M, int32_t F) { return M > 0 && F > 0 ? ffn3_hsave_bytes(M, F) : 0; }
extern "C" int64_t otr_ffn_split_padded_rows(int64_t M) { return M > 0 ? ffn3_padded_rows(M) : 0; }

extern "C" int32_t otr_ffn_ln_fwd_split(const float* x, const void* x16, const void* w1_pack, const float* b1, const void* w2_pack,
                                        const float* b2, const float* gamma, const float* beta, const uint64_t* seed, float p_drop,
                                        uint64_t rng_offset, float eps, float* y, void* y16, float* z, float* mean, float* rstd,
                                        void* hsave, void* usave, void* scratch, int64_t scratch_bytes, int32_t* sync,
                                        int64_t sync_ints, int64_t M, int32_t F, int32_t d_model, void* stream) {
  if (int32_t e = ffn_shape_check("ffn_ln_fwd_split", M, F, d_model)) return e;
  OTR_REQUIRE(ffn3_takes(F, 4), "ffn_ln_fwd_split: d_ff = %d does not split into 4 slices of whole 64-unit chunks", F);
  OTR_REQUIRE(x && x16 && w1_pack && b1 && w2_pack && b2 && gamma && beta && y && mean && rstd && scratch && sync, "ffn_ln_fwd_split: null pointer");
  OTR_REQUIRE((hsave == nullptr) == (usave == nullptr), "ffn_ln_fwd_split: hsave and usave come together");
  OTR_REQUIRE(p_drop >= 0.f && p_drop < 1.f && (p_drop == 0.f || seed), "ffn_ln_fwd_split: bad dropout arguments");
  OTR_REQUIRE(((uintptr_t)x16 | (uintptr_t)w1_pack | (uintptr_t)w2_pack | (uintptr_t)x | (uintptr_t)y | (uintptr_t)b1 | (uintptr_t)b2 |
               (uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)scratch | (uintptr_t)z | (uintptr_t)y16 | (uintptr_t)hsave | (uintptr_t)usave) % 16 == 0,
              "ffn_ln_fwd_split: buffers must be 16-byte aligned");
  OTR_REQUIRE(scratch_bytes >= ffn3_scratch_bytes(M) && sync_ints >= ffn3_sync_ints(M), "ffn_ln_fwd_split: scratch / sync too small");
  if (M == 0) return 0;
  return ffn3_ln_fwd_launch(x, x16, w1_pack, b1, w2_pack, b2, gamma, beta, seed, p_drop, rng_offset, eps, y, y16, z, mean, rstd,
                            hsave, usave, (float*)scratch, sync, M, F, (hipStream_t)stream);
}

extern "C" int32_t otr_ffn_bwd_split(const void* dy16, const void* hsave, const void* w2t_pack, const void* w1t_pack, void* dh,
                                     const float* skip, float* dx, void* scratch, int64_t scratch_bytes, int32_t* sync,
                                     int64_t sync_ints, int64_t M, int32_t F, int32_t d_model, void* stream) {
  if (int32_t e = ffn_shape_check("ffn_bwd_split", M, F, d_model)) return e;
  OTR_REQUIRE(ffn3_takes(F, 4), "ffn_bwd_split: d_ff = %d does not split into 4 slices of whole 64-unit chunks", F);
  OTR_REQUIRE(dy16 && hsave && w2t_pack && w1t_pack && dh && dx && scratch && sync, "ffn_bwd_split: null pointer");
  OTR_REQUIRE(((uintptr_t)dy16 | (uintptr_t)hsave | (uintptr_t)w2t_pack | (uintptr_t)w1t_pack | (uintptr_t)dh | (uintptr_t)skip |
               (uintptr_t)dx | (uintptr_t)scratch) % 16 == 0, "ffn_bwd_split: buffers must be 16-byte aligned");
  OTR_REQUIRE(scratch_bytes >= ffn3_scratch_bytes(M) && sync_ints >= ffn3_sync_ints(M), "ffn_bwd_split: scratch / sync too small");
  if (M == 0) return 0;
  return ffn3_bwd_launch(dy16, hsave, w2t_pack, w1t_pack, dh, skip, dx, (float*)scratch, sync, M, F, (hipStream_t)stream);
}

extern "C" int32_t otr_ffn_bwd(const void* x16, const void* dy16, const void* w1_pack, const float* b1, const void* w2t_pack,
                               const void* w1t_pack, void* dh, void* u, float* db1_part, const float* skip, float* dx, int64_t M,
                               int32_t F, int32_t d_model, void* stream) {
  if (int32_t e = ffn_shape_check("ffn_bwd", M, F, d_model)) return e;
  OTR_REQUIRE(x16 && dy16 && w1_pack && b1 && w2t_pack && w1t_pack && dh && u && dx && db1_part, "ffn_bwd: null pointer");
  OTR_REQUIRE((uintptr_t)db1_part % 16 == 0, "ffn_bwd: db1_part must be 16-byte aligned");
  OTR_REQUIRE(((uintptr_t)x16 | (uintptr_t)dy16 | (uintptr_t)w1_pack | (uintptr_t)w2t_pack | (uintptr_t)w1t_pack | (uintptr_t)dh |
               (uintptr_t)u | (uintptr_t)dx | (uintptr_t)skip | (uintptr_t)b1) % 16 == 0, "ffn_bwd: buffers must be 16-byte aligned");
  if (M == 0) return 0;
  FfnBwdArgs p{};
  p.x16 = (const uint16_t*)x16; p.dy16 = (const uint16_t*)dy16; p.p1 = (const uint4*)w1_pack; p.b1 = b1;
  p.p3 = (const uint4*)w2t_pack; p.p4 = (const uint4*)w1t_pack; p.dh = (uint16_t*)dh; p.u = (uint16_t*)u; p.bpart = db1_part; p.skip = skip; p.dx = dx;
  p.M = (int)M; p.F = F; p.ablate = g_otr_ffn2_ablate;
  hipLaunchKernelGGL(ffn_bwd_kernel<256>, dim3((unsigned)((M + FF_RB - 1) / FF_RB)), dim3(256), 0, (hipStream_t)stream, p);
  return otr_check_launch("ffn_bwd");
}

// ================================================================================================ v2: shared weight stream
// Measured on MI355X (tools/ubench/l2stream.hip, profiles/r02_l2stream.txt): a CU ingests global memory at ~22 B/clk
// with 4 waves (33 B/clk with 8), whether the lines come from L2, MALL or even L1 -- the v1 kernels above, whose waves
// each stream their own weight fragments into registers, sit exactly on that ceiling (3 MB per CU -> 60 us forward).
// v2 cuts the ingest per CU by 4: a workgroup owns 128 rows (wave w: rows 32w..32w+31) and 1/S of the hidden units
// (grid = row blocks x S = 63 x 4 at B=32); every weight fragment is fetched ONCE per workgroup, global -> LDS by
// direct-to-LDS loads (global_load_lds_dwordx4: the fragment-major packs are exactly the lane-linear image those
// need), and read from LDS by all four waves (ds_read_b128, conflict-free).  Chunk c+1 streams in while chunk c is
// multiplied: one barrier per chunk of 48 (forward) / 80 (backward) MFMAs per wave.  The hidden-dimension split leaves S
// partial fp32 output slabs; the LayerNorm kernel (otr_add_layernorm_fwd_slabs) / a small reduce kernel sums them.
struct Ffn2FwdArgs {
  const uint16_t* x16;     // [M, D]
  const uint4* p1; const float* b1; const uint4* p2;
  float* slabs;            // [S][M][D] f32 partial outputs (bias b_2 NOT included)
  int M, F, S;
  int ablate;              // tuning hook (otr_debug_set(4, v)); 0 in production
};

template <int D>
__global__ __launch_bounds__(256, 1) void ffn2_fwd_kernel(Ffn2FwdArgs p) {
  constexpr int NKS = D / 16, NT = D / 32;
  constexpr int FR = 2 * NKS + 2 * NT;             // fragments per chunk: 32 (w_1 value+gate) + 16 (w_2)
  constexpr int BUF = FR * 1024;
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * BUF + 8192];
  float* bias_s = reinterpret_cast<float*>(smem + 2 * BUF);      // b_1 of this workgroup's hidden units: [chunk][value 32 | gate 32]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 31, hi = lane >> 5;
  const int row0 = blockIdx.x * 128 + wid * 32;
  const int nchunk = p.F / 32, per = nchunk / p.S;               // chunks of this workgroup: [c0, c0 + per)
  const int c0 = blockIdx.y * per;

  for (int i = tid; i < per * 64; i += 256) {
    const int c = i >> 6, j = i & 63;
    bias_s[i] = p.b1[(j < 32 ? 0 : p.F) + (c0 + c) * 32 + (j & 31)];
  }
  // this wave's activation rows as MFMA B operands, held in registers for the whole kernel
  uint4 xf[NKS];
  {
    const int64_t row = min(row0 + m, p.M - 1);
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) xf[ks] = ld_global_b128(p.x16 + row * D + ks * 16 + hi * 8);
  }
  // fragment f of chunk c -> source fragment; every wave DMAs FR/4 of them
  auto issue = [&](int c, int buf) {
#pragma unroll
    for (int j = 0; j < FR / 4; ++j) {
      const int f = wid * (FR / 4) + j;              // wave-uniform
      const uint4* src;
      if (f < 2 * NKS) src = p.p1 + (int64_t)(((f & 1) ? nchunk + c : c) * NKS + (f >> 1)) * 64;
      else { const int t = f - 2 * NKS; src = p.p2 + (int64_t)((t >> 1) * (2 * nchunk) + 2 * c + (t & 1)) * 64; }
      dma_frag(src, smem + buf * BUF + f * 1024, lane);
    }
  };

  f32x16 yacc[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) yacc[i][r] = 0.f;

  // Software pipeline across chunks: the second GEMM of chunk i-1 (y += w_2 . u) is issued together with the GLU of
  // chunk i, so the matrix pipe has work while the VALU computes sigmoid / products / packs of the chunk just
  // multiplied (measured: with GEMM1 -> GLU -> GEMM2 in sequence a chunk took 4.2 k cycles for 1.5 k cycles of MFMAs).
  // The w_2 fragments of a chunk are therefore copied LDS -> registers while that chunk's first GEMM runs (its buffer
  // is refilled during the next iteration) and used one iteration later.
  constexpr int G = 8, NG1 = 2 * NKS / G;           // first GEMM: 32 fragments in groups of 8 (LDS -> register prefetch)
  static_assert((2 * NKS) % G == 0, "whole prefetch groups");
  uint4 w2r[2 * NT], uf0, uf1;

  // GEMM1 of chunk i (weights in LDS buffer i&1) into av / ag, with the chunk's bias and w_2 fragments fetched alongside
#define FFN2_GEMM1(I)                                                                                  \
  {                                                                                                     \
    const uint4* wb = reinterpret_cast<const uint4*>(smem + ((I) & 1) * BUF) + lane;                    \
    _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                     \
      bv[q] = *reinterpret_cast<const float4*>(bias_s + (I) * 64 + 8 * q + 4 * hi);                     \
      bg[q] = *reinterpret_cast<const float4*>(bias_s + (I) * 64 + 32 + 8 * q + 4 * hi);                \
    }                                                                                                   \
    _Pragma("unroll") for (int r = 0; r < 16; ++r) { av[r] = 0.f; ag[r] = 0.f; }                        \
    uint4 fr[2][G];                                                                                     \
    _Pragma("unroll") for (int j = 0; j < G; ++j) fr[0][j] = wb[j * 64];                                \
    _Pragma("unroll") for (int g = 0; g < NG1; ++g) {                                                   \
      if (g + 1 < NG1) {                                                                                \
        _Pragma("unroll") for (int j = 0; j < G; ++j) fr[(g + 1) & 1][j] = wb[((g + 1) * G + j) * 64];  \
      } else {                                                                                          \
        _Pragma("unroll") for (int j = 0; j < 2 * NT; ++j) nw[j] = wb[(2 * NKS + j) * 64];              \
      }                                                                                                 \
      __builtin_amdgcn_sched_barrier(0);                                                                \
      _Pragma("unroll") for (int j = 0; j < G; ++j) {                                                   \
        const int f = g * G + j;                                                                        \
        if (f & 1) mma32(ag, fr[g & 1][j], xf[f >> 1]); else mma32(av, fr[g & 1][j], xf[f >> 1]);       \
      }                                                                                                 \
      __builtin_amdgcn_sched_barrier(0);                                                                \
    }                                                                                                   \
  }
#define FFN2_GLU()                                                                                      \
  {                                                                                                     \
    float u[16];                                                                                        \
    _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                    \
      const float a = av[r] + reinterpret_cast<const float*>(&bv[r >> 2])[r & 3];                       \
      const float gt = ag[r] + reinterpret_cast<const float*>(&bg[r >> 2])[r & 3];                      \
      u[r] = a * fast_sigmoid(gt);                                                                      \
    }                                                                                                   \
    tile_to_frags(u, n0, n1);                                                                           \
  }
#define FFN2_GEMM2()                                                                                    \
  _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) {                                                   \
    mma32(yacc[nt], w2r[2 * nt], uf0);                                                                  \
    mma32(yacc[nt], w2r[2 * nt + 1], uf1);                                                              \
  }
#define FFN2_ROTATE()                                                                                   \
  {                                                                                                     \
    uf0 = n0; uf1 = n1;                                                                                 \
    _Pragma("unroll") for (int j = 0; j < 2 * NT; ++j) w2r[j] = nw[j];                                  \
  }
#define FFN2_ARRIVE(I)                                                                                  \
  {                                                                                                     \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                    \
    __syncthreads(); /* chunk I has landed for every wave; buffer (I+1)&1 is free */                    \
    if ((I) + 1 < per && !(p.ablate & 1)) issue(c0 + (I) + 1, ((I) + 1) & 1);                           \
    __builtin_amdgcn_sched_barrier(0);                                                                  \
  }
  f32x16 av, ag;
  float4 bv[4], bg[4];
  uint4 nw[2 * NT], n0, n1;
  issue(c0, 0);
  // The activation fragments must be KNOWN complete before the loop: hipcc's waitcnt pass otherwise carries "up to 16
  // loads may be pending" around the back edge and guards every xf use with vmcnt(15) ... vmcnt(0) -- which, since the
  // counter also holds the DMA just issued for the next chunk, waits for that DMA in the middle of the current chunk.
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) asm volatile("" :: "v"(xf[ks].x), "v"(xf[ks].y), "v"(xf[ks].z), "v"(xf[ks].w));
  FFN2_ARRIVE(0)
  if (!(p.ablate & 2)) {
    FFN2_GEMM1(0)
    FFN2_GLU()
    FFN2_ROTATE()
  }
  for (int i = 1; i < per; ++i) {
    FFN2_ARRIVE(i)
    if (p.ablate & 2) continue;
    FFN2_GEMM1(i)
    FFN2_GEMM2()                                      // chunk i-1, registers only: runs beside the GLU of chunk i
    FFN2_GLU()
    __builtin_amdgcn_sched_barrier(0);
    FFN2_ROTATE()
  }
  if (!(p.ablate & 2)) { FFN2_GEMM2() }
#undef FFN2_GEMM1
#undef FFN2_GLU
#undef FFN2_GEMM2
#undef FFN2_ROTATE
#undef FFN2_ARRIVE
  // partial output rows -> slab blockIdx.y, staged through the (now idle) buffer (per & 1) so that memory sees whole
  // 256-byte row segments: each wave uses a private 8 KiB slice, 64 columns at a time (no workgroup barrier needed)
  float* st = reinterpret_cast<float*>(smem + (per & 1) * BUF + wid * 8192);      // [32 rows][64 cols]
  float* out = p.slabs + ((int64_t)blockIdx.y * p.M) * D;
#pragma unroll
  for (int h = 0; h < NT / 2; ++h) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *reinterpret_cast<float4*>(st + m * 64 + t * 32 + 8 * q + 4 * hi) =
            make_float4(yacc[2 * h + t][4 * q], yacc[2 * h + t][4 * q + 1], yacc[2 * h + t][4 * q + 2], yacc[2 * h + t][4 * q + 3]);
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {                 // 4 rows x 16 lanes x float4 per pass
      const int r = rr * 4 + (lane >> 4);
      const int64_t row = (int64_t)row0 + r;
      const float4 v = *reinterpret_cast<const float4*>(st + r * 64 + (lane & 15) * 4);
      if (row < p.M) *reinterpret_cast<float4*>(out + row * D + h * 64 + (lane & 15) * 4) = v;
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
}

struct Ffn2BwdArgs {
  const uint16_t* x16; const uint16_t* dy16;
  const uint4* p1; const float* b1; const uint4* p3; const uint4* p4;
  uint16_t* dh; uint16_t* u;
  float* bpart;            // [ceil(M/32)][2F]: column sums of dh per 32-row block (= per wave)
  float* slabs;            // [S][M][D] f32 partial input gradients
  int M, F, S;
};

template <int D>
__global__ __launch_bounds__(256, 1) void ffn2_bwd_kernel(Ffn2BwdArgs p) {
  constexpr int NKS = D / 16, NT = D / 32;
  constexpr int S1 = 2 * NKS, S2 = S1 + NKS, FR = S2 + 4 * NT;    // 32 + 16 + 32 fragments per chunk
  constexpr int BUF = FR * 1024;                                  // 80 KiB: two of them are the whole LDS
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * BUF];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 31, hi = lane >> 5;
  const int row0 = blockIdx.x * 128 + wid * 32;
  const int nchunk = p.F / 32, per = nchunk / p.S;
  const int c0 = blockIdx.y * per;
  const int64_t grow = (int64_t)row0 + m;
  const bool live = grow < p.M;
  const int64_t crow = min(grow, (int64_t)p.M - 1);

  uint4 xf[NKS], df[NKS];
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) {
    xf[ks] = ld_global_b128(p.x16 + crow * D + ks * 16 + hi * 8);
    df[ks] = ld_global_b128(p.dy16 + crow * D + ks * 16 + hi * 8);
  }
  auto issue = [&](int c, int buf) {
#pragma unroll
    for (int j = 0; j < FR / 4; ++j) {
      const int f = wid * (FR / 4) + j;
      const uint4* src;
      if (f < S1) src = p.p1 + (int64_t)(((f & 1) ? nchunk + c : c) * NKS + (f >> 1)) * 64;
      else if (f < S2) src = p.p3 + (int64_t)(c * NKS + (f - S1)) * 64;
      else {
        const int t = f - S2, j4 = t & 3;
        const int ksf = (j4 < 2) ? 2 * c + j4 : 2 * nchunk + 2 * c + (j4 - 2);
        src = p.p4 + (int64_t)((t >> 2) * (4 * nchunk) + ksf) * 64;
      }
      dma_frag(src, smem + buf * BUF + f * 1024, lane);
    }
  };

  f32x16 xacc[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) xacc[i][r] = 0.f;

  issue(c0, 0);
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) {                 // see ffn2_fwd_kernel: the operand fragments are complete before the loop
    asm volatile("" :: "v"(xf[ks].x), "v"(xf[ks].y), "v"(xf[ks].z), "v"(xf[ks].w));
    asm volatile("" :: "v"(df[ks].x), "v"(df[ks].y), "v"(df[ks].z), "v"(df[ks].w));
  }
  for (int i = 0; i < per; ++i) {
    const int c = c0 + i;
    // b_1 of this chunk: plain loads BEFORE the wait below (they retire with it; nothing else is in flight then)
    float4 bv[4], bg[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      bv[q] = *reinterpret_cast<const float4*>(p.b1 + c * 32 + 8 * q + 4 * hi);
      bg[q] = *reinterpret_cast<const float4*>(p.b1 + p.F + c * 32 + 8 * q + 4 * hi);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (i + 1 < per) issue(c + 1, (i + 1) & 1);
    __builtin_amdgcn_sched_barrier(0);
    const uint4* wb = reinterpret_cast<const uint4*>(smem + (i & 1) * BUF) + lane;
    f32x16 av, ag, du;
#pragma unroll
    for (int r = 0; r < 16; ++r) { av[r] = 0.f; ag[r] = 0.f; du[r] = 0.f; }
    constexpr int G = 8, NG = FR / G;               // LDS -> register prefetch groups (see ffn2_fwd_kernel)
    static_assert(FR % G == 0 && S2 % G == 0, "group boundary must fall in front of the dx GEMM");
    uint4 fr[2][G], hf[4];
#pragma unroll
    for (int j = 0; j < G; ++j) fr[0][j] = wb[j * 64];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      if (g + 1 < NG) {
#pragma unroll
        for (int j = 0; j < G; ++j) fr[(g + 1) & 1][j] = wb[((g + 1) * G + j) * 64];
      }
      __builtin_amdgcn_sched_barrier(0);
      if (g * G == S2) {
        float uu[16], da_[16], dg_[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float a = av[r] + reinterpret_cast<const float*>(&bv[r >> 2])[r & 3];
          const float sg = fast_sigmoid(ag[r] + reinterpret_cast<const float*>(&bg[r >> 2])[r & 3]);
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
        float* bp = p.bpart + ((int64_t)blockIdx.x * 4 + wid) * (2 * p.F) + c * 32;
        tile_colsum_store(da_, bp, lane, hi, live);
        tile_colsum_store(dg_, bp + p.F, lane, hi, live);
      }
#pragma unroll
      for (int j = 0; j < G; ++j) {
        const int f = g * G + j;
        const uint4 w = fr[g & 1][j];
        if (f < S1) {
          if (f & 1) mma32(ag, w, xf[f >> 1]); else mma32(av, w, xf[f >> 1]);
        } else if (f < S2) {
          mma32(du, w, df[f - S1]);
        } else {
          const int t = f - S2;
          mma32(xacc[t >> 2], w, hf[t & 3]);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  __syncthreads();                                   // every wave is done with both weight buffers (the stores above are global)
  float* st = reinterpret_cast<float*>(smem + wid * 8192);
  float* out = p.slabs + ((int64_t)blockIdx.y * p.M) * D;
#pragma unroll
  for (int h = 0; h < NT / 2; ++h) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *reinterpret_cast<float4*>(st + m * 64 + t * 32 + 8 * q + 4 * hi) =
            make_float4(xacc[2 * h + t][4 * q], xacc[2 * h + t][4 * q + 1], xacc[2 * h + t][4 * q + 2], xacc[2 * h + t][4 * q + 3]);
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
      const int r = rr * 4 + (lane >> 4);
      const int64_t row = (int64_t)row0 + r;
      const float4 v = *reinterpret_cast<const float4*>(st + r * 64 + (lane & 15) * 4);
      if (row < p.M) *reinterpret_cast<float4*>(out + row * D + h * 64 + (lane & 15) * 4) = v;
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
}

// out[M, D] = sum of S slabs (+ skip): the input gradient of the FFN sub-layer after the hidden-dimension split
__global__ __launch_bounds__(256) void slab_sum_kernel(const float* __restrict__ slabs, int S, int64_t n4, const float* skip, float* out) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    float4 a = skip ? reinterpret_cast<const float4*>(skip)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    for (int s = 0; s < S; ++s) {
      const float4 v = reinterpret_cast<const float4*>(slabs)[(int64_t)s * n4 + i];
      a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    reinterpret_cast<float4*>(out)[i] = a;
  }
}

static int32_t ffn2_split_check(const char* who, int32_t F, int32_t S) {
  OTR_REQUIRE(S >= 1 && S <= 16 && (F / 32) % S == 0, "%s: d_ff / 32 = %d chunks do not split into %d parts", who, F / 32, S);
  return 0;
}

extern "C" int32_t otr_ffn_fwd_slabs(const void* x16, const void* w1_pack, const float* b1, const void* w2_pack, float* slabs,
                                     int32_t n_slabs, int64_t M, int32_t F, int32_t d_model, void* stream) {
  if (int32_t e = ffn_shape_check("ffn_fwd_slabs", M, F, d_model)) return e;
  if (int32_t e = ffn2_split_check("ffn_fwd_slabs", F, n_slabs)) return e;
  OTR_REQUIRE(x16 && w1_pack && b1 && w2_pack && slabs, "ffn_fwd_slabs: null pointer");
  OTR_REQUIRE(((uintptr_t)x16 | (uintptr_t)w1_pack | (uintptr_t)w2_pack | (uintptr_t)slabs) % 16 == 0, "ffn_fwd_slabs: buffers must be 16-byte aligned");
  OTR_REQUIRE((F / 32 / n_slabs) * 64 * 4 <= 8192, "ffn_fwd_slabs: too many hidden units per workgroup for the bias staging");
  if (M == 0) return 0;
  if (g_otr_ffn_waves != 2 && ffn3_takes(F, n_slabs))     // third form (ffn3.hip); otr_debug_set(5, 2) keeps the second for A/B runs
    return ffn3_fwd_launch(x16, w1_pack, b1, w2_pack, slabs, n_slabs, M, F, (hipStream_t)stream);
  Ffn2FwdArgs p{};
  p.x16 = (const uint16_t*)x16; p.p1 = (const uint4*)w1_pack; p.b1 = b1; p.p2 = (const uint4*)w2_pack; p.slabs = slabs;
  p.M = (int)M; p.F = F; p.S = n_slabs; p.ablate = g_otr_ffn2_ablate;
  hipLaunchKernelGGL(ffn2_fwd_kernel<256>, dim3((unsigned)((M + 127) / 128), (unsigned)n_slabs), dim3(256), 0, (hipStream_t)stream, p);
  return otr_check_launch("ffn_fwd_slabs");
}

extern "C" int32_t otr_ffn_bwd_slabs(const void* x16, const void* dy16, const void* w1_pack, const float* b1, const void* w2t_pack,
                                     const void* w1t_pack, void* dh, void* u, float* db1_part, float* slabs, int32_t n_slabs,
                                     int64_t M, int32_t F, int32_t d_model, void* stream) {
  if (int32_t e = ffn_shape_check("ffn_bwd_slabs", M, F, d_model)) return e;
  if (int32_t e = ffn2_split_check("ffn_bwd_slabs", F, n_slabs)) return e;
  OTR_REQUIRE(x16 && dy16 && w1_pack && b1 && w2t_pack && w1t_pack && dh && u && slabs && db1_part, "ffn_bwd_slabs: null pointer");
  OTR_REQUIRE((uintptr_t)db1_part % 16 == 0, "ffn_bwd_slabs: db1_part must be 16-byte aligned");
  OTR_REQUIRE(((uintptr_t)x16 | (uintptr_t)dy16 | (uintptr_t)w1_pack | (uintptr_t)w2t_pack | (uintptr_t)w1t_pack | (uintptr_t)dh |
               (uintptr_t)u | (uintptr_t)slabs | (uintptr_t)b1) % 16 == 0, "ffn_bwd_slabs: buffers must be 16-byte aligned");
  if (M == 0) return 0;
  Ffn2BwdArgs p{};
  p.x16 = (const uint16_t*)x16; p.dy16 = (const uint16_t*)dy16; p.p1 = (const uint4*)w1_pack; p.b1 = b1;
  p.p3 = (const uint4*)w2t_pack; p.p4 = (const uint4*)w1t_pack; p.dh = (uint16_t*)dh; p.u = (uint16_t*)u; p.bpart = db1_part; p.slabs = slabs;
  p.M = (int)M; p.F = F; p.S = n_slabs;
  hipLaunchKernelGGL(ffn2_bwd_kernel<256>, dim3((unsigned)((M + 127) / 128), (unsigned)n_slabs), dim3(256), 0, (hipStream_t)stream, p);
  return otr_check_launch("ffn_bwd_slabs");
}

extern "C" int32_t otr_slab_sum(const float* slabs, int32_t n_slabs, int64_t n, const float* skip, float* out, void* stream) {
  OTR_REQUIRE(slabs && out && n_slabs >= 1 && n >= 0 && n % 4 == 0, "slab_sum: bad arguments");
  OTR_REQUIRE(((uintptr_t)slabs | (uintptr_t)out | (uintptr_t)skip) % 16 == 0, "slab_sum: buffers must be 16-byte aligned");
  if (n == 0) return 0;
  const int64_t n4 = n / 4;
  unsigned g = (unsigned)((n4 + 255) / 256 > 4096 ? 4096 : (n4 + 255) / 256);
  hipLaunchKernelGGL(slab_sum_kernel, dim3(g), dim3(256), 0, (hipStream_t)stream, slabs, n_slabs, n4, skip, out);
  return otr_check_launch("slab_sum");
}
