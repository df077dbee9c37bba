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
// A workgroup packs PF_G consecutive table blocks (PF_G x 4 fragments).  Two things bounded the launch (~110 MB over 13 k blocks at the
// bench model, 59 us = 2 TB/s): (i) a binary search of the table in front of every block (7 dependent global loads) -- every wave now
// finds its items in ONE round trip (lane i loads the first block of item i; the item of block b is the number of entries <= b, minus
// one), the item rows come through the scalar cache and the source pieces of the PF_G blocks are in flight together; (ii) the loads in
// fragment layout touch one 128-byte line per lane PAIR (32 lines per instruction, every line four times from four waves) -- where the
// shapes allow it the workgroup fetches the 4 KiB source tile of its four fragments as whole lines (8 lanes per line, each line once)
// and the waves take their fragments out of LDS:
//   contraction index contiguous (cs == 1): fragments (rt, ks0 .. ks0 + 3) = 32 rows x 128 bytes;
//   free index contiguous (rs == 1):        fragments (rt0 .. rt0 + 1, ks0 .. ks0 + 1) = 32 contraction rows x 128 bytes -- block bi of
//                                           the item takes the rt pair bi / (nks / 2), the ks pair bi % (nks / 2).
// The destination of a fragment does not depend on which block packs it.
constexpr int PF_G = 2, PF_SCAN = 8;            // items scanned per wave: 64 * PF_SCAN (more: the binary search)
constexpr int PF_TS = 64 + 8;                   // elements per row of a source tile in LDS
__device__ __forceinline__ int pf_kmap(int perm, int hi, int j) { return perm ? (j < 4 ? 4 * hi + j : 8 + 4 * hi + (j - 4)) : hi * 8 + j; }
__global__ __launch_bounds__(256) void pack_frags_kernel(const uint16_t* __restrict__ src, uint16_t* __restrict__ dst,
                                                        const int64_t* __restrict__ table, int n, int64_t total_blocks) {
  __shared__ __attribute__((aligned(16))) uint16_t tile[PF_G][32][PF_TS];
  __shared__ __attribute__((aligned(16))) uint16_t wtile[PF_G][4][16][32 + 8];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, hi = lane >> 5;
  int64_t vb[PF_G];
  int item[PF_G];
#pragma unroll
  for (int g = 0; g < PF_G; ++g) { vb[g] = (int64_t)blockIdx.x * PF_G + g; item[g] = -1; }
  if (n <= 64 * PF_SCAN) {
    int64_t fb[PF_SCAN];
#pragma unroll
    for (int k = 0; k < PF_SCAN; ++k) {
      const int i = k * 64 + lane;
      fb[k] = i < n ? table[(int64_t)i * 8 + 7] : INT64_MAX;
    }
#pragma unroll
    for (int k = 0; k < PF_SCAN; ++k)
#pragma unroll
      for (int g = 0; g < PF_G; ++g) item[g] += __popcll(__ballot(fb[k] <= vb[g]));
  } else {
#pragma unroll
    for (int g = 0; g < PF_G; ++g) {
      int lo = 0, hi_ = n - 1;
      while (lo < hi_) {
        const int mid = (lo + hi_ + 1) >> 1;
        if (table[mid * 8 + 7] <= vb[g]) lo = mid; else hi_ = mid - 1;
      }
      item[g] = lo;
    }
  }
  enum { NONE, TILE_KC, TILE_MC, DIRECT, PAIR, TURN, SLOW };
  int kind[PF_G], perm_[PF_G];
  uint4 q[PF_G];
  int64_t out[PF_G];
  // ---- every source piece of the PF_G blocks goes out first
#pragma unroll
  for (int g = 0; g < PF_G; ++g) {
    kind[g] = NONE;
    if (vb[g] >= total_blocks) continue;
    const int64_t* t = table + (int64_t)__builtin_amdgcn_readfirstlane(max(item[g], 0)) * 8;
    const int64_t src_off = t[0], rs = t[1], cs = t[2], rows = t[3], cols = t[4], perm = t[5], dst_off = t[6];
    const uint32_t nks = (uint32_t)(cols >> 4), nrt = (uint32_t)(rows >> 5), nfrag = nrt * nks;     // < 2^31 fragments: otr_pack_frags checks the blocks
    const uint32_t bi = (uint32_t)(vb[g] - t[7]);
    perm_[g] = (int)perm;
    if (cs == 1 && nks % 4 == 0 && src_off % 8 == 0 && rs % 8 == 0) {
      const uint32_t frag0 = bi * 4;
      if (frag0 >= nfrag) continue;
      const uint32_t rt = frag0 / nks, ks0 = frag0 - rt * nks;
      kind[g] = TILE_KC;
      q[g] = *reinterpret_cast<const uint4*>(src + src_off + (int64_t)(rt * 32 + (tid >> 3)) * rs + ks0 * 16 + (tid & 7) * 8);
      out[g] = dst_off + (int64_t)(frag0 + wid) * 512 + lane * 8;
      continue;
    }
    if (rs == 1 && nks % 2 == 0 && nrt % 2 == 0 && src_off % 8 == 0 && cs % 8 == 0) {
      if (bi * 4 >= nfrag) continue;
      const uint32_t hk = nks / 2, rt0 = 2 * (bi / hk), ks0 = 2 * (bi % hk);
      kind[g] = TILE_MC;
      q[g] = *reinterpret_cast<const uint4*>(src + src_off + (int64_t)(ks0 * 16 + (tid >> 3)) * cs + rt0 * 32 + (tid & 7) * 8);
      out[g] = dst_off + (int64_t)((rt0 + (wid >> 1)) * nks + ks0 + (wid & 1)) * 512 + lane * 8;
      continue;
    }
    const uint32_t frag = bi * 4 + wid;
    if (frag >= nfrag) continue;
    const uint32_t rt = frag / nks, ks = frag - rt * nks;
    const int64_t r = (int64_t)rt * 32 + (lane & 31);
    const uint16_t* s = src + src_off + r * rs;
    out[g] = dst_off + (int64_t)frag * 512 + lane * 8;
    if (cs == 1 && perm == 0 && ((src_off + r * rs) % 8 == 0)) {        // contraction index contiguous: one 16-byte load
      kind[g] = DIRECT;
      q[g] = *reinterpret_cast<const uint4*>(s + ks * 16 + hi * 8);
    } else if (cs == 1 && ((src_off + r * rs) % 4 == 0)) {                // contiguous source, 8-byte aligned: two 8-byte loads
      kind[g] = PAIR;
      const uint2 a = *reinterpret_cast<const uint2*>(s + ks * 16 + (perm ? 4 * hi : 8 * hi));
      const uint2 b = *reinterpret_cast<const uint2*>(s + ks * 16 + (perm ? 8 + 4 * hi : 8 * hi + 4));
      q[g] = make_uint4(a.x, a.y, b.x, b.y);
    } else if (rs == 1 && ((src_off + rt * 32) % 8 == 0) && (cs % 8 == 0)) {
      // the FREE index is the contiguous one and the shape has no tile form: every lane fetches ONE 16-byte piece (row lane / 4, piece
      // lane % 4) and the wave turns its 16 x 32 tile through 1 KiB of LDS
      kind[g] = TURN;
      q[g] = *reinterpret_cast<const uint4*>(src + src_off + (int64_t)(ks * 16 + (lane >> 2)) * cs + rt * 32 + (lane & 3) * 8);
    } else {
      kind[g] = SLOW;
      uint16_t v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = s[(int64_t)(ks * 16 + pf_kmap((int)perm, hi, j)) * cs];
      q[g] = make_uint4(v[0] | ((uint32_t)v[1] << 16), v[2] | ((uint32_t)v[3] << 16), v[4] | ((uint32_t)v[5] << 16), v[6] | ((uint32_t)v[7] << 16));
    }
  }
  // ---- tiles -> LDS
#pragma unroll
  for (int g = 0; g < PF_G; ++g) {
    if (kind[g] == TILE_KC || kind[g] == TILE_MC) *reinterpret_cast<uint4*>(&tile[g][tid >> 3][(tid & 7) * 8]) = q[g];
    else if (kind[g] == TURN) *reinterpret_cast<uint4*>(&wtile[g][wid][lane >> 2][(lane & 3) * 8]) = q[g];
  }
  __syncthreads();
  // ---- fragments out of LDS, store
#pragma unroll
  for (int g = 0; g < PF_G; ++g) {
    if (kind[g] == NONE) continue;
    if (kind[g] == TILE_KC) {
      const uint16_t* row = &tile[g][lane & 31][wid * 16];
      if (perm_[g]) {
        const uint2 a = *reinterpret_cast<const uint2*>(row + 4 * hi), b = *reinterpret_cast<const uint2*>(row + 8 + 4 * hi);
        q[g] = make_uint4(a.x, a.y, b.x, b.y);
      } else {
        q[g] = *reinterpret_cast<const uint4*>(row + 8 * hi);
      }
    } else if (kind[g] == TILE_MC || kind[g] == TURN) {
      uint16_t v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int kk = pf_kmap(perm_[g], hi, j);
        v[j] = kind[g] == TILE_MC ? tile[g][(wid & 1) * 16 + kk][(wid >> 1) * 32 + (lane & 31)] : wtile[g][wid][kk][lane & 31];
      }
      q[g] = make_uint4(v[0] | ((uint32_t)v[1] << 16), v[2] | ((uint32_t)v[3] << 16), v[4] | ((uint32_t)v[5] << 16), v[6] | ((uint32_t)v[7] << 16));
    }
    *reinterpret_cast<uint4*>(dst + out[g]) = q[g];
  }
}

extern "C" int32_t otr_pack_frags(const void* src, void* dst, const int64_t* table, int32_t n_items, int64_t total_blocks,
                                  void* stream) {
  OTR_REQUIRE(src && dst && table, "pack_frags: null pointer");
  OTR_REQUIRE(n_items >= 0 && total_blocks >= 0 && total_blocks < (1ll << 31), "pack_frags: bad sizes");
  if (n_items == 0 || total_blocks == 0) return 0;
  hipLaunchKernelGGL(pack_frags_kernel, dim3((unsigned)((total_blocks + PF_G - 1) / PF_G)), dim3(256), 0, (hipStream_t)stream,
                     (const uint16_t*)src, (uint16_t*)dst, table, n_items, total_blocks);
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

// NW = waves per workgroup (4: an 8-wave form -- two waves per SIMD, <= 256 registers each, 12-deep rings -- measured 85 us against
// 61 us in round 2 and was removed).
template <int D, int NW, int PD>
__global__ __launch_bounds__(NW * 64, NW / 4) void ffn_ln_fwd_kernel(FfnFwdArgs p) {
  static_assert(D == 256, "the LayerNorm epilogue maps one float4 per lane: d_model = 256");
  constexpr int NKS = D / 16, NT = D / 32, YP = D + 4;
  constexpr int STEPS = 2 * NKS + 2 * NT;          // weight fragments (= MFMAs) per chunk: 32 + 16
  constexpr int NTHR = NW * 64, RPW = FF_RB / NW;  // PD = fragments in flight per wave (1 KiB each)
  static_assert(NW == 4, "one wave per SIMD");
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

  // the waves' partial y^T tiles meet in LDS: red[slot][m][n], n = nt*32 + 8q + 4hi + (r&3); four slots
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
int32_t ffn3_debug_block_map(int64_t M, int32_t map, int32_t* out, int32_t cap);
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
int32_t ffn3_fwd_slab_launch(const void* x16, const void* w1_pack, const float* b1, const void* w2_pack, void* hsave, void* usave, void* slab,
                             int64_t M, int32_t F, hipStream_t stream);
int32_t ffn3_bwd_slab_launch(const void* dy16, const void* hsave, const void* w2t_pack, const void* w1t_pack, void* dh, void* slab, int64_t M,
                             int32_t F, hipStream_t stream);
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
  hipLaunchKernelGGL((ffn_ln_fwd_kernel<256, 4, 24>), dim3(nblk), dim3(256), 0, (hipStream_t)stream, p);
  return otr_check_launch("ffn_ln_fwd");
}

extern "C" int32_t otr_debug_ffn_split_map(int64_t M, int32_t map, int32_t* out, int32_t cap) {
  OTR_REQUIRE(M > 0 && (map == 0 || map == 1) && cap >= 0, "debug_ffn_split_map: bad arguments");
  return ffn3_debug_block_map(M, map, out, cap);
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

extern "C" int32_t otr_ffn_fwd_split_slab(const void* x16, const void* w1_pack, const float* b1, const void* w2_pack, void* hsave, void* usave,
                                          void* slabs, int64_t M, int32_t F, int32_t d_model, void* stream) {
  if (int32_t e = ffn_shape_check("ffn_fwd_split_slab", M, F, d_model)) return e;
  OTR_REQUIRE(ffn3_takes(F, 4), "ffn_fwd_split_slab: d_ff = %d does not split into 4 slices of whole 64-unit chunks", F);
  OTR_REQUIRE(x16 && w1_pack && b1 && w2_pack && slabs, "ffn_fwd_split_slab: null pointer");
  OTR_REQUIRE((hsave == nullptr) == (usave == nullptr), "ffn_fwd_split_slab: hsave and usave come together");
  OTR_REQUIRE(((uintptr_t)x16 | (uintptr_t)w1_pack | (uintptr_t)w2_pack | (uintptr_t)b1 | (uintptr_t)hsave | (uintptr_t)usave | (uintptr_t)slabs) % 16 == 0,
              "ffn_fwd_split_slab: buffers must be 16-byte aligned");
  if (M == 0) return 0;
  return ffn3_fwd_slab_launch(x16, w1_pack, b1, w2_pack, hsave, usave, slabs, M, F, (hipStream_t)stream);
}

extern "C" int32_t otr_ffn_bwd_split_slab(const void* dy16, const void* hsave, const void* w2t_pack, const void* w1t_pack, void* dh, void* slabs,
                                          int64_t M, int32_t F, int32_t d_model, void* stream) {
  if (int32_t e = ffn_shape_check("ffn_bwd_split_slab", M, F, d_model)) return e;
  OTR_REQUIRE(ffn3_takes(F, 4), "ffn_bwd_split_slab: d_ff = %d does not split into 4 slices of whole 64-unit chunks", F);
  OTR_REQUIRE(dy16 && hsave && w2t_pack && w1t_pack && dh && slabs, "ffn_bwd_split_slab: null pointer");
  OTR_REQUIRE(((uintptr_t)dy16 | (uintptr_t)hsave | (uintptr_t)w2t_pack | (uintptr_t)w1t_pack | (uintptr_t)dh | (uintptr_t)slabs) % 16 == 0,
              "ffn_bwd_split_slab: buffers must be 16-byte aligned");
  if (M == 0) return 0;
  return ffn3_bwd_slab_launch(dy16, hsave, w2t_pack, w1t_pack, dh, slabs, M, F, (hipStream_t)stream);
}
