// Fused scaled-dot-product attention for gfx950, forward + backward, scores never leave registers.
// Replaces module/attention.py:23-46 (compute_context) and the bmm/softmax/bmm around it.
//
// Layout trick used throughout: a 16x16 MFMA accumulator tile (col = lane&15, row = (lane>>4)*4+r)
// can be fed straight back as an MFMA *input* chunk whose free index is the tile's column and whose
// contraction index is the tile's row (any slot->k bijection is legal if A and B agree).  So we
// always compute the score tile with the *contraction index of the next GEMM as its row*:
//   forward / dQ : S^T = K Q^T   (rows = keys,  cols = queries)  -> O^T = V^T P^T, dQ^T = K^T dS^T
//   dK,dV        : S   = Q K^T   (rows = queries, cols = keys)   -> dV^T = dO^T P, dK^T = Q^T dS
// and P / dS go from accumulator registers to MFMA operands with only a bf16 pack: no LDS round trip,
// no cross-lane shuffles except the 2-step row reductions across the four 16-lane groups.
//
// Workgroup = 4 waves; each wave owns 16 queries (fwd, dQ) or 16 keys (dKdV); the other side is
// streamed through LDS in blocks of 64 rows: a row-major [64][dk] image (XOR-swizzled 16-B chunks)
// for operands contracted over dk, and a transposed [dk][64] image for operands contracted over the
// streamed index.
#include <initializer_list>

#include "common.h"

struct AttnArgs {
  const void *q, *k, *v, *o, *do_;
  void *out, *dq, *dk, *dv;
  const uint8_t* key_mask;
  float* lse;
  float* delta;
  int B, H, Tq, Tk;
  int64_t q_bs, q_ts, k_bs, k_ts, v_bs, v_ts, o_bs, o_ts;
  int causal;
  float scale;
  int vec;  // all strides/base pointers allow 16-byte row chunks
  int xmap; // 1: 1-D grid, the blocks of one (head, utterance) share an XCD (attn_block)
  // optional additive score bias (pre-scale): S = (q.k + bias[b,h,i,col]) * scale, col = j (+ Tq-1-i if rel_shift):
  // rel_shift turns a [.., i, 2T-1] relative-position term into the Transformer-XL shifted matrix by index
  // arithmetic (module/attention.py:209-215 materialises and gathers it).
  const float* bias;
  int bias_vec4;                     // rel_shift: the rows are long enough for unclamped 16-byte loads (set_bias)
  float* dbias; int dbias_h16;     // dbias_h16: the gradient tensor is 16-bit (r06: half the bytes of a 64 MB band tensor written once and read twice)
  int64_t bias_bs, bias_hs, bias_rs;
  int rel_shift;
};

__device__ __forceinline__ int64_t bias_index(const AttnArgs& p, int b, int h, int i, int j) {
  return (int64_t)b * p.bias_bs + (int64_t)h * p.bias_hs + (int64_t)i * p.bias_rs + j + (p.rel_shift ? (p.Tq - 1 - i) : 0);
}
// four consecutive columns of one bias row as ONE 16-byte load at 4-byte alignment (the relative-position column j - i + T - 1 is aligned for
// no row in particular; global_load_dwordx4 only needs dword alignment).  The lane's four keys lg*4 .. + 3 of a 16-key tile are consecutive
// columns; past the last key the load runs on into the row's padding / the next row of the SAME tensor (never past its end: the last row's
// columns end before the padding does) and the values are masked at use.  r06: 16 -> 4 loads per lane and key block.
typedef float otr_f32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
__device__ __forceinline__ void bias_load4(const float* row, int col, float (&o)[4]) {
  const otr_f32x4_a4 v = *reinterpret_cast<const otr_f32x4_a4*>(row + col);
  o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
}

// Which (block of queries / keys, head, utterance) a workgroup works on.  xmap = 0: the 3-D grid as launched.  xmap = 1: a 1-D grid
// in which the nx blocks of one (head, utterance) get workgroup ids that are EQUAL modulo 8 -- consecutive ids go to consecutive
// XCDs (observed placement, used for locality only), so the blocks that stream the same K / V (or Q / dO) rows meet in ONE L2
// instead of four.  Groups past H * B (the grid is padded to whole XCD rounds) return false: the workgroup leaves.
__host__ __device__ __forceinline__ bool attn_block_of(int lid, int nx, int H, int B, int& bx, int& h, int& b) {
  const int xcd = lid & 7, k = lid >> 3;
  bx = k % nx;
  const int g = (k / nx) * 8 + xcd;
  if (g >= H * B) return false;
  h = g % H; b = g / H;
  return true;
}
__device__ __forceinline__ bool attn_block(const AttnArgs& p, int nx, int& bx, int& h, int& b) {
  if (!p.xmap) { bx = blockIdx.x; h = blockIdx.y; b = blockIdx.z; return true; }
  return attn_block_of((int)blockIdx.x, nx, p.H, p.B, bx, h, b);
}

template <class CT, int DK> struct ACfg {
  static constexpr int CE = MMA<CT>::CE;
  static constexpr int KSTEP = MMA<CT>::KSTEP;
  static constexpr int TPC = MMA<CT>::TPC;
  static constexpr int DKP = (DK + KSTEP - 1) / KSTEP * KSTEP;  // head dim padded to the MFMA k-step
  static constexpr int NCH = DKP / CE;                          // 16-B chunks per row-major row
  static constexpr int KS = DKP / KSTEP;                        // k-steps over the head dim
  static constexpr int ROWB = NCH * 16;                         // bytes per row-major row
  static constexpr int DT = DK / 16;                            // 16-wide d tiles
  static constexpr int TSTRIDE = 64 * (int)sizeof(CT) + (sizeof(CT) == 2 ? 8 : 16);  // transposed row bytes
  static constexpr int NC = 64 / KSTEP;                         // contraction chunks per 64-row block
  static constexpr int RM_BYTES = 64 * ROWB;
  static constexpr int TR_BYTES = DK * TSTRIDE;
  static_assert(DK % 16 == 0, "head dim must be a multiple of 16");
};

// ---- row-major [64][DKP] tile: rows row0.. of a [T, ...] tensor (row stride ts elements) ----------
template <class CT, int DK>
__device__ __forceinline__ void load_tile_rm(unsigned char* lds, const CT* g, int64_t ts, int nvalid, bool vec, int tid) {
  using C = ACfg<CT, DK>;
  for (int id = tid; id < 64 * C::NCH; id += 256) {
    int row = id / C::NCH, c = id - row * C::NCH;
    uint4 val = make_uint4(0, 0, 0, 0);
    if (row < nvalid && c * C::CE < DK) {
      const CT* p = g + (int64_t)row * ts + c * C::CE;
      if (vec) {
        val = *reinterpret_cast<const uint4*>(p);
      } else {
        float f[C::CE];
#pragma unroll
        for (int e = 0; e < C::CE; ++e) f[e] = ElemIO<CT>::ld(p + e);
        val = MMA<CT>::pack(f);
      }
    }
    *reinterpret_cast<uint4*>(lds + row * C::ROWB + ((c ^ swz<C::NCH>(row)) << 4)) = val;
  }
}
template <class CT, int DK>
__device__ __forceinline__ uint4 read_rm(const unsigned char* lds, int row, int chunk) {
  using C = ACfg<CT, DK>;
  return *reinterpret_cast<const uint4*>(lds + row * C::ROWB + ((chunk ^ swz<C::NCH>(row)) << 4));
}

// ---- transposed [DK][64] tile: element (d, j) = g[j*ts + d] -------------------------------------
template <class CT, int DK>
__device__ __forceinline__ void load_tile_tr(unsigned char* lds, const CT* g, int64_t ts, int nvalid, bool vec, int tid) {
  using C = ACfg<CT, DK>;
  constexpr int DP = DK / 2;
  for (int id = tid; id < DP * 16; id += 256) {
    int dp = id % DP, kg = id / DP;
    float v0[4], v1[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int row = kg * 4 + j;
      v0[j] = 0.f; v1[j] = 0.f;
      if (row < nvalid) {
        const CT* p = g + (int64_t)row * ts + 2 * dp;
        if (vec) {
          if constexpr (sizeof(CT) == 4) {
            float2 t = *reinterpret_cast<const float2*>(p);
            v0[j] = t.x; v1[j] = t.y;
          } else {
            uint32_t t = *reinterpret_cast<const uint32_t*>(p);
            v0[j] = h2f_lo(t); v1[j] = h2f_hi(t);
          }
        } else {
          v0[j] = ElemIO<CT>::ld(p); v1[j] = ElemIO<CT>::ld(p + 1);
        }
      }
    }
    unsigned char* d0 = lds + (2 * dp) * C::TSTRIDE + kg * 4 * (int)sizeof(CT);
    unsigned char* d1 = d0 + C::TSTRIDE;
    if constexpr (sizeof(CT) == 2) {
      *reinterpret_cast<uint2*>(d0) = make_uint2(pack2bf(v0[0], v0[1]), pack2bf(v0[2], v0[3]));
      *reinterpret_cast<uint2*>(d1) = make_uint2(pack2bf(v1[0], v1[1]), pack2bf(v1[2], v1[3]));
    } else {
      *reinterpret_cast<float4*>(d0) = make_float4(v0[0], v0[1], v0[2], v0[3]);
      *reinterpret_cast<float4*>(d1) = make_float4(v1[0], v1[1], v1[2], v1[3]);
    }
  }
}
// operand chunk c (KSTEP streamed rows) for row `row` of the transposed tile; slot layout matches
// MMA<CT>::from_tiles: elements 0-3 <- streamed rows c*KSTEP + g*4.., elements 4-7 <- +16 (bf16)
template <class CT, int DK>
__device__ __forceinline__ uint4 read_tr(const unsigned char* lds, int row, int c, int g) {
  using C = ACfg<CT, DK>;
  const unsigned char* p = lds + row * C::TSTRIDE;
  if constexpr (sizeof(CT) == 2) {
    uint2 lo = *reinterpret_cast<const uint2*>(p + ((2 * c) * 16 + g * 4) * 2);
    uint2 hi = *reinterpret_cast<const uint2*>(p + ((2 * c + 1) * 16 + g * 4) * 2);
    return make_uint4(lo.x, lo.y, hi.x, hi.y);
  } else {
    return *reinterpret_cast<const uint4*>(p + (c * 16 + g * 4) * 4);
  }
}

// ---- two-phase versions of the tile loaders (16-byte aligned operands only): ld_* issues every global load of a
// tile back to back into registers, st_* writes the LDS image later.  The kernels issue the loads of block i+1
// before the MFMAs of block i, so the HBM/L2 latency of the streamed operand hides behind compute instead of being
// paid once per block between two barriers.  All loads are unconditional (rows clamped, invalid rows zeroed with a
// mask at store time): a load inside a branch gets its own vmcnt(0) from hipcc.
template <class CT, int DK, int NT = 256> struct RmRegs {      // NT = threads of the workgroup
  static constexpr int NU = (64 * ACfg<CT, DK>::NCH) / NT;
  static_assert((64 * ACfg<CT, DK>::NCH) % NT == 0, "row-major tile must divide evenly over the workgroup's threads");
  uint4 v[NU];
};
template <class CT, int DK, int NT>
__device__ __forceinline__ void ld_rm(RmRegs<CT, DK, NT>& r, const CT* g, int64_t ts, int nvalid, int tid) {
  using C = ACfg<CT, DK>;
#pragma unroll
  for (int u = 0; u < RmRegs<CT, DK, NT>::NU; ++u) {
    const int id = tid + NT * u, row = id / C::NCH, c = id - row * C::NCH;
    const int rr = min(row, nvalid - 1), cc = (c * C::CE < DK) ? c * C::CE : 0;
    r.v[u] = *reinterpret_cast<const uint4*>(g + (int64_t)rr * ts + cc);
  }
}
template <class CT, int DK, int NT>
__device__ __forceinline__ void st_rm(unsigned char* lds, const RmRegs<CT, DK, NT>& r, int nvalid, int tid) {
  using C = ACfg<CT, DK>;
#pragma unroll
  for (int u = 0; u < RmRegs<CT, DK, NT>::NU; ++u) {
    const int id = tid + NT * u, row = id / C::NCH, c = id - row * C::NCH;
    const uint32_t m = (uint32_t)0 - (uint32_t)(row < nvalid && c * C::CE < DK);
    uint4 q = r.v[u];
    q.x &= m; q.y &= m; q.z &= m; q.w &= m;
    *reinterpret_cast<uint4*>(lds + row * C::ROWB + ((c ^ swz<C::NCH>(row)) << 4)) = q;
  }
}
template <class CT, int DK, int NT = 256> struct TrRegs {
  static constexpr int DP = DK / 2;
  static constexpr int NI = (DP * 16 + NT - 1) / NT;
  uint2 w[NI][4];   // two adjacent head-dim elements of 4 consecutive streamed rows (bf16 uses .x only)
};
template <class CT, int DK, int NT>
__device__ __forceinline__ void ld_tr(TrRegs<CT, DK, NT>& r, const CT* g, int64_t ts, int nvalid, int tid) {
  constexpr int DP = DK / 2;
#pragma unroll
  for (int it = 0; it < TrRegs<CT, DK, NT>::NI; ++it) {
    const int id = min(tid + NT * it, DP * 16 - 1), dp = id % DP, kg = id / DP;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int rr = min(kg * 4 + j, nvalid - 1);
      const CT* p = g + (int64_t)rr * ts + 2 * dp;
      if constexpr (sizeof(CT) == 4) r.w[it][j] = *reinterpret_cast<const uint2*>(p);
      else r.w[it][j].x = *reinterpret_cast<const uint32_t*>(p);
    }
  }
}
template <class CT, int DK, int NT>
__device__ __forceinline__ void st_tr(unsigned char* lds, const TrRegs<CT, DK, NT>& r, int nvalid, int tid) {
  using C = ACfg<CT, DK>;
  constexpr int DP = DK / 2;
#pragma unroll
  for (int it = 0; it < TrRegs<CT, DK, NT>::NI; ++it) {
    const int id = tid + NT * it;
    if (id < DP * 16) {                 // LDS-only branch (DK = 16: half of the threads have no patch)
      const int dp = id % DP, kg = id / DP;
      uint32_t lo[4], hi[4];            // element 2dp / 2dp+1 of rows kg*4..+3, as raw bits
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const uint32_t m = (uint32_t)0 - (uint32_t)(kg * 4 + j < nvalid);
        if constexpr (sizeof(CT) == 4) { lo[j] = r.w[it][j].x & m; hi[j] = r.w[it][j].y & m; }
        else { lo[j] = (r.w[it][j].x & 0xffffu) & m; hi[j] = (r.w[it][j].x >> 16) & m; }
      }
      unsigned char* d0 = lds + (2 * dp) * C::TSTRIDE + kg * 4 * (int)sizeof(CT);
      unsigned char* d1 = d0 + C::TSTRIDE;
      if constexpr (sizeof(CT) == 2) {
        *reinterpret_cast<uint2*>(d0) = make_uint2(lo[0] | (lo[1] << 16), lo[2] | (lo[3] << 16));
        *reinterpret_cast<uint2*>(d1) = make_uint2(hi[0] | (hi[1] << 16), hi[2] | (hi[3] << 16));
      } else {
        *reinterpret_cast<uint4*>(d0) = make_uint4(lo[0], lo[1], lo[2], lo[3]);
        *reinterpret_cast<uint4*>(d1) = make_uint4(hi[0], hi[1], hi[2], hi[3]);
      }
    }
  }
}

// per-wave register operand: 16 rows (row = lane&15) x DKP, chunk ks*4+g per k-step
template <class CT, int DK>
__device__ __forceinline__ void load_reg_frags(uint4* fr, const CT* g, int64_t ts, int row, int nrows, bool vec, int lg) {
  using C = ACfg<CT, DK>;
#pragma unroll
  for (int ks = 0; ks < C::KS; ++ks) {
    int e0 = (ks * 4 + lg) * C::CE;
    uint4 val = make_uint4(0, 0, 0, 0);
    if (row < nrows && e0 < DK) {
      const CT* p = g + (int64_t)row * ts + e0;
      if (vec) {
        val = *reinterpret_cast<const uint4*>(p);
      } else {
        float f[C::CE];
#pragma unroll
        for (int e = 0; e < C::CE; ++e) f[e] = ElemIO<CT>::ld(p + e);
        val = MMA<CT>::pack(f);
      }
    }
    fr[ks] = val;
  }
}

// bf16 mode works in the base-2 domain: scores are pre-multiplied by scale*log2(e) so that every exponential is a bare
// v_exp_f32 (natural exp costs an extra multiply each, and these kernels are VALU bound: PMC showed ~2000 VALU
// instructions per wave against 64 MFMAs).  fp32 (parity) mode keeps expf.
template <class CT> struct ExpDom {
  static constexpr float K = sizeof(CT) == 4 ? 1.f : 1.4426950408889634f;          // log2(e) in bf16 mode
  static __device__ __forceinline__ float ex(float x) {
    if constexpr (sizeof(CT) == 4) return expf(x);
    else return __builtin_amdgcn_exp2f(x);
  }
};
template <class CT> __device__ __forceinline__ float attn_exp(float x) { return ExpDom<CT>::ex(x); }
// validity of the 64 streamed keys of a block as one wave-uniform bit mask (one byte load per lane instead of 16)
__device__ __forceinline__ uint64_t key_block_mask(const uint8_t* km, int kb, int Tk, int lane) {
  const int key = kb * 64 + lane;
  const bool v = key < Tk && (!km || km[min(key, Tk - 1)] != 0);
  return __ballot(v);
}

// store 4 consecutive head-dim elements
template <class CT> __device__ __forceinline__ void store4(CT* p, float a, float b, float c, float d, bool vec) {
  if constexpr (sizeof(CT) == 4) {
    if (vec) *reinterpret_cast<float4*>(p) = make_float4(a, b, c, d);
    else { p[0] = a; p[1] = b; p[2] = c; p[3] = d; }
  } else {
    if (vec) *reinterpret_cast<uint2*>(p) = make_uint2(pack2bf(a, b), pack2bf(c, d));
    else { p[0] = f2bf(a); p[1] = f2bf(b); p[2] = f2bf(c); p[3] = f2bf(d); }
  }
}

#define NEG_INF (-__builtin_huge_valf())

// ================================================================================================ forward
// NW waves per workgroup (16 queries each).  NW = 8 (aligned 16-bit operands, head dim 64): every streamed key / value block is
// staged once for eight waves instead of four (attn_bwd_kernel)
template <class CT, int DK, bool PIPE, int NW = 4> __global__ __launch_bounds__(64 * NW) void attn_fwd_kernel(AttnArgs p) {
  using C = ACfg<CT, DK>;
  constexpr int NT = 64 * NW;
  static_assert(NW == 4 || PIPE, "more than four waves: aligned operands only");
  __shared__ __attribute__((aligned(16))) unsigned char smem[C::RM_BYTES + C::TR_BYTES];
  unsigned char* sK = smem;
  unsigned char* sVt = smem + C::RM_BYTES;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, lr = lane & 15, lg = lane >> 4;
  int bx, h, b;
  if (!attn_block(p, (p.Tq + 16 * NW - 1) / (16 * NW), bx, h, b)) return;
  const int q0 = bx * (16 * NW) + wid * 16;
  const bool vec = p.vec != 0;
  const CT* Q = reinterpret_cast<const CT*>(p.q) + b * p.q_bs + h * DK;
  const CT* K = reinterpret_cast<const CT*>(p.k) + b * p.k_bs + h * DK;
  const CT* V = reinterpret_cast<const CT*>(p.v) + b * p.v_bs + h * DK;
  const uint8_t* km = p.key_mask ? p.key_mask + (int64_t)b * p.Tk : nullptr;

  uint4 qf[C::KS];
  load_reg_frags<CT, DK>(qf, Q, p.q_ts, q0 + lr, p.Tq, vec, lg);
  const int qrow = q0 + lr;

  f32x4 acc[C::DT];
#pragma unroll
  for (int i = 0; i < C::DT; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m = NEG_INF, l = 0.f;                       // running max / sum in the ExpDom<CT> domain
  const float sc2 = p.scale * ExpDom<CT>::K;

  const int nkb = (p.Tk + 63) / 64;
  RmRegs<CT, DK, NT> kreg;
  TrRegs<CT, DK, NT> vreg;
  if constexpr (PIPE) {
    ld_rm<CT, DK>(kreg, K, p.k_ts, min(64, p.Tk), tid);
    ld_tr<CT, DK>(vreg, V, p.v_ts, min(64, p.Tk), tid);
  }
  for (int kb = 0; kb < nkb; ++kb) {
    __syncthreads();
    int nvalid = min(64, p.Tk - kb * 64);
    if constexpr (PIPE) {
      st_rm<CT, DK>(sK, kreg, nvalid, tid);
      st_tr<CT, DK>(sVt, vreg, nvalid, tid);
    } else {
      load_tile_rm<CT, DK>(sK, K + (int64_t)kb * 64 * p.k_ts, p.k_ts, nvalid, vec, tid);
      load_tile_tr<CT, DK>(sVt, V + (int64_t)kb * 64 * p.v_ts, p.v_ts, nvalid, vec, tid);
    }
    __syncthreads();
    if constexpr (PIPE) {   // next block's loads fly during this block's MFMAs (last round: redundant re-load, dropped)
      const int kn = min(kb + 1, nkb - 1), nv2 = min(64, p.Tk - kn * 64);
      ld_rm<CT, DK>(kreg, K + (int64_t)kn * 64 * p.k_ts, p.k_ts, nv2, tid);
      ld_tr<CT, DK>(vreg, V + (int64_t)kn * 64 * p.v_ts, p.v_ts, nv2, tid);
      __builtin_amdgcn_sched_barrier(0);
    }

    // score bias (relative-position term): the block's 16 values per lane go out BEFORE the MFMAs, unconditionally (indices clamped
    // into the tensor, masked at use) -- read at the point of use, each behind its own predicate, they were 16 exposed round trips
    // per key block (attention forward 38 us, backward 139 us per Conformer block; r05)
    float bv[4][4];
    if (p.bias) {
      const int qc = min(qrow, p.Tq - 1);
      const float* brow = p.bias + bias_index(p, b, h, qc, 0);
      if (p.rel_shift && p.bias_vec4) {
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) bias_load4(brow, kb * 64 + kt * 16 + lg * 4, bv[kt]);
      } else {
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) bv[kt][r] = brow[min(kb * 64 + kt * 16 + lg * 4 + r, p.Tk - 1)];
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    f32x4 st[4];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      st[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < C::KS; ++ks) MMA<CT>::mma(st[kt], read_rm<CT, DK>(sK, kt * 16 + lr, ks * 4 + lg), qf[ks]);
    }
    float bm = NEG_INF;
    const uint64_t kmask = key_block_mask(km, kb, p.Tk, lane);
    if (kmask == ~0ull && !p.causal && !p.bias) {       // whole block valid: no per-element predicate at all
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          st[kt][r] *= sc2;
          bm = fmaxf(bm, st[kt][r]);
        }
    } else {
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int kl = kt * 16 + lg * 4 + r, key = kb * 64 + kl;
          bool ok = ((kmask >> kl) & 1) && (!p.causal || key <= qrow);
          float raw = st[kt][r];
          if (p.bias && ok && qrow < p.Tq) raw += bv[kt][r];
          float s = ok ? raw * sc2 : NEG_INF;
          st[kt][r] = s;
          bm = fmaxf(bm, s);
        }
    }
    bm = fmaxf(bm, __shfl_xor(bm, 16));
    bm = fmaxf(bm, __shfl_xor(bm, 32));
    float m_new = fmaxf(m, bm);
    float m_safe = (m_new == NEG_INF) ? 0.f : m_new;
    float alpha = attn_exp<CT>(m - m_safe);
    float rs = 0.f;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float e = attn_exp<CT>(st[kt][r] - m_safe);
        st[kt][r] = e;
        rs += e;
      }
    rs += __shfl_xor(rs, 16);
    rs += __shfl_xor(rs, 32);
    l = l * alpha + rs;
    m = m_new;
#pragma unroll
    for (int i = 0; i < C::DT; ++i) acc[i] *= alpha;
#pragma unroll
    for (int c = 0; c < C::NC; ++c) {
      uint4 pf = MMA<CT>::from_tiles(&st[c * C::TPC]);
#pragma unroll
      for (int dt = 0; dt < C::DT; ++dt) MMA<CT>::mma(acc[dt], read_tr<CT, DK>(sVt, dt * 16 + lr, c, lg), pf);
    }
  }

  if (qrow < p.Tq) {
    float inv = l > 0.f ? 1.f / l : 0.f;
    CT* O = reinterpret_cast<CT*>(p.out) + b * p.o_bs + (int64_t)qrow * p.o_ts + h * DK;
#pragma unroll
    for (int dt = 0; dt < C::DT; ++dt)
      store4<CT>(O + dt * 16 + lg * 4, acc[dt][0] * inv, acc[dt][1] * inv, acc[dt][2] * inv, acc[dt][3] * inv, vec);
    if (lg == 0) p.lse[((int64_t)b * p.H + h) * p.Tq + qrow] = (l > 0.f) ? m * (1.f / ExpDom<CT>::K) + logf(l) : NEG_INF;
  }
}

// ================================================================================================ dK, dV
// wave owns 16 keys (columns); streams 64-query blocks.  OWN_DELTA: delta = rowsum(dO * O) of every query block is formed here,
// from the dO rows on their way to LDS and the matching O rows (two more 16-byte loads per thread and block), so that this body
// does not depend on the dQ body's output and both can share ONE launch (attn_bwd_kernel).
template <class CT, int DK> struct BwdSmem {
  using C = ACfg<CT, DK>;
  static constexpr int DKDV = 2 * C::RM_BYTES + 2 * C::TR_BYTES + 512;
  static constexpr int DQ = 2 * C::RM_BYTES + C::TR_BYTES;
  static constexpr int BOTH = DKDV > DQ ? DKDV : DQ;
};
// partial dot product of two 16-byte chunks of CT
template <class CT> __device__ __forceinline__ float chunk_dot(const uint4& a, const uint4& b) {
  const uint32_t aw[4] = {a.x, a.y, a.z, a.w}, bw[4] = {b.x, b.y, b.z, b.w};
  float acc = 0.f;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    if constexpr (sizeof(CT) == 4) {
      acc += __uint_as_float(aw[e]) * __uint_as_float(bw[e]);
    } else {
      acc += h2f_lo(aw[e]) * h2f_lo(bw[e]);
      acc += h2f_hi(aw[e]) * h2f_hi(bw[e]);
    }
  }
  return acc;
}
template <class CT, int DK, bool PIPE, bool OWN_DELTA, int NW = 4>     // NW waves: the workgroup owns 16 NW keys
__device__ __forceinline__ void attn_bwd_dkdv_body(const AttnArgs& p, const int bx, const int h, const int b, unsigned char* smem) {
  using C = ACfg<CT, DK>;
  static_assert(!OWN_DELTA || (PIPE && (C::NCH & (C::NCH - 1)) == 0 && C::NCH <= 64), "own delta: aligned rows, 2^n chunks per row");
  unsigned char* sQ = smem;
  unsigned char* sdO = smem + C::RM_BYTES;
  unsigned char* sQt = smem + 2 * C::RM_BYTES;
  unsigned char* sdOt = sQt + C::TR_BYTES;
  float* sLse = reinterpret_cast<float*>(sdOt + C::TR_BYTES);
  float* sDel = sLse + 64;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, lr = lane & 15, lg = lane >> 4;
  constexpr int NT = 64 * NW;
  static_assert(NW == 4 || PIPE, "more than four waves: aligned operands only");
  const int k0 = bx * (16 * NW) + wid * 16;
  const bool vec = p.vec != 0;
  const float sc2 = p.scale * ExpDom<CT>::K;
  const CT* Q = reinterpret_cast<const CT*>(p.q) + b * p.q_bs + h * DK;
  const CT* K = reinterpret_cast<const CT*>(p.k) + b * p.k_bs + h * DK;
  const CT* V = reinterpret_cast<const CT*>(p.v) + b * p.v_bs + h * DK;
  const CT* dO = reinterpret_cast<const CT*>(p.do_) + b * p.o_bs + h * DK;
  const CT* Og = reinterpret_cast<const CT*>(p.o) + b * p.o_bs + h * DK;
  const float* lse = p.lse + ((int64_t)b * p.H + h) * p.Tq;
  const float* del = p.delta + ((int64_t)b * p.H + h) * p.Tq;
  const int key = k0 + lr;
  bool key_ok = key < p.Tk && (!p.key_mask || p.key_mask[(int64_t)b * p.Tk + key]);

  uint4 kf[C::KS], vf[C::KS];
  load_reg_frags<CT, DK>(kf, K, p.k_ts, key, p.Tk, vec, lg);
  load_reg_frags<CT, DK>(vf, V, p.v_ts, key, p.Tk, vec, lg);

  f32x4 dk_acc[C::DT], dv_acc[C::DT];
#pragma unroll
  for (int i = 0; i < C::DT; ++i) { dk_acc[i] = f32x4{0.f, 0.f, 0.f, 0.f}; dv_acc[i] = f32x4{0.f, 0.f, 0.f, 0.f}; }

  const int nqb = (p.Tq + 63) / 64;
  RmRegs<CT, DK, NT> qreg, doreg, oreg;
  TrRegs<CT, DK, NT> qtreg, dotreg;
  float lreg = 0.f, dreg = 0.f;
  if constexpr (PIPE) {
    const int nv0 = min(64, p.Tq);
    ld_rm<CT, DK>(qreg, Q, p.q_ts, nv0, tid);
    ld_rm<CT, DK>(doreg, dO, p.o_ts, nv0, tid);
    if constexpr (OWN_DELTA) ld_rm<CT, DK>(oreg, Og, p.o_ts, nv0, tid);
    ld_tr<CT, DK>(qtreg, Q, p.q_ts, nv0, tid);
    ld_tr<CT, DK>(dotreg, dO, p.o_ts, nv0, tid);
    lreg = lse[min(tid & 63, p.Tq - 1)];
    if constexpr (!OWN_DELTA) dreg = del[min(tid & 63, p.Tq - 1)];
  }
  for (int qb = 0; qb < nqb; ++qb) {
    __syncthreads();
    int nvalid = min(64, p.Tq - qb * 64);
    if constexpr (PIPE) {
      st_rm<CT, DK>(sQ, qreg, nvalid, tid);
      st_rm<CT, DK>(sdO, doreg, nvalid, tid);
      st_tr<CT, DK>(sQt, qtreg, nvalid, tid);
      st_tr<CT, DK>(sdOt, dotreg, nvalid, tid);
      if constexpr (OWN_DELTA) {             // thread (u, tid) holds chunk id % NCH of row id / NCH: the row's chunks sit in NCH neighbouring lanes
#pragma unroll
        for (int u = 0; u < RmRegs<CT, DK, NT>::NU; ++u) {
          const int id = tid + NT * u, row = id / C::NCH, c = id - row * C::NCH;
          float part = (c * C::CE < DK) ? chunk_dot<CT>(doreg.v[u], oreg.v[u]) : 0.f;
#pragma unroll
          for (int sh = 1; sh < C::NCH; sh <<= 1) part += __shfl_xor(part, sh);
          if (c == 0) sDel[row] = (row < nvalid) ? part : 0.f;
        }
        if (tid < 64) sLse[tid] = (tid < nvalid) ? lreg * ExpDom<CT>::K : 0.f;
      } else if (tid < 64) {
        sLse[tid] = (tid < nvalid) ? lreg * ExpDom<CT>::K : 0.f;
        sDel[tid] = (tid < nvalid) ? dreg : 0.f;
      }
    } else {
      load_tile_rm<CT, DK>(sQ, Q + (int64_t)qb * 64 * p.q_ts, p.q_ts, nvalid, vec, tid);
      load_tile_rm<CT, DK>(sdO, dO + (int64_t)qb * 64 * p.o_ts, p.o_ts, nvalid, vec, tid);
      load_tile_tr<CT, DK>(sQt, Q + (int64_t)qb * 64 * p.q_ts, p.q_ts, nvalid, vec, tid);
      load_tile_tr<CT, DK>(sdOt, dO + (int64_t)qb * 64 * p.o_ts, p.o_ts, nvalid, vec, tid);
      if (tid < 64) {
        sLse[tid] = (tid < nvalid) ? lse[qb * 64 + tid] * ExpDom<CT>::K : 0.f;
        sDel[tid] = (tid < nvalid) ? del[qb * 64 + tid] : 0.f;
      }
    }
    __syncthreads();
    if constexpr (PIPE) {
      const int qn = min(qb + 1, nqb - 1), nv2 = min(64, p.Tq - qn * 64);
      ld_rm<CT, DK>(qreg, Q + (int64_t)qn * 64 * p.q_ts, p.q_ts, nv2, tid);
      ld_rm<CT, DK>(doreg, dO + (int64_t)qn * 64 * p.o_ts, p.o_ts, nv2, tid);
      if constexpr (OWN_DELTA) ld_rm<CT, DK>(oreg, Og + (int64_t)qn * 64 * p.o_ts, p.o_ts, nv2, tid);
      ld_tr<CT, DK>(qtreg, Q + (int64_t)qn * 64 * p.q_ts, p.q_ts, nv2, tid);
      ld_tr<CT, DK>(dotreg, dO + (int64_t)qn * 64 * p.o_ts, p.o_ts, nv2, tid);
      lreg = lse[min(qn * 64 + (tid & 63), p.Tq - 1)];
      if constexpr (!OWN_DELTA) dreg = del[min(qn * 64 + (tid & 63), p.Tq - 1)];
      __builtin_amdgcn_sched_barrier(0);
    }

    float bv[4][4];      // score bias of the block, loaded ahead of the MFMAs (see attn_fwd_kernel)
    if (p.bias) {
      const int kc = min(key, p.Tk - 1);
#pragma unroll
      for (int qt = 0; qt < 4; ++qt)
#pragma unroll
        for (int r = 0; r < 4; ++r) bv[qt][r] = p.bias[bias_index(p, b, h, min(qb * 64 + qt * 16 + lg * 4 + r, p.Tq - 1), kc)];
      __builtin_amdgcn_sched_barrier(0);
    }
    f32x4 pt[4], ds[4];  // tiles over q (rows), cols = keys
    // whole query block valid with finite lse, no causal mask, no bias: no per-element predicate (masked KEYS only
    // pollute their own dK/dV columns, which are zeroed at the store)
    const bool fastblk = !p.causal && !p.bias && nvalid == 64 && __ballot(sLse[lane] == NEG_INF) == 0;
#pragma unroll
    for (int qt = 0; qt < 4; ++qt) {
      f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f}, dp = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < C::KS; ++ks) {
        MMA<CT>::mma(s, read_rm<CT, DK>(sQ, qt * 16 + lr, ks * 4 + lg), kf[ks]);
        MMA<CT>::mma(dp, read_rm<CT, DK>(sdO, qt * 16 + lr, ks * 4 + lg), vf[ks]);
      }
      if (fastblk) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int ql = qt * 16 + lg * 4 + r;
          const float pe = attn_exp<CT>(s[r] * sc2 - sLse[ql]);
          pt[qt][r] = pe;
          ds[qt][r] = pe * (dp[r] - sDel[ql]) * p.scale;
        }
        continue;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int ql = qt * 16 + lg * 4 + r, qg = qb * 64 + ql;
        float ls = sLse[ql];                               // pre-multiplied by ExpDom<CT>::K when it was staged
        bool ok = key_ok && qg < p.Tq && (!p.causal || key <= qg) && ls != NEG_INF;
        float raw = s[r];
        if (p.bias && ok) raw += bv[qt][r];
        float pe = ok ? attn_exp<CT>(raw * sc2 - ls) : 0.f;
        pt[qt][r] = pe;
        float dsv = pe * (dp[r] - sDel[ql]) * p.scale;
        ds[qt][r] = dsv;
        // unique (i, col): plain store.  EVERY in-range (query, key) pair is written, masked ones with 0: a caller that keeps the
        // gradient tensor across steps (ops.RelPosAttentionFn: zeroed once, not per step) never finds a stale entry in the band
        if (p.dbias && key < p.Tk && qg < p.Tq) {
          const float gv = ok ? dsv : 0.f;
          if (p.dbias_h16) reinterpret_cast<bf16_t*>(p.dbias)[bias_index(p, b, h, qg, key)] = f2bf(gv);
          else p.dbias[bias_index(p, b, h, qg, key)] = gv;
        }
      }
    }
#pragma unroll
    for (int c = 0; c < C::NC; ++c) {
      uint4 pf = MMA<CT>::from_tiles(&pt[c * C::TPC]);
      uint4 dsf = MMA<CT>::from_tiles(&ds[c * C::TPC]);
#pragma unroll
      for (int dt = 0; dt < C::DT; ++dt) {
        MMA<CT>::mma(dv_acc[dt], read_tr<CT, DK>(sdOt, dt * 16 + lr, c, lg), pf);
        MMA<CT>::mma(dk_acc[dt], read_tr<CT, DK>(sQt, dt * 16 + lr, c, lg), dsf);
      }
    }
  }
  if (key < p.Tk) {
    CT* dK = reinterpret_cast<CT*>(p.dk) + b * p.k_bs + (int64_t)key * p.k_ts + h * DK;
    CT* dV = reinterpret_cast<CT*>(p.dv) + b * p.v_bs + (int64_t)key * p.v_ts + h * DK;
    // masked key: zero gradient (the fast path did not predicate it)
#pragma unroll
    for (int dt = 0; dt < C::DT; ++dt) {
      dk_acc[dt] = key_ok ? dk_acc[dt] : f32x4{0.f, 0.f, 0.f, 0.f};
      dv_acc[dt] = key_ok ? dv_acc[dt] : f32x4{0.f, 0.f, 0.f, 0.f};
      store4<CT>(dK + dt * 16 + lg * 4, dk_acc[dt][0], dk_acc[dt][1], dk_acc[dt][2], dk_acc[dt][3], vec);
      store4<CT>(dV + dt * 16 + lg * 4, dv_acc[dt][0], dv_acc[dt][1], dv_acc[dt][2], dv_acc[dt][3], vec);
    }
  }
}
template <class CT, int DK, bool PIPE> __global__ __launch_bounds__(256) void attn_bwd_dkdv_kernel(AttnArgs p) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[BwdSmem<CT, DK>::DKDV];
  attn_bwd_dkdv_body<CT, DK, PIPE, false>(p, (int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z, smem);
}

// ================================================================================================ dQ
// wave owns 16 queries (columns); streams 64-key blocks.
template <class CT, int DK, bool PIPE, int NW = 4>     // NW waves: the workgroup owns 16 NW queries
__device__ __forceinline__ void attn_bwd_dq_body(const AttnArgs& p, const int bx, const int h, const int b, unsigned char* smem) {
  using C = ACfg<CT, DK>;
  unsigned char* sK = smem;
  unsigned char* sV = smem + C::RM_BYTES;
  unsigned char* sKt = smem + 2 * C::RM_BYTES;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, lr = lane & 15, lg = lane >> 4;
  constexpr int NT = 64 * NW;
  static_assert(NW == 4 || PIPE, "more than four waves: aligned operands only");
  const int q0 = bx * (16 * NW) + wid * 16;
  const bool vec = p.vec != 0;
  const float sc2 = p.scale * ExpDom<CT>::K;
  const CT* Q = reinterpret_cast<const CT*>(p.q) + b * p.q_bs + h * DK;
  const CT* K = reinterpret_cast<const CT*>(p.k) + b * p.k_bs + h * DK;
  const CT* V = reinterpret_cast<const CT*>(p.v) + b * p.v_bs + h * DK;
  const CT* dO = reinterpret_cast<const CT*>(p.do_) + b * p.o_bs + h * DK;
  const uint8_t* km = p.key_mask ? p.key_mask + (int64_t)b * p.Tk : nullptr;
  const int qrow = q0 + lr;
  const bool q_ok = qrow < p.Tq;
  const float ls = q_ok ? p.lse[((int64_t)b * p.H + h) * p.Tq + qrow] * ExpDom<CT>::K : NEG_INF;
  const bool row_live = q_ok && ls != NEG_INF;
  const float lsf = row_live ? ls : 0.f;                // finite stand-in for the unpredicated fast path

  uint4 qf[C::KS], dof[C::KS];
  load_reg_frags<CT, DK>(qf, Q, p.q_ts, qrow, p.Tq, vec, lg);
  load_reg_frags<CT, DK>(dof, dO, p.o_ts, qrow, p.Tq, vec, lg);
  // delta = rowsum(dO * O) of this wave's 16 queries, from the dO fragments already in registers (each lane group holds
  // a quarter of the row); written out for the dK/dV kernel, which runs after this one (no separate delta launch)
  float dl;
  {
    uint4 of[C::KS];
    load_reg_frags<CT, DK>(of, reinterpret_cast<const CT*>(p.o) + b * p.o_bs + h * DK, p.o_ts, qrow, p.Tq, vec, lg);
    float acc_d = 0.f;
#pragma unroll
    for (int ks = 0; ks < C::KS; ++ks) {
      const uint32_t aw[4] = {dof[ks].x, dof[ks].y, dof[ks].z, dof[ks].w}, bw[4] = {of[ks].x, of[ks].y, of[ks].z, of[ks].w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if constexpr (sizeof(CT) == 4) {
          acc_d += __uint_as_float(aw[e]) * __uint_as_float(bw[e]);
        } else {
          acc_d += h2f_lo(aw[e]) * h2f_lo(bw[e]);
          acc_d += h2f_hi(aw[e]) * h2f_hi(bw[e]);
        }
      }
    }
    acc_d += __shfl_xor(acc_d, 16);
    acc_d += __shfl_xor(acc_d, 32);
    dl = q_ok ? acc_d : 0.f;
    if (lg == 0 && q_ok) p.delta[((int64_t)b * p.H + h) * p.Tq + qrow] = dl;
  }

  f32x4 acc[C::DT];
#pragma unroll
  for (int i = 0; i < C::DT; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nkb = (p.Tk + 63) / 64;
  RmRegs<CT, DK, NT> kreg, vreg;
  TrRegs<CT, DK, NT> ktreg;
  if constexpr (PIPE) {
    const int nv0 = min(64, p.Tk);
    ld_rm<CT, DK>(kreg, K, p.k_ts, nv0, tid);
    ld_rm<CT, DK>(vreg, V, p.v_ts, nv0, tid);
    ld_tr<CT, DK>(ktreg, K, p.k_ts, nv0, tid);
  }
  for (int kb = 0; kb < nkb; ++kb) {
    __syncthreads();
    int nvalid = min(64, p.Tk - kb * 64);
    if constexpr (PIPE) {
      st_rm<CT, DK>(sK, kreg, nvalid, tid);
      st_rm<CT, DK>(sV, vreg, nvalid, tid);
      st_tr<CT, DK>(sKt, ktreg, nvalid, tid);
    } else {
      load_tile_rm<CT, DK>(sK, K + (int64_t)kb * 64 * p.k_ts, p.k_ts, nvalid, vec, tid);
      load_tile_rm<CT, DK>(sV, V + (int64_t)kb * 64 * p.v_ts, p.v_ts, nvalid, vec, tid);
      load_tile_tr<CT, DK>(sKt, K + (int64_t)kb * 64 * p.k_ts, p.k_ts, nvalid, vec, tid);
    }
    __syncthreads();
    if constexpr (PIPE) {
      const int kn = min(kb + 1, nkb - 1), nv2 = min(64, p.Tk - kn * 64);
      ld_rm<CT, DK>(kreg, K + (int64_t)kn * 64 * p.k_ts, p.k_ts, nv2, tid);
      ld_rm<CT, DK>(vreg, V + (int64_t)kn * 64 * p.v_ts, p.v_ts, nv2, tid);
      ld_tr<CT, DK>(ktreg, K + (int64_t)kn * 64 * p.k_ts, p.k_ts, nv2, tid);
      __builtin_amdgcn_sched_barrier(0);
    }

    const uint64_t kmask = key_block_mask(km, kb, p.Tk, lane);
    float bv[4][4];      // score bias of the block, loaded ahead of the MFMAs (see attn_fwd_kernel)
    if (p.bias) {
      const float* brow = p.bias + bias_index(p, b, h, min(qrow, p.Tq - 1), 0);
      if (p.rel_shift && p.bias_vec4) {
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) bias_load4(brow, kb * 64 + kt * 16 + lg * 4, bv[kt]);
      } else {
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) bv[kt][r] = brow[min(kb * 64 + kt * 16 + lg * 4 + r, p.Tk - 1)];
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    f32x4 ds[4];  // tiles over keys (rows), cols = queries
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f}, dp = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < C::KS; ++ks) {
        MMA<CT>::mma(s, read_rm<CT, DK>(sK, kt * 16 + lr, ks * 4 + lg), qf[ks]);
        MMA<CT>::mma(dp, read_rm<CT, DK>(sV, kt * 16 + lr, ks * 4 + lg), dof[ks]);
      }
      if (kmask == ~0ull && !p.causal && !p.bias) {    // (rows without a finite lse are zeroed at the store)
#pragma unroll
        for (int r = 0; r < 4; ++r) ds[kt][r] = attn_exp<CT>(s[r] * sc2 - lsf) * (dp[r] - dl) * p.scale;
        continue;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int kl = kt * 16 + lg * 4 + r, key = kb * 64 + kl;
        bool ok = q_ok && ls != NEG_INF && ((kmask >> kl) & 1) && (!p.causal || key <= qrow);
        float raw = s[r];
        if (p.bias && ok) raw += bv[kt][r];
        float pe = ok ? attn_exp<CT>(raw * sc2 - ls) : 0.f;
        ds[kt][r] = pe * (dp[r] - dl) * p.scale;
      }
    }
#pragma unroll
    for (int c = 0; c < C::NC; ++c) {
      uint4 dsf = MMA<CT>::from_tiles(&ds[c * C::TPC]);
#pragma unroll
      for (int dt = 0; dt < C::DT; ++dt) MMA<CT>::mma(acc[dt], read_tr<CT, DK>(sKt, dt * 16 + lr, c, lg), dsf);
    }
  }
  if (q_ok) {
    CT* dQ = reinterpret_cast<CT*>(p.dq) + b * p.q_bs + (int64_t)qrow * p.q_ts + h * DK;
#pragma unroll
    for (int dt = 0; dt < C::DT; ++dt) {
      if (!row_live) acc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
      store4<CT>(dQ + dt * 16 + lg * 4, acc[dt][0], acc[dt][1], acc[dt][2], acc[dt][3], vec);
    }
  }
}
template <class CT, int DK, bool PIPE> __global__ __launch_bounds__(256) void attn_bwd_dq_kernel(AttnArgs p) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[BwdSmem<CT, DK>::DQ];
  attn_bwd_dq_body<CT, DK, PIPE>(p, (int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z, smem);
}
// dQ and dK/dV in ONE launch: workgroups 0 .. nqb-1 of a (head, utterance) own 64 queries each, the rest 64 keys each.  The two
// halves share nothing but the launch -- no ordering between them (the dK/dV half forms its own delta) -- so a CU holds twice
// the waves to hide latency behind and the step has one dependent launch less per attention (24 per step at the benchmark).
// NW = 8: 128 queries / keys per workgroup -- every streamed 64-row block is staged ONCE for eight waves instead of four, which
// halves the L2 -> LDS staging traffic of the launch (five tile images per block and workgroup; at 32 x 249 frames that traffic
// was ~0.6 MB per CU per launch, a quarter of the launch at the CU's ingest rate)
template <class CT, int DK, int NW> __global__ __launch_bounds__(64 * NW, 2) void attn_bwd_kernel(AttnArgs p, int nqb) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[BwdSmem<CT, DK>::BOTH];
  int bx, h, b;
  if (!attn_block(p, nqb + (p.Tk + 16 * NW - 1) / (16 * NW), bx, h, b)) return;
  if (bx < nqb) attn_bwd_dq_body<CT, DK, true, NW>(p, bx, h, b, smem);
  else attn_bwd_dkdv_body<CT, DK, true, true, NW>(p, bx - nqb, h, b, smem);
}

// ================================================================================================ host side
static int32_t fill_args(const otr_attn_desc_t* d, AttnArgs& a) {
  OTR_REQUIRE(d != nullptr, "attention: null descriptor");
  OTR_REQUIRE(d->B > 0 && d->H > 0 && d->Tq > 0 && d->Tk > 0, "attention: bad shape B=%d H=%d Tq=%d Tk=%d", d->B, d->H,
              d->Tq, d->Tk);
  OTR_REQUIRE(d->dtype == OTR_F32 || d->dtype == OTR_H16, "attention: bad dtype %d", d->dtype);
  OTR_REQUIRE(d->dk == 16 || d->dk == 32 || d->dk == 64 || d->dk == 96 || d->dk == 128,
              "attention: head dim %d not built (16/32/64/96/128)", d->dk);
  a.B = d->B; a.H = d->H; a.Tq = d->Tq; a.Tk = d->Tk;
  a.q_bs = d->q_bs; a.q_ts = d->q_ts; a.k_bs = d->k_bs; a.k_ts = d->k_ts;
  a.v_bs = d->v_bs; a.v_ts = d->v_ts; a.o_bs = d->o_bs; a.o_ts = d->o_ts;
  a.causal = d->causal; a.scale = d->scale;
  return 0;
}
extern int g_otr_attn_waves8;      // api.hip (otr_debug_set(20, v)): 8-wave workgroups (128 queries / keys) for 16-bit operands, head dim 64
extern int g_otr_attn_xmap;        // api.hip (otr_debug_set(16, v)): XCD-aware workgroup mapping of the attention launches
static dim3 attn_grid(AttnArgs& a, const otr_attn_desc_t* d, int nx) {
  a.xmap = g_otr_attn_xmap;
  return a.xmap ? dim3((unsigned)(8 * nx * ((d->H * d->B + 7) / 8))) : dim3((unsigned)nx, d->H, d->B);
}
static int vec_ok(const otr_attn_desc_t* d, std::initializer_list<const void*> ptrs) {
  int ce = d->dtype == OTR_F32 ? 4 : 8;
  int64_t strides[] = {d->q_bs, d->q_ts, d->k_bs, d->k_ts, d->v_bs, d->v_ts, d->o_bs, d->o_ts};
  for (int64_t s : strides)
    if (s % ce) return 0;
  if (d->dk % ce) return 0;
  for (const void* p : ptrs)
    if (p && ((uintptr_t)p % 16)) return 0;
  return 1;
}

#define DK_SWITCH_P(CTYPE, KERNEL, GRID, PIPE)                                                            \
  switch (d->dk) {                                                                                        \
    case 16: hipLaunchKernelGGL((KERNEL<CTYPE, 16, PIPE>), GRID, dim3(256), 0, s, a); break;              \
    case 32: hipLaunchKernelGGL((KERNEL<CTYPE, 32, PIPE>), GRID, dim3(256), 0, s, a); break;              \
    case 64: hipLaunchKernelGGL((KERNEL<CTYPE, 64, PIPE>), GRID, dim3(256), 0, s, a); break;              \
    case 96: hipLaunchKernelGGL((KERNEL<CTYPE, 96, PIPE>), GRID, dim3(256), 0, s, a); break;              \
    default: hipLaunchKernelGGL((KERNEL<CTYPE, 128, PIPE>), GRID, dim3(256), 0, s, a); break;             \
  }
// aligned operands take the software-pipelined (register-prefetch) instantiation
#define DK_SWITCH(CTYPE, KERNEL, GRID)                    \
  if (a.vec) { DK_SWITCH_P(CTYPE, KERNEL, GRID, true) }   \
  else { DK_SWITCH_P(CTYPE, KERNEL, GRID, false) }

extern "C" int32_t otr_attention_fwd(const otr_attn_desc_t* d, const void* q, const void* k, const void* v,
                                     const uint8_t* key_mask, void* o, float* lse, void* stream) {
  AttnArgs a{};
  if (int32_t e = fill_args(d, a)) return e;
  OTR_REQUIRE(q && k && v && o && lse, "attention_fwd: null pointer");
  a.q = q; a.k = k; a.v = v; a.out = o; a.lse = lse; a.key_mask = key_mask;
  a.vec = vec_ok(d, {q, k, v, o});
  hipStream_t s = (hipStream_t)stream;
  if (g_otr_attn_waves8 && a.vec && d->dtype == OTR_H16 && d->dk == 64) {
    hipLaunchKernelGGL((attn_fwd_kernel<bf16_t, 64, true, 8>), attn_grid(a, d, (d->Tq + 127) / 128), dim3(512), 0, s, a);
    return otr_check_launch("attention_fwd");
  }
  dim3 grid = attn_grid(a, d, (d->Tq + 63) / 64);
  if (d->dtype == OTR_H16) { DK_SWITCH(bf16_t, attn_fwd_kernel, grid) } else { DK_SWITCH(float, attn_fwd_kernel, grid) }
  return otr_check_launch("attention_fwd");
}

extern int g_otr_bias_vec4;
static void set_bias(AttnArgs& a, const float* bias, float* dbias, int64_t bs, int64_t hs, int64_t rs, int rel_shift) {
  a.bias = bias; a.dbias = dbias; a.bias_bs = bs; a.bias_hs = hs; a.bias_rs = rs; a.rel_shift = rel_shift;
  // 16-byte loads run up to key 64 ceil(Tk / 64) - 1 of every row: allowed when that stays inside what the strides say is there
  // (every row of a relative-position tensor has its 2T - 1 columns; the address is linear in h and i, so the corners decide)
  a.bias_vec4 = 0;
  if (rel_shift && g_otr_bias_vec4 && bs >= 0 && hs >= 0 && rs >= 0) {
    const int64_t T = a.Tq, k64 = ((int64_t)a.Tk + 63) / 64 * 64, H = a.H;
    int64_t extent = 0, reach = 0;
    for (int64_t h : {(int64_t)0, H - 1})
      for (int64_t i : {(int64_t)0, T - 1}) {
        extent = std::max(extent, h * hs + i * rs + 2 * T - 1);
        reach = std::max(reach, h * hs + i * rs + (k64 - 1) + (T - 1 - i) + 1);
      }
    a.bias_vec4 = reach <= extent;
  }
}

extern "C" int32_t otr_attention_bias_fwd(const otr_attn_desc_t* d, const void* q, const void* k, const void* v,
                                          const uint8_t* key_mask, const float* bias, int64_t bias_bs, int64_t bias_hs,
                                          int64_t bias_rs, int32_t rel_shift, void* o, float* lse, void* stream) {
  AttnArgs a{};
  if (int32_t e = fill_args(d, a)) return e;
  OTR_REQUIRE(q && k && v && o && lse && bias, "attention_bias_fwd: null pointer");
  OTR_REQUIRE(!rel_shift || d->Tq == d->Tk, "attention_bias_fwd: rel_shift needs Tq == Tk");
  a.q = q; a.k = k; a.v = v; a.out = o; a.lse = lse; a.key_mask = key_mask;
  a.vec = vec_ok(d, {q, k, v, o});
  set_bias(a, bias, nullptr, bias_bs, bias_hs, bias_rs, rel_shift);
  hipStream_t s = (hipStream_t)stream;
  if (g_otr_attn_waves8 && a.vec && d->dtype == OTR_H16 && d->dk == 64) {
    hipLaunchKernelGGL((attn_fwd_kernel<bf16_t, 64, true, 8>), attn_grid(a, d, (d->Tq + 127) / 128), dim3(512), 0, s, a);
    return otr_check_launch("attention_bias_fwd");
  }
  dim3 grid = attn_grid(a, d, (d->Tq + 63) / 64);
  if (d->dtype == OTR_H16) { DK_SWITCH(bf16_t, attn_fwd_kernel, grid) } else { DK_SWITCH(float, attn_fwd_kernel, grid) }
  return otr_check_launch("attention_bias_fwd");
}

static int32_t attention_bwd_impl(const otr_attn_desc_t* d, AttnArgs& a, void* stream);

extern "C" int32_t otr_attention_bias_bwd(const otr_attn_desc_t* d, const void* q, const void* k, const void* v,
                                          const uint8_t* key_mask, const float* bias, void* dbias, int32_t dbias_dtype, int64_t bias_bs,
                                          int64_t bias_hs, int64_t bias_rs, int32_t rel_shift, const void* o,
                                          const void* do_, const float* lse, float* delta, void* dq, void* dk, void* dv,
                                          void* stream) {
  AttnArgs a{};
  if (int32_t e = fill_args(d, a)) return e;
  OTR_REQUIRE(q && k && v && o && do_ && lse && delta && dq && dk && dv && bias, "attention_bias_bwd: null pointer");
  OTR_REQUIRE(dbias_dtype == OTR_F32 || dbias_dtype == OTR_H16, "attention_bias_bwd: bad dbias dtype %d", dbias_dtype);
  a.dbias_h16 = dbias_dtype == OTR_H16;
  a.q = q; a.k = k; a.v = v; a.o = o; a.do_ = do_; a.lse = const_cast<float*>(lse); a.delta = delta;
  a.dq = dq; a.dk = dk; a.dv = dv; a.key_mask = key_mask;
  a.vec = vec_ok(d, {q, k, v, o, do_, dq, dk, dv});
  set_bias(a, bias, reinterpret_cast<float*>(dbias), bias_bs, bias_hs, bias_rs, rel_shift);
  return attention_bwd_impl(d, a, stream);
}

extern "C" int32_t otr_attention_bwd(const otr_attn_desc_t* d, const void* q, const void* k, const void* v,
                                     const uint8_t* key_mask, const void* o, const void* do_, const float* lse,
                                     float* delta, void* dq, void* dk, void* dv, void* stream) {
  AttnArgs a{};
  if (int32_t e = fill_args(d, a)) return e;
  OTR_REQUIRE(q && k && v && o && do_ && lse && delta && dq && dk && dv, "attention_bwd: null pointer");
  a.q = q; a.k = k; a.v = v; a.o = o; a.do_ = do_; a.lse = const_cast<float*>(lse); a.delta = delta;
  a.dq = dq; a.dk = dk; a.dv = dv; a.key_mask = key_mask;
  a.vec = vec_ok(d, {q, k, v, o, do_, dq, dk, dv});
  return attention_bwd_impl(d, a, stream);
}

extern int g_otr_attn_bwd_split;   // api.hip (otr_debug_set(13, 1)): the two-launch form, for A/B runs
extern int g_otr_attn_enc;         // api.hip (otr_debug_set(21, v)): the whole-utterance-in-LDS backward kernel (encattn.hip) where it serves
bool encattn_bwd_takes(int dtype_is_h16, int dk, int Tq, int Tk, int causal, int has_bias, int vec);
// encattn96.hip: the Conformer's relative-position self-attention backward (head dim 96, score term) on the whole-utterance design
bool encattn96_bwd_takes(int dtype_is_h16, int dk, int Tq, int Tk, int causal, int has_bias, int rel_shift, int bias_vec4, int has_dbias, int vec);
int32_t encattn96_bwd_launch(const void* q, const void* k, const void* v, const void* o, const void* do_, const float* lse, const uint8_t* key_mask,
                             const float* bias, void* dbias, int dbias_h16, int64_t bias_bs, int64_t bias_hs, int64_t bias_rs, void* dq, void* dk,
                             void* dv, int B, int H, int T, int64_t q_bs, int64_t q_ts, int64_t k_bs, int64_t k_ts, int64_t v_bs, int64_t v_ts,
                             int64_t o_bs, int64_t o_ts, float scale, hipStream_t stream);
int32_t encattn_bwd_launch(const void* q, const void* k, const void* v, const void* o, const void* do_, const float* lse, const uint8_t* key_mask,
                           void* dq, void* dk, void* dv, int B, int H, int T, int64_t q_bs, int64_t q_ts, int64_t k_bs, int64_t k_ts, int64_t v_bs,
                           int64_t v_ts, int64_t o_bs, int64_t o_ts, float scale, hipStream_t stream);
template <class CT, int DK> static void attn_bwd_merged_launch(const otr_attn_desc_t* d, AttnArgs& a, hipStream_t s) {
  if constexpr (DK == 64 && sizeof(CT) == 2) {
    if (g_otr_attn_waves8) {
      const int nq8 = (d->Tq + 127) / 128, nk8 = (d->Tk + 127) / 128;
      hipLaunchKernelGGL((attn_bwd_kernel<CT, DK, 8>), attn_grid(a, d, nq8 + nk8), dim3(512), 0, s, a, nq8);
      return;
    }
  }
  const int nqb = (d->Tq + 63) / 64, nkb = (d->Tk + 63) / 64;
  hipLaunchKernelGGL((attn_bwd_kernel<CT, DK, 4>), attn_grid(a, d, nqb + nkb), dim3(256), 0, s, a, nqb);
}
#define ATTN_BWD_MERGED(CTYPE, DKV) attn_bwd_merged_launch<CTYPE, DKV>(d, a, s)
static int32_t attention_bwd_impl(const otr_attn_desc_t* d, AttnArgs& a, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  const int nqb = (d->Tq + 63) / 64, nkb = (d->Tk + 63) / 64;
  if (encattn96_bwd_takes(d->dtype == OTR_H16, d->dk, d->Tq, d->Tk, a.causal, a.bias != nullptr, a.rel_shift, a.bias_vec4, a.dbias != nullptr, a.vec))
    return encattn96_bwd_launch(a.q, a.k, a.v, a.o, a.do_, a.lse, a.key_mask, a.bias, a.dbias, a.dbias_h16, a.bias_bs, a.bias_hs, a.bias_rs, a.dq,
                                a.dk, a.dv, d->B, d->H, d->Tq, a.q_bs, a.q_ts, a.k_bs, a.k_ts, a.v_bs, a.v_ts, a.o_bs, a.o_ts, a.scale, s);
  if (g_otr_attn_enc && encattn_bwd_takes(d->dtype == OTR_H16, d->dk, d->Tq, d->Tk, a.causal, a.bias != nullptr || a.dbias != nullptr, a.vec))
    return encattn_bwd_launch(a.q, a.k, a.v, a.o, a.do_, a.lse, a.key_mask, a.dq, a.dk, a.dv, d->B, d->H, d->Tq, a.q_bs, a.q_ts, a.k_bs, a.k_ts,
                              a.v_bs, a.v_ts, a.o_bs, a.o_ts, a.scale, s);
  // one launch where both halves fit the 256-register budget of two waves per SIMD without spilling (16-bit: head dims up to 64,
  // fp32: up to 32; aligned operands)
  if (a.vec && !g_otr_attn_bwd_split && (d->dtype == OTR_H16 ? d->dk <= 64 : d->dk <= 32)) {
    if (d->dtype == OTR_H16) {
      switch (d->dk) {
        case 16: ATTN_BWD_MERGED(bf16_t, 16); break;
        case 32: ATTN_BWD_MERGED(bf16_t, 32); break;
        default: ATTN_BWD_MERGED(bf16_t, 64); break;
      }
    } else {
      if (d->dk == 16) ATTN_BWD_MERGED(float, 16); else ATTN_BWD_MERGED(float, 32);
    }
    return otr_check_launch("attention_bwd");
  }
  // dQ first: it also produces delta = rowsum(dO * O), which the dK/dV kernel reads
  dim3 gk((d->Tk + 63) / 64, d->H, d->B), gq((d->Tq + 63) / 64, d->H, d->B);
  if (d->dtype == OTR_H16) {
    DK_SWITCH(bf16_t, attn_bwd_dq_kernel, gq)
    DK_SWITCH(bf16_t, attn_bwd_dkdv_kernel, gk)
  } else {
    DK_SWITCH(float, attn_bwd_dq_kernel, gq)
    DK_SWITCH(float, attn_bwd_dkdv_kernel, gk)
  }
  return otr_check_launch("attention_bwd");
}

// host-side view of the XCD-aware 1-D grid (tests): returns its size for nx blocks per (head, utterance); out[3 i ..] = (block, head,
// utterance) of workgroup i, or (-1, -1, -1) for a padding workgroup
extern "C" int32_t otr_debug_attention_grid(int32_t nx, int32_t H, int32_t B, int32_t* out, int32_t cap) {
  OTR_REQUIRE(nx > 0 && H > 0 && B > 0 && cap >= 0, "debug_attention_grid: bad arguments");
  const int32_t grid = 8 * nx * ((H * B + 7) / 8);
  if (!out) return grid;
  for (int i = 0; i < grid && i < cap; ++i) {
    int bx, h, b;
    if (!attn_block_of(i, nx, H, B, bx, h, b)) bx = h = b = -1;
    out[3 * i] = bx; out[3 * i + 1] = h; out[3 * i + 2] = b;
  }
  return grid;
}
