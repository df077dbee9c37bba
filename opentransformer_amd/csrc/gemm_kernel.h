// MFMA GEMM core:  C[M,N] = act( sum_k A(m,k) * B(n,k) + bias[n] ) (+ C)
//
// One kernel template serves nn.Linear forward / dgrad / wgrad and the conv2 implicit GEMMs by
// changing how each operand tile is gathered from HBM ("mode"); the LDS image and the MFMA loop are
// always the same: tile rows x KCH 16-byte chunks (KCH/4 MFMA k-steps), XOR-swizzled so the 16-lane
// ds_read_b128 operand reads are bank-conflict free.
//
//   MODE_KC      operand stored [rows][K], K contiguous            (x and w in y = x w^T)
//   MODE_MC      operand stored [K][rows], rows contiguous         (w in dgrad; dy and x in wgrad)
//                -> each thread gathers a PM(rows) x CE(k) patch with PM-wide vector loads and
//                   transposes it in registers (8x4 bf16 patches: 8-16 B per lane per load)
//   MODE_IM2K    rows = conv2 output pixels, k = (kh,kw,c1) gathered from channel-last act1
//   MODE_IM2M    rows = (kh,kw,c1), contraction = output pixels   (conv2 wgrad B operand)
//
// Pipeline: double-buffered LDS; raw (same-type) operands additionally keep a 2-slot register ring, so the global
// loads of stage t+2 are in flight while stage t is multiplied and stage t+1 is written to the other LDS buffer
// -> one barrier per BK = KCH*CE contraction elements (64 for bf16).  Workgroup = 4 waves in a 2x2 grid; wave tile
// (BM/2)x(BN/2) of 16x16 MFMA tiles.  The accumulator is kept TRANSPOSED (mma(B-frag, A-frag)): lane holds row
// m = lane&15 and four consecutive columns n = (lane>>4)*4..; FAST kernels stage the finished tile through the idle
// operand LDS and write whole rows (16 B per lane), with optional fused GLU forward / backward epilogues (EPI).
//
// Persistent tile loop (no split-K, <= OTR_RESIDENT_WG workgroups): the next tile's operands are prefetched across the
// epilogue.  Split-K (blockIdx.y): k-slices write fp32 partial tiles to the caller's workspace and
// splitk_reduce_kernel sums them in a fixed order (deterministic; no atomics).  Weight gradients do not come through
// here one by one: gemm_grouped_kernel runs all of a backward pass's dW problems from a device-side descriptor table.
#pragma once
#include <type_traits>

#include "common.h"

enum { MODE_KC = 0, MODE_MC = 1, MODE_IM2K = 2, MODE_IM2M = 3 };

struct ConvGeom {
  int T1, F1, C1, T2, F2;
  FastDiv divF2, divT2, divC1;
  int64_t a1_elems;   // elements of act1 (clamp range of the unconditional im2col loads)
};

constexpr int OTR_ACT_GLU_BWD = 2;   // internal epilogue mode (otr_ffn_glu_bwd)
constexpr int OTR_ACT_GLU_FWD = 3;   // internal epilogue mode (otr_ffn_glu_fwd): h = x.W1^T + b and u = GLU(h) in one launch

struct GemmArgs {
  const void* A;
  const void* B;
  void* C;
  const float* bias;
  int M, N, K;
  int64_t lda, ldb, ldc;
  int act, accumulate;
  int a_vec, b_vec;  // operand base/ld satisfy the vector-load alignment
  int ksplit;        // number of k-slices (grid.y); > 1 => partial tiles to ws + splitk_reduce_kernel
  int allow_split;   // caller permits split-K (needs a workspace)
  float* ws;         // split-K workspace: ksplit partial [M,N] fp32 slabs, reduced in a fixed order
  int64_t ws_bytes;
  // act == OTR_ACT_GLU_BWD (FFN backward, see the epilogue): C is NOT written; aux_in = h [M, 2N] (GLU input saved by the
  // forward), aux_out = dh [M, 2N], aux_part = per-row-tile column sums of dh [tiles_m, 2N] (the w_1 bias gradient)
  const void* aux_in;
  void* aux_out;
  float* aux_part;
  int aux_flag;               // GLU_BWD: 1 = the second half of aux_in already holds sigmoid(gate)
  unsigned long long* trace;  // tuning hook (otr_debug_trace): per-workgroup phase timestamps, or NULL
  ConvGeom cg;
  // batched launch (otr_linear_fwd_batched): problem z = blockIdx.z uses A + z bsa, B + z bsb, C + z bsc (BYTE strides); no split-K
  int nbatch;
  int64_t bsa, bsb, bsc;
  int xcd_map;       // gemm_tile_of: positions equal modulo 8 share their row blocks of A (set by the launcher)
};

template <class CT> struct GemmCfg {
  static constexpr int CE = MMA<CT>::CE;
  static constexpr int KCH = sizeof(CT) == 2 ? 8 : 4;   // chunks per tile row
  static constexpr int BK = KCH * CE;                   // contraction elements per stage
  static constexpr int ROWB = KCH * 16;                 // bytes per tile row
  static constexpr int PM = sizeof(CT) == 2 ? 4 : 2;    // rows per transposing-loader patch
};

// pixel (b,t2,f2) -> element offset of act1[b, 2*t2, 2*f2-1, 0] (tap (0,0); may be "negative" in f)
__device__ __forceinline__ int64_t im2col_base(const ConvGeom& g, uint32_t m, int& f2_out) {
  uint32_t t = fdiv(m, g.divF2);
  uint32_t f2 = m - t * g.F2;
  uint32_t b = fdiv(t, g.divT2);
  uint32_t t2 = t - b * g.T2;
  f2_out = (int)f2;
  return (((int64_t)b * g.T1 + 2 * t2) * g.F1 + (2 * (int64_t)f2 - 1)) * g.C1;
}

template <class ST, int N> __device__ __forceinline__ void load_vec(const ST* p, float* out) {
  if constexpr (sizeof(ST) == 4) {
    if constexpr (N == 4) {
      float4 v = *reinterpret_cast<const float4*>(p);
      out[0] = v.x; out[1] = v.y; out[2] = v.z; out[3] = v.w;
    } else {
      float2 v = *reinterpret_cast<const float2*>(p);
      out[0] = v.x; out[1] = v.y;
    }
  } else {
    if constexpr (N == 4) {
      uint2 v = *reinterpret_cast<const uint2*>(p);
      out[0] = h2f_lo(v.x); out[1] = h2f_hi(v.x);
      out[2] = h2f_lo(v.y); out[3] = h2f_hi(v.y);
    } else {
      uint32_t v = *reinterpret_cast<const uint32_t*>(p);
      out[0] = h2f_lo(v); out[1] = h2f_hi(v);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Operand tile loader.  ROWS x KCH chunks per stage, 256 threads.
//
// FAST = true is the production path: every global load is UNCONDITIONAL (out-of-range rows are clamped to a
// valid row -- their accumulators are never stored -- and out-of-range k is loaded from k = 0 and zeroed with
// a select), so the compiler can issue a whole stage (or three) of loads back to back and wait with counted
// vmcnt.  With loads inside `if (in range) { if (aligned) ... else scalar fallback }` control flow hipcc put an
// `s_waitcnt vmcnt(0)` behind every single load (seen in the ISA; 2-3 us per k-stage).  FAST needs 16-byte
// aligned rows and K % CE == 0 (row-major modes) / nrows % PM == 0 (transposing modes); everything else runs
// the generic (FAST = false) instantiation.
template <class CT, class ST, int MODE, int ROWS, bool FAST> struct TileLoader {
  using G = GemmCfg<CT>;
  static constexpr int CE = G::CE, KCH = G::KCH, PM = G::PM;
  static constexpr bool ROWMAJOR = (MODE == MODE_KC || MODE == MODE_IM2K);
  // row-major modes: NU chunk units per thread (unit id = tid + 256u -> row = id / KCH, chunk = id % KCH)
  static constexpr int NU = ROWMAJOR ? (ROWS * KCH + 255) / 256 : 1;
  // transposing modes: (ROWS/PM) x KCH patches of PM rows x CE k; patch id = tid + 256u
  static constexpr int RG = ROWS / PM;
  static constexpr int NP = ROWMAJOR ? 1 : (RG * KCH + 255) / 256;
  // same-type row-major operands are staged as raw 16-byte chunks (no float round trip)
  static constexpr bool RAWQ = ROWMAJOR && std::is_same<ST, CT>::value;
  // same-type bf16 transposing operands (dy, x of every weight gradient; w of dgrad) stay raw 16-bit as well: 8-byte
  // loads of 4 rows per contraction index, transposed at LDS-store time with v_perm_b32 (one per output dword)
  // instead of a bf16 -> f32 unpack + re-pack of every element
  static constexpr bool RAWT = (MODE == MODE_MC) && FAST && std::is_same<ST, CT>::value && sizeof(CT) == 2;
  // implicit-im2col rows (conv2 weight gradient) the same way: (pixel, tap) -> 4 channels per 8-byte load, validity of
  // the frequency tap kept as one bit per contraction index and applied with the transposition
  static constexpr bool RAWI = (MODE == MODE_IM2M) && FAST && std::is_same<ST, CT>::value && sizeof(CT) == 2;
  // prefetch ring: FAST raw loaders keep DEPTH stages in registers (a 3-slot ring spilled: 96 VGPRs + 64 accumulators)
  // implicit-im2col rows of the conv2 FORWARD (r06): the (pixel, tap) chunk is loaded unconditionally from a clamped address and its validity
  // (row in range, frequency tap not padding, k inside K) kept as one bit per unit, applied when the chunk is written to LDS
  static constexpr bool RAWK = (MODE == MODE_IM2K) && FAST && RAWQ && sizeof(CT) == 2;
  static constexpr int DEPTH = ((RAWQ && FAST && MODE == MODE_KC) || RAWT || RAWI || RAWK) ? 2 : 1;
  // every thread owns a full set of units (true for all tile shapes instantiated): lets stores/loads drop the
  // per-unit activity test the compiler cannot fold (it does not know threadIdx.x < 256)
  static constexpr bool ALLACTIVE = ROWMAJOR ? ((ROWS * KCH) % 256 == 0) : ((RG * KCH) % 256 == 0);

  const ST* base;
  int64_t ld;
  int nrows, K, row0;
  bool vec;
  // stage-invariant part of every unit's global address (k = 0), computed once in init()
  const ST* p0[ROWMAJOR ? NU : NP];
  bool rok[ROWMAJOR ? NU : NP];
  float raw[(RAWQ || RAWT || RAWI) ? 1 : (ROWMAJOR ? NU : NP * PM)][CE];
  uint4 rawq[RAWQ ? DEPTH : 1][RAWQ ? NU : 1];
  uint2 rawt[(RAWT || RAWI) ? DEPTH : 1][(RAWT || RAWI) ? NP : 1][(RAWT || RAWI) ? CE : 1];
  uint32_t okb[RAWK ? DEPTH : 1];                     // RAWK: bit u = unit u of the stage is a real element
  uint32_t vbits[RAWI ? DEPTH : 1][RAWI ? NP : 1];   // RAWI: bit j = contraction index j of the patch is a real (unpadded) tap
  // im2col state
  int64_t pix[ROWMAJOR ? NU : 1];
  int f2v[ROWMAJOR ? NU : 1];
  int tapoff[ROWMAJOR ? 1 : NP], tapkw[ROWMAJOR ? 1 : NP];

  // pair_F > 0 (FAST KC only; GLU-forward fusion): the tile's second half of rows comes from pair_F rows further down, i.e.
  // tile row lr <-> operand row row0 + (lr mod ROWS/2) + (lr >= ROWS/2 ? pair_F : 0), so that one tile holds the GLU
  // value columns [n0, n0+ROWS/2) AND their gate columns [F+n0, ...)
  __device__ __forceinline__ void init(const void* p, int64_t ld_, int nrows_, int K_, int row0_, bool vec_,
                                       const ConvGeom& g, int tid, int pair_F = 0) {
    base = reinterpret_cast<const ST*>(p);
    ld = ld_; nrows = nrows_; K = K_; row0 = row0_; vec = vec_;
    if constexpr (MODE == MODE_KC) {
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        int id = tid + 256 * u;
        int lr = id / KCH;
        rok[u] = lr < ROWS && row0 + lr < nrows;
        if constexpr (FAST) {
          const int gr = pair_F > 0 ? row0 + (lr & (ROWS / 2 - 1)) + (lr >= ROWS / 2 ? pair_F : 0) : row0 + lr;
          p0[u] = base + (int64_t)min(gr, nrows - 1) * ld;      // row base, chunk added per stage
        } else p0[u] = base + (int64_t)(row0 + lr) * ld + (id % KCH) * CE;
      }
    }
    if constexpr (MODE == MODE_MC) {
#pragma unroll
      for (int u = 0; u < NP; ++u) {
        int id = tid + 256 * u;
        int rg = id % RG, c = id / RG;
        rok[u] = c < KCH;
        int r = row0 + PM * rg;
        if constexpr (FAST) p0[u] = base + ((r + PM <= nrows) ? r : 0);                     // k offset added per stage
        else p0[u] = base + (int64_t)(c * CE) * ld + r;
      }
    }
    if constexpr (MODE == MODE_IM2K) {
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        int row = row0 + (tid + 256 * u) / KCH;
        f2v[u] = 0;
        pix[u] = (row < nrows) ? im2col_base(g, (uint32_t)row, f2v[u]) : 0;
      }
    }
    if constexpr (MODE == MODE_IM2M) {
#pragma unroll
      for (int u = 0; u < NP; ++u) {
        int rg = (tid + 256 * u) % RG;
        int n0 = row0 + PM * rg;             // row index = tap*C1 + c1 (PM channels share a tap)
        if constexpr (FAST) n0 = (n0 + PM <= nrows) ? n0 : 0;      // rows past the end: any valid tap (results unused)
        uint32_t tap = fdiv((uint32_t)n0, g.divC1);
        int ch = n0 - (int)tap * g.C1;
        int kh = (int)tap / 3, kw = (int)tap - kh * 3;
        tapoff[u] = (kh * g.F1 + kw) * g.C1 + ch;
        tapkw[u] = kw;
      }
    }
  }

  // issue the global loads for the stage starting at k0 into ring slot SLOT (compile-time)
  template <int SLOT = 0> __device__ __forceinline__ void load(int k0, const ConvGeom& g, int tid) {
    if constexpr (RAWK) {
      uint32_t bits = 0;
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        const int id = tid + 256 * u, row = row0 + id / KCH, gk = k0 + (id % KCH) * CE;
        const int gkc = gk & (int)((uint32_t)0 - (uint32_t)(gk < K));          // K % CE == 0 and C1 % CE == 0 (host): a chunk is one tap's
        const uint32_t tap = fdiv((uint32_t)gkc, g.divC1);
        const int ch = gkc - (int)tap * g.C1, kh = (int)tap / 3, kw = (int)tap - kh * 3;
        const int fin = 2 * f2v[u] + kw - 1;
        const bool ok = row < nrows && gk < K && fin >= 0 && fin < g.F1;
        const int64_t off = min(max(pix[u] + (int64_t)(kh * g.F1 + kw) * g.C1 + ch, (int64_t)0), g.a1_elems - CE);
        rawq[SLOT][u] = ld_global_b128(base + off);
        bits |= (uint32_t)ok << u;
      }
      okb[SLOT] = bits;
    } else if constexpr (MODE == MODE_KC && FAST) {
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        const int gk = k0 + ((tid + 256 * u) % KCH) * CE;
        // k tail: load from k = 0 (always valid) and zero the chunk with an AND mask when it is written to LDS
        // (store()): the load stays unconditional and nothing consumes its result before the MFMAs of the
        // current stage -- a select here gets turned back into a branch + vmcnt(0) by the compiler
        const uint32_t m = (uint32_t)0 - (uint32_t)(gk < K);
        const ST* src = p0[u] + (gk & (int)m);
        if constexpr (RAWQ) {
          rawq[SLOT][u] = ld_global_b128(src);
        } else {
          load_row<ST, CE>(src, CE, true, raw[u]);
#pragma unroll
          for (int e = 0; e < CE; ++e) raw[u][e] = __uint_as_float(__float_as_uint(raw[u][e]) & m);
        }
      }
    } else if constexpr (RAWT) {
#pragma unroll
      for (int u = 0; u < NP; ++u) {
        const int kb = k0 + ((tid + 256 * u) / RG) * CE;          // K % CE == 0 (host): a chunk is valid as a whole
        const int kbc = kb & (int)((uint32_t)0 - (uint32_t)(kb < K));
        const ST* pj = p0[u] + (int64_t)kbc * ld;
#pragma unroll
        for (int j = 0; j < CE; ++j, pj += ld) rawt[SLOT][u][j] = ld_global_b64(pj);
      }
    } else if constexpr (RAWI) {
#pragma unroll
      for (int u = 0; u < NP; ++u) {
        const int kb = k0 + ((tid + 256 * u) / RG) * CE;          // K % CE == 0 (host): a chunk is valid as a whole
        const int kbc = kb & (int)((uint32_t)0 - (uint32_t)(kb < K));
        uint32_t bits = 0;
#pragma unroll
        for (int j = 0; j < CE; ++j) {
          int f2;
          const int64_t pb = im2col_base(g, (uint32_t)(kbc + j), f2);
          const int fin = 2 * f2 + tapkw[u] - 1;
          bits |= (uint32_t)(fin >= 0 && fin < g.F1) << j;
          const int64_t off = min(max(pb + tapoff[u], (int64_t)0), g.a1_elems - PM);   // padded taps read a valid address
          rawt[SLOT][u][j] = ld_global_b64(base + off);
        }
        vbits[SLOT][u] = bits;
      }
    } else if constexpr (MODE == MODE_MC && FAST) {
#pragma unroll
      for (int u = 0; u < NP; ++u) {
        const int c = (tid + 256 * u) / RG;
        if (ALLACTIVE || rok[u]) {             // wave-uniform: RG is a multiple of 16
          const int kb = k0 + c * CE;
#pragma unroll
          for (int j = 0; j < CE; ++j) {
            const uint32_t m = (uint32_t)0 - (uint32_t)(kb + j < K);
            float v[PM];
            load_vec<ST, PM>(p0[u] + (int64_t)((kb + j) & (int)m) * ld, v);
#pragma unroll
            for (int i = 0; i < PM; ++i) raw[u * PM + i][j] = __uint_as_float(__float_as_uint(v[i]) & m);
          }
        }
      }
    } else if constexpr (MODE == MODE_KC) {
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        int id = tid + 256 * u;
        int gk = k0 + (id % KCH) * CE;
        const bool ok = rok[u] && gk < K;
        if constexpr (RAWQ) {
          rawq[SLOT][u] = make_uint4(0, 0, 0, 0);
          if (ok) {
            const ST* src = p0[u] + k0;
            if (vec && K - gk >= CE) {
              rawq[SLOT][u] = *reinterpret_cast<const uint4*>(src);
            } else {
              float tmp[CE];
              load_row<ST, CE>(src, K - gk, false, tmp);
              rawq[SLOT][u] = MMA<CT>::pack(tmp);
            }
          }
        } else {
          if (ok) {
            load_row<ST, CE>(p0[u] + k0, K - gk, vec, raw[u]);
          } else {
#pragma unroll
            for (int e = 0; e < CE; ++e) raw[u][e] = 0.f;
          }
        }
      }
    } else if constexpr (MODE == MODE_IM2K) {
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        int id = tid + 256 * u;
        int lr = id / KCH;
        int row = row0 + lr, gk = k0 + (id % KCH) * CE;
        uint32_t tap = fdiv((uint32_t)gk, g.divC1);
        int ch = gk - (int)tap * g.C1;
        int kh = (int)tap / 3, kw = (int)tap - kh * 3;
        int fin = 2 * f2v[u] + kw - 1;
        const bool ok = lr < ROWS && row < nrows && gk < K && fin >= 0 && fin < g.F1;
        const ST* src = base + pix[u] + (int64_t)(kh * g.F1 + kw) * g.C1 + ch;
        if constexpr (RAWQ) {
          rawq[SLOT][u] = make_uint4(0, 0, 0, 0);
          if (ok) {
            if (vec) {
              rawq[SLOT][u] = *reinterpret_cast<const uint4*>(src);
            } else {
              float tmp[CE];
              load_row<ST, CE>(src, CE, false, tmp);
              rawq[SLOT][u] = MMA<CT>::pack(tmp);
            }
          }
        } else {
          if (ok) {
            load_row<ST, CE>(src, CE, vec, raw[u]);
          } else {
#pragma unroll
            for (int e = 0; e < CE; ++e) raw[u][e] = 0.f;
          }
        }
      }
    } else {
#pragma unroll
      for (int u = 0; u < NP; ++u) {
        int id = tid + 256 * u;
        int rg = id % RG, c = id / RG;
        int r = row0 + PM * rg;
        bool active = c < KCH;
        const ST* pstage = nullptr;
        if constexpr (MODE == MODE_MC) pstage = p0[u] + (int64_t)k0 * ld;
#pragma unroll
        for (int j = 0; j < CE; ++j) {
          int gk = k0 + c * CE + j;
          float v[PM];
#pragma unroll
          for (int i = 0; i < PM; ++i) v[i] = 0.f;
          if (active && gk < K) {
            const ST* p;
            bool ok = true;
            if constexpr (MODE == MODE_MC) {
              p = pstage + (int64_t)j * ld;
            } else {  // MODE_IM2M: contraction index gk = output pixel
              int f2;
              int64_t pb = im2col_base(g, (uint32_t)gk, f2);
              int fin = 2 * f2 + tapkw[u] - 1;
              ok = fin >= 0 && fin < g.F1;
              p = base + pb + tapoff[u];
            }
            if (ok) {
              if (vec && r + PM <= nrows) {
                load_vec<ST, PM>(p, v);
              } else {
#pragma unroll
                for (int i = 0; i < PM; ++i)
                  if (r + i < nrows) v[i] = ElemIO<ST>::ld(p + i);
              }
            }
          }
#pragma unroll
          for (int i = 0; i < PM; ++i) raw[u * PM + i][j] = v[i];
        }
      }
    }
  }

  // convert to CT and write the tile image: byte(row, chunk) = row*ROWB + ((chunk ^ swz(row)) << 4)
  // k0 = first contraction index of the stage held in SLOT (used for the FAST k-tail mask)
  template <int SLOT = 0> __device__ __forceinline__ void store(unsigned char* lds, int tid, int k0 = 0) const {
    if constexpr (ROWMAJOR) {
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        int id = tid + 256 * u;
        int row = id / KCH, c = id % KCH;
        if (ALLACTIVE || row < ROWS) {
          uint4 q;
          if constexpr (RAWQ) {
            q = rawq[SLOT][u];
            if constexpr (FAST && MODE == MODE_KC) {
              const uint32_t m = (uint32_t)0 - (uint32_t)(k0 + c * CE < K);
              q = make_uint4(q.x & m, q.y & m, q.z & m, q.w & m);
            }
            if constexpr (RAWK) {
              const uint32_t m = (uint32_t)0 - ((okb[SLOT] >> u) & 1u);
              q = make_uint4(q.x & m, q.y & m, q.z & m, q.w & m);
            }
          } else q = MMA<CT>::pack(raw[u]);
          *reinterpret_cast<uint4*>(lds + row * G::ROWB + ((c ^ swz<KCH>(row)) << 4)) = q;
        }
      }
    } else if constexpr (RAWT || RAWI) {
#pragma unroll
      for (int u = 0; u < NP; ++u) {
        const int id = tid + 256 * u, rg = id % RG, c = id / RG;
        if (ALLACTIVE || c < KCH) {
          const uint32_t m = (uint32_t)0 - (uint32_t)(k0 + c * CE < K);
          uint2 rr[CE];
#pragma unroll
          for (int j = 0; j < CE; ++j) {
            rr[j] = rawt[SLOT][u][j];
            if constexpr (RAWI) {                       // zero the padded frequency taps before the transposition
              const uint32_t mj = (uint32_t)0 - ((vbits[SLOT][u] >> j) & 1u);
              rr[j].x &= mj; rr[j].y &= mj;
            }
          }
          const uint2* r = rr;
#pragma unroll
          for (int i = 0; i < PM; ++i) {        // row i of the patch = 16-bit field i of every 8-byte load
            const uint32_t sel = (i & 1) ? 0x07060302u : 0x05040100u;
            uint4 q;
            if (i < 2) {
              q.x = __builtin_amdgcn_perm(r[1].x, r[0].x, sel); q.y = __builtin_amdgcn_perm(r[3].x, r[2].x, sel);
              q.z = __builtin_amdgcn_perm(r[5].x, r[4].x, sel); q.w = __builtin_amdgcn_perm(r[7].x, r[6].x, sel);
            } else {
              q.x = __builtin_amdgcn_perm(r[1].y, r[0].y, sel); q.y = __builtin_amdgcn_perm(r[3].y, r[2].y, sel);
              q.z = __builtin_amdgcn_perm(r[5].y, r[4].y, sel); q.w = __builtin_amdgcn_perm(r[7].y, r[6].y, sel);
            }
            q.x &= m; q.y &= m; q.z &= m; q.w &= m;
            const int row = PM * rg + i;
            *reinterpret_cast<uint4*>(lds + row * G::ROWB + ((c ^ swz<KCH>(row)) << 4)) = q;
          }
        }
      }
    } else {
#pragma unroll
      for (int u = 0; u < NP; ++u) {
        int id = tid + 256 * u;
        int rg = id % RG, c = id / RG;
        if (ALLACTIVE || c < KCH) {
#pragma unroll
          for (int i = 0; i < PM; ++i) {
            int row = PM * rg + i;
            *reinterpret_cast<uint4*>(lds + row * G::ROWB + ((c ^ swz<KCH>(row)) << 4)) = MMA<CT>::pack(raw[u * PM + i]);
          }
        }
      }
    }
  }
};

// ------------------------------------------------------------------------------------------------
// partial last chunk of a staged row (N not a multiple of the chunk width): out of line, keeps branches with memory
// operations out of the hot epilogue
template <class OT> __device__ __noinline__ void store_tail(OT* dst, const OT* src, int n) {
  for (int e = 0; e < n; ++e) dst[e] = src[e];
}
template <class OT> __device__ __noinline__ void store_tail_acc(OT* dst, const OT* src, int n, int relu) {
  for (int e = 0; e < n; ++e) {
    float v = ElemIO<OT>::ld(src + e) + ElemIO<OT>::ld(dst + e);
    ElemIO<OT>::st(dst + e, relu ? fmaxf(v, 0.f) : v);
  }
}

// One workgroup's share of a GEMM: tiles tile0, tile0 + tile_stride, ... (PERSIST) or just tile0, k-slice `kslice`.
// Shared by the plain kernel (blockIdx -> tile) and the grouped kernel (blockIdx -> problem -> tile).
// EPI: 0 = plain epilogue (bias / ReLU / accumulate), 1 = GLU forward, 2 = GLU backward (own instantiations: their
// registers must not weigh on the plain kernel -- folded into it they made it spill)
// Position in the launch -> output tile.  Consecutive workgroup ids go to consecutive XCDs (observed placement, used for locality
// only), and every XCD has its own L2: in the natural order (column tile fastest) the tiles_n workgroups that read ONE row block of A sit
// on tiles_n different XCDs and that block crosses the fabric tiles_n times -- the Conformer's 7968 x 384 input-gradient GEMMs moved
// 127 MB per launch for 24 MB of operands and results (rocprofv3 FETCH_SIZE / WRITE_SIZE, profiles/r05_pmc_conformer_before_xcdmap.txt;
// 42 MB with this map, profiles/r05_pmc_conformer.txt).  Here positions that are EQUAL modulo 8 walk the column tiles of the same row blocks: row blocks
// 8g + (position % 8), column tile fastest.  A bijection of [0, tiles_m * tiles_n) (the last tiles_m % 8 row blocks keep the natural
// order), so the persistent loop (stride = a multiple of 8) and split-K are untouched.
extern int g_otr_gemm_xcd_map;   // api.hip (otr_debug_set(26, v)): 0 = the natural order
__device__ __forceinline__ void gemm_tile_of(int t, int tiles_m, int tiles_n, bool xmap, int& tm, int& tn) {
  const int full = xmap ? (tiles_m >> 3) * 8 * tiles_n : 0;
  if (t < full) {
    const int idx = t >> 3, g = idx / tiles_n;
    tm = g * 8 + (t & 7);
    tn = idx - g * tiles_n;
  } else {
    const int r = t - full, q = r / tiles_n;
    tm = (xmap ? (tiles_m & ~7) : 0) + q;
    tn = r - q * tiles_n;
  }
}

template <class CT, class AT, class BT, class OT, int AMODE, int BMODE, int BM, int BN, bool FAST, bool PERSIST, int EPI = 0>
__device__ __forceinline__ void gemm_body(const GemmArgs& p, const int tile0, const int tile_stride, const int kslice) {
  using G = GemmCfg<CT>;
  constexpr int KCH = G::KCH, BK = G::BK, ROWB = G::ROWB;
  constexpr int WM = BM / 2, WN = BN / 2, FM = WM / 16, FN = WN / 16;
  constexpr int BUF = (BM + BN) * ROWB;  // bytes per pipeline stage: A tile then B tile
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * BUF];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int tiles_n = (p.N + BN - 1) / BN;
  const int tiles_m = (p.M + BM - 1) / BM;
  const int ntiles = tiles_n * tiles_m;
  const bool xmap = p.xcd_map != 0;
  // PERSIST (ksplit == 1 only): the grid is capped at the number of resident workgroups and each one walks tiles
  // tile0, +tile_stride, ...; the operand loads of the next tile are issued before the epilogue of the current
  // one, so the ~3 k-cycle stage-in latency and the epilogue overlap instead of adding up per tile.
  int tile = tile0;
  int tile_m, tile_n;
  gemm_tile_of(tile, tiles_m, tiles_n, xmap, tile_m, tile_n);

  // k-slice of this workgroup
  const int nk_total = (p.K + BK - 1) / BK;
  const int per = (nk_total + p.ksplit - 1) / p.ksplit;
  const int kt0 = kslice * per;
  const int kt1 = min(nk_total, kt0 + per);
  if (kt0 >= kt1) return;
#define OTR_TRACE(slot) \
  if (p.trace && tid == 0) p.trace[((int64_t)kslice * ntiles + tile) * 4 + (slot)] = __builtin_readcyclecounter();
  OTR_TRACE(0)

  // GLU-forward fusion (p.act == OTR_ACT_GLU_FWD, N = 2F): B tile n holds W1 rows [n*BN/2, +BN/2) and F + the same
  constexpr bool GLU_OK = FAST && AMODE == MODE_KC && BMODE == MODE_KC && sizeof(OT) == 2;
  static_assert(EPI == 0 || GLU_OK, "GLU epilogues exist for the bf16 KC/KC fast kernels only");
  const int pairF = (EPI == 1) ? p.N / 2 : 0;
  const int bstep = pairF > 0 ? BN / 2 : BN;               // operand-row advance per n-tile
  TileLoader<CT, AT, AMODE, BM, FAST> la;
  TileLoader<CT, BT, BMODE, BN, FAST> lb;
  la.init(p.A, p.lda, p.M, p.K, tile_m * BM, p.a_vec != 0, p.cg, tid);
  lb.init(p.B, p.ldb, p.N, p.K, tile_n * bstep, p.b_vec != 0, p.cg, tid, pairF);

  using LA = TileLoader<CT, AT, AMODE, BM, FAST>;
  using LB = TileLoader<CT, BT, BMODE, BN, FAST>;
  // prefetch distance: 2 register stages when both operands are staged raw, else the classic 1
  constexpr int D = (LA::DEPTH == 2 && LB::DEPTH == 2) ? 2 : 1;
  const int fr = lane & 15, fg = lane >> 4;
  f32x4 acc[FM][FN];

  auto compute = [&](int cur) {
    const unsigned char* sa = smem + cur * BUF;
    const unsigned char* sb = sa + BM * ROWB;
#pragma unroll
    for (int ks = 0; ks < KCH / 4; ++ks) {
      uint4 af[FM], bf[FN];
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        int row = wm * WM + i * 16 + fr;
        af[i] = *reinterpret_cast<const uint4*>(sa + row * ROWB + (((ks * 4 + fg) ^ swz<KCH>(row)) << 4));
      }
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        int row = wn * WN + j * 16 + fr;
        bf[j] = *reinterpret_cast<const uint4*>(sb + row * ROWB + (((ks * 4 + fg) ^ swz<KCH>(row)) << 4));
      }
      // transposed accumulator: tile rows = n (B side), cols = m (A side)
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) MMA<CT>::mma(acc[i][j], bf[j], af[i]);
    }
  };


  // Control flow inside the k-loop is kept free of conditional loads/stores: a load that is issued on only one
  // side of a join makes hipcc's waitcnt insertion fall back to `s_waitcnt vmcnt(0)` at the join, which drains the
  // whole prefetch ring every step (seen in the ISA).  So the steady-state bodies below issue their loads and LDS
  // stores unconditionally and the loop tails are peeled into straight-line code.
  const int klast = (kt1 - 1) * BK;                // a prologue load beyond the slice is clamped (redundant, unused)
  if constexpr (D == 2) {
    la.template load<0>(kt0 * BK, p.cg, tid);
    lb.template load<0>(kt0 * BK, p.cg, tid);
    la.template load<1>(min((kt0 + 1) * BK, klast), p.cg, tid);
    lb.template load<1>(min((kt0 + 1) * BK, klast), p.cg, tid);
  }

  for (;;) {
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  if constexpr (D == 1) {
    la.template load<0>(kt0 * BK, p.cg, tid);
    lb.template load<0>(kt0 * BK, p.cg, tid);
    la.template store<0>(smem, tid, kt0 * BK);
    lb.template store<0>(smem + BM * ROWB, tid, kt0 * BK);
    __syncthreads();
    OTR_TRACE(1)
    int kt = kt0;
    for (; kt + 1 < kt1; ++kt) {
      const int cur = (kt - kt0) & 1;
      la.template load<0>((kt + 1) * BK, p.cg, tid);
      lb.template load<0>((kt + 1) * BK, p.cg, tid);
      __builtin_amdgcn_sched_barrier(0);   // the scheduler otherwise sinks the loads below the MFMAs and hoists
      compute(cur);                        // their first use above them, i.e. waits for them with nothing to overlap
      __builtin_amdgcn_sched_barrier(0);
      la.template store<0>(smem + (cur ^ 1) * BUF, tid, (kt + 1) * BK);
      lb.template store<0>(smem + (cur ^ 1) * BUF + BM * ROWB, tid, (kt + 1) * BK);
      __syncthreads();
    }
    compute((kt - kt0) & 1);
  } else {
    // 2-slot register ring: stage s lives in slot (s - kt0) & 1 and is staged to LDS buffer (s - kt0) & 1.
    // Sub-step for stage t (already in LDS): issue the loads of stage t+2 into the slot stage t vacated -> MFMAs of
    // stage t -> write stage t+1 from its slot to the other LDS buffer (counted vmcnt: the 8 loads of stage t+2 stay
    // in flight across the barrier) -> barrier.
    la.template store<0>(smem, tid, kt0 * BK);
    lb.template store<0>(smem + BM * ROWB, tid, kt0 * BK);
    __syncthreads();
    OTR_TRACE(1)
#define OTR_GEMM_STEP(CUR, DO_LOAD, DO_STORE)                                                  \
    {                                                                                           \
      if (DO_LOAD) {                                                                            \
        la.template load<CUR>((kt + 2) * BK, p.cg, tid);                                        \
        lb.template load<CUR>((kt + 2) * BK, p.cg, tid);                                        \
      }                                                                                         \
      __builtin_amdgcn_sched_barrier(0); /* keep the loads ahead of the MFMAs ... */            \
      compute(CUR);                                                                             \
      __builtin_amdgcn_sched_barrier(0); /* ... and their first use (mask + LDS write) behind */ \
      if (DO_STORE) {                                                                           \
        la.template store<(CUR) ^ 1>(smem + ((CUR) ^ 1) * BUF, tid, (kt + 1) * BK);             \
        lb.template store<(CUR) ^ 1>(smem + ((CUR) ^ 1) * BUF + BM * ROWB, tid, (kt + 1) * BK); \
        __syncthreads();                                                                        \
      }                                                                                         \
      ++kt;                                                                                     \
    }
    int kt = kt0;
    while (kt1 - kt >= 4) {                        // both loads stay inside the slice
      OTR_GEMM_STEP(0, 1, 1)
      OTR_GEMM_STEP(1, 1, 1)
    }
    switch (kt1 - kt) {                            // 1..3 stages left: straight-line drains, loads only where valid
      case 3:
        OTR_GEMM_STEP(0, 1, 1)
        OTR_GEMM_STEP(1, 0, 1)
        OTR_GEMM_STEP(0, 0, 0)
        break;
      case 2:
        OTR_GEMM_STEP(0, 0, 1)
        OTR_GEMM_STEP(1, 0, 0)
        break;
      default:
        OTR_GEMM_STEP(0, 0, 0)
        break;
    }
#undef OTR_GEMM_STEP
  }

  OTR_TRACE(2)
  // epilogue: acc[i][j][r] = C[m = .. + i*16 + (lane&15)][n = .. + j*16 + (lane>>4)*4 + r]
  if (!PERSIST && p.ksplit > 1) {  // partial tile -> workspace slab [split][M][N]; reduced by splitk_reduce_kernel
    float* W = p.ws + (int64_t)kslice * p.M * p.N;
    const bool v4 = (p.N % 4 == 0);
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      int row = tile_m * BM + wm * WM + i * 16 + fr;
      if (row >= p.M) continue;
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        int col = tile_n * BN + wn * WN + j * 16 + fg * 4;
        if (col >= p.N) continue;
        float* dst = W + (int64_t)row * p.N + col;
        if (v4 && col + 4 <= p.N) {
          *reinterpret_cast<float4*>(dst) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (col + r < p.N) dst[r] = acc[i][j][r];
        }
      }
    }
    OTR_TRACE(3)
    return;
  }
  OT* C = reinterpret_cast<OT*>(p.C);
  const bool vec_out = (p.ldc % 4 == 0) && ((uintptr_t)p.C % 16 == 0);
  float bv[FN][4];
#pragma unroll
  for (int j = 0; j < FN; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) bv[j][r] = 0.f;
  if (p.bias) {   // clamped, unconditional loads (a per-element branch serialises them behind vmcnt(0))
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      const int tc = wn * WN + j * 16 + fg * 4;              // tile column -> output column (paired halves in GLU mode)
      const int col = pairF > 0 ? tile_n * (BN / 2) + (tc & (BN / 2 - 1)) + (tc >= BN / 2 ? pairF : 0) : tile_n * BN + tc;
#pragma unroll
      for (int r = 0; r < 4; ++r) bv[j][r] = p.bias[min(col + r, p.N - 1)];
    }
  }

  const int done_m = tile_m, done_n = tile_n;
  if constexpr (PERSIST && D == 2) {
    // operands of this workgroup's next tile (clamped: the last round re-loads a valid tile and drops it); issued
    // AFTER the bias loads so that waiting for the bias does not wait for them (vmcnt retires in order)
    const int nxt = min(tile + tile_stride, ntiles - 1);
    gemm_tile_of(nxt, tiles_m, tiles_n, xmap, tile_m, tile_n);
    la.init(p.A, p.lda, p.M, p.K, tile_m * BM, p.a_vec != 0, p.cg, tid);
    lb.init(p.B, p.ldb, p.N, p.K, tile_n * bstep, p.b_vec != 0, p.cg, tid, pairF);
    la.template load<0>(kt0 * BK, p.cg, tid);
    lb.template load<0>(kt0 * BK, p.cg, tid);
    la.template load<1>(min((kt0 + 1) * BK, klast), p.cg, tid);
    lb.template load<1>(min((kt0 + 1) * BK, klast), p.cg, tid);
  }
  // Staged epilogue: each lane owns 4 consecutive n of 16 different rows, so direct stores hit memory as 32-byte
  // (bf16) / 64-byte (f32) pieces -- measured with otr_debug_trace that was 54 % of a K=256 workgroup's lifetime
  // (11.4 k of 21 k cycles for the FFN w_1 GEMM).  Instead the tile goes through the (now idle) operand LDS and
  // leaves as whole rows: 16 bytes per lane, a wave covers >= 4 rows x 256 B.
  constexpr int EPC = 16 / (int)sizeof(OT);                  // elements per 16-byte chunk
  constexpr int CROW = BN * (int)sizeof(OT) + 16;            // staged row pitch (pad: rows land on different banks)
  constexpr int PASSES = (BM * CROW <= 2 * BUF) ? 1 : (BM / 2 * CROW <= 2 * BUF) ? 2 : 4;   // f32 tiles go in row slabs
  constexpr int PROWS = BM / PASSES;
  static_assert(PROWS * CROW <= 2 * BUF && PROWS % 16 == 0, "staged epilogue does not fit the operand LDS");
  const bool acc_st = p.accumulate != 0;                    // C += ...: added (and activated) at write-out, f32 only
  // FAST kernels are launched only when C is 16-byte aligned with a chunk-multiple pitch (and not bf16 +=): staged
  // epilogue only.  The generic kernels keep the direct per-lane stores.
  if constexpr (FAST) {
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
      __syncthreads();                                       // operand reads (ps = 0) / previous write-out (ps = 1) done
      {
#pragma unroll
        for (int i = 0; i < FM; ++i) {
          const int trow = wm * WM + i * 16;                 // first tile row of this fragment (wave-uniform)
          if (PASSES > 1 && trow / PROWS != ps) continue;
          const int lrow = trow - ps * PROWS + fr;
#pragma unroll
          for (int j = 0; j < FN; ++j) {
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              v[r] = acc[i][j][r] + bv[j][r];
              if (p.act == OTR_ACT_RELU && !acc_st) v[r] = fmaxf(v[r], 0.f);
            }
            unsigned char* q = smem + lrow * CROW + (wn * WN + j * 16 + fg * 4) * (int)sizeof(OT);
            if constexpr (sizeof(OT) == 4) *reinterpret_cast<float4*>(q) = make_float4(v[0], v[1], v[2], v[3]);
            else *reinterpret_cast<uint2*>(q) = make_uint2(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]));
          }
        }
      }
      __syncthreads();
      constexpr int CPR = BN / EPC;                          // chunks per row; 256 % CPR == 0, so a thread keeps its chunk
      constexpr int RSTEP = 256 / CPR;                       // column and its rows advance by RSTEP per unit
      static_assert(256 % CPR == 0 && (PROWS * CPR) % 256 == 0, "write-out mapping");
      const int lrow0 = tid / CPR, ch = tid % CPR;
      const int col = done_n * BN + ch * EPC;
      if constexpr (EPI == 1 && PASSES == 1) {
        {
          // staged tile = h (bias added) for value columns [n0, n0+BN/2) in its left half and their gates in the right
          // half.  Every thread writes its h chunk; threads on the left half also read the partner gate chunk from LDS
          // and write u = a * sigmoid(b): the separate GLU pass (re-read h, write u) disappears.
          const int F = pairF, n0 = done_n * (BN / 2);
          constexpr int VPR = CPR / 2;                       // value chunks per row; every thread gets whole items
          constexpr int VSTEP = 256 / VPR;                   // (value chunk + its gate chunk): the exp work is balanced
          const int vr0 = tid / VPR, vc = tid % VPR;
          const int vcol = n0 + vc * EPC;                    // value column; gate column = F + vcol
          bf16_t* Hh = reinterpret_cast<bf16_t*>(p.C);
          bf16_t* Uu = reinterpret_cast<bf16_t*>(p.aux_out);
#pragma unroll
          for (int u = 0; u < (PROWS * VPR) / 256; ++u) {
            const int lrow = vr0 + u * VSTEP, row = done_m * BM + lrow;
            if (row < p.M && vcol < F) {
              const uint4 q = *reinterpret_cast<const uint4*>(smem + lrow * CROW + vc * 16);
              const uint4 g = *reinterpret_cast<const uint4*>(smem + lrow * CROW + (vc + VPR) * 16);
              const uint32_t aw[4] = {q.x, q.y, q.z, q.w}, bw[4] = {g.x, g.y, g.z, g.w};
              float o[EPC], sg[EPC];
#pragma unroll
              for (int e = 0; e < EPC; ++e) {
                const float a = ((e & 1) ? h2f_hi(aw[e >> 1]) : h2f_lo(aw[e >> 1]));
                const float b = ((e & 1) ? h2f_hi(bw[e >> 1]) : h2f_lo(bw[e >> 1]));
                sg[e] = 1.f / (1.f + __expf(-b));
                o[e] = a * sg[e];
              }
              // h keeps (value | sigmoid(gate)): the backward epilogue then needs no exponential at all
              st_global_b128(Hh + (int64_t)row * p.ldc + vcol, q);
              st_global_b128(Hh + (int64_t)row * p.ldc + F + vcol, MMA<bf16_t>::pack(sg));
              st_global_b128(Uu + (int64_t)row * F + vcol, MMA<bf16_t>::pack(o));
            }
          }
          continue;
        }
      }
      if constexpr (EPI == 2 && PASSES == 1) {
        static_assert(EPI != 2 || PROWS * CROW + 2 * RSTEP * BN * 4 <= 2 * BUF, "GLU-backward partials do not fit the LDS");
        {
          // The staged tile is du = dy . W2 for hidden units [col, col+8) of PROWS rows.  GLU backward right here:
          //   dh[:, j] = du * sig(b),  dh[:, F+j] = du * a * sig(b) * (1 - sig(b)),   (a | b) = h[:, j], h[:, F+j]
          // du never goes to HBM and the separate GLU-backward pass (read h + du, write dh) disappears.
          constexpr int NU_ = (PROWS * CPR) / 256;
          const int F = p.N;
          const bf16_t* H = reinterpret_cast<const bf16_t*>(p.aux_in);
          bf16_t* DH = reinterpret_cast<bf16_t*>(p.aux_out);
          const int colc = min(col, F - EPC);                // clamped: loads unconditional, stores / sums masked
          uint4 ha[NU_], hb[NU_];
#pragma unroll
          for (int u = 0; u < NU_; ++u) {                    // all h loads of this thread in flight at once
            const int row = min(done_m * BM + lrow0 + u * RSTEP, p.M - 1);
            const bf16_t* hp = H + (int64_t)row * (2 * F) + colc;
            ha[u] = ld_global_b128(hp);
            hb[u] = ld_global_b128(hp + F);
          }
          float sa[EPC], sb[EPC];
#pragma unroll
          for (int e = 0; e < EPC; ++e) { sa[e] = 0.f; sb[e] = 0.f; }
#pragma unroll
          for (int u = 0; u < NU_; ++u) {
            const int lrow = lrow0 + u * RSTEP, row = done_m * BM + lrow;
            const bool live = row < p.M && col < F;
            const uint4 dq = *reinterpret_cast<const uint4*>(smem + lrow * CROW + ch * 16);
            const uint32_t dw[4] = {dq.x, dq.y, dq.z, dq.w}, aw[4] = {ha[u].x, ha[u].y, ha[u].z, ha[u].w},
                           bw[4] = {hb[u].x, hb[u].y, hb[u].z, hb[u].w};
            float oa[EPC], ob[EPC];
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
              const int sh = (e & 1) ? 0 : 16;               // element e of a packed pair: low half first
              const float d = ((e & 1) ? h2f_hi(dw[e >> 1]) : h2f_lo(dw[e >> 1]));
              const float a = ((e & 1) ? h2f_hi(aw[e >> 1]) : h2f_lo(aw[e >> 1]));
              const float b = ((e & 1) ? h2f_hi(bw[e >> 1]) : h2f_lo(bw[e >> 1]));
              (void)sh;
              const float sg = p.aux_flag ? b : 1.f / (1.f + __expf(-b));
              oa[e] = live ? d * sg : 0.f;
              ob[e] = live ? d * a * sg * (1.f - sg) : 0.f;
              sa[e] += oa[e]; sb[e] += ob[e];
            }
            if (live) {
              bf16_t* dp_ = DH + (int64_t)row * (2 * F) + col;
              st_global_b128(dp_, MMA<bf16_t>::pack(oa));
              st_global_b128(dp_ + F, MMA<bf16_t>::pack(ob));
            }
          }
          // column sums over the tile's rows: [RSTEP][BN] partials per half behind the staged tile, then one row of
          // aux_part per (row tile, column) -- deterministic, no atomics
          float* red = reinterpret_cast<float*>(smem + PROWS * CROW);
#pragma unroll
          for (int e = 0; e < EPC; ++e) {
            red[lrow0 * BN + ch * EPC + e] = sa[e];
            red[RSTEP * BN + lrow0 * BN + ch * EPC + e] = sb[e];
          }
          __syncthreads();
          for (int t = tid; t < 2 * BN; t += 256) {
            const int half = t / BN, c = t - half * BN, gc = done_n * BN + c;
            if (gc < F) {
              float acc_ = 0.f;
#pragma unroll
              for (int r = 0; r < RSTEP; ++r) acc_ += red[half * RSTEP * BN + r * BN + c];
              p.aux_part[(int64_t)done_m * (2 * F) + half * F + gc] = acc_;
            }
          }
          continue;                                          // (PASSES == 1: leaves the pass loop)
        }
      }
      // one base pointer per tile + a uniform row step: per-unit addresses would be loop-invariant in the persistent
      // tile loop, get hoisted, and spill (their reloads drained the prefetched loads with vmcnt(0))
      OT* dstp = C + (int64_t)(done_m * BM + ps * PROWS + lrow0) * p.ldc + col;
      const int64_t rstep = (int64_t)RSTEP * p.ldc;
#pragma unroll
      for (int u = 0; u < (PROWS * CPR) / 256; ++u, dstp += rstep) {     // incremental: u * rstep would be hoisted too
        const int lrow = lrow0 + u * RSTEP;
        const int row = done_m * BM + ps * PROWS + lrow;
        if (row < p.M && col < p.N) {
          const unsigned char* q = smem + lrow * CROW + ch * 16;
          OT* dst = dstp;
          const int nv = min(EPC, p.N - col);
          if constexpr (sizeof(OT) == 4) {
            if (acc_st) {
              if (nv == EPC) {
                float4 a = *reinterpret_cast<const float4*>(q);
                const uint4 cu = ld_global_b128(dst);
                const float4 c4 = make_float4(__uint_as_float(cu.x), __uint_as_float(cu.y), __uint_as_float(cu.z), __uint_as_float(cu.w));
                a.x += c4.x; a.y += c4.y; a.z += c4.z; a.w += c4.w;
                if (p.act == OTR_ACT_RELU) { a.x = fmaxf(a.x, 0.f); a.y = fmaxf(a.y, 0.f); a.z = fmaxf(a.z, 0.f); a.w = fmaxf(a.w, 0.f); }
                st_global_b128(dst, make_uint4(__float_as_uint(a.x), __float_as_uint(a.y), __float_as_uint(a.z), __float_as_uint(a.w)));
              } else {
                store_tail_acc<OT>(dst, reinterpret_cast<const OT*>(q), nv, p.act == OTR_ACT_RELU);
              }
              continue;
            }
          }
          if (nv == EPC) st_global_b128(dst, *reinterpret_cast<const uint4*>(q));
          else store_tail<OT>(dst, reinterpret_cast<const OT*>(q), nv);
        }
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      int row = done_m * BM + wm * WM + i * 16 + fr;
      if (row >= p.M) continue;
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        int col = done_n * BN + wn * WN + j * 16 + fg * 4;
        if (col >= p.N) continue;
        OT* dst = C + (int64_t)row * p.ldc + col;
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = acc[i][j][r] + bv[j][r];
        const bool full = col + 4 <= p.N && vec_out;
        if (p.accumulate) {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (col + r < p.N) v[r] += ElemIO<OT>::ld(dst + r);
        }
        if (p.act == OTR_ACT_RELU) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
        }
        if (full) {
          if constexpr (sizeof(OT) == 4) *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
          else *reinterpret_cast<uint2*>(dst) = make_uint2(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]));
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (col + r < p.N) ElemIO<OT>::st(dst + r, v[r]);
        }
      }
    }
  }
  OTR_TRACE(3)
  if constexpr (!PERSIST) {
    break;
  } else {
    tile += tile_stride;
    if (tile >= ntiles) break;
    if constexpr (D != 2) {
      gemm_tile_of(tile, tiles_m, tiles_n, xmap, tile_m, tile_n);
      la.init(p.A, p.lda, p.M, p.K, tile_m * BM, p.a_vec != 0, p.cg, tid);
      lb.init(p.B, p.ldb, p.N, p.K, tile_n * bstep, p.b_vec != 0, p.cg, tid, pairF);
    }
    __syncthreads();            // epilogue staging reads / last operand reads finish before the next tile's LDS writes
    OTR_TRACE(0)
  }
  }
#undef OTR_TRACE
}

template <class CT, class AT, class BT, class OT, int AMODE, int BMODE, int BM, int BN, bool FAST, bool PERSIST = false, int EPI = 0>
__global__ __launch_bounds__(256, 2) void gemm_kernel(GemmArgs p) {   // <= 256 VGPRs: two workgroups per CU
  int tile0 = (int)blockIdx.x, kslice = (int)blockIdx.y;
  if (!PERSIST && p.xcd_map && gridDim.y >= 8) {
    // split-K: the tiles of ONE k-slice read the same slices of both operands (the frontend's weight gradient: 5 column tiles x 48
    // pixel slices moved 418 MB for 119 MB of operands) -- launched as (tile, k-slice) with the tile fastest they sat on consecutive
    // XCDs.  Same bijection as gemm_tile_of with the k-slices in the role of the row blocks: ids equal modulo 8 share their k-slices.
    gemm_tile_of((int)(blockIdx.y * gridDim.x + blockIdx.x), (int)gridDim.y, (int)gridDim.x, true, kslice, tile0);
  }
  gemm_body<CT, AT, BT, OT, AMODE, BMODE, BM, BN, FAST, PERSIST, EPI>(p, tile0, (int)gridDim.x, kslice);
}

// ------------------------------------------------------------------------------------------------
// Grouped GEMM: ONE launch runs many independent problems of the same operand types (all weight gradients of a
// backward pass: dw_i += dy_i^T x_i).  Workgroup b serves problem i with first[i] <= b < first[i+1]; no split-K
// (problem-level parallelism fills the chip), results accumulate straight into C.  The descriptors travel in the
// kernel-argument segment (captured by value in a hipGraph; no device-side table to keep in sync).
constexpr int OTR_GROUP_MAX = 48;   // descriptors per writer launch (kernel-argument segment stays < 4 KB; < 64 threads)
struct GroupDesc {
  const void* A;
  const void* B;
  void* C;
  int M, N, K;
  int lda, ldb, ldc;
  int a_vec, b_vec;
};
struct GroupedArgs {
  int n;
  int first[OTR_GROUP_MAX + 1];
  GroupDesc d[OTR_GROUP_MAX];
};
// The descriptor table lives in device memory (head of the caller's workspace) so that ONE launch can serve every
// problem of a group (>100 for a full model: with launches of <= 48 the small long-contraction problems ended up
// alone in a second, badly filled launch).  It is filled by tiny writer kernels whose kernel-argument segments carry
// OTR_GROUP_MAX descriptors each: hipGraph-capturable by value, no host staging buffer to keep alive.
static __global__ void grouped_table_write_kernel(GroupedArgs g, GroupDesc* table, int* first, int offset) {
  const int i = (int)threadIdx.x;
  if (i < g.n) {
    table[offset + i] = g.d[i];
    first[offset + i] = g.first[i];
  }
  if (i == g.n) first[offset + i] = g.first[i];      // running end marker; overwritten by the next chunk's entry 0
}
template <class CT, class AT, class BT, class OT, int AMODE, int BMODE, int BM, int BN, bool FAST>
__global__ __launch_bounds__(256, 2) void gemm_grouped_kernel(const GroupDesc* __restrict__ table, const int* __restrict__ first,
                                                              int n) {
  const int b = (int)blockIdx.x;
  int lo = 0, hi = n - 1;                              // last problem whose first block <= b (uniform: scalar loads)
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (first[mid] <= b) lo = mid; else hi = mid - 1;
  }
  const GroupDesc d = table[lo];
  GemmArgs p{};
  p.A = d.A; p.B = d.B; p.C = d.C; p.bias = nullptr;   // (accessed through ld_global / st_global: see common.h)
  p.M = d.M; p.N = d.N; p.K = d.K;
  p.lda = d.lda; p.ldb = d.ldb; p.ldc = d.ldc;
  p.act = OTR_ACT_NONE; p.accumulate = 1;
  p.a_vec = d.a_vec; p.b_vec = d.b_vec;
  p.ksplit = 1; p.allow_split = 0; p.ws = nullptr; p.ws_bytes = 0; p.trace = nullptr;
  gemm_body<CT, AT, BT, OT, AMODE, BMODE, BM, BN, FAST, false>(p, b - first[lo], 1 << 30, 0);
}

// C = act( sum_s ws[s] + bias ) (+ C): the fixed-order (deterministic) second half of split-K
template <class OT>
static __global__ void splitk_reduce_kernel(const float* ws, int ks, int M, int N, OT* C, int64_t ldc, const float* bias,
                                            int act, int accumulate) {
  const int64_t total = (int64_t)M * N, slab = total;
  const bool v4 = (N % 4 == 0) && (ldc % 4 == 0) && ((uintptr_t)C % 16 == 0);
  if (v4) {
    for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < total; i += (int64_t)gridDim.x * blockDim.x * 4) {
      float4 a = *reinterpret_cast<const float4*>(ws + i);
      int s = 1;
      for (; s + 8 <= ks; s += 8) {       // eight slabs in flight (one at a time the loop was a chain of memory latencies: 26 us for
        float4 b[8];                      // the 64 slabs of the conv2 weight gradient); the additions keep their fixed order
#pragma unroll
        for (int j = 0; j < 8; ++j) b[j] = *reinterpret_cast<const float4*>(ws + (int64_t)(s + j) * slab + i);
#pragma unroll
        for (int j = 0; j < 8; ++j) { a.x += b[j].x; a.y += b[j].y; a.z += b[j].z; a.w += b[j].w; }
      }
      for (; s < ks; ++s) {
        float4 b = *reinterpret_cast<const float4*>(ws + (int64_t)s * slab + i);
        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
      }
      int64_t r = i / N;
      int c = (int)(i - r * N);
      float v[4] = {a.x, a.y, a.z, a.w};
      OT* dst = C + r * ldc + c;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (bias) v[e] += bias[c + e];
        if (accumulate) v[e] += ElemIO<OT>::ld(dst + e);
        if (act == OTR_ACT_RELU) v[e] = fmaxf(v[e], 0.f);
      }
      if constexpr (sizeof(OT) == 4) *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
      else *reinterpret_cast<uint2*>(dst) = make_uint2(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]));
    }
  } else {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
      float a = ws[i];
      for (int s = 1; s < ks; ++s) a += ws[s * slab + i];
      int64_t r = i / N;
      int c = (int)(i - r * N);
      OT* dst = C + r * ldc + c;
      if (bias) a += bias[c];
      if (accumulate) a += ElemIO<OT>::ld(dst);
      if (act == OTR_ACT_RELU) a = fmaxf(a, 0.f);
      ElemIO<OT>::st(dst, a);
    }
  }
}

// nbatch problems of one shape side by side (the per-head GEMMs of the relative-position attention, module/attention.py:217-253:
// four launches of 11-16 us each were four dependent launch boundaries for one head's worth of work each)
template <class CT, class AT, class BT, class OT, int BM, int BN>
__global__ __launch_bounds__(256, 2) void gemm_batched_kernel(GemmArgs p) {
  GemmArgs q = p;
  const int64_t z = blockIdx.z;
  q.A = reinterpret_cast<const char*>(p.A) + z * p.bsa;
  q.B = reinterpret_cast<const char*>(p.B) + z * p.bsb;
  q.C = reinterpret_cast<char*>(p.C) + z * p.bsc;
  gemm_body<CT, AT, BT, OT, MODE_KC, MODE_KC, BM, BN, true, false, 0>(q, (int)blockIdx.x, (int)gridDim.x, 0);
}

// ------------------------------------------------------------------------------------------------
// Tile / split-K selection.  Goal: >= ~512 workgroups (2 per CU) whenever the problem allows it; split-K
// only when the output grid alone cannot fill the chip and a workspace was provided.
extern int g_otr_force_tile;    // 0 = heuristic, 64 / 128 = forced (tuning hook: otr_debug_set(0, v))
extern int g_otr_force_ksplit;  // 0 = heuristic, n = forced                  (otr_debug_set(1, v))
extern int g_otr_force_generic; // 1 = never use the branch-free FAST loaders  (otr_debug_set(2, v))
extern int g_otr_no_persist;    // 1 = one workgroup per tile even without split-K (otr_debug_set(3, v))
extern int g_otr_gemm_resident64;      // api.hip (otr_debug_set(28, v))
extern int g_otr_im2k_fast;            // api.hip (otr_debug_set(34, v)): conv2 forward's implicit-im2col loader on unconditional loads
constexpr int OTR_RESIDENT_WG = 512;   // 256 CUs x 2 workgroups (launch_bounds(256, 2), 64 KB LDS each)

template <class CT, class AT, class BT, class OT, int AMODE, int BMODE>
static int32_t gemm_launch_tiles(GemmArgs a, hipStream_t s) {
  constexpr int BK = GemmCfg<CT>::BK;
  const int nk = (a.K + BK - 1) / BK;
  const int64_t slab_bytes = (int64_t)a.M * a.N * 4;
  const int max_by_ws = (a.ws && slab_bytes > 0) ? (int)(a.ws_bytes / slab_bytes) : 0;
  const bool can_split = a.allow_split && max_by_ws >= 2;
  const int64_t t128 = (int64_t)((a.M + 127) / 128) * ((a.N + 127) / 128);
  const int64_t t64 = (int64_t)((a.M + 63) / 64) * ((a.N + 63) / 64);
  if (a.nbatch > 1) {
    // batched: FAST KC/KC operands, plain epilogue, no split-K, for the operand types the callers use
    constexpr bool BUILT = AMODE == MODE_KC && BMODE == MODE_KC && std::is_same<CT, bf16_t>::value && std::is_same<BT, bf16_t>::value &&
                           ((std::is_same<AT, bf16_t>::value && std::is_same<OT, float>::value) ||
                            (std::is_same<AT, bf16_t>::value && std::is_same<OT, bf16_t>::value) ||      // r06: pt_i = W_i pe^T of all Conformer blocks
                            (std::is_same<AT, float>::value && std::is_same<OT, bf16_t>::value));
    if constexpr (BUILT) {
      constexpr int CE = GemmCfg<CT>::CE;
      const bool fast = a.a_vec && a.b_vec && (a.K % CE == 0) && ((uintptr_t)a.C % 16 == 0) && (a.ldc % (16 / (int)sizeof(OT)) == 0) &&
                        (a.bsa % 16 == 0) && (a.bsb % 16 == 0) && (a.bsc % 16 == 0) && !a.accumulate && !a.bias && a.act == OTR_ACT_NONE;
      if (!fast) return 1;                                            // not served: the caller loops over otr_linear_fwd
      // (N >= 96: d(q+v)_h = dbd_h p_h reads its fp32 operand once per column tile -- 96 columns are one 128-wide tile or two 64-wide ones)
      const bool big = a.M >= 128 && a.N >= 96 && t128 * a.nbatch >= 240;
      a.ksplit = 1;
      if (big) hipLaunchKernelGGL((gemm_batched_kernel<CT, AT, BT, OT, 128, 128>), dim3((unsigned)t128, 1, (unsigned)a.nbatch), dim3(256), 0, s, a);
      else hipLaunchKernelGGL((gemm_batched_kernel<CT, AT, BT, OT, 64, 64>), dim3((unsigned)t64, 1, (unsigned)a.nbatch), dim3(256), 0, s, a);
      return otr_check_launch("gemm(batched)");
    } else {
      return 1;
    }
  }
  auto splits_for = [&](int64_t tiles) {
    if (!can_split) return 1;
    // measured (tools/gemm_sweep.py): with >= 64 output tiles one k-slice per CU is enough (w_2 forward 7968x256x2048:
    // 35 us at 5 slices, 28 us at 2); tiny outputs need the second resident workgroup per CU as well
    // a short contraction (K <= 768: <= 24 stages) is over in a few microseconds whether split or not, and the reduce
    // launch that a split needs costs 5-7 us: 23 of them per training step bought nothing (r02 kernel trace)
    if (nk <= 24 && g_otr_force_ksplit == 0) return 1;
    int64_t want = tiles >= 64 ? (256 + tiles / 2) / tiles : (512 + tiles - 1) / tiles;   // >= 64 tiles: ~one wave of CUs
    int64_t cap = nk / 4 > 0 ? nk / 4 : 1;
    if (cap > max_by_ws) cap = max_by_ws;
    return (int)(want < cap ? want : cap);
  };
  bool big = a.M >= 128 && a.N >= 128;
  int ks = 1;
  if (big && t128 >= 256) {
    ks = 1;
  } else if (big && t128 * splits_for(t128) >= 192) {   // (126 tiles x 2 slices = 252 workgroups is a full wave of CUs; r05: with 189
    // -- the Conformer's 7968 x 384 outputs -- on 128-wide tiles the fp32-writing GEMMs gain what the 16-bit-writing ones lose: 12.67 vs 12.59 ms)
    ks = splits_for(t128);
  } else {
    big = false;
    ks = t64 >= 256 ? 1 : splits_for(t64);
  }
  if (g_otr_force_tile == 64) big = false;
  if (g_otr_force_tile == 128) big = true;
  if (g_otr_force_tile != 0) ks = splits_for(big ? t128 : t64);
  if (g_otr_force_ksplit > 0 && can_split) {
    ks = g_otr_force_ksplit < nk ? g_otr_force_ksplit : nk;
    if (ks > max_by_ws) ks = max_by_ws;
  }
  if (ks < 1) ks = 1;
  {  // every k-slice must own at least one stage: an empty slice would leave its workspace slab unwritten
    int per = (nk + ks - 1) / ks;
    ks = (nk + per - 1) / per;
  }
  a.ksplit = ks;
  a.xcd_map = g_otr_gemm_xcd_map;
  // branch-free loaders need aligned rows and whole chunks / whole row groups (see TileLoader)
  constexpr int CE = GemmCfg<CT>::CE, PM = GemmCfg<CT>::PM;
  auto side_fast = [&](int mode, int vec, int rows) {
    if (mode == MODE_KC) return vec && (a.K % CE == 0) && rows > 0;
    if (mode == MODE_MC) return vec && (rows % PM == 0) && (a.K % CE == 0);
    if (mode == MODE_IM2K)   // raw 16-bit path only (same-type operands); rows = output pixels: any count
      return sizeof(CT) == 2 && std::is_same<AT, CT>::value && vec && (a.K % CE == 0) && (a.cg.C1 % CE == 0) && a.cg.a1_elems >= CE && rows > 0 &&
             g_otr_im2k_fast != 0;
    if (mode == MODE_IM2M)   // raw bf16 path only (same-type operands)
      return sizeof(CT) == 2 && std::is_same<BT, CT>::value && vec && (rows % PM == 0) && (a.K % CE == 0) && (a.cg.C1 % PM == 0) &&
             a.cg.a1_elems >= PM;
    return false;
  };
  const bool fast = (AMODE == MODE_KC || AMODE == MODE_MC || AMODE == MODE_IM2K) && (BMODE == MODE_KC || BMODE == MODE_MC || BMODE == MODE_IM2M) &&
                    side_fast(AMODE, a.a_vec, a.M) && side_fast(BMODE, a.b_vec, a.N) && g_otr_force_generic == 0 &&
                    (a.ksplit > 1 ||          // split-K slabs go to the workspace; C is written by the reduce kernel
                     (((uintptr_t)a.C % 16 == 0) && (a.ldc % (16 / (int)sizeof(OT)) == 0) && !(a.accumulate && sizeof(OT) == 2)));
  const int64_t ntiles = big ? t128 : t64;
  // persistent variant (no split-K): at most OTR_RESIDENT_WG workgroups, each walking tiles b, b+grid, ...
  // (transposing operands: the cross-tile prefetch on top of their ring does not fit 256 VGPRs -- spills -- so they
  //  keep one workgroup per tile)
  constexpr bool CAN_PERSIST = (AMODE == MODE_KC && BMODE == MODE_KC);
  const bool persist = CAN_PERSIST && fast && a.ksplit == 1 && g_otr_no_persist == 0;
  // (64-wide tiles: 106 VGPRs and 32 KB of LDS -- FOUR workgroups fit a CU, so up to g_otr_gemm_resident64 = 1024 of them are resident:
  //  the Conformer's 7968 x 384 outputs are 750 tiles, which 512 workgroups walked as two rounds with the second one half empty)
  const int64_t resident = big ? OTR_RESIDENT_WG : g_otr_gemm_resident64;
  const dim3 grid((unsigned)(persist && ntiles > resident ? resident : ntiles), a.ksplit);
  if (a.act == OTR_ACT_GLU_FWD || a.act == OTR_ACT_GLU_BWD) {   // fused FFN epilogues: own (non-persistent) instantiations
    if constexpr (CAN_PERSIST && std::is_same<CT, bf16_t>::value && std::is_same<AT, bf16_t>::value &&
                  std::is_same<BT, bf16_t>::value && std::is_same<OT, bf16_t>::value) {
      if (!fast || a.ksplit != 1) {
        otr_set_error("gemm: fused GLU epilogue needs the fast path without split-K");
        return -1;
      }
      const dim3 g1((unsigned)ntiles, 1);
      if (a.act == OTR_ACT_GLU_FWD) {
        if (big) hipLaunchKernelGGL((gemm_kernel<CT, AT, BT, OT, AMODE, BMODE, 128, 128, true, false, 1>), g1, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((gemm_kernel<CT, AT, BT, OT, AMODE, BMODE, 64, 64, true, false, 1>), g1, dim3(256), 0, s, a);
      } else {
        if (big) hipLaunchKernelGGL((gemm_kernel<CT, AT, BT, OT, AMODE, BMODE, 128, 128, true, false, 2>), g1, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((gemm_kernel<CT, AT, BT, OT, AMODE, BMODE, 64, 64, true, false, 2>), g1, dim3(256), 0, s, a);
      }
      return otr_check_launch("gemm(glu)");
    } else {
      otr_set_error("gemm: fused GLU epilogue is built for bf16 KC/KC operands only");
      return -1;
    }
  }
  if constexpr (AMODE == MODE_KC || AMODE == MODE_MC || (AMODE == MODE_IM2K && sizeof(CT) == 2 && std::is_same<AT, CT>::value)) {
    if constexpr (BMODE == MODE_KC || BMODE == MODE_MC || BMODE == MODE_IM2M) {
      if (fast && persist) {
        if constexpr (CAN_PERSIST) {
          if (big) hipLaunchKernelGGL((gemm_kernel<CT, AT, BT, OT, AMODE, BMODE, 128, 128, true, true>), grid, dim3(256), 0, s, a);
          else hipLaunchKernelGGL((gemm_kernel<CT, AT, BT, OT, AMODE, BMODE, 64, 64, true, true>), grid, dim3(256), 0, s, a);
        }
      } else if (fast) {
        if (big) hipLaunchKernelGGL((gemm_kernel<CT, AT, BT, OT, AMODE, BMODE, 128, 128, true>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((gemm_kernel<CT, AT, BT, OT, AMODE, BMODE, 64, 64, true>), grid, dim3(256), 0, s, a);
      }
    }
  }
  if (!fast) {
    if (big) hipLaunchKernelGGL((gemm_kernel<CT, AT, BT, OT, AMODE, BMODE, 128, 128, false>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((gemm_kernel<CT, AT, BT, OT, AMODE, BMODE, 64, 64, false>), grid, dim3(256), 0, s, a);
  }
  if (a.ksplit > 1) {
    int64_t total = (int64_t)a.M * a.N;
    unsigned g = (unsigned)((total / 4 + 255) / 256 > 2048 ? 2048 : (total / 4 + 255) / 256);
    if (g < 1) g = 1;
    hipLaunchKernelGGL((splitk_reduce_kernel<OT>), dim3(g), dim3(256), 0, s, a.ws, a.ksplit, a.M, a.N, (OT*)a.C, a.ldc,
                       a.bias, a.act, a.accumulate);
  }
  return otr_check_launch("gemm");
}

// Launch ONE grouped kernel over n descriptors that all satisfy the FAST-loader conditions.  `table_mem` (device,
// table_bytes) receives the descriptor table; if it is too small the problems go in several launches.
static inline int64_t grouped_table_bytes(int n) { return (int64_t)n * (int64_t)sizeof(GroupDesc) + ((int64_t)n + 1) * 4 + 64; }
template <class CT, class AT, class BT, int BM, int BN>
static int32_t gemm_grouped_launch(const GroupDesc* d, int n, void* table_mem, int64_t table_bytes, hipStream_t s) {
  int cap = n;
  while (cap > 1 && grouped_table_bytes(cap) > table_bytes) cap /= 2;
  if (grouped_table_bytes(cap) > table_bytes) {
    otr_set_error("grouped gemm: workspace of %lld bytes cannot hold a descriptor table", (long long)table_bytes);
    return -1;
  }
  for (int base = 0; base < n; base += cap) {
    const int m = (n - base < cap) ? n - base : cap;
    GroupDesc* table = reinterpret_cast<GroupDesc*>(table_mem);
    int* first = reinterpret_cast<int*>(reinterpret_cast<unsigned char*>(table_mem) + (((int64_t)m * sizeof(GroupDesc) + 63) / 64) * 64);
    int blocks = 0;
    for (int c0 = 0; c0 < m; c0 += OTR_GROUP_MAX) {
      GroupedArgs g{};
      g.n = (m - c0 < OTR_GROUP_MAX) ? m - c0 : OTR_GROUP_MAX;
      for (int i = 0; i < g.n; ++i) {
        const GroupDesc& di = d[base + c0 + i];
        g.first[i] = blocks;
        g.d[i] = di;
        blocks += ((di.M + BM - 1) / BM) * ((di.N + BN - 1) / BN);
      }
      g.first[g.n] = blocks;
      hipLaunchKernelGGL(grouped_table_write_kernel, dim3(1), dim3(64), 0, s, g, table, first, c0);
    }
    if (blocks == 0) continue;
    hipLaunchKernelGGL((gemm_grouped_kernel<CT, AT, BT, float, MODE_MC, MODE_MC, BM, BN, true>), dim3((unsigned)blocks), dim3(256),
                       0, s, table, first, m);
  }
  return otr_check_launch("gemm_grouped");
}
int32_t gemm_grouped_wgrad_bf16(const GroupDesc* d, int n, int a_dtype, int b_dtype, int big, void* table_mem, int64_t table_bytes,
                                hipStream_t s);
int32_t gemm_grouped_wgrad_f32(const GroupDesc* d, int n, int a_dtype, int b_dtype, int big, void* table_mem, int64_t table_bytes,
                               hipStream_t s);

// dtype dispatch helpers implemented in gemm_bf16.hip / gemm_f32.hip
int32_t gemm_dispatch_bf16(const GemmArgs& a, int a_dtype, int b_dtype, int c_dtype, int amode, int bmode, hipStream_t s);
int32_t gemm_dispatch_f32(const GemmArgs& a, int a_dtype, int b_dtype, int c_dtype, int amode, int bmode, hipStream_t s);
