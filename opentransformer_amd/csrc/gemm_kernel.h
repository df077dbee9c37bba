// MFMA GEMM core:  C[M,N] = act( sum_k A(m,k) * B(n,k) + bias[n] ) (+ C)
//
// One kernel template serves nn.Linear forward / dgrad / wgrad and the conv2 implicit GEMMs by
// changing how each operand tile is gathered from HBM ("mode"); the LDS image and the MFMA loop are
// always the same: tile rows x 64 bytes (4 chunks of 16 B = one MFMA k-step), XOR-swizzled so the
// 16-lane ds_read_b128 operand reads are bank-conflict free.
//
//   MODE_KC      operand stored [rows][K], K contiguous            (x and w in y = x w^T)
//   MODE_MC      operand stored [K][rows], rows contiguous         (w in dgrad; dy and x in wgrad)
//                -> each thread gathers a 2(rows) x CE(k) patch and transposes it in registers
//   MODE_IM2K    rows = conv2 output pixels, k = (kh,kw,c1) gathered from channel-last act1
//   MODE_IM2M    rows = (kh,kw,c1), contraction = output pixels   (conv2 wgrad B operand)
//
// Pipeline: double-buffered LDS, global loads for tile t+1 are issued before the MFMAs of tile t
// and written to the other buffer afterwards -> one barrier per k-step.
// Workgroup = 4 waves in a 2x2 grid; wave tile (BM/2)x(BN/2) built from 16x16 MFMA tiles.
#pragma once
#include <type_traits>

#include "common.h"

enum { MODE_KC = 0, MODE_MC = 1, MODE_IM2K = 2, MODE_IM2M = 3 };

struct ConvGeom {
  int T1, F1, C1, T2, F2;
  FastDiv divF2, divT2, divC1;
};

struct GemmArgs {
  const void* A;
  const void* B;
  void* C;
  const float* bias;
  int M, N, K;
  int64_t lda, ldb, ldc;
  int act, accumulate;
  int a_vec, b_vec;  // operand base/ld satisfy the vector-load alignment
  ConvGeom cg;
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

// ------------------------------------------------------------------------------------------------
// Operand tile loader.  ROWS x (4 chunks) per k-step, 256 threads.
template <class CT, class ST, int MODE, int ROWS> struct TileLoader {
  static constexpr int CE = MMA<CT>::CE;
  static constexpr bool ROWMAJOR = (MODE == MODE_KC || MODE == MODE_IM2K);
  // row-major modes: NU chunk units per thread (unit u -> row = id>>2, chunk = id&3, id = tid+256u)
  static constexpr int NU = ROWMAJOR ? (ROWS * 4 + 255) / 256 : 1;
  // transposing modes: one (row pair, chunk) unit per thread: rp = tid % (ROWS/2), c = tid / (ROWS/2)

  const ST* base;
  int64_t ld;
  int nrows, K, row0;
  bool vec;
  float raw[ROWMAJOR ? NU : 2][CE];
  // im2col state
  int64_t pix[ROWMAJOR ? NU : 1];
  int f2v[ROWMAJOR ? NU : 1];
  int tapoff, tapkw;

  __device__ __forceinline__ void init(const void* p, int64_t ld_, int nrows_, int K_, int row0_, bool vec_,
                                       const ConvGeom& g, int tid) {
    base = reinterpret_cast<const ST*>(p);
    ld = ld_; nrows = nrows_; K = K_; row0 = row0_; vec = vec_;
    if constexpr (MODE == MODE_IM2K) {
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        int row = row0 + ((tid + 256 * u) >> 2);
        pix[u] = (row < nrows) ? im2col_base(g, (uint32_t)row, f2v[u]) : 0;
      }
    }
    if constexpr (MODE == MODE_IM2M) {
      int rp = tid % (ROWS / 2);
      int n0 = row0 + 2 * rp;              // row index = tap*C1 + c1
      uint32_t tap = fdiv((uint32_t)n0, g.divC1);
      int ch = n0 - (int)tap * g.C1;
      int kh = (int)tap / 3, kw = (int)tap - kh * 3;
      tapoff = (kh * g.F1 + kw) * g.C1 + ch;
      tapkw = kw;
    }
  }

  // issue the global loads for the k-step starting at k0
  __device__ __forceinline__ void load(int k0, const ConvGeom& g, int tid) {
    if constexpr (MODE == MODE_KC) {
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        int id = tid + 256 * u;
        int row = row0 + (id >> 2), gk = k0 + (id & 3) * CE;
        if ((id >> 2) < ROWS && row < nrows && gk < K) {
          load_row<ST, CE>(base + (int64_t)row * ld + gk, K - gk, vec, raw[u]);
        } else {
#pragma unroll
          for (int e = 0; e < CE; ++e) raw[u][e] = 0.f;
        }
      }
    } else if constexpr (MODE == MODE_IM2K) {
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        int id = tid + 256 * u;
        int row = row0 + (id >> 2), gk = k0 + (id & 3) * CE;
        uint32_t tap = fdiv((uint32_t)gk, g.divC1);
        int ch = gk - (int)tap * g.C1;
        int kh = (int)tap / 3, kw = (int)tap - kh * 3;
        int fin = 2 * f2v[u] + kw - 1;
        if ((id >> 2) < ROWS && row < nrows && gk < K && fin >= 0 && fin < g.F1) {
          load_row<ST, CE>(base + pix[u] + (int64_t)(kh * g.F1 + kw) * g.C1 + ch, CE, vec, raw[u]);
        } else {
#pragma unroll
          for (int e = 0; e < CE; ++e) raw[u][e] = 0.f;
        }
      }
    } else {
      constexpr int RP = ROWS / 2;
      int rp = tid % RP, c = tid / RP;
      int r = row0 + 2 * rp;
      bool active = c < 4;
#pragma unroll
      for (int j = 0; j < CE; ++j) {
        int gk = k0 + c * CE + j;
        float v0 = 0.f, v1 = 0.f;
        if (active && gk < K) {
          if constexpr (MODE == MODE_MC) {
            const ST* p = base + (int64_t)gk * ld + r;
            if (vec && r + 1 < nrows) {
              if constexpr (sizeof(ST) == 4) {
                float2 t = *reinterpret_cast<const float2*>(p);
                v0 = t.x; v1 = t.y;
              } else {
                uint32_t t = *reinterpret_cast<const uint32_t*>(p);
                v0 = __uint_as_float(t << 16); v1 = __uint_as_float(t & 0xffff0000u);
              }
            } else {
              if (r < nrows) v0 = ElemIO<ST>::ld(p);
              if (r + 1 < nrows) v1 = ElemIO<ST>::ld(p + 1);
            }
          } else {  // MODE_IM2M: contraction index gk = output pixel
            int f2;
            int64_t pb = im2col_base(g, (uint32_t)gk, f2);
            int fin = 2 * f2 + tapkw - 1;
            if (fin >= 0 && fin < g.F1) {
              const ST* p = base + pb + tapoff;
              if (r < nrows) v0 = ElemIO<ST>::ld(p);
              if (r + 1 < nrows) v1 = ElemIO<ST>::ld(p + 1);
            }
          }
        }
        raw[0][j] = v0;
        raw[1][j] = v1;
      }
    }
  }

  // convert to CT and write the tile image: byte(row, chunk) = row*64 + ((chunk ^ swz(row)) << 4)
  __device__ __forceinline__ void store(unsigned char* lds, int tid) const {
    if constexpr (ROWMAJOR) {
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        int id = tid + 256 * u;
        int row = id >> 2, c = id & 3;
        if (row < ROWS) *reinterpret_cast<uint4*>(lds + row * 64 + ((c ^ swz<4>(row)) << 4)) = MMA<CT>::pack(raw[u]);
      }
    } else {
      constexpr int RP = ROWS / 2;
      int rp = tid % RP, c = tid / RP;
      if (c < 4) {
        int row = 2 * rp;
        *reinterpret_cast<uint4*>(lds + row * 64 + ((c ^ swz<4>(row)) << 4)) = MMA<CT>::pack(raw[0]);
        *reinterpret_cast<uint4*>(lds + (row + 1) * 64 + ((c ^ swz<4>(row + 1)) << 4)) = MMA<CT>::pack(raw[1]);
      }
    }
  }
};

// ------------------------------------------------------------------------------------------------
template <class CT, class AT, class BT, class OT, int AMODE, int BMODE, int BM, int BN>
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs p) {
  constexpr int CE = MMA<CT>::CE;
  constexpr int BK = 4 * CE;
  constexpr int WM = BM / 2, WN = BN / 2, FM = WM / 16, FN = WN / 16;
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * (BM + BN) * 64];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int tiles_n = (p.N + BN - 1) / BN;
  const int tile_m = blockIdx.x / tiles_n, tile_n = blockIdx.x - tile_m * tiles_n;

  TileLoader<CT, AT, AMODE, BM> la;
  TileLoader<CT, BT, BMODE, BN> lb;
  la.init(p.A, p.lda, p.M, p.K, tile_m * BM, p.a_vec != 0, p.cg, tid);
  lb.init(p.B, p.ldb, p.N, p.K, tile_n * BN, p.b_vec != 0, p.cg, tid);

  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  constexpr int BUF = (BM + BN) * 64;  // bytes per pipeline stage: A tile then B tile

  const int nk = (p.K + BK - 1) / BK;
  la.load(0, p.cg, tid);
  lb.load(0, p.cg, tid);
  la.store(smem, tid);
  lb.store(smem + BM * 64, tid);
  __syncthreads();

  const int fr = lane & 15, fg = lane >> 4;
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    const bool more = kt + 1 < nk;
    if (more) {
      la.load((kt + 1) * BK, p.cg, tid);
      lb.load((kt + 1) * BK, p.cg, tid);
    }
    uint4 af[FM], bf[FN];
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      int row = wm * WM + i * 16 + fr;
      af[i] = *reinterpret_cast<const uint4*>(smem + cur * BUF + row * 64 + ((fg ^ swz<4>(row)) << 4));
    }
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      int row = wn * WN + j * 16 + fr;
      bf[j] = *reinterpret_cast<const uint4*>(smem + cur * BUF + BM * 64 + row * 64 + ((fg ^ swz<4>(row)) << 4));
    }
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) MMA<CT>::mma(acc[i][j], af[i], bf[j]);
    if (more) {
      la.store(smem + (cur ^ 1) * BUF, tid);
      lb.store(smem + (cur ^ 1) * BUF + BM * 64, tid);
    }
    __syncthreads();
  }

  // epilogue: C layout col = lane&15, row = (lane>>4)*4 + r
  OT* C = reinterpret_cast<OT*>(p.C);
#pragma unroll
  for (int i = 0; i < FM; ++i) {
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      int col = tile_n * BN + wn * WN + j * 16 + fr;
      if (col >= p.N) continue;
      float bv = p.bias ? p.bias[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int row = tile_m * BM + wm * WM + i * 16 + fg * 4 + r;
        if (row >= p.M) continue;
        float v = acc[i][j][r] + bv;
        OT* dst = C + (int64_t)row * p.ldc + col;
        if (p.accumulate) v += ElemIO<OT>::ld(dst);
        if (p.act == OTR_ACT_RELU) v = fmaxf(v, 0.f);
        ElemIO<OT>::st(dst, v);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
template <class CT, class AT, class BT, class OT, int AMODE, int BMODE>
static int32_t gemm_launch_tiles(const GemmArgs& a, hipStream_t s) {
  int64_t blocks128 = (int64_t)((a.M + 127) / 128) * ((a.N + 127) / 128);
  if (blocks128 >= 256) {
    hipLaunchKernelGGL((gemm_kernel<CT, AT, BT, OT, AMODE, BMODE, 128, 128>), dim3((unsigned)blocks128), dim3(256), 0, s, a);
  } else {
    int64_t blocks64 = (int64_t)((a.M + 63) / 64) * ((a.N + 63) / 64);
    hipLaunchKernelGGL((gemm_kernel<CT, AT, BT, OT, AMODE, BMODE, 64, 64>), dim3((unsigned)blocks64), dim3(256), 0, s, a);
  }
  return otr_check_launch("gemm");
}

// dtype dispatch helpers implemented in gemm_bf16.hip / gemm_f32.hip
int32_t gemm_dispatch_bf16(const GemmArgs& a, int a_dtype, int b_dtype, int c_dtype, int amode, int bmode, hipStream_t s);
int32_t gemm_dispatch_f32(const GemmArgs& a, int a_dtype, int b_dtype, int c_dtype, int amode, int bmode, hipStream_t s);
