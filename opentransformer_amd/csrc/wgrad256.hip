// Weight gradients of the wide 16-bit Linears, dw[N,K] += dy[M,N]^T x[M,K]  (train/trainer.py:208 loss.backward(): the
// aten mm_backward of module/ffn.py:38-41, module/attention.py:62-75), as ONE persistent launch over every qualifying
// problem of a backward pass.  Reached through otr_linear_wgrad_grouped (api.hip).
//
// Why a second kernel next to gemm_grouped_kernel: these problems have a long contraction (M = batch x frames = 7968)
// and a small output, both operands are contraction-major ([m][cols] row-major), and together they stream 2 GB for
// 377 GFLOP -- 189 flop/B, below the 312 flop/B ridge of the chip: the launch is HBM-bound (floor 250 us at 8 TB/s).
// The 128x128-tile kernel needs 64 flop per byte a CU ingests and tops out on the CU's vector-memory return path
// (22-33 B/clk) at 550 TFLOP/s = 690 us.  Here:
//  * 256 x 256 output tile per workgroup of 8 waves (2 x 4, 128 x 64 each: 4 x 2 accumulators of v_mfma_f32_32x32x16):
//    128 flop per ingested byte.
//  * operands go global -> LDS by direct-to-LDS DMA (global_load_lds_dwordx4, 8 rows x 128 B per wave instruction), as a
//    ring of eight 16-row slabs, staged six slabs ahead; waits are counted (vmcnt(8)), never zero, in the steady state.
//  * the LDS image of a slab is made of [8 rows][32 columns] subtiles (512 B, exactly what ONE ds_read_b64_tr_b16 of a
//    wave covers): the transposing read hands every lane 4 consecutive contraction rows of one column, two reads make
//    one MFMA operand, both operands are read the same way so the contraction order agrees; no swizzle, no VALU.
//  * fragments of slab q+1 are read while the MFMAs of slab q run (two register sets); one barrier per slab.
//  * work is cut stream-K style: the (tile, row) space of all problems is divided into gridDim.x equal chunks of whole
//    16-row slabs; a tile that straddles chunks is accumulated by its pieces one after the other (a turnstile per tile:
//    flag == my slice index; payload and flag travel with sc0 sc1 accesses, cdna_hip_programming.md G16 valid forms).
//    The piece that finishes first in time (the head of the later chunk) goes first, so nobody waits for work that has
//    not started; the order is fixed, the sum is deterministic.
#include "common.h"
#include "wgrad256.h"

#include <algorithm>

extern int g_otr_spin_limit;       // api.hip: bound of the turnstile spin (otr_debug_set(11, v))
extern int32_t* g_otr_fault;       // api.hip: sticky device fault word (otr_set_fault_counter) or NULL

namespace {

typedef __attribute__((address_space(3))) unsigned char lds_byte;
typedef short v4s __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) v4s lds_v4s;

constexpr int SLAB_ROWS = 16;
constexpr int SLAB_BYTES = 16384;                 // 16 rows x 256 columns x 2 B, dy part then x part
constexpr int RING = 8, AHEAD = 6;
constexpr int LDS_BYTES = RING * SLAB_BYTES;      // 128 KB; after a piece's last slab the ring doubles as the epilogue's scratch (16 KB per wave)

// one wave instruction: 64 lanes x 16 B, per-lane global source -> LDS [dst, dst + 1024) lane-linear.  Inline asm so
// that hipcc does not see a pending LDS write (it would wait vmcnt(0) before the next ds_read, ffn_fused.hip).
__device__ __forceinline__ void dma16(const void* src, uint32_t lds_dst) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(src), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void dma16_nt(const void* src, uint32_t lds_dst) {     // non-temporal: a stream no other tile reads
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(src), "s"(lds_dst) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

__device__ __forceinline__ uint2 tr_read(uint32_t lds_addr) {
  v4s r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s*)(uintptr_t)lds_addr);
  return __builtin_bit_cast(uint2, r);
}

struct Frags {
  uint4 a[4], b[2];
};

// the wave's operands of one slab: 4 dy fragments (32 columns each) + 2 x fragments, 12 transposing reads, in two
// halves so that they can be spread between the MFMAs of the slab before
__device__ __forceinline__ void read_frags_lo(Frags& f, uint32_t slab_addr, uint32_t a_off, uint32_t b_off) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const uint2 lo = tr_read(slab_addr + b_off + i * 512), hi = tr_read(slab_addr + b_off + 4096 + i * 512);
    f.b[i] = make_uint4(lo.x, lo.y, hi.x, hi.y);
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const uint2 lo = tr_read(slab_addr + a_off + i * 512), hi = tr_read(slab_addr + a_off + 4096 + i * 512);
    f.a[i] = make_uint4(lo.x, lo.y, hi.x, hi.y);
  }
}
__device__ __forceinline__ void read_frags_hi(Frags& f, uint32_t slab_addr, uint32_t a_off, uint32_t b_off) {
#pragma unroll
  for (int i = 2; i < 4; ++i) {
    const uint2 lo = tr_read(slab_addr + a_off + i * 512), hi = tr_read(slab_addr + a_off + 4096 + i * 512);
    f.a[i] = make_uint4(lo.x, lo.y, hi.x, hi.y);
  }
}

// column sums of the dy operand for the bias gradient: lane (column l & 31, rows 8 (l >> 5) ..) adds the 8 values of each of
// its 4 fragments with v_dot2c against ones (16 VALU per slab, on the two waves of a workgroup that own wc == 0)
__device__ __forceinline__ void frag_colsum(float (&cs)[4], const Frags& f) {
#ifdef OTR_HALF_FP16
  typedef _Float16 hv2 __attribute__((ext_vector_type(2)));
  const hv2 one = {(_Float16)1.f, (_Float16)1.f};
#define W256_DOT(w, c) __builtin_amdgcn_fdot2(__builtin_bit_cast(hv2, w), one, c, false)
#else
  typedef __bf16 hv2 __attribute__((ext_vector_type(2)));
  const hv2 one = {(__bf16)1.f, (__bf16)1.f};
#define W256_DOT(w, c) __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(hv2, w), one, c, false)
#endif
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    cs[a] = W256_DOT(f.a[a].x, cs[a]);
    cs[a] = W256_DOT(f.a[a].y, cs[a]);
    cs[a] = W256_DOT(f.a[a].z, cs[a]);
    cs[a] = W256_DOT(f.a[a].w, cs[a]);
  }
#undef W256_DOT
}

struct Stager {                 // this lane's share of the two DMA instructions its wave issues per slab
  const unsigned char* base_a;  // dy: (row = lane row of slab 0, this lane's 16 bytes)
  const unsigned char* base_b;  // x
  int64_t step_a, step_b;       // bytes per 16 rows
  int row0, M;                  // lane row inside a slab; rows of the problem
  bool col_a, col_b;            // this lane's 8 columns exist (ragged last tile of N / K)
  const unsigned char* zeros;
  uint32_t dst;                 // wave's byte offset inside a slab's dy part (the x part is + 8192)
  // CONV (W256Args mode 2): base_b = act1 + this lane's 16 bytes of the tile's channels; the row comes from the output pixel
  int cT1, cF1, cT2, cF2, kh, kw;
  int64_t pix_bytes;            // C1 * 2
  FastDiv divF2, divT2;
  // NTA / NTB: the operand strip is read by this tile only -> non-temporal, it stays out of the L2 the shared strips live in
  template <bool NTA, bool NTB, bool SKIPB = false, bool CONV = false> __device__ __forceinline__ void issue(int slab, int slot) {
    const bool ok = slab * SLAB_ROWS + row0 < M;
    const unsigned char* pa = ok && col_a ? base_a + slab * step_a : zeros;
    const unsigned char* pb;
    if constexpr (CONV) {
      const uint32_t m = (uint32_t)(slab * SLAB_ROWS + row0);
      const uint32_t t = fdiv(m, divF2), f2 = m - t * (uint32_t)cF2, b = fdiv(t, divT2), t2 = t - b * (uint32_t)cT2;
      const int fin = 2 * (int)f2 + kw - 1;
      const bool in = ok && col_b && fin >= 0 && fin < cF1;
      pb = in ? base_b + (((int64_t)b * cT1 + 2 * (int)t2 + kh) * cF1 + fin) * pix_bytes : zeros;
    } else {
      pb = ok && col_b ? base_b + slab * step_b : zeros;
    }
    if constexpr (NTA) dma16_nt(pa, (uint32_t)(slot * SLAB_BYTES) + dst); else dma16(pa, (uint32_t)(slot * SLAB_BYTES) + dst);
    if constexpr (SKIPB) dma16(zeros, (uint32_t)(slot * SLAB_BYTES) + 8192u + dst);        // ablation 6: the x part is not fetched (one line, L1-hot)
    else if constexpr (NTB) dma16_nt(pb, (uint32_t)(slot * SLAB_BYTES) + 8192u + dst);
    else dma16(pb, (uint32_t)(slot * SLAB_BYTES) + 8192u + dst);
  }
};

// One piece = slabs [sb, sb + P) of one tile, walked in the rotated order sb + (i + rot) % P (a sum: any order is right;
// the rotation staggers the problems of a launch so that their pipeline fills and epilogues do not coincide).
// Per slab and wave: 8 MFMAs of the current fragments with the two DMA instructions of slab i+6 and the 12 reads of slab
// i+1 spread between them, then the counted wait that retires slab i+2 and the barrier that publishes it.  Two slabs per
// trip (static register sets); the steady-state trips carry no conditionals.
template <bool NTA, bool NTB, int ABL, bool BIAS, bool CONV = false>
__device__ __forceinline__ void stream_piece(Stager& sg, f32x16 (&acc)[4][2], float (&cs)[4], int wc, int sb, int P, int rot,
                                             uint32_t a_off, uint32_t b_off) {
  constexpr bool no_mma = (ABL & 1) && ABL != 6, no_dma = (ABL & 2) && ABL != 6;
  int stage = rot;                                     // next slab to stage, relative to sb, walks rot .. P-1, 0 .. rot-1
  auto issue_next = [&](int slot) {
    sg.issue<NTA, NTB, ABL == 6, CONV>(sb + stage, slot);
    stage = stage + 1 == P ? 0 : stage + 1;
  };
  // ---- prologue: slabs 0 .. AHEAD-1 in flight, slabs 0 and 1 landed, fragments of slab 0 in registers
#pragma unroll
  for (int j = 0; j < AHEAD; ++j)
    if (j < P) issue_next(j);
  if (P >= AHEAD) wait_vm<2 * (AHEAD - 2)>(); else wait_vm<0>();
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("" ::: "memory");
  Frags f0, f1;
  read_frags_lo(f0, 0u, a_off, b_off);
  read_frags_hi(f0, 0u, a_off, b_off);
#define W256_MMA(A, CUR)                                                                                     \
  if constexpr (!no_mma) {                                                                                   \
    mma32(acc[A][0], CUR.a[A], CUR.b[0]);                                                                    \
    mma32(acc[A][1], CUR.a[A], CUR.b[1]);                                                                    \
  } else {                                                                                                   \
    asm volatile("" ::"v"(CUR.a[A].x), "v"(CUR.b[0].x), "v"(CUR.b[1].w), "v"(CUR.a[A].w));                   \
  }                                                                                                          \
  __builtin_amdgcn_sched_barrier(0);
#define W256_PHASE(Q, CUR, NXT, ST)                                                                          \
  {                                                                                                          \
    const int q_ = (Q);                                                                                      \
    const bool more_ = ((ST) || q_ + AHEAD < P) && !no_dma, next_ = (ST) || q_ + 1 < P;                      \
    const uint32_t nslab_ = (uint32_t)(((q_ + 1) & (RING - 1)) * SLAB_BYTES);                                \
    W256_MMA(0, CUR)                                                                                         \
    if (more_) issue_next((q_ + AHEAD) & (RING - 1));                                                        \
    __builtin_amdgcn_sched_barrier(0);                                                                       \
    W256_MMA(1, CUR)                                                                                         \
    if (next_) read_frags_lo(NXT, nslab_, a_off, b_off);                                                     \
    __builtin_amdgcn_sched_barrier(0);                                                                       \
    W256_MMA(2, CUR)                                                                                         \
    if (next_) read_frags_hi(NXT, nslab_, a_off, b_off);                                                     \
    __builtin_amdgcn_sched_barrier(0);                                                                       \
    W256_MMA(3, CUR)                                                                                         \
    if constexpr (BIAS) { if ((q_ & 3) == wc) frag_colsum(cs, CUR); }   /* the four waves of a row half take turns */  \
    if (more_) wait_vm<2 * (AHEAD - 2)>(); else wait_vm<0>();                                                \
    __builtin_amdgcn_s_barrier();                                                                            \
    asm volatile("" ::: "memory");                                                                           \
    __builtin_amdgcn_sched_barrier(0);                                                                       \
  }
  int q = 0;
  for (; q + AHEAD + 1 < P; q += 2) {     // steady state
    W256_PHASE(q, f0, f1, true)
    W256_PHASE(q + 1, f1, f0, true)
  }
  for (; q + 1 < P; q += 2) {             // the last AHEAD slabs: nothing left to stage
    W256_PHASE(q, f0, f1, false)
    W256_PHASE(q + 1, f1, f0, false)
  }
  if (q < P) W256_PHASE(q, f0, f1, false)
#undef W256_PHASE
#undef W256_MMA
}

// FIRST: dw is known to hold zeros (a freshly cleared gradient buffer that this launch is the first to write: W256Item.overwrite) --
// the tile is stored, the 16 loads of the read-modify-write are not issued (133 MB of the 1.9 GB the M = B x T' launch moves).
template <bool COH, bool FIRST = false>
__device__ __forceinline__ void flush_tile(f32x16 (&acc)[4][2], float* dw, int ldw, int N, int K, int n_base, int k_base,
                                           unsigned char* scr, int lane) {
  // acc[a][b]: rows n_base + 32a + (r&3) + 8(r>>2) + 4(lane>>5), column k_base + 32b + (lane&31).  Through 4 KB of the
  // (now idle) ring a 32 x 32 tile becomes 8 rows x 128 B per instruction: 16-byte read-modify-write of dw, in two rounds of
  // four tiles with all 16 loads of a round in flight at once (one round per tile cost 8 memory latencies per piece:
  // 0.13 ms of the launch).  dw came out of a table (no known address space): buffer accesses, which also carry the
  // cache policy.  COH: sc0 sc1 -- loads are served by memory and not by an L2 line another XCD has since rewritten,
  // stores write through and drop the line.
  constexpr int AUX = COH ? 17 : 0;
  const int rr = lane >> 3, cq = lane & 7;
  // the descriptor ends with the matrix: rows >= N fall outside (loads give 0, stores are dropped); columns >= K of a
  // ragged last tile are sent outside by their offset
  auto rs = __builtin_amdgcn_make_buffer_rsrc(dw, 0, ((N - 1) * ldw + K) * 4, 0x00020000);
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    otr_u32x4 old[2][2][4];
    uint32_t off[2][2][4];
#pragma unroll
    for (int a2 = 0; a2 < 2; ++a2)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int row = n_base + 32 * (2 * h + a2) + 8 * j + rr, col = k_base + 32 * b + 4 * cq;
          off[a2][b][j] = col < K ? (uint32_t)((row * ldw + col) * 4) : 0xfffffff0u;
          if constexpr (FIRST) old[a2][b][j] = otr_u32x4{0u, 0u, 0u, 0u};
          else old[a2][b][j] = __builtin_amdgcn_raw_buffer_load_b128(rs, off[a2][b][j], 0, AUX);
        }
#pragma unroll
    for (int a2 = 0; a2 < 2; ++a2)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        float* sw = reinterpret_cast<float*>(scr + (a2 * 2 + b) * 4096);
#pragma unroll
        for (int r = 0; r < 16; ++r) sw[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 32 + (lane & 31)] = acc[2 * h + a2][b][r];
      }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
#pragma unroll
    for (int a2 = 0; a2 < 2; ++a2)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const float* sw = reinterpret_cast<const float*>(scr + (a2 * 2 + b) * 4096);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float4 t = *reinterpret_cast<const float4*>(sw + (8 * j + rr) * 32 + 4 * cq);
          const otr_u32x4 o = old[a2][b][j];
          otr_u32x4 v = {__float_as_uint(__uint_as_float(o.x) + t.x), __float_as_uint(__uint_as_float(o.y) + t.y),
                         __float_as_uint(__uint_as_float(o.z) + t.z), __float_as_uint(__uint_as_float(o.w) + t.w)};
          __builtin_amdgcn_raw_buffer_store_b128(v, rs, off[a2][b][j], 0, AUX);
        }
      }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // the second round overwrites the scratch: reads above first
  }
}

// dbias[n_base + 32a + lane] += cs[a] (lanes 0..31), same cache policy and the same turnstile as the tile it belongs to
template <bool COH> __device__ __forceinline__ void flush_bias(const float (&cs)[4], float* dbias, int N, int n_base, int lane) {
  constexpr int AUX = COH ? 17 : 0;
  auto rs = __builtin_amdgcn_make_buffer_rsrc(dbias, 0, N * 4, 0x00020000);
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const uint32_t off = (uint32_t)((n_base + 32 * a + lane) * 4);
    const float old = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, off, 0, AUX));
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(old + cs[a]), rs, off, 0, AUX);
  }
}

__global__ void trread_probe_kernel(const uint16_t* image, const int* addr, uint16_t* out) {
  __shared__ __attribute__((aligned(16))) uint16_t img[2048];
  for (int i = threadIdx.x; i < 2048; i += 64) img[i] = image[i];
  __syncthreads();
  const uint2 r = tr_read((uint32_t)(uintptr_t)(lds_byte*)reinterpret_cast<unsigned char*>(img) + (uint32_t)addr[threadIdx.x]);
  out[threadIdx.x * 4 + 0] = (uint16_t)(r.x & 0xffffu); out[threadIdx.x * 4 + 1] = (uint16_t)(r.x >> 16);
  out[threadIdx.x * 4 + 2] = (uint16_t)(r.y & 0xffffu); out[threadIdx.x * 4 + 3] = (uint16_t)(r.y >> 16);
}

__global__ void wgrad256_init_kernel(int* flags, int n, uint32_t* zeros) {
  for (int i = threadIdx.x; i < n; i += blockDim.x) flags[i] = 0;
  if (threadIdx.x < 16) zeros[threadIdx.x] = 0u;
}

template <int ABL, bool CONV = false>
__global__ __launch_bounds__(512) void wgrad256_kernel(W256Args g) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[LDS_BYTES];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wid >> 2, wc = wid & 3;
  const uint32_t smem0 = (uint32_t)(uintptr_t)(lds_byte*)smem;

  // transposing read: lane p of 16-lane group gq supplies the 8-byte piece (row 4(gq>>1) + (p>>2), columns 16(gq&1) + 4(p&3)..)
  // of the [8][32] subtile and receives column 16(gq&1) + p of rows 4(gq>>1) .. +3
  const int gq = lane >> 4, p16 = lane & 15;
  const uint32_t lane_off = (uint32_t)((4 * (gq >> 1) + (p16 >> 2)) * 64 + (gq & 1) * 32 + (p16 & 3) * 8);
  const uint32_t a_off = smem0 + lane_off + (uint32_t)(4 * wr) * 512u;             // dy subtiles (mb, 4wr + i)
  const uint32_t b_off = smem0 + lane_off + 8192u + (uint32_t)(2 * wc) * 512u;     // x subtiles (mb, 2wc + i)

  // staging: wave w fills subtiles (mb = w>>2, columns 64(w&3) .. +63) of both parts; lane -> (subtile, row, 16-byte unit)
  const int st_row = 8 * (wid >> 2) + ((lane >> 2) & 7);
  const int st_col = 64 * (wid & 3) + 32 * (lane >> 5) + 8 * (lane & 3);

  // Slot <-> block: consecutive slots (tiles that share an operand panel and walk it in step) go to ONE XCD (block b runs on
  // XCD b % 8: observed placement, used for speed only), so a shared panel is fetched into that L2 once instead of once
  // per tile.  In the stream-K schedule the order inside an XCD's group is reversed: a piece only waits for a piece of the
  // NEXT chunk, i.e. (except at the 7 group seams) of a LOWER block index, which the dispatcher starts no later than
  // this one; every spin is bounded in any case.
  const int G = (int)gridDim.x, gx = G >> 3, bid = (int)blockIdx.x;
  const bool rounds = g.mode >= 1;
  const int slot = (G & 7) == 0 ? (bid & 7) * gx + (rounds ? (bid >> 3) : gx - 1 - (bid >> 3)) : (rounds ? bid : G - 1 - bid);
  // stream-K: this workgroup's chunk of the (tile, slab) space
  int pos = slot * g.chunk;
  const int c_end = pos + g.chunk < g.total ? pos + g.chunk : g.total;
  int round = 0;

  for (;;) {
    // ---- next piece: slabs [sb, sb + P) of tile `tile` of problem pi; `slice` of `nslices` pieces of that tile
    int pi = 0, tile, sb, P, rot = 0, slice = 0, nslices = 1;
    if constexpr (CONV) {
      // one round: slot = part * tiles + tile -- the tiles (taps) of one row range are neighbours on one XCD, where they share the
      // dy rows and most of the act1 rows (the nine taps of an output pixel overlap)
      if (round > 0 || slot >= g.rem_tiles * g.parts) break;
      ++round;
      const int R = g.chunk, part = slot / g.rem_tiles;
      tile = slot - part * g.rem_tiles;
      sb = (int)((int64_t)R * part / g.parts);
      P = (int)((int64_t)R * (part + 1) / g.parts) - sb;
      slice = part; nslices = g.parts;
    } else if (rounds) {
      // every tile is R slabs long.  Rounds 0 .. nfull-1: slot s walks the whole tile round*G + s; last round: the remaining
      // tiles cut into `parts` row ranges each, the parts of a tile in neighbouring slots
      const int R = g.chunk;
      int t;
      if (round < g.nfull) {
        t = round * G + slot; sb = 0; P = R;
      } else if (round == g.nfull && slot < g.rem_tiles * g.parts) {
        const int part = slot % g.parts;
        t = g.nfull * G + slot / g.parts;
        sb = (int)((int64_t)R * part / g.parts);
        P = (int)((int64_t)R * (part + 1) / g.parts) - sb;
        slice = part; nslices = g.parts;
      } else {
        break;
      }
      ++round;
      const int tpos = t * R;
      while (pi + 1 < g.nprob && g.p[pi + 1].start <= tpos) ++pi;
      tile = (tpos - g.p[pi].start) / R;
      if (nslices == 1) rot = (int)(((unsigned)pi * 40503u & 0xffu) * (unsigned)P >> 8);       // stagger the problems
    } else {
      if (pos >= c_end) break;
      while (pi + 1 < g.nprob && g.p[pi + 1].start <= pos) ++pi;
      const int R = (g.p[pi].M + SLAB_ROWS - 1) / SLAB_ROWS;
      const int rel = pos - g.p[pi].start;
      tile = rel / R;
      sb = rel - tile * R;
      const int tile_begin = g.p[pi].start + tile * R, tile_end = tile_begin + R;
      const int pe = tile_end < c_end ? tile_end : c_end;
      P = pe - pos;
      const int first_chunk = tile_begin / g.chunk, last_chunk = (tile_end - 1) / g.chunk;
      nslices = last_chunk - first_chunk + 1; slice = last_chunk - slot;      // the piece that finishes first goes first
      pos = pe;
    }
    const W256Prob& pr = g.p[pi];
    const int tiles_k = (pr.K + 255) >> 8;
    const int tn = tile / tiles_k, tk = tile - tn * tiles_k;
    const int n0 = tn * 256, k0 = tk * 256;

    Stager sg;
    sg.M = pr.M; sg.row0 = st_row; sg.zeros = reinterpret_cast<const unsigned char*>(g.zeros);
    sg.col_a = n0 + st_col < pr.N; sg.col_b = k0 + st_col < pr.K;
    sg.base_a = reinterpret_cast<const unsigned char*>(pr.dy + (int64_t)st_row * pr.ldy + n0 + st_col);
    sg.base_b = reinterpret_cast<const unsigned char*>(pr.x + (int64_t)st_row * pr.ldx + k0 + st_col);
    sg.step_a = (int64_t)pr.ldy * SLAB_ROWS * 2; sg.step_b = (int64_t)pr.ldx * SLAB_ROWS * 2;
    sg.dst = smem0 + (uint32_t)wid * 1024u;
    if constexpr (CONV) {
      const int tap = k0 / g.cC1, ch0 = k0 - tap * g.cC1;
      sg.kh = tap / 3; sg.kw = tap - 3 * sg.kh;
      sg.cT1 = g.cT1; sg.cF1 = g.cF1; sg.cT2 = g.cT2; sg.cF2 = g.cF2; sg.divF2 = g.cdivF2; sg.divT2 = g.cdivT2;
      sg.pix_bytes = (int64_t)g.cC1 * 2;
      sg.base_b = reinterpret_cast<const unsigned char*>(pr.x + ch0 + st_col);
    }

    f32x16 acc[4][2];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // bias gradient: column sums of the dy strip, taken by the tk == 0 tile of each strip.  The four waves of a row half hold
    // the same dy fragments: they take the slabs in turns (16 v_dot2c on every 4th slab; on every slab for all waves it cost
    // 0.05 ms of the launch) and meet in LDS before the epilogue.
    float cs[4] = {0.f, 0.f, 0.f, 0.f};
    const bool bias = pr.dbias != nullptr && tk == 0;
    // the dy strip of this tile is shared with the other tk tiles of its problem, the x strip with the other tn tiles
    const bool nt_a = (g.policy & 1) && tiles_k == 1, nt_b = (g.policy & 1) && pr.N <= 256;
#define W256_RUN(A, B)                                                                                       \
  {                                                                                                          \
    if (bias) stream_piece<A, B, ABL, true>(sg, acc, cs, wc, sb, P, rot, a_off, b_off);                     \
    else stream_piece<A, B, ABL, false>(sg, acc, cs, wc, sb, P, rot, a_off, b_off);                          \
  }
    if constexpr (CONV) stream_piece<false, false, ABL, false, true>(sg, acc, cs, wc, sb, P, rot, a_off, b_off);
    else if (nt_a && nt_b) W256_RUN(true, true)
    else if (nt_a) W256_RUN(true, false)
    else if (nt_b) W256_RUN(false, true)
    else W256_RUN(false, false)
#undef W256_RUN
    if (bias) {                                                      // workgroup-uniform
      float* bx = reinterpret_cast<float*>(smem);                    // [8 waves][4 fragments][32 columns] (the ring is idle)
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        cs[a] += __shfl_xor(cs[a], 32);                              // the two row halves of a column
        if (lane < 32) bx[(wid * 4 + a) * 32 + lane] = cs[a];
      }
      __syncthreads();
      if (wc == 0 && lane < 32) {
#pragma unroll
        for (int a = 0; a < 4; ++a)
          cs[a] = bx[((wr * 4 + 0) * 4 + a) * 32 + lane] + bx[((wr * 4 + 1) * 4 + a) * 32 + lane] +
                  bx[((wr * 4 + 2) * 4 + a) * 32 + lane] + bx[((wr * 4 + 3) * 4 + a) * 32 + lane];
      }
      __syncthreads();                                               // the epilogue reuses this LDS
    }
    // ---- accumulate into dw (every DMA has landed and every wave has passed the last barrier: the ring is idle)
    float* dw = pr.dw;
    const int n_base = n0 + 128 * wr, k_base = k0 + 64 * wc;
    unsigned char* scr = smem + wid * 16384;
    if constexpr (CONV) {
      // this row range's partial tile: stored (nobody else writes slab `slice`), summed by w256_reduce_kernel
      flush_tile<false, true>(acc, g.split_ws + (int64_t)slice * pr.N * pr.ldw, pr.ldw, pr.N, pr.K, n_base, k_base, scr, lane);
    } else if constexpr ((ABL & 4) != 0 && ABL != 6) {
    } else if (nslices == 1) {
      if (pr.flag0 & (1 << 30)) flush_tile<false, true>(acc, dw, pr.ldw, pr.N, pr.K, n_base, k_base, scr, lane);
      else flush_tile<false>(acc, dw, pr.ldw, pr.N, pr.K, n_base, k_base, scr, lane);
      if (bias && wc == 0 && lane < 32) flush_bias<false>(cs, pr.dbias, pr.N, n_base, lane);
    } else {
      int* flag = g.flags + (pr.flag0 & 0x3fffffff) + tile;
      if (slice > 0) {
        if (tid == 0) {
          int spins = 0;                                               // bounded: a lost piece gives a wrong sum, never a hang
          while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != slice && spins < g.spin_limit) {
            __builtin_amdgcn_s_sleep(8);
            ++spins;
          }
          if (spins >= g.spin_limit) {                                 // give-up code: word 15 of the zero line (otr_debug_wgrad256_errors)
            __hip_atomic_fetch_add(const_cast<int*>(reinterpret_cast<const int*>(g.zeros)) + 15, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // ... and the caller's sticky fault word: otr_optimizer_step skips the update of a step whose gradients may be wrong
            if (g.fault) __hip_atomic_fetch_add(g.fault, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
        __syncthreads();
      }
      if (slice == 0 && (pr.flag0 & (1 << 30))) flush_tile<true, true>(acc, dw, pr.ldw, pr.N, pr.K, n_base, k_base, scr, lane);   // the turnstile's first piece
      else flush_tile<true>(acc, dw, pr.ldw, pr.N, pr.K, n_base, k_base, scr, lane);
      if (bias && wc == 0 && lane < 32) flush_bias<true>(cs, pr.dbias, pr.N, n_base, lane);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // this wave's write-through stores are at memory
      __syncthreads();
      if (tid == 0) __hip_atomic_store(flag, slice + 1 == nslices ? 0 : slice + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();                                                 // the ring (epilogue scratch) is reused by the next piece
  }
}

}  // namespace

// Host side: lay the problems out in the (tile, slab) space, cut it into one chunk per workgroup, launch.
static inline int w256_tiles(const W256Item& it) { return ((it.N + 255) / 256) * ((it.K + 255) / 256); }

// The schedule of a launch (host only, no device work): the problem table with its slab offsets, the mode and its parameters,
// the grid.  Shared by wgrad256_launch and otr_debug_wgrad256_plan (tests/test_cabi.py replays the kernel's piece decoding on it).
static int32_t wgrad256_plan(const W256Item* it, int n, int grid_cap, W256Args& g, int& grid, int& flags) {
  if (n > W256_MAX_PROBS) {
    otr_set_error("wgrad256: %d problems exceed the table of %d", n, W256_MAX_PROBS);
    return -1;
  }
  int64_t total = 0;
  flags = 0;
  bool same_rows = true;
  for (int i = 0; i < n; ++i) {
    W256Prob& p = g.p[i];
    p.dy = reinterpret_cast<const uint16_t*>(it[i].dy); p.x = reinterpret_cast<const uint16_t*>(it[i].x); p.dw = it[i].dw;
    p.dbias = it[i].dbias;
    p.M = it[i].M; p.N = it[i].N; p.K = it[i].K; p.ldy = (int)it[i].ldy; p.ldx = (int)it[i].ldx; p.ldw = (int)it[i].ldw;
    p.start = (int)total;
    p.flag0 = flags | (it[i].overwrite ? (1 << 30) : 0);
    const int tiles = w256_tiles(it[i]), slabs = (it[i].M + SLAB_ROWS - 1) / SLAB_ROWS;
    same_rows = same_rows && it[i].M == it[0].M;
    total += (int64_t)tiles * slabs;
    flags += tiles;
  }
  if (total >= (1ll << 30)) {
    otr_set_error("wgrad256: %lld slabs do not fit the 32-bit work index", (long long)total);
    return -1;
  }
  int cap = grid_cap > 0 ? grid_cap : (grid_cap < 0 ? -grid_cap : 248);    // < 0: that many workgroups, stream-K schedule
  g.nprob = n; g.total = (int)total; g.spin_limit = g_otr_spin_limit; g.fault = g_otr_fault;
  if (same_rows && grid_cap >= 0 && flags >= 8) {
    // Rounds: every tile is R slabs long.  G slots (whole XCD groups) each walk one whole tile per round -- the tiles of a
    // problem sit in neighbouring slots of one XCD and read their shared operand panel in step -- and the tiles left over
    // after the full rounds are cut into `parts` row ranges so that they fill the slots once more.
    const int R = (it[0].M + SLAB_ROWS - 1) / SLAB_ROWS, T = flags;
    double best = 1e30;
    int bestG = 8;
    for (int G = cap / 8 * 8; G >= 8 && G >= cap / 2; G -= 8) {
      const int nfull = T / G, rem = T - nfull * G, parts = rem ? std::min(G / rem, 8) : 1;
      const double cost = nfull + (rem ? 1.0 / parts + 0.02 : 0.0);           // + a piece's fill / drain
      if (cost < best - 1e-9) { best = cost; bestG = G; }
    }
    grid = bestG;
    g.mode = 1; g.chunk = R;
    g.nfull = T / grid; g.rem_tiles = T - g.nfull * grid; g.parts = g.rem_tiles ? std::min(grid / g.rem_tiles, 8) : 1;
    if (g.parts > R) g.parts = R;                   // never an empty row range
  } else {
    // Stream-K: short chunks make pieces whose epilogue outweighs their work: at least 64 slabs (1024 rows) per workgroup
    while (cap > 1 && total / cap < 64) cap /= 2;
    const int64_t chunk = (total + cap - 1) / cap;
    grid = (int)((total + chunk - 1) / chunk);
    if (grid >= 8) grid = (grid + 7) / 8 * 8;       // whole XCD groups; the extra chunks are empty
    g.mode = 0; g.chunk = (int)chunk;
  }
  return 0;
}

int32_t wgrad256_launch(const W256Item* it, int n, void* workspace, int64_t workspace_bytes, int grid_cap, int ablate, hipStream_t s) {
  if (n <= 0) return 0;
  W256Args g{};
  int grid = 0, flags = 0;
  if (int32_t e = wgrad256_plan(it, n, grid_cap, g, grid, flags)) return e;
  const int64_t need = 64 + (int64_t)flags * 4;   // (== wgrad256_workspace_bytes)
  if (!workspace || workspace_bytes < need) {
    otr_set_error("wgrad256: workspace of %lld bytes, need %lld", (long long)workspace_bytes, (long long)need);
    return -1;
  }
  g.ablate = ablate & 7; g.policy = ablate >> 3 ? (ablate >> 3) - 1 : 1;   // (ablate >> 3) - 1: bit 0 non-temporal strips
  g.zeros = workspace;
  g.flags = reinterpret_cast<int*>(reinterpret_cast<unsigned char*>(workspace) + 64);
  hipLaunchKernelGGL(wgrad256_init_kernel, dim3(1), dim3(256), 0, s, g.flags, flags, reinterpret_cast<uint32_t*>(workspace));
  switch (ablate & 7) {
    case 0: hipLaunchKernelGGL(wgrad256_kernel<0>, dim3((unsigned)grid), dim3(512), 0, s, g); break;
    case 1: hipLaunchKernelGGL(wgrad256_kernel<1>, dim3((unsigned)grid), dim3(512), 0, s, g); break;
    case 2: hipLaunchKernelGGL(wgrad256_kernel<2>, dim3((unsigned)grid), dim3(512), 0, s, g); break;
    case 3: hipLaunchKernelGGL(wgrad256_kernel<3>, dim3((unsigned)grid), dim3(512), 0, s, g); break;
    case 4: hipLaunchKernelGGL(wgrad256_kernel<4>, dim3((unsigned)grid), dim3(512), 0, s, g); break;
    case 5: hipLaunchKernelGGL(wgrad256_kernel<5>, dim3((unsigned)grid), dim3(512), 0, s, g); break;
    case 6: hipLaunchKernelGGL(wgrad256_kernel<6>, dim3((unsigned)grid), dim3(512), 0, s, g); break;
    default: hipLaunchKernelGGL(wgrad256_kernel<7>, dim3((unsigned)grid), dim3(512), 0, s, g); break;
  }
  return otr_check_launch("wgrad256");
}

// dw[i] = sum over the row ranges, in order (deterministic), float4 per thread
__global__ __launch_bounds__(256) void w256_reduce_kernel(const float* __restrict__ ws, int nslab, int64_t elems, float* __restrict__ dw) {
  const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= elems) return;
  float4 a = *reinterpret_cast<const float4*>(ws + i);
  for (int s = 1; s < nslab; ++s) {
    const float4 v = *reinterpret_cast<const float4*>(ws + (int64_t)s * elems + i);
    a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
  }
  *reinterpret_cast<float4*>(dw + i) = a;
}

int32_t wgrad256_conv_launch(const void* g2, const void* act1, float* dw2r, int B, int T1, int F1, int T2, int F2, int C1, int C2,
                             void* workspace, int64_t workspace_bytes, hipStream_t s) {
  const int64_t M = (int64_t)B * T2 * F2, K = 9ll * C1, elems = (int64_t)C2 * K;
  if (C1 % 256 != 0 || C2 % 8 != 0 || M < 16 * 2 * RING || M >= (1ll << 31) - 16) return 1;
  if (((uintptr_t)g2 | (uintptr_t)act1 | (uintptr_t)dw2r) % 16 != 0 || !workspace) return 1;
  const int tiles = ((C2 + 255) / 256) * (int)(K / 256);
  if (tiles > 256) return 1;
  // the row ranges: as many as fill one workgroup per CU, and as the workspace holds partial matrices of
  const int64_t head = 256;                                   // the zero line (64 bytes) and alignment
  int parts = 256 / tiles;
  const int64_t fit = (workspace_bytes - head) / (elems * 4);
  if (fit < parts) parts = (int)fit;
  const int R = (int)((M + SLAB_ROWS - 1) / SLAB_ROWS);
  if (parts > R / (2 * RING)) parts = R / (2 * RING);           // a row range is at least two turns of the ring long
  if (parts < 1) return 1;
  W256Args g{};
  W256Prob& p = g.p[0];
  p.dy = reinterpret_cast<const uint16_t*>(g2); p.x = reinterpret_cast<const uint16_t*>(act1); p.dw = dw2r; p.dbias = nullptr;
  p.M = (int)M; p.N = C2; p.K = (int)K; p.ldy = C2; p.ldx = C1; p.ldw = (int)K; p.start = 0; p.flag0 = 0;
  g.nprob = 1; g.total = tiles * R; g.chunk = R; g.mode = 2; g.nfull = 0; g.rem_tiles = tiles; g.parts = parts;
  g.spin_limit = g_otr_spin_limit; g.fault = g_otr_fault; g.ablate = 0; g.policy = 1;
  g.cT1 = T1; g.cF1 = F1; g.cT2 = T2; g.cF2 = F2; g.cC1 = C1; g.cdivF2 = make_fastdiv((uint32_t)F2); g.cdivT2 = make_fastdiv((uint32_t)T2);
  g.zeros = workspace; g.flags = nullptr;
  g.split_ws = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(workspace) + head);
  hipLaunchKernelGGL(wgrad256_init_kernel, dim3(1), dim3(256), 0, s, nullptr, 0, reinterpret_cast<uint32_t*>(workspace));   // the zero line
  const int grid = (tiles * parts + 7) / 8 * 8;               // whole XCD groups (the slot <-> block map); the extra ones leave at once
  hipLaunchKernelGGL((wgrad256_kernel<0, true>), dim3((unsigned)grid), dim3(512), 0, s, g);
  if (int32_t e = otr_check_launch("wgrad256(conv)")) return e;
  hipLaunchKernelGGL(w256_reduce_kernel, dim3((unsigned)((elems / 4 + 255) / 256)), dim3(256), 0, s, g.split_ws, parts, elems, dw2r);
  return otr_check_launch("wgrad256(conv reduce)");
}

int64_t wgrad256_workspace_bytes(const W256Item* it, int n) {
  int64_t flags = 0;
  for (int i = 0; i < n; ++i) flags += w256_tiles(it[i]);
  return 64 + flags * 4;
}

extern "C" int32_t otr_debug_trread(const void* image, const int32_t* addr, void* out, void* stream) {
  OTR_REQUIRE(image && addr && out, "debug_trread: null pointer");
  hipLaunchKernelGGL(trread_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, reinterpret_cast<const uint16_t*>(image), addr,
                     reinterpret_cast<uint16_t*>(out));
  return otr_check_launch("debug_trread");
}

// out: {mode, grid, chunk, nfull, rem_tiles, parts, total slabs, problems} then the first slab of every problem
extern "C" int32_t otr_debug_wgrad256_plan(const otr_wgrad_item_t* items, int32_t n, int32_t grid_cap, int32_t* out) {
  OTR_REQUIRE(items && out && n > 0 && n <= W256_MAX_PROBS, "debug_wgrad256_plan: bad arguments");
  W256Item it[W256_MAX_PROBS];
  for (int i = 0; i < n; ++i) it[i] = W256Item{items[i].dy, items[i].x, items[i].dw, items[i].dbias, items[i].M, items[i].N, items[i].K,
                                                items[i].ldy, items[i].ldx, items[i].ldw};
  W256Args g{};
  int grid = 0, flags = 0;
  if (int32_t e = wgrad256_plan(it, n, grid_cap, g, grid, flags)) return e;
  out[0] = g.mode; out[1] = grid; out[2] = g.chunk; out[3] = g.nfull; out[4] = g.rem_tiles; out[5] = g.parts; out[6] = g.total; out[7] = n;
  for (int i = 0; i < n; ++i) out[8 + i] = g.p[i].start;
  return 0;
}

// pieces of the LAST launch on this workspace that gave up waiting at a turnstile (0 unless workgroups could not be
// co-resident); synchronises the device
extern "C" int32_t otr_debug_wgrad256_errors(const void* workspace) {
  OTR_REQUIRE(workspace, "debug_wgrad256_errors: null workspace");
  int v = -1;
  if (hipMemcpy(&v, reinterpret_cast<const int*>(workspace) + 15, 4, hipMemcpyDeviceToHost) != hipSuccess) {
    otr_set_error("debug_wgrad256_errors: copy failed");
    return -1;
  }
  return v;
}
