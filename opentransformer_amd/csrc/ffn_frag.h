// Device helpers shared by the fused FFN kernels (ffn_fused.hip: 32-row workgroups, weights L2 -> VGPR; ffn3.hip: 128-row
// workgroups, weights through an LDS-DMA ring): the fragment conventions of otr_pack_frags, the GLU on accumulator tiles, the
// accumulator-tile -> operand-fragment / row-major / column-sum conversions, and the direct-to-LDS fragment copy.
#pragma once
#include "common.h"

constexpr int FF_RB = 32;   // rows of one MFMA B-operand tile (= rows per workgroup of the v1 kernels)

__device__ __forceinline__ float fast_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }

// 32 rows x D 16-bit activations -> LDS as 16-byte chunks, chunk index XOR (row & 15): the B-operand read of lane
// (m = lane&31, hi) -- chunk (2*ks + hi) of row m -- is then bank-conflict free for ds_read_b128
template <int D, int NTHR = 256>
__device__ __forceinline__ void stage_rows(uint4* dst, const uint16_t* src, int row0, int M, int tid) {
  constexpr int CPR = D / 8;
#pragma unroll
  for (int i = tid; i < FF_RB * CPR; i += NTHR) {
    const int r = i / CPR, ch = i % CPR;
    const int gr = min(row0 + r, M - 1);
    dst[r * CPR + (ch ^ (r & 15))] = ld_global_b128(src + (int64_t)gr * D + ch * 8);
  }
}
template <int D> __device__ __forceinline__ uint4 frag_b(const uint4* rows, int m, int hi, int ks) {
  return rows[m * (D / 8) + ((2 * ks + hi) ^ (m & 15))];
}

// accumulator tile (16 floats: hidden units 8q + 4hi + (r&3), q = r>>2, of row m = lane&31) -> two B-operand fragments
__device__ __forceinline__ void tile_to_frags(const float* v, uint4& f0, uint4& f1) {
  f0 = make_uint4(pack2h(v[0], v[1]), pack2h(v[2], v[3]), pack2h(v[4], v[5]), pack2h(v[6], v[7]));
  f1 = make_uint4(pack2h(v[8], v[9]), pack2h(v[10], v[11]), pack2h(v[12], v[13]), pack2h(v[14], v[15]));
}

// store an accumulator tile as 32 consecutive 16-bit elements of row m (row-major consumer: the weight-gradient GEMM).
// The row's 64 bytes are split over lanes m and m+32 in 8-byte pieces; one v_permlane32_swap per dword turns them into
// 16-byte pieces (cdna_hip_programming.md T21): lane (m, hi) then owns elements [8(q0+hi), 8(q0+hi)+8) for q0 = 0, 2.
__device__ __forceinline__ void store_tile_row(uint16_t* rowp, const uint4& f0, const uint4& f1, int hi, bool live) {
  uint32_t w[8] = {f0.x, f0.y, f0.z, f0.w, f1.x, f1.y, f1.z, f1.w};
#pragma unroll
  for (int q0 = 0; q0 < 4; q0 += 2) {
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      auto r = __builtin_amdgcn_permlane32_swap(w[2 * q0 + e], w[2 * q0 + 2 + e], false, false);
      w[2 * q0 + e] = r[0];
      w[2 * q0 + 2 + e] = r[1];
    }
    if (live) st_global_b128(rowp + 8 * (q0 + hi), make_uint4(w[2 * q0], w[2 * q0 + 1], w[2 * q0 + 2], w[2 * q0 + 3]));
  }
}

// column sums of an accumulator tile over its 32 rows: the 16 registers of lane (m, hi) are hidden units 8q + 4hi + (r&3) of
// row m.  Reduce-scatter butterfly over the 32 lanes of a half-wave (xor 16, 8, 4, 2 halve the register set each step, xor 1
// finishes): 16 shuffles per tile instead of 80 for sixteen independent butterflies; lane m ends up with the total of
// register r = (m4 m3 m2 m1) and the even lanes store it.  dst = the 32 floats of this tile in the partial-sum row.
__device__ __forceinline__ void tile_colsum_store(const float* v, float* dst, int lane, int hi, bool rows_live) {
  const int m = lane & 31;
  const bool b4 = m & 16, b3 = m & 8, b2 = m & 4, b1 = m & 2;
  float a[8], b[4], c[2];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float lo = rows_live ? v[i] : 0.f, hi_ = rows_live ? v[8 + i] : 0.f;
    a[i] = (b4 ? hi_ : lo) + __shfl_xor(b4 ? lo : hi_, 16);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) b[i] = (b3 ? a[4 + i] : a[i]) + __shfl_xor(b3 ? a[i] : a[4 + i], 8);
#pragma unroll
  for (int i = 0; i < 2; ++i) c[i] = (b2 ? b[2 + i] : b[i]) + __shfl_xor(b2 ? b[i] : b[2 + i], 4);
  float d = (b1 ? c[1] : c[0]) + __shfl_xor(b1 ? c[0] : c[1], 2);
  d += __shfl_xor(d, 1);
  const int r = ((m >> 4) & 1) * 8 + ((m >> 3) & 1) * 4 + ((m >> 2) & 1) * 2 + ((m >> 1) & 1);
  if ((m & 1) == 0) dst[8 * (r >> 2) + 4 * hi + (r & 3)] = d;
}

// ------------------------------------------------------------------------------------------------ direct-to-LDS fragment copy
typedef __attribute__((address_space(3))) unsigned char lds_byte;
typedef __attribute__((address_space(1))) const unsigned char gbl_byte;

// one 1 KiB fragment: 64 lanes x 16 B, global (fragment-major pack) -> LDS, asynchronous (vmcnt)
// Inline asm, not __builtin_amdgcn_global_load_lds: with the builtin hipcc tracks the pending LDS write and puts
// `s_waitcnt vmcnt(0)` in front of the next ds_read of ANY address -- i.e. it waited for the chunk it had just started to
// fetch before multiplying the current one (seen in the ISA: the whole DMA latency exposed per chunk).  The asm form is
// invisible to that bookkeeping; the kernels below wait themselves (vmcnt(0) + barrier right before a buffer is read).
// M0 carries the wave-uniform LDS byte address and is restored afterwards (cdna_hip_programming.md 5.7).
__device__ __forceinline__ void dma_frag(const uint4* src_frag, unsigned char* lds_frag, int lane) {
  const uint32_t dst = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lds_byte*)lds_frag);
  const uint4* src = src_frag + lane;
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
}

// the same with a wave-uniform source (SGPR base + per-lane byte offset): scalar address arithmetic only
typedef __attribute__((address_space(3))) unsigned char ffn_lds_byte;
__device__ __forceinline__ void ffn_dma(const void* uniform_src, uint32_t lane_off, uint32_t lds_dst) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(lane_off), "s"(uniform_src), "s"(lds_dst) : "memory");
}
