// Shared device/host helpers for libotrans_hip.so (gfx950 / CDNA4 only).
//
// Conventions used by every kernel in this library:
//  * wave = 64 lanes; workgroups are 256 threads (4 waves) unless stated.
//  * "chunk" = 16 bytes of the compute type CT: 8 bf16 or 4 f32.  One MFMA k-step consumes one
//    chunk per lane for A and one for B:
//      bf16: v_mfma_f32_16x16x32_bf16, lane l holds row (l&15), k = (l>>4)*8 + 0..7
//      f32 : 4 x v_mfma_f32_16x16x4_f32; instruction i contracts k = {g*4+i : g=0..3}, so lane l
//            again holds row (l&15), k = (l>>4)*4 + 0..3   (any k-bijection is legal as long as
//            A and B use the same one)
//    so both compute types share one LDS geometry: a 16-byte chunk per (row, l>>4).
//  * C/D layout of every 16x16 MFMA (dtype independent): col = lane&15, row = (lane>>4)*4 + reg.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/otrans_hip.h"

// The 16-bit storage / MFMA-input type is a BUILD parameter: bf16 (libotrans_hip.so, dtype code OTR_BF16) or IEEE fp16
// (-DOTR_HALF_FP16: libotrans_hip_f16.so, dtype code OTR_F16).  fp16 has 3 more mantissa bits -- the full model's
// logits land 5e-4 from the fp32 reference instead of 3.9e-3 (tools/precision_study.py) at the same MFMA rate -- and
// needs loss scaling for the gradients (ops.ScaleGradFn + otr_optimizer_step).  Kernels call the 16-bit type "bf16_t"
// (raw bits) throughout; only the conversions and the MFMA opcode below differ.
#ifdef OTR_HALF_FP16
typedef _Float16 otr_hreal;
#define OTR_H16 OTR_F16
#else
typedef __bf16 otr_hreal;
#define OTR_H16 OTR_BF16
#endif
typedef __attribute__((ext_vector_type(8))) otr_hreal bf16x8;
typedef __attribute__((ext_vector_type(2))) otr_hreal bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef uint16_t bf16_t;  // raw bf16 bits in memory

// ---------------------------------------------------------------- error plumbing (host)
void otr_set_error(const char* fmt, ...);
int32_t otr_check_launch(const char* what);
// Zero n floats with a kernel.  The library never uses hipMemset*: under hipGraph capture on ROCm 7.2
// a captured memset node was observed to leave every 4th float of a pool buffer stale on replay.
void otr_zero_f32(float* p, int64_t n, hipStream_t s);
#define OTR_REQUIRE(cond, ...)          \
  do {                                  \
    if (!(cond)) {                      \
      otr_set_error(__VA_ARGS__);       \
      return -1;                        \
    }                                   \
  } while (0)

// ---------------------------------------------------------------- explicit global-memory accesses
// A pointer that reaches a kernel through a table in memory (grouped GEMM) has no known address space and its
// accesses compile to flat_load / flat_store, which count on lgkmcnt as well: every LDS wait then also waits for
// the global prefetch.  These helpers pin the access to addrspace(1).
typedef uint32_t otr_u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t otr_u32x2 __attribute__((ext_vector_type(2)));
#define OTR_GLOBAL __attribute__((address_space(1)))
__device__ __forceinline__ uint4 ld_global_b128(const void* p) {
  otr_u32x4 v = *(const OTR_GLOBAL otr_u32x4*)(p);
  return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ uint2 ld_global_b64(const void* p) {
  otr_u32x2 v = *(const OTR_GLOBAL otr_u32x2*)(p);
  return make_uint2(v.x, v.y);
}
__device__ __forceinline__ void st_global_b128(void* p, uint4 v) {
  otr_u32x4 t = {v.x, v.y, v.z, v.w};
  *(OTR_GLOBAL otr_u32x4*)(p) = t;
}

// streaming store: data nobody reads soon (tiles saved for the backward pass).  A launch that leaves tens of MB dirty in the L2s pays
// for their write-back when it ENDS (the L2s of the XCDs are not coherent: kernel end flushes them); marked non-temporal the lines
// leave while the launch still computes
__device__ __forceinline__ void st_global_b128_nt(void* p, uint4 v) {
  otr_u32x4 t = {v.x, v.y, v.z, v.w};
#if defined(OTR_ST_WT) && OTR_ST_WT
  // experiment: system-scope write-through (sc0 sc1) + nt: the line goes to memory now instead of lingering dirty in the L2 until the
  // end-of-kernel release writes it back
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt\n\ts_nop 1" ::"v"(p), "v"(t) : "memory");
#else
  __builtin_nontemporal_store(t, (OTR_GLOBAL otr_u32x4*)(p));
#endif
}

// ---------------------------------------------------------------- scalar conversions
#ifdef OTR_HALF_FP16
__device__ __forceinline__ float bf2f(bf16_t v) { return (float)__builtin_bit_cast(_Float16, v); }
__device__ __forceinline__ float h2f_lo(uint32_t w) { return bf2f((bf16_t)(w & 0xffffu)); }   // low / high half of a packed pair
__device__ __forceinline__ float h2f_hi(uint32_t w) { return bf2f((bf16_t)(w >> 16)); }
#else
__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ float h2f_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float h2f_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
#endif
__device__ __forceinline__ bf16_t f2bf(float f) {  // RNE (v_cvt_pk_bf16_f32 / v_cvt_f16_f32)
  otr_hreal h = (otr_hreal)f;
  return __builtin_bit_cast(bf16_t, h);
}
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
  f32x2 v = {lo, hi};
  bf16x2 h = __builtin_convertvector(v, bf16x2);
  return __builtin_bit_cast(uint32_t, h);
}

// half-type-neutral names (the 16-bit storage type is a build parameter, see the top of this file)
__device__ __forceinline__ float h2f(bf16_t v) { return bf2f(v); }
__device__ __forceinline__ bf16_t f2h(float f) { return f2bf(f); }
__device__ __forceinline__ uint32_t pack2h(float lo, float hi) { return pack2bf(lo, hi); }

// 32x32x16 MFMA on the 16-bit storage type.  A: lane l holds row (l&31), k = (l>>5)*8 + 0..7; B: col (l&31), same k;
// D: col = l&31, row = (reg&3) + 8*(reg>>2) + 4*(l>>5)  (MI355X_MICROARCH / cdna_hip_programming.md section 3)
typedef __attribute__((ext_vector_type(16))) float f32x16;
__device__ __forceinline__ void mma32(f32x16& acc, const uint4& a, const uint4& b) {
#ifdef OTR_HALF_FP16
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
#else
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
#endif
}

template <class T> struct ElemIO;
template <> struct ElemIO<float> {
  static __device__ __forceinline__ float ld(const float* p) { return *p; }
  static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct ElemIO<bf16_t> {
  static __device__ __forceinline__ float ld(const bf16_t* p) { return bf2f(*p); }
  static __device__ __forceinline__ void st(bf16_t* p, float v) { *p = f2bf(v); }
};

// ---------------------------------------------------------------- compute-type traits
// CT = float | bf16_t.  Chunk = uint4 (16 B) holding CE elements of CT.
template <class CT> struct MMA;
template <> struct MMA<bf16_t> {
  static constexpr int CE = 8;       // elements per 16-byte chunk
  static constexpr int KSTEP = 32;   // contraction length of one mma() call
  static constexpr int TPC = 2;      // 16-wide C tiles that make up one contraction chunk
  static __device__ __forceinline__ void mma(f32x4& acc, const uint4& a, const uint4& b) {
#ifdef OTR_HALF_FP16
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
#else
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
#endif
  }
  // pack CE floats into a chunk
  static __device__ __forceinline__ uint4 pack(const float* f) {
    uint4 r;
    r.x = pack2bf(f[0], f[1]);
    r.y = pack2bf(f[2], f[3]);
    r.z = pack2bf(f[4], f[5]);
    r.w = pack2bf(f[6], f[7]);
    return r;
  }
  // Build the contraction-operand chunk from TPC accumulator tiles (C layout, rows = contraction
  // index).  Element j<4 <- tile0[j] (contraction row g*4+j of the first 16), j>=4 <- tile1[j-4].
  static __device__ __forceinline__ uint4 from_tiles(const f32x4* t) {
    uint4 r;
    r.x = pack2bf(t[0][0], t[0][1]);
    r.y = pack2bf(t[0][2], t[0][3]);
    r.z = pack2bf(t[1][0], t[1][1]);
    r.w = pack2bf(t[1][2], t[1][3]);
    return r;
  }
};
template <> struct MMA<float> {
  static constexpr int CE = 4;
  static constexpr int KSTEP = 16;
  static constexpr int TPC = 1;
  static __device__ __forceinline__ void mma(f32x4& acc, const uint4& a, const uint4& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.x), __uint_as_float(b.x), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.y), __uint_as_float(b.y), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.z), __uint_as_float(b.z), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.w), __uint_as_float(b.w), acc, 0, 0, 0);
  }
  static __device__ __forceinline__ uint4 pack(const float* f) {
    return make_uint4(__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3]));
  }
  static __device__ __forceinline__ uint4 from_tiles(const f32x4* t) {
    return make_uint4(__float_as_uint(t[0][0]), __float_as_uint(t[0][1]), __float_as_uint(t[0][2]),
                      __float_as_uint(t[0][3]));
  }
};

// Load N consecutive elements of source type ST starting at p into floats.  `nvalid` elements are
// in range (rest become 0); `vec` says p is 16-byte aligned for the vector path.
template <class ST, int N>
__device__ __forceinline__ void load_row(const ST* p, int nvalid, bool vec, float* out) {
  if (vec && nvalid >= N) {
    if constexpr (sizeof(ST) == 4) {
#pragma unroll
      for (int i = 0; i < N; i += 4) {
        float4 v = *reinterpret_cast<const float4*>(p + i);
        out[i] = v.x; out[i + 1] = v.y; out[i + 2] = v.z; out[i + 3] = v.w;
      }
    } else {
      if constexpr (N == 8) {
        uint4 v = *reinterpret_cast<const uint4*>(p);
        out[0] = h2f_lo(v.x); out[1] = h2f_hi(v.x);
        out[2] = h2f_lo(v.y); out[3] = h2f_hi(v.y);
        out[4] = h2f_lo(v.z); out[5] = h2f_hi(v.z);
        out[6] = h2f_lo(v.w); out[7] = h2f_hi(v.w);
      } else {
        static_assert(N == 4, "bf16 rows are read 4 or 8 at a time");
        uint2 v = *reinterpret_cast<const uint2*>(p);
        out[0] = h2f_lo(v.x); out[1] = h2f_hi(v.x);
        out[2] = h2f_lo(v.y); out[3] = h2f_hi(v.y);
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < N; ++i) out[i] = (i < nvalid) ? ElemIO<ST>::ld(p + i) : 0.f;
  }
}

// ---------------------------------------------------------------- LDS swizzles
// Row-major tile of 16-byte chunks, NCH chunks per row.  A 16-lane MFMA operand read touches 16
// consecutive rows at one logical chunk; swz() spreads those 16 accesses over all 64 banks.
// NCH=4 (64-B rows): ds_read_b128 is serviced in the lane groups {0-3,12-15,20-27},{4-11,16-19,28-31}
// (+32), i.e. rows {0-3,12-15} at chunk g together with rows {4-11} at chunk g^1; the lookup
// {0,2,3,1}[(row>>2)&3] makes all 16 (row&3, chunk) slots of such a group distinct.
// NCH=12 (192-B rows: head dim 96 in 16-bit, the Conformer): a row starts 12 units of 16 B further, i.e. -4 units mod the
// 16-unit bank window -- the NCH=4 geometry with the row classes relabelled (row&3 -> -row&3) and the k-step adding whole
// windows; the same lookup on the low two chunk bits is conflict-free (chunks stay inside their aligned group of four).
template <int NCH> __device__ __forceinline__ int swz(int row) {
  if constexpr (NCH == 4 || NCH == 12) return (0x78 >> (((row >> 2) & 3) * 2)) & 3;
  else if constexpr (NCH == 8) return (row >> 1) & 7;
  else if constexpr (NCH == 16) return row & 15;
  else return 0;
}

// ---------------------------------------------------------------- wave reductions (64 lanes)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

// sinusoidal position table entry (module/pos.py:30-42): PE[t,2i] = sin(t*exp(-2i ln(1e4)/d)), PE[t,2i+1] = cos(same)
__device__ __forceinline__ float pe_value(int t, int col, float neg_ln_over_d) {
  float div = expf((float)(col & ~1) * neg_ln_over_d);
  float ang = (float)t * div;
  return (col & 1) ? cosf(ang) : sinf(ang);
}

// counter-based RNG: one 32-bit draw per (seed, element index), SplitMix64 finaliser -- the optimizer's gradient noise
__device__ __forceinline__ uint32_t otr_rand32_sm64(uint64_t seed, uint64_t idx) {
  uint64_t z = seed + 0x9E3779B97F4A7C15ull * (idx + 1);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  return (uint32_t)(z >> 32);
}
// The DROPOUT masks' draw (every kernel that applies or regenerates a mask calls this with the element's index: forward and backward
// agree by construction).  Round 5: a 32-bit keyed mixer instead of SplitMix64.  Three 64-bit multiplies per ELEMENT were ~60 VALU
// issue slots; with residual_dropout 0.1 the 60 LayerNorm-bearing launches of a training step draw ~100 M masks, and the step ran
// 0.13 ms (3 %) slower than with dropout off (same box, tools/gpu_dropout_ab.sh).  Here: the index is xor-ed with a key, two rounds
// of the `lowbias32` finaliser (xorshift-multiply, constants 0x7FEB352D / 0x846CA68B) with a second key between them -- for a fixed
// seed a bijection of the low 32 index bits (exactly uniform over a full period), both keys wave-uniform functions of the seed
// (scalar ALU).  Keep rate, lag-1 / lag-256 / cross-step / cross-site correlations and row / column means of 2 M-element masks match
// an i.i.d. source to sampling noise (tools/dropout_hash_check.py); tests/test_gpu_dropout.py measures the kernels themselves.
__device__ __forceinline__ uint32_t otr_rand32(uint64_t seed, uint64_t idx) {
  const uint32_t s0 = (uint32_t)seed, s1 = (uint32_t)(seed >> 32);
  const uint32_t k0 = (s0 * 0x9E3779B1u) ^ s1;
  const uint32_t k1 = (s1 * 0x85EBCA77u) ^ (s0 >> 15) ^ 0xC2B2AE3Du;
  uint32_t x = (uint32_t)idx ^ k0;
  x += (uint32_t)(idx >> 32) * 0x27D4EB2Fu;        // indices past 2^32 (a step's offsets never get there; kept for totality)
  x ^= x >> 16; x *= 0x7FEB352Du;
  x ^= x >> 15; x ^= k1; x *= 0x846CA68Bu;
  x ^= x >> 16;
  return x;
}

// fast unsigned division by a runtime constant (host-computed magic), valid for n < 2^31
struct FastDiv {
  uint32_t d, magic, shift;
};
static inline FastDiv make_fastdiv(uint32_t d) {
  FastDiv f;
  f.d = d;
  if (d == 1) {
    f.magic = 0;
    f.shift = 0;
    return f;
  }
  uint32_t s = 0;
  while ((1u << s) < d) ++s;
  uint64_t m = ((1ull << (32 + s)) + d - 1) / d;  // ceil(2^(32+s)/d); fits in 33 bits
  f.magic = (uint32_t)(m - (1ull << 32));         // store low 32 bits (the 2^32 term is added back)
  f.shift = s;
  return f;
}
__device__ __forceinline__ uint32_t fdiv(uint32_t n, const FastDiv& f) {
  if (f.d == 1) return n;
  uint32_t t = __umulhi(n, f.magic);
  // (n*(2^32+magic)) >> (32+shift) == (t + n) >> shift, computed without overflow
  return (t + ((n - t) >> 1)) >> (f.shift - 1);
}
