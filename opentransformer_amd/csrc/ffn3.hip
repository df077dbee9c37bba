// Fused FFN sub-layer, third form: 128-row workgroups, weights through an LDS-DMA ring, 64 rows per wave
// (encoder/transformer.py:58-63, decoder/transformer.py:82-86, module/ffn.py:38-41 with activation 'glu').
//
//   forward :  partial y_s = w_2[:, slice s] glu(w_1[slice s] x + b_1[slice s]) for S = 4 hidden slices; the four workgroups of a
//              row block exchange their partial sums and each finishes 32 rows: + b_2, dropout, residual, LayerNorm
//   backward:  dh for the weight gradients from the SAVED (value, sigmoid) tiles, partial dx_s = dh[slice s] . w_1[slice s],
//              exchanged the same way (+ skip)
//
// Why a third form: the 32-row kernels (ffn_fused.hip) stream ALL packed weights into every CU and sit on the
// CU's ~22 B/clk vector-memory ingest (60 us forward for 3 MB per CU).  A workgroup here owns 128 rows x 1/S of the hidden
// units, so a CU ingests 1/S of the weights (0.75 MB at S = 4), once, by direct-to-LDS DMA, and the four waves share every
// fragment.  What the second form (32 rows per wave) ran into was the LDS READ rate: every 32-cycle MFMA consumed a fresh
// 1 KiB fragment.  Here the waves form a 2 x 2 grid -- wave (wr, wc) owns rows 64 wr .. +63 (two 32-row B-operand tiles, kept in
// REGISTERS for the whole kernel: 128 VGPRs) -- so every weight fragment read from LDS feeds TWO MFMAs:
//   GEMM1  h^T[32 hidden of sub-chunk wc, 64 rows] = w_1 frags (LDS) x x^T frags (registers)          64 MFMAs / 32 KiB read
//   GLU    on the accumulators; u (16-bit) stays in registers as B operands (the accumulator layout IS an operand layout once
//          the contraction index is permuted, and the permutation is applied to w_2 when it is packed, perm = 1) and is also
//          handed to the partner wave (same rows, other sub-chunk) through 4 KiB of LDS
//   GEMM2  y^T[128 output columns of half wc, 64 rows] += w_2 frags (LDS) x u^T frags (own: registers, partner's: LDS)
//                                                                                                    32 MFMAs / 20 KiB read
// i.e. 0.54 KiB of LDS reads per MFMA instead of 1, on the 512-register budget of one wave per SIMD (y accumulators 128 +
// x 128 + h 64 + u 48 + fragment staging 32).
//
// The hidden slice is walked in chunks of 64 units = 3 phases of 32 fragments (32 KiB) each -- A: w_1, contraction steps 0-7;
// B: steps 8-15; G: w_2 of the PREVIOUS chunk (its GEMM2 runs beside the GLU of this chunk, so the matrix pipe has work
// during the VALU phase) -- through a ring of four 32 KiB slots: phase p's barrier releases its slot for phase p+4, whose 8
// DMAs per wave are issued behind the first MFMA group of phase p+1 (scalar instructions only: wave-uniform source,
// SGPR base + lane offset) and waited for with a counted vmcnt(16) at the end of phase p+3's predecessor, so a phase's data
// has nearly three whole phases (> 3000 cycles) to land.  One barrier per phase (32 MFMAs per wave).
#include "ffn_frag.h"

namespace {

constexpr int F3_PHASE = 32 * 1024;      // one phase = 32 fragments
constexpr int F3_SLOTS = 4;
constexpr int F3_RING = F3_SLOTS * F3_PHASE;
constexpr int F3_UBUF = 16 * 1024;       // u hand-over: [wr][wc][row tile][k-step] fragments of 1 KiB
constexpr int F3_BIAS = 8 * 1024;        // b_1 of the slice: [v1 chunk][value 32 | gate 32] floats (<= 32 chunks)

// stores of what only the backward pass / the weight-gradient launch reads (OTR_F3_NT=0 at build time: plain stores, for A/B runs)
#ifndef OTR_F3_NT
#define OTR_F3_NT 1
#endif
#if OTR_F3_NT
#define F3_ST_SAVE(P, V) st_global_b128_nt((P), (V))
#else
#define F3_ST_SAVE(P, V) st_global_b128((P), (V))
#endif
template <int N> __device__ __forceinline__ void f3_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void f3_wait_lds() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void f3_barrier() {
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("" ::: "memory");
}

// GEMM1's MFMA with the register classes spelled out: accumulator in VGPRs (the GLU's VALU code reads it without
// v_accvgpr_read), weight fragment in VGPRs (fresh from ds_read), activation fragment in the ACCUMULATOR half of the register
// file ("a": it is an MFMA operand only).  With the builtin hipcc keeps both operands in VGPRs; the 128 registers of x then
// overflow the 256 architectural VGPRs and are shuttled through AGPRs (160 v_accvgpr moves per 32 MFMAs in the first build).
// Hazards (hipcc pads nothing inside asm, cdna_hip_programming.md 5.7): the accumulate chain D -> C of the next MFMA needs no
// wait states; the VALU readers of the result sit behind a barrier and an explicit s_nop (F3_MFMA_DRAIN); and every statement
// opens with `s_nop 1`: hipcc is free to place a VALU write of an operand (a register copy of the bias-initialised
// accumulator, seen in one build: the first chunk's row tile 0 came out wrong) directly in front of the statement, and a VALU
// write -> MFMA read needs two wait states.  Inside a back-to-back MFMA stream the two states hide behind the busy pipe.
#ifdef OTR_HALF_FP16
#define F3_MFMA_OP "v_mfma_f32_32x32x16_f16"
#else
#define F3_MFMA_OP "v_mfma_f32_32x32x16_bf16"
#endif
__device__ __forceinline__ void f3_mma_xa(f32x16& acc, const otr_u32x4& w, const otr_u32x4& x_acc) {
  asm volatile("s_nop 1\n\t" F3_MFMA_OP " %0, %1, %2, %0" : "+v"(acc) : "v"(w), "a"(x_acc));
}
#define F3_MFMA_DRAIN() asm volatile("s_nop 15\n\ts_nop 3" ::: "memory")   /* >= 12 wait states: MFMA result -> VALU reader */

// ---- partial-sum exchange between the four workgroups of a row block (fused forms).  Accumulator layout throughout: tile
// (ct, q): lane (m, hi) holds columns 32 ct + 8 q + 4 hi .. + 3 of row m as one float4, so every piece is 1 KiB lane-linear.
constexpr int F3_AUX_COH = 17;           // sc0 | sc1: stores write through to memory, loads are served by memory
typedef decltype(__builtin_amdgcn_make_buffer_rsrc((void*)nullptr, (short)0, 0, 0)) f3_rsrc_t;

// Cache policy of a transfer.  The XCDs' L2s are not coherent with each other, so in general the sender writes THROUGH
// (sc0 sc1) and the receiver reads past its L2 (sc0 sc1) -- with the default workgroup mapping (f3_block_map, map 1: the four
// slices of a row block on four XCDs) that is what every transfer does, and it was measured no slower.  Under map 0 the four
// workgroups of a row block share an XCD in practice; each publishes its XCC id in the row block's sync record at kernel
// entry, and a transfer whose two ends read EQUAL ids uses the shared L2: plain stores (at the L2 when vmcnt drains) and sc1
// loads (past the L1, served by the L2).  Mixed decisions stay correct: an id not (yet) valid or different -> write-through; a
// write-through store on the same XCD drops the line from that L2, so an L2-served load refetches it.
//   sync record of a row block (8 ints): [0] arrivals, MONOTONIC: launches are stream-ordered, so launch n finds 4n and a
//   workgroup waits for 4n + 4 -- nothing is reset, nobody is the "last reader"; [2 + s] = (n << 6) | (XCC id of slice s + 1),
//   valid only with this launch's n (a stale id of an earlier launch never vouches for a placement).
// Every agent-scope load is a round trip of 1-2 us, so the record is read TWICE per workgroup, both off the critical path: the
// generation at kernel entry, the four ids while the closing MFMAs run (ids still invalid then are re-read after the arrival
// wait, when they are certain to be there).
__device__ __forceinline__ int f3_xcc_id() {
  int v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 0xf;
}
struct F3Sync {
  uint32_t* rec;
  uint32_t gen;            // arrivals at this launch's start (a multiple of 4).  UNSIGNED: the counter gains 4 per launch and wraps
                           // after 2^30 launches (~30 h of training); all comparisons below are modular, so the wrap is harmless
  int xcc;
  uint32_t ids[4];         // the slices' entries as read early (f3_sync_read_ids)
};
__device__ __forceinline__ uint32_t f3_agent_load(const uint32_t* p) {
  return (uint32_t)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
// the tag a workgroup publishes next to its XCC id: this launch's number modulo 2^26 (equality tests only)
__device__ __forceinline__ uint32_t f3_tag(uint32_t gen, int xcc) { return (((gen >> 2) & 0x3ffffffu) << 6) | (uint32_t)(xcc + 1); }
// kernel entry: the load only (its round trip hides behind the prologue); f3_sync_publish after the prologue's first barrier
__device__ __forceinline__ void f3_sync_begin(F3Sync& y, int* rec, int xcc) {
  y.rec = reinterpret_cast<uint32_t*>(rec); y.xcc = xcc;
  y.gen = __hip_atomic_load(y.rec, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void f3_sync_publish(F3Sync& y, int sl, int tid) {
  y.gen = (uint32_t)__builtin_amdgcn_readfirstlane((int)y.gen) & ~3u;   // a late starter may see partners' arrivals of THIS launch: < 4
  if (tid == 0) __hip_atomic_store(y.rec + 2 + sl, f3_tag(y.gen, y.xcc), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void f3_sync_read_ids(F3Sync& y) {
#pragma unroll
  for (int i = 0; i < 4; ++i) y.ids[i] = __hip_atomic_load(y.rec + 2 + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool f3_same_xcd(const F3Sync& y, uint32_t entry) { return entry == f3_tag(y.gen, y.xcc); }
__device__ __forceinline__ bool f3_entry_valid(const F3Sync& y, uint32_t entry) { return (entry >> 6) == ((y.gen >> 2) & 0x3ffffffu) && (entry & 63u) != 0; }
template <int AUX>
__device__ __forceinline__ void f3_store_tile4(const f32x16 (&acc)[4], f3_rsrc_t rs, uint32_t base) {
#pragma unroll
  for (int ct = 0; ct < 4; ++ct)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const otr_u32x4 v = {__float_as_uint(acc[ct][4 * q]), __float_as_uint(acc[ct][4 * q + 1]),
                           __float_as_uint(acc[ct][4 * q + 2]), __float_as_uint(acc[ct][4 * q + 3])};
      __builtin_amdgcn_raw_buffer_store_b128(v, rs, base + (uint32_t)((ct * 4 + q) * 1024), 0, AUX);
    }
}
// wave (wr, wc) holds acc[rt][ct]: rows 64 wr + 32 rt .., columns 128 wc + 32 ct ..; the row tile that belongs to quarter `sl`
// (this workgroup finishes it) goes to LDS `own` [8 column tiles][4 q][64 lanes] float4, the others to
// scratch[sender sl][quarter] of the row block
__device__ __forceinline__ void f3_send_partials(const f32x16 (&acc)[2][4], f3_rsrc_t rs, float* own, const F3Sync& y, int sl, int wr, int wc,
                                                 int lane) {
#pragma unroll
  for (int rt = 0; rt < 2; ++rt) {
    const int qt = 2 * wr + rt;                                  // the quarter these 32 rows belong to (wave-uniform)
    if (qt == sl) {
#pragma unroll
      for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          *reinterpret_cast<float4*>(own + (((4 * wc + ct) * 4 + q) * 64 + lane) * 4) =
              make_float4(acc[rt][ct][4 * q], acc[rt][ct][4 * q + 1], acc[rt][ct][4 * q + 2], acc[rt][ct][4 * q + 3]);
    } else {
      const uint32_t base = (uint32_t)((sl * 4 + qt) * 32768 + (4 * wc) * 4096 + lane * 16);
      const uint32_t e = (uint32_t)__builtin_amdgcn_readfirstlane((int)(qt == 0 ? y.ids[0] : qt == 1 ? y.ids[1] : qt == 2 ? y.ids[2] : y.ids[3]));
      if (f3_same_xcd(y, e)) f3_store_tile4<0>(acc[rt], rs, base); else f3_store_tile4<F3_AUX_COH>(acc[rt], rs, base);
    }
  }
}
// one lane: arrive, then wait for all four workgroups of the row block (bounded; a give-up bumps the fault word)
__device__ __forceinline__ void f3_arrive_wait(const F3Sync& y, int spin_limit, int* fault) {
  __hip_atomic_fetch_add(y.rec, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const uint32_t target = y.gen + 4u;                            // modular: (int32_t)(counter - target) < 0 <=> not all four have arrived
  int spins = 0;
  while ((int32_t)(__hip_atomic_load(y.rec, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0 && spins < spin_limit) {
    __builtin_amdgcn_s_sleep(2);
    ++spins;
  }
  if (spins >= spin_limit && fault) __hip_atomic_fetch_add(fault, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <int AUX>
__device__ __forceinline__ void f3_load_tile2(otr_u32x4 (&part)[2][4], f3_rsrc_t rs, uint32_t base) {
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q) part[t][q] = __builtin_amdgcn_raw_buffer_load_b128(rs, base + (uint32_t)((t * 4 + q) * 1024), 0, AUX);
}
// the three partners' partials of quarter `sl`, column tiles 2 wid, 2 wid + 1 (after the arrival wait: every id is published)
__device__ __forceinline__ void f3_recv_partials(otr_u32x4 (&part)[3][2][4], f3_rsrc_t rs, const F3Sync& y, int sl, int wid, int lane) {
#pragma unroll
  for (int n = 0; n < 3; ++n) {
    const int s2 = n + (n >= sl ? 1 : 0);                        // the three other slices (wave-uniform)
    const uint32_t base = (uint32_t)((s2 * 4 + sl) * 32768 + (2 * wid) * 4096 + lane * 16);
    uint32_t e = (uint32_t)__builtin_amdgcn_readfirstlane((int)(s2 == 0 ? y.ids[0] : s2 == 1 ? y.ids[1] : s2 == 2 ? y.ids[2] : y.ids[3]));
    if (!f3_entry_valid(y, e)) e = f3_agent_load(y.rec + 2 + s2);   // read too early (rare): the sender may have seen OUR id
    if (f3_same_xcd(y, e)) f3_load_tile2<16>(part[n], rs, base); else f3_load_tile2<F3_AUX_COH>(part[n], rs, base);
  }
}

// The workgroup's 128 activation rows (16-bit, D = 256) -> LDS, then this wave's 64 rows as MFMA B-operand fragments.  Loading
// the fragments straight from global memory (lane (m, hi) fetches 16 bytes of ITS row per contraction step) touches 64
// different 128-byte lines per instruction, and the 32 KiB of a row tile thrash the L1: the forward prologue took 11.9 k
// cycles (5.7 us of a 45 us kernel, per-workgroup clock stamps).  Coalesced loads (a wave instruction covers two whole rows)
// into an XOR-swizzled row-major image (chunk index ^ (row & 15): conflict-free for the fragment reads, ffn_frag.h), two
// barriers, 32 ds_read_b128 per lane.
template <int D>
__device__ __forceinline__ void f3_stage_rows128(uint4* xs, const uint16_t* src, int rb, int M, int tid) {
  constexpr int CPR = D / 8;                                     // 16-byte chunks per row
  uint4 v[128 * CPR / 256];
#pragma unroll
  for (int k = 0; k < 128 * CPR / 256; ++k) {
    const int pi = tid + 256 * k, r = pi / CPR, ch = pi % CPR;
    const int64_t gr = min((int64_t)rb * 128 + r, (int64_t)M - 1);
    v[k] = ld_global_b128(src + gr * D + ch * 8);
  }
#pragma unroll
  for (int k = 0; k < 128 * CPR / 256; ++k) {
    const int pi = tid + 256 * k, r = pi / CPR, ch = pi % CPR;
    xs[r * CPR + (ch ^ (r & 15))] = v[k];
  }
}

struct Ffn3FwdArgs {
  const uint16_t* x16;     // [M, D]
  const uint4* p1; const float* b1; const uint4* p2;
  float* scratch;          // exchange of the partial sums: [row block][sender slice][quarter][32 KiB of accumulator tiles]
  int M, F, S;
  // ---- epilogue (in-kernel reduction of the S = 4 partial sums + bias + dropout + residual + LayerNorm)
  const float* x; const float* b2; const float* gamma; const float* beta; const uint64_t* seed;
  float* y; uint16_t* y16; float* z; float* mean; float* rstd;
  uint4* hsave;            // SAVE: (value + bias, sigmoid(gate)) of every hidden unit, 16-bit, in ACCUMULATOR-TILE order for the
                           // backward kernel: [row block][slice][64-unit chunk][wave][8 pieces][64 lanes] x 16 B; piece
                           // rt*2 + j = value registers 8j .. 8j+7 of row tile rt, piece 4 + rt*2 + j = the sigmoids
  uint16_t* usave;         // SAVE: u = glu output [128 * row blocks, F] row-major (operand of the w_2 weight gradient)
  uint16_t* slab;          // SLAB: [4 slices][M][256] 16-bit out -- this slice's partial sums w_2[:, slice] glu(...) (no bias): the launch that
                           // consumes them finishes the LayerNorm in its prologue (ln_pro.h); nothing of the exchange / epilogue fields
                           // below is touched then
  int* sync;               // [8 * row blocks] zero before the first launch: the row blocks' sync records (F3Sync)
  int* fault;              // NULL or the sticky fault word (otr_set_fault_counter)
  int spin_limit, coh_only, map;
  unsigned long long* trace;   // tuning hook (otr_debug_trace): wave 0 of every workgroup stamps the shader clock: [48 per workgroup]
  float eps, p_drop;
  uint64_t rng_offset;
};

// workgroup -> (row block, hidden slice).  Consecutive workgroup ids go to consecutive XCDs (observed, not promised: used for
// locality only).  map 0: the four slices of a row block share an XCD (their x rows are fetched into that L2 once, the partial
// sums meet in it) and every XCD streams all four weight slices.  map 1: an XCD owns ONE slice (XCDs s and s + 4 share slice s,
// the row blocks alternate between them): its 32 workgroups re-read 0.75 MB of weights instead of 3 MB, which survive the store
// stream of the saved tiles in the 4 MB L2 -- the four slices of a row block then sit on four XCDs (consecutive workgroup ids), their
// exchange takes the write-through path (measured: no slower) and x is fetched by four L2s.
__host__ __device__ __forceinline__ void f3_block_map(int b, int map, int& rb, int& s) {
  const int xcd = b & 7, j = b >> 3;
  if (map) {
    s = xcd & 3;
    rb = 2 * j + (xcd >> 2);
  } else {
    s = j & 3;
    rb = (j >> 2) * 8 + xcd;
  }
}

// accumulator tiles acc[rt][ct] of wave (wr, wc) (rows 64 wr + 32 rt .., columns 128 wc + 32 ct ..) -> the slice's 16-bit slab rows:
// through LDS (the ring, free after the main loop) as a row-major [128][256] tile, then whole 512-byte rows, two per wave
// instruction.  (Straight from the accumulator layout an instruction would touch 32 rows with 8 bytes each: the CU's address path
// takes about a cycle per line touched.)
__device__ __forceinline__ void f3_store_slab(const f32x16 (&acc)[2][4], unsigned char* lds, uint16_t* slab_rows, int rb, int M, int wid, int wr, int wc,
                                              int lane) {
  constexpr int TS = 256 * 2 + 16;                                 // bytes per tile row
  const int m = lane & 31, hi = lane >> 5;
#pragma unroll
  for (int rt = 0; rt < 2; ++rt)
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *reinterpret_cast<uint2*>(lds + (64 * wr + 32 * rt + m) * TS + (128 * wc + 32 * ct + 8 * q + 4 * hi) * 2) =
            make_uint2(pack2h(acc[rt][ct][4 * q], acc[rt][ct][4 * q + 1]), pack2h(acc[rt][ct][4 * q + 2], acc[rt][ct][4 * q + 3]));
  f3_wait_lds();
  f3_barrier();
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int r = 32 * wid + 2 * k + hi;
    const int64_t row = (int64_t)rb * 128 + r;
    const uint4 v = *reinterpret_cast<const uint4*>(lds + r * TS + m * 16);
    if (row < M) st_global_b128(slab_rows + row * 256 + m * 8, v);
  }
}

template <int D, int ABL, bool SAVE, bool SLAB = false>
__global__ __launch_bounds__(256, 1) void ffn3_fwd_kernel(Ffn3FwdArgs p) {
  static_assert(D == 256, "two 128-column halves, 16 contraction steps");
  constexpr int NKS = D / 16;
  __shared__ __attribute__((aligned(1024))) unsigned char smem[F3_RING + F3_UBUF + F3_BIAS];
  unsigned char* ring = smem;
  unsigned char* ubuf = smem + F3_RING;
  float* bias_s = reinterpret_cast<float*>(smem + F3_RING + F3_UBUF);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wid >> 1, wc = wid & 1;
  const int m = lane & 31, hi = lane >> 5;
  int rb, sl;
  f3_block_map((int)blockIdx.x, p.map, rb, sl);
  if (rb * 128 >= p.M) return;                                   // whole workgroup: the grid is padded to whole XCD groups
  const int row0 = rb * 128 + wr * 64;
  int stamp_i = 0;
#define F3_STAMP() if constexpr ((ABL & 16) != 0) { if (p.trace && tid == 0 && stamp_i < 48) p.trace[(int64_t)blockIdx.x * 48 + stamp_i++] = __builtin_amdgcn_s_memtime(); }
  // the constant-rate clock all XCDs share (100 MHz): when each workgroup starts / ends, at [256 * 48 + 2 workgroup + 0 / 1]
#define F3_REALTIME(K) if constexpr ((ABL & 16) != 0) { if (p.trace && tid == 0) p.trace[256 * 48 + 2 * (int64_t)blockIdx.x + (K)] = __builtin_amdgcn_s_memrealtime(); }
  F3_REALTIME(0)
  F3_STAMP()
  F3Sync sy{};
  if constexpr (!SLAB) f3_sync_begin(sy, p.sync + 8 * rb, p.coh_only ? 16 + sl : f3_xcc_id());
  const int nchunk = p.F / 32, per = nchunk / p.S, NC = per >> 1; // v1 chunks (32 units) of the layer / of this slice; 64-unit chunks
  const int c_base = sl * per;

  // ---- DMA schedule.  Phase p = 3C + k (k = 0: A(C), 1: B(C), 2: G(C) = w_2 of chunk C-1), the closing phase 3 NC = w_2 of the
  // last chunk; anything later is a placeholder (valid addresses, never read) that keeps the counted waits uniform.  A wave
  // owns fragments wid*8 .. wid*8 + 7 of every phase; fragment j of them sits at source offset
  // (pa (j >> 2) + pb ((j >> 1) & 1) + pc (j & 1)) KiB from `psrc`: w_1 phases (f = (ksl*2 + wcc)*2 + vg): pa = 1 (next
  // contraction step), pb = NKS (the other sub-chunk), pc = nchunk NKS (gate rows); w_2 phases (f = ct*4 + k): pa = 2 nchunk
  // (next column tile), pb = 2, pc = 1 (contraction steps).
  const unsigned char* psrc = nullptr;
  uint32_t pa = 0, pb = 0, pc = 0, pdst = 0;
  int pC = 0, pk = 0;                                            // the next phase to schedule
  const uint32_t ring0 = (uint32_t)(uintptr_t)(ffn_lds_byte*)ring;
  const uint32_t lane_off = (uint32_t)lane * 16u;
  auto schedule = [&](int slot) {
    const bool w2 = pk == 2 || pC >= NC;
    if (!w2) {
      const int c0 = c_base + 2 * pC;
      psrc = reinterpret_cast<const unsigned char*>(p.p1) + ((int64_t)(c0 * NKS + pk * 8 + wid * 2) << 10);
      pa = 1; pb = NKS; pc = (uint32_t)(nchunk * NKS);
    } else {
      const int cp = pC >= NC ? NC - 1 : (pC > 0 ? pC - 1 : 0);
      psrc = reinterpret_cast<const unsigned char*>(p.p2) + ((int64_t)(wid * 2 * (2 * nchunk) + 2 * (c_base + 2 * cp)) << 10);
      pa = (uint32_t)(2 * nchunk); pb = 2; pc = 1;
    }
    pdst = ring0 + (uint32_t)(slot * F3_PHASE + wid * 8192);
    if (++pk == 3) { pk = 0; ++pC; }
  };
  auto issue2 = [&](int i) {                                     // fragments 2i, 2i + 1 of the scheduled phase
    const uint32_t o = pa * (uint32_t)(i >> 1) + pb * (uint32_t)(i & 1);
    ffn_dma(psrc + ((uint64_t)o << 10), lane_off, pdst + (uint32_t)(2 * i) * 1024u);
    ffn_dma(psrc + ((uint64_t)(o + pc) << 10), lane_off, pdst + (uint32_t)(2 * i + 1) * 1024u);
  };

  schedule(0);
#pragma unroll
  for (int i = 0; i < 4; ++i) issue2(i);
  schedule(1);
#pragma unroll
  for (int i = 0; i < 4; ++i) issue2(i);
  // this wave's 64 activation rows as MFMA B operands, in registers for the whole kernel; staged through ring slots 2 and 3
  // (64 KiB), which receive their first DMAs only after the second barrier below
  uint4* xs = reinterpret_cast<uint4*>(ring + 2 * F3_PHASE);
  f3_stage_rows128<D>(xs, p.x16, rb, p.M, tid);
  for (int i = tid; i < per * 64; i += 256) {
    const int c = i >> 6, j = i & 63;
    bias_s[i] = p.b1[(j < 32 ? 0 : p.F) + (c_base + c) * 32 + (j & 31)];
  }
  f3_wait_lds();
  f3_barrier();
  if constexpr (!SLAB) f3_sync_publish(sy, sl, tid);
  otr_u32x4 xf[2][NKS];
#pragma unroll
  for (int rt = 0; rt < 2; ++rt)
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      const uint4 t = xs[(wr * 64 + 32 * rt + m) * (D / 8) + ((2 * ks + hi) ^ (m & 15))];
      xf[rt][ks] = otr_u32x4{t.x, t.y, t.z, t.w};
    }
  // The operand fragments must be KNOWN complete before the loop (ffn_fused.hip, lesson 2); "+a": they are MFMA operands only
  // -> the accumulator half of the register file
#pragma unroll
  for (int rt = 0; rt < 2; ++rt)
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) asm volatile("" : "+a"(xf[rt][ks]));
  f3_wait_lds();
  f3_barrier();                                                  // every wave has its fragments: slots 2 and 3 may be filled
  schedule(2);
#pragma unroll
  for (int i = 0; i < 4; ++i) issue2(i);
  schedule(3);                                                   // issued between the MFMA groups of phase 0
  f3_wait_vm<16>();                                              // phase 0 has landed (phases 1, 2 may fly)
  f3_barrier();
  F3_STAMP()

  f32x16 yacc[2][4];
#pragma unroll
  for (int rt = 0; rt < 2; ++rt)
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
      for (int r = 0; r < 16; ++r) yacc[rt][ct][r] = 0.f;
  f32x16 hv[2], hg[2];
  uint4 uown[2][2], upart[2][2];
#pragma unroll
  for (int rt = 0; rt < 2; ++rt)
#pragma unroll
    for (int j = 0; j < 2; ++j) uown[rt][j] = upart[rt][j] = make_uint4(0u, 0u, 0u, 0u);

  constexpr bool no_dma = (ABL & 1) != 0, no_mma = (ABL & 2) != 0, no_st = (ABL & 4) != 0;   // ablations are compile-time: a run-time flag would split
  // the G phase into basic blocks, and hipcc interleaves the GLU's VALU code with the MFMAs only inside one block
  uint4* my_u = reinterpret_cast<uint4*>(ubuf) + ((wr * 2 + wc) * 4) * 64 + lane;
  const uint4* partner_u = reinterpret_cast<const uint4*>(ubuf) + ((wr * 2 + (wc ^ 1)) * 4) * 64 + lane;
  // SAVE: row-major u leaves as whole 128-byte lines, one chunk late, from the hand-over buffer (ffn3_bwd_kernel: dh_store): a
  // row's 64 elements of a chunk = 8 pieces (sub-chunk w, step j, half h); this wave takes rows 32 wid .. + 31 of the row block,
  // instruction g rows 8g .. 8g + 7
  const int st_r = lane >> 3, st_q = lane & 7;
  const unsigned char* st_src = ubuf + ((((wid >> 1) * 2 + (st_q >> 2)) * 4 + (wid & 1) * 2 + ((st_q >> 1) & 1)) * 64 + st_r) * 16 + 8 * (st_q & 1);
  uint16_t* st_dst = SAVE ? p.usave + ((int64_t)rb * 128 + 32 * wid + st_r) * (int64_t)p.F + c_base * 32 + 8 * st_q : nullptr;
  auto u_store = [&](int CH, int g) {
    const unsigned char* sp = st_src + g * 128;
    const uint2 lo = *reinterpret_cast<const uint2*>(sp), up = *reinterpret_cast<const uint2*>(sp + 512);
    F3_ST_SAVE(st_dst + (int64_t)(8 * g) * (int64_t)p.F + CH * 64, make_uint4(lo.x, lo.y, up.x, up.y));
  };
  const int k_own = 2 * wc, k_par = 2 * (wc ^ 1);

  // all eight DMAs of the scheduled phase go out right behind the FIRST MFMA group of a phase (not two behind every group): every
  // fragment then has three phases to land, which matters when the weights come from beyond the L2 (5.41 -> 5.39 ms per step)
#define F3_ISSUE2(I) if constexpr (!no_dma) { if ((I) == 0) { issue2(0); issue2(1); issue2(2); issue2(3); } }
  // accumulators of GEMM1 start from the biases (register r of lane (m, hi) is hidden unit 8 (r >> 2) + 4 hi + (r & 3) of the
  // sub-chunk, for both row tiles), so the GLU adds nothing
#define F3_BIAS_INIT(CL)                                                                                       \
  _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                              \
    const float4 bv = *reinterpret_cast<const float4*>(bias_s + (CL) * 64 + 8 * q + 4 * hi);                   \
    const float4 bg = *reinterpret_cast<const float4*>(bias_s + (CL) * 64 + 32 + 8 * q + 4 * hi);              \
    _Pragma("unroll") for (int rt = 0; rt < 2; ++rt) {                                                         \
      hv[rt][4 * q] = bv.x; hv[rt][4 * q + 1] = bv.y; hv[rt][4 * q + 2] = bv.z; hv[rt][4 * q + 3] = bv.w;       \
      hg[rt][4 * q] = bg.x; hg[rt][4 * q + 1] = bg.y; hg[rt][4 * q + 2] = bg.z; hg[rt][4 * q + 3] = bg.w;       \
    }                                                                                                          \
  }                                                                                                            \
  asm volatile("s_nop 3" ::: "memory");     /* VALU / LDS write of an accumulator -> asm MFMA reading it as C */
  // GEMM1 over 8 contraction steps (HALF = 0: steps 0-7, 1: steps 8-15) of the phase in ring slot SLOT; the two DMAs of the
  // scheduled phase ride behind every group of 8 MFMAs
#define F3_GEMM1(HALF, SLOT, STP, CHP)                                                                         \
  if constexpr (!no_mma) {                                                                                     \
    const otr_u32x4* wb = reinterpret_cast<const otr_u32x4*>(ring + (SLOT) * F3_PHASE) + (wc * 2) * 64 + lane; \
    otr_u32x4 fr[2][4];                                                                                        \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) fr[0][j] = wb[(((j >> 1)) * 4 + (j & 1)) * 64];               \
    _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                                            \
      if (g + 1 < 4) {                                                                                         \
        _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                          \
          fr[(g + 1) & 1][j] = wb[((2 * (g + 1) + (j >> 1)) * 4 + (j & 1)) * 64];                              \
      }                                                                                                        \
      __builtin_amdgcn_sched_barrier(0);                                                                       \
      _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                          \
        const int ks = (HALF) * 8 + 2 * g + (j >> 1);                                                          \
        if (j & 1) { f3_mma_xa(hg[0], fr[g & 1][j], xf[0][ks]); f3_mma_xa(hg[1], fr[g & 1][j], xf[1][ks]); }   \
        else       { f3_mma_xa(hv[0], fr[g & 1][j], xf[0][ks]); f3_mma_xa(hv[1], fr[g & 1][j], xf[1][ks]); }   \
      }                                                                                                        \
      if constexpr (SAVE && (STP) && !no_st) u_store(CHP, g);    /* the previous chunk's u */                    \
      F3_ISSUE2(g)                                                                                             \
      __builtin_amdgcn_sched_barrier(0);                                                                       \
    }                                                                                                          \
  } else {                                                                                                     \
    _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                                            \
      if constexpr (SAVE && (STP) && !no_st) u_store(CHP, g);                                                  \
      F3_ISSUE2(g)                                                                                             \
    }                                                                                                          \
  }
  // end of a phase: this wave's DMAs of the NEXT phase have landed, its LDS traffic is done; after the barrier the slot just
  // consumed is free for the phase four ahead, which is scheduled here and issued during the next phase.  The wait is COUNTED
  // (vmcnt retires in issue order, stores included): everything this wave issued after the last DMA of the next phase's
  // group may still fly -- the 16 DMAs of the two phases after it plus EXTRA = the global stores of those two phases (SAVE: 12
  // per G phase, unconditional so that the count is exact)
#define F3_PHASE_END(EXTRA)                                                                                    \
  if constexpr (no_dma) f3_wait_vm<0>(); else f3_wait_vm<16 + (EXTRA)>();                                      \
  f3_wait_lds();                                                                                               \
  f3_barrier();                                                                                                \
  schedule(slot);                                                                                              \
  slot = (slot + 1) & 3;                                                                                       \
  F3_STAMP()
  // Phase G of chunk C.  G2: GEMM2 of chunk C-1 -- w_2 fragment (ct, k) at ring[(ct*4 + k) KiB]; this wave's column tiles are
  // 4 wc .. 4 wc + 3; contraction steps k_own, k_own + 1 take u from registers (uown), k_par, k_par + 1 the partner's (upart,
  // read from the hand-over buffer during phase B).  GLU: the GLU of chunk C in quarters; quarter kk REPLACES
  // uown[kk >> 1][kk & 1], which step (kk & 1) <= kk of GEMM2 has already consumed.  One basic block; per step 8 MFMAs with the
  // quarter's ~45 VALU instructions pinned between them (sched_group_barrier: 1 MFMA, then 6 VALU), so the matrix pipe runs
  // during the VALU phase.
#define F3_PHASE_G(G2, GLU, SLOT, CHUNK)                                                                              \
  if constexpr (!no_mma) {                                                                                     \
    const uint4* wb = reinterpret_cast<const uint4*>(ring + (SLOT) * F3_PHASE) + (wc * 16) * 64 + lane;        \
    uint4 fr[2][4];                                                                                            \
    if constexpr (GLU) F3_MFMA_DRAIN();      /* GEMM1's asm MFMAs -> the GLU's VALU reads (also behind the barrier) */ \
    if constexpr (G2) {                                                                                        \
      _Pragma("unroll") for (int ct = 0; ct < 4; ++ct) fr[0][ct] = wb[(ct * 4 + k_own) * 64];                  \
    }                                                                                                          \
    _Pragma("unroll") for (int kk = 0; kk < 4; ++kk) {                                                         \
      if constexpr (G2) {                                                                                      \
        if (kk + 1 < 4) {                                                                                      \
          const int kn = (kk + 1 < 2 ? k_own : k_par) + ((kk + 1) & 1);                                        \
          _Pragma("unroll") for (int ct = 0; ct < 4; ++ct) fr[(kk + 1) & 1][ct] = wb[(ct * 4 + kn) * 64];      \
        }                                                                                                      \
      }                                                                                                        \
      __builtin_amdgcn_sched_barrier(0);                                                                       \
      if constexpr (G2) {                                                                                      \
        const uint4 u0 = kk < 2 ? uown[0][kk & 1] : upart[0][kk & 1], u1 = kk < 2 ? uown[1][kk & 1] : upart[1][kk & 1]; \
        _Pragma("unroll") for (int ct = 0; ct < 4; ++ct) {                                                     \
          mma32(yacc[0][ct], fr[kk & 1][ct], u0);                                                              \
          mma32(yacc[1][ct], fr[kk & 1][ct], u1);                                                              \
        }                                                                                                      \
      }                                                                                                        \
      if constexpr (GLU) {                   /* quarter kk: row tile kk >> 1, registers 8 (kk & 1) .. + 7 */       \
        const int rt = kk >> 1, j0 = (kk & 1) * 8;                                                             \
        float u[8], sg[8];                                                                                     \
        _Pragma("unroll") for (int e = 0; e < 8; ++e) { sg[e] = fast_sigmoid(hg[rt][j0 + e]); u[e] = hv[rt][j0 + e] * sg[e]; } \
        const uint4 nu = make_uint4(pack2h(u[0], u[1]), pack2h(u[2], u[3]), pack2h(u[4], u[5]), pack2h(u[6], u[7])); \
        uown[rt][kk & 1] = nu;               /* kept for GEMM2 of this chunk one iteration later ... */            \
        my_u[(rt * 2 + (kk & 1)) * 64] = nu; /* ... and handed to the partner wave */                            \
        if constexpr (SAVE && !no_st) {      /* 2 global stores per quarter (row-major u: u_store) */              \
          uint4* hs = p.hsave + ((int64_t)(((rb * 4 + sl) * NC + (CHUNK)) * 4 + wid) * 8) * 64 + lane;         \
          F3_ST_SAVE(hs + (rt * 2 + (kk & 1)) * 64,                                                            \
                         make_uint4(pack2h(hv[rt][j0], hv[rt][j0 + 1]), pack2h(hv[rt][j0 + 2], hv[rt][j0 + 3]), \
                                    pack2h(hv[rt][j0 + 4], hv[rt][j0 + 5]), pack2h(hv[rt][j0 + 6], hv[rt][j0 + 7]))); \
          F3_ST_SAVE(hs + (4 + rt * 2 + (kk & 1)) * 64,                                                        \
                         make_uint4(pack2h(sg[0], sg[1]), pack2h(sg[2], sg[3]), pack2h(sg[4], sg[5]), pack2h(sg[6], sg[7]))); \
        }                                                                                                      \
      }                                                                                                        \
      if constexpr (G2 && GLU) {                                                                               \
        _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                                        \
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     /* one MFMA */                                \
          __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);     /* six VALU */                                \
        }                                                                                                      \
      }                                                                                                        \
      __builtin_amdgcn_sched_barrier(0);                                                                       \
      if constexpr (GLU) { F3_ISSUE2(kk) }   /* the closing phase (no GLU) schedules nothing */                  \
      __builtin_amdgcn_sched_barrier(0);                                                                       \
    }                                                                                                          \
  } else if constexpr (GLU) {                                                                                  \
    _Pragma("unroll") for (int kk = 0; kk < 4; ++kk) { F3_ISSUE2(kk) }                                         \
  }
  // the partner's u of the previous chunk: written before the barrier that closed G(C-1), rewritten in G(C) behind the
  // barrier that closes this phase B(C)
#define F3_READ_PARTNER()                                                                                      \
  _Pragma("unroll") for (int rt = 0; rt < 2; ++rt)                                                             \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) upart[rt][j] = partner_u[(rt * 2 + j) * 64];

  // Stores do NOT retire in issue order with loads (measured: with the 12 stores of a G phase added to the count, ring slots were
  // read before their DMA had landed -- 0.5 % wrong outputs): a store is acknowledged by the L2 long before an older LDS-DMA
  // returns.  Loads retire in order among themselves, so the count keeps the 16 YOUNGER LOADS only; the stores in flight merely
  // make the wait a little longer than needed.
  constexpr int KG = 0;
  int slot = 0;                                                  // ring slot of the current phase (phase index mod 4)
  // ---- chunk 0: A, B, GLU only
  F3_BIAS_INIT(wc)
  F3_GEMM1(0, slot, false, 0)
  F3_PHASE_END(0)
  F3_GEMM1(1, slot, false, 0)
  F3_PHASE_END(0)
  F3_PHASE_G(false, true, slot, 0)
  F3_PHASE_END(KG)
  // ---- chunks 1 .. NC-1: A, B, GLU beside the previous chunk's GEMM2
  for (int C = 1; C < NC; ++C) {
    F3_BIAS_INIT(2 * C + wc)
    F3_GEMM1(0, slot, true, C - 1)
    F3_PHASE_END(KG)                                             // the G phase before this one
    F3_READ_PARTNER()
    F3_GEMM1(1, slot, false, 0)
    F3_PHASE_END(0)
    F3_PHASE_G(true, true, slot, C)
    F3_PHASE_END(KG)
  }
  // ---- closing phase: GEMM2 of chunk NC-1 (its w_2 took the place of a phase A(NC)); the partners' ids travel meanwhile
  if constexpr (!SLAB) f3_sync_read_ids(sy);
  F3_READ_PARTNER()
  if constexpr (SAVE && !no_st) {
#pragma unroll
    for (int g = 0; g < 4; ++g) u_store(NC - 1, g);              // the last chunk's u
  }
  F3_PHASE_G(true, false, slot, 0)
#undef F3_ISSUE2
#undef F3_BIAS_INIT
#undef F3_GEMM1
#undef F3_PHASE_END
#undef F3_PHASE_G
#undef F3_READ_PARTNER
  F3_STAMP()
  f3_wait_vm<0>();                                               // the placeholder DMAs have landed: the ring becomes scratch
  f3_wait_lds();
  f3_barrier();
  F3_STAMP()
  if constexpr (SLAB) {
    // ---- slab form: no exchange, no LayerNorm here.  The exchange below is a chain of far round trips (write-through drain,
    // arrival atomic, partners' data, LayerNorm: 36 k of this kernel's 108 k cycles by clock stamps); a launch boundary is cheaper,
    // and the q|k|v projection that reads this sub-layer's output finishes the LayerNorm in its prologue (rowblock.hip)
    f3_store_slab(yacc, ring, p.slab + (int64_t)sl * p.M * 256, rb, p.M, wid, wr, wc, lane);
    F3_STAMP()
    F3_REALTIME(1)
    return;
  }

  // ---- fused form (S = 4): the four workgroups of a row block exchange their partial sums and each finishes ONE quarter of
  // the rows (32 rows: quarter `sl`): y = LayerNorm(x + dropout(sum of the four partials + b_2)).  Everything stays in the
  // accumulator layout -- tile (ct, q): lane (m, hi) holds columns 32 ct + 8 q + 4 hi .. + 3 of row m as one float4 -- so the
  // exchange is lane-linear 1 KiB pieces (coalesced both ways) and a row's statistics need one cross-lane step (hi) plus the
  // four waves' column shares meeting in LDS.
  //   1. own quarter -> LDS, the three foreign quarters -> scratch[row block][sl][quarter] with write-through stores
  //      (sc0 sc1: at memory when vmcnt drains, whatever XCD the reader sits on)
  //   2. arrive (one atomic per workgroup), wait for all four (bounded spin; a give-up is reported through the fault word)
  //   3. wave w sums column tiles 2w, 2w+1 of its quarter: own (LDS) + three partners (sc1 loads: served past the L1)
  //   4. bias, dropout (the mask otr_add_layernorm_bwd regenerates), residual, LayerNorm; y / y16 / z / mean / rstd
  float* own = reinterpret_cast<float*>(ring);                   // [8 column tiles][4 q][64 lanes] float4 = 32 KiB
  float* red = reinterpret_cast<float*>(ring + 32768);           // [2 passes][4 waves][32 rows]
  auto rs = __builtin_amdgcn_make_buffer_rsrc(p.scratch + (int64_t)rb * (4 * 4 * 8192), 0, 4 * 4 * 32768, 0x00020000);
  f3_send_partials(yacc, rs, own, sy, sl, wr, wc, lane);
  F3_STAMP()
  // everything the quarter's epilogue reads besides the partials is fetched before the arrival wait
  const int64_t row = (int64_t)rb * 128 + 32 * sl + m;
  const bool live = row < p.M;
  const int64_t crow = live ? row : (int64_t)p.M - 1;
  float4 xr[2][4], b2r[2][4], gmr[2][4], btr[2][4];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int col = 32 * (2 * wid + t) + 8 * q + 4 * hi;
      xr[t][q] = *reinterpret_cast<const float4*>(p.x + crow * D + col);
      b2r[t][q] = *reinterpret_cast<const float4*>(p.b2 + col);
      gmr[t][q] = *reinterpret_cast<const float4*>(p.gamma + col);
      btr[t][q] = *reinterpret_cast<const float4*>(p.beta + col);
    }
  F3_STAMP()
  f3_wait_vm<0>();                                               // this wave's write-through stores are at memory
  f3_wait_lds();
  f3_barrier();
  F3_STAMP()
  if (tid == 0) f3_arrive_wait(sy, p.spin_limit, p.fault);
  f3_barrier();
  F3_STAMP()
  // ---- the quarter's rows: lane (m, hi) of wave `wid` owns row m, columns 32 ct + 8 q + 4 hi .. + 3 for ct = 2 wid, 2 wid + 1
  otr_u32x4 part[3][2][4];
  f3_recv_partials(part, rs, sy, sl, wid, lane);
  const bool drop = p.p_drop > 0.f;
  const uint64_t seed = drop ? *p.seed : 0;
  const uint32_t thr = drop ? (uint32_t)fminf(p.p_drop * 4294967296.f, 4294967295.f) : 0;
  const float inv_keep = drop ? 1.f / (1.f - p.p_drop) : 1.f;
  float v[2][16];
  float sm = 0.f;
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int col = 32 * (2 * wid + t) + 8 * q + 4 * hi;
      const float4 o = *reinterpret_cast<const float4*>(own + (((2 * wid + t) * 4 + q) * 64 + lane) * 4);
      const float4 b2 = b2r[t][q];
      float a4[4] = {o.x + b2.x, o.y + b2.y, o.z + b2.z, o.w + b2.w};
#pragma unroll
      for (int n = 0; n < 3; ++n) {
        a4[0] += __uint_as_float(part[n][t][q].x); a4[1] += __uint_as_float(part[n][t][q].y);
        a4[2] += __uint_as_float(part[n][t][q].z); a4[3] += __uint_as_float(part[n][t][q].w);
      }
      const float x4[4] = {xr[t][q].x, xr[t][q].y, xr[t][q].z, xr[t][q].w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float sc = 1.f;
        if (drop) sc = otr_rand32(seed, p.rng_offset + (uint64_t)(crow * D + col + e)) >= thr ? inv_keep : 0.f;
        const float zz = x4[e] + a4[e] * sc;
        v[t][4 * q + e] = zz;
        sm += zz;
      }
    }
  F3_STAMP()
  sm += __shfl_xor(sm, 32);
  if (hi == 0) red[wid * 32 + m] = sm;
  f3_wait_lds();
  f3_barrier();
  const float mean = (red[m] + red[32 + m] + red[64 + m] + red[96 + m]) * (1.f / D);
  float qq = 0.f;
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) { const float d_ = v[t][r] - mean; qq += d_ * d_; }
  qq += __shfl_xor(qq, 32);
  if (hi == 0) red[128 + wid * 32 + m] = qq;
  f3_wait_lds();
  f3_barrier();
  const float rstd = rsqrtf((red[128 + m] + red[160 + m] + red[192 + m] + red[224 + m]) * (1.f / D) + p.eps);
  if (live) {
    if (p.z) {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          *reinterpret_cast<float4*>(p.z + row * D + 32 * (2 * wid + t) + 8 * q + 4 * hi) =
              make_float4(v[t][4 * q], v[t][4 * q + 1], v[t][4 * q + 2], v[t][4 * q + 3]);
    }
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 gm = gmr[t][q], bt = btr[t][q];
        float* vv = &v[t][4 * q];
        vv[0] = (vv[0] - mean) * rstd * gm.x + bt.x; vv[1] = (vv[1] - mean) * rstd * gm.y + bt.y;
        vv[2] = (vv[2] - mean) * rstd * gm.z + bt.z; vv[3] = (vv[3] - mean) * rstd * gm.w + bt.w;
        *reinterpret_cast<float4*>(p.y + row * D + 32 * (2 * wid + t) + 8 * q + 4 * hi) = make_float4(vv[0], vv[1], vv[2], vv[3]);
      }
    if (p.y16) {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          *reinterpret_cast<uint2*>(p.y16 + row * D + 32 * (2 * wid + t) + 8 * q + 4 * hi) =
              make_uint2(pack2h(v[t][4 * q], v[t][4 * q + 1]), pack2h(v[t][4 * q + 2], v[t][4 * q + 3]));
    }
    if (wid == 0 && hi == 0) { p.mean[row] = mean; p.rstd[row] = rstd; }
  }
  F3_STAMP()
#undef F3_STAMP
}


// ================================================================================================ backward
// dx = skip + dh . w_1 with dh = GLU'(saved (value, sigmoid), du), du = dy . w_2; dh (16-bit, row-major) is the operand of the
// w_1 weight gradient (its column sums = the w_1 bias gradient ride along in that launch).  The hidden pre-activations are
// NOT recomputed: the forward kernel left (value + bias, sigmoid(gate)) in accumulator-tile order (Ffn3FwdArgs::hsave), which
// this kernel reads back piece by piece (1 KiB coalesced loads) -- the backward pass is then the forward pass mirrored:
//   D   du^T[32 hidden of sub-chunk wc, 64 rows] = w_2^T frags (LDS) x dy^T frags (ACCUMULATOR registers, 128)   32 MFMAs
//   GLU' on the accumulators (du) and the saved tiles -> dvalue, dgate: 16-bit B-operand fragments (registers + hand-over to
//       the partner wave) and row-major dh (global)
//   XA  dx^T[128 columns of half wc, 64 rows] += w_1^T frags x OWN dh frags                                         32 MFMAs
//   XB  ... += w_1^T frags x the PARTNER's dh frags (read from the hand-over buffer during the next D phase)           32 MFMAs
// software-pipelined like the forward kernel: iteration C = D(C), XA(C-1) beside the first half of GLU'(C), XB(C-1) beside
// the second half.  Same ring, same counted waits; the 4 loads of the NEXT chunk's saved tiles an X phase issues (two phases ahead
// of their use) sit behind the phase's eight DMAs, so the counts are exact; row-major dh leaves one chunk late, during
// the D phase, as whole lines (dh_store).
struct Ffn3BwdArgs {
  const uint16_t* dy16;    // [M, D]
  const uint4* hsave;      // forward's (value, sigmoid) tiles
  const uint4* p3;         // w_2^T packed: rows = F hidden units, contraction = D, perm 0
  const uint4* p4;         // w_1^T packed: rows = D, contraction = 2F (value then gate), perm 1
  uint16_t* dh;            // [128 * row blocks, 2F] row-major out
  const float* skip;       // [M, D] or NULL
  float* dx;               // [M, D] out (may alias skip)
  float* scratch; int* sync; int* fault; int spin_limit, coh_only, map;
  int M, F;
  uint16_t* slab;          // SLAB: [4 slices][M][256] 16-bit out -- this slice's share dh[slice] . w_1[slice] of the input gradient (skip, dx,
                           // scratch, sync are not touched then: the consumer adds the shares to the skip gradient, rowblock.hip)
  unsigned long long* trace;   // tuning hook (otr_debug_trace), as Ffn3FwdArgs::trace
};

// a global load hipcc does not count (its own s_waitcnt would drain the DMAs in flight): 16 B per lane from a wave-uniform base.
// Form (ii) of cdna_hip_programming.md 5.7: the load, and before the first consumer a counted wait that names the destination.
// The destination is "+v", not "=v": it lives across the loop back-edge, and with a plain output hipcc may give the loop-carried
// value another register and copy the "defined" one there while the load is still in flight; read-modify-write
// ties the asm to the ONE register the loop carries.
__device__ __forceinline__ void f3_gload(otr_u32x4& dst, const void* uniform_src, uint32_t lane_off) {
  asm volatile("global_load_dwordx4 %0, %1, %2" : "+v"(dst) : "v"(lane_off), "s"(uniform_src) : "memory");
}
template <int N> __device__ __forceinline__ void f3_wait_vm_for(otr_u32x4& a, otr_u32x4& b, otr_u32x4& c, otr_u32x4& d) {
  asm volatile("s_waitcnt vmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N) : "memory");
}
__device__ __forceinline__ void f3_mma_acc(f32x16& acc, const otr_u32x4& w, const otr_u32x4& b_acc) {
  asm volatile("s_nop 1\n\t" F3_MFMA_OP " %0, %1, %2, %0" : "+v"(acc) : "v"(w), "a"(b_acc));
}
__device__ __forceinline__ float f3_h2f_lo(uint32_t w) { return h2f_lo(w); }

template <int D, int ABL, bool SLAB = false>
__global__ __launch_bounds__(256, 1) void ffn3_bwd_kernel(Ffn3BwdArgs p) {
  static_assert(D == 256, "two 128-column halves, 16 contraction steps");
  constexpr int NKS = D / 16;
  constexpr int HAND = 32 * 1024;        // dh hand-over: [wave][row tile][k: dvalue 0, 1, dgate 0, 1] fragments of 1 KiB
  __shared__ __attribute__((aligned(1024))) unsigned char smem[F3_RING + HAND];
  unsigned char* ring = smem;
  unsigned char* hand = smem + F3_RING;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wid >> 1, wc = wid & 1;
  const int m = lane & 31, hi = lane >> 5;
  int rb, sl;
  f3_block_map((int)blockIdx.x, p.map, rb, sl);
  if (rb * 128 >= p.M) return;
  const int row0 = rb * 128 + wr * 64;
  int stamp_i = 0;
#define F3B_STAMP() if constexpr ((ABL & 16) != 0) { if (p.trace && tid == 0 && stamp_i < 48) p.trace[(int64_t)blockIdx.x * 48 + stamp_i++] = __builtin_amdgcn_s_memtime(); }
  F3B_STAMP()
  F3Sync sy{};
  if constexpr (!SLAB) f3_sync_begin(sy, p.sync + 8 * rb, p.coh_only ? 16 + sl : f3_xcc_id());
  const int nchunk = p.F / 32, per = nchunk / 4, NC = per >> 1;
  const int c_base = sl * per;
  constexpr bool no_dma = (ABL & 1) != 0, no_mma = (ABL & 2) != 0, no_st = (ABL & 4) != 0, no_hl = (ABL & 8) != 0;

  // ---- DMA schedule (see ffn3_fwd_kernel).  Phase 3C + k: k = 0: D(C) = w_2^T of chunk C (fragment f = ks*2 + wcc); k = 1 / 2:
  // XA / XB of chunk C-1 = w_1^T fragments for the OWN / PARTNER step: f = (wcc*4 + ctl)*4 + j4 -- column tile 4 wcc + ctl,
  // contraction steps j4 = {value 0, 1, gate 0, 1} of sub-chunk wcc (own step) or 1 - wcc (partner step); phases 3 NC and
  // 3 NC + 1 close with XA / XB of the last chunk.  This wave's fragments are wid*8 + j; source offset of fragment j:
  // (pa (j >> 2) + pb ((j >> 1) & 1) + pc (j & 1)) KiB.
  const unsigned char* psrc = nullptr;
  uint32_t pa = 0, pb = 0, pc = 0, pdst = 0;
  int pC = 0, pk = 0;
  const uint32_t ring0 = (uint32_t)(uintptr_t)(ffn_lds_byte*)ring;
  const uint32_t lane_off = (uint32_t)lane * 16u;
  auto schedule = [&](int slot) {
    int kind, ch;                                                // kind 0: w_2^T, 1: own step, 2: partner step
    if (pC < NC) { kind = pk; ch = pk == 0 ? pC : (pC > 0 ? pC - 1 : 0); }
    else { kind = (pC == NC && pk == 0) ? 1 : 2; ch = NC - 1; }
    if (kind == 0) {
      psrc = reinterpret_cast<const unsigned char*>(p.p3) + ((int64_t)((c_base + 2 * ch) * NKS + wid * 4) << 10);
      pa = 2; pb = 1; pc = NKS;                                  // f = wid*8 + j: ks = wid*4 + (j >> 1), wcc = j & 1
    } else {
      const int wcc = wid >> 1, sc = kind == 1 ? wcc : (wcc ^ 1);
      const int c = c_base + 2 * ch + sc;
      psrc = reinterpret_cast<const unsigned char*>(p.p4) + ((int64_t)((4 * wcc + (wid & 1) * 2) * (4 * nchunk) + 2 * c) << 10);
      pa = (uint32_t)(4 * nchunk); pb = (uint32_t)(2 * nchunk); pc = 1;   // f = wid*8 + j: ctl = (wid & 1)*2 + (j >> 2), j4 = j & 3
    }
    pdst = ring0 + (uint32_t)(slot * F3_PHASE + wid * 8192);
    if (++pk == 3) { pk = 0; ++pC; }
  };
  auto issue2 = [&](int i) {
    const uint32_t o = pa * (uint32_t)(i >> 1) + pb * (uint32_t)(i & 1);
    ffn_dma(psrc + ((uint64_t)o << 10), lane_off, pdst + (uint32_t)(2 * i) * 1024u);
    ffn_dma(psrc + ((uint64_t)(o + pc) << 10), lane_off, pdst + (uint32_t)(2 * i + 1) * 1024u);
  };
#define F3B_ISSUE2(I) if constexpr (!no_dma) { if ((I) == 0) { issue2(0); issue2(1); issue2(2); issue2(3); } }   /* see ffn3_fwd_kernel */

  schedule(0);
#pragma unroll
  for (int i = 0; i < 4; ++i) issue2(i);
  schedule(1);
#pragma unroll
  for (int i = 0; i < 4; ++i) issue2(i);
  // this wave's 64 rows of dy as MFMA B operands, in (accumulator) registers for the whole kernel; staged through ring slots 2
  // and 3 (f3_stage_rows128)
  uint4* xs = reinterpret_cast<uint4*>(ring + 2 * F3_PHASE);
  f3_stage_rows128<D>(xs, p.dy16, rb, p.M, tid);
  f3_wait_lds();
  f3_barrier();
  if constexpr (!SLAB) f3_sync_publish(sy, sl, tid);
  otr_u32x4 dyf[2][NKS];
#pragma unroll
  for (int rt = 0; rt < 2; ++rt)
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      const uint4 t = xs[(wr * 64 + 32 * rt + m) * (D / 8) + ((2 * ks + hi) ^ (m & 15))];
      dyf[rt][ks] = otr_u32x4{t.x, t.y, t.z, t.w};
    }
#pragma unroll
  for (int rt = 0; rt < 2; ++rt)
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) asm volatile("" : "+a"(dyf[rt][ks]));
  f3_wait_lds();
  f3_barrier();                                                  // every wave has its fragments: slots 2 and 3 may be filled
  // the saved (value, sigmoid) tiles of chunk C for this wave: 8 pieces; hp[0..3] = row tile 0 (value 0, 1, sigmoid 0, 1),
  // hp[4..7] = row tile 1
  const unsigned char* hbase = reinterpret_cast<const unsigned char*>(p.hsave) + ((int64_t)((rb * 4 + sl) * NC) * 4 + wid) * 8192;
  otr_u32x4 hp[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) hp[i] = otr_u32x4{0u, 0u, 0u, 0u};
  auto hload = [&](int C, int half) {                            // 4 pieces of row tile `half`
    const unsigned char* b = hbase + (int64_t)C * (4 * 8192);
    f3_gload(hp[4 * half + 0], b + (half * 2 + 0) * 1024, lane_off);
    f3_gload(hp[4 * half + 1], b + (half * 2 + 1) * 1024, lane_off);
    f3_gload(hp[4 * half + 2], b + (4 + half * 2 + 0) * 1024, lane_off);
    f3_gload(hp[4 * half + 3], b + (4 + half * 2 + 1) * 1024, lane_off);
  };
  hload(0, 0);
  hload(0, 1);
  schedule(2);
#pragma unroll
  for (int i = 0; i < 4; ++i) issue2(i);
  schedule(3);
  f3_wait_vm_for<8>(hp[0], hp[1], hp[2], hp[3]);                 // phases 0, 1 and the first tiles have landed (phase 2 may fly)
  f3_wait_vm_for<8>(hp[4], hp[5], hp[6], hp[7]);
  f3_barrier();

  f32x16 xacc[2][4];
#pragma unroll
  for (int rt = 0; rt < 2; ++rt)
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
      for (int r = 0; r < 16; ++r) xacc[rt][ct][r] = 0.f;
  f32x16 du[2];
  uint4 dho[2][4], dhp[2][4];                                    // own / partner dh fragments of the previous chunk
#pragma unroll
  for (int rt = 0; rt < 2; ++rt)
#pragma unroll
    for (int k = 0; k < 4; ++k) dho[rt][k] = dhp[rt][k] = make_uint4(0u, 0u, 0u, 0u);
  uint4* my_h = reinterpret_cast<uint4*>(hand) + (wid * 8) * 64 + lane;
  const uint4* partner_h = reinterpret_cast<const uint4*>(hand) + ((wid ^ 1) * 8) * 64 + lane;
  // Row-major dh leaves as WHOLE 128-byte lines, one chunk late, from the hand-over buffer (where both partner waves' fragments
  // of a chunk meet anyway): a row's 64 dvalue (64 dgate) elements of a chunk are 8 pieces of 16 B = (sub-chunk w, contraction
  // step k, half h), and piece (w, k, h) is the 8-byte half h of lanes (row, hi = 0) and (row, hi = 1) of fragment (wave (wr, w),
  // row tile, k).  This wave takes rows 32 wid .. + 31 of the row block (row tile wid & 1 of wave row wid >> 1); instruction
  // t = 4 seg + g covers rows 8g .. 8g + 7 of segment seg (0 dvalue, 1 dgate), 8 lanes per row.  (Stored straight from the
  // accumulator layout -- 32-byte pieces of 32 different rows per instruction -- the stores cost 20 us of a 63 us launch and
  // held up the tile loads behind them: the CU's address path takes about a cycle per line touched.)
  const int st_r = lane >> 3, st_q = lane & 7;
  const unsigned char* st_src = hand + ((((wid >> 1) * 2 + (st_q >> 2)) * 8 + (wid & 1) * 4 + ((st_q >> 1) & 1)) * 64 + st_r) * 16 + 8 * (st_q & 1);
  uint16_t* st_dst = p.dh + ((int64_t)rb * 128 + 32 * wid + st_r) * (2 * (int64_t)p.F) + c_base * 32 + 8 * st_q;
  auto dh_store = [&](int CH, int t) {
    const unsigned char* sp = st_src + (t >> 2) * 2048 + (t & 3) * 128;
    const uint2 lo = *reinterpret_cast<const uint2*>(sp), up = *reinterpret_cast<const uint2*>(sp + 512);
    F3_ST_SAVE(st_dst + (int64_t)(8 * (t & 3)) * (2 * (int64_t)p.F) + (t >> 2) * (int64_t)p.F + CH * 64, make_uint4(lo.x, lo.y, up.x, up.y));
  };

  // D: du = w_2^T . dy over 16 contraction steps, 8 MFMAs per group of 4 fragments, two DMAs behind every group
#define F3B_PHASE_D(SLOT, STP, CHP)                                                                            \
  if constexpr (!no_mma) {                                                                                     \
    _Pragma("unroll") for (int rt = 0; rt < 2; ++rt)                                                           \
      _Pragma("unroll") for (int r = 0; r < 16; ++r) du[rt][r] = 0.f;                                          \
    asm volatile("s_nop 3" ::: "memory");                                                                      \
    const otr_u32x4* wb = reinterpret_cast<const otr_u32x4*>(ring + (SLOT) * F3_PHASE) + wc * 64 + lane;       \
    otr_u32x4 fr[2][4];                                                                                        \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) fr[0][j] = wb[(j * 2) * 64];                                 \
    _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                                            \
      if (g + 1 < 4) {                                                                                         \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) fr[(g + 1) & 1][j] = wb[((4 * (g + 1) + j) * 2) * 64];   \
      }                                                                                                        \
      __builtin_amdgcn_sched_barrier(0);                                                                       \
      _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                          \
        f3_mma_acc(du[0], fr[g & 1][j], dyf[0][4 * g + j]);                                                    \
        f3_mma_acc(du[1], fr[g & 1][j], dyf[1][4 * g + j]);                                                    \
      }                                                                                                        \
      if constexpr ((STP) && !no_st) { dh_store(CHP, 2 * g); dh_store(CHP, 2 * g + 1); }   /* the previous chunk's dh */ \
      F3B_ISSUE2(g)                                                                                            \
      __builtin_amdgcn_sched_barrier(0);                                                                       \
    }                                                                                                          \
  } else {                                                                                                     \
    _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                                            \
      if constexpr ((STP) && !no_st) { dh_store(CHP, 2 * g); dh_store(CHP, 2 * g + 1); }                       \
      F3B_ISSUE2(g)                                                                                            \
    }                                                                                                          \
  }
#define F3B_PHASE_END(KEEP)                                                                                    \
  if constexpr (no_dma) f3_wait_vm<0>(); else f3_wait_vm<(KEEP)>();                                            \
  f3_wait_lds();                                                                                               \
  f3_barrier();                                                                                                \
  schedule(slot);                                                                                              \
  slot = (slot + 1) & 3;
  // GLU' of quarter (RT, J) of chunk CH: registers 8J .. 8J+7 of row tile RT.  value a, sigmoid s (saved), d = du:
  //   dvalue = d s,  dgate = d a s (1 - s);  fragments: dvalue -> contraction step J, dgate -> step 2 + J of this sub-chunk
#define F3B_GLU(RT, J, NDH)                                                                                    \
  {                                                                                                            \
    const otr_u32x4 pa_ = hp[4 * (RT) + (J)], ps_ = hp[4 * (RT) + 2 + (J)];                                    \
    const uint32_t aw[4] = {pa_.x, pa_.y, pa_.z, pa_.w}, sw[4] = {ps_.x, ps_.y, ps_.z, ps_.w};                 \
    float da[8], dg[8];                                                                                        \
    _Pragma("unroll") for (int e = 0; e < 8; ++e) {                                                            \
      const float a = (e & 1) ? h2f_hi(aw[e >> 1]) : h2f_lo(aw[e >> 1]);                                       \
      const float sg = (e & 1) ? h2f_hi(sw[e >> 1]) : h2f_lo(sw[e >> 1]);                                      \
      const float d_ = du[RT][8 * (J) + e];                                                                    \
      da[e] = d_ * sg;                                                                                         \
      const float t_ = d_ * a * sg;                                                                            \
      dg[e] = t_ - t_ * sg;                                                                                    \
    }                                                                                                          \
    NDH[RT][J] = make_uint4(pack2h(da[0], da[1]), pack2h(da[2], da[3]), pack2h(da[4], da[5]), pack2h(da[6], da[7])); \
    NDH[RT][2 + (J)] = make_uint4(pack2h(dg[0], dg[1]), pack2h(dg[2], dg[3]), pack2h(dg[4], dg[5]), pack2h(dg[6], dg[7])); \
    my_h[((RT) * 4 + (J)) * 64] = NDH[RT][J];                                                                  \
    my_h[((RT) * 4 + 2 + (J)) * 64] = NDH[RT][2 + (J)];                                                        \
  }
  // row tile RT of chunk CH complete: the loads of the NEXT chunk's saved tiles for this row tile (4 global loads), in front of
  // the phase's last DMA pair
#define F3B_STORE_LOAD(RT, CH, NDH)                                                                            \
  {                                                                                                            \
    if constexpr (!no_hl) hload((CH) + 1 < NC ? (CH) + 1 : (CH), RT);   /* past the end: a reload of valid tiles, never used */ \
  }
  // X phase: 32 MFMAs in 4 steps (j4) of 8 over w_1^T fragments [(wc*4 + ctl)*4 + j4] with the dh fragments DHF[rt][j4], the GLU'
  // quarters Q0 / Q1 (row tile RTQ) of chunk CH beside steps 0-1 / 2-3
#define F3B_PHASE_X(X, GLUQ, SLOT, DHF, RTQ, CH, NDH)                                                          \
  if constexpr (!no_mma) {                                                                                     \
    const uint4* wb = reinterpret_cast<const uint4*>(ring + (SLOT) * F3_PHASE) + (wc * 16) * 64 + lane;        \
    uint4 fr[2][4];                                                                                            \
    if constexpr (GLUQ) {                                                                                      \
      F3_MFMA_DRAIN();                                                                                         \
      if constexpr (!no_dma && !no_hl) { f3_wait_vm_for<20>(hp[4 * (RTQ)], hp[4 * (RTQ) + 1], hp[4 * (RTQ) + 2], hp[4 * (RTQ) + 3]); } \
      else if constexpr (no_hl) { }                                                                            \
      else { f3_wait_vm_for<0>(hp[4 * (RTQ)], hp[4 * (RTQ) + 1], hp[4 * (RTQ) + 2], hp[4 * (RTQ) + 3]); }      \
    }                                                                                                          \
    if constexpr (X) {                                                                                         \
      _Pragma("unroll") for (int ct = 0; ct < 4; ++ct) fr[0][ct] = wb[(ct * 4 + 0) * 64];                      \
    }                                                                                                          \
    _Pragma("unroll") for (int j4 = 0; j4 < 4; ++j4) {                                                         \
      if constexpr (X) {                                                                                       \
        if (j4 + 1 < 4) {                                                                                      \
          _Pragma("unroll") for (int ct = 0; ct < 4; ++ct) fr[(j4 + 1) & 1][ct] = wb[(ct * 4 + j4 + 1) * 64];  \
        }                                                                                                      \
      }                                                                                                        \
      __builtin_amdgcn_sched_barrier(0);                                                                       \
      if constexpr (X) {                                                                                       \
        _Pragma("unroll") for (int ct = 0; ct < 4; ++ct) {                                                     \
          mma32(xacc[0][ct], fr[j4 & 1][ct], DHF[0][j4]);                                                      \
          mma32(xacc[1][ct], fr[j4 & 1][ct], DHF[1][j4]);                                                      \
        }                                                                                                      \
      }                                                                                                        \
      if constexpr (GLUQ) {                                                                                    \
        if (j4 == 0) F3B_GLU(RTQ, 0, NDH)                                                                      \
        if (j4 == 2) F3B_GLU(RTQ, 1, NDH)                                                                      \
        if (j4 == 3) F3B_STORE_LOAD(RTQ, CH, NDH)                                                              \
      }                                                                                                        \
      __builtin_amdgcn_sched_barrier(0);                                                                       \
      if constexpr (GLUQ) { F3B_ISSUE2(j4) }                                                                   \
      __builtin_amdgcn_sched_barrier(0);                                                                       \
    }                                                                                                          \
  } else if constexpr (GLUQ) {                                                                                 \
    _Pragma("unroll") for (int j4 = 0; j4 < 4; ++j4) { F3B_ISSUE2(j4) }                                        \
  }
#define F3B_READ_PARTNER()                                                                                     \
  _Pragma("unroll") for (int rt = 0; rt < 2; ++rt)                                                             \
    _Pragma("unroll") for (int k = 0; k < 4; ++k) dhp[rt][k] = partner_h[(rt * 4 + k) * 64];

  // Counted waits (ffn3_fwd_kernel: loads retire in issue order, stores do not count on).  A phase end keeps 16: the DMAs of
  // the two phases after the next phase's group -- also right if the tile loads (4 per X phase, HBM) retired late or early.
  // The tile loads themselves are waited for at the start of the X phase that consumes them with the 20 loads issued after
  // them kept (12 + 8: an X phase -- 8 DMAs and 4 tile loads --, a D phase)
  int slot = 0;
  uint4 dhn[2][4];                                               // the fragments the GLU' of this iteration produces
  // ---- chunk 0: D, then GLU' only
  F3B_STAMP()
  F3B_PHASE_D(slot, false, 0)
  F3B_PHASE_END(16)
  F3B_STAMP()
  F3B_PHASE_X(false, true, slot, dho, 0, 0, dhn)
  F3B_PHASE_END(16)
  F3B_STAMP()
  F3B_PHASE_X(false, true, slot, dhp, 1, 0, dhn)
  F3B_PHASE_END(16)
  F3B_STAMP()
#pragma unroll
  for (int rt = 0; rt < 2; ++rt)
#pragma unroll
    for (int k = 0; k < 4; ++k) dho[rt][k] = dhn[rt][k];
  // ---- chunks 1 .. NC-1
  for (int C = 1; C < NC; ++C) {
    F3B_READ_PARTNER()                                           // dh of chunk C-1 (written before the barriers of its X phases)
    F3B_PHASE_D(slot, true, C - 1)
    F3B_PHASE_END(16)
    F3B_STAMP()
    F3B_PHASE_X(true, true, slot, dho, 0, C, dhn)
    F3B_PHASE_END(16)
    F3B_STAMP()
    F3B_PHASE_X(true, true, slot, dhp, 1, C, dhn)
    F3B_PHASE_END(16)
    F3B_STAMP()
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int k = 0; k < 4; ++k) dho[rt][k] = dhn[rt][k];
  }
  // ---- closing phases: XA, XB of the last chunk (no DMA, no global traffic: only the last X phase's 16 may still fly); the
  // partners' ids travel meanwhile
  if constexpr (!SLAB) f3_sync_read_ids(sy);
  F3B_READ_PARTNER()
  if constexpr (!no_st) {
#pragma unroll
    for (int t = 0; t < 8; ++t) dh_store(NC - 1, t);             // the last chunk's dh
  }
  F3B_PHASE_X(true, false, slot, dho, 0, 0, dhn)
  if constexpr (no_dma) f3_wait_vm<0>(); else f3_wait_vm<8>();    // of the last X phase: its 4 tile reloads + 4 of its DMAs at most
  f3_wait_lds();
  f3_barrier();
  slot = (slot + 1) & 3;
  F3B_PHASE_X(true, false, slot, dhp, 1, 0, dhn)
#undef F3B_ISSUE2
#undef F3B_PHASE_D
#undef F3B_PHASE_END
#undef F3B_GLU
#undef F3B_STORE_LOAD
#undef F3B_PHASE_X
#undef F3B_READ_PARTNER
  f3_wait_vm_for<0>(hp[0], hp[1], hp[2], hp[3]);                 // the last (unused) tile reloads land in registers that stay
  f3_wait_vm_for<0>(hp[4], hp[5], hp[6], hp[7]);                 // theirs until here; everything else has drained too
  f3_wait_lds();
  f3_barrier();
  F3B_STAMP()
  if constexpr (SLAB) {
    f3_store_slab(xacc, ring, p.slab + (int64_t)sl * p.M * 256, rb, p.M, wid, wr, wc, lane);    // see ffn3_fwd_kernel
    F3B_STAMP()
    return;
  }

  // ---- exchange the four partial input gradients of the row block; this workgroup finishes quarter `sl`: dx = skip + sum
  float* own = reinterpret_cast<float*>(ring);
  auto rs = __builtin_amdgcn_make_buffer_rsrc(p.scratch + (int64_t)rb * (4 * 4 * 8192), 0, 4 * 4 * 32768, 0x00020000);
  f3_send_partials(xacc, rs, own, sy, sl, wr, wc, lane);
  const int64_t row = (int64_t)rb * 128 + 32 * sl + m;
  const bool live = row < p.M;
  const int64_t crow = live ? row : (int64_t)p.M - 1;
  float4 sk[2][4];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q)
      sk[t][q] = p.skip ? *reinterpret_cast<const float4*>(p.skip + crow * D + 32 * (2 * wid + t) + 8 * q + 4 * hi)
                        : make_float4(0.f, 0.f, 0.f, 0.f);
  f3_wait_vm<0>();
  f3_wait_lds();
  f3_barrier();
  if (tid == 0) f3_arrive_wait(sy, p.spin_limit, p.fault);
  f3_barrier();
  otr_u32x4 part[3][2][4];
  f3_recv_partials(part, rs, sy, sl, wid, lane);
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 o = *reinterpret_cast<const float4*>(own + (((2 * wid + t) * 4 + q) * 64 + lane) * 4);
      float4 v = make_float4(sk[t][q].x + o.x, sk[t][q].y + o.y, sk[t][q].z + o.z, sk[t][q].w + o.w);
#pragma unroll
      for (int n = 0; n < 3; ++n) {
        v.x += __uint_as_float(part[n][t][q].x); v.y += __uint_as_float(part[n][t][q].y);
        v.z += __uint_as_float(part[n][t][q].z); v.w += __uint_as_float(part[n][t][q].w);
      }
      if (live) *reinterpret_cast<float4*>(p.dx + row * D + 32 * (2 * wid + t) + 8 * q + 4 * hi) = v;
    }
}

}  // namespace

extern int g_otr_ffn2_ablate;

// hidden slices x row blocks, padded to whole XCD groups (f3_block_map)
extern int g_otr_ffn_map;   // api.hip (otr_debug_set(15, v))
static inline unsigned f3_grid(int64_t M, int S) {
  const int64_t blocks = (M + 127) / 128;
  return g_otr_ffn_map ? (unsigned)(8 * ((blocks + 1) / 2)) : (unsigned)(8 * S * ((blocks + 7) / 8));
}

extern int g_otr_spin_limit;
extern int32_t* g_otr_fault;
extern int g_otr_ffn_coh_only;
extern unsigned long long* g_otr_trace;

#define F3_LAUNCH_FWD_SLAB(SAVE)                                                                                   \
  switch (g_otr_ffn2_ablate & 31) {                                                                                      \
    case 16: hipLaunchKernelGGL((ffn3_fwd_kernel<256, 16, SAVE, true>), dim3(f3_grid(M, S)), dim3(256), 0, stream, p); break; \
    default: hipLaunchKernelGGL((ffn3_fwd_kernel<256, 0, SAVE, true>), dim3(f3_grid(M, S)), dim3(256), 0, stream, p); break;  \
  }
#define F3_LAUNCH_FWD(SAVE)                                                                                        \
  switch (g_otr_ffn2_ablate & 31) {                                                                                      \
    case 16: hipLaunchKernelGGL((ffn3_fwd_kernel<256, 16, SAVE>), dim3(f3_grid(M, S)), dim3(256), 0, stream, p); break; \
    case 0: hipLaunchKernelGGL((ffn3_fwd_kernel<256, 0, SAVE>), dim3(f3_grid(M, S)), dim3(256), 0, stream, p); break;   \
    case 1: hipLaunchKernelGGL((ffn3_fwd_kernel<256, 1, SAVE>), dim3(f3_grid(M, S)), dim3(256), 0, stream, p); break;   \
    case 2: hipLaunchKernelGGL((ffn3_fwd_kernel<256, 2, SAVE>), dim3(f3_grid(M, S)), dim3(256), 0, stream, p); break;   \
    case 4: hipLaunchKernelGGL((ffn3_fwd_kernel<256, 4, SAVE>), dim3(f3_grid(M, S)), dim3(256), 0, stream, p); break;   \
    default: hipLaunchKernelGGL((ffn3_fwd_kernel<256, 3, SAVE>), dim3(f3_grid(M, S)), dim3(256), 0, stream, p); break;  \
  }

// scratch bytes / sync ints of the fused form for M rows (S = 4); bytes of the saved (value, sigmoid) tiles; padded rows of u / dh
int64_t ffn3_scratch_bytes(int64_t M) { return ((M + 127) / 128) * (int64_t)(4 * 4 * 32768); }
int64_t ffn3_hsave_bytes(int64_t M, int32_t F) { return ((M + 127) / 128) * 128 * (int64_t)F * 4; }
int64_t ffn3_padded_rows(int64_t M) { return ((M + 127) / 128) * 128; }
int64_t ffn3_sync_ints(int64_t M) { return 8 * ((M + 127) / 128); }

int32_t ffn3_ln_fwd_launch(const float* x, const void* x16, const void* w1_pack, const float* b1, const void* w2_pack, const float* b2,
                           const float* gamma, const float* beta, const uint64_t* seed, float p_drop, uint64_t rng_offset, float eps,
                           float* y, void* y16, float* z, float* mean, float* rstd, void* hsave, void* usave, float* scratch,
                           int32_t* sync, int64_t M, int32_t F, hipStream_t stream) {
  constexpr int S = 4;
  Ffn3FwdArgs p{};
  p.x16 = (const uint16_t*)x16; p.p1 = (const uint4*)w1_pack; p.b1 = b1; p.p2 = (const uint4*)w2_pack; p.scratch = scratch;
  p.M = (int)M; p.F = F; p.S = S;
  p.x = x; p.b2 = b2; p.gamma = gamma; p.beta = beta; p.seed = seed; p.y = y; p.y16 = (uint16_t*)y16; p.z = z; p.mean = mean; p.rstd = rstd;
  p.sync = sync; p.fault = g_otr_fault; p.spin_limit = g_otr_spin_limit; p.coh_only = g_otr_ffn_coh_only; p.map = g_otr_ffn_map; p.eps = eps; p.p_drop = p_drop; p.rng_offset = rng_offset;
  p.hsave = (uint4*)hsave; p.usave = (uint16_t*)usave; p.trace = g_otr_trace;
  if (hsave) { F3_LAUNCH_FWD(true) } else { F3_LAUNCH_FWD(false) }
  return otr_check_launch("ffn3_ln_fwd");
}

// slab form: the slices' partial sums leave as 16-bit slabs [4][M][256]; the consumer finishes the LayerNorm
int32_t ffn3_fwd_slab_launch(const void* x16, const void* w1_pack, const float* b1, const void* w2_pack, void* hsave, void* usave, void* slab,
                             int64_t M, int32_t F, hipStream_t stream) {
  constexpr int S = 4;
  Ffn3FwdArgs p{};
  p.x16 = (const uint16_t*)x16; p.p1 = (const uint4*)w1_pack; p.b1 = b1; p.p2 = (const uint4*)w2_pack;
  p.M = (int)M; p.F = F; p.S = S; p.map = g_otr_ffn_map;
  p.hsave = (uint4*)hsave; p.usave = (uint16_t*)usave; p.slab = (uint16_t*)slab; p.trace = g_otr_trace;
  if (hsave) { F3_LAUNCH_FWD_SLAB(true) } else { F3_LAUNCH_FWD_SLAB(false) }
  return otr_check_launch("ffn3_fwd_slab");
}

int32_t ffn3_bwd_launch(const void* dy16, const void* hsave, const void* w2t_pack, const void* w1t_pack, void* dh, const float* skip,
                        float* dx, float* scratch, int32_t* sync, int64_t M, int32_t F, hipStream_t stream) {
  Ffn3BwdArgs p{};
  p.dy16 = (const uint16_t*)dy16; p.hsave = (const uint4*)hsave; p.p3 = (const uint4*)w2t_pack; p.p4 = (const uint4*)w1t_pack;
  p.dh = (uint16_t*)dh; p.skip = skip; p.dx = dx; p.scratch = scratch; p.sync = sync; p.fault = g_otr_fault;
  p.spin_limit = g_otr_spin_limit; p.coh_only = g_otr_ffn_coh_only; p.map = g_otr_ffn_map; p.M = (int)M; p.F = F;
  switch (g_otr_ffn2_ablate & 15) {
    case 0: hipLaunchKernelGGL((ffn3_bwd_kernel<256, 0>), dim3(f3_grid(M, 4)), dim3(256), 0, stream, p); break;
    case 1: hipLaunchKernelGGL((ffn3_bwd_kernel<256, 1>), dim3(f3_grid(M, 4)), dim3(256), 0, stream, p); break;
    case 2: hipLaunchKernelGGL((ffn3_bwd_kernel<256, 2>), dim3(f3_grid(M, 4)), dim3(256), 0, stream, p); break;
    case 4: hipLaunchKernelGGL((ffn3_bwd_kernel<256, 4>), dim3(f3_grid(M, 4)), dim3(256), 0, stream, p); break;
    case 8: hipLaunchKernelGGL((ffn3_bwd_kernel<256, 8>), dim3(f3_grid(M, 4)), dim3(256), 0, stream, p); break;
    case 12: hipLaunchKernelGGL((ffn3_bwd_kernel<256, 12>), dim3(f3_grid(M, 4)), dim3(256), 0, stream, p); break;
    default: hipLaunchKernelGGL((ffn3_bwd_kernel<256, 3>), dim3(f3_grid(M, 4)), dim3(256), 0, stream, p); break;
  }
  return otr_check_launch("ffn3_bwd");
}

int32_t ffn3_bwd_slab_launch(const void* dy16, const void* hsave, const void* w2t_pack, const void* w1t_pack, void* dh, void* slab, int64_t M,
                             int32_t F, hipStream_t stream) {
  Ffn3BwdArgs p{};
  p.dy16 = (const uint16_t*)dy16; p.hsave = (const uint4*)hsave; p.p3 = (const uint4*)w2t_pack; p.p4 = (const uint4*)w1t_pack;
  p.dh = (uint16_t*)dh; p.slab = (uint16_t*)slab; p.map = g_otr_ffn_map; p.M = (int)M; p.F = F;
  p.trace = g_otr_trace;
  if ((g_otr_ffn2_ablate & 31) == 16) hipLaunchKernelGGL((ffn3_bwd_kernel<256, 16, true>), dim3(f3_grid(M, 4)), dim3(256), 0, stream, p);
  else hipLaunchKernelGGL((ffn3_bwd_kernel<256, 0, true>), dim3(f3_grid(M, 4)), dim3(256), 0, stream, p);
  return otr_check_launch("ffn3_bwd_slab");
}

// 1 when the 128-row kernels take this (hidden size, split): whole 64-unit chunks per slice, biases fit their LDS staging
int32_t ffn3_takes(int32_t F, int32_t S) { return S >= 1 && (F / 32) % S == 0 && ((F / 32) / S) % 2 == 0 && (F / 32) / S <= 32; }

// host-side view of the launch geometry, for tests: grid size of the split kernels for M rows under mapping `map`, and
// out[2 b], out[2 b + 1] = (row block, slice) of workgroup b (row blocks >= ceil(M / 128) are padding that exits at once)
int32_t ffn3_debug_block_map(int64_t M, int32_t map, int32_t* out, int32_t cap) {
  const int64_t blocks = (M + 127) / 128;
  const int32_t grid = map ? (int32_t)(8 * ((blocks + 1) / 2)) : (int32_t)(8 * 4 * ((blocks + 7) / 8));
  if (!out) return grid;
  for (int b = 0; b < grid && b < cap; ++b) {
    int rb, sl;
    f3_block_map(b, map, rb, sl);
    out[2 * b] = rb; out[2 * b + 1] = sl;
  }
  return grid;
}
