// Input gradient of the second Conv2dLayer of a WIDE frontend (256 -> 256 channels: conformer_baseline.yaml; frontend/conv.py:50-83,
// 141-142) as an implicit GEMM whose WEIGHTS stream through LDS.  One launch section per parity class (t1 & 1, f1 & 1) of the act1 pixels
// with its 2 / 4 / 1 / 2 taps (conv.hip has the derivation):
//
//   dact1[b, 2i + pt, 2j + pf, n] = [act1 > 0] * sum over the class's taps, sum over c < 256 of  g2[row(pixel, tap), c] * W_tap[c, n],   n < 256
//
// Why: conv.hip's weight-stationary form keeps a class's A fragments resident in LDS, which at 256 x 256 channels only fits for a
// 64-channel slice of the output -- four copies of the grid then each read the g2 rows again, from HBM (77 MB x 4 classes x 4 slices;
// 485 us, HBM-bound at 15 % MFMA use).  Here a workgroup (8 waves x 32 pixels) owns ALL 256 output channels of its 256 pixels: the pixel
// rows of a tap are read ONCE, straight into the MFMA B operand (16 x 16 bytes per lane), and the tap's weights -- pre-packed in MFMA
// A-fragment order by a small kernel -- arrive as eight 16 KB chunks (k half x 64 output channels) in an LDS ring of eight, by DMA six
// chunks ahead; one barrier per chunk (16 MFMAs 32x32x16 per wave).  A k half's row pieces are spent after its fourth chunk and
// reloaded there with the next tap's, five chunks before they are used.  Accumulators: 8 x 16 registers per lane.  One workgroup per CU.
// Measured (batch 32 x 1000 frames): 350 us against 485; the launch moves ~0.95 GB (g2 once per class, the act1 mask, dact1), ~200 us at
// the HBM rate -- the rest is the 256-register budget of 8 waves per CU (accumulators 128 + row pieces 64 + fragments: the allocator
// spills ~70 registers around the tap boundaries) and one barrier per 16 MFMAs.  otr_debug_set(32, v) ablates its parts (CwArgs.ablate):
// no MFMAs 328, no fragment reads 342, no weight DMA 326, no row reloads 273, all four 218 us.
// (The FORWARD of the same layer was built on this skeleton as well -- 9 taps, bias + ReLU epilogue -- and measured 330-430 us against
//  the 320 us of gemm_kernel's im2col loader: not kept.)
#include "common.h"

namespace {

struct CwArgs {
  const uint16_t* in;        // g2 [B,T2,F2,256]
  const uint4* wp;           // packed weights: [9 taps][2 k halves][4 slices][2 rt][8 ks][64 lanes] x 16 bytes
  const uint16_t* mask;      // act1 (addressed like out)
  uint16_t* out;             // dact1 [B,T1,F1,256]
  int B, T1, F1, T2, F2;
  int ablate;                // tuning hook (otr_debug_set(32, v)): 1 = no MFMAs, 2 = one fragment read per chunk, 4 = no weight DMA, 8 = no row reloads
  int wg0[5];                // parity class c = 2 pt + pf owns workgroups [wg0[c], wg0[c+1])
};

typedef __attribute__((address_space(3))) unsigned char cw_lds_byte;
// one wave instruction: 64 lanes x 16 B, per-lane global source -> LDS [dst, dst + 1024) lane-linear (see wgrad256.hip dma16)
// (scalar base + 32-bit lane offset: per-lane 64-bit addresses of every chunk are loop invariants that hipcc hoists out of the tile loop --
//  288 registers for the forward's 36 chunks)
__device__ __forceinline__ void cw_dma16(const void* sbase, uint32_t voff, uint32_t lds_dst) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}

constexpr int CW_C = 256, CW_KS = 16, CW_CHUNK = 1024, CW_RING = 3;   // CW_CHUNK: uint4 entries per (tap, k half, slice) chunk = 16 KB
constexpr int CW_NBUF = 8, CW_AHEAD = 6;                              // LDS ring of chunks; how many chunks ahead the DMA runs      // channels (both sides); k-steps per tap; uint4 entries per (tap, slice) chunk

// ---- weights -> A-fragment order.  Entry (tap, k half, slice, rt, ks & 7, lane): row n = 64 slice + 32 rt + (lane & 31) of A, k = 16 ks + 8 (lane >> 5) .. + 7
// A[n = c1][k = c2] = w2r[c2][tap][c1]
__global__ __launch_bounds__(256) void cw_pack_kernel(const uint16_t* __restrict__ w2r, uint4* __restrict__ wp) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= 9 * 8 * CW_CHUNK) return;
  const int ln = e & 63, ks = ((e >> 12) & 1) * 8 + ((e >> 6) & 7), rt = (e >> 9) & 1, slice = (e >> 10) & 3, tap = e >> 13;
  const int n = slice * 64 + rt * 32 + (ln & 31), k0 = ks * 16 + (ln >> 5) * 8;
  uint32_t v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = w2r[((int64_t)(k0 + j) * 9 + tap) * CW_C + n];
  wp[e] = make_uint4(v[0] | (v[1] << 16), v[2] | (v[3] << 16), v[4] | (v[5] << 16), v[6] | (v[7] << 16));
}

struct CwPix { int b, i, j; };

// the parity class (PT, PF)
template <int PT, int PF>
__device__ __forceinline__ void cw_body(const CwArgs& p, uint4* wbuf, uint4* ebuf_all, const int w, const int nwg) {
  constexpr int NKW = PF ? 2 : 1;
  constexpr int NT = (PT ? 1 : 2) * NKW;
  constexpr int NCH = NT * 8;                                  // chunks per tile (a multiple of the ring: the LDS buffer of a chunk is its index & 7)
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, hi = lane >> 5, pl = lane & 31;
  const int nT = PT ? p.T1 / 2 : (p.T1 + 1) / 2, nF = PF ? p.F1 / 2 : (p.F1 + 1) / 2;
  const int Mc = p.B * nT * nF, ntile = (Mc + 255) / 256;
  if (w >= ntile) return;                                       // (workgroup-uniform)

  auto pix_of = [&](int m) {
    const uint32_t mc = (uint32_t)min(m, Mc - 1), bi = mc / (uint32_t)nF, b = bi / (uint32_t)nT;
    return CwPix{(int)b, (int)(bi - b * (uint32_t)nT), (int)(mc - bi * (uint32_t)nF)};
  };
  // lane (pixel, hi)'s 16-byte pieces of the row tap tt reads: a 32-bit element offset into p.in (one base per pixel + a uniform delta per
  // tap, clamped into the tensor: nothing is kept per tap); false: the tap falls outside the operand for this pixel (operand = 0)
  const int TR = p.T2, FR = p.F2;
  const int in_last = p.B * TR * FR * CW_C - CW_C;            // (< 2^31: checked by the host)
  auto base_of = [&](const CwPix& px) { return ((px.b * TR + px.i) * FR + px.j) * CW_C + hi * 8; };     // the tap-independent part
  auto src_of = [&](const CwPix& px, int base, int tt, int& off) {
    const int a = tt / NKW, b2 = tt - a * NKW;
    const int tr = PT ? px.i : px.i - a, fr = PF ? px.j + 1 - b2 : px.j, delta = ((PT ? 0 : -a) * FR + (PF ? 1 - b2 : 0)) * CW_C;
    off = min(max(base + delta, hi * 8), in_last + hi * 8);
    return tr >= 0 && tr < TR && fr >= 0 && fr < FR;
  };
  // weights of tap tt (compile-time position), 64-channel slice s: where its chunk starts in the packed image
  // (wp_t = p.wp behind an opaque zero renewed per tile: the chunks' base addresses are loop invariants, and hipcc keeps all 72 of the
  //  forward's in SGPR pairs across the tile loop -- 98 of them spilled -- unless they depend on something the loop changes)
  const uint4* wp_t = p.wp;
  auto chunk_of = [&](int c) {                                   // chunk c of a tile: tap c >> 3, k half (c >> 2) & 1, slice c & 3
    const int tt = c >> 3, js = c & 7;
    const int a = tt / NKW, b2 = tt - a * NKW, tap = (PT ? 1 : 2 * a) * 3 + (PF ? 2 * b2 : 1);
    return wp_t + (tap * 8 + js) * CW_CHUNK;                     // (uniform: a scalar base)
  };

  f32x16 acc[8];
#pragma unroll
  for (int a = 0; a < 8; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;

  // chunk c -> LDS buffer c & 7 by direct-to-LDS DMA, CW_AHEAD chunks ahead of its use: wave w moves entries [512 q + 64 w, + 64) of the
  // chunk, q < 2 (the packed image is already in LDS order).  The compiler does not see these writes: the wait that publishes a chunk is
  // written by hand.  vmcnt retires in order; behind the two DMA instructions of chunk c this wave has issued, by the time it needs the
  // chunk, those of the 5 chunks after it (10) and at least one group of 8 row pieces (every 4th chunk issues one): vmcnt(18) never lets
  // chunk c itself stay in flight -- except while the pipeline fills (the first tile's first chunks: drained with vmcnt(0)).  The waits
  // hipcc adds for the loads it knows (row pieces, masks, bias) count too few instructions and therefore only ever wait longer.
  const uint32_t wdst = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(cw_lds_byte*)reinterpret_cast<unsigned char*>(wbuf) + (uint32_t)(wid * 1024));
  auto dma_chunk = [&](int c) {
    const uint4* src = chunk_of(c % NCH);
#pragma unroll
    for (int q = 0; q < 2; ++q) cw_dma16(src, (uint32_t)((tid + 512 * q) * 16), wdst + (uint32_t)(((c & (CW_NBUF - 1)) * CW_CHUNK + 512 * q) * 16));
  };
  uint4 bq[CW_KS];                                              // B operand: this lane's pieces of its pixel's row of the current tap
  CwPix px = pix_of(w * 256 + wid * 32 + pl);
  int pbase = base_of(px), off0;
  bool okc = src_of(px, pbase, 0, off0);
#pragma unroll
  for (int ks = 0; ks < CW_KS; ++ks) bq[ks] = ld_global_b128(p.in + off0 + ks * 16);
#pragma unroll
  for (int c = 0; c < CW_AHEAD; ++c) dma_chunk(c);
  bool filling = true;

  uint4* ebuf = ebuf_all + wid * 64;
  unsigned char* ebytes = reinterpret_cast<unsigned char*>(ebuf);
  const int erow = lane >> 1, ehalf = lane & 1;                 // the pixel row / 8-channel half this lane stores (see conv.hip's epilogue)

  for (int tile = w; tile < ntile; tile += nwg) {
    {
      int z = 0;
      asm volatile("" : "+s"(z));
      wp_t = p.wp + z;
    }
    const int me = tile * 256 + wid * 32 + erow;
    const bool elive = me < Mc;
    const CwPix e = pix_of(me);                                 // (32-bit element offsets: the tensors hold < 2^31 elements, checked by the host)
    const int obase = (((e.b * p.T1 + 2 * e.i + PT) * p.F1) + 2 * e.j + PF) * CW_C + 8 * ehalf;
    int off_next_tile = 0;
#pragma unroll
    for (int tt = 0; tt < NT; ++tt) {
      int offn;
      bool okn;
      if (tt + 1 < NT) {
        okn = src_of(px, pbase, tt + 1, offn);
      } else {                                                  // the next tile's pixel (the last tile: this one again, loaded and never used)
        if (tile + nwg < ntile) { px = pix_of((tile + nwg) * 256 + wid * 32 + pl); pbase = base_of(px); }
        okn = src_of(px, pbase, 0, offn);
        off_next_tile = offn;
      }
      const uint16_t* srcn = p.in + offn;
#pragma unroll
      for (int js = 0; js < 8; ++js) {
        const int c = tt * 8 + js, kh = js >> 2, s = js & 3;
        if (filling && c < CW_AHEAD) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
        __syncthreads();                                        // chunk c is in LDS for everybody; the buffer of chunk c - 2 is free
        if (!(p.ablate & 4)) dma_chunk(c + CW_AHEAD);
        __builtin_amdgcn_sched_barrier(0);
        const uint4* af = wbuf + (c & (CW_NBUF - 1)) * CW_CHUNK + lane;
        // A fragments: a ring of CW_RING k-steps in registers, read CW_RING - 1 steps ahead of the MFMAs that use them (every index is a
        // compile-time constant: no moves)
        uint4 fr[CW_RING][2];
#pragma unroll
        for (int k = 0; k < CW_RING - 1; ++k) { fr[k][0] = af[k * 64]; fr[k][1] = af[(8 + k) * 64]; }
#pragma unroll
        for (int k8 = 0; k8 < 8; ++k8) {
          const int ks = kh * 8 + k8;
          if (s == 0 && !okc) bq[ks] = make_uint4(0u, 0u, 0u, 0u);
          if (k8 + CW_RING - 1 < 8 && !(p.ablate & 2)) {
            fr[(k8 + CW_RING - 1) % CW_RING][0] = af[(k8 + CW_RING - 1) * 64];
            fr[(k8 + CW_RING - 1) % CW_RING][1] = af[(8 + k8 + CW_RING - 1) * 64];
          }
          if (!(p.ablate & 1)) {
            mma32(acc[2 * s], fr[k8 % CW_RING][0], bq[ks]);
            mma32(acc[2 * s + 1], fr[k8 % CW_RING][1], bq[ks]);
          } else {
            asm volatile("" ::"v"(fr[k8 % CW_RING][0].x), "v"(fr[k8 % CW_RING][1].w), "v"(bq[ks].x));
          }
          if (s == 3 && !(p.ablate & 8)) {                                        // the registers just consumed take the next tap's pieces: five chunks until they are used
            if (tt == NT - 1) bq[ks] = ld_global_b128(p.mask + obase + ks * 16);      // ... or the ReLU-mask rows of this tile's pixels
            else bq[ks] = ld_global_b128(srcn + ks * 16);
          }
          __builtin_amdgcn_sched_barrier(0);                   // (without it hipcc issues a chunk's fragment reads at once)
        }
      }
      okc = okn;
    }
    filling = false;
    // ---- epilogue: 16 slabs of 16 channels through 1 KB of LDS per wave (8-byte pieces per lane -> 16 bytes per lane, two lanes per row)
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      const int a = g >> 1, q0 = 2 * (g & 1);
#pragma unroll
      for (int qq = 0; qq < 2; ++qq) {
        const int q = q0 + qq;
        const float v0 = acc[a][4 * q], v1 = acc[a][4 * q + 1], v2 = acc[a][4 * q + 2], v3 = acc[a][4 * q + 3];
        otr_u32x2 wv = {pack2h(v0, v1), pack2h(v2, v3)};
        *reinterpret_cast<otr_u32x2*>(ebytes + pl * 32 + qq * 16 + hi * 8) = wv;
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[a][4 * q + r] = 0.f;
      }
      asm volatile("" ::: "memory");                           // the pieces above are read back by other lanes of this wave
      const uint4 v = ebuf[lane];                              // (LDS operations of a wave complete in order: no barrier)
      asm volatile("" ::: "memory");
      const uint4 m = bq[g];
      auto keep = [](uint32_t act, uint32_t val) {
        return ((int16_t)(act & 0xffffu) > 0 ? (val & 0xffffu) : 0u) | ((int16_t)(act >> 16) > 0 ? (val & 0xffff0000u) : 0u);
      };
      if (elive) st_global_b128(p.out + obase + g * 16, make_uint4(keep(m.x, v.x), keep(m.y, v.y), keep(m.z, v.z), keep(m.w, v.w)));
      bq[g] = ld_global_b128(p.in + off_next_tile + g * 16);    // the mask piece is spent: the next tile's first tap moves in
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // the last DMA (a chunk nobody uses) must not outlive the workgroup's LDS
}

__global__ __launch_bounds__(512, 2) void conv2wide_dgrad_kernel(CwArgs p) {
  __shared__ uint4 wbuf[CW_NBUF * CW_CHUNK];                    // 128 KB
  __shared__ uint4 ebuf[8 * 64];                                // 1 KB per wave
  int cls = 0;
  while (cls < 3 && (int)blockIdx.x >= p.wg0[cls + 1]) ++cls;
  const int w = (int)blockIdx.x - p.wg0[cls], nwg = p.wg0[cls + 1] - p.wg0[cls];
  if (cls == 0) cw_body<0, 0>(p, wbuf, ebuf, w, nwg);
  else if (cls == 1) cw_body<0, 1>(p, wbuf, ebuf, w, nwg);
  else if (cls == 2) cw_body<1, 0>(p, wbuf, ebuf, w, nwg);
  else cw_body<1, 1>(p, wbuf, ebuf, w, nwg);
}

}  // namespace

int g_otr_conv2wide_ablate = 0;
int64_t conv2wide_workspace_bytes() { return (int64_t)9 * 8 * CW_CHUNK * 16; }

// input gradient; wg0 = the parity classes' workgroup ranges (conv.hip conv2_dgrad_plan with one workgroup per CU)
int32_t conv2wide_dgrad(const void* g2, const void* w2r, const void* act1, void* dact1, int B, int T1, int F1, int T2, int F2, const int* wg0,
                        void* scratch, hipStream_t s) {
  CwArgs a{};
  a.in = reinterpret_cast<const uint16_t*>(g2); a.wp = reinterpret_cast<const uint4*>(scratch); a.mask = reinterpret_cast<const uint16_t*>(act1);
  a.ablate = g_otr_conv2wide_ablate; a.out = reinterpret_cast<uint16_t*>(dact1); a.B = B; a.T1 = T1; a.F1 = F1; a.T2 = T2; a.F2 = F2;
  if ((int64_t)B * T1 * F1 * CW_C >= (1ll << 31)) return 1;
  for (int c = 0; c < 5; ++c) a.wg0[c] = wg0[c];
  if (a.wg0[4] <= 0) return 0;
  hipLaunchKernelGGL(cw_pack_kernel, dim3(9 * 8 * CW_CHUNK / 256), dim3(256), 0, s, reinterpret_cast<const uint16_t*>(w2r),
                     reinterpret_cast<uint4*>(scratch));
  hipLaunchKernelGGL(conv2wide_dgrad_kernel, dim3((unsigned)a.wg0[4]), dim3(512), 0, s, a);
  return otr_check_launch("conv2wide_dgrad");
}
