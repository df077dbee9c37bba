// conv2 forward of the Conv2d-subsampling frontend (frontend/conv.py:63-66, second Conv2dLayer: C1 -> C2 channels, 3 x 3, stride 2,
// pad (0, 1), + bias, ReLU) for the shipped channel counts (64 -> 128) and 80- / 40-bin inputs, 16-bit channel-last activations.
//
// The generic implicit-GEMM path (gemm_kernel.h MODE_IM2K: 128 x 128 tiles, the A operand gathered tap by tap) took 60 us for
// 23.5 GFLOP: every workgroup re-reads its input pixels 9/4 times through the L1 and re-streams the 147 KB of weights per tile.
// Here the WEIGHTS are stationary: wave w of a workgroup owns output channels 32 w .. + 31 and keeps their 36 MFMA A fragments
// (9 taps x 4 contraction steps of 16 input channels) in 144 registers for the whole (persistent) kernel.  The input arrives as
// whole rows: a work item is RPI consecutive output rows of one utterance (RPI F2 = 160 output pixels = five 32-pixel tiles), its
// 2 RPI + 1 input rows go to LDS once -- pixel stride 144 B (bank spread for lanes two pixels apart), a zero pixel on either side
// for the frequency padding -- and every B operand is one ds_read_b128 at an immediate offset from the lane's pixel base:
//   D[c2, pixel] += W[c2, tap, c1 16 ks ..] . in[pixel + tap, c1 16 ks ..]           (v_mfma_f32_32x32x16, 36 per tile and wave)
// Bias starts the accumulators, ReLU and the 16-bit conversion run on them, a lane stores 4 consecutive channels of its pixel.
#include "common.h"

namespace {

constexpr int C2F_PS = 144;               // bytes per staged pixel (64 channels x 2 B + 16)
constexpr int C2F_LDS = 108 * 1024;        // 2 RPI + 2 staged rows: the fused form's last work item of an utterance computes one row more
constexpr int C2F_NT = 5;                 // 32-pixel tiles per work item

// The MFMA with the register classes spelled out (as csrc/ffn3.hip): the weight fragment lives in the ACCUMULATOR half of the register
// file (it is an MFMA operand only), which leaves the architectural registers to the accumulators and a ring of B operands read
// ahead from LDS.  (With the builtin, hipcc kept the 144 weight registers in VGPRs, had none left to read ahead, and every MFMA
// waited for its own ds_read: 50 us.)  `s_nop 1`: hipcc pads nothing in front of an asm statement (VALU write -> MFMA read hazard).
#ifdef OTR_HALF_FP16
#define C2F_MFMA_OP "v_mfma_f32_32x32x16_f16"
#else
#define C2F_MFMA_OP "v_mfma_f32_32x32x16_bf16"
#endif
typedef uint32_t c2f_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void c2f_mma(f32x16& acc, const c2f_u32x4& w_acc, const c2f_u32x4& b) {
  asm volatile("s_nop 1\n\t" C2F_MFMA_OP " %0, %1, %2, %0" : "+v"(acc) : "a"(w_acc), "v"(b));
}
__device__ __forceinline__ c2f_u32x4 c2f_lds(const unsigned char* p) {
  const uint4 t = *reinterpret_cast<const uint4*>(p);
  return c2f_u32x4{t.x, t.y, t.z, t.w};
}

// tuning hook (otr_debug_trace): thread 0 stamps the shader clock into trace[16384 + (9 * 256 + workgroup) * 16 + k]
#define C2F_STAMP(K) do { if (p.trace && threadIdx.x == 0) p.trace[16384 + (9 * 256 + (int)blockIdx.x) * 16 + (K)] = __builtin_amdgcn_s_memtime(); } while (0)
struct Conv2FwdArgs {
  unsigned long long* trace;
  const uint16_t* act1; const uint16_t* w2r; const float* b2; uint16_t* act2;
  int B, T1, T2, nblk, nitems;
  // FUSE1: conv1 (frontend/conv.py:63-66, first Conv2dLayer: 1 -> 64 channels) is computed here, tile by tile, straight into the staged
  // image -- the arithmetic of conv1_fwd_mfma_kernel (conv.hip), bit for bit -- and act1 leaves for the backward pass on the way
  const float* x; const float* w1; const float* b1; uint16_t* act1_out;
  int T, F;
};

// FUSE1 = true: the input rows of a work item are not loaded but COMPUTED from the filterbank frames (35 rows x 80 floats instead of
// 17 rows x 40 pixels x 128 B): one launch and one 82 MB read less per step; act1 is still written (streaming stores: only the
// backward pass reads it).
template <int F1, bool FUSE1>              // F1 = 40 (80-bin fbank) or 20
__global__ __launch_bounds__(256, 1) void conv2_fwd_kernel(Conv2FwdArgs p) {
  constexpr int F2 = F1 / 2, RPI = 160 / F2, NIN = 2 * RPI + 1, C1 = 64, C2 = 128;
  constexpr int RS = (F1 + 2) * C2F_PS;                            // bytes per staged input row (pixels -1 .. F1)
  static_assert((NIN + 1) * RS <= C2F_LDS, "the input rows of a work item must fit the LDS");
  constexpr int NCH = NIN * F1 * 8;                                // 16-byte pieces per work item
  constexpr int PER = (NCH + 255) / 256, HALF = (PER + 1) / 2;
  __shared__ __attribute__((aligned(16))) unsigned char smem[C2F_LDS];
  const int tid = threadIdx.x, lane = tid & 63, m = lane & 31, hi = lane >> 5;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  C2F_STAMP(0);

  // the padding pixels, once: nothing ever overwrites them
  for (int i = tid; i < (NIN + 1) * 18; i += 256) {
    const int r = i / 18, rem = i - r * 18, side = rem / 9, c = rem - side * 9;
    *reinterpret_cast<uint4*>(smem + r * RS + (side ? (F1 + 1) * C2F_PS : 0) + c * 16) = make_uint4(0u, 0u, 0u, 0u);
  }
  // this wave's weights: fragment (tap, ks): lane (m, hi) holds W[32 wid + m][tap][16 ks + 8 hi .. + 7]
  c2f_u32x4 wf[36];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const uint4 q = ld_global_b128(p.w2r + ((int64_t)(32 * wid + m) * 9 + t) * C1 + 16 * ks + 8 * hi);
      wf[t * 4 + ks] = c2f_u32x4{q.x, q.y, q.z, q.w};
    }
#pragma unroll
  for (int i = 0; i < 36; ++i) asm volatile("" : "+a"(wf[i]));        // pinned to the accumulator registers
  float4 bias4[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) bias4[q] = *reinterpret_cast<const float4*>(p.b2 + 32 * wid + 8 * q + 4 * hi);
  // the lane's pixel of every tile: the geometry is the same for every work item
  int boff[C2F_NT], orow[C2F_NT];
#pragma unroll
  for (int pt = 0; pt < C2F_NT; ++pt) {
    const int pid = pt * 32 + m, r = pid / F2, f2 = pid - r * F2;
    orow[pt] = r;
    boff[pt] = (2 * r) * RS + (2 * f2) * C2F_PS + hi * 16;        // staged pixel index = f + 1 = 2 f2 + kw
  }

  // FUSE1: the filterbank values of the NEXT item's conv1 tiles are fetched while this item's conv2 tiles run (scalar 4-byte loads,
  // one line per lane: ~5 k cycles of latency when they were waited for in place)
  constexpr int MAXT1 = ((NIN + 1) * F1 + 31) / 32, TPWV = (MAXT1 + 3) / 4;
  float xb[FUSE1 ? TPWV : 1][5];
  auto load_x = [&](int it) {
    if constexpr (FUSE1) {
      const int b_ = it / p.nblk, blk_ = it - b_ * p.nblk, r0_ = 2 * blk_ * RPI;
      const int npx_ = (NIN + (blk_ == p.nblk - 1 ? 1 : 0)) * F1;
#pragma unroll
      for (int k = 0; k < TPWV; ++k) {
        const int rp = min(32 * (wid + 4 * k) + m, npx_ - 1), row = rp / F1, f1 = rp - row * F1;
        const int r = min(r0_ + row, p.T1 - 1);
        const float* xin = p.x + ((int64_t)b_ * p.T + 2 * r) * p.F;
#pragma unroll
        for (int s5 = 0; s5 < 5; ++s5) {
          const int tap = 2 * s5 + hi, kh = tap / 3, kw = tap - 3 * kh, f = 2 * f1 + kw - 1;
          const bool ok = tap < 9 && f >= 0 && f < p.F;
          xb[k][s5] = xin[(ok ? kh : 0) * p.F + (ok ? f : 0)];     // raw: the padding mask is applied where the value is USED (a select here
        }                                                          // would wait for the load on the spot)
      }
    }
  };
  if ((int)blockIdx.x < p.nitems) load_x((int)blockIdx.x);

  for (int item = blockIdx.x; item < p.nitems; item += gridDim.x) {
    const int b = item / p.nblk, blk = item - b * p.nblk;
    const int t2_0 = blk * RPI, r0 = 2 * t2_0;
    const uint16_t* src = p.act1 + (int64_t)b * p.T1 * (F1 * C1);
    __syncthreads();                                               // the previous item's reads are done
    if (item == (int)blockIdx.x) C2F_STAMP(1);
    int nst = 0;
    uint16_t* a1 = nullptr;
    auto store_act1 = [&](int i0, int i1) {
      for (int i = i0; i < i1; ++i) {
        const int idx = tid + 256 * i;
        if (idx >= nst) break;
        const int row = idx / (F1 * 8), rem = idx - row * (F1 * 8), pxi = rem >> 3, c = rem & 7;
        st_global_b128_nt(a1 + (int64_t)idx * 8, *reinterpret_cast<const uint4*>(smem + row * RS + (pxi + 1) * C2F_PS + c * 16));
      }
    };
    if constexpr (FUSE1) {
      // conv1 into the image: tiles of 32 consecutive pixels of the item's rows, wave w takes tiles w, w + 4, ...; the last item of an
      // utterance also computes (and owns) the one or two rows of act1 past its conv2 window
      typedef __attribute__((ext_vector_type(16))) float f32x16_t;
      const bool last = blk == p.nblk - 1;
      const int nr = NIN + (last ? 1 : 0), npx = nr * F1;
      float wa[2][5];
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int s5 = 0; s5 < 5; ++s5) wa[ct][s5] = (2 * s5 + hi < 9) ? p.w1[(32 * ct + m) * 9 + 2 * s5 + hi] : 0.f;
      float4 bq[2][4];
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int q = 0; q < 4; ++q) bq[ct][q] = *reinterpret_cast<const float4*>(p.b1 + 32 * ct + 8 * q + 4 * hi);
#pragma unroll
      for (int k = 0; k < TPWV; ++k) {
        const int rp = 32 * (wid + 4 * k) + m;
        if (32 * (wid + 4 * k) >= npx) break;                      // wave-uniform
        const int rpc = min(rp, npx - 1), row = rpc / F1, f1 = rpc - row * F1;
        const bool live = rp < npx, rowok = r0 + row < p.T1;
        f32x16_t acc[2];
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
#pragma unroll
          for (int q = 0; q < 4; ++q) { acc[ct][4 * q] = bq[ct][q].x; acc[ct][4 * q + 1] = bq[ct][q].y; acc[ct][4 * q + 2] = bq[ct][q].z; acc[ct][4 * q + 3] = bq[ct][q].w; }
#pragma unroll
          for (int s5 = 0; s5 < 5; ++s5) {
            const int tap = 2 * s5 + hi, kw = tap - 3 * (tap / 3), f = 2 * f1 + kw - 1;
            const float xv = (tap < 9 && f >= 0 && f < p.F) ? xb[k][s5] : 0.f;
            acc[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[ct][s5], xv, acc[ct], 0, 0, 0);
          }
        }
        if (live) {
          unsigned char* px = smem + row * RS + (f1 + 1) * C2F_PS;
#pragma unroll
          for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              uint2 v = make_uint2(pack2h(fmaxf(acc[ct][4 * q], 0.f), fmaxf(acc[ct][4 * q + 1], 0.f)),
                                   pack2h(fmaxf(acc[ct][4 * q + 2], 0.f), fmaxf(acc[ct][4 * q + 3], 0.f)));
              if (!rowok) v = make_uint2(0u, 0u);                  // rows past T1 do not exist: conv2 must see zeros there
              *reinterpret_cast<uint2*>(px + (32 * ct + 8 * q + 4 * hi) * 2) = v;
            }
        }
      }
      __syncthreads();
      if (item == (int)blockIdx.x) C2F_STAMP(2);
      if (item + (int)gridDim.x < p.nitems) load_x(item + (int)gridDim.x);
      // act1 for the backward pass: the rows this item OWNS (the first 2 RPI of its window; the last item of an utterance: all that
      // are left), as whole 128-byte pixel rows
      // (issued in five portions BETWEEN the conv2 tiles below: 80 KB of stores in one burst fill the CU's memory queue and the waves
      //  stand at their next store for ~4 k cycles)
      nst = (last ? min(nr, p.T1 - r0) : min(2 * RPI, p.T1 - r0)) * (F1 * 8);
      a1 = p.act1_out + ((int64_t)b * p.T1 + r0) * (F1 * C1);
    } else {
#pragma unroll
    for (int h = 0; h < 2; ++h) {                                  // two batches of loads: 2 x HALF x 4 registers in flight
      uint4 v[HALF];
#pragma unroll
      for (int k = 0; k < HALF; ++k) {
        const int idx = tid + 256 * (h * HALF + k), row = idx / (F1 * 8);
        const bool live = idx < NCH && r0 + row < p.T1;
        v[k] = ld_global_b128(src + (live ? (int64_t)(r0 + row) * (F1 * C1) + (idx - row * (F1 * 8)) * 8 : 0));
      }
#pragma unroll
      for (int k = 0; k < HALF; ++k) {
        const int idx = tid + 256 * (h * HALF + k), row = idx / (F1 * 8), rem = idx - row * (F1 * 8), px = rem >> 3, c = rem & 7;
        if (idx >= NCH) continue;
        uint4 q = v[k];
        if (r0 + row >= p.T1) q = make_uint4(0u, 0u, 0u, 0u);
        *reinterpret_cast<uint4*>(smem + row * RS + (px + 1) * C2F_PS + c * 16) = q;
      }
    }
    __syncthreads();
    if (item == (int)blockIdx.x) C2F_STAMP(2);
    }
    uint16_t* dst = p.act2 + ((int64_t)b * p.T2 + t2_0) * (F2 * C2) + 32 * wid + 4 * hi;
    // the 5 x 36 (tile, tap, step) products as ONE stream with the B operands read PD steps ahead (across tile boundaries)
    constexpr int PD = 8, NSTEP = C2F_NT * 36;
    auto b_of = [&](int g) { const int pt = g / 36, s = g % 36, t = s >> 2, ks = s & 3;
                             return c2f_lds(smem + boff[pt] + (t / 3) * RS + (t % 3) * C2F_PS + ks * 32); };
    c2f_u32x4 ring[PD];
#pragma unroll
    for (int g = 0; g < PD; ++g) ring[g] = b_of(g);
#pragma unroll
    for (int pt = 0; pt < C2F_NT; ++pt) {
      f32x16 acc;
#pragma unroll
      for (int q = 0; q < 4; ++q) { acc[4 * q] = bias4[q].x; acc[4 * q + 1] = bias4[q].y; acc[4 * q + 2] = bias4[q].z; acc[4 * q + 3] = bias4[q].w; }
#pragma unroll
      for (int s = 0; s < 36; ++s) {
        const int g = pt * 36 + s;
        c2f_mma(acc, wf[s], ring[g % PD]);
        if (g + PD < NSTEP) ring[g % PD] = b_of(g + PD);
        __builtin_amdgcn_sched_barrier(0);
      }
      asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");              // MFMA result -> VALU reader (hipcc does not know the asm is an MFMA)
      if constexpr (FUSE1) store_act1(4 * pt, 4 * pt + 4);
      if (t2_0 + orow[pt] < p.T2) {
        uint16_t* o = dst + (int64_t)(pt * 32 + m) * C2;
#pragma unroll
        for (int q = 0; q < 4; ++q)
          *reinterpret_cast<uint2*>(o + 8 * q) = make_uint2(pack2h(fmaxf(acc[4 * q], 0.f), fmaxf(acc[4 * q + 1], 0.f)),
                                                            pack2h(fmaxf(acc[4 * q + 2], 0.f), fmaxf(acc[4 * q + 3], 0.f)));
      }
    }
    if constexpr (FUSE1) store_act1(4 * C2F_NT, 4 * C2F_NT + 8);      // the last item of an utterance owns up to two rows more
    if (item == (int)blockIdx.x) C2F_STAMP(3);
  }
  C2F_STAMP(4);
}

}  // namespace

extern unsigned long long* g_otr_trace;
extern int g_otr_conv2_fwd_direct;         // api.hip (otr_debug_set(22, v)): 0 = the implicit-GEMM path everywhere

// 0 = launched, 1 = not served (the caller takes the implicit-GEMM path)
int32_t conv2_fwd_direct(const void* act1, const void* w2r, const float* b2, void* act2, int B, int T1, int F1, int T2, int F2, int C1, int C2,
                         int act_is_h16, int w_is_h16, hipStream_t stream) {
  if (!g_otr_conv2_fwd_direct || !act_is_h16 || !w_is_h16 || C1 != 64 || C2 != 128 || F2 * 2 != F1 || (F1 != 40 && F1 != 20)) return 1;
  if (((uintptr_t)act1 | (uintptr_t)w2r | (uintptr_t)b2 | (uintptr_t)act2) % 16 != 0) return 1;
  Conv2FwdArgs p{};
  p.trace = g_otr_trace;
  p.act1 = (const uint16_t*)act1; p.w2r = (const uint16_t*)w2r; p.b2 = b2; p.act2 = (uint16_t*)act2;
  p.B = B; p.T1 = T1; p.T2 = T2;
  const int RPI = 160 / F2;
  p.nblk = (T2 + RPI - 1) / RPI; p.nitems = B * p.nblk;
  const unsigned grid = (unsigned)(p.nitems < 256 ? p.nitems : 256);
  if (F1 == 40) hipLaunchKernelGGL((conv2_fwd_kernel<40, false>), dim3(grid), dim3(256), 0, stream, p);
  else hipLaunchKernelGGL((conv2_fwd_kernel<20, false>), dim3(grid), dim3(256), 0, stream, p);
  return otr_check_launch("conv2_fwd");
}

// both Conv2dLayers in one launch (see FUSE1).  0 = launched, 1 = not served (the caller runs otr_conv1_fwd + otr_conv2_fwd)
int32_t conv12_fwd_direct(const float* x, const float* w1, const float* b1, void* act1, const void* w2r, const float* b2, void* act2, int B, int T,
                          int F, int T1, int F1, int T2, int F2, int C1, int C2, int act_is_h16, int w_is_h16, hipStream_t stream) {
  if (!g_otr_conv2_fwd_direct || g_otr_conv2_fwd_direct == 2 || !act_is_h16 || !w_is_h16 || C1 != 64 || C2 != 128 || F2 * 2 != F1 || (F1 != 40 && F1 != 20))
    return 1;
  if (!x || !w1 || !b1 || !b2 || (((uintptr_t)act1 | (uintptr_t)w2r | (uintptr_t)b1 | (uintptr_t)b2 | (uintptr_t)act2) % 16 != 0)) return 1;
  Conv2FwdArgs p{};
  p.trace = g_otr_trace;
  p.w2r = (const uint16_t*)w2r; p.b2 = b2; p.act2 = (uint16_t*)act2;
  p.x = x; p.w1 = w1; p.b1 = b1; p.act1_out = (uint16_t*)act1; p.T = T; p.F = F;
  p.B = B; p.T1 = T1; p.T2 = T2;
  const int RPI = 160 / F2;
  p.nblk = (T2 + RPI - 1) / RPI; p.nitems = B * p.nblk;
  const unsigned grid = (unsigned)(p.nitems < 256 ? p.nitems : 256);
  if (F1 == 40) hipLaunchKernelGGL((conv2_fwd_kernel<40, true>), dim3(grid), dim3(256), 0, stream, p);
  else hipLaunchKernelGGL((conv2_fwd_kernel<20, true>), dim3(grid), dim3(256), 0, stream, p);
  return otr_check_launch("conv12_fwd");
}
