// Relative-position self-attention backward of the Conformer encoder (module/attention.py:196-253 under autograd; conformer_baseline.yaml:
// d_model 384, 4 heads -> head dim 96): encattn.hip's design -- one (utterance, head), two workgroups of 8 waves, one per ORIENTATION
// (lane = query -> dQ; lane = key -> dK, dV and the score term's gradient), the streamed side staged in LDS once per super-chunk and walked
// by every wave on its own with 32 x 32 x 16 MFMAs, P and dS going from the accumulators straight back into MFMA operands -- for head dim 96
// and WITH the score term  S = (q.k + bias[b, i, h, j - i + T - 1]) * scale  (the Transformer-XL shifted matrix by index arithmetic).
// A separate file on purpose: encattn.hip is the headline's kernel and stays as it is.
//
// What changes against head dim 64:
//   * an image of 256 rows x 96 columns would need 206 KB for the four of the key orientation: the images hold 128 rows (TI), and the streamed
//     side is ALWAYS walked in super-chunks of 128 rows (two for the bench's T' = 249), the own side in blocks of 256 rows;
//   * a row is 12 pieces of 16 bytes: a staging chunk is 32 rows, 16 thread slots per row of which 12 are live (so the delta = rowsum(dO . O)
//     reduction stays a 16-lane butterfly);
//   * three 32-column accumulator tiles per output instead of two, six contraction steps over the head dim instead of four;
//   * the score term: lane = query reads four consecutive columns of its row per register quad (16-byte loads at dword alignment, as
//     attention.hip's bias_load4); lane = key reads one column per register (the lanes of a wave are consecutive columns of one row: coalesced)
//     and writes d bias = scale . dS the same way, EVERY in-range pair, masked ones with 0 (the caller keeps the tensor across steps);
//   * it replaced attn_bwd_dq_kernel<96> + attn_bwd_dkdv_kernel<96> (27 + 55 us per Conformer block at the bench batch).
#include "common.h"

namespace {

constexpr int E9_OWN = 256;               // own rows of a workgroup (8 waves x 32)
constexpr int E9_TI = 128;                // rows of one LDS image = one streamed super-chunk
constexpr int E9_DK = 96, E9_KS = 6, E9_CT = 3, E9_PPR = 12;
constexpr int E9_HS = E9_DK * 2 + 16;     // 208 bytes per row of a row-major [TI][96] 16-bit image
constexpr int E9_TS = E9_TI * 2 + 8;      // 264 bytes per row of a transposed [96][TI] 16-bit image
constexpr int E9_RM = E9_TI * E9_HS;      // 26624
constexpr int E9_TR = E9_DK * E9_TS;      // 25344
constexpr int E9_TMAX = 512;              // most frames served (-lse and delta of every query stay in LDS)
constexpr int E9_SMEM = 2 * E9_RM + 2 * E9_TR + 2 * E9_TMAX * 4;    // 108032
constexpr float E9_LOG2E = 1.4426950408889634f;
static_assert(E9_OWN * E9_HS <= 2 * E9_RM, "the epilogue stages 256 own rows in the two row-major images");

struct E9Args {
  const uint16_t *q, *k, *v, *o, *do_;
  uint16_t *dq, *dk, *dv;
  const uint8_t* key_mask;
  const float* lse;
  const float* bias;        // [.., i, .., col]: element b bias_bs + h bias_hs + i bias_rs + (j - i + T - 1)
  void* dbias;              // same addressing; f32 or the 16-bit type
  int dbias_h16;
  int64_t bias_bs, bias_hs, bias_rs;
  int B, H, T;
  int64_t q_bs, q_ts, k_bs, k_ts, v_bs, v_ts, o_bs, o_ts;
  float scale;
  int ablate;               // tuning hook (otr_debug_set(33, 1 | 2 a)): a & 1 = no score-term loads, a & 2 = no d bias stores, a & 4 = neither orientation computes
};

__device__ __forceinline__ uint4 e9_frag(const unsigned char* img, int row, int hi, int ks) {
  return *reinterpret_cast<const uint4*>(img + row * E9_HS + (2 * ks + hi) * 16);
}
// contraction slots of step k2 (16 streamed rows from `col0`) in accumulator order: rows col0 + 16 k2 + 4 hi + e, then + 8
__device__ __forceinline__ uint4 e9_tfrag(const unsigned char* timg, int row, int col0, int hi, int k2) {
  const unsigned char* vr = timg + row * E9_TS + (col0 + 16 * k2 + 4 * hi) * 2;
  const uint2 lo = *reinterpret_cast<const uint2*>(vr), up = *reinterpret_cast<const uint2*>(vr + 16);
  return make_uint4(lo.x, lo.y, up.x, up.y);
}
__device__ __forceinline__ uint4 e9_pack8(const float* v) {
  return make_uint4(pack2h(v[0], v[1]), pack2h(v[2], v[3]), pack2h(v[4], v[5]), pack2h(v[6], v[7]));
}
__device__ __forceinline__ void e9_zero(f32x16& a) {
#pragma unroll
  for (int r = 0; r < 16; ++r) a[r] = 0.f;
}
// staging chunk c = rows 32 c .. + 31 of the super-chunk: thread t holds piece (row 32 c + t / 16, 16 bytes t % 16) -- slots 12 .. 15 idle
// (they load piece 11 again and drop it).  Rows past the end are clamped at the load and zeroed when they are written.
// -> row-major image.  DOT: w is the same piece of a second matrix and sdot[row] = sum_d a[row][d] b[row][d] is left in LDS.
template <bool DOT>
__device__ __forceinline__ void e9_stage_rm(unsigned char* img, uint4 q, uint4 w, float* sdot, int T, int tid, int c) {
  const int row = 32 * c + (tid >> 4), ch = tid & 15;
  const bool slot = ch < E9_PPR;
  const uint32_t live = (uint32_t)0 - (uint32_t)(row < T && slot);
  q.x &= live; q.y &= live; q.z &= live; q.w &= live;
  if (slot) *reinterpret_cast<uint4*>(img + row * E9_HS + ch * 16) = q;
  if constexpr (DOT) {
    const uint32_t a[4] = {q.x, q.y, q.z, q.w}, b[4] = {w.x, w.y, w.z, w.w};
    float part = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) part += h2f_lo(a[e]) * h2f_lo(b[e]) + h2f_hi(a[e]) * h2f_hi(b[e]);
    part += __shfl_xor(part, 1);
    part += __shfl_xor(part, 2);
    part += __shfl_xor(part, 4);
    part += __shfl_xor(part, 8);
    if (ch == 0) sdot[row] = part;
  }
}
// transposed image of chunk c (32 rows = 16 row pairs) of a staged row-major image: timg[d][row], two rows per 32-bit store; thread t
// (0 .. 255) takes the row pair 16 c + t % 16 and the 16-byte piece t / 16 (< 12), elements 2 e0 .. 2 e0 + 2 ne - 1 of it.
__device__ __forceinline__ void e9_transpose(unsigned char* timg, const unsigned char* img, int c, int t, int e0, int ne) {
  const int rp = 16 * c + (t & 15), ch = t >> 4;
  if (ch >= E9_PPR) return;
  const uint4 a = *reinterpret_cast<const uint4*>(img + (2 * rp) * E9_HS + ch * 16);
  const uint4 b = *reinterpret_cast<const uint4*>(img + (2 * rp + 1) * E9_HS + ch * 16);
  const uint32_t aw[4] = {a.x, a.y, a.z, a.w}, bw[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    if (e < e0 || e >= e0 + ne) continue;
    *reinterpret_cast<uint32_t*>(timg + (8 * ch + 2 * e) * E9_TS + 4 * rp) = (aw[e] & 0xffffu) | (bw[e] << 16);
    *reinterpret_cast<uint32_t*>(timg + (8 * ch + 2 * e + 1) * E9_TS + 4 * rp) = (aw[e] >> 16) | (bw[e] & 0xffff0000u);
  }
}
// own side: the six contraction-step fragments of row `row` (lane (m, hi) holds elements 16 ks + 8 hi .. + 7)
__device__ __forceinline__ void e9_load_frags(uint4 (&f)[E9_KS], const uint16_t* src, int64_t ts, int row, int hi) {
#pragma unroll
  for (int ks = 0; ks < E9_KS; ++ks) f[ks] = ld_global_b128(src + (int64_t)row * ts + 16 * ks + 8 * hi);
}
// accumulator tiles (lane = own row m, registers = head dim 32 ct + 8 q + 4 hi + (r & 3)) x scale -> this wave's 32 rows of a row-major
// staging area -> memory as whole 192-byte rows.  live = false: the lane's row is written as zeros (whatever the accumulators hold)
__device__ __forceinline__ void e9_store_rows(const f32x16 (&acc)[E9_CT], float scale, bool live, unsigned char* og, uint16_t* dst, int64_t ts,
                                              int row0, int T, int lane) {
  const int m = lane & 31, hi = lane >> 5;
#pragma unroll
  for (int ct = 0; ct < E9_CT; ++ct)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      uint2 v = make_uint2(pack2h(acc[ct][4 * q] * scale, acc[ct][4 * q + 1] * scale), pack2h(acc[ct][4 * q + 2] * scale, acc[ct][4 * q + 3] * scale));
      if (!live) v = make_uint2(0u, 0u);
      *reinterpret_cast<uint2*>(og + m * E9_HS + (32 * ct + 8 * q + 4 * hi) * 2) = v;
    }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
  for (int i = 0; i < 32 * E9_PPR / 64; ++i) {
    const int idx = lane + 64 * i, j = idx / E9_PPR, ch = idx - j * E9_PPR;
    if (row0 + j < T) st_global_b128(dst + (int64_t)(row0 + j) * ts + 8 * ch, *reinterpret_cast<const uint4*>(og + j * E9_HS + ch * 16));
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}
typedef float e9_f32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));

__global__ __launch_bounds__(512, 1) void encattn96_bwd_kernel(E9Args p) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[E9_SMEM];
  const int tid = threadIdx.x, lane = tid & 63, m = lane & 31, hi = lane >> 5;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  // the two orientations of one (utterance, head) get workgroup ids equal modulo 8: one XCD (placement observed, used for locality only)
  const int xcd = (int)blockIdx.x & 7, kk = (int)blockIdx.x >> 3;
  const int T = p.T;
  const int nob = (T + E9_OWN - 1) / E9_OWN;                       // own blocks
  const int orient = kk & 1, ob = (kk >> 1) % nob, g = ((kk >> 1) / nob) * 8 + xcd;
  if (g >= p.H * p.B) return;
  const int h = g % p.H, b = g / p.H;
  const int o0 = ob * E9_OWN;                                      // first own row of this workgroup
  const float sc2 = p.scale * E9_LOG2E;
  const uint16_t* Q = p.q + (int64_t)b * p.q_bs + h * E9_DK;
  const uint16_t* K = p.k + (int64_t)b * p.k_bs + h * E9_DK;
  const uint16_t* V = p.v + (int64_t)b * p.v_bs + h * E9_DK;
  const uint16_t* O = p.o + (int64_t)b * p.o_bs + h * E9_DK;
  const uint16_t* dO = p.do_ + (int64_t)b * p.o_bs + h * E9_DK;
  const float* lse = p.lse + ((int64_t)b * p.H + h) * T;
  const uint8_t* km = p.key_mask ? p.key_mask + (int64_t)b * T : nullptr;
  const int64_t bb = (int64_t)b * p.bias_bs + (int64_t)h * p.bias_hs;
  const int own = o0 + 32 * wid + m;                             // this lane's own row (query or key)
  const int ownc = min(own, T - 1);
  const bool wave_live = o0 + 32 * wid < T;                      // this wave owns at least one real row

  const int nchunk = (T + 31) >> 5;                                // 32-row chunks of the streamed side, super-chunks of four
  // piece (row 32 cc + tid / 16, 16 bytes tid % 16) of streamed chunk cc (rows past T clamped; zeroed when written)
  auto issue = [&](const uint16_t* src, int64_t ts, int cc) {
    const int row = min(32 * cc + (tid >> 4), T - 1), ch = min(tid & 15, E9_PPR - 1);
    return ld_global_b128(src + (int64_t)row * ts + 8 * ch);
  };

  if (orient == 0) {
    // ------------------------------------------------------------------ lane = query: dQ = scale . dS K
    unsigned char* krm = smem;
    unsigned char* vrm = smem + E9_RM;
    unsigned char* kt = smem + 2 * E9_RM;
    float* kbias = reinterpret_cast<float*>(smem + 2 * E9_RM + E9_TR);          // 0 for a live key, -inf for a masked one / past T
    uint4 qf[E9_KS], dof[E9_KS];
    float del = 0.f;
    e9_load_frags(qf, Q, p.q_ts, ownc, hi);                                      // the own side first: the first tile needs it
    e9_load_frags(dof, dO, p.o_ts, ownc, hi);
    uint4 gk = issue(K, p.k_ts, 0), gv = issue(V, p.v_ts, 0);                    // the streamed side one chunk ahead
    {
      uint4 of[E9_KS];
      e9_load_frags(of, O, p.o_ts, ownc, hi);
#pragma unroll
      for (int ks = 0; ks < E9_KS; ++ks) {
        const uint32_t a[4] = {dof[ks].x, dof[ks].y, dof[ks].z, dof[ks].w}, c[4] = {of[ks].x, of[ks].y, of[ks].z, of[ks].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) del += h2f_lo(a[e]) * h2f_lo(c[e]) + h2f_hi(a[e]) * h2f_hi(c[e]);
      }
      del += __shfl_xor(del, 32);
    }
    const float l0 = lse[ownc];
    // a query row with no live key at all (lse = -inf) has P = 0 everywhere; rows past T contribute nothing and are not stored
    const float nl = (own < T && l0 != -__builtin_huge_valf()) ? -l0 * E9_LOG2E : -__builtin_huge_valf();
    // the score term of this lane's query row, shifted so that key j sits at brow[j]
    const float* brow = p.bias + bb + (int64_t)ownc * p.bias_rs + (T - 1 - ownc);
    f32x16 dq[E9_CT];
#pragma unroll
    for (int ct = 0; ct < E9_CT; ++ct) e9_zero(dq[ct]);
    for (int cc = 0; cc < nchunk; ++cc) {
      const int c = cc & 3, s0 = (cc >> 2) * E9_TI, Ts = min(E9_TI, T - s0);
      if (c == 0) {
        if (cc > 0) __syncthreads();                               // every wave is done with the previous super-chunk's images
        const uint8_t kmb = km ? km[min(s0 + tid, T - 1)] : (uint8_t)1;
        if (tid < E9_TI) kbias[tid] = (tid < Ts && kmb) ? 0.f : -__builtin_huge_valf();
      }
      // the tile's score-term columns go out FIRST: the staging, its two barriers and the first products hide their way (loaded at the
      // head of the tile they were a round trip to L2 on every tile's critical path: 46 -> 4x us per launch)
      float bz[16];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const e9_f32x4_a4 v4 = *reinterpret_cast<const e9_f32x4_a4*>(brow + ((p.ablate & 1) ? 0 : cc * 32 + 8 * q + 4 * hi));
        bz[4 * q] = v4.x; bz[4 * q + 1] = v4.y; bz[4 * q + 2] = v4.z; bz[4 * q + 3] = v4.w;
      }
      e9_stage_rm<false>(krm, gk, gk, nullptr, Ts, tid, c);
      e9_stage_rm<false>(vrm, gv, gv, nullptr, Ts, tid, c);
      gk = issue(K, p.k_ts, min(cc + 1, nchunk - 1));              // (the last chunk again after the last one: dropped)
      gv = issue(V, p.v_ts, min(cc + 1, nchunk - 1));
      __syncthreads();
      e9_transpose(kt, krm, c, tid & 255, 2 * (tid >> 8), 2);
      __syncthreads();
      if (wave_live && !(p.ablate & 4)) {
        const unsigned char* kr = krm + c * 32 * E9_HS;
        const unsigned char* vr = vrm + c * 32 * E9_HS;
        f32x16 st, dp;
        e9_zero(st); e9_zero(dp);
#pragma unroll
        for (int ks = 0; ks < E9_KS; ++ks) { mma32(st, e9_frag(kr, m, hi, ks), qf[ks]); mma32(dp, e9_frag(vr, m, hi, ks), dof[ks]); }
        float dsv[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 kb = *reinterpret_cast<const float4*>(kbias + c * 32 + 8 * q + 4 * hi);
          const float kb4[4] = {kb.x, kb.y, kb.z, kb.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int r = 4 * q + e;
            const float pe = __builtin_amdgcn_exp2f(__builtin_fmaf(st[r] + bz[r], sc2, nl) + kb4[e]);
            dsv[r] = pe * (dp[r] - del);
          }
        }
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
          const uint4 pb = e9_pack8(dsv + 8 * k2);
#pragma unroll
          for (int ct = 0; ct < E9_CT; ++ct) mma32(dq[ct], e9_tfrag(kt, 32 * ct + m, c * 32, hi, k2), pb);
        }
      }
    }
    __syncthreads();                                               // every wave is done with the images: they become staging space
    if (wave_live) e9_store_rows(dq, p.scale, true, smem + 32 * wid * E9_HS, p.dq + (int64_t)b * p.q_bs + h * E9_DK, p.q_ts, o0 + 32 * wid, T, lane);
  } else {
    // ------------------------------------------------------------------ lane = key: dV = P^T dO, dK = scale . dS^T Q, d bias = scale . dS
    unsigned char* qrm = smem;
    unsigned char* dorm = smem + E9_RM;
    unsigned char* qt = smem + 2 * E9_RM;
    unsigned char* dot = smem + 2 * E9_RM + E9_TR;
    float* nls = reinterpret_cast<float*>(smem + 2 * E9_RM + 2 * E9_TR);       // -lse log2(e) per query (-inf: no live key / past T), all T of them
    float* dels = nls + E9_TMAX;                                               // rowsum(dO . O) per query
    uint4 kf[E9_KS], vf[E9_KS];
    e9_load_frags(kf, K, p.k_ts, ownc, hi);                                      // the own side first: the first tile needs it
    e9_load_frags(vf, V, p.v_ts, ownc, hi);
    uint4 gq = issue(Q, p.q_ts, 0), gdo = issue(dO, p.o_ts, 0);                  // the streamed side one chunk ahead
    const uint8_t kmb = km ? km[ownc] : (uint8_t)1;
    // a masked key (or one past T) only pollutes ITS OWN dk / dv rows -- the lane is a column of every product here -- so the loop
    // carries no mask at all and the rows are zeroed on their way out (its d bias entries are written as zeros)
    const bool keyok = own < T && kmb;
    // -lse and delta of EVERY query first (one pass over dO and O, 16 lanes per row: the loop then carries neither the O pieces nor the
    // butterfly -- in the first version its registers, spilled, were most of the launch)
    for (int r0 = 0; r0 < T; r0 += 256) {                            // eight 32-row groups per trip, all their loads in flight together
      uint4 a[8], c4[8];
      const int ch = min(tid & 15, E9_PPR - 1);
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int rowc = min(r0 + 32 * u + (tid >> 4), T - 1);
        a[u] = ld_global_b128(dO + (int64_t)rowc * p.o_ts + 8 * ch);
        c4[u] = ld_global_b128(O + (int64_t)rowc * p.o_ts + 8 * ch);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int row = r0 + 32 * u + (tid >> 4);
        const uint32_t aw[4] = {a[u].x, a[u].y, a[u].z, a[u].w}, cw[4] = {c4[u].x, c4[u].y, c4[u].z, c4[u].w};
        float part = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) part += h2f_lo(aw[e]) * h2f_lo(cw[e]) + h2f_hi(aw[e]) * h2f_hi(cw[e]);
        if ((tid & 15) >= E9_PPR) part = 0.f;
        part += __shfl_xor(part, 1);
        part += __shfl_xor(part, 2);
        part += __shfl_xor(part, 4);
        part += __shfl_xor(part, 8);
        if ((tid & 15) == 0 && row < T) dels[row] = part;
      }
    }
    for (int r = tid; r < E9_TMAX; r += 512) {
      const float l0 = lse[min(r, T - 1)];
      nls[r] = (r < T && l0 != -__builtin_huge_valf()) ? -l0 * E9_LOG2E : -__builtin_huge_valf();
      if (r >= T) dels[r] = 0.f;
    }
    // this lane's key column of the score term: element (i, own) sits at bcol + i (rs - 1)
    const int64_t bcol = bb + (T - 1) + ownc, bstep = p.bias_rs - 1;
    f32x16 dk[E9_CT], dv[E9_CT];
#pragma unroll
    for (int ct = 0; ct < E9_CT; ++ct) { e9_zero(dk[ct]); e9_zero(dv[ct]); }
    for (int cc = 0; cc < nchunk; ++cc) {
      const int c = cc & 3, s0 = (cc >> 2) * E9_TI, Ts = min(E9_TI, T - s0);
      if (c == 0 && cc > 0) __syncthreads();                       // every wave is done with the previous super-chunk's images
      float bz[16];                                                // the tile's score-term entries of this key column (queries clamped into T), out first
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int i = min(cc * 32 + 8 * (r >> 2) + 4 * hi + (r & 3), T - 1);
        bz[r] = p.bias[(p.ablate & 1) ? bb : bcol + (int64_t)i * bstep];
      }
      e9_stage_rm<false>(qrm, gq, gq, nullptr, Ts, tid, c);
      e9_stage_rm<false>(dorm, gdo, gdo, nullptr, Ts, tid, c);
      gq = issue(Q, p.q_ts, min(cc + 1, nchunk - 1));              // (the last chunk again after the last one: dropped)
      gdo = issue(dO, p.o_ts, min(cc + 1, nchunk - 1));
      __syncthreads();                                             // (the first one also publishes nls / dels)
      if (tid < 256) e9_transpose(qt, qrm, c, tid, 0, 4);         // wave-uniform split: four waves per image
      else e9_transpose(dot, dorm, c, tid - 256, 0, 4);
      __syncthreads();
      if (wave_live && !(p.ablate & 4)) {
        const unsigned char* qr = qrm + c * 32 * E9_HS;
        const unsigned char* dr = dorm + c * 32 * E9_HS;
        f32x16 st, dp;
        e9_zero(st); e9_zero(dp);
#pragma unroll
        for (int ks = 0; ks < E9_KS; ++ks) { mma32(st, e9_frag(qr, m, hi, ks), kf[ks]); mma32(dp, e9_frag(dr, m, hi, ks), vf[ks]); }
        float pv[16], dsv[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 nl = *reinterpret_cast<const float4*>(nls + cc * 32 + 8 * q + 4 * hi);
          const float4 de = *reinterpret_cast<const float4*>(dels + cc * 32 + 8 * q + 4 * hi);
          const float nl4[4] = {nl.x, nl.y, nl.z, nl.w}, de4[4] = {de.x, de.y, de.z, de.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int r = 4 * q + e;
            pv[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(st[r] + bz[r], sc2, nl4[e]));
            dsv[r] = pv[r] * (dp[r] - de4[e]);
          }
        }
        if (own < T && !(p.ablate & 2)) {                          // d bias: every in-range (query, key) pair, masked ones with 0
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int i = cc * 32 + 8 * (r >> 2) + 4 * hi + (r & 3);
            if (i < T) {
              const float gvv = keyok ? dsv[r] * p.scale : 0.f;
              const int64_t idx = bcol + (int64_t)i * bstep;
              if (p.dbias_h16) reinterpret_cast<uint16_t*>(p.dbias)[idx] = (uint16_t)(pack2h(gvv, 0.f) & 0xffffu);
              else reinterpret_cast<float*>(p.dbias)[idx] = gvv;
            }
          }
        }
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
          const uint4 pb = e9_pack8(pv + 8 * k2), sb = e9_pack8(dsv + 8 * k2);
#pragma unroll
          for (int ct = 0; ct < E9_CT; ++ct) {
            mma32(dv[ct], e9_tfrag(dot, 32 * ct + m, c * 32, hi, k2), pb);
            mma32(dk[ct], e9_tfrag(qt, 32 * ct + m, c * 32, hi, k2), sb);
          }
        }
      }
    }
    __syncthreads();
    if (wave_live) {                                               // both through the wave's own 32 rows of the staging area, one after the other
      e9_store_rows(dk, p.scale, keyok, smem + 32 * wid * E9_HS, p.dk + (int64_t)b * p.k_bs + h * E9_DK, p.k_ts, o0 + 32 * wid, T, lane);
      e9_store_rows(dv, 1.f, keyok, smem + 32 * wid * E9_HS, p.dv + (int64_t)b * p.v_bs + h * E9_DK, p.v_ts, o0 + 32 * wid, T, lane);
    }
  }
}

}  // namespace

extern int g_otr_attn_enc96;       // api.hip (otr_debug_set(33, v))

// shapes this kernel serves (attention.hip asks before it takes its own path): 16-bit, head dim 96, self-attention without a causal mask,
// WITH the relative-position score term in fp32 whose rows allow the 16-byte loads (AttnArgs.bias_vec4), aligned operands
bool encattn96_bwd_takes(int dtype_is_h16, int dk, int Tq, int Tk, int causal, int has_bias, int rel_shift, int bias_vec4, int has_dbias, int vec) {
  return g_otr_attn_enc96 && dtype_is_h16 && dk == E9_DK && Tq == Tk && Tq >= 1 && Tq <= E9_TMAX && !causal && has_bias && rel_shift && bias_vec4 &&
         has_dbias && vec;
}

int32_t encattn96_bwd_launch(const void* q, const void* k, const void* v, const void* o, const void* do_, const float* lse, const uint8_t* key_mask,
                             const float* bias, void* dbias, int dbias_h16, int64_t bias_bs, int64_t bias_hs, int64_t bias_rs, void* dq, void* dk,
                             void* dv, int B, int H, int T, int64_t q_bs, int64_t q_ts, int64_t k_bs, int64_t k_ts, int64_t v_bs, int64_t v_ts,
                             int64_t o_bs, int64_t o_ts, float scale, hipStream_t stream) {
  E9Args p{};
  p.q = (const uint16_t*)q; p.k = (const uint16_t*)k; p.v = (const uint16_t*)v; p.o = (const uint16_t*)o; p.do_ = (const uint16_t*)do_;
  p.dq = (uint16_t*)dq; p.dk = (uint16_t*)dk; p.dv = (uint16_t*)dv; p.key_mask = key_mask; p.lse = lse;
  p.bias = bias; p.dbias = dbias; p.dbias_h16 = dbias_h16; p.bias_bs = bias_bs; p.bias_hs = bias_hs; p.bias_rs = bias_rs;
  p.B = B; p.H = H; p.T = T; p.q_bs = q_bs; p.q_ts = q_ts; p.k_bs = k_bs; p.k_ts = k_ts; p.v_bs = v_bs; p.v_ts = v_ts; p.o_bs = o_bs; p.o_ts = o_ts;
  p.scale = scale; p.ablate = g_otr_attn_enc96 >> 1;
  const unsigned nob = (unsigned)((T + E9_OWN - 1) / E9_OWN);
  const unsigned grid = 8u * 2u * nob * (unsigned)((H * B + 7) / 8);
  hipLaunchKernelGGL(encattn96_bwd_kernel, dim3(grid), dim3(512), 0, stream, p);
  return otr_check_launch("encattn96_bwd");
}
