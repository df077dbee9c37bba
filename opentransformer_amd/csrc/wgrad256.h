// Host <-> kernel types of the 256 x 256-tile weight-gradient launch (wgrad256.hip), used by api.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "common.h"

constexpr int W256_MAX_PROBS = 56;      // the table travels in the kernel-argument segment (< 4 KB): hipGraph-capturable by value

struct W256Item {                       // dw[N,K] += dy[M,N]^T x[M,K]; 16-bit operands, N % 8 == 0, K % 8 == 0
  const void* dy;
  const void* x;
  float* dw;
  float* dbias;                         // NULL or [N]: dbias += column sums of dy (the bias gradient of the same Linear)
  int M, N, K;
  int64_t ldy, ldx, ldw;
  int overwrite;                        // != 0: dw holds zeros nobody else has written in this pass: the first partial sum is STORED
};

struct W256Prob {
  const uint16_t* dy;
  const uint16_t* x;
  float* dw;
  float* dbias;
  int start;                            // first slab of this problem in the launch's (tile, slab) space
  int M, N, K, ldy, ldx, ldw;           // tiles: ceil(N/256) x ceil(K/256), each ceil(M/16) slabs long
  int flag0;                            // first turnstile flag of this problem (one per tile); bit 30: overwrite (W256Item)
};                                      // 64 bytes: 56 of them + the scalars below stay under the 4 KB argument segment

struct W256Args {
  W256Prob p[W256_MAX_PROBS];
  int total, chunk;                     // slabs in the launch; slabs per workgroup (stream-K) / per tile (rounds)
  int mode, nfull, rem_tiles, parts;    // 1 = rounds schedule: full rounds, tiles left for the last round, row ranges per tile there
  int nprob, spin_limit;
  int ablate, policy;                      // tuning hook (otr_debug_set(8, v)): 1 = no MFMA, 2 = no DMA after the prologue, 4 = no accumulation into dw; v >> 3: 0 = default policy (non-temporal unshared strips); else (v >> 3) - 1 = bit 0 non-temporal strips.  Ablation 6 = the x part is not fetched (its DMA reads one zero line)
  // mode 2 (one problem: the conv2 weight gradient of a C1 % 256 == 0 frontend, wgrad256_conv_launch): the x operand of the k-tile
  // (tap, 256 channels) is GATHERED -- row m = output pixel (b, t2, f2) reads act1[b, 2 t2 + kh, 2 f2 + kw - 1, channels], the
  // zero line where the frequency tap falls into the padding -- and every tile is cut into `parts` row ranges whose partial
  // tiles go to split_ws[part] (plain stores, no turnstile); w256_reduce_kernel sums them in a fixed order.
  int cT1, cF1, cT2, cF2, cC1;
  FastDiv cdivF2, cdivT2;
  float* split_ws;
  int* flags;
  const void* zeros;                    // >= 64 zero bytes: source of rows past M
  int* fault;                           // NULL or the sticky device fault word (otr_set_fault_counter): +1 per piece that gave up
};

// Eligibility is the caller's business (api.hip); workspace holds 64 zero bytes + one int per tile.
int32_t wgrad256_launch(const W256Item* items, int n, void* workspace, int64_t workspace_bytes, int grid_cap, int ablate, hipStream_t s);
int64_t wgrad256_workspace_bytes(const W256Item* items, int n);
// conv2 weight gradient dw2r[C2, 9 C1] = g2[M, C2]^T im2col(act1) on the same kernel (x rows gathered; see W256Args mode 2).
// 0 = launched, 1 = shape / workspace not served (the caller's generic GEMM takes it), < 0 = error
int32_t wgrad256_conv_launch(const void* g2, const void* act1, float* dw2r, int B, int T1, int F1, int T2, int F2, int C1, int C2,
                             void* workspace, int64_t workspace_bytes, hipStream_t s);
