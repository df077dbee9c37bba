// bf16-MFMA instantiations of the GEMM core (v_mfma_f32_16x16x32_bf16, fp32 accumulate).
#include "gemm_kernel.h"

#define CASE(AT, BT, OT, AM, BM_) return gemm_launch_tiles<bf16_t, AT, BT, OT, AM, BM_>(a, s)

int32_t gemm_dispatch_bf16(const GemmArgs& a, int ad, int bd, int cd, int amode, int bmode, hipStream_t s) {
  // dtype codes -> 0 (f32) / 1 (the 16-bit type of this build)
  const int key = (amode << 12) | (bmode << 8) | ((ad != OTR_F32) << 2) | ((bd != OTR_F32) << 1) | (cd != OTR_F32);
  constexpr int H16 = 1, F32 = 0;
  switch (key) {
    // linear forward: x[M,K] KC (f32|bf16) x w[N,K] KC (f32|bf16)
    case (MODE_KC << 12) | (MODE_KC << 8) | (F32 << 2) | (F32 << 1) | F32: CASE(float, float, float, MODE_KC, MODE_KC);
    case (MODE_KC << 12) | (MODE_KC << 8) | (F32 << 2) | (F32 << 1) | H16: CASE(float, float, bf16_t, MODE_KC, MODE_KC);
    case (MODE_KC << 12) | (MODE_KC << 8) | (H16 << 2) | (F32 << 1) | F32: CASE(bf16_t, float, float, MODE_KC, MODE_KC);
    case (MODE_KC << 12) | (MODE_KC << 8) | (H16 << 2) | (F32 << 1) | H16: CASE(bf16_t, float, bf16_t, MODE_KC, MODE_KC);
    case (MODE_KC << 12) | (MODE_KC << 8) | (F32 << 2) | (H16 << 1) | F32: CASE(float, bf16_t, float, MODE_KC, MODE_KC);
    case (MODE_KC << 12) | (MODE_KC << 8) | (F32 << 2) | (H16 << 1) | H16: CASE(float, bf16_t, bf16_t, MODE_KC, MODE_KC);
    case (MODE_KC << 12) | (MODE_KC << 8) | (H16 << 2) | (H16 << 1) | F32: CASE(bf16_t, bf16_t, float, MODE_KC, MODE_KC);
    case (MODE_KC << 12) | (MODE_KC << 8) | (H16 << 2) | (H16 << 1) | H16: CASE(bf16_t, bf16_t, bf16_t, MODE_KC, MODE_KC);
    // dgrad: dy[M,N] KC x w[N,K] MC
    case (MODE_KC << 12) | (MODE_MC << 8) | (F32 << 2) | (F32 << 1) | F32: CASE(float, float, float, MODE_KC, MODE_MC);
    case (MODE_KC << 12) | (MODE_MC << 8) | (F32 << 2) | (F32 << 1) | H16: CASE(float, float, bf16_t, MODE_KC, MODE_MC);
    case (MODE_KC << 12) | (MODE_MC << 8) | (H16 << 2) | (F32 << 1) | F32: CASE(bf16_t, float, float, MODE_KC, MODE_MC);
    case (MODE_KC << 12) | (MODE_MC << 8) | (H16 << 2) | (F32 << 1) | H16: CASE(bf16_t, float, bf16_t, MODE_KC, MODE_MC);
    case (MODE_KC << 12) | (MODE_MC << 8) | (F32 << 2) | (H16 << 1) | F32: CASE(float, bf16_t, float, MODE_KC, MODE_MC);
    case (MODE_KC << 12) | (MODE_MC << 8) | (F32 << 2) | (H16 << 1) | H16: CASE(float, bf16_t, bf16_t, MODE_KC, MODE_MC);
    case (MODE_KC << 12) | (MODE_MC << 8) | (H16 << 2) | (H16 << 1) | F32: CASE(bf16_t, bf16_t, float, MODE_KC, MODE_MC);
    case (MODE_KC << 12) | (MODE_MC << 8) | (H16 << 2) | (H16 << 1) | H16: CASE(bf16_t, bf16_t, bf16_t, MODE_KC, MODE_MC);
    // wgrad: dy^T (MC) x x (MC) -> f32
    case (MODE_MC << 12) | (MODE_MC << 8) | (F32 << 2) | (F32 << 1) | F32: CASE(float, float, float, MODE_MC, MODE_MC);
    case (MODE_MC << 12) | (MODE_MC << 8) | (F32 << 2) | (H16 << 1) | F32: CASE(float, bf16_t, float, MODE_MC, MODE_MC);
    case (MODE_MC << 12) | (MODE_MC << 8) | (H16 << 2) | (F32 << 1) | F32: CASE(bf16_t, float, float, MODE_MC, MODE_MC);
    case (MODE_MC << 12) | (MODE_MC << 8) | (H16 << 2) | (H16 << 1) | F32: CASE(bf16_t, bf16_t, float, MODE_MC, MODE_MC);
    // conv2 forward (implicit im2col A) and wgrad (implicit im2col B)
    case (MODE_IM2K << 12) | (MODE_KC << 8) | (H16 << 2) | (F32 << 1) | H16: CASE(bf16_t, float, bf16_t, MODE_IM2K, MODE_KC);
    case (MODE_IM2K << 12) | (MODE_KC << 8) | (H16 << 2) | (H16 << 1) | H16: CASE(bf16_t, bf16_t, bf16_t, MODE_IM2K, MODE_KC);
    case (MODE_MC << 12) | (MODE_IM2M << 8) | (H16 << 2) | (H16 << 1) | F32: CASE(bf16_t, bf16_t, float, MODE_MC, MODE_IM2M);
    default:
      otr_set_error("gemm(bf16): unsupported combination amode=%d bmode=%d a=%d b=%d c=%d", amode, bmode, ad, bd, cd);
      return -2;
  }
}

int32_t gemm_grouped_wgrad_bf16(const GroupDesc* d, int n, int ad, int bd, int big, void* tm, int64_t tb, hipStream_t s) {
#define GCASE(AT, BT)                                                                  \
  return big ? gemm_grouped_launch<bf16_t, AT, BT, 128, 128>(d, n, tm, tb, s) : gemm_grouped_launch<bf16_t, AT, BT, 64, 64>(d, n, tm, tb, s)
  if (ad == OTR_F32 && bd == OTR_F32) { GCASE(float, float); }
  if (ad == OTR_F32 && bd == OTR_H16) { GCASE(float, bf16_t); }
  if (ad == OTR_H16 && bd == OTR_F32) { GCASE(bf16_t, float); }
  GCASE(bf16_t, bf16_t);
#undef GCASE
}
