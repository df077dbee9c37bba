// fp32-MFMA instantiations of the GEMM core (v_mfma_f32_16x16x4_f32: exact fp32, the parity mode).
#include "gemm_kernel.h"

#define CASE(AM, BM_) return gemm_launch_tiles<float, float, float, float, AM, BM_>(a, s)

int32_t gemm_dispatch_f32(const GemmArgs& a, int ad, int bd, int cd, int amode, int bmode, hipStream_t s) {
  if (ad != OTR_F32 || bd != OTR_F32 || cd != OTR_F32) {
    otr_set_error("gemm(f32): all operands must be f32 (got a=%d b=%d c=%d)", ad, bd, cd);
    return -2;
  }
  const int key = (amode << 4) | bmode;
  switch (key) {
    case (MODE_KC << 4) | MODE_KC: CASE(MODE_KC, MODE_KC);
    case (MODE_KC << 4) | MODE_MC: CASE(MODE_KC, MODE_MC);
    case (MODE_MC << 4) | MODE_MC: CASE(MODE_MC, MODE_MC);
    case (MODE_IM2K << 4) | MODE_KC: CASE(MODE_IM2K, MODE_KC);
    case (MODE_MC << 4) | MODE_IM2M: CASE(MODE_MC, MODE_IM2M);
    default:
      otr_set_error("gemm(f32): unsupported combination amode=%d bmode=%d", amode, bmode);
      return -2;
  }
}

int32_t gemm_grouped_wgrad_f32(const GroupDesc* d, int n, int ad, int bd, int big, void* tm, int64_t tb, hipStream_t s) {
#define GCASE(AT, BT)                                                                  \
  return big ? gemm_grouped_launch<float, AT, BT, 128, 128>(d, n, tm, tb, s) : gemm_grouped_launch<float, AT, BT, 64, 64>(d, n, tm, tb, s)
  if (ad == OTR_F32 && bd == OTR_F32) { GCASE(float, float); }
  if (ad == OTR_F32 && bd == OTR_H16) { GCASE(float, bf16_t); }
  if (ad == OTR_H16 && bd == OTR_F32) { GCASE(bf16_t, float); }
  GCASE(bf16_t, bf16_t);
#undef GCASE
}
