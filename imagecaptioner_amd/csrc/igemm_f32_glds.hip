// exact-fp32 instantiation (TERMS 0, v_mfma_f32_32x32x2_f32) of the LDS-DMA GEMM family + the eligibility rule shared by all variants.
// tools/ablate builds this file with -DICK_ABL=n.
#define ICK_GLDS_TERMS 0
#define ICK_GLDS_ENTRY run_glds
#include "igemm_glds_impl.h"

namespace ickg {
// true when the LDS-DMA kernel can run this problem (16-byte chunks must not straddle the K end or a filter tap)
bool glds_eligible(const IckGemm* d) {
  switch (d->op) {
    case ICK_OP_NT: case ICK_OP_NN: return d->K % 4 == 0;
    case ICK_OP_TN: case ICK_OP_CONV_FWD_C4: return true;
    case ICK_OP_CONV_WGRAD: {
      // rounds 1-2: both operands x-contiguous + a per-lane gather with two integer divisions per piece per k-tile: the
      // register-staged kernel was faster.  Round 3 keeps the pixel decomposition incrementally (igemm_glds_impl.h w_img /
      // w_oy / w_ox) and the LDS-DMA kernel wins by 7-13 % on every layer3 / layer4 shape (tools/bench_wgrad_glds.py,
      // profiles/r03_wgrad_glds_vs_regs.log: 256x2304x12544 157 against 169-180 us).  IckGemm.tile bit 8 still selects the
      // register-staged kernel; ICK_WGRAD_GLDS=0 switches the LDS-DMA form off (A/B).
      static const bool on = [] { const char* e = getenv("ICK_WGRAD_GLDS"); return !(e && e[0] == '0'); }();
      return on;
    }
    case ICK_OP_CONV_FWD: return d->Cin % BK == 0;
    case ICK_OP_CONV_DGRAD: case ICK_OP_CONV_DGRAD_S2: return d->Cout % BK == 0;
    default: return false;
  }
}
}  // namespace ickg
