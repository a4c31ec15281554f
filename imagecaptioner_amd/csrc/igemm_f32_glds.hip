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
    case ICK_OP_CONV_WGRAD: return false;   // both operands x-contiguous + per-lane gather: the register-staged kernel is faster (measured)
    case ICK_OP_CONV_FWD: return d->Cin % BK == 0;
    case ICK_OP_CONV_DGRAD: case ICK_OP_CONV_DGRAD_S2: return d->Cout % BK == 0;
    default: return false;
  }
}
}  // namespace ickg
