// LDS-DMA GEMM family, TERMS = 5: NATIVE bf16 operands (16-bit in HBM and LDS).
#define ICK_GLDS_TERMS 5
#define ICK_GLDS_ENTRY run_glds_h16_t5
#include "igemm_glds_impl.h"
