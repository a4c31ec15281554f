// LDS-DMA GEMM family, TERMS = 6: NATIVE fp16 operands (16-bit in HBM and LDS).
#define ICK_GLDS_TERMS 6
#define ICK_GLDS_ENTRY run_glds_h16_t6
#include "igemm_glds_impl.h"
