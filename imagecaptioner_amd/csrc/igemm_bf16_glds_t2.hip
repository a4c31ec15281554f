// LDS-DMA GEMM family, TERMS = 2: fp16 products (v_mfma_f32_32x32x16_f16), the reference's autocast dtype.
#define ICK_GLDS_TERMS 2
#define ICK_GLDS_ENTRY run_glds_bf16_t2
#include "igemm_glds_impl.h"
