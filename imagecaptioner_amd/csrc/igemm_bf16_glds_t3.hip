// LDS-DMA GEMM family, TERMS = 3: split bf16: hi*hi + hi*lo + lo*hi.
#define ICK_GLDS_TERMS 3
#define ICK_GLDS_ENTRY run_glds_bf16_t3
#include "igemm_glds_impl.h"
