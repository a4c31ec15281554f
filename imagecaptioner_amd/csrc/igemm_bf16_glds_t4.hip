// LDS-DMA GEMM family, TERMS = 4: fp32 by three fp16 products (hi*hi + 2^-11 (hi*lo' + lo'*hi)), NT / CONV_FWD.
#define ICK_GLDS_TERMS 4
#define ICK_GLDS_ENTRY run_glds_bf16_t4
#include "igemm_glds_impl.h"
