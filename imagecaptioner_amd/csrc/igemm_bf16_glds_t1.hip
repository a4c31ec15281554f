// LDS-DMA GEMM family, TERMS = 1: bf16 products.
#define ICK_GLDS_TERMS 1
#define ICK_GLDS_ENTRY run_glds_bf16_t1
#include "igemm_glds_impl.h"

namespace ickg {
int run_glds_bf16_t3(const IckGemm* d, const P& p, int nz, hipStream_t st);   // igemm_bf16_glds_t3.hip
int run_glds_bf16_t2(const IckGemm* d, const P& p, int nz, hipStream_t st);   // igemm_bf16_glds_t2.hip (fp16)
int run_glds_bf16_t4(const IckGemm* d, const P& p, int nz, hipStream_t st);   // igemm_bf16_glds_t4.hip (fp32 by three fp16 products)

// the dispatcher over TERMS for ick_gemm_bf16 (igemm_bf16.hip)
int run_glds_bf16(const IckGemm* d, int terms, const P& p, int nz, hipStream_t st) {
  return terms == 4 ? run_glds_bf16_t4(d, p, nz, st) : terms == 3 ? run_glds_bf16_t3(d, p, nz, st) : terms == 2 ? run_glds_bf16_t2(d, p, nz, st) : run_glds_bf16_t1(d, p, nz, st);
}
}  // namespace ickg
