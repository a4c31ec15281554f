// igemm_h16.hip — entry of the NATIVE 16-bit GEMM variants (operands bf16 / fp16 in HBM, fp32 accumulate and fp32 output) and
// the cast kernel that produces such operands.  The reference's autocast keeps activations and a weight copy in fp16
// (train_student_kd.py:271): this is that storage regime for the k-contiguous contractions (Linear forward, convolution
// forward, stride-1 data gradients run as forward convolutions).  The kernel is igemm_glds_impl.h addressed in units of two
// halves — this file only halves the k-extents.
#include "igemm_params.h"

namespace ickg {
bool glds_eligible(const IckGemm* d);                                           // igemm_f32_glds.hip
int run_glds_h16_t5(const IckGemm* d, const P& p, int nz, hipStream_t st);      // bf16
int run_glds_h16_t6(const IckGemm* d, const P& p, int nz, hipStream_t st);      // fp16
}

namespace {
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
template <typename H4>
__global__ void cast16_kernel(const float* __restrict__ x, void* __restrict__ y, long n4) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const float4 v = reinterpret_cast<const float4*>(x)[i];
    const f32x4 f = {v.x, v.y, v.z, v.w};
    reinterpret_cast<H4*>(y)[i] = __builtin_convertvector(f, H4);
  }
}
}  // namespace

extern "C" int ick_cast_f32_to_16(const float* x, void* y, int64_t n, int fp16, void* stream) {
  ICK_REQUIRE(x && y && n > 0 && n % 4 == 0, "ick_cast_f32_to_16: n %% 4");
  long g = (n / 4 + 255) / 256; if (g > 4096) g = 4096;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (fp16) ICK_LAUNCH(cast16_kernel<f16x4>, dim3((int)g), dim3(256), 0, st, x, y, (long)(n / 4));
  else ICK_LAUNCH(cast16_kernel<bf16x4>, dim3((int)g), dim3(256), 0, st, x, y, (long)(n / 4));
  return ick::launch_status("cast_f32_to_16");
}

// A [M][K] (lda) and B [N][K] (ldb) hold 16-bit elements; M, N, K, lda, ldb and the conv geometry are given in ELEMENTS.
extern "C" int ick_gemm_h16(const IckGemm* d0, int fp16, void* stream) {
  using namespace ickg;
  ICK_REQUIRE(d0 != nullptr, "ick_gemm_h16: null descriptor");
  ICK_REQUIRE(d0->op == ICK_OP_NT || d0->op == ICK_OP_CONV_FWD, "ick_gemm_h16: NT and CONV_FWD only (k-contiguous operands)");
  ICK_REQUIRE(d0->K % 8 == 0 && d0->lda % 8 == 0 && d0->ldb % 8 == 0, "ick_gemm_h16: K, lda, ldb must be multiples of 8 halves");
  ICK_REQUIRE((d0->batch_outer <= 1 && d0->batch_inner <= 1) && d0->splitk <= 1, "ick_gemm_h16: no batching / split-K");
  IckGemm dd = *d0;
  dd.K /= 2; dd.lda /= 2; dd.ldb /= 2;                       // units of two halves: the kernel then runs the fp32 addressing
  if (dd.op == ICK_OP_CONV_FWD) {
    ICK_REQUIRE(dd.Cin % 64 == 0, "ick_gemm_h16: Cin must be a multiple of 64 (32 units)");
    dd.Cin /= 2;
  }
  dd.tile &= 255;
  ICK_REQUIRE(glds_eligible(&dd), "ick_gemm_h16: shape not eligible for the LDS-DMA kernel");
  P p; int nz = 1;
  if (int rc = prepare(&dd, 32, p, nz, "ick_gemm_h16")) return rc;
  if (dd.op == ICK_OP_CONV_FWD)
    ICK_REQUIRE(p.M == p.Nb * p.Ho * p.Wo && p.N == p.Cout && p.K == p.R * p.S * p.Cin && p.ldb == p.K, "ick_gemm_h16: M/N/K do not match the geometry");
  hipStream_t st = static_cast<hipStream_t>(stream);
  return fp16 ? run_glds_h16_t6(&dd, p, nz, st) : run_glds_h16_t5(&dd, p, nz, st);
}
