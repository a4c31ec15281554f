// igemm_h16.hip — entry of the NATIVE 16-bit GEMM family (operands bf16 / fp16 in HBM and in LDS, fp32 accumulation, C and the
// residual fp32 or 16-bit) and the casts that produce / consume such storage.  The reference's autocast keeps activations and a
// weight copy in fp16 (train_student_kd.py:271): this is that storage regime for every op of the implicit-GEMM family.
//   * both operands k-contiguous (Linear forward, convolution forward, stride-1 data gradients run as forward convolutions):
//     the LDS-DMA kernel of igemm_glds_impl.h (TERMS 5 / 6) addressed in units of two halves — this file halves K, lda, ldb, Cin;
//   * everything else (weight gradients, stride-2 data gradients, NN / TN): the register-staged kernel of igemm_bf16.hip with
//     IN16 = true (8-byte fetches of four halves, transposed LDS reads).
#include "igemm_params.h"
#include <cstdlib>

namespace ickg {
bool glds_eligible(const IckGemm* d);                                           // igemm_f32_glds.hip
int run_glds_h16_t5(const IckGemm* d, const P& p, int nz, hipStream_t st);      // bf16, LDS-DMA (igemm_glds_impl.h)
int run_glds_h16_t6(const IckGemm* d, const P& p, int nz, hipStream_t st);      // fp16
int check_bf16_shapes(const IckGemm* d, const P& p, int nz, int* to_f32, const char* who);     // igemm_bf16.hip
int run_regs_h16(const IckGemm* d, int fp16, const P& p, int nz, hipStream_t st);             // register-staged, IN16
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

template <typename H4>
__global__ void cast32_kernel(const void* __restrict__ x, float* __restrict__ y, long n4) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const f4 f = __builtin_convertvector(reinterpret_cast<const H4*>(x)[i], f4);
    reinterpret_cast<float4*>(y)[i] = make_float4(f.x, f.y, f.z, f.w);
  }
}

extern "C" int ick_cast_16_to_f32(const void* x, float* y, int64_t n, int fp16, void* stream) {
  ICK_REQUIRE(x && y && n > 0 && n % 4 == 0, "ick_cast_16_to_f32: n %% 4");
  long g = (n / 4 + 255) / 256; if (g > 4096) g = 4096;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (fp16) ICK_LAUNCH(cast32_kernel<f16x4>, dim3((int)g), dim3(256), 0, st, x, y, (long)(n / 4));
  else ICK_LAUNCH(cast32_kernel<bf16x4>, dim3((int)g), dim3(256), 0, st, x, y, (long)(n / 4));
  return ick::launch_status("cast_16_to_f32");
}

// All extents in ELEMENTS.  k-contiguous ops whose extents allow it run the LDS-DMA kernel addressed in units of two halves;
// everything else (and IckGemm.tile bit 8) the register-staged IN16 kernel.
extern "C" int ick_gemm_h16(const IckGemm* d0, int fp16, void* stream) {
  using namespace ickg;
  ICK_REQUIRE(d0 != nullptr, "ick_gemm_h16: null descriptor");
  hipStream_t st = static_cast<hipStream_t>(stream);
  {
    P p; int nz = 1;
    if (int rc = prepare(d0, 32, p, nz, "ick_gemm_h16")) return rc;
    int to_f32 = 0;
    if (int rc = check_bf16_shapes(d0, p, nz, &to_f32, "ick_gemm_h16")) return rc;
    ICK_REQUIRE(!to_f32, "ick_gemm_h16: conv channel counts must be multiples of 32");
    ICK_REQUIRE(d0->op != ICK_OP_CONV_FWD_C4, "ick_gemm_h16: the 4-channel stem convolution reads fp32 images (ick_gemm_bf16)");
    ICK_REQUIRE((d0->op != ICK_OP_NT && d0->op != ICK_OP_NN) || p.K % 4 == 0, "ick_gemm_h16: NT / NN need K %% 4 == 0 (four k per fetch)");
    ICK_REQUIRE(!(p.c16 && p.splitk > 1), "ick_gemm_h16: no split-K into a 16-bit C");
    const bool kc = d0->op == ICK_OP_NT || d0->op == ICK_OP_CONV_FWD;
    const bool units = kc && !(d0->tile & 256) && p.K % 8 == 0 && p.lda % 8 == 0 && p.ldb % 8 == 0 && nz == 1 && p.splitk == 1 &&
                       (d0->op != ICK_OP_CONV_FWD || p.Cin % 64 == 0);
    // x-contiguous operands on the LDS-DMA kernel (round 3; igemm_glds_impl.h X16): weight gradients and TN products in
    // ELEMENT extents, k-tiles of 64 rows, transposed LDS reads.  8-half chunks must not straddle a row end or a filter tap.
    static const bool x16_on = [] { const char* e = getenv("ICK_X16_GLDS"); return !(e && e[0] == '0'); }();
    const bool xc = (d0->op == ICK_OP_TN || d0->op == ICK_OP_CONV_WGRAD) && x16_on && !(d0->tile & 256) && nz == (p.splitk > 1 ? p.splitk : 1) &&
                    p.M % 8 == 0 && p.N % 8 == 0 && !p.c16 && !p.r16 &&
                    (d0->op == ICK_OP_TN ? (p.lda % 8 == 0 && p.ldb % 8 == 0) : (p.Cin % 8 == 0 && p.Cout % 8 == 0));
    if (xc) {
      IckGemm dx = *d0;
      dx.tile &= 255;
      P px; int nzx = 1;
      if (int rc = prepare(&dx, 64, px, nzx, "ick_gemm_h16")) return rc;      // split-K slices in multiples of the 64-row k-tile
      return fp16 ? run_glds_h16_t6(&dx, px, nzx, st) : run_glds_h16_t5(&dx, px, nzx, st);
    }
    if (!units) return run_regs_h16(d0, fp16, p, nz, st);
  }
  IckGemm dd = *d0;
  dd.K /= 2; dd.lda /= 2; dd.ldb /= 2;                       // units of two halves: the kernel then runs the fp32 addressing
  if (dd.op == ICK_OP_CONV_FWD) dd.Cin /= 2;
  dd.tile &= 255;
  ICK_REQUIRE(glds_eligible(&dd), "ick_gemm_h16: internal: unit-addressed shape not eligible for the LDS-DMA kernel");
  P p; int nz = 1;
  if (int rc = prepare(&dd, 32, p, nz, "ick_gemm_h16")) return rc;
  return fp16 ? run_glds_h16_t6(&dd, p, nz, st) : run_glds_h16_t5(&dd, p, nz, st);
}
