// tools/ablate/mfma_chain.hip — does a chain of DEPENDENT v_mfma_f32_32x32x2_f32 (same accumulator back to back) issue at the
// matrix pipe's full rate?  Pure-register kernel: every wave runs ITER rounds over NACC accumulators; RUN = how many consecutive
// MFMAs hit the same accumulator before moving on (the tiled kernel: NACC 4, RUN 4; the fused attention's S^T phase: NACC 1).
//   hipcc -O3 --offload-arch=gfx950 -shared -fPIC tools/ablate/mfma_chain.hip -o tools/ablate/libmfma.so ; run_mfma_chain.py
#include <hip/hip_runtime.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int NACC, int RUN, int F16>
__global__ __launch_bounds__(256) void chain_kernel(float* out, int iters, float a0, float b0) {
  f32x16 acc[NACC];
#pragma unroll
  for (int c = 0; c < NACC; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
  float a = a0 + threadIdx.x * 1e-9f, b = b0;
  f16x8 ah, bh;
#pragma unroll
  for (int j = 0; j < 8; ++j) { ah[j] = (_Float16)(a0 + j); bh[j] = (_Float16)b0; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16 / (NACC * RUN) + (16 % (NACC * RUN) ? 1 : 0); ++u)
#pragma unroll
      for (int c = 0; c < NACC; ++c)
#pragma unroll
        for (int q = 0; q < RUN; ++q) {
          if (F16) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[c], 0, 0, 0);
          else acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
        }
  }
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < NACC; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[c][r];
  if (s == 1.2345e38f) out[threadIdx.x] = s;
}

template <int NACC, int RUN, int F16>
static long go(float* out, int blocks, int iters, void* st) {
  hipLaunchKernelGGL((chain_kernel<NACC, RUN, F16>), dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(st), out, iters, 1.0f, 0.5f);
  const int per_iter = (16 / (NACC * RUN) + (16 % (NACC * RUN) ? 1 : 0)) * NACC * RUN;
  return (long)per_iter;
}
// returns MFMAs per iteration per wave (so that the caller can compute the rate)
extern "C" long chain_run(int nacc, int run, int f16, float* out, int blocks, int iters, void* st) {
#define CASE(N, R) if (nacc == N && run == R) return f16 ? go<N, R, 1>(out, blocks, iters, st) : go<N, R, 0>(out, blocks, iters, st);
  CASE(1, 1) CASE(2, 1) CASE(4, 1) CASE(8, 1) CASE(4, 4) CASE(2, 2) CASE(2, 4) CASE(1, 4)
  return -1;
}
