// tools/ablate/k64_stream.hip — the weight-stationary K = 64 streaming kernel of round 3 (not in libick.so: no faster than the
// tiled kernel) kept as an ablation harness for the question it raised: what bounds a product that writes 205 MB
// and reads 51 MB at 100 us?  MODE 0 = full kernel, 1 = no statistics, 2 = no stores (results consumed by an impossible branch),
// 3 = non-temporal stores, 4 = stores but no MFMAs (the accumulators keep their initial values).
//   hipcc -O3 --offload-arch=gfx950 -shared -fPIC tools/ablate/k64_stream.hip -o tools/ablate/libk64.so ; run_k64.py
#include <hip/hip_runtime.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int WN, int MODE>
__global__ __launch_bounds__(256, 2) void k64_stream_kernel(const float* __restrict__ A, long lda, const float* __restrict__ W, long ldb,
                                                           float* __restrict__ C, long ldc, int M, int N, double* __restrict__ stat_sum,
                                                           double* __restrict__ stat_sq, int copies, long stat_stride, int ntiles) {
  constexpr int WM = 4 / WN;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = lane & 31, half = lane >> 5;
  const int wn = wave % WN, wm = wave / WN;
  const int n0 = (blockIdx.y * WN + wn) * 64;
  float b[2][32];
#pragma unroll
  for (int sl = 0; sl < 2; ++sl) {
    const float* wr = W + (long)(n0 + 32 * sl + col) * ldb + 32 * half;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const float4 v = *reinterpret_cast<const float4*>(wr + 4 * q);
      b[sl][4 * q] = v.x; b[sl][4 * q + 1] = v.y; b[sl][4 * q + 2] = v.z; b[sl][4 * q + 3] = v.w;
    }
  }
  double s1[2] = {0.0, 0.0}, s2[2] = {0.0, 0.0};
  auto load = [&](int t, float (&a)[32]) {
    int row = t * 32 + col;
    row = row < M ? row : M - 1;
    const float* ar = A + (long)row * lda + 32 * half;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const float4 v = *reinterpret_cast<const float4*>(ar + 4 * q);
      a[4 * q] = v.x; a[4 * q + 1] = v.y; a[4 * q + 2] = v.z; a[4 * q + 3] = v.w;
    }
  };
  float a0[32], a1[32];
  int t = blockIdx.x * WM + wm;
  const int tstep = gridDim.x * WM;
  if (t < ntiles) load(t, a0);
  auto tile = [&](int tt, const float (&a)[32]) {
    f32x16 acc[2];
#pragma unroll
    for (int sl = 0; sl < 2; ++sl)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[sl][r] = MODE == 4 ? a[r] : 0.f;
    if (MODE != 4) {
#pragma unroll
      for (int kk = 0; kk < 32; ++kk) {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk], b[0][kk], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk], b[1][kk], acc[1], 0, 0, 0);
      }
    }
    const int r0 = tt * 32;
#pragma unroll
    for (int sl = 0; sl < 2; ++sl) {
      float* cp = C + n0 + 32 * sl + col;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = r0 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (row < M) {
          const float v = acc[sl][r];
          if (MODE == 2) { if (v == 1.2345e38f) cp[(long)row * ldc] = v; }
          else if (MODE == 3) __builtin_nontemporal_store(v, cp + (long)row * ldc);
          else cp[(long)row * ldc] = v;
          if (MODE == 0) { s1[sl] += (double)v; s2[sl] += (double)v * (double)v; }
        }
      }
    }
  };
  for (; t < ntiles; t += 2 * tstep) {
    if (t + tstep < ntiles) load(t + tstep, a1);
    tile(t, a0);
    if (t + tstep < ntiles) {
      if (t + 2 * tstep < ntiles) load(t + 2 * tstep, a0);
      tile(t + tstep, a1);
    }
  }
  if (MODE == 0 && stat_sum) {
#pragma unroll
    for (int sl = 0; sl < 2; ++sl) {
      double u = s1[sl], v = s2[sl];
      u += __shfl_xor(u, 32); v += __shfl_xor(v, 32);
      if (lane < 32) {
        const long so = (long)((blockIdx.x * WM + wm) % copies) * stat_stride + n0 + 32 * sl + col;
        atomicAdd(stat_sum + so, u); atomicAdd(stat_sq + so, v);
      }
    }
  }
}

template <int MODE>
static int go(const float* A, const float* W, float* C, int M, int N, double* ss, double* sq, int copies, void* stream, int gx_cap) {
  const int ntiles = (M + 31) / 32;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (N % 256 == 0) {
    int gx = ntiles < gx_cap ? ntiles : gx_cap;
    hipLaunchKernelGGL((k64_stream_kernel<4, MODE>), dim3(gx, N / 256), dim3(256), 0, st, A, 64L, W, 64L, C, (long)N, M, N, ss, sq, copies, (long)N, ntiles);
  } else {
    int gx = (ntiles + 3) / 4; if (gx > gx_cap) gx = gx_cap;
    hipLaunchKernelGGL((k64_stream_kernel<1, MODE>), dim3(gx, N / 64), dim3(256), 0, st, A, 64L, W, 64L, C, (long)N, M, N, ss, sq, copies, (long)N, ntiles);
  }
  return (int)hipGetLastError();
}
extern "C" int k64_run(int mode, const float* A, const float* W, float* C, int M, int N, double* ss, double* sq, int copies, void* stream, int gx_cap) {
  switch (mode) {
    case 0: return go<0>(A, W, C, M, N, ss, sq, copies, stream, gx_cap);
    case 1: return go<1>(A, W, C, M, N, ss, sq, copies, stream, gx_cap);
    case 2: return go<2>(A, W, C, M, N, ss, sq, copies, stream, gx_cap);
    case 3: return go<3>(A, W, C, M, N, ss, sq, copies, stream, gx_cap);
    default: return go<4>(A, W, C, M, N, ss, sq, copies, stream, gx_cap);
  }
}
