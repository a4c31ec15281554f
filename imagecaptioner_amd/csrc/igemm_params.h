// igemm_params.h — kernel parameter block and guarded-fetch helpers shared by the implicit-GEMM kernel families
// (igemm_f32.hip: exact fp32 MFMA; igemm_bf16.hip: bf16 / split-bf16 MFMA over the same fp32 operands in HBM).
#pragma once
#include "ick_common.h"
#include <cstdlib>

namespace ickg {

struct P {  // kernel parameters (by value)
  const float* A; const float* B; float* C;
  const float* bias; const float* residual; double* stat_sum; double* stat_sq;
  int M, N, K;
  long lda, ldb, ldc, ldr;
  int batch_inner;
  long sAo, sAi, sBo, sBi, sCo, sCi;
  int splitk, kps, accumulate, act, tiles_n;
  float alpha;
  int Nb, H, W, Cin, Ho, Wo, Cout, R, S, stride, pad;
  int stat_copies; long stat_stride;
  const float* col_scale;
  int no_ep_vec;   // IckGemm.tile bit 9: force the direct (dword) epilogue (A/B and diagnostics)
  int m_base;        // first row of this launch (M-split dispatch; 0 for a whole-problem launch)
  int chunk_tiles;   // fold the MFMA accumulators into the master sum every chunk_tiles k-tiles (0 = never)
  int ep_vec;   // LDS-staged 16-byte epilogue allowed (set by the launcher from the alignment of C / residual)
  int c16, r16;   // native 16-bit kernels only (IckGemm.io16): C / the residual hold bf16 or fp16 elements (the operand type)
  const float* a_absmax;   // TERMS 4 only: device pointer to max |A| (IckGemm.a_absmax); NULL = A is used as it is
};

// which fetch pattern each op uses for its A and B operands
__host__ __device__ constexpr bool a_kcontig(int op) { return op != ICK_OP_TN && op != ICK_OP_CONV_WGRAD; }
__host__ __device__ constexpr bool is_dgrad(int op) { return op == ICK_OP_CONV_DGRAD || op == ICK_OP_CONV_DGRAD_S2; }
__host__ __device__ constexpr bool b_kcontig(int op) {
  return op == ICK_OP_NT || op == ICK_OP_CONV_FWD || op == ICK_OP_CONV_FWD_C4;
}

// Branch-free guarded fetch: a lane whose element is out of range reads a valid dummy address and its value is
// zeroed where it is CONSUMED (the LDS stash), so the 16-byte loads of tile t+1 stay in flight across tile t's MFMAs.
// (A branchy `ok ? load : 0` makes hipcc emit s_waitcnt vmcnt(0) right behind the loads: load->compute serialised.)
__device__ __forceinline__ float4 ldg4u(const float* p, bool ok, const float* safe) {
  return *reinterpret_cast<const float4*>(ok ? p : safe);
}
__device__ __forceinline__ float4 keep_if(float4 v, bool ok) {
  return make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
}
// k-contiguous rows whose valid range ends at kend (any kend: the row pitch is a multiple of 4, so the 16-byte load
// stays inside the row): zero the components at k+1..k+3 that lie past kend
__device__ __forceinline__ float4 ktail(float4 v, int k, int kend) {
  if (k + 3 >= kend) {
    if (k + 1 >= kend) v.y = 0.f;
    if (k + 2 >= kend) v.z = 0.f;
    v.w = 0.f;
  }
  return v;
}

// native 16-bit operands (IN16 kernels): four halves = 8 bytes, same guarded branch-free form
__device__ __forceinline__ uint2 ldg4u(const unsigned short* p, bool ok, const unsigned short* safe) {
  return *reinterpret_cast<const uint2*>(ok ? p : safe);
}
__device__ __forceinline__ uint2 keep_if(uint2 v, bool ok) { return make_uint2(ok ? v.x : 0u, ok ? v.y : 0u); }
__device__ __forceinline__ uint2 ktail(uint2 v, int, int) { return v; }   // IN16 requires K % 4 == 0: no partial group

// 16-bit C / residual rows of the native 16-bit kernels: four elements = 8 bytes (H4 = bf16x4 or f16x4)
template <typename H4>
__device__ __forceinline__ float4 ld4h(const void* base, long idx) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  const f4 f = __builtin_convertvector(*reinterpret_cast<const H4*>(reinterpret_cast<const unsigned short*>(base) + idx), f4);
  return make_float4(f.x, f.y, f.z, f.w);
}
template <typename H4>
__device__ __forceinline__ void st4h(void* base, long idx, float4 v) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  const f4 f = {v.x, v.y, v.z, v.w};
  *reinterpret_cast<H4*>(reinterpret_cast<unsigned short*>(base) + idx) = __builtin_convertvector(f, H4);
}

// erf to < 1 ulp of fp32 (max |error| 7.6e-8 over [-6, 6], checked against scipy in double), BRANCH-FREE: both minimax
// branches are evaluated and selected, ~25 VALU instructions.  libm's erff carries real branches that a wave takes both ways,
// and 64 inlined copies of it per lane are what made a GELU epilogue 20 us long.
__device__ __forceinline__ float erf_fast(float a) {
  const float t = fabsf(a), s = a * a;
  float r = fmaf(-1.72853470e-5f, t, 3.83197126e-4f);
  const float u = fmaf(-3.88396438e-3f, t, 2.42546219e-2f);
  r = fmaf(r, s, u);
  r = fmaf(r, t, -1.06777877e-1f);
  r = fmaf(r, t, -6.34846687e-1f);
  r = fmaf(r, t, -1.28717512e-1f);
  r = fmaf(r, t, -t);
  const float hi = copysignf(1.0f - __expf(r), a);
  float q = -5.96761703e-4f;
  q = fmaf(q, s, 4.99119423e-3f);
  q = fmaf(q, s, -2.67681349e-2f);
  q = fmaf(q, s, 1.12819925e-1f);
  q = fmaf(q, s, -3.76125336e-1f);
  q = fmaf(q, s, 1.28379166e-1f);
  const float lo = fmaf(q, a, a);
  return t > 0.927734375f ? hi : lo;
}

// The activation as a COMPILE-TIME choice.  The epilogues are fully unrolled over the accumulator registers (16 TM TN values
// per lane); with a run-time `act` every one of those copies carried the code of all four activations behind uniform
// branches — ~350 instructions per element, 176 KB of sparsely executed code per kernel, and in-kernel time stamps showed a
// 128x128 tile spending 20-38 us in its epilogue on instruction fetch (round 3, tools/ablate/run_kloop.py stamps).
template <int ACT>
__device__ __forceinline__ float act_c(float v) {
  if constexpr (ACT == ICK_ACT_RELU) return v > 0.f ? v : 0.f;
  else if constexpr (ACT == ICK_ACT_GELU) return 0.5f * v * (1.f + erf_fast(v * 0.70710678118654752440f));
  else if constexpr (ACT == ICK_ACT_TANH) return tanhf(v);
  else return v;
}
template <int ACT> struct ActTag { static constexpr int value = ACT; };
// calls f(ActTag<act>{}): one specialised copy of the caller's unrolled loop per activation, selected by a single switch
template <typename F>
__device__ __forceinline__ void act_dispatch(int act, F&& f) {
  switch (act) {
    case ICK_ACT_RELU: f(ActTag<ICK_ACT_RELU>{}); break;
    case ICK_ACT_GELU: f(ActTag<ICK_ACT_GELU>{}); break;
    case ICK_ACT_TANH: f(ActTag<ICK_ACT_TANH>{}); break;
    default: f(ActTag<ICK_ACT_NONE>{}); break;
  }
}
// run-time form, for ROLLED loops only (one copy of the code)
__device__ __forceinline__ float act_fn(float v, int act) {
  if (act == ICK_ACT_RELU) return act_c<ICK_ACT_RELU>(v);
  if (act == ICK_ACT_GELU) return act_c<ICK_ACT_GELU>(v);
  if (act == ICK_ACT_TANH) return act_c<ICK_ACT_TANH>(v);
  return v;
}

// Copies the descriptor into the kernel parameter block and derives the split-K partition for a k-tile depth of `bk`.
// Returns 0 or a negative status with ick_last_error() set; nz = grid.z.
inline int prepare(const IckGemm* d, int bk, P& p, int& nz, const char* who) {
  ICK_REQUIRE(d != nullptr, "%s: null descriptor", who);
  ICK_REQUIRE(d->A && d->B && d->C, "%s: null operand pointer", who);
  ICK_REQUIRE(d->M > 0 && d->N > 0 && d->K > 0, "%s: empty problem M=%d N=%d K=%d", who, d->M, d->N, d->K);
  ICK_REQUIRE(ick::aligned16(d->A) && ick::aligned16(d->B), "%s: A/B must be 16-byte aligned", who);
  p = P{};
  p.A = d->A; p.B = d->B; p.C = d->C; p.bias = d->bias; p.residual = d->residual;
  p.stat_sum = d->stat_sum; p.stat_sq = d->stat_sq;
  p.col_scale = d->col_scale;
  p.a_absmax = d->a_absmax;
  p.c16 = d->io16 & 1; p.r16 = (d->io16 >> 1) & 1;
  p.stat_copies = d->stat_copies > 1 ? d->stat_copies : 1; p.stat_stride = d->stat_stride;
  ICK_REQUIRE(p.stat_copies == 1 || p.stat_stride >= d->N, "%s: stat_stride must be >= N", who);
  p.M = d->M; p.N = d->N; p.K = d->K;
  p.lda = d->lda; p.ldb = d->ldb; p.ldc = d->ldc; p.ldr = d->ldr;
  p.batch_inner = d->batch_inner > 0 ? d->batch_inner : 1;
  const int bo = d->batch_outer > 0 ? d->batch_outer : 1;
  p.sAo = d->sAo; p.sAi = d->sAi; p.sBo = d->sBo; p.sBi = d->sBi; p.sCo = d->sCo; p.sCi = d->sCi;
  p.splitk = d->splitk > 1 ? d->splitk : 1;
  p.accumulate = d->accumulate; p.act = d->act; p.alpha = d->alpha;
  p.Nb = d->Nb; p.H = d->H; p.W = d->W; p.Cin = d->Cin; p.Ho = d->Ho; p.Wo = d->Wo; p.Cout = d->Cout;
  p.R = d->R; p.S = d->S; p.stride = d->stride; p.pad = d->pad;
  nz = bo * p.batch_inner;
  {
    static const int env_chunk = [] { const char* e = getenv("ICK_KCHUNK"); return e ? atoi(e) : 0; }();   // A/B runs
    const int kc = d->kchunk != 0 ? d->kchunk : (env_chunk != 0 ? env_chunk : 64);
    p.chunk_tiles = kc > 0 ? (kc + bk - 1) / bk : 0;
  }
  ICK_REQUIRE((d->stat_sum == nullptr) == (d->stat_sq == nullptr), "%s: stat_sum and stat_sq go together", who);
  if (p.splitk > 1) {
    ICK_REQUIRE(nz == 1, "%s: split-K and batching are exclusive", who);
    ICK_REQUIRE(p.act == ICK_ACT_NONE && !p.stat_sum && !p.col_scale, "%s: split-K cannot apply an activation or statistics", who);
    const int tiles = (p.K + bk - 1) / bk;
    const int per = (tiles + p.splitk - 1) / p.splitk;
    p.kps = per * bk;
    p.splitk = (tiles + per - 1) / per;
    nz = p.splitk;
  } else {
    p.kps = p.K;
  }
  ICK_REQUIRE(nz <= 65535, "%s: grid.z %d too large", who, nz);
  return 0;
}

}  // namespace ickg
