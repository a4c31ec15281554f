// igemm_f32.hip — fp32 implicit-GEMM family on the CDNA4 matrix cores (gfx950).
//
//   C[z][m][n] = act(alpha * sum_k A(m,k) B(n,k) + bias[n]) + residual[m][n]
//
// One LDS-tiled kernel template built on v_mfma_f32_32x32x2_f32 (exact fp32: the result is
// a k-ordered fmaf chain, so parity with the reference's fp32 CPU path holds to ~1e-6) serves
// every dense contraction of the KD step: Linear fwd/bwd, batched attention products, and the
// ResNet-50 convolutions (fwd / dgrad / wgrad) as implicit GEMM over NHWC activations.
//
// Tiling (64-wide wavefronts): 256 threads = 4 waves in a 2x2 grid; block tile BMxBNx16;
// each wave owns (BM/2)x(BN/2) = TMxTN MFMA tiles of 32x32 (16 accumulator VGPRs each).
// Operand tiles live in LDS k-major, As[k][m] / Bs[k][n], so the MFMA fragment read
// (lane l: A[m = l&31][k = l>>5]) is one conflict-free ds_read_b32 per tile per k-pair.
//   * a source that is contiguous along K (row-major activations, conv im2col over NHWC
//     channels, [N][K] weights) is fetched as float4 along k and TRANSPOSED on the LDS write
//     (4 x ds_write_b32; row pitch = BX+2 makes the 32-lane write group conflict-free);
//   * a source that is contiguous along M/N ([K][N] weights for dgrad, dY^T for wgrad) is
//     fetched as float4 along x and stored with one ds_write_b128 (row pitch BX+4).
// Global->LDS is software pipelined through registers (issue tile t+1's loads, run tile t's
// MFMAs, then write t+1 into the other LDS buffer): one barrier per k-tile.
#include "igemm_params.h"
#include <cstdlib>
#include <type_traits>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BK = 16;
constexpr int NT = 256;
#ifndef ICK_STASH_AT
#define ICK_STASH_AT 10
#endif
constexpr int STASH_AT = ICK_STASH_AT;   // k offset inside a tile after which the next tile is written to LDS

using namespace ickg;

template <int OP, int BM, int BN>
__global__ __launch_bounds__(NT, 2) void igemm_f32_kernel(const P p) {
  constexpr bool AK = a_kcontig(OP), BKc = b_kcontig(OP);
  constexpr int BMP = BM + (AK ? 2 : 4), BNP = BN + (BKc ? 2 : 4);
  constexpr int WM = BM / 2, WN = BN / 2, TM = WM / 32, TN = WN / 32;
  // k-contiguous fetch: 4 threads per row, 64 rows per pass; x-contiguous: BX/4 threads per k-row
  constexpr int PA = AK ? BM / 64 : (BK * BM) / 1024;
  constexpr int PB = BKc ? BN / 64 : (BK * BN) / 1024;
  constexpr int A_TPK = BM / 4, B_TPK = BN / 4;        // threads per k-row (x-contiguous)
  constexpr int A_KR = NT / A_TPK, B_KR = NT / B_TPK;  // k-rows per pass   (x-contiguous)

  __shared__ __attribute__((aligned(16))) float As[2][BK][BMP];
  __shared__ __attribute__((aligned(16))) float Bs[2][BK][BNP];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  // XCD-aware tile order: the dispatcher deals workgroup ids round-robin over the 8 XCDs (each with its own L2);
  // remap so that every XCD walks a CONTIGUOUS range of tiles (row-major over N then M): the A row panel
  // (activations, the large operand) is then fetched into one L2 instead of up to eight (bijective for any grid).
  const int nwg = gridDim.x, q8 = nwg >> 3, r8 = nwg & 7, xcd = blockIdx.x & 7;
  const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (blockIdx.x >> 3);
  const int tile_m = wg / p.tiles_n, tile_n = wg - tile_m * p.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  // batch / split-K decomposition of blockIdx.z
  int z = blockIdx.z, split = 0;
  if (p.splitk > 1) { split = z; z = 0; }
  // stride-2 dgrad: blockIdx.z = parity class (py,px) of the input pixel; only the taps r = r0, r0+2, .. with
  // (iy + pad - r) even can reach it, so each class is a dense GEMM over its own nr*ns taps (no multiplies by zero)
  int py = 0, px = 0, r0 = 0, s0 = 0, ns = 1, kcls = 0;
  if constexpr (OP == ICK_OP_CONV_DGRAD_S2) {
    // classes in DESCENDING order of work (workgroups are issued in blockIdx.z order): for a 3x3 / pad 1 kernel class (1,1)
    // reaches 4 taps and (0,0) one — dispatched last the heavy class was the launch's tail (253 -> 218 us on layer4's) — while a
    // 1x1 kernel has all its taps in class (0,0), which must stay first so that the other classes' zero-fill overlaps it
    {
      auto taps = [&](int par, int R) { const int q = (par + p.pad) & 1; return R > q ? (R - q + 1) / 2 : 0; };
      const int hy = taps(1, p.R) > taps(0, p.R) ? 1 : 0, hx = taps(1, p.S) > taps(0, p.S) ? 1 : 0;
      py = (z >> 1) ^ hy; px = (z & 1) ^ hx; z = 0;
    }
    r0 = (py + p.pad) & 1; s0 = (px + p.pad) & 1;
    const int nr = p.R > r0 ? (p.R - r0 + 1) / 2 : 0;
    ns = p.S > s0 ? (p.S - s0 + 1) / 2 : 0;
    kcls = nr * ns * p.Cout;
  }
  const int zo = z / p.batch_inner, zi = z - zo * p.batch_inner;
  const float* __restrict__ Ag = p.A + zo * p.sAo + zi * p.sAi;
  const float* __restrict__ Bg = p.B + zo * p.sBo + zi * p.sBi;
  const long coff = zo * p.sCo + zi * p.sCi;
  const int kbeg = split * p.kps;
  const int kend = OP == ICK_OP_CONV_DGRAD_S2 ? kcls : min(p.K, kbeg + p.kps);
  const int nkt = (kend - kbeg + BK - 1) / BK;

  // ---------------------------------------------------------------- per-thread fetch state
  // A side
  const float* a_ptr[PA]; bool a_ok[PA]; int a_y[PA], a_x[PA];
  const int a_k4 = (tid & 3) * 4;  // k offset inside the tile (k-contiguous fetch)
#pragma unroll
  for (int i = 0; i < PA; ++i) {
    if constexpr (AK) {
      const int m = m0 + i * 64 + (tid >> 2);
      a_ok[i] = m < p.M;
      if constexpr (OP == ICK_OP_NT || OP == ICK_OP_NN) {
        a_ptr[i] = Ag + (long)m * p.lda + a_k4; a_y[i] = a_x[i] = 0;
      } else if constexpr (OP == ICK_OP_CONV_FWD || OP == ICK_OP_CONV_FWD_C4) {
        const int hw = p.Ho * p.Wo; const int b = m / hw; const int r = m - b * hw;
        const int oy = r / p.Wo, ox = r - oy * p.Wo;
        a_y[i] = oy * p.stride - p.pad; a_x[i] = ox * p.stride - p.pad;
        a_ptr[i] = Ag + (long)b * p.H * p.W * p.Cin;
      } else if constexpr (OP == ICK_OP_CONV_DGRAD_S2) {  // rows = input pixels of this parity class
        const int w2 = p.W >> 1; const int hw = (p.H >> 1) * w2; const int b = m / hw; const int r = m - b * hw;
        const int iy = 2 * (r / w2) + py, ix = 2 * (r % w2) + px;
        a_y[i] = iy + p.pad; a_x[i] = ix + p.pad;
        a_ptr[i] = Ag + (long)b * p.Ho * p.Wo * p.Cout;
      } else {  // CONV_DGRAD: rows are input pixels, gather from dY
        const int hw = p.H * p.W; const int b = m / hw; const int r = m - b * hw;
        const int iy = r / p.W, ix = r - iy * p.W;
        a_y[i] = iy + p.pad; a_x[i] = ix + p.pad;
        a_ptr[i] = Ag + (long)b * p.Ho * p.Wo * p.Cout;
      }
    } else {  // x-contiguous: A stored [K][M]
      const int m = m0 + (tid % A_TPK) * 4;
      a_ok[i] = m < p.M;
      a_ptr[i] = Ag + m; a_y[i] = i * A_KR + tid / A_TPK; a_x[i] = 0;
    }
  }
  // B side
  const float* b_ptr[PB]; bool b_ok[PB]; int b_y[PB], b_x[PB];
  int b_r = 0, b_s = 0;
#pragma unroll
  for (int i = 0; i < PB; ++i) {
    if constexpr (BKc) {
      const int n = n0 + i * 64 + (tid >> 2);
      b_ok[i] = n < p.N;
      b_ptr[i] = Bg + (long)n * p.ldb + a_k4; b_y[i] = b_x[i] = 0;
    } else {
      const int n = n0 + (tid % B_TPK) * 4;
      b_ok[i] = n < p.N;
      b_y[i] = i * B_KR + tid / B_TPK; b_x[i] = 0;
      if constexpr (OP == ICK_OP_CONV_WGRAD) {
        const int tap = n / p.Cin; const int ci = n - tap * p.Cin;
        b_r = tap / p.S; b_s = tap - b_r * p.S;
        b_ptr[i] = Bg + ci;
      } else {
        b_ptr[i] = Bg + n;
      }
    }
  }

  float4 ra[PA], rb[PB];
  unsigned amask = 0, bmask = 0;   // bit i: fetch i of the tile in flight is in range

  auto fetch = [&](int kt) {
    const int k0 = kbeg + kt * BK;
    amask = 0; bmask = 0;
    // ---- A
    if constexpr (OP == ICK_OP_NT || OP == ICK_OP_NN) {
#pragma unroll
      for (int i = 0; i < PA; ++i) {
        const bool ok = a_ok[i] && (k0 + a_k4 < kend);
        amask |= ok ? (1u << i) : 0u;
        ra[i] = ldg4u(a_ptr[i] + k0, ok, Ag);
      }
    } else if constexpr (OP == ICK_OP_CONV_FWD) {
      const int tap = k0 / p.Cin; const int ci = k0 - tap * p.Cin + a_k4;
      const int r = tap / p.S, s = tap - r * p.S;
#pragma unroll
      for (int i = 0; i < PA; ++i) {
        const int iy = a_y[i] + r, ix = a_x[i] + s;
        const bool ok = a_ok[i] && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W && (k0 + a_k4 < kend);
        amask |= ok ? (1u << i) : 0u;
        ra[i] = ldg4u(a_ptr[i] + ((long)iy * p.W + ix) * p.Cin + ci, ok, Ag);
      }
    } else if constexpr (OP == ICK_OP_CONV_FWD_C4) {
      const int tap = (k0 + a_k4) >> 2; const int r = tap / p.S, s = tap - r * p.S;
#pragma unroll
      for (int i = 0; i < PA; ++i) {
        const int iy = a_y[i] + r, ix = a_x[i] + s;
        const bool ok = a_ok[i] && r < p.R && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        amask |= ok ? (1u << i) : 0u;
        ra[i] = ldg4u(a_ptr[i] + ((long)iy * p.W + ix) * 4, ok, Ag);
      }
    } else if constexpr (OP == ICK_OP_CONV_DGRAD) {
      const int tap = k0 / p.Cout; const int co = k0 - tap * p.Cout + a_k4;
      const int r = tap / p.S, s = tap - r * p.S;
#pragma unroll
      for (int i = 0; i < PA; ++i) {
        const int ty = a_y[i] - r, tx = a_x[i] - s;
        const int oy = ty / p.stride, ox = tx / p.stride;
        const bool ok = a_ok[i] && ty >= 0 && tx >= 0 && oy * p.stride == ty && ox * p.stride == tx &&
                        oy < p.Ho && ox < p.Wo && (k0 + a_k4 < kend);
        amask |= ok ? (1u << i) : 0u;
        ra[i] = ldg4u(a_ptr[i] + ((long)oy * p.Wo + ox) * p.Cout + co, ok, Ag);
      }
    } else if constexpr (OP == ICK_OP_CONV_DGRAD_S2) {
      const int q = k0 / p.Cout; const int co = k0 - q * p.Cout + a_k4;
      const int r = r0 + 2 * (q / ns), s = s0 + 2 * (q % ns);
#pragma unroll
      for (int i = 0; i < PA; ++i) {
        const int ty = a_y[i] - r, tx = a_x[i] - s;        // even by construction
        const int oy = ty >> 1, ox = tx >> 1;
        const bool ok = a_ok[i] && ty >= 0 && tx >= 0 && oy < p.Ho && ox < p.Wo && (k0 + a_k4 < kend);
        amask |= ok ? (1u << i) : 0u;
        ra[i] = ldg4u(a_ptr[i] + ((long)oy * p.Wo + ox) * p.Cout + co, ok, Ag);
      }
    } else {  // A [K][M]
#pragma unroll
      for (int i = 0; i < PA; ++i) {
        const int k = k0 + a_y[i];
        const bool ok = a_ok[i] && k < kend;
        amask |= ok ? (1u << i) : 0u;
        ra[i] = ldg4u(a_ptr[i] + (long)k * p.lda, ok, Ag);
      }
    }
    // ---- B
    if constexpr (BKc) {
#pragma unroll
      for (int i = 0; i < PB; ++i) {
        const bool ok = b_ok[i] && (k0 + a_k4 < kend);
        bmask |= ok ? (1u << i) : 0u;
        rb[i] = ldg4u(b_ptr[i] + k0, ok, Bg);
      }
    } else if constexpr (OP == ICK_OP_CONV_DGRAD) {
      const int tap = k0 / p.Cout; const int co0 = k0 - tap * p.Cout;
#pragma unroll
      for (int i = 0; i < PB; ++i) {
        const int co = co0 + b_y[i];
        const bool ok = b_ok[i] && (k0 + b_y[i] < kend);
        bmask |= ok ? (1u << i) : 0u;
        rb[i] = ldg4u(b_ptr[i] + ((long)co * p.R * p.S + tap) * p.Cin, ok, Bg);
      }
    } else if constexpr (OP == ICK_OP_CONV_DGRAD_S2) {
      const int q = k0 / p.Cout; const int co0 = k0 - q * p.Cout;
      const int tap = (r0 + 2 * (q / ns)) * p.S + s0 + 2 * (q % ns);
#pragma unroll
      for (int i = 0; i < PB; ++i) {
        const int co = co0 + b_y[i];
        const bool ok = b_ok[i] && (k0 + b_y[i] < kend);
        bmask |= ok ? (1u << i) : 0u;
        rb[i] = ldg4u(b_ptr[i] + ((long)co * p.R * p.S + tap) * p.Cin, ok, Bg);
      }
    } else if constexpr (OP == ICK_OP_CONV_WGRAD) {
      const int hw = p.Ho * p.Wo;
#pragma unroll
      for (int i = 0; i < PB; ++i) {
        const int k = k0 + b_y[i];
        const int b = k / hw; const int rem = k - b * hw; const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
        const int iy = oy * p.stride - p.pad + b_r, ix = ox * p.stride - p.pad + b_s;
        const bool ok = b_ok[i] && k < kend && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        bmask |= ok ? (1u << i) : 0u;
        rb[i] = ldg4u(b_ptr[i] + (((long)b * p.H + iy) * p.W + ix) * p.Cin, ok, Bg);
      }
    } else {  // B [K][N]
#pragma unroll
      for (int i = 0; i < PB; ++i) {
        const int k = k0 + b_y[i];
        const bool ok = b_ok[i] && k < kend;
        bmask |= ok ? (1u << i) : 0u;
        rb[i] = ldg4u(b_ptr[i] + (long)k * p.ldb, ok, Bg);
      }
    }
  };

  // write the fetched tile `kt` into LDS buffer `buf` (zeroing what was out of range)
  auto stash = [&](int buf, int kt) {
    const int kq = kbeg + kt * BK + a_k4;
#pragma unroll
    for (int i = 0; i < PA; ++i) {
      float4 v = keep_if(ra[i], (amask >> i) & 1u);
      if constexpr (AK) {
        if constexpr (OP == ICK_OP_NT || OP == ICK_OP_NN) v = ktail(v, kq, kend);
        const int x = i * 64 + (tid >> 2);
        As[buf][a_k4 + 0][x] = v.x; As[buf][a_k4 + 1][x] = v.y;
        As[buf][a_k4 + 2][x] = v.z; As[buf][a_k4 + 3][x] = v.w;
      } else {
        *reinterpret_cast<float4*>(&As[buf][i * A_KR + tid / A_TPK][(tid % A_TPK) * 4]) = v;
      }
    }
#pragma unroll
    for (int i = 0; i < PB; ++i) {
      float4 v = keep_if(rb[i], (bmask >> i) & 1u);
      if constexpr (BKc) {
        v = ktail(v, kq, kend);
        const int x = i * 64 + (tid >> 2);
        Bs[buf][a_k4 + 0][x] = v.x; Bs[buf][a_k4 + 1][x] = v.y;
        Bs[buf][a_k4 + 2][x] = v.z; Bs[buf][a_k4 + 3][x] = v.w;
      } else {
        *reinterpret_cast<float4*>(&Bs[buf][i * B_KR + tid / B_TPK][(tid % B_TPK) * 4]) = v;
      }
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  f32x16 tot[TM][TN];      // chunked accumulation (see igemm_f32_glds.hip): master sum of the folded MFMA chains
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) tot[i][j][r] = 0.f;
  int chain = 0;

  if (nkt > 0) {
    fetch(0);
    stash(0, 0);
  }
  __syncthreads();

  const int fa = wm * WM + (lane & 31), fb = wn * WN + (lane & 31), fk = lane >> 5;
  for (int kt = 0; kt < nkt; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nkt) fetch(kt + 1);
    // fragment registers are double-buffered by hand: the ds_reads of k-pair kk+2 are issued BEFORE the MFMAs of
    // k-pair kk, so their LDS latency hides behind 4 x 64 MFMA cycles even with a single wave on the SIMD
    float av[2][TM], bv[2][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) av[0][i] = As[buf][fk][fa + i * 32];
#pragma unroll
    for (int j = 0; j < TN; ++j) bv[0][j] = Bs[buf][fk][fb + j * 32];
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
      const int cur = (kk >> 1) & 1, nxt = cur ^ 1;
      if (kk + 2 < BK) {
#pragma unroll
        for (int i = 0; i < TM; ++i) av[nxt][i] = As[buf][kk + 2 + fk][fa + i * 32];
#pragma unroll
        for (int j = 0; j < TN; ++j) bv[nxt][j] = Bs[buf][kk + 2 + fk][fb + j * 32];
      }
      // the LDS write of tile kt+1 (other buffer: nobody reads it during tile kt) goes in the shadow of the
      // second half of this tile's MFMAs instead of between the last MFMA and the barrier
      if (kk == STASH_AT && kt + 1 < nkt) stash(buf ^ 1, kt + 1);
      // scheduling fence: keeps "next fragments' LDS reads" ahead of this k-pair's MFMAs (hipcc otherwise sinks the
      // reads behind the MFMAs and waits lgkmcnt(0) in front of every group)
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cur][i], bv[cur][j], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (++chain == p.chunk_tiles && kt + 1 < nkt) {
      chain = 0;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          tot[i][j] += acc[i][j];
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        }
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] += tot[i][j];

  // ---------------------------------------------------------------- epilogue
  // C/D layout of the 32x32 tile: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
  // Interior tiles (the vast majority) take a check-free path; flags are wave-uniform scalars.
  float* __restrict__ Cg = p.C + coff;
  const float* __restrict__ Rg = (p.residual && split == 0) ? p.residual + coff : nullptr;
  const float* __restrict__ biasp = (p.bias && split == 0) ? p.bias : nullptr;
  const int mode = p.splitk > 1 ? 2 : (p.accumulate ? 1 : 0);
  const int act = p.act & 15;
  const float alpha = p.alpha;
  auto epilogue = [&](auto full_tag, auto act_tag) {
    constexpr bool FULL = decltype(full_tag)::value;
    constexpr int ACT = decltype(act_tag)::value;     // compile-time activation: see igemm_params.h act_c
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + wn * WN + j * 32 + (lane & 31);
      const bool nok = FULL || n < p.N;
      const float bias = (biasp && nok) ? biasp[n] : 0.f;
      float ssum = 0.f, ssq = 0.f;   // ssq by explicit fmaf: the epilogue variants must agree bit for bit
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int mb = m0 + wm * WM + i * 32 + 4 * (lane >> 5);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = mb + (r & 3) + 8 * (r >> 2);
          if (FULL || (m < p.M && nok)) {
            float v = acc[i][j][r] * alpha;
            ssum += v; ssq = fmaf(v, v, ssq);
            if constexpr (ACT < 0) v = act_fn(v + bias, act); else v = act_c<ACT>(v + bias);
            long mr = m;
            if constexpr (OP == ICK_OP_CONV_DGRAD_S2) {   // class row -> pixel row of the full-resolution dX
              const int w2 = p.W >> 1; const int hw = (p.H >> 1) * w2; const int b = m / hw; const int q = m - b * hw;
              mr = ((long)b * p.H + 2 * (q / w2) + py) * p.W + 2 * (q % w2) + px;
            }
            const long o = mr * p.ldc + n;
            if (Rg) v += Rg[mr * p.ldr + n];
            if (mode == 2) atomicAdd(Cg + o, v);
            else if (mode == 1) Cg[o] += v;
            else Cg[o] = v;
          }
        }
      }
      if (p.stat_sum) {  // BatchNorm batch statistics of the raw product; fp64 so that E[x^2]-E[x]^2 cannot cancel
        ssum += __shfl_xor(ssum, 32);
        ssq += __shfl_xor(ssq, 32);
        if (lane < 32 && nok) {
          const long so = (long)(tile_m % p.stat_copies) * p.stat_stride + n;   // copy of this row tile
          atomicAdd(p.stat_sum + so, (double)ssum); atomicAdd(p.stat_sq + so, (double)ssq);
        }
      }
    }
  };
  if (act == ICK_ACT_NONE) {
    if (m0 + BM <= p.M && n0 + BN <= p.N) epilogue(std::true_type{}, ActTag<ICK_ACT_NONE>{});
    else epilogue(std::false_type{}, ActTag<ICK_ACT_NONE>{});
  } else {
    epilogue(std::false_type{}, ActTag<-1>{});       // -1: the run-time form (rare on this path; saves three unrolled copies)
  }
}

template <int OP, int BM, int BN>
int launch(const P& p0, int nz, hipStream_t st) {
  P p = p0;
  p.tiles_n = (p.N + BN - 1) / BN;
  dim3 grid(p.tiles_n * ((p.M + BM - 1) / BM), 1, nz);
  ICK_LAUNCH((igemm_f32_kernel<OP, BM, BN>), grid, dim3(NT), 0, st, p);
  return ick::launch_status("igemm_f32");
}

// Tile choice.  fp32 MFMA is slow enough (64 cycles per 32x32x2) that all four tile shapes keep the matrix pipe
// fed; what separates them on the step's shapes is WAVE QUANTISATION over the 256 CUs: a grid of T workgroups
// with R resident per CU finishes in ceil(T / 256 / R) * R "slots" of unequal value.  Cost model: every
// workgroup costs its MFMA work (tile area) plus a fixed per-k-tile overhead that weighs more on small tiles;
// the grid costs max-per-CU work.  tile: 0 = model, 1 = 128x128, 2 = 64x64, 3 = 128x64, 4 = 64x128.
template <int OP>
int dispatch_tile(const P& p, int nz, hipStream_t st, int tile) {
  tile &= 15;    // the +16 (three LDS buffers) / +32 (M-split) variants exist in the LDS-DMA kernel only
  if (tile == 0) {
    static const int bm[4] = {128, 64, 128, 64}, bn[4] = {128, 64, 64, 128};
    static const double eff[4] = {1.00, 0.80, 0.90, 0.90};   // relative MFMA efficiency of the tile shape
    double best = 1e300;
    for (int t = 0; t < 4; ++t) {
      const long blocks = (long)((p.M + bm[t] - 1) / bm[t]) * ((p.N + bn[t] - 1) / bn[t]) * nz;
      const long per_cu = (blocks + 255) / 256;               // workgroups the busiest CU executes
      const double cost = (double)per_cu * bm[t] * bn[t] / eff[t];
      if (cost < best * 0.999) { best = cost; tile = t + 1; }
    }
  }
  switch (tile) {
    case 2: return launch<OP, 64, 64>(p, nz, st);
    case 3: return launch<OP, 128, 64>(p, nz, st);
    case 4: return launch<OP, 64, 128>(p, nz, st);
    default: return launch<OP, 128, 128>(p, nz, st);
  }
}

}  // namespace

namespace ickg {
bool glds_eligible(const IckGemm* d);                                   // igemm_f32_glds.hip
int run_glds(const IckGemm* d, const P& p, int nz, hipStream_t st);
}

// the LDS-DMA kernel (igemm_f32_glds.hip) is the default; IckGemm.tile bit 8 or ICK_NO_GLDS=1 selects this file's
// register-staged kernel (kept for shapes whose 16-byte chunks would straddle the K end or a filter tap, and for A/B runs)
static bool want_glds(const IckGemm* d) {
  static const bool off = [] { const char* e = getenv("ICK_NO_GLDS"); return e && e[0] == '1'; }();
  return !off && !(d->tile & 256) && glds_eligible(d);
}

extern "C" int ick_gemm_f32(const IckGemm* d0, void* stream) {
  ICK_REQUIRE(d0 != nullptr, "ick_gemm_f32: null descriptor");
  ICK_REQUIRE(d0->io16 == 0, "ick_gemm_f32: C and the residual are fp32 here (io16 belongs to ick_gemm_h16 / ick_gemm_bf16)");
  const bool fused_bn = d0->col_scale || (d0->act & ICK_ACT_POST_RESIDUAL);   // only the LDS-DMA kernel implements these
  const bool glds = want_glds(d0) || (fused_bn && glds_eligible(d0));
  ICK_REQUIRE(glds || (!d0->col_scale && !(d0->act & ICK_ACT_POST_RESIDUAL)),
              "ick_gemm_f32: col_scale / ICK_ACT_POST_RESIDUAL are implemented by the LDS-DMA kernel only (K %% 4, Cin %% 32)");
  IckGemm dd = *d0; dd.tile &= 255;
  const int no_ep_vec = (d0->tile >> 9) & 1;
  const IckGemm* d = &dd;
  P p; int nz = 1;
  if (int rc = prepare(d, glds ? 32 : BK, p, nz, "ick_gemm_f32")) return rc;
  p.no_ep_vec = no_ep_vec;
  hipStream_t st = static_cast<hipStream_t>(stream);
  switch (d->op) {
    case ICK_OP_NT:
      ICK_REQUIRE(p.lda % 4 == 0 && p.ldb % 4 == 0, "NT: lda, ldb must be multiples of 4 (rows readable up to roundup4(K))");
      ICK_REQUIRE((p.sAo | p.sAi | p.sBo | p.sBi) % 4 == 0, "NT: batch strides must be multiples of 4");
      return glds ? run_glds(d, p, nz, st) : dispatch_tile<ICK_OP_NT>(p, nz, st, d->tile);
    case ICK_OP_NN:
      ICK_REQUIRE(p.lda % 4 == 0 && p.ldb % 4 == 0, "NN: lda, ldb must be multiples of 4 (rows readable up to roundup4)");
      ICK_REQUIRE((p.sAo | p.sAi | p.sBo | p.sBi) % 4 == 0, "NN: batch strides must be multiples of 4");
      return glds ? run_glds(d, p, nz, st) : dispatch_tile<ICK_OP_NN>(p, nz, st, d->tile);
    case ICK_OP_TN:
      ICK_REQUIRE(p.lda % 4 == 0 && p.ldb % 4 == 0, "TN: lda, ldb must be multiples of 4 (rows readable up to roundup4)");
      ICK_REQUIRE((p.sAo | p.sAi | p.sBo | p.sBi) % 4 == 0, "TN: batch strides must be multiples of 4");
      return glds ? run_glds(d, p, nz, st) : dispatch_tile<ICK_OP_TN>(p, nz, st, d->tile);
    case ICK_OP_CONV_FWD:
      ICK_REQUIRE(p.Cin % 16 == 0, "CONV_FWD: Cin=%d must be a multiple of 16", p.Cin);
      ICK_REQUIRE(p.M == p.Nb * p.Ho * p.Wo && p.N == p.Cout && p.K == p.R * p.S * p.Cin && p.ldb == p.K,
                  "CONV_FWD: M/N/K do not match the geometry");
      return glds ? run_glds(d, p, nz, st) : dispatch_tile<ICK_OP_CONV_FWD>(p, nz, st, d->tile);
    case ICK_OP_CONV_FWD_C4:
      ICK_REQUIRE(p.Cin == 4, "CONV_FWD_C4: Cin must be 4");
      ICK_REQUIRE(p.M == p.Nb * p.Ho * p.Wo && p.N == p.Cout && p.K == p.R * p.S * 4 && p.ldb == p.K,
                  "CONV_FWD_C4: M/N/K do not match the geometry");
      return glds ? run_glds(d, p, nz, st) : dispatch_tile<ICK_OP_CONV_FWD_C4>(p, nz, st, d->tile);
    case ICK_OP_CONV_DGRAD:
      ICK_REQUIRE(p.Cout % 16 == 0 && p.Cin % 4 == 0, "CONV_DGRAD: Cout %% 16 and Cin %% 4 required");
      ICK_REQUIRE(p.M == p.Nb * p.H * p.W && p.N == p.Cin && p.K == p.R * p.S * p.Cout,
                  "CONV_DGRAD: M/N/K do not match the geometry");
      return glds ? run_glds(d, p, nz, st) : dispatch_tile<ICK_OP_CONV_DGRAD>(p, nz, st, d->tile);
    case ICK_OP_CONV_DGRAD_S2:
      ICK_REQUIRE(p.stride == 2 && p.H % 2 == 0 && p.W % 2 == 0, "CONV_DGRAD_S2: stride 2 and even H, W required");
      ICK_REQUIRE(p.Cout % 16 == 0 && p.Cin % 4 == 0, "CONV_DGRAD_S2: Cout %% 16 and Cin %% 4 required");
      ICK_REQUIRE(p.M == p.Nb * (p.H / 2) * (p.W / 2) && p.N == p.Cin && p.K == p.R * p.S * p.Cout && nz == 1 &&
                  p.splitk == 1, "CONV_DGRAD_S2: M must be the rows of ONE parity class; no batching / split-K");
      return glds ? run_glds(d, p, 4, st) : dispatch_tile<ICK_OP_CONV_DGRAD_S2>(p, 4, st, d->tile);
    case ICK_OP_CONV_WGRAD:
      ICK_REQUIRE(p.Cout % 4 == 0 && p.Cin % 4 == 0, "CONV_WGRAD: Cout %% 4 and Cin %% 4 required");
      ICK_REQUIRE(p.M == p.Cout && p.N == p.R * p.S * p.Cin && p.K == p.Nb * p.Ho * p.Wo && p.lda == p.Cout,
                  "CONV_WGRAD: M/N/K do not match the geometry");
      return glds ? run_glds(d, p, nz, st) : dispatch_tile<ICK_OP_CONV_WGRAD>(p, nz, st, d->tile);
    default:
      return ick::fail(-1, "ick_gemm_f32: unknown op %d", d->op);
  }
}
