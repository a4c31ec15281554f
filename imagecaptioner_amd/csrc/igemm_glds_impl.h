// igemm_glds_impl.h — the MFMA implicit-GEMM family with DIRECT global->LDS staging (LDS-DMA, global_load_lds_dwordx4),
// ONE source for all arithmetic variants (template parameter TERMS; one translation unit per value, see the *_glds*.hip
// stubs): addressing, the DMA ring, the zero page, the LDS-staged epilogue, the chunked accumulation and the M-split
// dispatch are shared; only the fragment reads and the MFMA differ.
//   TERMS 0  exact fp32       v_mfma_f32_32x32x2_f32: one ds_read_b128 per 32x32 tile per 8 k (lane l reads
//                              X[row l&31][8j + 4(l>>5) + 0..3]; MFMA e of the group multiplies k-slots {8j+e, 8j+4+e} —
//                              the SAME pairing on both operands, so the sum is unchanged)
//   TERMS 1  bf16             v_mfma_f32_32x32x16_bf16: two ds_read_b128 = 8 consecutive k per lane, rounded to bf16 IN
//   TERMS 2  fp16             v_mfma_f32_32x32x16_f16   REGISTERS (the LDS images stay fp32: operands are fp32 in HBM)
//   TERMS 3  split bf16       hi + lo parts in registers, hi*hi + hi*lo + lo*hi (~1e-5 of the fp32 product)
//   TERMS 4  fp32 BY THREE fp16 PRODUCTS ("f32x3"; round 3, NT / CONV_FWD only): a = hi + 2^-11 lo', hi = fp16(a),
//                              lo' = fp16(2^11 (a - hi)) — 22-23 of a's 24 significand bits; hi*hi goes to one accumulator,
//                              hi*lo' + lo'*hi to a second one that is folded in with 2^-11 after the k-loop (the scaling
//                              keeps lo' out of fp16's subnormal range; only lo*lo, <= 2^-22 of the product, is dropped).
//                              3 MFMAs of 32 cycles per 16 k against 8 of 64 for TERMS 0; measured error against float64
//                              on the ViT shapes = that of TERMS 0 (tests/test_gemm_gpu.py).  Operands must be < 65504 in
//                              magnitude (fp16's range: larger values give inf — loud, not silent).
//   TERMS 5 / 6  NATIVE bf16 / fp16 operands: A and B are 16-bit in HBM and in LDS.  The kernel addresses them in
//                              "units" of two halves (the host halves K, lda, ldb, Cin), so every line of addressing / DMA /
//                              swizzle code is the fp32 one; the fp32-style fragment (ds_read_b128 = 4 units) IS the 8
//                              consecutive k a lane feeds to v_mfma_f32_32x32x16_*: one MFMA per fragment, no conversion,
//                              half the L2->LDS stream of TERMS 1 / 2.  k-contiguous operands only (NT, CONV_FWD).
//
// Why LDS-DMA: an ablation of the register-staged kernel (igemm_f32.hip; tools/ablate, profiles/r01g_ablation_f32.log) on
// MI355X shows the MFMA loop alone runs at 142 TF (90 % of the fp32 MFMA peak) but the full kernel at 102 TF: issuing the
// global loads + exposed load latency cost 18-25 % and the register->LDS write pass (ds_write, zero-masking VALU) another
// 10-15 %.  LDS-DMA removes both: no staging VGPRs, no ds_write instructions, and a k-tile twice as deep (32) so that
// one tile of prefetch covers the load latency.  At 16-bit MFMA rates a k-tile is 256 matrix-pipe cycles per wave while
// a global load takes ~2 us: the DMA ring keeps 1-2 whole tiles in flight per workgroup (385 -> 417 TF at 4096^3).
//
//   C[z][m][n] = act(alpha * sum_k A(m,k) B(n,k) + bias[n]) + residual[m][n]        (same contract as igemm_f32.hip)
//
// LDS images (one LDS-DMA wave-instruction writes 64 lanes x 16 B = 1 KiB, lane-linear):
//   * k-contiguous source: image [x][32 k] (128-B rows, 8 sixteen-byte chunks), chunk position XOR-swizzled by
//     (row>>1)&7 THROUGH THE SOURCE ADDRESS (lane l of the DMA fetches the chunk that belongs at its linear slot).
//     With the swizzle the four 16-lane groups of ds_read_b128 each touch 16 distinct 16-B slots: conflict-free.
//   * x-contiguous source: image [32 k][x] (lane-linear as is), fragment = ds_read_b32 per k (32 consecutive floats).
// Out-of-range elements (conv padding, M/N/K tails) are fetched from a 16-byte zero page instead of being masked
// afterwards.  Two LDS buffers, one barrier per k-tile:  wait own DMA -> barrier -> issue DMA of tile t+1 -> MFMAs of t.
#pragma once
#include "igemm_params.h"
#include <cstdlib>
#include <type_traits>
#ifndef ICK_ABL
#define ICK_ABL 0   // tools/ablate builds set 1..3 to price the epilogue and the DMA stream (never in libick.so)
#endif

namespace {

using namespace ickg;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
template <int TERMS> struct Half16 { using x8 = bf16x8; using x4 = bf16x4; };
template <> struct Half16<4> { using x8 = f16x8; using x4 = f16x4; };
template <> struct Half16<2> { using x8 = f16x8; using x4 = f16x4; };   // fp16: 10 mantissa bits, 5-bit exponent — run under the device GradScaler
template <> struct Half16<6> { using x8 = f16x8; using x4 = f16x4; };
template <int TERMS, typename V>
__device__ __forceinline__ f32x16 mfma16(V a, V b, f32x16 c) {
  if constexpr (TERMS == 2 || TERMS == 4 || TERMS == 6) return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

constexpr int BK = 32;
constexpr int kCUs = 256;            // MI355X

// Experiment knobs of the k-loop (tools/ablate/run_kloop.py builds one library per value; libick.so uses ICK_EXP_DEFAULT):
//   1  the zero page's address is pinned in a VGPR pair (hipcc otherwise re-loads it from the GOT with s_load + lgkmcnt(0) in EVERY k-tile)
//   2  the LDS-DMA of the next k-tile is issued in PIECES between the MFMA blocks of the first half of the k-tile instead of in
//      one burst behind the barrier (the burst holds the wave's instruction stream for ~50 cycles per piece with the matrix pipe idle)
//   4  s_setprio 1 across the MFMA blocks (the co-resident workgroup's address arithmetic yields to a wave that has MFMAs to issue)
//   8  s_memtime stamps per workgroup into g_dbg (diagnostic builds only)
// libick.so is built with knobs 1 + 4 (measured +1..3 % on the step's shapes, +5 % at 4096^3; profiles/r03_kloop_knobs.log);
// 2 / 8 / 32 / 64 exist for tools/ablate only.
#ifndef ICK_EXP
#define ICK_EXP 5
#endif

__device__ __attribute__((aligned(16))) float g_zero16[4];   // the zero page
#if (ICK_EXP & 32)
__device__ int g_stagger = 0;            // experiment: initial delay (units of 64 shader cycles) of the workgroups in odd wave slots
#endif
#if (ICK_EXP & 8)
__device__ long long* g_dbg = nullptr;   // diagnostic builds: 16 words per workgroup (phase stamps, HW_ID, XCC_ID; 8..15: one k-tile's s_memtime stamps)
#endif
const bool g_no_vec_epilogue = [] { const char* e = getenv("ICK_NO_VEC_EPILOGUE"); return e && e[0] == '1'; }();   // A/B runs

// One LDS-DMA wave-instruction: lane l's 16 bytes at `src` land at LDS byte address lds_wave_base + 16*l.
// Issued through inline asm on purpose: hipcc tracks the builtin form as an LDS write on the VM counter and then
// waits vmcnt(0) in front of EVERY later ds_read (it cannot prove the other LDS buffer does not alias), which
// serialises "load tile t+1" against "compute tile t".  The k-loop has no other vector-memory loads, so the only
// vmcnt wait that matters is the explicit one at the top of each iteration.
__device__ __forceinline__ void glds16(const float* src, unsigned lds_wave_base) {
  const unsigned m0v = __builtin_amdgcn_readfirstlane(lds_wave_base);
#if ICK_ABL == 5   // timing experiment: every DMA is issued but fetches the zero page (issue cost without operand traffic)
  src = g_zero16;
#endif
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(src), "s"(m0v) : "memory");
}

// NW = waves per workgroup: 4 (2 x 2 wave grid) or 8 (4 x 2, BM = 128 only: twice the waves per SIMD behind the same LDS
// footprint — the 128-row tiles otherwise leave 2-3 waves per SIMD to cover the per-k-tile barrier and the epilogues)
// (tiles above 128 x 128 — the 256 x 256 tile of the native 16-bit variants — hold their two buffers in 128 KB of the CU's
//  160 KB of LDS, one workgroup per CU, and keep their 128 accumulator registers per lane with the full 256-VGPR budget)
// LW = LOADER waves (round 3).  In-kernel cycle stamps (tools/ablate/run_kloop.py stamps) showed where a k-tile's time goes
// when a 128 x 128 workgroup has its CU to itself: 632 cycles waiting for its DMA, 584 at the barrier, 2248 ISSUING its eight
// LDS-DMA pieces, 1464 + 3508 in the LDS reads and the 64 MFMAs (4096) — the vector-memory pipe of a CU takes in ~70 bytes per
// cycle at best, an issuing wave simply blocks until its piece is accepted, and a blocked wave issues no MFMAs.  Moving the
// pieces between the MFMA blocks (knob 2) changes nothing: the same wave still blocks.  With LW > 0 the workgroup carries LW
// extra waves that do nothing but address arithmetic + DMA issue for ALL pieces of a k-tile and wait for them to land; the NW
// compute waves never touch the vector-memory pipe inside the k-loop: barrier -> LDS reads -> MFMAs.  Every wave still meets
// at ONE barrier per k-tile (tile kt landed / the buffer of tile kt-1 is free), roles are wave-uniform scalar branches.
template <int OP, int BM, int BN, int NBUF, int TERMS, int NW, int LW = 0>
// launch bounds: second argument = minimum waves per SIMD.  Two workgroups per CU for every tile up to 128 x 128: a 4-wave
// workgroup then needs 2 (<= 256 VGPRs), an 8-wave one 4 (<= 128 VGPRs) — with 2 the eight-wave kernels silently fell to one
// workgroup per CU once the epilogue grew past 128 registers (round 3, seen in the in-kernel stamps).
// (loader variant: NW + LW = 8 waves with three buffers = one workgroup per CU, two waves per SIMD, 256 VGPRs.  Forms that
//  keep TWO workgroups per CU were tried and dropped: 4 + 2 waves on 128 x 128 needs <= 168 VGPRs and spills 130-200
//  registers; 4 + 4 waves on 128 x 64 / 64 x 128 (<= 128 VGPRs, no spills, bit-identical results) ran 5-15 % SLOWER than the
//  plain 4-wave tiles on every ViT shape — profiles/r03_kloop_knobs.log, last block)
__global__ __launch_bounds__((NW + LW) * 64, (BM * BN > 128 * 128 ? 1 : (LW > 0 ? (NBUF == 2 ? 4 : 2) : (NW == 8 ? 4 : 2))))
void igemm_glds_kernel(const P p) {
  constexpr bool AK = a_kcontig(OP), BKc = b_kcontig(OP);
  constexpr int NT = (NW + LW) * 64;
  constexpr int DW = LW > 0 ? LW : NW;                    // waves that issue the DMA
  constexpr int WM = BM / (NW / 2), WN = BN / 2, TM = WM / 32, TN = WN / 32;
  static_assert(WM >= 32 && WN >= 32, "a wave owns at least one 32x32 MFMA tile");
  constexpr int PA = BM / (8 * DW), PB = BN / (8 * DW);   // LDS-DMA instructions per issuing thread per k-tile
  constexpr int A_SZ = BM * BK, B_SZ = BN * BK, BUF = A_SZ + B_SZ;
  // X16: NATIVE 16-bit operands that are x-contiguous (TN, CONV_WGRAD with TERMS 5 / 6; round 3).  Extents stay in ELEMENTS
  // (halves).  A k-tile is 64 k-rows; the LDS images are [64 k][BX halves] (the same bytes as the fp32 [32 k][BX] image), a
  // lane's 16 bytes are 8 halves along x, and the fragment is TRANSPOSED ON THE READ (ds_read_b64_tr_b16, the read pattern of
  // igemm_bf16.hip).  The DMA cannot pad rows, so the 64-byte granules of a row are XOR-swizzled by (k-row & (granules - 1))
  // through the source address: the four k-rows a 32-lane read group touches then sit in four different 16-bank groups.
  constexpr bool X16 = TERMS >= 5 && !AK && !BKc;
  static_assert(TERMS < 5 || (AK && BKc) || X16, "native 16-bit operands: both k-contiguous or both x-contiguous");
  constexpr int BKE = X16 ? 64 : BK;                    // k extent of a k-tile in the kernel's k units
  constexpr int A_TPK = X16 ? BM / 8 : BM / 4, B_TPK = X16 ? BN / 8 : BN / 4;   // lanes per k-row of an x-contiguous image
  constexpr int A_RPI = 64 / A_TPK, B_RPI = 64 / B_TPK;  // k-rows per DMA instruction (x-contiguous)
  constexpr int A_NG = X16 ? BM / 32 : 1, B_NG = X16 ? BN / 32 : 1;             // 64-byte granules per image row (X16)
  constexpr bool H16OUT = TERMS != 0 && TERMS != 3 && TERMS != 4;       // variants that can write C / read the residual as 16-bit (P.c16 / P.r16)

  __shared__ __attribute__((aligned(16))) float lds[NBUF * BUF];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = LW > 0 ? __builtin_amdgcn_readfirstlane(tid >> 6) : (tid >> 6);   // (scalar: the role branches must be s_cbranch)
  const bool loader = LW > 0 && wave >= NW;          // this wave only feeds the LDS ring
  const bool issuer = LW == 0 || loader;             // this wave issues DMA pieces
  const int dw = LW > 0 ? (loader ? wave - NW : 0) : wave;   // index among the issuing waves
  const int wm = wave >> 1, wn = wave & 1;
  const int nwg = gridDim.x, q8 = nwg >> 3, r8 = nwg & 7, xcd = blockIdx.x & 7;   // XCD-aware tile order (igemm_f32.hip)
  const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (blockIdx.x >> 3);
  const int tile_m = wg / p.tiles_n, tile_n = wg - tile_m * p.tiles_n;
  const int m0 = p.m_base + tile_m * BM, n0 = tile_n * BN;   // m_base: first row of this launch (M-split dispatch)

  int z = blockIdx.z, split = 0;
  if (p.splitk > 1) { split = z; z = 0; }
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
  const int nkt = (kend - kbeg + BKE - 1) / BKE;

  // ---------------------------------------------------------------- per-thread DMA state
  // k-contiguous: DMA instruction q = wave*P + i covers rows 8q..8q+7; lane l -> row 8q + (l>>3), linear chunk slot l&7,
  // which holds logical chunk (l&7) ^ ((row>>1)&7).
  const float* a_ptr[PA]; bool a_ok[PA]; int a_y[PA], a_x[PA], a_kq[PA];
#pragma unroll
  for (int i = 0; i < PA; ++i) {
    const int q = dw * PA + i;
    if constexpr (AK) {
      const int row = q * 8 + (lane >> 3);
      a_kq[i] = (((lane & 7) ^ ((row >> 1) & 7)) << 2);
      const int m = m0 + row;
      a_ok[i] = m < p.M;
      if constexpr (OP == ICK_OP_NT || OP == ICK_OP_NN) {
        a_ptr[i] = Ag + (long)m * p.lda + a_kq[i]; a_y[i] = a_x[i] = 0;
      } else if constexpr (OP == ICK_OP_CONV_FWD || OP == ICK_OP_CONV_FWD_C4) {
        const int hw = p.Ho * p.Wo; const int b = m / hw; const int r = m - b * hw;
        const int oy = r / p.Wo, ox = r - oy * p.Wo;
        a_y[i] = oy * p.stride - p.pad; a_x[i] = ox * p.stride - p.pad;
        a_ptr[i] = Ag + (long)b * p.H * p.W * p.Cin;
      } else if constexpr (OP == ICK_OP_CONV_DGRAD_S2) {
        const int w2 = p.W >> 1; const int hw = (p.H >> 1) * w2; const int b = m / hw; const int r = m - b * hw;
        const int iy = 2 * (r / w2) + py, ix = 2 * (r % w2) + px;
        a_y[i] = iy + p.pad; a_x[i] = ix + p.pad;
        a_ptr[i] = Ag + (long)b * p.Ho * p.Wo * p.Cout;
      } else {  // CONV_DGRAD
        const int hw = p.H * p.W; const int b = m / hw; const int r = m - b * hw;
        const int iy = r / p.W, ix = r - iy * p.W;
        a_y[i] = iy + p.pad; a_x[i] = ix + p.pad;
        a_ptr[i] = Ag + (long)b * p.Ho * p.Wo * p.Cout;
      }
    } else if constexpr (X16) {   // A [K][M] halves: lane -> (k-row, 8 m) with the row's 64-byte granules swizzled by the k-row
      const int krow = q * A_RPI + lane / A_TPK, slot = lane % A_TPK;
      const int m = m0 + (((((slot >> 2) ^ (krow & (A_NG - 1))) << 2) | (slot & 3)) << 3);
      a_ok[i] = m < p.M;
      a_ptr[i] = reinterpret_cast<const float*>(reinterpret_cast<const unsigned short*>(Ag) + m);
      a_y[i] = krow; a_x[i] = 0; a_kq[i] = 0;
    } else {  // A stored [K][M]: instruction q covers k-rows q*A_RPI .. ; lane -> (k-row, 4 m)
      const int m = m0 + (lane % A_TPK) * 4;
      a_ok[i] = m < p.M;
      a_ptr[i] = Ag + m; a_y[i] = q * A_RPI + lane / A_TPK; a_x[i] = 0; a_kq[i] = 0;
    }
  }
  const float* b_ptr[PB]; bool b_ok[PB]; int b_y[PB], b_kq[PB];
  int b_r = 0, b_s = 0;
  // weight gradient: the output pixel (image, row, column) behind each piece's k index, kept INCREMENTALLY — a k-tile advances
  // every lane's pixel by BK, so the two integer divisions per piece per k-tile of the direct form (~50 VALU instructions
  // each) become a handful of adds and compares.  Pieces are issued in k-tile order, which is what makes this valid.
  int w_img[PB], w_oy[PB], w_ox[PB];
  const int w_q = OP == ICK_OP_CONV_WGRAD ? BKE / max(p.Wo, 1) : 0, w_r = OP == ICK_OP_CONV_WGRAD ? BKE - w_q * p.Wo : 0;
#pragma unroll
  for (int i = 0; i < PB; ++i) {
    const int q = dw * PB + i;
    if constexpr (BKc) {
      const int row = q * 8 + (lane >> 3);
      b_kq[i] = (((lane & 7) ^ ((row >> 1) & 7)) << 2);
      const int n = n0 + row;
      b_ok[i] = n < p.N;
      b_ptr[i] = Bg + (long)n * p.ldb + b_kq[i]; b_y[i] = 0;
    } else {
      const int krow = q * B_RPI + lane / B_TPK, slot = lane % B_TPK;
      const int n = X16 ? n0 + (((((slot >> 2) ^ (krow & (B_NG - 1))) << 2) | (slot & 3)) << 3) : n0 + slot * 4;
      b_ok[i] = n < p.N;
      b_y[i] = krow; b_kq[i] = 0;
      auto belem = [&](long e) {   // Bg + e elements (halves for X16)
        if constexpr (X16) return reinterpret_cast<const float*>(reinterpret_cast<const unsigned short*>(Bg) + e);
        else return Bg + e;
      };
      if constexpr (OP == ICK_OP_CONV_WGRAD) {
        const int tap = n / p.Cin; const int ci = n - tap * p.Cin;
        b_r = tap / p.S; b_s = tap - b_r * p.S;
        b_ptr[i] = belem(ci);
        const int hw = p.Ho * p.Wo;
        const int k = kbeg + b_y[i];
        w_img[i] = k / hw; const int rem = k - w_img[i] * hw; w_oy[i] = rem / p.Wo; w_ox[i] = rem - w_oy[i] * p.Wo;
      } else {
        b_ptr[i] = belem(n);
      }
    }
  }

  const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)lds;
  const float* zpage = g_zero16;
  if constexpr ((ICK_EXP & 1) != 0) asm volatile("" : "+v"(zpage));   // opaque: stays in two VGPRs, never re-materialised from the GOT

  // What one k-tile's DMA needs besides the per-thread state: its first k and (convolutions) the filter tap it lies in.
  // `valid` false turns every piece into a zero-page fetch (the piecewise schedule issues unconditionally: no branches
  // between the MFMA blocks; the epilogue drains them before it reuses the buffers).
  struct KTile { int k0, r, s, c0, tap; bool valid; };
  auto ktile = [&](int kt, bool valid) -> KTile {
    KTile t; t.k0 = kbeg + kt * BKE; t.r = t.s = t.c0 = t.tap = 0; t.valid = valid;
    if constexpr (OP == ICK_OP_CONV_FWD) {
      const int tap = t.k0 / p.Cin; t.c0 = t.k0 - tap * p.Cin; t.r = tap / p.S; t.s = tap - t.r * p.S;
    } else if constexpr (OP == ICK_OP_CONV_DGRAD) {
      t.tap = t.k0 / p.Cout; t.c0 = t.k0 - t.tap * p.Cout; t.r = t.tap / p.S; t.s = t.tap - t.r * p.S;
    } else if constexpr (OP == ICK_OP_CONV_DGRAD_S2) {
      const int q = t.k0 / p.Cout; t.c0 = t.k0 - q * p.Cout;
      t.r = r0 + 2 * (q / ns); t.s = s0 + 2 * (q % ns); t.tap = t.r * p.S + t.s;
    }
    return t;
  };
  // pointer + element offset for the x-contiguous operands (elements are halves in the X16 variants)
  auto eoff = [](const float* ptr, long e) -> const float* {
    if constexpr (X16) return reinterpret_cast<const float*>(reinterpret_cast<const unsigned short*>(ptr) + e);
    else return ptr + e;
  };
  // piece q of a k-tile's DMA into buffer buf: q < PA -> A instruction q of this wave, else B instruction q - PA
  auto issue_piece = [&](const KTile& t, int q, int buf) {
    const int k0 = t.k0;
    if (q < PA) {
      const int i = q;
      const unsigned dst = lds_base + 4u * (buf * BUF + dw * (PA * 256)) + i * 1024;       // LDS byte address (wave-uniform)
      bool ok; const float* src;
      if constexpr (OP == ICK_OP_NT || OP == ICK_OP_NN) {
        ok = a_ok[i] && (k0 + a_kq[i] < kend); src = a_ptr[i] + k0;
      } else if constexpr (OP == ICK_OP_CONV_FWD) {
        const int iy = a_y[i] + t.r, ix = a_x[i] + t.s;
        ok = a_ok[i] && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W && (k0 + a_kq[i] < kend);
        src = a_ptr[i] + ((long)iy * p.W + ix) * p.Cin + t.c0 + a_kq[i];
      } else if constexpr (OP == ICK_OP_CONV_FWD_C4) {
        const int tap = (k0 + a_kq[i]) >> 2; const int r = tap / p.S, s = tap - r * p.S;
        const int iy = a_y[i] + r, ix = a_x[i] + s;
        ok = a_ok[i] && r < p.R && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        src = a_ptr[i] + ((long)iy * p.W + ix) * 4;
      } else if constexpr (OP == ICK_OP_CONV_DGRAD) {
        const int ty = a_y[i] - t.r, tx = a_x[i] - t.s;
        const int oy = ty / p.stride, ox = tx / p.stride;
        ok = a_ok[i] && ty >= 0 && tx >= 0 && oy * p.stride == ty && ox * p.stride == tx &&
             oy < p.Ho && ox < p.Wo && (k0 + a_kq[i] < kend);
        src = a_ptr[i] + ((long)oy * p.Wo + ox) * p.Cout + t.c0 + a_kq[i];
      } else if constexpr (OP == ICK_OP_CONV_DGRAD_S2) {
        const int ty = a_y[i] - t.r, tx = a_x[i] - t.s;
        const int oy = ty >> 1, ox = tx >> 1;
        ok = a_ok[i] && ty >= 0 && tx >= 0 && oy < p.Ho && ox < p.Wo && (k0 + a_kq[i] < kend);
        src = a_ptr[i] + ((long)oy * p.Wo + ox) * p.Cout + t.c0 + a_kq[i];
      } else {  // A [K][M]
        const int k = k0 + a_y[i];
        ok = a_ok[i] && k < kend; src = eoff(a_ptr[i], (long)k * p.lda);
      }
      glds16((ok && t.valid) ? src : zpage, dst);
    } else {
      const int i = q - PA;
      const unsigned dst = lds_base + 4u * (buf * BUF + A_SZ + dw * (PB * 256)) + i * 1024;
      bool ok; const float* src;
      if constexpr (BKc) {
        ok = b_ok[i] && (k0 + b_kq[i] < kend); src = b_ptr[i] + k0;
      } else if constexpr (OP == ICK_OP_CONV_DGRAD || OP == ICK_OP_CONV_DGRAD_S2) {
        const int co = t.c0 + b_y[i];
        ok = b_ok[i] && (k0 + b_y[i] < kend); src = b_ptr[i] + ((long)co * p.R * p.S + t.tap) * p.Cin;
      } else if constexpr (OP == ICK_OP_CONV_WGRAD) {
        const int k = k0 + b_y[i];
        const int iy = w_oy[i] * p.stride - p.pad + b_r, ix = w_ox[i] * p.stride - p.pad + b_s;
        ok = b_ok[i] && k < kend && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        src = eoff(b_ptr[i], (((long)w_img[i] * p.H + iy) * p.W + ix) * p.Cin);
        w_ox[i] += w_r; w_oy[i] += w_q;                     // the same piece of the NEXT k-tile: BK pixels on
        if (w_ox[i] >= p.Wo) { w_ox[i] -= p.Wo; ++w_oy[i]; }
        while (w_oy[i] >= p.Ho) { w_oy[i] -= p.Ho; ++w_img[i]; }
      } else {  // B [K][N]
        const int k = k0 + b_y[i];
        ok = b_ok[i] && k < kend; src = eoff(b_ptr[i], (long)k * p.ldb);
      }
      glds16((ok && t.valid) ? src : zpage, dst);
    }
  };
  constexpr int NP = PA + PB;                      // DMA pieces per wave per k-tile
  // issue the LDS-DMA of k-tile kt into buffer buf (one burst)
  auto issue = [&](int kt, int buf) {
    const KTile t = ktile(kt, true);
#pragma unroll
    for (int q = 0; q < NP; ++q) issue_piece(t, q, buf);
  };

  // fragments of k-group j (8 k) of tile-row/column t: element e feeds MFMA e (k-slots {8j+e, 8j+4+e})
  const int frow = lane & 31, fh = lane >> 5, fsw = (frow >> 1) & 7;
  // X16: transposed fragment of MFMA k-step j (16 k-rows) of 32 image columns starting at col0 (ds_read_b64_tr_b16 twice: the
  // read pattern of igemm_bf16.hip tr_read8 — 16-lane group tg, block row tq, 4-column piece tp), swizzled like the DMA wrote it
  const int xtg = lane >> 4, xtq = (lane >> 2) & 3, xtp = lane & 3;
  auto frag_x16 = [&](const float* img, int j, int col0, int bx, int ng) -> float4 {
    typedef short s16x4 __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) s16x4* lds_p;
    const int krow = j * 16 + (xtg >> 1) * 8 + xtq;
    const int col = col0 + (xtg & 1) * 16 + xtp * 4;                       // halves; 64-byte granule = col >> 5
    const unsigned short* base = reinterpret_cast<const unsigned short*>(img);
    const unsigned short* q = base + krow * bx + ((((col >> 5) ^ (krow & (ng - 1))) << 5) | (col & 31));
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(q));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(q + 4 * bx));   // k-rows + 4: same swizzle (granule counts are 2 or 4)
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(float4, v);
  };
  auto frag_a = [&](const float* Ab, int j, int i) -> float4 {
    if constexpr (X16) {
      return frag_x16(Ab, j, wm * WM + i * 32, BM, A_NG);
    } else if constexpr (AK) {
      return *reinterpret_cast<const float4*>(Ab + (wm * WM + i * 32 + frow) * BK + (((2 * j + fh) ^ fsw) << 2));
    } else {
      const float* q = Ab + (8 * j + 4 * fh) * BM + wm * WM + i * 32 + frow;
      return make_float4(q[0], q[BM], q[2 * BM], q[3 * BM]);
    }
  };
  auto frag_b = [&](const float* Bb, int j, int t) -> float4 {
    if constexpr (X16) {
      return frag_x16(Bb, j, wn * WN + t * 32, BN, B_NG);
    } else if constexpr (BKc) {
      return *reinterpret_cast<const float4*>(Bb + (wn * WN + t * 32 + frow) * BK + (((2 * j + fh) ^ fsw) << 2));
    } else {
      const float* q = Bb + (8 * j + 4 * fh) * BN + wn * WN + t * 32 + frow;
      return make_float4(q[0], q[BN], q[2 * BN], q[3 * BN]);
    }
  };

  // 16-bit variants: fp32 fragment of MFMA k-step s (16 k): lane l holds X[row l&31][k = 16 s + 8 (l>>5) + 0..7]
  auto frag_a8 = [&](const float* Ab, int s, int i) -> f32x8 {
    if constexpr (AK) {
      const float* r = Ab + (wm * WM + i * 32 + frow) * BK;
      const float4 lo = *reinterpret_cast<const float4*>(r + (((4 * s + 2 * fh) ^ fsw) << 2));
      const float4 hi = *reinterpret_cast<const float4*>(r + (((4 * s + 2 * fh + 1) ^ fsw) << 2));
      return f32x8{lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    } else {
      const float* q = Ab + (16 * s + 8 * fh) * BM + wm * WM + i * 32 + frow;
      return f32x8{q[0], q[BM], q[2 * BM], q[3 * BM], q[4 * BM], q[5 * BM], q[6 * BM], q[7 * BM]};
    }
  };
  auto frag_b8 = [&](const float* Bb, int s, int t) -> f32x8 {
    if constexpr (BKc) {
      const float* r = Bb + (wn * WN + t * 32 + frow) * BK;
      const float4 lo = *reinterpret_cast<const float4*>(r + (((4 * s + 2 * fh) ^ fsw) << 2));
      const float4 hi = *reinterpret_cast<const float4*>(r + (((4 * s + 2 * fh + 1) ^ fsw) << 2));
      return f32x8{lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    } else {
      const float* q = Bb + (16 * s + 8 * fh) * BN + wn * WN + t * 32 + frow;
      return f32x8{q[0], q[BN], q[2 * BN], q[3 * BN], q[4 * BN], q[5 * BN], q[6 * BN], q[7 * BN]};
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  // Chunked accumulation: one MFMA accumulator is a strictly sequential fp32 chain over k, whose rounding error grows
  // with the chain length (measured: a K = 4608 chain is 4x further from fp64 than a K-blocked CPU GEMM, and that excess
  // shows up 1.5-3x in every gradient behind the ResNet trunk).  Every p.chunk_tiles k-tiles (64 k by default; CPU
  // GEMMs block K at ~256) the chain is folded into a master sum and restarted from zero.  Measured on the KD step
  // (tools/diag_grads.py, profiles/r02_diag_grads_B{2,8}.log, chunk 128): gradient error vs fp64 relative to torch's
  // CPU fp32 falls from 4.3-4.5x (refinement / decoder, B = 8) to 0.98x, trunk 1.10x -> 0.92x; an isolated K = 2304
  // data gradient goes from 1.8e-6 to 2.6e-7 of scale.  Cost: none measurable (29.83 vs 29.81 ms/step at chunk 64 / 128,
  // 194.3 vs 195.1 us for the ViT fc1 GEMM with / without folding): the fold's VALU adds hide under other waves' MFMAs.
  f32x16 tot[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) tot[i][j][r] = 0.f;
  int chain = 0;
  // TERMS 4 with IckGemm.a_absmax: s = 2^(11 - e) for max |A| = m 2^e, m in [0.5, 1): max |A| s lies in [2^10, 2^11); a zero
  // or non-finite maximum leaves A alone
  bool a_scaled = false;
  float a_sc = 1.f, a_inv = 1.f;
  if constexpr (TERMS == 4) {
    if (p.a_absmax) {
      const float am = *p.a_absmax;                       // (uniform: every lane loads the same word)
      if (am > 0.f && am < 3.0e38f) {
        int e;
        (void)frexpf(am, &e);
        a_sc = ldexpf(1.f, 11 - e);
        a_inv = ldexpf(1.f, e - 11);
        a_scaled = true;
      }
    }
  }

  constexpr bool IL = (ICK_EXP & 2) != 0 && TERMS == 0 && LW == 0;   // piecewise DMA schedule (knob 2)
  auto stamp = [&](int slot, bool on) {
#if (ICK_EXP & 8)
    if (on && g_dbg) {
      const long long tm = __builtin_amdgcn_s_memrealtime();   // 100 MHz wall clock
      if (tid == 0) {
        long long* d = g_dbg + ((long)blockIdx.z * gridDim.x + blockIdx.x) * 16;
        d[slot] = tm;
        if (slot == 0) { d[6] = __builtin_amdgcn_s_getreg((32 - 1) << 11 | 4); d[7] = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 20); }
      }
    }
#endif
  };
  // cycle stamps inside ONE k-tile (the third), wave 0 of every workgroup: where a k-tile's non-MFMA time goes
  auto kstamp = [&](int slot, int kt) {
#if (ICK_EXP & 64)
    if (kt == 2 && g_dbg) {
      const long long tm = __builtin_amdgcn_s_memtime();
      if (tid == 0) g_dbg[((long)blockIdx.z * gridDim.x + blockIdx.x) * 16 + 8 + slot] = tm;
    }
#endif
  };
  stamp(0, true);
#if (ICK_EXP & 32)
  // Phase stagger (experiment): the two workgroups that share a CU start together and stay in lock-step — both in their
  // prologues, both in their epilogues, the matrix pipe idle under both.  The workgroup whose waves sit in ODD wave slots of
  // their SIMDs sleeps for g_stagger x 64 cycles once, in the first round of the grid only: later workgroups inherit the phase
  // of the slot they replace.
  if (g_stagger > 0 && (long)blockIdx.z * gridDim.x + blockIdx.x < 2 * kCUs) {
    const unsigned wave_slot = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 4) & 15u;   // HW_ID[3:0] = wave id within the SIMD
    if (wave_slot & 1u)
      for (int i = 0; i < g_stagger; i += 16) __builtin_amdgcn_s_sleep(16);
  }
#endif
  // NBUF = 2: DMA of tile t+1 is issued at the top of iteration t (one compute phase to land).
  // NBUF = 3: DMA of tile t+2 is issued at the top of iteration t (two compute phases to land); the wait at the top of
  //           iteration t leaves the PA+PB most recent DMAs (tile t+1) in flight.
  if constexpr (IL) {   // the piecewise schedule counts on NP pieces per k-tile slot being in flight, real or zero-page
    const KTile t0 = ktile(0, nkt > 0);
#pragma unroll
    for (int q = 0; q < NP; ++q) issue_piece(t0, q, 0);
    if (NBUF == 3) {
      const KTile t1 = ktile(1, nkt > 1);
#pragma unroll
      for (int q = 0; q < NP; ++q) issue_piece(t1, q, 1);
    }
  } else if (issuer) {
    if (nkt > 0) issue(0, 0);
    if (NBUF == 3 && nkt > 1) issue(1, 1);
  }
  int cb = 0;                                     // buffer of tile kt
  for (int kt = 0; kt < nkt; ++kt) {
    // my DMA of tile kt has landed; after the barrier so has everybody's, and everybody is done reading the buffer
    // that the next DMA overwrites
    kstamp(0, kt);
    if (issuer) {
      if (NBUF == 3 && (IL || kt + 1 < nkt)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PA + PB) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    kstamp(1, kt);
    __syncthreads();
    kstamp(2, kt);
    stamp(1, kt == 0);
    const int nbuf = cb == 0 ? NBUF - 1 : cb - 1;   // the buffer the next DMA fills
    KTile nxt_t = ktile(kt + NBUF - 1, kt + NBUF - 1 < nkt);
    if constexpr (!IL) {
#if ICK_ABL == 3
      if (issuer && kt + NBUF - 1 < nkt && p.alpha == 12345.f) issue(kt + NBUF - 1, nbuf);
#else
      if (issuer && kt + NBUF - 1 < nkt) issue(kt + NBUF - 1, nbuf);
#endif
    }
    kstamp(3, kt);
    const float* Ab = lds + cb * BUF;
    cb = cb + 1 == NBUF ? 0 : cb + 1;
    if (LW > 0 && loader) continue;                  // a loader's k-tile ends here: on to waiting for the next tile's pieces
    const float* Bb = Ab + A_SZ;
    if constexpr (TERMS == 0 || TERMS >= 5) {
      // fragment registers: two sets (the next group's reads fly under this group's MFMAs); ONE set for the 256 x 256 tile,
      // whose 128 accumulator registers leave no room for a second (spills otherwise) — its two waves per SIMD cover each
      // other's LDS latency instead, and its reads for group j + 1 are issued behind group j's eight MFMAs
      constexpr int FB = BM * BN > 128 * 128 ? 1 : 2;
      float4 av[FB][TM], bv[FB][TN];
  #pragma unroll
      for (int i = 0; i < TM; ++i) av[0][i] = frag_a(Ab, 0, i);
  #pragma unroll
      for (int t = 0; t < TN; ++t) bv[0][t] = frag_b(Bb, 0, t);
  #pragma unroll
      for (int j = 0; j < BK / 8; ++j) {
        const int cur = FB == 2 ? (j & 1) : 0, nxt = FB == 2 ? (cur ^ 1) : 0;
        if (FB == 2 && j + 1 < BK / 8) {
  #pragma unroll
          for (int i = 0; i < TM; ++i) av[nxt][i] = frag_a(Ab, j + 1, i);
  #pragma unroll
          for (int t = 0; t < TN; ++t) bv[nxt][t] = frag_b(Bb, j + 1, t);
        }
        // keep the next group's LDS reads AHEAD of this group's MFMAs (hipcc otherwise sinks them behind the MFMAs and
        // exposes the LDS latency at the head of every group)
        __builtin_amdgcn_sched_barrier(0);
        if constexpr ((ICK_EXP & 4) != 0) if (j == 0) __builtin_amdgcn_s_setprio(1);
  #pragma unroll
        for (int i = 0; i < TM; ++i)
  #pragma unroll
          for (int t = 0; t < TN; ++t) {
            if constexpr (TERMS >= 5) {      // native 16-bit operands: the four units ARE the lane's eight halves
              using h8 = typename Half16<TERMS>::x8;
              acc[i][t] = mfma16<TERMS>(__builtin_bit_cast(h8, av[cur][i]), __builtin_bit_cast(h8, bv[cur][t]), acc[i][t]);
              continue;
            }
            acc[i][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cur][i].x, bv[cur][t].x, acc[i][t], 0, 0, 0);
            acc[i][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cur][i].y, bv[cur][t].y, acc[i][t], 0, 0, 0);
            acc[i][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cur][i].z, bv[cur][t].z, acc[i][t], 0, 0, 0);
            acc[i][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cur][i].w, bv[cur][t].w, acc[i][t], 0, 0, 0);
            if constexpr (IL) {
              // piecewise DMA: block (j, i, t) is slot sl of 4 TM TN; the NP pieces go behind the first MFMA blocks of
              // the k-tile, PPS per block, so that the last one still has half a k-tile of MFMAs to land under
              constexpr int NSLOT = (BK / 8) * TM * TN, PPS = (NP + NSLOT / 2 - 1) / (NSLOT / 2);
              const int sl = (j * TM + i) * TN + t;
  #pragma unroll
              for (int u = 0; u < PPS; ++u)
                if (sl * PPS + u < NP) { __builtin_amdgcn_sched_barrier(0); issue_piece(nxt_t, sl * PPS + u, nbuf); __builtin_amdgcn_sched_barrier(0); }
            }
          }
        if constexpr ((ICK_EXP & 4) != 0) if (j == BK / 8 - 1) __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        if (j == 0) kstamp(4, kt);
        if (j == BK / 8 - 1) kstamp(5, kt);
        if (FB == 1 && j + 1 < BK / 8) {
  #pragma unroll
          for (int i = 0; i < TM; ++i) av[0][i] = frag_a(Ab, j + 1, i);
  #pragma unroll
          for (int t = 0; t < TN; ++t) bv[0][t] = frag_b(Bb, j + 1, t);
        }
      }
    } else {
      f32x8 av[2][TM], bv[2][TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) av[0][i] = frag_a8(Ab, 0, i);
#pragma unroll
      for (int t = 0; t < TN; ++t) bv[0][t] = frag_b8(Bb, 0, t);
#pragma unroll
      for (int s = 0; s < BK / 16; ++s) {
        const int cur = s & 1, nxt = cur ^ 1;
        if (s + 1 < BK / 16) {
#pragma unroll
          for (int i = 0; i < TM; ++i) av[nxt][i] = frag_a8(Ab, s + 1, i);
#pragma unroll
          for (int t = 0; t < TN; ++t) bv[nxt][t] = frag_b8(Bb, s + 1, t);
        }
        using h8 = typename Half16<TERMS>::x8;
        h8 ah[TM], bh[TN], al[TM], bl[TN];
        if constexpr (TERMS == 4) {
          if (a_scaled) {            // IckGemm.a_absmax: A times the power of two that puts its maximum into fp16's best range (exact)
#pragma unroll
            for (int i = 0; i < TM; ++i) av[cur][i] = av[cur][i] * a_sc;
          }
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          ah[i] = __builtin_convertvector(av[cur][i], h8);
          if constexpr (TERMS == 3) al[i] = __builtin_convertvector(av[cur][i] - __builtin_convertvector(ah[i], f32x8), h8);
          if constexpr (TERMS == 4) al[i] = __builtin_convertvector((av[cur][i] - __builtin_convertvector(ah[i], f32x8)) * 2048.f, h8);
        }
#pragma unroll
        for (int t = 0; t < TN; ++t) {
          bh[t] = __builtin_convertvector(bv[cur][t], h8);
          if constexpr (TERMS == 3) bl[t] = __builtin_convertvector(bv[cur][t] - __builtin_convertvector(bh[t], f32x8), h8);
          if constexpr (TERMS == 4) bl[t] = __builtin_convertvector((bv[cur][t] - __builtin_convertvector(bh[t], f32x8)) * 2048.f, h8);
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int t = 0; t < TN; ++t) {
            if constexpr (TERMS == 3) {   // small terms first
              acc[i][t] = mfma16<TERMS>(al[i], bh[t], acc[i][t]);
              acc[i][t] = mfma16<TERMS>(ah[i], bl[t], acc[i][t]);
            }
            if constexpr (TERMS == 4) {   // the cross terms, scaled by 2^11, in their own accumulator
              tot[i][t] = mfma16<TERMS>(al[i], bh[t], tot[i][t]);
              tot[i][t] = mfma16<TERMS>(ah[i], bl[t], tot[i][t]);
            }
            acc[i][t] = mfma16<TERMS>(ah[i], bh[t], acc[i][t]);
          }
      }
    }
    if constexpr (TERMS == 0)          // (16-bit products: operand rounding dominates the chain's — and `acc + 0.0f` is not
    if (++chain == p.chunk_tiles && kt + 1 < nkt) {   //  foldable, so an unconditional master sum would cost 16 TM TN live registers)
      chain = 0;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int t = 0; t < TN; ++t) {
          tot[i][t] += acc[i][t];
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][t][r] = 0.f;
        }
    }
  }
  if constexpr (TERMS == 0) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int t = 0; t < TN; ++t) acc[i][t] += tot[i][t];
  }
  if constexpr (TERMS == 4) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int t = 0; t < TN; ++t) acc[i][t] = (acc[i][t] + tot[i][t] * (1.f / 2048.f)) * a_inv;    // (a_inv = 1 without a_absmax)
  }
  if constexpr (IL) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the last k-tiles' zero-page pieces must land before the buffers become the C slab
  stamp(2, true);

  // ---------------------------------------------------------------- epilogue
  // C/D layout of the 32x32 tile: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
  // Interior tiles (the vast majority) take a check-free path; flags are wave-uniform scalars.
  float* __restrict__ Cg = p.C + coff;
  const float* __restrict__ Rg = (p.residual && split == 0) ? p.residual + coff : nullptr;
  const float* __restrict__ biasp = (p.bias && split == 0) ? p.bias : nullptr;
  const int mode = p.splitk > 1 ? 2 : (p.accumulate ? 1 : 0);
  const int act = p.act & 15;
  const bool post = (p.act & ICK_ACT_POST_RESIDUAL) != 0;   // activation after the residual (staged epilogue only)
  const float alpha = p.alpha;
  if (p.ep_vec && mode != 2) {
    // LDS-staged epilogue: the block tile goes registers -> LDS (bias / activation applied, statistics taken on the
    // way) and leaves as 16-byte stores along n, 2 x 512-byte rows (128-wide tiles) per wave-instruction instead of
    // 2 x 128 bytes: the direct path below issues 4x the store instructions and prices 5-30 % of a short-K GEMM
    // (profiles/r01g_ablation_glds.log, V1 -> V2).  The residual is read in the same coalesced pattern.
    // A tile larger than the buffers (256 x 256: 256 KB of fp32 against 128 KB of LDS) leaves in SLABS row slabs.
    constexpr int SLABS = (BM * BN + NBUF * BUF - 1) / (NBUF * BUF);
    constexpr int SROWS = BM / SLABS;
    static_assert(BM % SLABS == 0 && SROWS % WM == 0, "a wave's rows lie in one slab");
    float* ct = lds;
#pragma unroll
    for (int sl = 0; sl < SLABS; ++sl) {
    const int srow0 = sl * SROWS;
    __syncthreads();                       // every wave is out of the k-loop / the previous slab is stored: the LDS buffers become the C slab [SROWS][BN]
    if (!loader && (SLABS == 1 || (wm * WM) / SROWS == sl)) {
    // registers -> LDS, one specialised copy per activation (igemm_params.h act_dispatch); the statistics only where asked for
    act_dispatch(post ? ICK_ACT_NONE : act, [&](auto act_tag) {
    constexpr int ACT = decltype(act_tag)::value;
    const bool want_stats = p.stat_sum != nullptr;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int nl = wn * WN + j * 32 + (lane & 31);
      const int n = n0 + nl;
      const bool nok = n < p.N;
      const float bias = (biasp && nok) ? biasp[n] : 0.f;
      const float csc = (p.col_scale && nok) ? p.col_scale[n] : 1.f;   // eval-mode BatchNorm scale (else exactly v + bias)
      float ssum = 0.f, ssq = 0.f;   // ssq by explicit fmaf: the epilogue variants must agree bit for bit
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int ml = wm * WM + i * 32 + 4 * (lane >> 5);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = ml + (r & 3) + 8 * (r >> 2);
          const float v = acc[i][j][r] * alpha;
          if (want_stats && nok && m0 + row < p.M) { ssum += v; ssq = fmaf(v, v, ssq); }
          const float u = fmaf(v, csc, bias);
          ct[(row - srow0) * BN + nl] = act_c<ACT>(u);
        }
      }
      if (want_stats) {
        ssum += __shfl_xor(ssum, 32);
        ssq += __shfl_xor(ssq, 32);
        if (lane < 32 && nok) {
          const long so = (long)(tile_m % p.stat_copies) * p.stat_stride + n;
          atomicAdd(p.stat_sum + so, (double)ssum); atomicAdd(p.stat_sq + so, (double)ssq);
        }
      }
    }
    });
    }
    stamp(4, sl == 0);
    __syncthreads();
    stamp(5, sl == 0);
    constexpr int C4 = BN / 4;
    for (int c = tid; c < SROWS * C4; c += NT) {
      const int row = srow0 + c / C4, col = (c % C4) * 4;
      const int m = m0 + row, n = n0 + col;
      if (m < p.M && n < p.N) {            // N % 4 == 0 on this path: the whole chunk is in range
        long mr = m;
        if constexpr (OP == ICK_OP_CONV_DGRAD_S2) {
          const int w2 = p.W >> 1; const int hw = (p.H >> 1) * w2; const int b = m / hw; const int q = m - b * hw;
          mr = ((long)b * p.H + 2 * (q / w2) + py) * p.W + 2 * (q % w2) + px;
        }
        float4 v = *reinterpret_cast<const float4*>(ct + (row - srow0) * BN + col);
        using h4 = typename Half16<TERMS>::x4;
        if (Rg) {
          float4 q;
          if (H16OUT && p.r16) q = ld4h<h4>(p.residual, coff + mr * p.ldr + n);
          else q = *reinterpret_cast<const float4*>(Rg + mr * p.ldr + n);
          v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
        }
        if (post) { v.x = act_fn(v.x, act); v.y = act_fn(v.y, act); v.z = act_fn(v.z, act); v.w = act_fn(v.w, act); }
        if (H16OUT && p.c16) {           // 16-bit C: 8-byte row chunks
          const long o = coff + mr * p.ldc + n;
          if (mode == 1) { const float4 q = ld4h<h4>(p.C, o); v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w; }
          st4h<h4>(p.C, o, v);
          continue;
        }
        float4* dst = reinterpret_cast<float4*>(Cg + mr * p.ldc + n);
        if (mode == 1) { const float4 o = *dst; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
        *dst = v;
      }
    }
    }
    stamp(3, true);
    return;
  }
  auto epilogue = [&](auto full_tag, auto act_tag) {
    constexpr bool FULL = decltype(full_tag)::value;
    constexpr int ACT = decltype(act_tag)::value;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + wn * WN + j * 32 + (lane & 31);
      const bool nok = FULL || n < p.N;
      const float bias = (biasp && nok) ? biasp[n] : 0.f;
      float ssum = 0.f, ssq = 0.f;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int mb = m0 + wm * WM + i * 32 + 4 * (lane >> 5);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = mb + (r & 3) + 8 * (r >> 2);
          if ((FULL || (m < p.M && nok)) && (ICK_ABL < 2 || ICK_ABL >= 4 || alpha == 12345.f)) {
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
      if (p.stat_sum && (ICK_ABL < 1 || ICK_ABL >= 4 || alpha == 12345.f)) {  // BatchNorm batch statistics of the raw product; fp64 so that E[x^2]-E[x]^2 cannot cancel
        ssum += __shfl_xor(ssum, 32);
        ssq += __shfl_xor(ssq, 32);
        if (lane < 32 && nok) {
          const long so = (long)(tile_m % p.stat_copies) * p.stat_stride + n;   // copy of this row tile
#if ICK_ABL == 4   // timing experiment only: fp32 atomics instead of fp64
          atomicAdd(reinterpret_cast<float*>(p.stat_sum) + so, ssum); atomicAdd(reinterpret_cast<float*>(p.stat_sq) + so, ssq);
#else
          atomicAdd(p.stat_sum + so, (double)ssum); atomicAdd(p.stat_sq + so, (double)ssq);
#endif
        }
      }
    }
  };
  // (interior tiles check-free; activations other than none / ReLU reach this dword path only with an unaligned C: they share
  //  the boundary-checked copy)
  if (loader) return;
  if (act == ICK_ACT_NONE) {
    if (m0 + BM <= p.M && n0 + BN <= p.N) epilogue(std::true_type{}, ActTag<ICK_ACT_NONE>{});
    else epilogue(std::false_type{}, ActTag<ICK_ACT_NONE>{});
  } else {
    epilogue(std::false_type{}, ActTag<-1>{});       // -1: the run-time form (one more copy would cost registers for a rare path)
  }
  stamp(3, true);
}

// rows [m_begin, m_end) of the problem (m_end <= 0: all of M).  A launch over a row range is what the M-split dispatch
// uses: bounds are checked against p.M = m_end, addresses are formed from the global row index.
template <int OP, int BM, int BN, int NBUF, int TERMS, int NW = 4, int LW = 0>
int launch(const P& p0, int nz, hipStream_t st, int m_begin = 0, int m_end = 0) {
  P p = p0;
  if (m_end <= 0 || m_end > p.M) m_end = p.M;
  p.m_base = m_begin;
  p.M = m_end;
  p.tiles_n = (p.N + BN - 1) / BN;
  // 16-byte stores need every row start of C (and of the residual) 16-byte aligned
  p.ep_vec = p.N % 4 == 0 && p.ldc % 4 == 0 && (p.sCo | p.sCi) % 4 == 0 && ick::aligned16(p.C) &&
             (!p.residual || (p.ldr % 4 == 0 && ick::aligned16(p.residual))) && !g_no_vec_epilogue && !p.no_ep_vec;
  if ((p.col_scale || (p.act & ICK_ACT_POST_RESIDUAL)) && !(p.ep_vec && p.splitk == 1))
    return ick::fail(-1, "igemm: col_scale / ICK_ACT_POST_RESIDUAL need 16-byte aligned C rows (N %% 4, ldc %% 4) and no split-K");
  if ((p.c16 || p.r16) && !((TERMS != 0 && TERMS != 3) && p.ep_vec && p.splitk == 1))
    return ick::fail(-1, "igemm: a 16-bit C / residual needs a 16-bit MFMA variant of the LDS-DMA kernel, N %% 4, ldc %% 4, ldr %% 4 and no split-K");
  dim3 grid(p.tiles_n * ((p.M - m_begin + BM - 1) / BM), 1, nz);
  ICK_LAUNCH((igemm_glds_kernel<OP, BM, BN, NBUF, TERMS, NW, LW>), grid, dim3((NW + LW) * 64), 0, st, p);
  return ick::launch_status("igemm_glds");
}

constexpr int kBodySlots = 2 * kCUs; // 128x128 workgroups resident at once (64 KiB of LDS each: two per CU)

template <int OP, int TERMS>
int launch_tile(const P& p, int nz, hipStream_t st, int tile, int m_begin = 0, int m_end = 0) {
  if constexpr (TERMS >= 5 && !a_kcontig(OP)) {   // native 16-bit x-contiguous operands (TN, CONV_WGRAD): the two-buffer tiles only
    switch (tile & (64 | 15)) {
      case 1: return launch<OP, 128, 128, 2, TERMS>(p, nz, st, m_begin, m_end);
      case 2: return launch<OP, 64, 64, 2, TERMS>(p, nz, st, m_begin, m_end);
      case 4: return launch<OP, 64, 128, 2, TERMS>(p, nz, st, m_begin, m_end);
      case 65: return launch<OP, 128, 128, 2, TERMS, 8>(p, nz, st, m_begin, m_end);
      case 67: return launch<OP, 128, 64, 2, TERMS, 8>(p, nz, st, m_begin, m_end);
      default: return launch<OP, 128, 64, 2, TERMS>(p, nz, st, m_begin, m_end);
    }
  } else
  switch (tile) {           // +16: three LDS buffers (two tiles of prefetch) instead of two
    case 2: return launch<OP, 64, 64, 2, TERMS>(p, nz, st, m_begin, m_end);
    case 3: return launch<OP, 128, 64, 2, TERMS>(p, nz, st, m_begin, m_end);
    case 4: return launch<OP, 64, 128, 2, TERMS>(p, nz, st, m_begin, m_end);
    case 18: return launch<OP, 64, 64, 3, TERMS>(p, nz, st, m_begin, m_end);
    case 19: return launch<OP, 128, 64, 3, TERMS>(p, nz, st, m_begin, m_end);
    case 20: return launch<OP, 64, 128, 3, TERMS>(p, nz, st, m_begin, m_end);
    case 65: return launch<OP, 128, 128, 2, TERMS, 8>(p, nz, st, m_begin, m_end);     // +64: eight waves per workgroup
    case 67: return launch<OP, 128, 64, 2, TERMS, 8>(p, nz, st, m_begin, m_end);
    case 83: return launch<OP, 128, 64, 3, TERMS, 8>(p, nz, st, m_begin, m_end);
    case 129:     // +128: loader waves.  129 = 4 compute + 4 loader waves, three buffers: ONE workgroup per CU (long-K shapes)
      if constexpr ((TERMS == 0 || TERMS == 4) && (OP == ICK_OP_NT || OP == ICK_OP_CONV_FWD)) return launch<OP, 128, 128, 3, TERMS, 4, 4>(p, nz, st, m_begin, m_end);
      else return launch<OP, 128, 128, 2, TERMS>(p, nz, st, m_begin, m_end);
    case 69:      // 256 x 256 x (64 halves), eight waves as 4 x 2 (64 x 128 per wave): large plain GEMMs on native 16-bit operands
      if constexpr (TERMS >= 5 && OP == ICK_OP_NT) return launch<OP, 256, 256, 2, TERMS, 8>(p, nz, st, m_begin, m_end);
      else return launch<OP, 128, 128, 2, TERMS, 8>(p, nz, st, m_begin, m_end);
    default: return launch<OP, 128, 128, 2, TERMS>(p, nz, st, m_begin, m_end);
  }
}

// M-split (IckGemm.tile +32): the 128x128 tile is the family's most efficient at long K (half the operand stream per FLOP
// of the 64x64 one: 119-127 TF against ~80 at 4096^3) but a grid of T such workgroups runs in ceil(T / 512) rounds, and
// the step's shapes leave the last round mostly empty (ViT: M = 12608 -> 2.32 rounds).  So: the rows that fill WHOLE
// rounds go to the 128x128 kernel, the remaining rows to a second launch with a small tile whose own partial round is
// short.  Measured (profiles/r02b_step_gemm_shapes_tile_sweep.log): a wash on the step's short-K shapes — the ViT fc1
// GEMM takes 177 us split vs 175 (128x128 alone) vs 166-169 (64-row tiles): with K = 384 a 128x128 workgroup spends as
// long in its prologue + GELU epilogue as two of its twelve k-tiles, and only two of them share a CU — it wins 2-3 % on
// the K >= 512 Linear shapes (fc2, teacher decoder).  Kept as a tuned-table option, not chosen by the cost model.
// Returns the first row of the tail (0: no full round exists -> everything is tail; M: no tail).
inline int msplit_row(const P& p, int nz) {
  if (nz != 1) return 0;
  const long tiles_n = (p.N + 127) / 128, tiles_m = (p.M + 127) / 128;
  const long rounds = tiles_m * tiles_n / kBodySlots;
  long rows = rounds * kBodySlots / tiles_n;                 // tile rows of the body
  if (rows >= tiles_m) return p.M;
  return (int)(rows * 128);
}

template <int OP, int TERMS>
int dispatch_tile(const P& p, int nz, hipStream_t st, int tile) {
  static const int bm[4] = {128, 64, 128, 64}, bn[4] = {128, 64, 64, 128};
  static const double eff[4] = {1.00, 0.85, 0.93, 0.93};   // relative efficiency of the tile shape
  bool split = (tile & 32) != 0 && !(tile & 128);
  tile &= 128 | 64 | 31;
  if ((tile & 31) == 0) {
    double best = 1e300;
    auto cost_of = [&](int t, long rows) {                   // busiest CU's share of tile area / efficiency
      const long blocks = ((rows + bm[t] - 1) / bm[t]) * ((p.N + bn[t] - 1) / bn[t]) * nz;
      return (double)((blocks + kCUs - 1) / kCUs) * bm[t] * bn[t] / eff[t];
    };
    for (int t = 0; t < 4; ++t) {
      const double cost = cost_of(t, p.M);
      if (cost < best * 0.999) { best = cost; tile = t + 1; }
    }
  }
  if (split && OP != ICK_OP_CONV_DGRAD_S2) {
    const int ms = msplit_row(p, nz);
    if (ms >= p.M) return launch<OP, 128, 128, 2, TERMS>(p, nz, st);
    if (ms > 0) {
      if (int rc = launch<OP, 128, 128, 2, TERMS>(p, nz, st, 0, ms)) return rc;
      return launch_tile<OP, TERMS>(p, nz, st, tile, ms, p.M);
    }
  }
  return launch_tile<OP, TERMS>(p, nz, st, tile);
}

}  // namespace

namespace ickg {
// argument checks are done by the caller; p/nz come from prepare(d, 32, ...); eligibility = glds_eligible(d)
int ICK_GLDS_ENTRY(const IckGemm* d, const P& p, int nz, hipStream_t st) {
  constexpr int TERMS = ICK_GLDS_TERMS;
  if constexpr (TERMS >= 5) {   // both operands k-contiguous: the only ops whose 16-bit image the fp32 addressing can carry
    if (d->op == ICK_OP_NT) return dispatch_tile<ICK_OP_NT, TERMS>(p, nz, st, d->tile);
    if (d->op == ICK_OP_CONV_FWD) return dispatch_tile<ICK_OP_CONV_FWD, TERMS>(p, nz, st, d->tile);
    if (d->op == ICK_OP_TN) return dispatch_tile<ICK_OP_TN, TERMS>(p, nz, st, d->tile);               // x-contiguous operands (X16)
    if (d->op == ICK_OP_CONV_WGRAD) return dispatch_tile<ICK_OP_CONV_WGRAD, TERMS>(p, nz, st, d->tile);
    return ick::fail(-1, "igemm (native 16-bit, LDS-DMA): NT, CONV_FWD, TN and CONV_WGRAD only, got op %d", d->op);
  } else if constexpr (TERMS == 4) {   // forward products, data gradients run as forward convolutions, 1x1 weight gradients (TN)
    if (d->op == ICK_OP_NT) return dispatch_tile<ICK_OP_NT, TERMS>(p, nz, st, d->tile);
    if (d->op == ICK_OP_CONV_FWD) return dispatch_tile<ICK_OP_CONV_FWD, TERMS>(p, nz, st, d->tile);
    if (d->op == ICK_OP_TN) return dispatch_tile<ICK_OP_TN, TERMS>(p, nz, st, d->tile);
    if (d->op == ICK_OP_CONV_WGRAD) return dispatch_tile<ICK_OP_CONV_WGRAD, TERMS>(p, nz, st, d->tile);   // (gather = the DMA's source addressing: shared)
    return ick::fail(-1, "igemm (f32x3, LDS-DMA): NT, CONV_FWD, TN and CONV_WGRAD only, got op %d", d->op);
  } else
  switch (d->op) {
    case ICK_OP_NT: return dispatch_tile<ICK_OP_NT, TERMS>(p, nz, st, d->tile);
    case ICK_OP_CONV_FWD: return dispatch_tile<ICK_OP_CONV_FWD, TERMS>(p, nz, st, d->tile);
#if !(ICK_EXP & 16)   // (experiment builds instantiate NT and CONV_FWD only)
    case ICK_OP_NN: return dispatch_tile<ICK_OP_NN, TERMS>(p, nz, st, d->tile);
    case ICK_OP_TN: return dispatch_tile<ICK_OP_TN, TERMS>(p, nz, st, d->tile);
    case ICK_OP_CONV_FWD_C4: return dispatch_tile<ICK_OP_CONV_FWD_C4, TERMS>(p, nz, st, d->tile);
    case ICK_OP_CONV_DGRAD: return dispatch_tile<ICK_OP_CONV_DGRAD, TERMS>(p, nz, st, d->tile);
    case ICK_OP_CONV_DGRAD_S2: return dispatch_tile<ICK_OP_CONV_DGRAD_S2, TERMS>(p, 4, st, d->tile);
    case ICK_OP_CONV_WGRAD:
      if constexpr (TERMS == 0) return dispatch_tile<ICK_OP_CONV_WGRAD, TERMS>(p, nz, st, d->tile);
      else return ick::fail(-1, "ick_gemm_bf16: op %d has no LDS-DMA variant", d->op);
#endif
    default: return ick::fail(-1, "igemm (LDS-DMA): unknown op %d", d->op);
  }
}
#if (ICK_EXP & 32)
extern "C" int ick_exp_set_stagger(int units) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_stagger), &units, sizeof(units)); }
#endif
#if (ICK_EXP & 8)
extern "C" int ick_exp_set_dbg(void* buf) { long long* b = static_cast<long long*>(buf); return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_dbg), &b, sizeof(b)); }
#endif
}  // namespace ickg
