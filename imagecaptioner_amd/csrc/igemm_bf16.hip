// igemm_bf16.hip — the implicit-GEMM family of igemm_f32.hip on the bf16 matrix cores (v_mfma_f32_32x32x16_bf16,
// 16x the fp32 MFMA rate), for the mixed-precision (AMP) regime the reference trains in
// (torch.cuda.amp.autocast + GradScaler, src/train_student_kd.py:263-290).
//
//   C[z][m][n] = act(alpha * sum_k A(m,k) B(n,k) + bias[n]) + residual[m][n]        (fp32 in HBM, fp32 accumulate)
//
// Operands stay fp32 in HBM (same descriptors, same layouts, same epilogues as the fp32 family); they are rounded
// to bf16 on the way into LDS, so one step can run in either precision without touching the host code.
//   TERMS == 1 : a ~ bf16(a)                    one MFMA per k-step         (autocast semantics, ~3 significant digits)
//   TERMS == 3 : a = hi + lo, both bf16          hi*hi + hi*lo + lo*hi       (~16 mantissa bits per operand: within
//                                                                              ~1e-5 relative of the exact fp32 product)
//
// Tiling: 256 threads = 4 waves (2x2), block tile BM x BN x 32, wave tile (BM/2)x(BN/2) as 32x32 MFMA tiles.
// Two LDS images, chosen by how the SOURCE is contiguous (coalesced 16-byte global loads either way):
//   * k-contiguous source (activations [M][K], [N][K] weights, im2col over NHWC channels): image [x][k], row pitch
//     80 B; written with ds_write_b64 (4 k), fragment = one ds_read_b128 (8 k at row = lane&31, k-half = lane>>5);
//     pitch 80 B = 5 sixteen-byte slots makes the four 16-lane groups of ds_read_b128 conflict-free;
//   * x-contiguous source ([K][N] weights, dY^T, wgrad gathers): image [k][x], row pitch 2*BX+64 B, written with
//     ds_write_b64 (4 x); the fragment is TRANSPOSED ON THE READ by two ds_read_b64_tr_b16 (4 k-rows x 16 columns per
//     16-lane group): no register transpose, no scattered 2-byte writes; pitch mod 256 B = 64 B makes the reads
//     conflict-free per 32-lane half.
// Global->LDS is register-staged (the fp32->bf16 rounding has to happen in flight): tile t+1's loads are issued
// before tile t's MFMAs and written to the other LDS buffer after them; one barrier per k-tile.
//
// IN16 instantiations (ick_gemm_h16): A and B already hold bf16 / fp16 elements in HBM.  Same thread -> element map
// (4 elements per fetch, now 8 bytes), no rounding in flight; C and the residual may be 16-bit too (P.c16 / P.r16).  This is
// the kernel of the 16-bit training regime for the products whose operands are NOT k-contiguous (weight gradients,
// stride-2 data gradients); k-contiguous ones take the LDS-DMA kernel (igemm_glds_impl.h, TERMS 5 / 6).
#include "igemm_params.h"
#include <cstdlib>
#include <type_traits>

namespace {

using namespace ickg;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
// TERMS == 2: fp16 instead of bf16 (v_mfma_f32_32x32x16_f16) — same 16-bit LDS images, different rounding + MFMA
template <int TERMS> struct Half16 { using x8 = bf16x8; using x4 = bf16x4; };
template <> struct Half16<2> { using x8 = f16x8; using x4 = f16x4; };
template <int TERMS, typename V>
__device__ __forceinline__ f32x16 mfma16(V a, V b, f32x16 c) {
  if constexpr (TERMS == 2) return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;

constexpr int BK = 32;
constexpr int NT = 256;
constexpr int KCP = BK + 8;   // row pitch (elements) of a k-contiguous image: 80 bytes

template <bool KC, int BX>
constexpr int img_elems() { return KC ? BX * KCP : BK * (BX + 32); }

// 4 fp32 -> 4 bf16 (round to nearest even) stored as 8 bytes; with SPLIT also the bf16 of the rounding residual
template <int TERMS>
__device__ __forceinline__ void store4(u16* dst, int lo_off, float4 v) {
  using h4 = typename Half16<TERMS>::x4;
  const f32x4 x = {v.x, v.y, v.z, v.w};
  const h4 h = __builtin_convertvector(x, h4);
  *reinterpret_cast<h4*>(dst) = h;
  if constexpr (TERMS == 3) {
    const f32x4 r = x - __builtin_convertvector(h, f32x4);
    *reinterpret_cast<h4*>(dst + lo_off) = __builtin_convertvector(r, h4);
  }
}

template <int TERMS>
__device__ __forceinline__ void store4(u16* dst, int, uint2 v) { *reinterpret_cast<uint2*>(dst) = v; }

template <typename H8>
__device__ __forceinline__ H8 tr_read8(const u16* p, int row_pitch) {
  typedef __attribute__((address_space(3))) s16x4* lds_p;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(p));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(p + 4 * row_pitch));
  return __builtin_bit_cast(H8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

template <int OP, int BM, int BN, int TERMS, bool IN16 = false>
__global__ __launch_bounds__(NT, 2) void igemm_bf16_kernel(const P p) {
  using T = std::conditional_t<IN16, u16, float>;          // element type of A / B in HBM
  using R4 = std::conditional_t<IN16, uint2, float4>;      // four of them in registers
  constexpr bool AK = a_kcontig(OP), BKc = b_kcontig(OP);
  constexpr bool SPLIT = TERMS == 3;
  constexpr int IMGS = SPLIT ? 2 : 1;
  constexpr int WM = BM / 2, WN = BN / 2, TM = WM / 32, TN = WN / 32;
  constexpr int PA = BM / 32, PB = BN / 32;             // 16-byte fetches per thread per k-tile
  constexpr int A_TPK = BM / 4, B_TPK = BN / 4;         // threads per k-row (x-contiguous fetch)
  constexpr int A_KR = NT / A_TPK, B_KR = NT / B_TPK;   // k-rows per pass   (x-contiguous fetch)
  constexpr int A_XP = BM + 32, B_XP = BN + 32;         // row pitch of the [k][x] images
  constexpr int A_SZ = img_elems<AK, BM>(), B_SZ = img_elems<BKc, BN>();
  constexpr int BUF = IMGS * (A_SZ + B_SZ);

  constexpr int LDS_ELEMS = 2 * BUF > 2 * BM * BN ? 2 * BUF : 2 * BM * BN;   // also holds the fp32 C tile of the staged epilogue
  __shared__ __attribute__((aligned(16))) u16 lds[LDS_ELEMS];   // [buffer][A hi, A lo, B hi, B lo]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  // XCD-aware tile order (see igemm_f32.hip): every XCD walks a contiguous range of tiles
  const int nwg = gridDim.x, q8 = nwg >> 3, r8 = nwg & 7, xcd = blockIdx.x & 7;
  const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (blockIdx.x >> 3);
  const int tile_m = wg / p.tiles_n, tile_n = wg - tile_m * p.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  int z = blockIdx.z, split = 0;
  if (p.splitk > 1) { split = z; z = 0; }
  int py = 0, px = 0, r0 = 0, s0 = 0, ns = 1, kcls = 0;
  if constexpr (OP == ICK_OP_CONV_DGRAD_S2) {   // parity class of the input pixel (see igemm_f32.hip)
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
  const T* __restrict__ Ag = reinterpret_cast<const T*>(p.A) + zo * p.sAo + zi * p.sAi;
  const T* __restrict__ Bg = reinterpret_cast<const T*>(p.B) + zo * p.sBo + zi * p.sBi;
  const long coff = zo * p.sCo + zi * p.sCi;
  const int kbeg = split * p.kps;
  const int kend = OP == ICK_OP_CONV_DGRAD_S2 ? kcls : min(p.K, kbeg + p.kps);
  const int nkt = (kend - kbeg + BK - 1) / BK;

  // ---------------------------------------------------------------- per-thread fetch state
  const T* a_ptr[PA]; bool a_ok[PA]; int a_y[PA], a_x[PA];
  const int a_k4 = (tid & 7) * 4;   // k offset inside the tile (k-contiguous fetch: 8 threads x 16 B = one 128-B row)
#pragma unroll
  for (int i = 0; i < PA; ++i) {
    if constexpr (AK) {
      const int m = m0 + i * 32 + (tid >> 3);
      a_ok[i] = m < p.M;
      if constexpr (OP == ICK_OP_NT || OP == ICK_OP_NN) {
        a_ptr[i] = Ag + (long)m * p.lda + a_k4; a_y[i] = a_x[i] = 0;
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
    } else {  // A stored [K][M]
      const int m = m0 + (tid % A_TPK) * 4;
      a_ok[i] = m < p.M;
      a_ptr[i] = Ag + m; a_y[i] = i * A_KR + tid / A_TPK; a_x[i] = 0;
    }
  }
  const T* b_ptr[PB]; bool b_ok[PB]; int b_y[PB];
  int b_r = 0, b_s = 0;
#pragma unroll
  for (int i = 0; i < PB; ++i) {
    if constexpr (BKc) {
      const int n = n0 + i * 32 + (tid >> 3);
      b_ok[i] = n < p.N;
      b_ptr[i] = Bg + (long)n * p.ldb + a_k4; b_y[i] = 0;
    } else {
      const int n = n0 + (tid % B_TPK) * 4;
      b_ok[i] = n < p.N;
      b_y[i] = i * B_KR + tid / B_TPK;
      if constexpr (OP == ICK_OP_CONV_WGRAD) {
        const int tap = n / p.Cin; const int ci = n - tap * p.Cin;
        b_r = tap / p.S; b_s = tap - b_r * p.S;
        b_ptr[i] = Bg + ci;
      } else {
        b_ptr[i] = Bg + n;
      }
    }
  }

  R4 ra[PA], rb[PB];
  unsigned amask = 0, bmask = 0;

  auto fetch = [&](int kt) {
    const int k0 = kbeg + kt * BK;
    amask = 0; bmask = 0;
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
        const int ty = a_y[i] - r, tx = a_x[i] - s;
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

  // round the fetched tile `kt` to bf16 and write it into LDS buffer `buf` (zeroing what was out of range)
  auto stash = [&](int buf, int kt) {
    const int kq = kbeg + kt * BK + a_k4;
    u16* Ab = lds + buf * BUF;
    u16* Bb = Ab + IMGS * A_SZ;
#pragma unroll
    for (int i = 0; i < PA; ++i) {
      R4 v = keep_if(ra[i], (amask >> i) & 1u);
      if constexpr (AK) {
        if constexpr (OP == ICK_OP_NT || OP == ICK_OP_NN) v = ktail(v, kq, kend);
        store4<TERMS>(Ab + (i * 32 + (tid >> 3)) * KCP + a_k4, A_SZ, v);
      } else {
        store4<TERMS>(Ab + (i * A_KR + tid / A_TPK) * A_XP + (tid % A_TPK) * 4, A_SZ, v);
      }
    }
#pragma unroll
    for (int i = 0; i < PB; ++i) {
      R4 v = keep_if(rb[i], (bmask >> i) & 1u);
      if constexpr (BKc) {
        v = ktail(v, kq, kend);
        store4<TERMS>(Bb + (i * 32 + (tid >> 3)) * KCP + a_k4, B_SZ, v);
      } else {
        store4<TERMS>(Bb + (i * B_KR + tid / B_TPK) * B_XP + (tid % B_TPK) * 4, B_SZ, v);
      }
    }
  };

  // MFMA operand fragment of k-step s (16 k): lane l holds X[row l&31][k = 8*(l>>5) + 0..7]
  const int frow = lane & 31, fk = lane >> 5;
  const int tg = lane >> 4, tq = (lane >> 2) & 3, tp = lane & 3;   // transposed read: group, block row, 4-column piece
  using h8 = typename Half16<TERMS>::x8;
  auto frag_a = [&](int buf, int img, int s, int i) -> h8 {
    const u16* base = lds + buf * BUF + img * A_SZ;
    if constexpr (AK) {
      return *reinterpret_cast<const h8*>(base + (wm * WM + i * 32 + frow) * KCP + s * 16 + fk * 8);
    } else {
      return tr_read8<h8>(base + (s * 16 + (tg >> 1) * 8 + tq) * A_XP + wm * WM + i * 32 + (tg & 1) * 16 + tp * 4, A_XP);
    }
  };
  auto frag_b = [&](int buf, int img, int s, int j) -> h8 {
    const u16* base = lds + buf * BUF + IMGS * A_SZ + img * B_SZ;
    if constexpr (BKc) {
      return *reinterpret_cast<const h8*>(base + (wn * WN + j * 32 + frow) * KCP + s * 16 + fk * 8);
    } else {
      return tr_read8<h8>(base + (s * 16 + (tg >> 1) * 8 + tq) * B_XP + wn * WN + j * 32 + (tg & 1) * 16 + tp * 4, B_XP);
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  if (nkt > 0) {
    fetch(0);
    stash(0, 0);
  }
  __syncthreads();

  for (int kt = 0; kt < nkt; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nkt) fetch(kt + 1);
    h8 av[2][IMGS][TM], bv[2][IMGS][TN];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int g = 0; g < IMGS; ++g) {
#pragma unroll
        for (int i = 0; i < TM; ++i) av[s][g][i] = frag_a(buf, g, s, i);
#pragma unroll
        for (int j = 0; j < TN; ++j) bv[s][g][j] = frag_b(buf, g, s, j);
      }
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          if constexpr (SPLIT) {   // small terms first
            acc[i][j] = mfma16<TERMS>(av[s][1][i], bv[s][0][j], acc[i][j]);
            acc[i][j] = mfma16<TERMS>(av[s][0][i], bv[s][1][j], acc[i][j]);
          }
          acc[i][j] = mfma16<TERMS>(av[s][0][i], bv[s][0][j], acc[i][j]);
        }
    if (kt + 1 < nkt) stash(buf ^ 1, kt + 1);
    __syncthreads();
  }

  // ---------------------------------------------------------------- epilogue (identical to the fp32 family)
  // C/D layout of the 32x32 tile: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
  float* __restrict__ Cg = p.C + coff;
  const float* __restrict__ Rg = (p.residual && split == 0) ? p.residual + coff : nullptr;
  const float* __restrict__ biasp = (p.bias && split == 0) ? p.bias : nullptr;
  const int mode = p.splitk > 1 ? 2 : (p.accumulate ? 1 : 0);
  const int act = p.act & 15;
  const bool post = (p.act & ICK_ACT_POST_RESIDUAL) != 0;   // activation after the residual (staged epilogue only)
  const float alpha = p.alpha;
  if (p.ep_vec && mode != 2) {
    // LDS-staged epilogue (see igemm_f32_glds.hip): registers -> LDS tile [BM][BN] -> 16-byte row stores; at bf16 MFMA
    // rates the epilogue is a large share of a short-K convolution
    __syncthreads();
    float* ct = reinterpret_cast<float*>(lds);
    const int tid = threadIdx.x;
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
          ct[row * BN + nl] = act_c<ACT>(u);
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
    __syncthreads();
    constexpr int C4 = BN / 4;
    for (int c = tid; c < BM * C4; c += NT) {
      const int row = c / C4, col = (c - row * C4) * 4;
      const int m = m0 + row, n = n0 + col;
      if (m < p.M && n < p.N) {
        long mr = m;
        if constexpr (OP == ICK_OP_CONV_DGRAD_S2) {
          const int w2 = p.W >> 1; const int hw = (p.H >> 1) * w2; const int b = m / hw; const int q = m - b * hw;
          mr = ((long)b * p.H + 2 * (q / w2) + py) * p.W + 2 * (q % w2) + px;
        }
        float4 v = *reinterpret_cast<const float4*>(ct + row * BN + col);
        using h4 = typename Half16<TERMS>::x4;
        if (Rg) {
          float4 q;
          if (IN16 && p.r16) q = ld4h<h4>(p.residual, coff + mr * p.ldr + n);
          else q = *reinterpret_cast<const float4*>(Rg + mr * p.ldr + n);
          v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
        }
        if (post) { v.x = act_fn(v.x, act); v.y = act_fn(v.y, act); v.z = act_fn(v.z, act); v.w = act_fn(v.w, act); }
        if (IN16 && p.c16) {                 // 16-bit C: 8-byte row chunks
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
    return;
  }
  auto epilogue = [&](auto full_tag, auto act_tag) {
    constexpr bool FULL = decltype(full_tag)::value;
    constexpr int ACT = decltype(act_tag)::value;     // compile-time activation: see igemm_params.h act_c
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
          if (FULL || (m < p.M && nok)) {
            float v = acc[i][j][r] * alpha;
            ssum += v; ssq = fmaf(v, v, ssq);
            if constexpr (ACT < 0) v = act_fn(v + bias, act); else v = act_c<ACT>(v + bias);
            long mr = m;
            if constexpr (OP == ICK_OP_CONV_DGRAD_S2) {
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
      if (p.stat_sum) {
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

template <int OP, int BM, int BN, int TERMS, bool IN16>
int launch(const P& p0, int nz, hipStream_t st) {
  P p = p0;
  p.tiles_n = (p.N + BN - 1) / BN;
  p.ep_vec = p.N % 4 == 0 && p.ldc % 4 == 0 && (p.sCo | p.sCi) % 4 == 0 && ick::aligned16(p.C) &&
             (!p.residual || (p.ldr % 4 == 0 && ick::aligned16(p.residual)));
  if ((p.col_scale || (p.act & ICK_ACT_POST_RESIDUAL)) && !(p.ep_vec && p.splitk == 1))
    return ick::fail(-1, "igemm: col_scale / ICK_ACT_POST_RESIDUAL need 16-byte aligned C rows (N %% 4, ldc %% 4) and no split-K");
  if ((p.c16 || p.r16) && !(IN16 && p.ep_vec && p.splitk == 1))
    return ick::fail(-1, "igemm: a 16-bit C / residual needs the native 16-bit kernel, N %% 4, ldc %% 4, ldr %% 4 and no split-K");
  dim3 grid(p.tiles_n * ((p.M + BM - 1) / BM), 1, nz);
  ICK_LAUNCH((igemm_bf16_kernel<OP, BM, BN, TERMS, IN16>), grid, dim3(NT), 0, st, p);
  return ick::launch_status("igemm_bf16");
}

// Tile choice: the same wave-quantisation model as the fp32 family (max work per CU), with a flatter efficiency
// ladder: at bf16 MFMA rates every tile shape is fed by the fp32 operand stream from L2, not by the matrix pipe.
// The split (TERMS == 3) 128x128 tile would need 80 KB of LDS per workgroup; it is not built.
template <int OP, int TERMS, bool IN16>
int dispatch_tile(const P& p, int nz, hipStream_t st, int tile) {
  constexpr bool BIG = TERMS != 3;
  if (tile == 0 || (!BIG && tile == 1)) {
    static const int bm[4] = {128, 64, 128, 64}, bn[4] = {128, 64, 64, 128};
    static const double eff[4] = {1.00, 0.70, 0.85, 0.85};
    double best = 1e300;
    tile = 2;
    for (int t = BIG ? 0 : 1; t < 4; ++t) {
      const long blocks = (long)((p.M + bm[t] - 1) / bm[t]) * ((p.N + bn[t] - 1) / bn[t]) * nz;
      const long per_cu = (blocks + 255) / 256;
      const double cost = (double)per_cu * bm[t] * bn[t] / eff[t];
      if (cost < best * 0.999) { best = cost; tile = t + 1; }
    }
  }
  switch (tile) {
    case 2: return launch<OP, 64, 64, TERMS, IN16>(p, nz, st);
    case 3: return launch<OP, 128, 64, TERMS, IN16>(p, nz, st);
    case 4: return launch<OP, 64, 128, TERMS, IN16>(p, nz, st);
    default:
      if constexpr (BIG) return launch<OP, 128, 128, TERMS, IN16>(p, nz, st);
      else return launch<OP, 128, 64, TERMS, IN16>(p, nz, st);
  }
}

template <int TERMS, bool IN16 = false>
int run(const IckGemm* d, const P& p, int nz, hipStream_t st) {
  switch (d->op) {
    case ICK_OP_NT: return dispatch_tile<ICK_OP_NT, TERMS, IN16>(p, nz, st, d->tile);
    case ICK_OP_NN: return dispatch_tile<ICK_OP_NN, TERMS, IN16>(p, nz, st, d->tile);
    case ICK_OP_TN: return dispatch_tile<ICK_OP_TN, TERMS, IN16>(p, nz, st, d->tile);
    case ICK_OP_CONV_FWD: return dispatch_tile<ICK_OP_CONV_FWD, TERMS, IN16>(p, nz, st, d->tile);
    case ICK_OP_CONV_FWD_C4: return dispatch_tile<ICK_OP_CONV_FWD_C4, TERMS, IN16>(p, nz, st, d->tile);
    case ICK_OP_CONV_DGRAD: return dispatch_tile<ICK_OP_CONV_DGRAD, TERMS, IN16>(p, nz, st, d->tile);
    case ICK_OP_CONV_DGRAD_S2: return dispatch_tile<ICK_OP_CONV_DGRAD_S2, TERMS, IN16>(p, 4, st, d->tile);
    case ICK_OP_CONV_WGRAD: return dispatch_tile<ICK_OP_CONV_WGRAD, TERMS, IN16>(p, nz, st, d->tile);
    default: return ick::fail(-1, "ick_gemm_bf16: unknown op %d", d->op);
  }
}

}  // namespace

namespace ickg {
bool glds_eligible(const IckGemm* d);                                                     // igemm_f32_glds.hip
int run_glds_bf16(const IckGemm* d, int terms, const P& p, int nz, hipStream_t st);       // igemm_bf16_glds_impl.h


// shape rules shared by ick_gemm_bf16 and ick_gemm_h16; *to_f32 = 1: the channel count has no 32-wide k-tile (exact-fp32 kernel)
int check_bf16_shapes(const IckGemm* d, const P& p, int nz, int* to_f32, const char* who) {
  *to_f32 = 0;
  switch (d->op) {
    case ICK_OP_NT: case ICK_OP_NN: case ICK_OP_TN:
      ICK_REQUIRE(p.lda % 4 == 0 && p.ldb % 4 == 0, "%s: lda, ldb must be multiples of 4 (rows readable up to roundup4)", who);
      ICK_REQUIRE((p.sAo | p.sAi | p.sBo | p.sBi) % 4 == 0, "%s: batch strides must be multiples of 4", who);
      break;
    case ICK_OP_CONV_FWD:
      ICK_REQUIRE(p.M == p.Nb * p.Ho * p.Wo && p.N == p.Cout && p.K == p.R * p.S * p.Cin && p.ldb == p.K,
                  "CONV_FWD: M/N/K do not match the geometry");
      if (p.Cin % BK != 0) *to_f32 = 1;   // a k-tile must not straddle two taps: exact-fp32 kernel (BK 16)
      break;
    case ICK_OP_CONV_FWD_C4:
      ICK_REQUIRE(p.Cin == 4, "CONV_FWD_C4: Cin must be 4");
      ICK_REQUIRE(p.M == p.Nb * p.Ho * p.Wo && p.N == p.Cout && p.K == p.R * p.S * 4 && p.ldb == p.K,
                  "CONV_FWD_C4: M/N/K do not match the geometry");
      break;
    case ICK_OP_CONV_DGRAD:
      ICK_REQUIRE(p.Cin % 4 == 0, "CONV_DGRAD: Cin %% 4 required");
      ICK_REQUIRE(p.M == p.Nb * p.H * p.W && p.N == p.Cin && p.K == p.R * p.S * p.Cout,
                  "CONV_DGRAD: M/N/K do not match the geometry");
      if (p.Cout % BK != 0) *to_f32 = 1;
      break;
    case ICK_OP_CONV_DGRAD_S2:
      ICK_REQUIRE(p.stride == 2 && p.H % 2 == 0 && p.W % 2 == 0, "CONV_DGRAD_S2: stride 2 and even H, W required");
      ICK_REQUIRE(p.Cin % 4 == 0, "CONV_DGRAD_S2: Cin %% 4 required");
      ICK_REQUIRE(p.M == p.Nb * (p.H / 2) * (p.W / 2) && p.N == p.Cin && p.K == p.R * p.S * p.Cout && nz == 1 &&
                  p.splitk == 1, "CONV_DGRAD_S2: M must be the rows of ONE parity class; no batching / split-K");
      if (p.Cout % BK != 0) *to_f32 = 1;
      break;
    case ICK_OP_CONV_WGRAD:
      ICK_REQUIRE(p.Cout % 4 == 0 && p.Cin % 4 == 0, "CONV_WGRAD: Cout %% 4 and Cin %% 4 required");
      ICK_REQUIRE(p.M == p.Cout && p.N == p.R * p.S * p.Cin && p.K == p.Nb * p.Ho * p.Wo && p.lda == p.Cout,
                  "CONV_WGRAD: M/N/K do not match the geometry");
      break;
    default:
      return ick::fail(-1, "%s: unknown op %d", who, d->op);
  }
  return 0;
}

// register-staged kernel on native 16-bit operands (entry: igemm_h16.hip); shapes already checked by the caller
int run_regs_h16(const IckGemm* d, int fp16, const P& p, int nz, hipStream_t st) {
  IckGemm dd = *d; dd.tile &= 15;
  return fp16 ? run<2, true>(&dd, p, nz, st) : run<1, true>(&dd, p, nz, st);
}
}  // namespace ickg

extern "C" int ick_gemm_bf16(const IckGemm* d, int terms, void* stream) {
  using namespace ickg;
  ICK_REQUIRE(terms >= 1 && terms <= 4, "ick_gemm_bf16: terms must be 1 (bf16), 2 (fp16), 3 (split bf16) or 4 (fp32 by three fp16 products), got %d", terms);
  ICK_REQUIRE(d != nullptr, "ick_gemm_bf16: null descriptor");
  ICK_REQUIRE(d->io16 == 0 || terms < 3, "ick_gemm_bf16: a 16-bit C / residual needs terms 1 (bf16) or 2 (fp16)");
  if (terms == 4) {   // fp32-grade results: the three-product kernel where it exists (k-contiguous forward products), exact fp32 MFMA elsewhere
    const bool fwd = (d->op == ICK_OP_NT || d->op == ICK_OP_CONV_FWD || d->op == ICK_OP_TN || d->op == ICK_OP_CONV_WGRAD) && !(d->tile & 256) &&
                     glds_eligible(d);
    if (!fwd) return ick_gemm_f32(d, stream);
    P p4; int nz4 = 1;
    if (int rc = prepare(d, BK, p4, nz4, "ick_gemm_bf16")) return rc;
    IckGemm d4 = *d; d4.tile &= 255;
    return run_glds_bf16(&d4, 4, p4, nz4, static_cast<hipStream_t>(stream));
  }
  P p; int nz = 1;
  if (int rc = prepare(d, BK, p, nz, "ick_gemm_bf16")) return rc;
  hipStream_t st = static_cast<hipStream_t>(stream);
  int to_f32 = 0;
  if (int rc = check_bf16_shapes(d, p, nz, &to_f32, "ick_gemm_bf16")) return rc;
  if (to_f32) return ick_gemm_f32(d, stream);
  // the LDS-DMA variant (igemm_bf16_glds_impl.h) is the default where it exists; IckGemm.tile bit 8 or ICK_NO_GLDS_BF16=1
  // selects this file's register-staged kernel (conv wgrad always: both operands x-contiguous + per-lane gather)
  static const bool no_glds = [] { const char* e = getenv("ICK_NO_GLDS_BF16"); return e && e[0] == '1'; }();
  if ((d->io16 || (!no_glds && !(d->tile & 256))) && d->op != ICK_OP_CONV_WGRAD && glds_eligible(d)) {   // (a 16-bit C exists on the LDS-DMA kernel only)
    IckGemm dd = *d; dd.tile &= 255;
    return run_glds_bf16(&dd, terms, p, nz, st);
  }
  IckGemm dd = *d; dd.tile &= 15;      // the register-staged family has the four plain tile shapes only
  return terms == 3 ? run<3>(&dd, p, nz, st) : terms == 2 ? run<2>(&dd, p, nz, st) : run<1>(&dd, p, nz, st);
}
