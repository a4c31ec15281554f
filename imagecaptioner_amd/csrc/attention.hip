// attention.hip — fused forward attention softmax(Q K^T / sqrt(d)) V on the fp32 matrix cores (gfx950), d = 64.
//
// Used by the frozen teacher (no backward needed): ViT-S/16 self-attention (197 tokens x 6 heads,
// timm blocks; /root/reference/src/teacher_model.py:82) and the Transformer-decoder self/cross attention
// (nn.MultiheadAttention inside nn.TransformerDecoderLayer, /root/reference/src/teacher_model.py:60-67), and by
// the student's refinement block in eval mode.  The (B,H,Lq,Lk) score matrix never reaches HBM
// (unfused: 2 x 60 MB per ViT layer written and re-read).
//
// Structure (64-wide wavefronts, v_mfma_f32_32x32x2_f32, exact fp32):
//   * one workgroup per (batch, head); wave w owns queries [32w, 32w+32) of that head;
//   * keys/values stream through LDS in chunks of 32 keys (double buffered, register-staged prefetch);
//   * S^T = K Q^T ("swapped" product): the accumulator of a 32x32 tile then holds, per lane, ONE query column
//     and 16 of the 32 keys in its registers, so the online-softmax row statistics are register reductions plus a
//     single lane<->lane+32 exchange, and the per-query rescale is a per-lane scalar;
//   * O^T += V^T P^T takes that accumulator DIRECTLY as the MFMA B operand: the k-pairing of the 32x32x2 MFMA is
//     chosen as (key held by lane half 0, key held by lane half 1) of register r, so no data moves between the
//     two products (the V^T fragment is read from LDS in the matching key order);
//   * K chunk rows are padded to 65 floats (the A-fragment read walks keys across lanes), V rows to 64.
#include "ick_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int D = 64;       // head dim
constexpr int KC = 32;      // keys per chunk
constexpr int KP = D + 1;   // K row pitch in LDS

struct AP {
  const float* q; const float* k; const float* v; float* o;
  long qld, kld, vld, old;          // row pitches (elements)
  long qbs, kbs, vbs, obs;          // batch strides (elements); head h adds h*D columns
  int H, Lq, Lk, causal;
  float scale;
};

// (seven-wave workgroups — the ViT's 197 tokens — are held to 128 VGPRs = four waves per SIMD: TWO workgroups per CU, so that
//  the 384 (batch, head) workgroups of a B = 64 launch are resident at once instead of running as 256 + 128)
template <int NW>
__global__ __launch_bounds__(NW * 64, NW == 7 ? 4 : 1) void attn_fwd_kernel(const AP p) {
  __shared__ __attribute__((aligned(16))) float Ks[2][KC][KP];
  __shared__ __attribute__((aligned(16))) float Vs[2][KC][D];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int col = lane & 31, half = lane >> 5;
  const int b = blockIdx.x / p.H, h = blockIdx.x - b * p.H;
  const float* Q = p.q + b * p.qbs + h * D;
  const float* K = p.k + b * p.kbs + h * D;
  const float* V = p.v + b * p.vbs + h * D;
  float* O = p.o + b * p.obs + h * D;
  const int q0 = (blockIdx.y * NW + wave) * 32;
  const int qrow = q0 + col;                       // this lane's query
  const bool qok = qrow < p.Lq;

  // B operand of S^T = K Q^T: lane (query col, half) needs Q[query][d = 2kk + half], kk = 0..31, pre-scaled
  float qf[32];
  const float qscale = p.scale * 1.44269504088896340736f;   // scores in the log2 domain: the softmax is exp2 (one v_exp_f32 each)
#pragma unroll
  for (int kk = 0; kk < 32; ++kk) qf[kk] = qok ? Q[(long)qrow * p.qld + 2 * kk + half] * qscale : 0.f;

  f32x16 ot[2];                                    // O^T tiles: rows d = 32t + (r&3)+8(r>>2)+4half, column = query
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) ot[t][r] = 0.f;
  float mrun = -INFINITY, lrun = 0.f;

  // chunk staging: NW*64 threads move 32 keys x 64 floats of K and V each (float4 per thread per pass)
  constexpr int NTH = NW * 64;
  constexpr int F4 = KC * D / 4;                   // 512 float4 per operand
  constexpr int PASS = (F4 + NTH - 1) / NTH;
  float4 rk[PASS], rv[PASS];
  const int nchunks = (p.Lk + KC - 1) / KC;
  auto fetch = [&](int c) {
#pragma unroll
    for (int i = 0; i < PASS; ++i) {
      const int f = tid + i * NTH;
      const int key = c * KC + f / (D / 4), d4 = (f % (D / 4)) * 4;
      const bool ok = f < F4 && key < p.Lk;
      rk[i] = ok ? *reinterpret_cast<const float4*>(K + (long)key * p.kld + d4) : make_float4(0, 0, 0, 0);
      rv[i] = ok ? *reinterpret_cast<const float4*>(V + (long)key * p.vld + d4) : make_float4(0, 0, 0, 0);
    }
  };
  auto stash = [&](int buf) {
#pragma unroll
    for (int i = 0; i < PASS; ++i) {
      const int f = tid + i * NTH;
      if (f < F4) {
        const int key = f / (D / 4), d4 = (f % (D / 4)) * 4;
        Ks[buf][key][d4 + 0] = rk[i].x; Ks[buf][key][d4 + 1] = rk[i].y;
        Ks[buf][key][d4 + 2] = rk[i].z; Ks[buf][key][d4 + 3] = rk[i].w;
        *reinterpret_cast<float4*>(&Vs[buf][key][d4]) = rv[i];
      }
    }
  };

  fetch(0);
  stash(0);
  __syncthreads();
  for (int c = 0; c < nchunks; ++c) {
    const int buf = c & 1;
    if (c + 1 < nchunks) fetch(c + 1);
    // ---- S^T (keys x queries) = K_chunk (32 x 64) . Q^T (64 x 32)
    f32x16 st;
#pragma unroll
    for (int r = 0; r < 16; ++r) st[r] = 0.f;
#pragma unroll
    for (int kk = 0; kk < 32; ++kk)
      st = __builtin_amdgcn_mfma_f32_32x32x2f32(Ks[buf][col][2 * kk + half], qf[kk], st, 0, 0, 0);
    // ---- online softmax over this chunk's keys; lane holds keys (r&3) + 8(r>>2) + 4half of column `col`
    float cmax = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = c * KC + (r & 3) + 8 * (r >> 2) + 4 * half;
      const bool ok = key < p.Lk && (!p.causal || key <= qrow);
      st[r] = ok ? st[r] : -INFINITY;
      cmax = fmaxf(cmax, st[r]);
    }
    cmax = fmaxf(cmax, __shfl_xor(cmax, 32));
    const float mnew = fmaxf(mrun, cmax);
    const float msafe = mnew == -INFINITY ? 0.f : mnew;           // fully masked so far: keep everything at 0
    const float alpha = __builtin_amdgcn_exp2f(mrun - msafe);                       // exp(-inf) = 0 on the first chunk
    float csum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) { st[r] = __builtin_amdgcn_exp2f(st[r] - msafe); csum += st[r]; }
    csum += __shfl_xor(csum, 32);
    lrun = lrun * alpha + csum;
    mrun = mnew;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) ot[t][r] *= alpha;
    // ---- O^T (d x queries) += V^T (d x keys) . P^T (keys x queries); P^T register r IS the B fragment of the
    // k-pair {key (r&3)+8(r>>2) [half 0], that key + 4 [half 1]}; the A fragment reads V in the same key order
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = (r & 3) + 8 * (r >> 2) + 4 * half;
#pragma unroll
      for (int t = 0; t < 2; ++t)
        ot[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(Vs[buf][key][32 * t + col], st[r], ot[t], 0, 0, 0);
    }
    if (c + 1 < nchunks) stash(buf ^ 1);
    __syncthreads();
  }
  // ---- normalise and store O[query][d]: registers 4g..4g+3 of a tile are 4 consecutive d -> one float4 store
  if (qok) {
    const float inv = lrun > 0.f ? 1.f / lrun : 0.f;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d0 = 32 * t + 8 * g + 4 * half;
        *reinterpret_cast<float4*>(O + (long)qrow * p.old + d0) =
            make_float4(ot[t][4 * g] * inv, ot[t][4 * g + 1] * inv, ot[t][4 * g + 2] * inv, ot[t][4 * g + 3] * inv);
      }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// The same attention with fp32-GRADE products from three fp16 MFMAs each ("f32x3", round 3; igemm_glds_impl.h TERMS 4 has
// the derivation): x = hi + 2^-11 lo', hi = fp16(x), lo' = fp16(2^11 (x - hi)); hi*hi into one accumulator, hi*lo' + lo'*hi
// into a second one that is folded in with 2^-11.  v_mfma_f32_32x32x16_f16 multiplies 16 k per 32 cycles where the fp32 MFMA
// multiplies 2 per 64: a chunk of 32 keys costs 24 MFMAs x 32 cycles per wave instead of 64 x 64.
//   * K and V are split ONCE, when a chunk is stashed: LDS holds K as two fp16 planes [key][64 d] (pitch 144 B: the four
//     16-lane groups of a ds_read_b128 hit 16 distinct 16-byte slots) and V TRANSPOSED as two planes [d][32 keys] (pitch
//     72 B), so that both A fragments are 8 consecutive halves (K: 8 d of one key; V^T: 2 x 4 keys of one d);
//   * S^T = K Q^T keeps the 32x32 accumulator layout of the fp32 kernel (lane = one query column, 16 of the 32 keys in its
//     registers), so the online softmax is unchanged;
//   * O^T += V^T P^T: the B fragment of k-step s is the lane's registers 8s..8s+7, i.e. keys {16s + 4 half + 0..3,
//     16s + 8 + 4 half + 0..3} — the A fragment reads V^T at exactly those keys (two ds_read_b64), so P never moves
//     between lanes here either; it is split in registers (probabilities lie in [0, 1]: pre-scaled by 2^11 they sit in the
//     range fp16 is best at, and this product needs a single accumulator — see the comment at `ot`).
// Operands must lie in fp16's range (LayerNorm'd activations through a Linear: they do); the host layer calls this entry only
// under ops.precision("f32x3").
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef float f32x8v __attribute__((ext_vector_type(8)));
constexpr int KHP = 72;     // K plane row pitch in halves (144 B)
constexpr int VTP = 36;     // V^T plane row pitch in halves (72 B)

__device__ __forceinline__ void split4(const float4 v, f16x4& hi, f16x4& lo) {
  const f32x4v x = {v.x, v.y, v.z, v.w};
  hi = __builtin_convertvector(x, f16x4);
  lo = __builtin_convertvector((x - __builtin_convertvector(hi, f32x4v)) * 2048.f, f16x4);
}

template <int NW>
__global__ __launch_bounds__(NW * 64) void attn_fwd_x3_kernel(const AP p) {
  __shared__ __attribute__((aligned(16))) _Float16 Kh[2][KC][KHP], Kl[2][KC][KHP];
  __shared__ __attribute__((aligned(16))) _Float16 Vh[2][D][VTP], Vl[2][D][VTP];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int col = lane & 31, half = lane >> 5;
  const int b = blockIdx.x / p.H, h = blockIdx.x - b * p.H;
  const float* Q = p.q + b * p.qbs + h * D;
  const float* K = p.k + b * p.kbs + h * D;
  const float* V = p.v + b * p.vbs + h * D;
  float* O = p.o + b * p.obs + h * D;
  const int q0 = (blockIdx.y * NW + wave) * 32;
  const int qrow = q0 + col;
  const bool qok = qrow < p.Lq;

  // B operand of S^T = K Q^T, k-step s (16 d): lane (query col, half) supplies d = 16 s + 8 half + 0..7, pre-scaled
  const float qscale = p.scale * 1.44269504088896340736f;   // log2 domain, as in attn_fwd_kernel
  f16x8 qh[4], ql[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    f32x8v x;
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = qok ? Q[(long)qrow * p.qld + 16 * s + 8 * half + j] * qscale : 0.f;
    qh[s] = __builtin_convertvector(x, f16x8);
    ql[s] = __builtin_convertvector((x - __builtin_convertvector(qh[s], f32x8v)) * 2048.f, f16x8);
  }

  // O^T tiles, ONE accumulator at scale 2^11: the probabilities are <= 1, so P' = 2^11 p = Ph + Pl splits with an UNSCALED low
  // part that stays a normal fp16 number down to p = 6e-5, and with ph = fp16(p) = 2^-11 Ph all three products
  // V_h Ph + V_h Pl + V_l' ph carry the same factor 2^11 (V_l' = 2^11 (v - V_h) as everywhere): 32 registers fewer than a
  // second accumulator (154 VGPRs; forcing 128 for two seven-wave workgroups per CU spills 18 and gains nothing: measured).
  f32x16 ot[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) ot[t][r] = 0.f;
  float mrun = -INFINITY, lrun = 0.f;

  constexpr int NTH = NW * 64;
  constexpr int F4 = KC * D / 4;
  constexpr int PASS = (F4 + NTH - 1) / NTH;
  float4 rk[PASS], rv[PASS];
  const int nchunks = (p.Lk + KC - 1) / KC;
  auto fetch = [&](int c) {
#pragma unroll
    for (int i = 0; i < PASS; ++i) {
      const int f = tid + i * NTH;
      const int key = c * KC + f / (D / 4), d4 = (f % (D / 4)) * 4;
      const bool ok = f < F4 && key < p.Lk;
      rk[i] = ok ? *reinterpret_cast<const float4*>(K + (long)key * p.kld + d4) : make_float4(0, 0, 0, 0);
      rv[i] = ok ? *reinterpret_cast<const float4*>(V + (long)key * p.vld + d4) : make_float4(0, 0, 0, 0);
    }
  };
  auto stash = [&](int buf) {
#pragma unroll
    for (int i = 0; i < PASS; ++i) {
      const int f = tid + i * NTH;
      if (f < F4) {
        const int key = f / (D / 4), d4 = (f % (D / 4)) * 4;
        f16x4 hi, lo;
        split4(rk[i], hi, lo);
        *reinterpret_cast<f16x4*>(&Kh[buf][key][d4]) = hi;
        *reinterpret_cast<f16x4*>(&Kl[buf][key][d4]) = lo;
        split4(rv[i], hi, lo);
#pragma unroll
        for (int j = 0; j < 4; ++j) { Vh[buf][d4 + j][key] = hi[j]; Vl[buf][d4 + j][key] = lo[j]; }
      }
    }
  };

  fetch(0);
  stash(0);
  __syncthreads();
  for (int c = 0; c < nchunks; ++c) {
    const int buf = c & 1;
    if (c + 1 < nchunks) fetch(c + 1);
    // ---- S^T (keys x queries) = K_chunk (32 x 64) . Q^T (64 x 32): 4 k-steps x 3 MFMAs
    f32x16 st, sx;
#pragma unroll
    for (int r = 0; r < 16; ++r) { st[r] = 0.f; sx[r] = 0.f; }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const f16x8 kh = *reinterpret_cast<const f16x8*>(&Kh[buf][col][16 * s + 8 * half]);
      const f16x8 kl = *reinterpret_cast<const f16x8*>(&Kl[buf][col][16 * s + 8 * half]);
      sx = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl, qh[s], sx, 0, 0, 0);
      sx = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, ql[s], sx, 0, 0, 0);
      st = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, qh[s], st, 0, 0, 0);
    }
    // ---- online softmax over this chunk's keys; lane holds keys (r&3) + 8(r>>2) + 4half of column `col`
    float cmax = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = c * KC + (r & 3) + 8 * (r >> 2) + 4 * half;
      const bool ok = key < p.Lk && (!p.causal || key <= qrow);
      st[r] = ok ? st[r] + sx[r] * (1.f / 2048.f) : -INFINITY;
      cmax = fmaxf(cmax, st[r]);
    }
    cmax = fmaxf(cmax, __shfl_xor(cmax, 32));
    const float mnew = fmaxf(mrun, cmax);
    const float msafe = mnew == -INFINITY ? 0.f : mnew;
    const float alpha = __builtin_amdgcn_exp2f(mrun - msafe);
    float csum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) { st[r] = __builtin_amdgcn_exp2f(st[r] - msafe); csum += st[r]; }
    csum += __shfl_xor(csum, 32);
    lrun = lrun * alpha + csum;
    mrun = mnew;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) ot[t][r] *= alpha;
    // ---- O^T (d x queries) += V^T (d x keys) . P^T (keys x queries): 2 k-steps x 2 tiles x 3 MFMAs
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      f32x8v pv;
#pragma unroll
      for (int j = 0; j < 8; ++j) pv[j] = st[8 * s + j];
      const f16x8 ph = __builtin_convertvector(pv, f16x8);                  // fp16(p) = 2^-11 Ph
      const f32x8v pw = pv * 2048.f;
      const f16x8 Ph = __builtin_convertvector(pw, f16x8);
      const f16x8 Pl = __builtin_convertvector(pw - __builtin_convertvector(Ph, f32x8v), f16x8);
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const _Float16* vh = &Vh[buf][32 * t + col][16 * s + 4 * half];
        const _Float16* vl = &Vl[buf][32 * t + col][16 * s + 4 * half];
        const f16x4 vh0 = *reinterpret_cast<const f16x4*>(vh), vh1 = *reinterpret_cast<const f16x4*>(vh + 8);
        const f16x4 vl0 = *reinterpret_cast<const f16x4*>(vl), vl1 = *reinterpret_cast<const f16x4*>(vl + 8);
        const f16x8 vhf = __builtin_shufflevector(vh0, vh1, 0, 1, 2, 3, 4, 5, 6, 7);
        const f16x8 vlf = __builtin_shufflevector(vl0, vl1, 0, 1, 2, 3, 4, 5, 6, 7);
        ot[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vlf, ph, ot[t], 0, 0, 0);
        ot[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vhf, Pl, ot[t], 0, 0, 0);
        ot[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vhf, Ph, ot[t], 0, 0, 0);
      }
    }
    if (c + 1 < nchunks) stash(buf ^ 1);
    __syncthreads();
  }
  if (qok) {
    const float inv = lrun > 0.f ? (1.f / 2048.f) / lrun : 0.f;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d0 = 32 * t + 8 * g + 4 * half;
        float o4[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) o4[j] = ot[t][4 * g + j] * inv;
        *reinterpret_cast<float4*>(O + (long)qrow * p.old + d0) = make_float4(o4[0], o4[1], o4[2], o4[3]);
      }
  }
}

}  // namespace

// q, k, v, o: row matrices; element (b, row, h, c) lives at base + b*bs + row*ld + h*64 + c.
extern "C" int ick_attention_fwd_d64(const float* q, int64_t qld, int64_t qbs, const float* k, int64_t kld, int64_t kbs,
                                     const float* v, int64_t vld, int64_t vbs, float* o, int64_t old, int64_t obs,
                                     int B, int H, int Lq, int Lk, int causal, float scale, void* stream) {
  ICK_REQUIRE(q && k && v && o && B > 0 && H > 0 && Lq > 0 && Lk > 0, "ick_attention_fwd_d64: bad arguments");
  ICK_REQUIRE((kld | vld | old | kbs | vbs | obs) % 4 == 0 && ick::aligned16(k) && ick::aligned16(v) && ick::aligned16(o),
              "ick_attention_fwd_d64: K/V/O rows must be 16-byte aligned");
  AP p{q, k, v, o, qld, kld, vld, old, qbs, kbs, vbs, obs, H, Lq, Lk, causal, scale};
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int tiles = (Lq + 31) / 32;
  if (tiles == 7) {
    // the ViT's 197 tokens: seven waves in ONE workgroup per (batch, head) share each K / V chunk — with four-wave
    // workgroups the second one (69 valid queries of 128) re-stages every chunk for 2.2 waves of work
    ICK_LAUNCH((attn_fwd_kernel<7>), dim3(B * H, 1), dim3(448), 0, st, p);
  } else if (tiles >= 4) {
    ICK_LAUNCH((attn_fwd_kernel<4>), dim3(B * H, (tiles + 3) / 4), dim3(256), 0, st, p);
  } else if (tiles >= 2) {
    ICK_LAUNCH((attn_fwd_kernel<2>), dim3(B * H, (tiles + 1) / 2), dim3(128), 0, st, p);
  } else {
    ICK_LAUNCH((attn_fwd_kernel<1>), dim3(B * H, 1), dim3(64), 0, st, p);
  }
  return ick::launch_status("attention_fwd_d64");
}

// the same contract, products as three fp16 MFMAs each (fp32-grade; see attn_fwd_x3_kernel)
extern "C" int ick_attention_fwd_d64_x3(const float* q, int64_t qld, int64_t qbs, const float* k, int64_t kld, int64_t kbs,
                                        const float* v, int64_t vld, int64_t vbs, float* o, int64_t old, int64_t obs,
                                        int B, int H, int Lq, int Lk, int causal, float scale, void* stream) {
  ICK_REQUIRE(q && k && v && o && B > 0 && H > 0 && Lq > 0 && Lk > 0, "ick_attention_fwd_d64_x3: bad arguments");
  ICK_REQUIRE((kld | vld | old | kbs | vbs | obs) % 4 == 0 && ick::aligned16(k) && ick::aligned16(v) && ick::aligned16(o),
              "ick_attention_fwd_d64_x3: K/V/O rows must be 16-byte aligned");
  AP p{q, k, v, o, qld, kld, vld, old, qbs, kbs, vbs, obs, H, Lq, Lk, causal, scale};
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int tiles = (Lq + 31) / 32;
  if (tiles == 7) ICK_LAUNCH((attn_fwd_x3_kernel<7>), dim3(B * H, 1), dim3(448), 0, st, p);
  else if (tiles >= 4) ICK_LAUNCH((attn_fwd_x3_kernel<4>), dim3(B * H, (tiles + 3) / 4), dim3(256), 0, st, p);
  else if (tiles >= 2) ICK_LAUNCH((attn_fwd_x3_kernel<2>), dim3(B * H, (tiles + 1) / 2), dim3(128), 0, st, p);
  else ICK_LAUNCH((attn_fwd_x3_kernel<1>), dim3(B * H, 1), dim3(64), 0, st, p);
  return ick::launch_status("attention_fwd_d64_x3");
}
