// decoder_fused.hip — the student's LSTM + spatial-attention decode step as THREE launches per token and direction
// (reference loop body: /root/reference/src/student_model.py:232-251; attention_mechanism :173-203; nn.LSTM :142-148).
//
// Round 1 ran a token as 1 + 1 + 2L split-K GEMV-sized GEMMs (fp32 atomics into zero-filled arenas) + 1 + L elementwise
// kernels: ~19 launches, each 8-10 us of mostly idle GPU.  Here the chain of a token is cut only where the data
// dependence forces a chip-wide exchange:
//     forward   A: per image   hW = W_h h_top ; scores / softmax / context ; x = W_c2 ctx + Xe[t]          (dec_attn_x_fwd)
//               L_l: per layer  gates = [inp ; h_prev] [W_ih | W_hh]^T + b -> LSTM cell (+ inter-layer dropout)  (lstm_layer_fwd)
//     backward  G_l: per layer  dG_l [W_hh | W_ih] -> carry into h_l(t-1) ; input gradient -> cell adjoint of layer l-1 (or dX)
//               Z: per image    dctx = W_c2^T dX ; attention adjoint ; dh_top(t-1) += W_h^T dhW ; cell adjoint of the top layer
// i.e. L + 1 launches per token.  Why launches and not one persistent kernel with grid barriers: on this chip a
// dependent kernel boundary inside a hipGraph costs 1.2-1.9 us, an in-kernel chip-wide barrier with a release/acquire
// pair 4-7 us (MI355X_MICROARCH.md, "boundary" vs "barrier-xcd" price rows) — the boundary IS the cheaper barrier.
//
// The two GEMM-shaped stages (64 rows x 4H columns x K <= 1024) run on v_mfma_f32_16x16x4_f32 with operands read straight
// from L2 into registers (the weights of a layer are 4-8 MB: L2 / Infinity-Cache resident across the 15 tokens): one
// workgroup = 16 batch rows x 16 output columns over the FULL K (4 waves split K, LDS combine), so a workgroup owns whole
// hidden units and the cell / cell-adjoint is its epilogue — no split-K atomics, no zero-filled scratch, no second pass.
#include "ick_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

// same generator as dropout_kernel (norm_act.hip): keep(i) is a pure function of (seed, step, i)
__device__ __forceinline__ unsigned mix32(unsigned long long z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return (unsigned)((z ^ (z >> 31)) >> 32);
}
struct Drop {
  float p, inv_keep; unsigned long long seed; const long* step;
  __device__ __forceinline__ float apply(float x, long i) const {
    unsigned long long s = seed;
    if (step) s += (unsigned long long)(*step) * 0x9E3779B97F4A7C15ull;
    const unsigned thr = (unsigned)(p * 4294967296.0);
    return mix32(s * 0x100000001B3ull + (unsigned long long)i) >= thr ? x * inv_keep : 0.f;
  }
};

constexpr int ANT = 1024;   // threads of a per-image workgroup (16 waves)
constexpr int GNT = 256;    // threads of a skinny-GEMM workgroup (4 waves)

// out[e] = sum_k v[k] * W[e][k] for e in [0, n_out), n_out % 4 == 0: one wave per FOUR outputs at a time (lanes stride k by
// float4), v in LDS.  Four rows per trip keep 4-8 independent 16-byte loads in flight per lane: the weights come from L2
// (every image's workgroup re-reads them), and one row per trip left the wave waiting out a full L2 round trip per output
// (34 us for the stage instead of the ~13 us of its attention part).
__device__ __forceinline__ void matvec_rows(const float* __restrict__ W, long ldw, const float* v, int K, int n_out,
                                            float* out_lds, int wave, int lane, int nwaves) {
  for (int e = wave * 4; e < n_out; e += nwaves * 4) {
    const float* wr = W + (long)e * ldw;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    for (int k = lane * 4; k < K; k += 256) {
      const float4 a = *reinterpret_cast<const float4*>(wr + k);
      const float4 b = *reinterpret_cast<const float4*>(wr + ldw + k);
      const float4 c = *reinterpret_cast<const float4*>(wr + 2 * ldw + k);
      const float4 d = *reinterpret_cast<const float4*>(wr + 3 * ldw + k);
      const float4 v4 = *reinterpret_cast<const float4*>(v + k);
      s0 = fmaf(a.x, v4.x, s0); s0 = fmaf(a.y, v4.y, s0); s0 = fmaf(a.z, v4.z, s0); s0 = fmaf(a.w, v4.w, s0);
      s1 = fmaf(b.x, v4.x, s1); s1 = fmaf(b.y, v4.y, s1); s1 = fmaf(b.z, v4.z, s1); s1 = fmaf(b.w, v4.w, s1);
      s2 = fmaf(c.x, v4.x, s2); s2 = fmaf(c.y, v4.y, s2); s2 = fmaf(c.z, v4.z, s2); s2 = fmaf(c.w, v4.w, s2);
      s3 = fmaf(d.x, v4.x, s3); s3 = fmaf(d.y, v4.y, s3); s3 = fmaf(d.z, v4.z, s3); s3 = fmaf(d.w, v4.w, s3);
    }
    s0 = wave_sum(s0); s1 = wave_sum(s1); s2 = wave_sum(s2); s3 = wave_sum(s3);
    if (lane == 0) { out_lds[e] = s0; out_lds[e + 1] = s1; out_lds[e + 2] = s2; out_lds[e + 3] = s3; }
  }
}

// ------------------------------------------------------------------------------------------------ forward, stage A
// one workgroup per image b:  hW[b] = W_h h_top[b]   (W_h = Wa[:, 0:H], row pitch ldwa)
//                             scores_j = sum_e tanh(Uf[b,j,e] + hW[b,e]) ; w = softmax_j ; ctx = sum_j w_j feats[b,j]
//                             x[b] = W_c2 ctx + Xe[b]   (W_c2 = Wc[:, E:2E], row pitch ldwc)
__global__ __launch_bounds__(ANT) void dec_attn_x_fwd_kernel(const float* __restrict__ h_top, const float* __restrict__ Wh, long ldwa,
                                                            const float* __restrict__ Uf, const float* __restrict__ feats,
                                                            const float* __restrict__ Wc2, long ldwc, const float* __restrict__ Xe,
                                                            float* __restrict__ hW_out, float* __restrict__ w_out,
                                                            float* __restrict__ ctx_out, float* __restrict__ x_out, int L, int E, int H) {
  extern __shared__ __attribute__((aligned(16))) float sh[];   // [Lp] scores | [E] hW row | [E] ctx | [max(H, ng*E)] h vector / partial ctx
  const int Lp = (L + 3) & ~3;
  float* sc = sh;
  float* hrow = sh + Lp;
  float* cx = hrow + E;
  float* scratch = cx + E;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* U = Uf + (long)b * L * E;
  const float* F = feats + (long)b * L * E;
  if (h_top) {
    for (int k = tid; k < H; k += ANT) scratch[k] = h_top[(long)b * H + k];
    __syncthreads();
    matvec_rows(Wh, ldwa, scratch, H, E, hrow, wave, lane, ANT / 64);
  } else {
    for (int e = tid; e < E; e += ANT) hrow[e] = 0.f;          // t = 0: h_top is the zero state
  }
  __syncthreads();
  for (int e = tid; e < E; e += ANT) hW_out[(long)b * E + e] = hrow[e];
  // scores: a sum of E tanh values (|score| up to ~E) feeding a softmax — summed in fp64, rounded once (see decoder.hip)
  for (int j = wave; j < L; j += ANT / 64) {
    double s = 0.0;
    for (int e = lane; e < E; e += 64) s += (double)tanhf(U[(long)j * E + e] + hrow[e]);
    s = wave_sum_d(s);
    if (lane == 0) sc[j] = (float)s;
  }
  __syncthreads();
  if (wave == 0) {
    float mx = -INFINITY;
    for (int j = lane; j < L; j += 64) mx = fmaxf(mx, sc[j]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    float sum = 0.f;
    for (int j = lane; j < L; j += 64) sum += expf(sc[j] - mx);
    sum = wave_sum(sum);
    const float inv = 1.f / sum;
    for (int j = lane; j < L; j += 64) {
      const float w = expf(sc[j] - mx) * inv;
      sc[j] = w;
      w_out[(long)b * L + j] = w;
    }
  }
  __syncthreads();
  const int ng = ANT / E;                       // position groups (4 for E = 256)
  const int e = tid % E, jg = tid / E;
  if (jg < ng) {
    float a = 0.f;
    for (int j = jg; j < L; j += ng) a += sc[j] * F[(long)j * E + e];
    scratch[jg * E + e] = a;
  }
  __syncthreads();
  if (tid < E) {
    float a = 0.f;
    for (int g = 0; g < ng; ++g) a += scratch[g * E + tid];   // fixed order: deterministic
    cx[tid] = a;
    ctx_out[(long)b * E + tid] = a;
  }
  __syncthreads();
  matvec_rows(Wc2, ldwc, cx, E, E, scratch, wave, lane, ANT / 64);
  __syncthreads();
  for (int q = tid; q < E; q += ANT) x_out[(long)b * E + q] = scratch[q] + Xe[(long)b * E + q];
}

// ------------------------------------------------------------------------------------------------ skinny MFMA GEMM core
// acc (16 rows x 16 cols of this workgroup, this wave's share of K) over two operand segments:
//   C[r][c] = sum_k A1[r][k] B1[c][k] + sum_k A2[r][k] B2[c][k]
// Lane l feeds v_mfma_f32_16x16x4_f32 with A[row l%16][k-slot l/16] and B[k-slot l/16][col l%16]; it loads FOUR consecutive
// k per operand as one 16-byte access (k = 16 s + 4 (l/16) + 0..3) and MFMA e of the group consumes element e, so the
// instruction's four k-slots are {16 s + 4 q + e : q = 0..3} — the same permutation on both operands: the sum is unchanged.
// K-steps of 16 are dealt round-robin to the 4 waves.  arow / brow = this lane's row pointers (clamped to a valid row).
__device__ __forceinline__ f32x4 skinny_mfma(const float* __restrict__ a1, const float* __restrict__ b1, int K1,
                                             const float* __restrict__ a2, const float* __restrict__ b2, int K2, int wave, int kq) {
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const int s1 = K1 >> 4, S = s1 + (K2 >> 4);
  for (int s0 = wave; s0 < S; s0 += 16) {            // 4 steps of this wave per trip: 8 loads in flight, then 16 MFMAs
    float4 av[4], bv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int s = s0 + 4 * u;
      const bool ok = s < S;
      const bool first = s < s1;
      const float* pa = first ? a1 + (s << 4) : a2 + ((s - s1) << 4);
      const float* pb = first ? b1 + (s << 4) : b2 + ((s - s1) << 4);
      if (ok) {
        av[u] = *reinterpret_cast<const float4*>(pa + 4 * kq);
        bv[u] = *reinterpret_cast<const float4*>(pb + 4 * kq);
      } else {
        av[u] = make_float4(0.f, 0.f, 0.f, 0.f); bv[u] = av[u];
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u].x, bv[u].x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u].y, bv[u].y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u].z, bv[u].z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u].w, bv[u].w, acc, 0, 0, 0);
    }
  }
  return acc;
}

// D layout of the 16x16 tile: lane l holds rows 4*(l/16) + 0..3 of column l%16.  Sum the 4 waves' partials through LDS.
__device__ __forceinline__ void combine_tile(f32x4 acc, float (*red)[16][17], int wave, int lane) {
#pragma unroll
  for (int r = 0; r < 4; ++r) red[wave][4 * (lane >> 4) + r][lane & 15] = acc[r];
}

// ------------------------------------------------------------------------------------------------ forward, stage L
// grid (H/4 unit groups, row tiles of 16).  Column c of the tile = gate c/4 (i,f,g,o) of hidden unit u0 + c%4, i.e. weight
// row (c/4)*H + u0 + c%4: the workgroup holds all four gates of its four units -> the LSTM cell is its epilogue.
__global__ __launch_bounds__(GNT) void lstm_layer_fwd_kernel(const float* __restrict__ inp, int K1, const float* __restrict__ h_prev,
                                                            const float* __restrict__ Wih, const float* __restrict__ Whh,
                                                            const float* __restrict__ bih, const float* __restrict__ bhh,
                                                            const float* __restrict__ c_prev, float* __restrict__ gates,
                                                            float* __restrict__ c_out, float* __restrict__ h_out,
                                                            float* __restrict__ h_drop, Drop drop, int B, int H) {
  __shared__ float red[4][16][17];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int u0 = blockIdx.x * 4, r0 = blockIdx.y * 16;
  const int lr = lane & 15, kq = lane >> 4;
  const int arow = min(r0 + lr, B - 1);                     // rows past the batch read row B-1 and are never stored
  const int wrow = (lr >> 2) * H + u0 + (lr & 3);
  const f32x4 acc = skinny_mfma(inp + (long)arow * K1, Wih + (long)wrow * K1, K1,
                                h_prev ? h_prev + (long)arow * H : nullptr, Whh + (long)wrow * H, h_prev ? H : 0, wave, kq);
  combine_tile(acc, red, wave, lane);
  __syncthreads();
  if (tid < 64) {
    const int i = tid >> 2, ui = tid & 3, b = r0 + i, u = u0 + ui;
    if (b < B) {
      float g[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c = q * 4 + ui;
        g[q] = ((red[0][i][c] + red[1][i][c]) + (red[2][i][c] + red[3][i][c])) + bih[q * H + u] + bhh[q * H + u];
      }
      const float gi = sigmoidf_(g[0]), gf = sigmoidf_(g[1]), gg = tanhf(g[2]), go = sigmoidf_(g[3]);
      const long o = (long)b * H + u;
      const float c = gf * (c_prev ? c_prev[o] : 0.f) + gi * gg;
      const float h = go * tanhf(c);
      c_out[o] = c;
      h_out[o] = h;
      if (h_drop) h_drop[o] = drop.apply(h, o);
      if (gates) {
        float* s = gates + (long)b * 4 * H;
        s[u] = gi; s[H + u] = gf; s[2 * H + u] = gg; s[3 * H + u] = go;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ backward, stage G
// out[b][n] = sum_k dG[b][k] WT[n][k], WT = [W_hh | W_ih]^T stored [(H + K1)][4H] (prepared once per step).
// grid ((H + K1)/16 column groups [first_col offsets it], row tiles).  Columns n < H: the carry into h(t-1) of this layer
// (raw store).  Columns n >= H: the gradient of the layer's input; either stored raw (layer 0: dX[t]) or — layer l > 0 —
// masked by the inter-layer dropout, added to the carry of the layer below and pushed through that layer's cell adjoint
// (column n - H is a hidden unit of the layer below; its four gates are elementwise in (b, unit)).
struct CellBwd {           // the layer below, token t
  const float* carry_h;    // [B][Hb] dL/dh from token t+1 (NULL at the last token)
  float* carry_c;          // [B][Hb] in: dL/dc from token t+1 (ignored when first != 0); out: dL/dc(t-1)
  const float* gates;      // [B][4 Hb] saved activations i,f,g,o
  const float* c;          // [B][Hb] c(t)
  const float* c_prev;     // [B][Hb] c(t-1) or NULL at t = 0
  float* dG;               // [B][4 Hb] out
  int first;               // last token: no incoming carries
};

__device__ __forceinline__ void cell_adjoint(const CellBwd& cb, float dh, int b, int u, int Hb) {
  const long o = (long)b * Hb + u;
  const float* s = cb.gates + (long)b * 4 * Hb;
  const float gi = s[u], gf = s[Hb + u], gg = s[2 * Hb + u], go = s[3 * Hb + u];
  const float tc = tanhf(cb.c[o]);
  const float dc = (cb.first ? 0.f : cb.carry_c[o]) + dh * go * (1.f - tc * tc);
  float* d = cb.dG + (long)b * 4 * Hb;
  d[u] = dc * gg * gi * (1.f - gi);
  d[Hb + u] = dc * (cb.c_prev ? cb.c_prev[o] : 0.f) * gf * (1.f - gf);
  d[2 * Hb + u] = dc * gi * (1.f - gg * gg);
  d[3 * Hb + u] = dh * tc * go * (1.f - go);
  cb.carry_c[o] = dc * gf;
}

__global__ __launch_bounds__(GNT) void lstm_layer_bwd_kernel(const float* __restrict__ dG, const float* __restrict__ WT, int first_col,
                                                            float* __restrict__ carry_h_out, float* __restrict__ d_inp_out,
                                                            CellBwd below, Drop drop, int use_drop, int B, int K1, int H) {
  __shared__ float red[4][16][17];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n0 = first_col + blockIdx.x * 16, r0 = blockIdx.y * 16;
  const int lr = lane & 15, kq = lane >> 4;
  const int K = 4 * H;
  const int arow = min(r0 + lr, B - 1);
  const f32x4 acc = skinny_mfma(dG + (long)arow * K, WT + (long)(n0 + lr) * K, K, nullptr, nullptr, 0, wave, kq);
  combine_tile(acc, red, wave, lane);
  __syncthreads();
  const int i = tid >> 4, j = tid & 15, b = r0 + i, n = n0 + j;
  if (b >= B) return;
  const float v = (red[0][i][j] + red[1][i][j]) + (red[2][i][j] + red[3][i][j]);
  if (n < H) {
    carry_h_out[(long)b * H + n] = v;
  } else {
    const int u = n - H;
    if (!below.gates) {
      d_inp_out[(long)b * K1 + u] = v;
    } else {
      float dh = use_drop ? drop.apply(v, (long)b * K1 + u) : v;
      if (!below.first) dh += below.carry_h[(long)b * K1 + u];
      cell_adjoint(below, dh, b, u, K1);
    }
  }
}

// ------------------------------------------------------------------------------------------------ backward, stage Z
// one workgroup per image b (adjoint of stage A at token t, then the top layer's cell adjoint at token t-1):
//   dctx = W_c2^T dX[b] ; attention adjoint (dUf += , dfeats += , dhW) ; dh_top(t-1)[b] = dHs[t-1][b] + carry + W_h^T dhW
// mode bit 0: run the attention part (a token exists) ; bit 1: run the cell adjoint (a token t-1 exists)
__global__ __launch_bounds__(ANT) void dec_attn_x_bwd_kernel(const float* __restrict__ dX, const float* __restrict__ Wc2, long ldwc,
                                                            const float* __restrict__ w, const float* __restrict__ Uf,
                                                            const float* __restrict__ hW, const float* __restrict__ feats,
                                                            float* __restrict__ dUf, float* __restrict__ dfeats,
                                                            float* __restrict__ dhW_out, const float* __restrict__ Wh, long ldwa,
                                                            const float* __restrict__ dHs_prev, CellBwd top, int mode,
                                                            int L, int E, int H) {
  extern __shared__ __attribute__((aligned(16))) float sh[];  // [Lp] ds | [Lp] w | [E] dctx | [E] dhW | [max(ng*E, 2*Lp, ANT)] scratch
  const int Lp = (L + 3) & ~3;
  float* ds = sh;
  float* wl = sh + Lp;
  float* dc = sh + 2 * Lp;
  float* dhw = dc + E;
  float* part = dhw + E;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long base = (long)b * L * E;
  const int ng = ANT / E;
  const int e = tid % E, jg = tid / E;
  if (mode & 1) {
    // dctx[e] = sum_e' dX[e'] W_c2[e'][e]: thread (e, group) walks its share of the rows e' (coalesced along e)
    for (int q = tid; q < E; q += ANT) dhw[q] = dX[(long)b * E + q];     // dX row staged in the dhW slot for now
    __syncthreads();
    if (jg < ng) {
      float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;            // 4 independent chains: the row loads come from L2
      int r = jg;
      for (; r + 3 * ng < E; r += 4 * ng) {
        const float w0 = Wc2[(long)r * ldwc + e], w1 = Wc2[(long)(r + ng) * ldwc + e];
        const float w2 = Wc2[(long)(r + 2 * ng) * ldwc + e], w3 = Wc2[(long)(r + 3 * ng) * ldwc + e];
        a0 = fmaf(dhw[r], w0, a0); a1 = fmaf(dhw[r + ng], w1, a1);
        a2 = fmaf(dhw[r + 2 * ng], w2, a2); a3 = fmaf(dhw[r + 3 * ng], w3, a3);
      }
      for (; r < E; r += ng) a0 = fmaf(dhw[r], Wc2[(long)r * ldwc + e], a0);
      part[jg * E + e] = (a0 + a1) + (a2 + a3);
    }
    for (int j = tid; j < L; j += ANT) wl[j] = w[(long)b * L + j];
    __syncthreads();
    if (tid < E) {
      float a = 0.f;
      for (int g = 0; g < ng; ++g) a += part[g * E + tid];
      dc[tid] = a;
    }
    __syncthreads();
    // softmax adjoint in fp64 (see decoder.hip attn_step_bwd): ds_j = w_j (dw_j - sum_i w_i dw_i), dw_j = <dctx, f_j>
    double* dwd = reinterpret_cast<double*>(part);
    for (int j = wave; j < L; j += ANT / 64) {
      double s = 0.0;
      for (int q = lane; q < E; q += 64) s = fma((double)dc[q], (double)feats[base + (long)j * E + q], s);
      s = wave_sum_d(s);
      if (lane == 0) dwd[j] = s;
    }
    __syncthreads();
    if (wave == 0) {
      double d = 0.0;
      for (int j = lane; j < L; j += 64) d = fma((double)wl[j], dwd[j], d);
      d = wave_sum_d(d);
      for (int j = lane; j < L; j += 64) ds[j] = (float)((double)wl[j] * (dwd[j] - d));
    }
    __syncthreads();
    if (jg < ng) {
      const float h = hW[(long)b * E + e], dce = dc[e];
      float acc = 0.f;
      for (int j = jg; j < L; j += ng) {
        const long o = base + (long)j * E + e;
        const float ex = expf(-2.f * fabsf(Uf[o] + h));          // sech^2 = 4 e^{-2|x|} / (1 + e^{-2|x|})^2
        const float dp = ds[j] * (4.f * ex / ((1.f + ex) * (1.f + ex)));
        dUf[o] += dp;
        dfeats[o] += wl[j] * dce;
        acc += dp;
      }
      part[jg * E + e] = acc;
    }
    __syncthreads();
    if (tid < E) {
      float a = 0.f;
      for (int g = 0; g < ng; ++g) a += part[g * E + tid];
      dhw[tid] = a;
      dhW_out[(long)b * E + tid] = a;
    }
    __syncthreads();
  }
  if (mode & 2) {
    // dh_top(t-1)[k] = dHs[t-1][b][k] + carry[b][k] + sum_e dhW[e] W_h[e][k]   (coalesced along k; the rows e are
    // shared out over the ANT / H thread groups and combined through LDS)
    const int gh = H <= ANT ? ANT / H : 1;
    if (mode & 1) {
      for (int k0 = 0; k0 < H; k0 += ANT) {                       // one trip unless H > 1024
        const int k = k0 + tid % (H <= ANT ? H : ANT), g = H <= ANT ? tid / H : 0;
        if (g < gh && k < H) {
          float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
          int q = g;
          for (; q + 3 * gh < E; q += 4 * gh) {
            const float w0 = Wh[(long)q * ldwa + k], w1 = Wh[(long)(q + gh) * ldwa + k];
            const float w2 = Wh[(long)(q + 2 * gh) * ldwa + k], w3 = Wh[(long)(q + 3 * gh) * ldwa + k];
            a0 = fmaf(dhw[q], w0, a0); a1 = fmaf(dhw[q + gh], w1, a1);
            a2 = fmaf(dhw[q + 2 * gh], w2, a2); a3 = fmaf(dhw[q + 3 * gh], w3, a3);
          }
          for (; q < E; q += gh) a0 = fmaf(dhw[q], Wh[(long)q * ldwa + k], a0);
          part[g * (H <= ANT ? H : ANT) + (k - k0)] = (a0 + a1) + (a2 + a3);
        }
        __syncthreads();
        const int kk = k0 + tid;
        if (tid < (H <= ANT ? H : ANT) && kk < H) {
          float a = dHs_prev[(long)b * H + kk];
          for (int g2 = 0; g2 < gh; ++g2) a += part[g2 * (H <= ANT ? H : ANT) + tid];
          if (!top.first) a += top.carry_h[(long)b * H + kk];
          cell_adjoint(top, a, b, kk, H);
        }
        __syncthreads();
      }
    } else {
      for (int k = tid; k < H; k += ANT) {
        float a = dHs_prev[(long)b * H + k];
        if (!top.first) a += top.carry_h[(long)b * H + k];
        cell_adjoint(top, a, b, k, H);
      }
    }
  }
}

// dst[c][r] = src[r][c]: 32x32 tiles through LDS (weight transposes for stage G, once per step)
__global__ void transpose2d_kernel(const float* __restrict__ src, long lds_, float* __restrict__ dst, long ldd, int rows, int cols) {
  __shared__ float tile[32][33];
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
  for (int r = ty; r < 32; r += 8) {
    const int rr = r0 + r, cc = c0 + tx;
    tile[r][tx] = (rr < rows && cc < cols) ? src[(long)rr * lds_ + cc] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int r = ty; r < 32; r += 8) {
    const int cc = c0 + r, rr = r0 + tx;
    if (cc < cols && rr < rows) dst[(long)cc * ldd + rr] = tile[tx][r];
  }
}

inline Drop make_drop(float p, uint64_t seed, const int64_t* step) {
  Drop d; d.p = p; d.inv_keep = p < 1.f ? 1.f / (1.f - p) : 0.f; d.seed = seed; d.step = (const long*)step; return d;
}

}  // namespace

#define ST static_cast<hipStream_t>(stream)

extern "C" {

int ick_dec_attn_x_fwd(const float* h_top, const float* Wh, int64_t ldwa, const float* Uf, const float* feats, const float* Wc2,
                       int64_t ldwc, const float* Xe, float* hW_out, float* w_out, float* ctx_out, float* x_out, int B, int L, int E,
                       int H, void* stream) {
  ICK_REQUIRE(Wh && Uf && feats && Wc2 && Xe && hW_out && w_out && ctx_out && x_out && B > 0 && L > 0, "ick_dec_attn_x_fwd: bad arguments");
  ICK_REQUIRE(E > 0 && E <= ANT && E % 4 == 0 && H % 4 == 0 && ldwa % 4 == 0 && ldwc % 4 == 0,
              "ick_dec_attn_x_fwd: E <= 1024 and E, H, ldwa, ldwc multiples of 4 (E=%d H=%d)", E, H);
  ICK_REQUIRE(ick::aligned16(Wh) && ick::aligned16(Wc2), "ick_dec_attn_x_fwd: weight slices must be 16-byte aligned");
  const int Lp = (L + 3) & ~3;
  const int scr = H > (ANT / E) * E ? H : (ANT / E) * E;
  const size_t sh = (size_t)(Lp + 2 * E + scr) * sizeof(float);
  ICK_LAUNCH(dec_attn_x_fwd_kernel, dim3(B), dim3(ANT), sh, ST, h_top, Wh, (long)ldwa, Uf, feats, Wc2, (long)ldwc, Xe, hW_out, w_out,
             ctx_out, x_out, L, E, H);
  return ick::launch_status("dec_attn_x_fwd");
}

int ick_lstm_layer_fwd(const float* inp, int K1, const float* h_prev, const float* Wih, const float* Whh, const float* bih,
                       const float* bhh, const float* c_prev, float* gates, float* c_out, float* h_out, float* h_drop, float p_drop,
                       uint64_t seed, const int64_t* step, int B, int H, void* stream) {
  ICK_REQUIRE(inp && Wih && Whh && bih && bhh && c_out && h_out && B > 0, "ick_lstm_layer_fwd: bad arguments");
  ICK_REQUIRE(K1 > 0 && K1 % 16 == 0 && H > 0 && H % 16 == 0, "ick_lstm_layer_fwd: input width and H must be multiples of 16 (K1=%d H=%d)", K1, H);
  ICK_REQUIRE(ick::aligned16(inp) && ick::aligned16(Wih) && ick::aligned16(Whh) && (!h_prev || ick::aligned16(h_prev)),
              "ick_lstm_layer_fwd: operands must be 16-byte aligned");
  ICK_REQUIRE(!h_drop || (p_drop > 0.f && p_drop < 1.f), "ick_lstm_layer_fwd: h_drop needs 0 < p < 1");
  ICK_LAUNCH(lstm_layer_fwd_kernel, dim3(H / 4, (B + 15) / 16), dim3(GNT), 0, ST, inp, K1, h_prev, Wih, Whh, bih, bhh, c_prev, gates,
             c_out, h_out, h_drop, make_drop(p_drop, seed, step), B, H);
  return ick::launch_status("lstm_layer_fwd");
}

int ick_lstm_layer_bwd(const float* dG, const float* WT, float* carry_h_out, float* d_inp_out, const float* below_carry_h,
                       float* below_carry_c, const float* below_gates, const float* below_c, const float* below_c_prev,
                       float* below_dG, int below_first, float p_drop, uint64_t seed, const int64_t* step, int B, int K1, int H,
                       void* stream) {
  ICK_REQUIRE(dG && WT && B > 0 && K1 > 0 && K1 % 16 == 0 && H > 0 && H % 16 == 0, "ick_lstm_layer_bwd: bad arguments (K1=%d H=%d)", K1, H);
  ICK_REQUIRE(ick::aligned16(dG) && ick::aligned16(WT), "ick_lstm_layer_bwd: operands must be 16-byte aligned");
  ICK_REQUIRE((d_inp_out != nullptr) != (below_gates != nullptr), "ick_lstm_layer_bwd: exactly one of d_inp_out / the layer below");
  ICK_REQUIRE(!below_gates || (below_carry_c && below_c && below_dG && (below_first || below_carry_h)),
              "ick_lstm_layer_bwd: incomplete cell-adjoint arguments");
  CellBwd cb{below_carry_h, below_carry_c, below_gates, below_c, below_c_prev, below_dG, below_first};
  const int first_col = carry_h_out ? 0 : H;                 // t = 0: no token t-1 -> skip the recurrent half
  const int groups = (H + K1 - first_col) / 16;
  ICK_LAUNCH(lstm_layer_bwd_kernel, dim3(groups, (B + 15) / 16), dim3(GNT), 0, ST, dG, WT, first_col, carry_h_out, d_inp_out, cb,
             make_drop(p_drop, seed, step), p_drop > 0.f ? 1 : 0, B, K1, H);
  return ick::launch_status("lstm_layer_bwd");
}

int ick_dec_attn_x_bwd(const float* dX, const float* Wc2, int64_t ldwc, const float* w, const float* Uf, const float* hW,
                       const float* feats, float* dUf, float* dfeats, float* dhW_out, const float* Wh, int64_t ldwa,
                       const float* dHs_prev, const float* top_carry_h, float* top_carry_c, const float* top_gates,
                       const float* top_c, const float* top_c_prev, float* top_dG, int top_first, int B, int L, int E, int H,
                       void* stream) {
  const int mode = (dX ? 1 : 0) | (dHs_prev ? 2 : 0);
  ICK_REQUIRE(mode != 0 && B > 0 && L > 0, "ick_dec_attn_x_bwd: nothing to do");
  ICK_REQUIRE(E > 0 && E <= ANT && E % 2 == 0 && H > 0, "ick_dec_attn_x_bwd: E <= 1024 and even (E=%d)", E);
  if (mode & 1) ICK_REQUIRE(Wc2 && w && Uf && hW && feats && dUf && dfeats && dhW_out && Wh, "ick_dec_attn_x_bwd: missing attention operands");
  if (mode & 2) ICK_REQUIRE(top_carry_c && top_gates && top_c && top_dG && (top_first || top_carry_h), "ick_dec_attn_x_bwd: missing cell operands");
  const int Lp = (L + 3) & ~3;
  int scr = ANT;                                             // >= (ANT / E) * E and >= (ANT / H) * H
  if (scr < 2 * Lp) scr = 2 * Lp;
  if (scr < H) scr = H;
  const size_t sh = (size_t)(2 * Lp + 2 * E + scr) * sizeof(float);
  CellBwd cb{top_carry_h, top_carry_c, top_gates, top_c, top_c_prev, top_dG, top_first};
  ICK_LAUNCH(dec_attn_x_bwd_kernel, dim3(B), dim3(ANT), sh, ST, dX, Wc2, (long)ldwc, w, Uf, hW, feats, dUf, dfeats, dhW_out, Wh,
             (long)ldwa, dHs_prev, cb, mode, L, E, H);
  return ick::launch_status("dec_attn_x_bwd");
}

int ick_transpose2d(const float* src, int64_t ld_src, float* dst, int64_t ld_dst, int rows, int cols, void* stream) {
  ICK_REQUIRE(src && dst && rows > 0 && cols > 0 && ld_src >= cols && ld_dst >= rows, "ick_transpose2d: bad arguments");
  ICK_LAUNCH(transpose2d_kernel, dim3((cols + 31) / 32, (rows + 31) / 32), dim3(256), 0, ST, src, (long)ld_src, dst, (long)ld_dst, rows, cols);
  return ick::launch_status("transpose2d");
}

}  // extern "C"
