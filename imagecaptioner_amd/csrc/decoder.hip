// decoder.hip — per-time-step kernels of the student's LSTM + spatial-attention decoder
// (reference: LSTMDecoder.attention_mechanism /root/reference/src/student_model.py:173-203 and the loop body
// :232-251).  The decoder is recast so that everything time-invariant is hoisted out of the loop:
//   W_a [h ; f_j] = W_h h + (W_f f_j + b_a)   -> Uf = feats W_f^T + b_a is ONE GEMM per image batch,
// and the per-step work is GEMV-sized contractions (igemm_f32) plus these HBM/LDS-bound wavefront kernels:
//   attn_step_fwd : scores_j = sum_e tanh(Uf[b,j,e] + hW[b,e]); w = softmax_j; ctx = sum_j w_j f_j
//   attn_step_bwd : the exact adjoint (recomputes tanh from Uf + hW; accumulates dUf, dfeats in place)
//   lstm_cell_fwd / lstm_cell_bwd : gate nonlinearities + state update / adjoint (gate order i,f,g,o)
//   argmax_rows   : greedy token selection (first maximum, like torch.argmax)
#include "ick_common.h"

namespace {

constexpr int NT = 256;

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

// one 1024-thread workgroup (16 waves) per image b; L positions (49), E features (<= 1024).
// Only B workgroups exist per step, so the work of one image is spread over as many waves as a CU takes:
// scores: one wave per position (lanes stride the E axis); context: threads = (feature e) x (position group).
constexpr int ANT = 1024;

__global__ __launch_bounds__(ANT) void attn_step_fwd_kernel(const float* __restrict__ Uf, const float* __restrict__ hW,
                                                           const float* __restrict__ feats, float* __restrict__ w_out,
                                                           float* __restrict__ ctx, int L, int E) {
  extern __shared__ __attribute__((aligned(16))) float sh[];  // [Lp] scores/weights, [E] hW row, [ng*E] partial ctx
  const int Lp = (L + 3) & ~3;
  float* sc = sh;
  float* hrow = sh + Lp;
  float* part = hrow + E;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* U = Uf + (long)b * L * E;
  const float* F = feats + (long)b * L * E;
  for (int e = tid; e < E; e += ANT) hrow[e] = hW[(long)b * E + e];
  __syncthreads();
  // A score is a sum of E tanh values (|score| up to ~E) feeding a softmax: its ABSOLUTE rounding error is what the
  // attention weights see, so the E terms are summed in fp64 (4 per lane + a shuffle tree: free) and rounded once.
  for (int j = wave; j < L; j += ANT / 64) {
    double s = 0.0;
    for (int e = lane; e < E; e += 64) s += (double)tanhf(U[(long)j * E + e] + hrow[e]);
    s = wave_sum_d(s);
    if (lane == 0) sc[j] = (float)s;
  }
  __syncthreads();
  if (wave == 0) {  // softmax over the L positions
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
  const int ng = ANT / E;                       // position groups (4 for E=256)
  const int e = tid % E, jg = tid / E;
  if (jg < ng) {
    float a = 0.f;
    for (int j = jg; j < L; j += ng) a += sc[j] * F[(long)j * E + e];
    part[jg * E + e] = a;
  }
  __syncthreads();
  if (tid < E) {
    float a = 0.f;
    for (int g = 0; g < ng; ++g) a += part[g * E + tid];   // fixed order: deterministic
    ctx[(long)b * E + tid] = a;
  }
}

// adjoint of attn_step_fwd for one step: dUf += dpre, dfeats += w (x) dctx, dhW = sum_j dpre
__global__ __launch_bounds__(ANT) void attn_step_bwd_kernel(const float* __restrict__ dctx, const float* __restrict__ w,
                                                           const float* __restrict__ Uf, const float* __restrict__ hW,
                                                           const float* __restrict__ feats, float* __restrict__ dUf,
                                                           float* __restrict__ dfeats, float* __restrict__ dhW, int L, int E) {
  extern __shared__ __attribute__((aligned(16))) float sh[];  // [Lp] ds, [Lp] w, [E] dctx, [ng*E] partial dhW
  const int Lp = (L + 3) & ~3;
  float* ds = sh;
  float* wl = sh + Lp;
  float* dc = sh + 2 * Lp;
  float* part = dc + E;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long base = (long)b * L * E;
  for (int e = tid; e < E; e += ANT) dc[e] = dctx[(long)b * E + e];
  for (int j = tid; j < L; j += ANT) wl[j] = w[(long)b * L + j];
  __syncthreads();
  // softmax adjoint ds_j = w_j (dw_j - sum_i w_i dw_i), dw_j = <dctx, f_j>: with a peaky softmax the bracket cancels
  // to a small fraction of dw_j, so dw_j and the weighted mean are formed in fp64 (49 x E products per image: free)
  double* dwd = reinterpret_cast<double*>(part);      // [Lp] fp64 scratch (part is not live yet; 8-byte aligned: see launcher)
  for (int j = wave; j < L; j += ANT / 64) {
    double s = 0.0;
    for (int e = lane; e < E; e += 64) s = fma((double)dc[e], (double)feats[base + (long)j * E + e], s);
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
  const int ng = ANT / E;
  const int e = tid % E, jg = tid / E;
  if (jg < ng) {
    const float h = hW[(long)b * E + e], dce = dc[e];
    float acc = 0.f;
    for (int j = jg; j < L; j += ng) {
      const long o = base + (long)j * E + e;
      // d tanh = sech^2 = 4 e^{-2|x|} / (1 + e^{-2|x|})^2 : relative accuracy also where tanh saturates
      // (1 - tanh^2 loses all digits there)
      const float ex = expf(-2.f * fabsf(Uf[o] + h));
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
    dhW[(long)b * E + tid] = a;
  }
}

// gates G (B,4H) = x W_ih^T + h W_hh^T (raw sums from the GEMMs) + b_ih + b_hh ; order i,f,g,o
__global__ void lstm_cell_fwd_kernel(const float* __restrict__ G, const float* __restrict__ bih, const float* __restrict__ bhh,
                                     const float* __restrict__ c_prev, float* __restrict__ gates, float* __restrict__ c_out,
                                     float* __restrict__ h_out, int B, int H) {
  const long total = (long)B * H;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long b = i / H; const int h = (int)(i - b * H);
    const float* g = G + b * 4 * H;
    const float gi = sigmoidf_(g[h] + bih[h] + bhh[h]);
    const float gf = sigmoidf_(g[H + h] + bih[H + h] + bhh[H + h]);
    const float gg = tanhf(g[2 * H + h] + bih[2 * H + h] + bhh[2 * H + h]);
    const float go = sigmoidf_(g[3 * H + h] + bih[3 * H + h] + bhh[3 * H + h]);
    const float c = gf * (c_prev ? c_prev[i] : 0.f) + gi * gg;
    c_out[i] = c;
    h_out[i] = go * tanhf(c);
    if (gates) {
      float* s = gates + b * 4 * H;
      s[h] = gi; s[H + h] = gf; s[2 * H + h] = gg; s[3 * H + h] = go;
    }
  }
}

// dh = dh_a (+ dh_b); dc = dc_in + dh*o*(1-tanh(c)^2); writes dG (pre-activation) and dc_prev (may alias dc_in)
__global__ void lstm_cell_bwd_kernel(const float* __restrict__ dh_a, const float* __restrict__ dh_b,
                                     const float* dc_in, const float* __restrict__ gates, const float* __restrict__ c,
                                     const float* __restrict__ c_prev, float* __restrict__ dG, float* dc_prev, int B, int H) {
  const long total = (long)B * H;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long b = i / H; const int h = (int)(i - b * H);
    const float* s = gates + b * 4 * H;
    const float gi = s[h], gf = s[H + h], gg = s[2 * H + h], go = s[3 * H + h];
    float dh = dh_a[i];
    if (dh_b) dh += dh_b[i];
    const float tc = tanhf(c[i]);
    const float dc = (dc_in ? dc_in[i] : 0.f) + dh * go * (1.f - tc * tc);
    float* d = dG + b * 4 * H;
    d[h] = dc * gg * gi * (1.f - gi);
    d[H + h] = dc * (c_prev ? c_prev[i] : 0.f) * gf * (1.f - gf);
    d[2 * H + h] = dc * gi * (1.f - gg * gg);
    d[3 * H + h] = dh * tc * go * (1.f - go);
    dc_prev[i] = dc * gf;
  }
}

// ids[r] = torch.argmax(logits[r][0..V)): first index of the maximum, NaN counting as the largest value (one wave per
// row).  A row that is all -inf (or all NaN) yields 0 like torch — never an out-of-range id: the next greedy step gathers
// an embedding row with it.
__device__ __forceinline__ bool amax_better(float v, float best) { return v > best || (v != v && best == best); }
__global__ void argmax_rows_kernel(const float* __restrict__ x, long* __restrict__ ids, long rows, int V, long ld) {
  const int lane = threadIdx.x & 63, wpb = blockDim.x >> 6;
  for (long r = blockIdx.x * (long)wpb + (threadIdx.x >> 6); r < rows; r += (long)gridDim.x * wpb) {
    const float* xr = x + r * ld;
    float best = -INFINITY; int bi = V;
    // 16-byte loads, 4 of them in flight per lane (a 4-byte load per dependent compare left the wave latency-bound:
    // 78 round trips per 5000-word row); indices grow with every step, a strict comparison keeps the first maximum
    if ((ld & 3) == 0 && (V & 3) == 0) {
      const int V4 = V >> 2;
      int i = lane;
      for (; i + 192 < V4; i += 256) {
        float4 q[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) q[u] = reinterpret_cast<const float4*>(xr)[i + 64 * u];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int b = (i + 64 * u) * 4;
          if (amax_better(q[u].x, best)) { best = q[u].x; bi = b; }
          if (amax_better(q[u].y, best)) { best = q[u].y; bi = b + 1; }
          if (amax_better(q[u].z, best)) { best = q[u].z; bi = b + 2; }
          if (amax_better(q[u].w, best)) { best = q[u].w; bi = b + 3; }
        }
      }
      for (; i < V4; i += 64) {
        const float4 q = reinterpret_cast<const float4*>(xr)[i];
        const int b = i * 4;
        if (amax_better(q.x, best)) { best = q.x; bi = b; }
        if (amax_better(q.y, best)) { best = q.y; bi = b + 1; }
        if (amax_better(q.z, best)) { best = q.z; bi = b + 2; }
        if (amax_better(q.w, best)) { best = q.w; bi = b + 3; }
      }
    } else {
      for (int i = lane; i < V; i += 64) {
        const float v = xr[i];
        if (amax_better(v, best)) { best = v; bi = i; }
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(best, o); const int oi = __shfl_xor(bi, o);
      const bool eq = ov == best || (ov != ov && best != best);
      if (amax_better(ov, best) || (eq && oi < bi)) { best = ov; bi = oi; }
    }
    if (lane == 0) ids[r] = bi < V ? bi : 0;
  }
}

// beam-search expansion (reference: CaptioningTeacher.caption_image, /root/reference/src/teacher_model.py:170-179):
// candidate[b][v] = scores[b] + log_softmax(logits[b])[v]; the k best of the flattened (b,v) space, best first
// (ties: lower flat index).  One 1024-thread workgroup: Bl*V is ~25k.  No host round trip between the softmax
// and the selection; the caller copies 2k numbers back per step instead of the reference's per-candidate .item()s.
__global__ __launch_bounds__(1024) void beam_topk_kernel(const float* __restrict__ logits, const float* __restrict__ scores,
                                                        int Bl, int V, int k, float* __restrict__ out_vals,
                                                        long* __restrict__ out_idx) {
  __shared__ float lse[64];
  __shared__ float rv[16];
  __shared__ long ri[16];
  __shared__ long chosen[32];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int b = 0; b < Bl; ++b) {                      // log-sum-exp of every live beam's row
    const float* row = logits + (long)b * V;
    float mx = -INFINITY;
    for (int v = tid; v < V; v += 1024) mx = fmaxf(mx, row[v]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if (lane == 0) rv[wave] = mx;
    __syncthreads();
    mx = rv[0];
    for (int w = 1; w < 16; ++w) mx = fmaxf(mx, rv[w]);
    __syncthreads();
    float s = 0.f;
    for (int v = tid; v < V; v += 1024) s += expf(row[v] - mx);
    s = wave_sum(s);
    if (lane == 0) rv[wave] = s;
    __syncthreads();
    if (tid == 0) {
      float t = 0.f;
      for (int w = 0; w < 16; ++w) t += rv[w];
      lse[b] = mx + logf(t);
    }
    __syncthreads();
  }
  const long total = (long)Bl * V;
  for (int r = 0; r < k; ++r) {
    float best = -INFINITY; long bi = total;
    for (long i = tid; i < total; i += 1024) {
      bool taken = false;
      for (int c = 0; c < r; ++c) taken |= (chosen[c] == i);
      if (taken) continue;
      const int b = (int)(i / V);
      const float val = scores[b] + (logits[i] - lse[b]);
      if (val > best || (val == best && i < bi) || bi == total) { best = val; bi = i; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(best, o); const long oi = __shfl_xor(bi, o);
      if (oi != total && (bi == total || ov > best || (ov == best && oi < bi))) { best = ov; bi = oi; }
    }
    if (lane == 0) { rv[wave] = best; ri[wave] = bi; }
    __syncthreads();
    if (tid == 0) {
      float bb = rv[0]; long ii = ri[0];
      for (int w = 1; w < 16; ++w)
        if (ri[w] != total && (ii == total || rv[w] > bb || (rv[w] == bb && ri[w] < ii))) { bb = rv[w]; ii = ri[w]; }
      chosen[r] = ii; out_vals[r] = bb; out_idx[r] = ii;
    }
    __syncthreads();
  }
}

}  // namespace

#define ST static_cast<hipStream_t>(stream)

extern "C" {

int ick_attn_step_fwd(const float* Uf, const float* hW, const float* feats, float* w_out, float* ctx, int B, int L, int E,
                      void* stream) {
  ICK_REQUIRE(Uf && hW && feats && w_out && ctx && B > 0 && L > 0 && E > 0 && E <= ANT, "ick_attn_step_fwd: bad arguments (E <= 1024)");
  const size_t sh = (((L + 3) & ~3) + E + (size_t)(ANT / E) * E) * sizeof(float);
  ICK_LAUNCH(attn_step_fwd_kernel, dim3(B), dim3(ANT), sh, ST, Uf, hW, feats, w_out, ctx, L, E);
  return ick::launch_status("attn_step_fwd");
}

int ick_attn_step_bwd(const float* dctx, const float* w, const float* Uf, const float* hW, const float* feats, float* dUf,
                      float* dfeats, float* dhW, int B, int L, int E, void* stream) {
  ICK_REQUIRE(dctx && w && Uf && hW && feats && dUf && dfeats && dhW && B > 0 && L > 0 && E > 0,
              "ick_attn_step_bwd: bad arguments");
  ICK_REQUIRE(E <= ANT && E % 2 == 0, "ick_attn_step_bwd: E <= 1024 and even");
  ICK_REQUIRE(2 * ((L + 3) & ~3) <= (ANT / E) * E, "ick_attn_step_bwd: L too large for the fp64 scratch (L=%d, E=%d)", L, E);
  const size_t sh = (2 * ((L + 3) & ~3) + E + (size_t)(ANT / E) * E) * sizeof(float);
  ICK_LAUNCH(attn_step_bwd_kernel, dim3(B), dim3(ANT), sh, ST, dctx, w, Uf, hW, feats, dUf, dfeats, dhW, L, E);
  return ick::launch_status("attn_step_bwd");
}

int ick_lstm_cell_fwd(const float* G, const float* b_ih, const float* b_hh, const float* c_prev, float* gates, float* c_out,
                      float* h_out, int B, int H, void* stream) {
  ICK_REQUIRE(G && b_ih && b_hh && c_out && h_out && B > 0 && H > 0, "ick_lstm_cell_fwd: bad arguments");
  long g = ((long)B * H + NT - 1) / NT; if (g > 2048) g = 2048;
  ICK_LAUNCH(lstm_cell_fwd_kernel, dim3((int)g), dim3(NT), 0, ST, G, b_ih, b_hh, c_prev, gates, c_out, h_out, B, H);
  return ick::launch_status("lstm_cell_fwd");
}

int ick_lstm_cell_bwd(const float* dh_a, const float* dh_b, const float* dc_in, const float* gates, const float* c,
                      const float* c_prev, float* dG, float* dc_prev, int B, int H, void* stream) {
  ICK_REQUIRE(dh_a && gates && c && dG && dc_prev && B > 0 && H > 0, "ick_lstm_cell_bwd: bad arguments");
  long g = ((long)B * H + NT - 1) / NT; if (g > 2048) g = 2048;
  ICK_LAUNCH(lstm_cell_bwd_kernel, dim3((int)g), dim3(NT), 0, ST, dh_a, dh_b, dc_in, gates, c, c_prev, dG, dc_prev, B, H);
  return ick::launch_status("lstm_cell_bwd");
}

int ick_argmax_rows(const float* x, int64_t* ids, int64_t rows, int V, int64_t ld, void* stream) {
  ICK_REQUIRE(x && ids && rows > 0 && V > 0 && ld >= V, "ick_argmax_rows: bad arguments");
  long g = (rows + 3) / 4; if (g > 2048) g = 2048;
  ICK_LAUNCH(argmax_rows_kernel, dim3((int)g), dim3(NT), 0, ST, x, (long*)ids, (long)rows, V, (long)ld);
  return ick::launch_status("argmax_rows");
}

int ick_beam_topk(const float* logits, const float* scores, int Bl, int V, int k, float* out_vals, int64_t* out_idx,
                  void* stream) {
  ICK_REQUIRE(logits && scores && out_vals && out_idx && Bl > 0 && Bl <= 64 && V > 0 && k > 0 && k <= 32 && (long)Bl * V >= k,
              "ick_beam_topk: bad arguments (Bl <= 64, k <= 32)");
  ICK_LAUNCH(beam_topk_kernel, dim3(1), dim3(1024), 0, ST, logits, scores, Bl, V, k, out_vals, (long*)out_idx);
  return ick::launch_status("beam_topk");
}

}  // extern "C"
