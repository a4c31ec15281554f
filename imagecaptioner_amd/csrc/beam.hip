// beam.hip — the two kernels that make the teacher's beam search KV-cached, batched over images and free of host
// round trips (reference: CaptioningTeacher.caption_image, /root/reference/src/teacher_model.py:108-252, which re-runs
// the whole decoder on the growing prefixes, one image at a time, with three .item() syncs per candidate).
//
// State for B images x W beam slots (row = b*W + j), Tcap = max_length + 1 token positions:
//   * K / V cache per decoder layer: [Tcap][B*W][E], written by the step that consumed the position;
//   * anc[t'][row] = the ROW whose cache line at position t' belongs to this beam's history.  A beam that survives a
//     step inherits its origin's ancestry, so the caches are never copied or permuted (a physical gather would move
//     ~100 MB per step at B = 64): the attention kernel follows the indirection instead;
//   * seq[row][Tcap] (int32 tokens incl. <START>), score[row], width[b] = live beams of image b (shrinks by one per
//     finished hypothesis, exactly as the reference's `beam_size = B_live`), finished hypotheses per image in
//     finishing order (at most W over the whole search: every finish narrows the beam by one).
#include "ick_common.h"
#include <cmath>

namespace {

constexpr int HD = 64;   // head dim of every attention of the teacher (E / heads = 512 / 8)

__device__ __forceinline__ float wave_sum64(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// One wave per (row, head): lane = head dimension.  qkv [rows][3E] holds this step's packed in_proj output of the new
// token (position t): its key / value are stored into the caches, then the query attends to positions 0..t through anc.
__global__ __launch_bounds__(64) void beam_self_attn_kernel(const float* __restrict__ qkv, float* __restrict__ Kc,
                                                            float* __restrict__ Vc, const int* __restrict__ anc,
                                                            float* __restrict__ out, int rows, int E, int t, float scale) {
  const int row = blockIdx.x, h = blockIdx.y, lane = threadIdx.x;
  const long col = (long)h * HD + lane;
  const float* x = qkv + (long)row * 3 * E;
  const float q = x[col] * scale, kown = x[E + col], vown = x[2 * E + col];
  const long plane = (long)rows * E;
  Kc[t * plane + (long)row * E + col] = kown;
  Vc[t * plane + (long)row * E + col] = vown;
  float m = -INFINITY, l = 0.f, acc = 0.f;         // online softmax over the t + 1 keys (every lane holds the same m, l)
  for (int p = 0; p <= t; ++p) {
    float kk, vv;
    if (p < t) {
      const long src = (long)anc[(long)p * rows + row] * E + col;
      kk = Kc[p * plane + src]; vv = Vc[p * plane + src];
    } else { kk = kown; vv = vown; }
    const float s = wave_sum64(q * kk);
    const float mn = fmaxf(m, s);
    const float c = expf(m - mn), e = expf(s - mn);
    l = l * c + e; acc = acc * c + e * vv; m = mn;
  }
  out[(long)row * E + col] = acc / l;
}

// One workgroup per image.  logits [B*W][V] of the live rows' last position; expands every live beam with
// log-probabilities, keeps the `width` best (beam, token) pairs (ties: lower flat index, as ick_beam_topk), retires the
// ones ending in <END> as finished hypotheses (raw score + length: the host normalises in double like the reference's
// Python arithmetic, :193-197) and compacts the survivors into slots 0..nlive-1 in rank order.
__global__ __launch_bounds__(1024) void beam_step_kernel(const float* __restrict__ logits, float* __restrict__ score,
                                                        int* __restrict__ width, const int* __restrict__ seq_in,
                                                        int* __restrict__ seq_out, const int* __restrict__ anc_in,
                                                        int* __restrict__ anc_out, long* __restrict__ next_tok,
                                                        int* __restrict__ fin_seq, float* __restrict__ fin_score,
                                                        int* __restrict__ fin_len, int* __restrict__ nfin, int B, int W,
                                                        int V, int Tcap, int t, int end_id) {
  __shared__ float lse[32], sc[32];
  __shared__ float rv[16];
  __shared__ long ri[16];
  __shared__ long chosen[32];
  __shared__ float cval[32];
  __shared__ int dst[32];            // >= 0: live slot, < 0: -(finished index + 1)
  __shared__ int s_nlive;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int w = width[b];
  const int rows = B * W;
  if (w == 0) {                      // image done: keep feeding valid token ids to the (ignored) rows
    if (tid < W) next_tok[b * W + tid] = 0;
    return;
  }
  const float* lg = logits + (long)b * W * V;
  if (tid < w) sc[tid] = score[b * W + tid];
  for (int r = 0; r < w; ++r) {      // log-sum-exp of every live row
    const float* rowp = lg + (long)r * V;
    float mx = -INFINITY;
    for (int v = tid; v < V; v += 1024) mx = fmaxf(mx, rowp[v]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if (lane == 0) rv[wave] = mx;
    __syncthreads();
    mx = rv[0];
    for (int i = 1; i < 16; ++i) mx = fmaxf(mx, rv[i]);
    __syncthreads();
    float s = 0.f;
    for (int v = tid; v < V; v += 1024) s += expf(rowp[v] - mx);
    s = wave_sum64(s);
    if (lane == 0) rv[wave] = s;
    __syncthreads();
    if (tid == 0) {
      float tt = 0.f;
      for (int i = 0; i < 16; ++i) tt += rv[i];
      lse[r] = mx + logf(tt);
    }
    __syncthreads();
  }
  const long total = (long)w * V;
  for (int r = 0; r < w; ++r) {      // the w best candidates, best first
    float best = -INFINITY; long bi = total;
    for (long i = tid; i < total; i += 1024) {
      bool taken = false;
      for (int c = 0; c < r; ++c) taken |= (chosen[c] == i);
      if (taken) continue;
      const int rb = (int)(i / V);
      const float val = sc[rb] + (lg[i] - lse[rb]);
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
      for (int i = 1; i < 16; ++i)
        if (ri[i] != total && (ii == total || rv[i] > bb || (rv[i] == bb && ri[i] < ii))) { bb = rv[i]; ii = ri[i]; }
      chosen[r] = ii; cval[r] = bb;
    }
    __syncthreads();
  }
  if (tid == 0) {                    // classify in rank order (finishing order == rank order within a step)
    int nl = 0, nf = nfin[b];
    for (int r = 0; r < w; ++r) {
      const int tok = (int)(chosen[r] % V);
      if (end_id >= 0 && tok == end_id) { dst[r] = -(nf + 1); ++nf; }
      else dst[r] = nl++;
    }
    nfin[b] = nf; s_nlive = nl; width[b] = nl;
  }
  __syncthreads();
  const int len = t + 1;             // tokens of the prefix (positions 0..t); the grown sequences hold len + 1
  for (int r = 0; r < w; ++r) {
    const int origin = (int)(chosen[r] / V), tok = (int)(chosen[r] % V);
    const int orow = b * W + origin;
    if (dst[r] >= 0) {
      const int nrow = b * W + dst[r];
      for (int p = tid; p < len; p += 1024) {
        seq_out[(long)nrow * Tcap + p] = seq_in[(long)orow * Tcap + p];
        anc_out[(long)p * rows + nrow] = p < t ? anc_in[(long)p * rows + orow] : orow;
      }
      if (tid == 0) { seq_out[(long)nrow * Tcap + len] = tok; score[nrow] = cval[r]; next_tok[nrow] = tok; }
    } else {
      const int f = -dst[r] - 1;
      int* fs = fin_seq + ((long)b * W + f) * Tcap;
      for (int p = tid; p < len; p += 1024) fs[p] = seq_in[(long)orow * Tcap + p];
      if (tid == 0) { fs[len] = tok; fin_score[b * W + f] = cval[r]; fin_len[b * W + f] = len + 1; }
    }
  }
  if (tid >= s_nlive && tid < W) next_tok[b * W + tid] = 0;
}

}  // namespace

#define ST static_cast<hipStream_t>(stream)

extern "C" {

int ick_beam_self_attn(const float* qkv, float* kcache, float* vcache, const int32_t* anc, float* out, int rows, int E, int heads,
                       int t, int Tcap, void* stream) {
  ICK_REQUIRE(qkv && kcache && vcache && anc && out && rows > 0 && heads > 0 && E == heads * HD && t >= 0 && t < Tcap,
              "ick_beam_self_attn: bad arguments (head dim 64, 0 <= t < Tcap)");
  ICK_LAUNCH(beam_self_attn_kernel, dim3(rows, heads), dim3(64), 0, ST, qkv, kcache, vcache, anc, out, rows, E, t,
             1.0f / sqrtf((float)HD));
  return ick::launch_status("beam_self_attn");
}

int ick_beam_step(const float* logits, float* score, int32_t* width, const int32_t* seq_in, int32_t* seq_out, const int32_t* anc_in,
                  int32_t* anc_out, int64_t* next_tok, int32_t* fin_seq, float* fin_score, int32_t* fin_len, int32_t* nfin, int B,
                  int W, int V, int Tcap, int t, int end_id, void* stream) {
  ICK_REQUIRE(logits && score && width && seq_in && seq_out && anc_in && anc_out && next_tok && fin_seq && fin_score && fin_len &&
              nfin && B > 0 && W > 0 && W <= 32 && V >= W && t >= 0 && t + 1 < Tcap && seq_in != seq_out && anc_in != anc_out,
              "ick_beam_step: bad arguments (W <= 32, t + 1 < Tcap, double-buffered seq / anc)");
  ICK_LAUNCH(beam_step_kernel, dim3(B), dim3(1024), 0, ST, logits, score, (int*)width, (const int*)seq_in, (int*)seq_out,
             (const int*)anc_in, (int*)anc_out, (long*)next_tok, (int*)fin_seq, fin_score, (int*)fin_len, (int*)nfin, B, W, V, Tcap, t,
             end_id);
  return ick::launch_status("beam_step");
}

}  // extern "C"
