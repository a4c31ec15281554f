// losses.hip — fused forward+backward of the knowledge-distillation losses
// (reference: DistillationLoss, /root/reference/src/distillation_utils.py:8-200).
// HBM-bound: each kernel reads its inputs once, produces the per-row loss terms AND the gradient in the same
// pass (token KL + CE: read s, read t, write ds = the algorithmic 3 x T*B*V*4 bytes), then a single small
// kernel reduces the per-row terms deterministically and combines them with the alpha/beta/gamma weights.
#include "ick_common.h"

namespace {

constexpr int NT = 256;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

// block-wide reductions through LDS scratch red[NT/64]
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int i = 0; i < NT / 64; ++i) t += red[i];
  return t;
}
__device__ __forceinline__ float block_max(float v, float* red) {
  v = wave_max(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float t = red[0];
#pragma unroll
  for (int i = 1; i < NT / 64; ++i) t = fmaxf(t, red[i]);
  return t;
}

// ------------------------------------------------------------------ token-level KL (temperature tau) + cross entropy
// (distillation_utils.py:30-54 and :154).  One workgroup per row r of the (T*B, V) logits; the row pair is staged
// in LDS so HBM sees each logit once.  Outputs: row_kl[r] = sum_v p_t (log p_t - log p_s) (the caller's reduce
// applies tau^2 / rows), row_ce[r] = -log softmax(s)[target] (0 for PAD), and
//   ds[r][v] = g_kd * (p_s^tau - p_t^tau) + g_ce[r] * (softmax(s) - onehot(target))
// with g_kd = grad_scale*alpha*tau/rows and g_ce = grad_scale*w_ce/n_valid (0 on PAD rows).
__global__ __launch_bounds__(NT) void token_kd_ce_kernel(const float* __restrict__ s, const float* __restrict__ t,
                                                        const long* __restrict__ targets, float* __restrict__ ds,
                                                        float* __restrict__ row_kl, float* __restrict__ row_ce,
                                                        const int* __restrict__ n_valid, int V, float inv_tau,
                                                        float g_kd, float g_ce_num) {
  extern __shared__ __attribute__((aligned(16))) float sh[];  // [V] s, [V] t, [8] scratch
  float* ss = sh;
  float* ts = sh + V;
  float* red = sh + 2 * V;
  const long r = blockIdx.x;
  const float* sr = s + r * V;
  const float* tr = t + r * V;
  const int tid = threadIdx.x;
  float ms = -INFINITY, mt = -INFINITY;
  if ((V & 3) == 0) {
    for (int i = tid; i < (V >> 2); i += NT) {
      const float4 a = reinterpret_cast<const float4*>(sr)[i], b = reinterpret_cast<const float4*>(tr)[i];
      reinterpret_cast<float4*>(ss)[i] = a; reinterpret_cast<float4*>(ts)[i] = b;
      ms = fmaxf(ms, fmaxf(fmaxf(a.x, a.y), fmaxf(a.z, a.w)));
      mt = fmaxf(mt, fmaxf(fmaxf(b.x, b.y), fmaxf(b.z, b.w)));
    }
  } else {
    for (int i = tid; i < V; i += NT) {
      const float a = sr[i], b = tr[i];
      ss[i] = a; ts[i] = b; ms = fmaxf(ms, a); mt = fmaxf(mt, b);
    }
  }
  ms = block_max(ms, red);
  mt = block_max(mt, red);
  // sums of exp at temperature tau (student, teacher) and at temperature 1 (student, for CE)
  float zs = 0.f, zt = 0.f, z1 = 0.f;
  for (int i = tid; i < V; i += NT) {
    zs += expf((ss[i] - ms) * inv_tau);
    zt += expf((ts[i] - mt) * inv_tau);
    z1 += expf(ss[i] - ms);
  }
  zs = block_sum(zs, red);
  zt = block_sum(zt, red);
  z1 = block_sum(z1, red);
  const float lzs = logf(zs), lzt = logf(zt), lz1 = logf(z1);
  const long tgt = targets ? targets[r] : 0;
  const bool valid = tgt != 0;
  const float g_ce = valid ? g_ce_num / (float)max(*n_valid, 1) : 0.f;
  const float izs = 1.f / zs, izt = 1.f / zt, iz1 = 1.f / z1;
  float kl = 0.f;
  float* dr = ds + r * V;
  for (int i = tid; i < V; i += NT) {
    const float a = (ss[i] - ms) * inv_tau, b = (ts[i] - mt) * inv_tau;
    const float ps = expf(a) * izs, pt = expf(b) * izt;
    kl += pt * ((b - lzt) - (a - lzs));
    float g = g_kd * (ps - pt);
    if (valid) g += g_ce * (expf(ss[i] - ms) * iz1 - (i == tgt ? 1.f : 0.f));
    dr[i] = g;
  }
  kl = block_sum(kl, red);
  if (tid == 0) {
    row_kl[r] = kl;
    row_ce[r] = valid ? -((ss[tgt] - ms) - lz1) : 0.f;
  }
}

__global__ void count_valid_kernel(const long* __restrict__ targets, int n, int* __restrict__ out) {
  __shared__ float red[NT / 64];
  float c = 0.f;
  for (int i = threadIdx.x; i < n; i += NT) c += targets[i] != 0 ? 1.f : 0.f;
  c = block_sum(c, red);
  if (threadIdx.x == 0) *out = (int)(c + 0.5f);
}

// ------------------------------------------------------------------ encoder-feature loss (distillation_utils.py:56-94)
// one workgroup per image b; s,t (L,E).  Writes part[b] = {sum_e (mean_s-mean_t)^2, sum_e (sw-tw)^2} and the gradients
// ds, dt (either may be NULL).  gscale = grad_scale*beta; MSE means over B*E.
__global__ __launch_bounds__(NT) void feature_kd_kernel(const float* __restrict__ s, const float* __restrict__ t,
                                                       float* __restrict__ ds, float* __restrict__ dt,
                                                       float* __restrict__ part, int B, int L, int E, float gscale) {
  extern __shared__ __attribute__((aligned(16))) float sh[];
  const int Lp = (L + 3) & ~3;
  float* as_ = sh;             // [L] softmax weights (student)
  float* at_ = sh + Lp;        // [L] (teacher)
  float* qs = sh + 2 * Lp;     // [L] q_j student
  float* qt = sh + 3 * Lp;     // [L]
  float* Dg = sh + 4 * Lp;     // [E] d(loss)/d(mean diff)
  float* Da = Dg + E;          // [E] d(loss)/d(weighted diff)
  float* red = Da + E;         // [8]
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* S = s + (long)b * L * E;
  const float* T = t + (long)b * L * E;
  // row sums -> softmax over positions
  for (int j = wave; j < L; j += NT / 64) {
    float a = 0.f, c = 0.f;
    for (int e = lane; e < E; e += 64) { a += S[(long)j * E + e]; c += T[(long)j * E + e]; }
    a = wave_sum(a); c = wave_sum(c);
    if (lane == 0) { as_[j] = a; at_[j] = c; }
  }
  __syncthreads();
  if (wave < 2) {
    float* a = wave == 0 ? as_ : at_;
    float mx = -INFINITY;
    for (int j = lane; j < L; j += 64) mx = fmaxf(mx, a[j]);
    mx = wave_max(mx);
    float z = 0.f;
    for (int j = lane; j < L; j += 64) z += expf(a[j] - mx);
    z = wave_sum(z);
    for (int j = lane; j < L; j += 64) a[j] = expf(a[j] - mx) / z;
  }
  __syncthreads();
  const float invBE = 1.f / ((float)B * (float)E);
  float pg = 0.f, pa = 0.f;
  for (int e = tid; e < E; e += NT) {
    float ms = 0.f, mt = 0.f, ws = 0.f, wt = 0.f;
    for (int j = 0; j < L; ++j) {
      const float x = S[(long)j * E + e], y = T[(long)j * E + e];
      ms += x; mt += y; ws += as_[j] * x; wt += at_[j] * y;
    }
    const float dg = (ms - mt) / L, da = ws - wt;
    pg += dg * dg; pa += da * da;
    Dg[e] = gscale * 0.6f * 2.f * dg * invBE / L;   // d/ds[j,e] of the global term
    Da[e] = gscale * 0.4f * 2.f * da * invBE;       // D[e] of the attention term
  }
  pg = block_sum(pg, red);
  pa = block_sum(pa, red);
  if (tid == 0) { part[2 * b] = pg; part[2 * b + 1] = pa; }
  __syncthreads();
  // q_j = sum_e Da[e] * x[j,e]
  for (int j = wave; j < L; j += NT / 64) {
    float a = 0.f, c = 0.f;
    for (int e = lane; e < E; e += 64) { a += Da[e] * S[(long)j * E + e]; c += Da[e] * T[(long)j * E + e]; }
    a = wave_sum(a); c = wave_sum(c);
    if (lane == 0) { qs[j] = a; qt[j] = c; }
  }
  __syncthreads();
  float qbs = 0.f, qbt = 0.f;  // sum_i a_i q_i (every thread computes it: L is tiny)
  for (int j = 0; j < L; ++j) { qbs += as_[j] * qs[j]; qbt += at_[j] * qt[j]; }
  for (int i = tid; i < L * E; i += NT) {
    const int j = i / E, e = i - j * E;
    if (ds) ds[(long)b * L * E + i] = Dg[e] + as_[j] * (Da[e] + qs[j] - qbs);
    if (dt) dt[(long)b * L * E + i] = -(Dg[e] + at_[j] * (Da[e] + qt[j] - qbt));
  }
}

// ------------------------------------------------------------------ hidden-state loss (distillation_utils.py:96-136)
// rows = n_steps*B; one wave per row: 0.7*MSE + 0.3*(1-cos) with nn.CosineEmbeddingLoss's EPSILON=1e-12.
// part[r] = {sum_h (s-t)^2, 1-cos}; ds = gscale/n_steps * (0.7*2(s-t)/(B*H) + 0.3/B * d(1-cos)/ds)
__global__ void hidden_kd_kernel(const float* __restrict__ s, const float* __restrict__ t, float* __restrict__ ds,
                                 float* __restrict__ part, long rows, int B, int H, float gscale_over_n) {
  const int lane = threadIdx.x & 63, wpb = blockDim.x >> 6;
  for (long r = blockIdx.x * (long)wpb + (threadIdx.x >> 6); r < rows; r += (long)gridDim.x * wpb) {
    const float* sr = s + r * H; const float* tr = t + r * H;
    float dot = 0.f, ns = 0.f, nt = 0.f, se = 0.f;
    for (int i = lane; i < H; i += 64) {
      const float a = sr[i], b = tr[i];
      dot += a * b; ns += a * a; nt += b * b; se += (a - b) * (a - b);
    }
    dot = wave_sum(dot); ns = wave_sum(ns) + 1e-12f; nt = wave_sum(nt) + 1e-12f; se = wave_sum(se);
    const float den = sqrtf(ns * nt);
    const float cosv = dot / den;
    if (lane == 0) { part[2 * r] = se; part[2 * r + 1] = 1.f - cosv; }
    if (ds) {
      const float kmse = gscale_over_n * 0.7f * 2.f / ((float)B * (float)H);
      const float kcos = gscale_over_n * 0.3f / (float)B;
      for (int i = lane; i < H; i += 64) {
        const float a = sr[i], b = tr[i];
        ds[r * H + i] = kmse * (a - b) - kcos * (b / den - cosv * a / ns);
      }
    }
  }
}

// ------------------------------------------------------------------ deterministic reduce + combine
// out[0..4] = total, ce, token_kd, feature_kd, hidden_kd  (loss_dict order of distillation_utils.py:192-198)
__global__ void kd_combine_kernel(const float* __restrict__ row_kl, const float* __restrict__ row_ce, int rows,
                                  const int* __restrict__ n_valid, const float* __restrict__ feat_part, int Bf, int Ef,
                                  const float* __restrict__ hid_part, int hid_rows, int hid_B, int hid_H, int hid_steps,
                                  float w_ce, float alpha, float beta, float gamma, float tau, float* __restrict__ out) {
  __shared__ float red[NT / 64];
  float kl = 0.f, ce = 0.f;
  for (int i = threadIdx.x; i < rows; i += NT) { kl += row_kl[i]; ce += row_ce[i]; }
  kl = block_sum(kl, red); ce = block_sum(ce, red);
  float fg = 0.f, fa = 0.f;
  if (feat_part) for (int i = threadIdx.x; i < Bf; i += NT) { fg += feat_part[2 * i]; fa += feat_part[2 * i + 1]; }
  fg = block_sum(fg, red); fa = block_sum(fa, red);
  float hm = 0.f, hc = 0.f;
  if (hid_part) for (int i = threadIdx.x; i < hid_rows; i += NT) { hm += hid_part[2 * i]; hc += hid_part[2 * i + 1]; }
  hm = block_sum(hm, red); hc = block_sum(hc, red);
  if (threadIdx.x == 0) {
    const float kd = rows > 0 ? kl / rows * tau * tau : 0.f;
    const float cel = (n_valid && *n_valid > 0) ? ce / *n_valid : 0.f;
    const float feat = feat_part ? (0.6f * fg + 0.4f * fa) / ((float)Bf * Ef) : 0.f;
    const float hid = hid_part ? (0.7f * hm / ((float)hid_B * hid_H) + 0.3f * hc / hid_B) / hid_steps : 0.f;
    out[0] = w_ce * cel + alpha * kd + beta * feat + gamma * hid;
    out[1] = cel; out[2] = kd; out[3] = feat; out[4] = hid;
  }
}

// ------------------------------------------------------------------ OptimizedDistillationLoss
// reference: /root/reference/src/train_student_kd_optimized.py:34-128.  Same row-in-LDS structure as token_kd_ce_kernel:
//   row_kd[r]   = -sum_v p_t log p_s            (soft-target cross entropy at temperature tau; the reduce applies tau^2/rows)
//   row_hard[r] = focal_alpha (1 - p)^focal_gamma ce,  ce = -log softmax(s)[target], p = exp(-ce)   (every row: no ignore_index)
//   ds[r][v]    = g_kd (p_s - p_t) + g_hard focal'(ce) (softmax(s)[v] - [v == target])
// with g_kd = grad_scale * alpha_now * tau / rows, g_hard = grad_scale * (1 - alpha_now) / rows and
// focal'(ce) = focal_alpha [(1-p)^gamma + gamma (1-p)^(gamma-1) p ce].
__global__ __launch_bounds__(NT) void token_softce_focal_kernel(const float* __restrict__ s, const float* __restrict__ t,
                                                               const long* __restrict__ targets, float* __restrict__ ds,
                                                               float* __restrict__ row_kd, float* __restrict__ row_hard,
                                                               int V, float inv_tau, float g_kd, float g_hard,
                                                               float focal_alpha, float focal_gamma) {
  extern __shared__ __attribute__((aligned(16))) float sh[];
  float* ss = sh;
  float* ts = sh + V;
  float* red = sh + 2 * V;
  const long r = blockIdx.x;
  const float* sr = s + r * V;
  const float* tr = t + r * V;
  const int tid = threadIdx.x;
  float ms = -INFINITY, mt = -INFINITY;
  for (int i = tid; i < V; i += NT) {
    const float a = sr[i], b = tr[i];
    ss[i] = a; ts[i] = b; ms = fmaxf(ms, a); mt = fmaxf(mt, b);
  }
  ms = block_max(ms, red);
  mt = block_max(mt, red);
  float zs = 0.f, zt = 0.f, z1 = 0.f;
  for (int i = tid; i < V; i += NT) {
    zs += expf((ss[i] - ms) * inv_tau);
    zt += expf((ts[i] - mt) * inv_tau);
    z1 += expf(ss[i] - ms);
  }
  zs = block_sum(zs, red);
  zt = block_sum(zt, red);
  z1 = block_sum(z1, red);
  const float lzs = logf(zs), lz1 = logf(z1);
  const long tgt = targets[r];
  const float ce = -((ss[tgt] - ms) - lz1);
  const float p = expf(-ce), om = fmaxf(1.f - p, 0.f);
  const float pw = powf(om, focal_gamma);
  const float dfocal = focal_alpha * (pw + (om > 0.f ? focal_gamma * powf(om, focal_gamma - 1.f) * p * ce : 0.f));
  const float gh = g_hard * dfocal;
  const float izs = 1.f / zs, izt = 1.f / zt, iz1 = 1.f / z1;
  float kd = 0.f;
  float* dr = ds + r * V;
  for (int i = tid; i < V; i += NT) {
    const float a = (ss[i] - ms) * inv_tau;
    const float ps = expf(a) * izs, pt = expf((ts[i] - mt) * inv_tau) * izt;
    kd -= pt * (a - lzs);
    dr[i] = g_kd * (ps - pt) + gh * (expf(ss[i] - ms) * iz1 - (i == tgt ? 1.f : 0.f));
  }
  kd = block_sum(kd, red);
  if (tid == 0) { row_kd[r] = kd; row_hard[r] = focal_alpha * pw * ce; }
}

// per-token cosine feature loss (:84-94): one wave per token row; cos = <s,t> / (max(|s|,eps) max(|t|,eps)), eps = 1e-12
// (F.normalize).  part[row] = cos; ds, dt (either may be NULL) = -gscale/rows * d cos.
__global__ __launch_bounds__(NT) void feature_cosine_kernel(const float* __restrict__ s, const float* __restrict__ t,
                                                           float* __restrict__ ds, float* __restrict__ dt,
                                                           float* __restrict__ part, long rows, int E, float g) {
  const int lane = threadIdx.x & 63;
  const long row = blockIdx.x * (long)(NT / 64) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* a = s + row * E;
  const float* b = t + row * E;
  float saa = 0.f, sbb = 0.f, sab = 0.f;
  for (int e = lane; e < E; e += 64) { const float x = a[e], y = b[e]; saa += x * x; sbb += y * y; sab += x * y; }
  saa = wave_sum(saa); sbb = wave_sum(sbb); sab = wave_sum(sab);
  const float na = fmaxf(sqrtf(saa), 1e-12f), nb = fmaxf(sqrtf(sbb), 1e-12f);
  const float cosv = sab / (na * nb);
  if (lane == 0) part[row] = cosv;
  // d cos / d a = b / (na nb) - cos a / na^2   (for |a| > eps)
  const float ia = 1.f / na, ib = 1.f / nb;
  for (int e = lane; e < E; e += 64) {
    const float x = a[e], y = b[e];
    if (ds) ds[row * E + e] = -g * (y * ia * ib - cosv * x * ia * ia);
    if (dt) dt[row * E + e] = -g * (x * ia * ib - cosv * y * ib * ib);
  }
}

// hidden term (:101-110): MSE over (B,H) of the attention-weighted time sums; w (T,B) already softmaxed over time.
// one workgroup per batch row b: part[b] = sum_h (ws - wt)^2 ; ds[t][b][h] = g * 2 (ws - wt) w[t][b] / (B H)
__global__ __launch_bounds__(NT) void weighted_hidden_mse_kernel(const float* __restrict__ s, const float* __restrict__ t,
                                                                const float* __restrict__ w, float* __restrict__ ds,
                                                                float* __restrict__ part, int T, int B, int H, float g) {
  __shared__ float red[NT / 64];
  const int b = blockIdx.x;
  float acc = 0.f;
  for (int h = threadIdx.x; h < H; h += NT) {
    float d = 0.f;
    for (int k = 0; k < T; ++k) d += w[k * B + b] * (s[((long)k * B + b) * H + h] - t[((long)k * B + b) * H + h]);
    acc += d * d;
    if (ds) for (int k = 0; k < T; ++k) ds[((long)k * B + b) * H + h] = g * 2.f * d * w[k * B + b] / ((float)B * H);
  }
  acc = block_sum(acc, red);
  if (threadIdx.x == 0) part[b] = acc;
}

// out7 = {total, token, feature, hidden, kd, hard, hard} (the reference's dict order :120-128), deterministic reduce
__global__ void optloss_combine_kernel(const float* __restrict__ row_kd, const float* __restrict__ row_hard, int rows,
                                       const float* __restrict__ cos_part, long cos_rows, const float* __restrict__ hid_part,
                                       int hid_B, int hid_H, float alpha_now, float beta_now, float gamma_now, float tau,
                                       float* __restrict__ out) {
  __shared__ float red[NT / 64];
  float kd = 0.f, hd = 0.f, cs = 0.f, hm = 0.f;
  for (int i = threadIdx.x; i < rows; i += NT) { kd += row_kd[i]; hd += row_hard[i]; }
  if (cos_part) for (long i = threadIdx.x; i < cos_rows; i += NT) cs += cos_part[i];
  if (hid_part) for (int i = threadIdx.x; i < hid_B; i += NT) hm += hid_part[i];
  kd = block_sum(kd, red); hd = block_sum(hd, red); cs = block_sum(cs, red); hm = block_sum(hm, red);
  if (threadIdx.x == 0) {
    const float kdl = kd / rows * tau * tau, hard = hd / rows;
    const float token = alpha_now * kdl + (1.f - alpha_now) * hard;
    const float feat = cos_part ? 1.f - cs / (float)cos_rows : 0.f;
    const float hid = hid_part ? hm / ((float)hid_B * hid_H) : 0.f;
    out[0] = token + beta_now * feat + gamma_now * hid;
    out[1] = token; out[2] = feat; out[3] = hid; out[4] = kdl; out[5] = hard; out[6] = hard;
  }
}

// x[i] *= *scalar (device scalar) — applies autograd's incoming grad_output to a precomputed gradient
__global__ void scale_by_scalar_kernel(float* __restrict__ x, const float* __restrict__ sc, long n) {
  const float k = *sc;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) x[i] *= k;
}

}  // namespace

#define ST static_cast<hipStream_t>(stream)

extern "C" {

int ick_count_valid(const int64_t* targets, int n, int* out, void* stream) {
  ICK_REQUIRE(targets && out && n > 0, "ick_count_valid: bad arguments");
  ICK_LAUNCH(count_valid_kernel, dim3(1), dim3(NT), 0, ST, (const long*)targets, n, out);
  return ick::launch_status("count_valid");
}

int ick_token_kd_ce(const float* s, const float* t, const int64_t* targets, float* ds, float* row_kl, float* row_ce,
                    const int* n_valid, int rows, int V, float tau, float g_kd, float g_ce_num, void* stream) {
  ICK_REQUIRE(s && t && ds && row_kl && row_ce && n_valid && rows > 0 && V > 0 && tau > 0, "ick_token_kd_ce: bad arguments");
  const size_t sh = ((size_t)2 * V + 8) * sizeof(float);
  ICK_REQUIRE(sh <= 160 * 1024, "ick_token_kd_ce: vocabulary %d too large for the LDS-staged row (max %d)", V, (160 * 1024 / 4 - 8) / 2);
  if (sh > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(token_kd_ce_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
    if (e != hipSuccess) return ick::fail((int)e, "ick_token_kd_ce: cannot raise dynamic LDS to %zu", sh);
  }
  ICK_LAUNCH(token_kd_ce_kernel, dim3(rows), dim3(NT), sh, ST, s, t, (const long*)targets, ds, row_kl, row_ce, n_valid,
                     V, 1.0f / tau, g_kd, g_ce_num);
  return ick::launch_status("token_kd_ce");
}

int ick_feature_kd(const float* s, const float* t, float* ds, float* dt, float* part, int B, int L, int E, float gscale,
                   void* stream) {
  ICK_REQUIRE(s && t && part && B > 0 && L > 0 && E > 0, "ick_feature_kd: bad arguments");
  const size_t sh = (4 * ((L + 3) & ~3) + 2 * E + 8) * sizeof(float);
  ICK_LAUNCH(feature_kd_kernel, dim3(B), dim3(NT), sh, ST, s, t, ds, dt, part, B, L, E, gscale);
  return ick::launch_status("feature_kd");
}

int ick_hidden_kd(const float* s, const float* t, float* ds, float* part, int steps, int B, int H, float gscale, void* stream) {
  ICK_REQUIRE(s && t && part && steps > 0 && B > 0 && H > 0, "ick_hidden_kd: bad arguments");
  const long rows = (long)steps * B;
  long g = (rows + 3) / 4; if (g > 2048) g = 2048;
  ICK_LAUNCH(hidden_kd_kernel, dim3((int)g), dim3(NT), 0, ST, s, t, ds, part, rows, B, H, gscale / steps);
  return ick::launch_status("hidden_kd");
}

int ick_kd_combine(const float* row_kl, const float* row_ce, int rows, const int* n_valid, const float* feat_part, int Bf,
                   int Ef, const float* hid_part, int hid_steps, int hid_B, int hid_H, float w_ce, float alpha, float beta,
                   float gamma, float tau, float* out5, void* stream) {
  ICK_REQUIRE(row_kl && row_ce && out5 && rows > 0, "ick_kd_combine: bad arguments");
  ICK_LAUNCH(kd_combine_kernel, dim3(1), dim3(NT), 0, ST, row_kl, row_ce, rows, n_valid, feat_part, Bf, Ef, hid_part,
                     hid_steps * hid_B, hid_B, hid_H, hid_steps > 0 ? hid_steps : 1, w_ce, alpha, beta, gamma, tau, out5);
  return ick::launch_status("kd_combine");
}

int ick_token_softce_focal(const float* s, const float* t, const int64_t* targets, float* ds, float* row_kd, float* row_hard,
                           int rows, int V, float tau, float g_kd, float g_hard, float focal_alpha, float focal_gamma,
                           void* stream) {
  ICK_REQUIRE(s && t && targets && ds && row_kd && row_hard && rows > 0 && V > 0 && tau > 0, "ick_token_softce_focal: bad arguments");
  const size_t sh = ((size_t)2 * V + 8) * sizeof(float);
  ICK_REQUIRE(sh <= 160 * 1024, "ick_token_softce_focal: vocabulary %d too large for the LDS-staged row", V);
  if (sh > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(token_softce_focal_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
    if (e != hipSuccess) return ick::fail((int)e, "ick_token_softce_focal: cannot raise dynamic LDS to %zu", sh);
  }
  ICK_LAUNCH(token_softce_focal_kernel, dim3(rows), dim3(NT), sh, ST, s, t, (const long*)targets, ds, row_kd, row_hard, V,
             1.0f / tau, g_kd, g_hard, focal_alpha, focal_gamma);
  return ick::launch_status("token_softce_focal");
}

int ick_feature_cosine(const float* s, const float* t, float* ds, float* dt, float* part, int64_t rows, int E, float gscale,
                       void* stream) {
  ICK_REQUIRE(s && t && part && rows > 0 && E > 0, "ick_feature_cosine: bad arguments");
  ICK_LAUNCH(feature_cosine_kernel, dim3((int)((rows + NT / 64 - 1) / (NT / 64))), dim3(NT), 0, ST, s, t, ds, dt, part, (long)rows, E,
             gscale / (float)rows);
  return ick::launch_status("feature_cosine");
}

int ick_weighted_hidden_mse(const float* s, const float* t, const float* w, float* ds, float* part, int T, int B, int H,
                            float gscale, void* stream) {
  ICK_REQUIRE(s && t && w && part && T > 0 && B > 0 && H > 0, "ick_weighted_hidden_mse: bad arguments");
  ICK_LAUNCH(weighted_hidden_mse_kernel, dim3(B), dim3(NT), 0, ST, s, t, w, ds, part, T, B, H, gscale);
  return ick::launch_status("weighted_hidden_mse");
}

int ick_optloss_combine(const float* row_kd, const float* row_hard, int rows, const float* cos_part, int64_t cos_rows,
                        const float* hid_part, int hid_B, int hid_H, float alpha_now, float beta_now, float gamma_now,
                        float tau, float* out7, void* stream) {
  ICK_REQUIRE(row_kd && row_hard && out7 && rows > 0, "ick_optloss_combine: bad arguments");
  ICK_LAUNCH(optloss_combine_kernel, dim3(1), dim3(NT), 0, ST, row_kd, row_hard, rows, cos_part, (long)cos_rows, hid_part,
             hid_B, hid_H, alpha_now, beta_now, gamma_now, tau, out7);
  return ick::launch_status("optloss_combine");
}

int ick_scale_by_scalar(float* x, const float* scalar, int64_t n, void* stream) {
  ICK_REQUIRE(x && scalar && n > 0, "ick_scale_by_scalar: bad arguments");
  long g = (n + NT - 1) / NT; if (g > 2048) g = 2048;
  ICK_LAUNCH(scale_by_scalar_kernel, dim3((int)g), dim3(NT), 0, ST, x, scalar, (long)n);
  return ick::launch_status("scale_by_scalar");
}

}  // extern "C"
