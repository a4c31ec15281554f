// optim.hip — the optimizer tail of the KD step as flat-buffer kernels
// (reference: /root/reference/src/train_student_kd.py:292-299 — clip_grad_norm_(max_norm=1.0) then AdamW(wd=0.01)).
// Parameters, gradients and both Adam moments live in flat fp32 buffers (one segment per LR group), so the
// global gradient norm is one streaming reduction and the update is one fused pass per group:
// read p,g,m,v / write p,m,v = 28 B per parameter (HBM-bound).
#include "ick_common.h"

namespace {

constexpr int NT = 256;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

__global__ void sumsq_partial_kernel(const float* __restrict__ x, long n, float* __restrict__ partial) {
  __shared__ float red[NT / 64];
  float s = 0.f;
  const long n4 = n >> 2;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const float4 v = reinterpret_cast<const float4*>(x)[i];
    s += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
  }
  if (blockIdx.x == 0) for (long i = (n4 << 2) + threadIdx.x; i < n; i += blockDim.x) s += x[i] * x[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// norm_out[0] = sqrt(sum partial [+ norm_out[0]^2 if accumulate]) ; deterministic single-block tree
__global__ void norm_finalize_kernel(const float* __restrict__ partial, int n, float* __restrict__ norm_out, int accumulate) {
  __shared__ float red[NT / 64];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += NT) s += partial[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = (red[0] + red[1]) + (red[2] + red[3]);
    if (accumulate) t += norm_out[0] * norm_out[0];
    norm_out[0] = sqrtf(t);
  }
}

// torch.optim.AdamW single step with the clip coefficient folded in:
// (hyper != NULL: lr, 1-beta1^t, 1-beta2^t are read from device memory so a captured graph follows the schedule)
//   g' = g * min(1, max_norm / (norm + 1e-6)) * inv_scale ; p *= 1 - lr*wd ; m,v EMA ; p -= lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps)
__global__ void adamw_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                             long n, float lr, float b1, float b2, float eps, float wd, float bc1, float bc2,
                             const float* __restrict__ norm, float max_norm, float inv_scale, int write_clipped,
                             const float* __restrict__ hyper, const float* __restrict__ scaler) {
  if (hyper) {  // device-resident schedule (hipGraph replay); hyper[3] > 0: this step's beta1 (OneCycleLR's momentum cycling)
    lr = hyper[0]; bc1 = hyper[1]; bc2 = hyper[2];
    if (hyper[3] > 0.f) b1 = hyper[3];
  }
  if (scaler) {   // dynamic loss scaling (torch.amp.GradScaler semantics): scaler = {scale, 1/scale, found_inf}
    if (scaler[2] != 0.f) return;                                 // scaler.step(): skip the update on inf / nan gradients
    inv_scale *= scaler[1];                                       // scaler.unscale_()
  }
  float coef = inv_scale;
  if (norm) { const float c = max_norm / (norm[0] * inv_scale + 1e-6f); coef *= (c < 1.f ? c : 1.f); }
  const float step = lr / bc1, isb2 = rsqrtf(bc2), decay = 1.f - lr * wd;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float gg = g[i] * coef;
    const float mm = b1 * m[i] + (1.f - b1) * gg;
    const float vv = b2 * v[i] + (1.f - b2) * gg * gg;
    m[i] = mm; v[i] = vv;
    p[i] = p[i] * decay - step * mm / (sqrtf(vv) * isb2 + eps);
    if (write_clipped) g[i] = gg;
  }
}

// GradScaler bookkeeping on device (so a captured graph carries it): st = {scale, 1/scale, found_inf, good_steps}.
// check: found_inf = any of the n_norms gradient norms (of the SCALED gradients) is inf / nan       (scaler.unscale_)
// update: found_inf ? scale *= backoff, good_steps = 0 : (++good_steps == interval ? scale *= growth, good_steps = 0)
__global__ void loss_scale_check_kernel(const float* __restrict__ norms, int n_norms, float* __restrict__ st) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    bool bad = false;
    for (int i = 0; i < n_norms; ++i) bad = bad || !isfinite(norms[i]);
    st[2] = bad ? 1.f : 0.f;
  }
}
__global__ void loss_scale_update_kernel(float* __restrict__ st, float growth, float backoff, int interval) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    float scale = st[0], good = st[3];
    if (st[2] != 0.f) { scale *= backoff; good = 0.f; }
    else if (++good >= (float)interval) { scale *= growth; good = 0.f; }
    st[0] = scale; st[1] = 1.f / scale; st[3] = good;
  }
}

__global__ void adam_bias_correction_kernel(long* __restrict__ steps, const float* __restrict__ scaler, double b1, double b2,
                                            float* __restrict__ hyper, int n_groups, int stride,
                                            const float* __restrict__ lr_in, const float* __restrict__ beta1_in) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  long t = steps[0];
  if (!scaler || scaler[2] == 0.f) steps[0] = ++t; else ++t;   // skipped step: the values are unused (adamw returns early)
  const float bc2 = (float)(1.0 - pow(b2, (double)t));
  for (int g = 0; g < n_groups; ++g) {
    // beta1_in: this step's beta1 per group (torch's OneCycleLR cycles it against the learning rate); torch.optim.AdamW then
    // forms 1 - beta1^t from the CURRENT beta1 — so does this
    const double b1g = beta1_in ? (double)beta1_in[g] : b1;
    hyper[g * stride + 1] = (float)(1.0 - pow(b1g, (double)t)); hyper[g * stride + 2] = bc2;
    if (beta1_in) hyper[g * stride + 3] = beta1_in[g];
    if (lr_in) hyper[g * stride] = lr_in[g];   // this step's learning rates, uploaded as ONE contiguous block by the host
  }
}

}  // namespace

#define ST static_cast<hipStream_t>(stream)

extern "C" {

// norm_out[0] = ||x||_2 (accumulate != 0: combines with the norm already stored there: sqrt(old^2 + ||x||^2)).
// workspace: >= 1024 floats.
int ick_grad_norm(const float* x, int64_t n, float* workspace, float* norm_out, int accumulate, void* stream) {
  ICK_REQUIRE(x && workspace && norm_out && n > 0 && ick::aligned16(x), "ick_grad_norm: bad arguments");
  long g = (n / 4 + NT - 1) / NT; if (g > 1024) g = 1024; if (g < 1) g = 1;
  ICK_LAUNCH(sumsq_partial_kernel, dim3((int)g), dim3(NT), 0, ST, x, (long)n, workspace);
  ICK_LAUNCH(norm_finalize_kernel, dim3(1), dim3(NT), 0, ST, workspace, (int)g, norm_out, accumulate);
  return ick::launch_status("grad_norm");
}

int ick_adamw_step(float* p, float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                   float weight_decay, int step, const float* norm, float max_norm, float inv_scale, int write_clipped,
                   const float* hyper, const float* scaler, void* stream) {
  ICK_REQUIRE(p && g && m && v && n > 0 && (step >= 1 || hyper), "ick_adamw_step: bad arguments");
  const float bc1 = 1.f - powf(beta1, (float)step), bc2 = 1.f - powf(beta2, (float)step);
  long gr = (n + NT - 1) / NT; if (gr > 4096) gr = 4096;
  ICK_LAUNCH(adamw_kernel, dim3((int)gr), dim3(NT), 0, ST, p, g, m, v, (long)n, lr, beta1, beta2, eps, weight_decay,
                     bc1, bc2, norm, max_norm, inv_scale, write_clipped, hyper, scaler);
  return ick::launch_status("adamw_step");
}

int ick_adam_bias_correction(int64_t* applied_steps, const float* scaler, double beta1, double beta2, float* hyper,
                             int n_groups, int stride, const float* lr_in, const float* beta1_in, void* stream) {
  ICK_REQUIRE(applied_steps && hyper && n_groups > 0 && stride >= 3 && beta1 > 0. && beta1 < 1. && beta2 > 0. && beta2 < 1.,
              "ick_adam_bias_correction: bad arguments");
  ICK_REQUIRE(!beta1_in || stride >= 4, "ick_adam_bias_correction: beta1_in needs hyper rows of at least 4 floats");
  ICK_LAUNCH(adam_bias_correction_kernel, dim3(1), dim3(64), 0, ST, (long*)applied_steps, scaler, beta1, beta2,
             hyper, n_groups, stride, lr_in, beta1_in);
  return ick::launch_status("adam_bias_correction");
}

int ick_loss_scale_check(const float* norms, int n_norms, float* state, void* stream) {
  ICK_REQUIRE(norms && state && n_norms > 0, "ick_loss_scale_check: bad arguments");
  ICK_LAUNCH(loss_scale_check_kernel, dim3(1), dim3(64), 0, ST, norms, n_norms, state);
  return ick::launch_status("loss_scale_check");
}

int ick_loss_scale_update(float* state, float growth_factor, float backoff_factor, int growth_interval, void* stream) {
  ICK_REQUIRE(state && growth_factor >= 1.f && backoff_factor > 0.f && backoff_factor <= 1.f && growth_interval > 0,
              "ick_loss_scale_update: bad arguments");
  ICK_LAUNCH(loss_scale_update_kernel, dim3(1), dim3(64), 0, ST, state, growth_factor, backoff_factor, growth_interval);
  return ick::launch_status("loss_scale_update");
}

}  // extern "C"
