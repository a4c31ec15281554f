// depthwise.hip — what MobileNetV2 (the compact student's backbone, /root/reference/src/student_model_compact.py:19-22)
// needs beyond the dense implicit-GEMM family: 3x3 depthwise convolution forward / data gradient / weight gradient over
// NHWC fp32, per-channel batch statistics of an activation, and the dot-product attention step of the compact decoder
// (:114-138).  All HBM-bound: one float4 of channels per thread, no MFMA (9 MACs per output element).
// Weights keep nn.Conv2d's depthwise layout (C, 1, 3, 3) contiguous, i.e. w[c][tap]: 36 scalar loads per thread, L1-resident.
#include "ick_common.h"

namespace {

constexpr int NT = 256;

inline int grid_for(long work_items, int per_block = NT, int cap = 4096) {
  long g = (work_items + per_block - 1) / per_block;
  if (g < 1) g = 1;
  if (g > cap) g = cap;
  return (int)g;
}

__device__ __forceinline__ void load_w9(const float* __restrict__ w, int c, float4 (&wk)[9]) {
  const float* p = w + (long)c * 36;          // 4 consecutive channels x 9 taps
#pragma unroll
  for (int t = 0; t < 9; ++t) wk[t] = make_float4(p[t], p[9 + t], p[18 + t], p[27 + t]);
}

// y[b][oy][ox][c] = sum_{r,s} x[b][oy*st - 1 + r][ox*st - 1 + s][c] * w[c][r][s]      (padding 1)
__global__ void dwconv3x3_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ y,
                                     int B, int H, int W, int C4, int Ho, int Wo, int st) {
  const long total = (long)B * Ho * Wo * C4;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4);
    long r = i / C4;
    const int ox = (int)(r % Wo); r /= Wo;
    const int oy = (int)(r % Ho);
    const long b = r / Ho;
    float4 wk[9];
    load_w9(w, c, wk);
    float4 a = make_float4(0, 0, 0, 0);
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      const int iy = oy * st - 1 + dy;
      if ((unsigned)iy >= (unsigned)H) continue;
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const int ix = ox * st - 1 + dx;
        if ((unsigned)ix >= (unsigned)W) continue;
        const float4 v = reinterpret_cast<const float4*>(x)[((b * H + iy) * W + ix) * C4 + c];
        const float4 k = wk[dy * 3 + dx];
        a.x = fmaf(v.x, k.x, a.x); a.y = fmaf(v.y, k.y, a.y); a.z = fmaf(v.z, k.z, a.z); a.w = fmaf(v.w, k.w, a.w);
      }
    }
    reinterpret_cast<float4*>(y)[i] = a;
  }
}

// dx[b][iy][ix][c] = sum over taps (r,s) and output pixels with oy*st - 1 + r == iy, ox*st - 1 + s == ix of dy * w[c][r][s]
__global__ void dwconv3x3_dgrad_kernel(const float* __restrict__ dy, const float* __restrict__ w, float* __restrict__ dx,
                                       int B, int H, int W, int C4, int Ho, int Wo, int st) {
  const long total = (long)B * H * W * C4;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4);
    long r = i / C4;
    const int ix = (int)(r % W); r /= W;
    const int iy = (int)(r % H);
    const long b = r / H;
    float4 wk[9];
    load_w9(w, c, wk);
    float4 a = make_float4(0, 0, 0, 0);
#pragma unroll
    for (int rr = 0; rr < 3; ++rr) {
      const int ty = iy + 1 - rr;
      if (ty < 0 || ty % st != 0) continue;
      const int oy = ty / st;
      if (oy >= Ho) continue;
#pragma unroll
      for (int ss = 0; ss < 3; ++ss) {
        const int tx = ix + 1 - ss;
        if (tx < 0 || tx % st != 0) continue;
        const int ox = tx / st;
        if (ox >= Wo) continue;
        const float4 v = reinterpret_cast<const float4*>(dy)[((b * Ho + oy) * Wo + ox) * C4 + c];
        const float4 k = wk[rr * 3 + ss];
        a.x = fmaf(v.x, k.x, a.x); a.y = fmaf(v.y, k.y, a.y); a.z = fmaf(v.z, k.z, a.z); a.w = fmaf(v.w, k.w, a.w);
      }
    }
    reinterpret_cast<float4*>(dx)[i] = a;
  }
}

// dw[c][r][s] += sum_{b,oy,ox} dy[b][oy][ox][c] * x[b][oy*st-1+r][ox*st-1+s][c].
// Block = ny x C4 threads (ny = 256 / C4 pixel lanes); grid-stride over output pixels; LDS combine over the pixel lanes;
// fp32 atomics (~200 blocks: the same contention lesson as bn_bwd_reduce).
__global__ void dwconv3x3_wgrad_kernel(const float* __restrict__ dy, const float* __restrict__ x, float* __restrict__ dw,
                                       int B, int H, int W, int C4, int Ho, int Wo, int st, int ny) {
  extern __shared__ __attribute__((aligned(16))) float sh[];      // [ny][C4][9] float4
  const int tc = threadIdx.x % C4, tr = threadIdx.x / C4;
  float4 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) acc[t] = make_float4(0, 0, 0, 0);
  const long npix = (long)B * Ho * Wo;
  for (long p = blockIdx.x * (long)ny + tr; p < npix; p += (long)gridDim.x * ny) {
    const int ox = (int)(p % Wo);
    const int oy = (int)((p / Wo) % Ho);
    const long b = p / ((long)Wo * Ho);
    const float4 g = reinterpret_cast<const float4*>(dy)[p * C4 + tc];
#pragma unroll
    for (int rr = 0; rr < 3; ++rr) {
      const int iy = oy * st - 1 + rr;
      if ((unsigned)iy >= (unsigned)H) continue;
#pragma unroll
      for (int ss = 0; ss < 3; ++ss) {
        const int ix = ox * st - 1 + ss;
        if ((unsigned)ix >= (unsigned)W) continue;
        const float4 v = reinterpret_cast<const float4*>(x)[((b * H + iy) * W + ix) * C4 + tc];
        float4& a = acc[rr * 3 + ss];
        a.x = fmaf(g.x, v.x, a.x); a.y = fmaf(g.y, v.y, a.y); a.z = fmaf(g.z, v.z, a.z); a.w = fmaf(g.w, v.w, a.w);
      }
    }
  }
  float4* s4 = reinterpret_cast<float4*>(sh);
#pragma unroll
  for (int t = 0; t < 9; ++t) s4[(tr * C4 + tc) * 9 + t] = acc[t];
  __syncthreads();
  if (tr == 0) {
    for (int r = 1; r < ny; ++r)
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const float4 v = s4[(r * C4 + tc) * 9 + t];
        acc[t].x += v.x; acc[t].y += v.y; acc[t].z += v.z; acc[t].w += v.w;
      }
    float* o = dw + (long)tc * 36;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      atomicAdd(o + t, acc[t].x); atomicAdd(o + 9 + t, acc[t].y); atomicAdd(o + 18 + t, acc[t].z); atomicAdd(o + 27 + t, acc[t].w);
    }
  }
}

// per-channel sum / sum of squares of x[M][C] in fp64 (BatchNorm batch statistics behind a depthwise convolution, whose
// kernel has no GEMM epilogue to produce them): block = ny x C4 threads, fp64 per thread, LDS combine, fp64 atomics
struct d4 { double x, y, z, w; };
__global__ void colstats_kernel(const float* __restrict__ x, double* __restrict__ sum, double* __restrict__ sq, long M, int C4, int ny) {
  extern __shared__ __attribute__((aligned(16))) float shf[];
  d4* sh = reinterpret_cast<d4*>(shf);                                 // [2][ny][C4]
  const int tc = threadIdx.x % C4, tr = threadIdx.x / C4;
  d4 s = {0, 0, 0, 0}, q = {0, 0, 0, 0};
  for (long m = blockIdx.x * (long)ny + tr; m < M; m += (long)gridDim.x * ny) {
    const float4 v = reinterpret_cast<const float4*>(x)[m * C4 + tc];
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    q.x = fma((double)v.x, (double)v.x, q.x); q.y = fma((double)v.y, (double)v.y, q.y);
    q.z = fma((double)v.z, (double)v.z, q.z); q.w = fma((double)v.w, (double)v.w, q.w);
  }
  sh[tr * C4 + tc] = s; sh[(ny + tr) * C4 + tc] = q;
  __syncthreads();
  if (tr == 0) {
    for (int r = 1; r < ny; ++r) {
      const d4 a = sh[r * C4 + tc], b = sh[(ny + r) * C4 + tc];
      s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
      q.x += b.x; q.y += b.y; q.z += b.z; q.w += b.w;
    }
    double* ps = sum + tc * 4; double* pq = sq + tc * 4;
    atomicAdd(ps, s.x); atomicAdd(ps + 1, s.y); atomicAdd(ps + 2, s.z); atomicAdd(ps + 3, s.w);
    atomicAdd(pq, q.x); atomicAdd(pq + 1, q.y); atomicAdd(pq + 2, q.z); atomicAdd(pq + 3, q.w);
  }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// dot-product attention step of the compact decoder (student_model_compact.py:114-138), one workgroup per image:
//   scores_j = <hp[b], f[b][j]> ; w = softmax_j ; ctx = sum_j w_j f[b][j] ; x = emb[b] + ctx
__global__ void dot_attn_fwd_kernel(const float* __restrict__ hp, const float* __restrict__ feats, const float* __restrict__ emb,
                                    float* __restrict__ w_out, float* __restrict__ x_out, int L, int E) {
  extern __shared__ __attribute__((aligned(16))) float sh[];          // [L] scores
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
  const float* F = feats + (long)b * L * E;
  const float* h = hp + (long)b * E;
  for (int j = wave; j < L; j += nw) {
    float s = 0.f;
    for (int e = lane; e < E; e += 64) s = fmaf(h[e], F[(long)j * E + e], s);
    s = wave_sum(s);
    if (lane == 0) sh[j] = s;
  }
  __syncthreads();
  if (wave == 0) {
    float mx = -INFINITY;
    for (int j = lane; j < L; j += 64) mx = fmaxf(mx, sh[j]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    float sum = 0.f;
    for (int j = lane; j < L; j += 64) sum += expf(sh[j] - mx);
    sum = wave_sum(sum);
    const float inv = 1.f / sum;
    for (int j = lane; j < L; j += 64) {
      const float wv = expf(sh[j] - mx) * inv;
      sh[j] = wv;
      w_out[(long)b * L + j] = wv;
    }
  }
  __syncthreads();
  for (int e = tid; e < E; e += blockDim.x) {
    float a = 0.f;
    for (int j = 0; j < L; ++j) a = fmaf(sh[j], F[(long)j * E + e], a);
    x_out[(long)b * E + e] = a + emb[(long)b * E + e];
  }
}

// adjoint: dctx = dx ; dw_j = <dctx, f_j> ; ds = w (dw - sum w dw) ; dfeats[j] += w_j dctx + ds_j hp ; dhp = sum_j ds_j f_j
__global__ void dot_attn_bwd_kernel(const float* __restrict__ dxv, const float* __restrict__ w, const float* __restrict__ hp,
                                    const float* __restrict__ feats, float* __restrict__ dfeats, float* __restrict__ dhp, int L, int E) {
  extern __shared__ __attribute__((aligned(16))) float sh[];          // [L] ds | [L] w
  float* ds = sh;
  float* wl = sh + L;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
  const long base = (long)b * L * E;
  const float* dc = dxv + (long)b * E;
  for (int j = tid; j < L; j += blockDim.x) wl[j] = w[(long)b * L + j];
  for (int j = wave; j < L; j += nw) {
    float s = 0.f;
    for (int e = lane; e < E; e += 64) s = fmaf(dc[e], feats[base + (long)j * E + e], s);
    s = wave_sum(s);
    if (lane == 0) ds[j] = s;
  }
  __syncthreads();
  if (wave == 0) {
    float d = 0.f;
    for (int j = lane; j < L; j += 64) d = fmaf(wl[j], ds[j], d);
    d = wave_sum(d);
    for (int j = lane; j < L; j += 64) ds[j] = wl[j] * (ds[j] - d);
  }
  __syncthreads();
  for (int e = tid; e < E; e += blockDim.x) {
    const float dce = dc[e], he = hp[(long)b * E + e];
    float a = 0.f;
    for (int j = 0; j < L; ++j) {
      const long o = base + (long)j * E + e;
      dfeats[o] += wl[j] * dce + ds[j] * he;
      a = fmaf(ds[j], feats[o], a);
    }
    dhp[(long)b * E + e] = a;
  }
}

inline int ny_for(int C4) { return C4 >= NT ? 1 : NT / C4; }

}  // namespace

#define ST static_cast<hipStream_t>(stream)

extern "C" {

int ick_dwconv3x3_fwd(const float* x, const float* w, float* y, int B, int H, int W, int C, int stride, void* stream) {
  ICK_REQUIRE(x && w && y && B > 0 && H > 0 && W > 0 && C % 4 == 0 && (stride == 1 || stride == 2), "ick_dwconv3x3_fwd: bad arguments");
  const int Ho = (H + 2 - 3) / stride + 1, Wo = (W + 2 - 3) / stride + 1;
  ICK_LAUNCH(dwconv3x3_fwd_kernel, dim3(grid_for((long)B * Ho * Wo * (C / 4))), dim3(NT), 0, ST, x, w, y, B, H, W, C / 4, Ho, Wo, stride);
  return ick::launch_status("dwconv3x3_fwd");
}

int ick_dwconv3x3_dgrad(const float* dy, const float* w, float* dx, int B, int H, int W, int C, int stride, void* stream) {
  ICK_REQUIRE(dy && w && dx && B > 0 && H > 0 && W > 0 && C % 4 == 0 && (stride == 1 || stride == 2), "ick_dwconv3x3_dgrad: bad arguments");
  const int Ho = (H + 2 - 3) / stride + 1, Wo = (W + 2 - 3) / stride + 1;
  ICK_LAUNCH(dwconv3x3_dgrad_kernel, dim3(grid_for((long)B * H * W * (C / 4))), dim3(NT), 0, ST, dy, w, dx, B, H, W, C / 4, Ho, Wo, stride);
  return ick::launch_status("dwconv3x3_dgrad");
}

int ick_dwconv3x3_wgrad(const float* dy, const float* x, float* dw, int B, int H, int W, int C, int stride, void* stream) {
  ICK_REQUIRE(dy && x && dw && B > 0 && H > 0 && W > 0 && C % 4 == 0 && C / 4 <= NT && (stride == 1 || stride == 2),
              "ick_dwconv3x3_wgrad: bad arguments (C <= 1024)");
  const int C4 = C / 4, ny = ny_for(C4);
  const int Ho = (H + 2 - 3) / stride + 1, Wo = (W + 2 - 3) / stride + 1;
  long g = ((long)B * Ho * Wo + ny - 1) / ny;
  if (g > 200) g = 200;
  ICK_LAUNCH(dwconv3x3_wgrad_kernel, dim3((int)g), dim3(ny * C4), (size_t)ny * C4 * 9 * sizeof(float4), ST, dy, x, dw, B, H, W, C4, Ho,
             Wo, stride, ny);
  return ick::launch_status("dwconv3x3_wgrad");
}

int ick_colstats(const float* x, double* sum, double* sq, int64_t M, int C, void* stream) {
  ICK_REQUIRE(x && sum && sq && M > 0 && C % 4 == 0 && C / 4 <= NT, "ick_colstats: bad arguments (C <= 1024)");
  const int C4 = C / 4, ny = ny_for(C4);
  long g = (M + (long)ny * 8 - 1) / ((long)ny * 8);
  if (g > 200) g = 200;
  if (g < 1) g = 1;
  ICK_LAUNCH(colstats_kernel, dim3((int)g), dim3(ny * C4), (size_t)2 * ny * C4 * sizeof(d4), ST, x, sum, sq, (long)M, C4, ny);
  return ick::launch_status("colstats");
}

int ick_dot_attn_fwd(const float* hp, const float* feats, const float* emb, float* w_out, float* x_out, int B, int L, int E, void* stream) {
  ICK_REQUIRE(hp && feats && emb && w_out && x_out && B > 0 && L > 0 && E > 0, "ick_dot_attn_fwd: bad arguments");
  ICK_LAUNCH(dot_attn_fwd_kernel, dim3(B), dim3(NT), (size_t)L * sizeof(float), ST, hp, feats, emb, w_out, x_out, L, E);
  return ick::launch_status("dot_attn_fwd");
}

int ick_dot_attn_bwd(const float* dx, const float* w, const float* hp, const float* feats, float* dfeats, float* dhp, int B, int L, int E,
                     void* stream) {
  ICK_REQUIRE(dx && w && hp && feats && dfeats && dhp && B > 0 && L > 0 && E > 0, "ick_dot_attn_bwd: bad arguments");
  ICK_LAUNCH(dot_attn_bwd_kernel, dim3(B), dim3(NT), (size_t)2 * L * sizeof(float), ST, dx, w, hp, feats, dfeats, dhp, L, E);
  return ick::launch_status("dot_attn_bwd");
}

}  // extern "C"
