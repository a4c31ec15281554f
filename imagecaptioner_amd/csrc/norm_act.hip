// norm_act.hip — the HBM-bound companions of the contraction kernels: BatchNorm (train/eval,
// fwd/bwd), LayerNorm fwd/bwd, row softmax fwd/bwd, max-pool, layout transforms, column sums,
// embedding gather/scatter, adaptive token pooling.  All fp32, float4-vectorised along the
// contiguous (channel / feature) axis, grid-stride loops capped at ~2048 workgroups, wave64
// shuffles for reductions.
#include "ick_common.h"
#include <cstdlib>

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

inline int grid_for(long work_items, int per_block = NT, int cap = 2048) {
  long g = (work_items + per_block - 1) / per_block;
  if (g < 1) g = 1;
  if (g > cap) g = cap;
  return (int)g;
}

// Storage type of the trunk's activations: float, or __bf16 / _Float16 in the 16-bit training regime (the reference's
// autocast keeps them in fp16, train_student_kd.py:271).  ld4 / st4 move four consecutive elements (16 or 8 bytes), index in
// units of four; arithmetic is always fp32 (statistics fp64).
typedef float f32x4_t __attribute__((ext_vector_type(4)));
template <typename T> struct Vec4 { typedef T type __attribute__((ext_vector_type(4))); };
template <typename T>
__device__ __forceinline__ float4 ld4(const T* p, long i) {
  if constexpr (sizeof(T) == 4) return reinterpret_cast<const float4*>(p)[i];
  else {
    const f32x4_t f = __builtin_convertvector(reinterpret_cast<const typename Vec4<T>::type*>(p)[i], f32x4_t);
    return make_float4(f.x, f.y, f.z, f.w);
  }
}
template <typename T>
__device__ __forceinline__ void st4(T* p, long i, float4 v) {
  if constexpr (sizeof(T) == 4) reinterpret_cast<float4*>(p)[i] = v;
  else {
    const f32x4_t f = {v.x, v.y, v.z, v.w};
    reinterpret_cast<typename Vec4<T>::type*>(p)[i] = __builtin_convertvector(f, typename Vec4<T>::type);
  }
}

// ------------------------------------------------------------------ layout transforms
// images (B,3,H,W) fp32 NCHW -> (B,H,W,4) NHWC with a zero 4th channel (16-B pixels for the stem conv)
__global__ void nchw3_to_nhwc4_kernel(const float* __restrict__ x, float* __restrict__ y, long npix, long hw) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < npix; i += (long)gridDim.x * blockDim.x) {
    const long b = i / hw, r = i - b * hw;
    const float* s = x + b * 3 * hw + r;
    reinterpret_cast<float4*>(y)[i] = make_float4(s[0], s[hw], s[2 * hw], 0.f);
  }
}

// dst3[p][0..2] += src4[p][0..2] (the stem's weight gradient leaves its zero-padded 4-channel layout)
__global__ void nhwc4_to_nhwc3_add_kernel(const float* __restrict__ s4, float* __restrict__ d3, long npix) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < npix; i += (long)gridDim.x * blockDim.x) {
    const float4 v = reinterpret_cast<const float4*>(s4)[i];
    d3[3 * i] += v.x; d3[3 * i + 1] += v.y; d3[3 * i + 2] += v.z;
  }
}

// images (B,3,224,224) -> rows [B*196][768], k = c*256 + py*16 + px (the flattened Conv2d(3,384,16,16) weight order)
__global__ void patchify16_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int HW, int G) {
  const long total4 = (long)B * G * G * 192;  // float4 units
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
    const int q = (int)(i % 192);
    const long pr = i / 192;
    const int gx = (int)(pr % G), gy = (int)((pr / G) % G);
    const long b = pr / ((long)G * G);
    const int c = q >> 6, py = (q >> 2) & 15, px = (q & 3) * 4;
    const float* s = x + ((b * 3 + c) * HW + gy * 16 + py) * (long)HW + gx * 16 + px;
    reinterpret_cast<float4*>(y)[i] = *reinterpret_cast<const float4*>(s);
  }
}

// conv weight [Cout][R][S][Cin] -> the layout its stride-1 data gradient reads as a FORWARD convolution over dY:
// wt[ci][R-1-r][S-1-s][co] = w[co][r][s][ci]  (taps flipped, channels swapped).  One 32x32 (co x ci) tile per block
// and tap, transposed through LDS so that both the read (along ci) and the write (along co) are 128-byte rows.
template <typename T>
__global__ void conv_weight_dgrad_layout_kernel(const T* __restrict__ w, T* __restrict__ wt, int Cout, int taps, int Cin) {
  __shared__ T tile[32][33];
  const int tap = blockIdx.z, ci0 = blockIdx.x * 32, co0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 256 threads: 8 rows per pass
#pragma unroll
  for (int r = ty; r < 32; r += 8) {
    const int co = co0 + r, ci = ci0 + tx;
    tile[r][tx] = (co < Cout && ci < Cin) ? w[((long)co * taps + tap) * Cin + ci] : T(0);
  }
  __syncthreads();
  const int tflip = taps - 1 - tap;    // (R-1-r)*S + (S-1-s) = R*S-1 - (r*S+s)
#pragma unroll
  for (int r = ty; r < 32; r += 8) {
    const int ci = ci0 + r, co = co0 + tx;
    if (ci < Cin && co < Cout) wt[((long)ci * taps + tflip) * Cout + co] = tile[tx][r];
  }
}

// ViT token assembly: x[b][0] = cls + pos[0]; x[b][1+p] = patch[b][p] + pos[1+p]
__global__ void vit_assemble_kernel(const float* __restrict__ patch, const float* __restrict__ cls,
                                    const float* __restrict__ pos, float* __restrict__ x, int B, int Ntok, int D4) {
  const long total = (long)B * Ntok * D4;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int d = (int)(i % D4);
    const long r = i / D4;
    const int tkn = (int)(r % Ntok);
    const long b = r / Ntok;
    const float4 pe = reinterpret_cast<const float4*>(pos)[(long)tkn * D4 + d];
    float4 v = tkn == 0 ? reinterpret_cast<const float4*>(cls)[d]
                        : reinterpret_cast<const float4*>(patch)[(b * (Ntok - 1) + tkn - 1) * D4 + d];
    v.x += pe.x; v.y += pe.y; v.z += pe.z; v.w += pe.w;
    reinterpret_cast<float4*>(x)[i] = v;
  }
}

// ------------------------------------------------------------------ BatchNorm
// finalize batch statistics gathered by the conv epilogue: mean/var -> (scale, shift), saved mean / invstd,
// running-stat update with the unbiased variance (nn.BatchNorm2d train mode, momentum 0.1)
// the conv epilogue may spread its fp64 atomics over `copies` accumulator rows (IckGemm.stat_copies): fold them
__device__ __forceinline__ double fold_copies(const double* __restrict__ a, int c, int copies, long stride) {
  double s = a[c];
  for (int r = 1; r < copies; ++r) s += a[r * stride + c];
  return s;
}

__global__ void bn_finalize_kernel(const double* __restrict__ sum, const double* __restrict__ sq, int copies, long stride, float count,
                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                   float* __restrict__ rmean, float* __restrict__ rvar, float momentum, float eps,
                                   float* __restrict__ scale, float* __restrict__ shift, float* __restrict__ smean,
                                   float* __restrict__ sinv, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const double meand = fold_copies(sum, c, copies, stride) / (double)count;
  double vard = fold_copies(sq, c, copies, stride) / (double)count - meand * meand;      // fp64 accumulators: no cancellation
  vard = vard > 0.0 ? vard : 0.0;
  const float mean = (float)meand;
  const float var = (float)vard;
  const float inv = (float)(1.0 / sqrt(vard + (double)eps));
  const float g = gamma[c];
  scale[c] = g * inv;
  shift[c] = beta[c] - mean * g * inv;
  smean[c] = mean;
  sinv[c] = inv;
  if (rmean) {
    const float unb = count > 1.f ? var * count / (count - 1.f) : var;
    rmean[c] = (1.f - momentum) * rmean[c] + momentum * mean;
    rvar[c] = (1.f - momentum) * rvar[c] + momentum * unb;
  }
}

__global__ void bn_eval_coeffs_kernel(const float* __restrict__ gamma, const float* __restrict__ beta,
                                      const float* __restrict__ rmean, const float* __restrict__ rvar, float eps,
                                      float* __restrict__ scale, float* __restrict__ shift, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float inv = rsqrtf(rvar[c] + eps);
  scale[c] = gamma[c] * inv;
  shift[c] = beta[c] - rmean[c] * gamma[c] * inv;
}

// y = [relu](x*scale[c] + shift[c] [+ residual]); x,y [M][C] (NHWC rows)
__global__ void scale_shift_act_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                       const float* __restrict__ shift, const float* __restrict__ res,
                                       float* __restrict__ y, long total4, int C4, int relu) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4);
    const float4 v = reinterpret_cast<const float4*>(x)[i];
    const float4 a = reinterpret_cast<const float4*>(scale)[c];
    const float4 b = reinterpret_cast<const float4*>(shift)[c];
    float4 o = make_float4(fmaf(v.x, a.x, b.x), fmaf(v.y, a.y, b.y), fmaf(v.z, a.z, b.z), fmaf(v.w, a.w, b.w));
    if (res) {
      const float4 r = reinterpret_cast<const float4*>(res)[i];
      o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
    }
    if (relu) {      // 1 = ReLU, 2 = ReLU6 (MobileNetV2)
      const float hi = relu == 2 ? 6.f : INFINITY;
      o.x = fminf(fmaxf(o.x, 0.f), hi); o.y = fminf(fmaxf(o.y, 0.f), hi); o.z = fminf(fmaxf(o.z, 0.f), hi); o.w = fminf(fmaxf(o.w, 0.f), hi);
    }
    reinterpret_cast<float4*>(y)[i] = o;
  }
}

template <typename T>
__global__ void bn_train_apply_kernel(const T* __restrict__ x, const double* __restrict__ sum,
                                      const double* __restrict__ sq, int copies, long stride, float count, const float* __restrict__ gamma,
                                      const float* __restrict__ beta, float* __restrict__ rmean, float* __restrict__ rvar,
                                      float momentum, float eps, const T* __restrict__ res, T* __restrict__ y,
                                      float* __restrict__ smean, float* __restrict__ sinv, long total4, int C4, int relu) {
  const long i0 = blockIdx.x * (long)blockDim.x + threadIdx.x;
  const int c4 = (int)(i0 % C4);
  float sc[4], sh[4];
  if (copies > 1) {
    // large-M layers (few channels, several accumulator copies): fold the copies once per block, cooperatively, into
    // LDS — per THREAD the 2 x copies fp64 loads per channel would dwarf the elementwise work
    __shared__ float s_sc[1024], s_sh[1024];
    const int C = C4 * 4;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
      const double meand = fold_copies(sum, c, copies, stride) / (double)count;
      double vard = fold_copies(sq, c, copies, stride) / (double)count - meand * meand;
      vard = vard > 0.0 ? vard : 0.0;
      const float inv = (float)(1.0 / sqrt(vard + (double)eps));
      const float g = gamma[c];
      s_sc[c] = g * inv;
      s_sh[c] = beta[c] - (float)meand * g * inv;
      if (blockIdx.x == 0) {    // exactly one block publishes the statistics
        smean[c] = (float)meand;
        sinv[c] = inv;
        if (rmean) {
          const float var = (float)vard;
          const float unb = count > 1.f ? var * count / (count - 1.f) : var;
          rmean[c] = (1.f - momentum) * rmean[c] + momentum * (float)meand;
          rvar[c] = (1.f - momentum) * rvar[c] + momentum * unb;
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) { sc[k] = s_sc[c4 * 4 + k]; sh[k] = s_sh[c4 * 4 + k]; }
  } else {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int c = c4 * 4 + k;
      const double meand = sum[c] / (double)count;
      double vard = sq[c] / (double)count - meand * meand;
      vard = vard > 0.0 ? vard : 0.0;
      const float inv = (float)(1.0 / sqrt(vard + (double)eps));
      const float g = gamma[c];
      sc[k] = g * inv;
      sh[k] = beta[c] - (float)meand * g * inv;
      if (i0 < C4) {            // exactly one thread per channel group publishes the statistics
        smean[c] = (float)meand;
        sinv[c] = inv;
        if (rmean) {
          const float var = (float)vard;
          const float unb = count > 1.f ? var * count / (count - 1.f) : var;
          rmean[c] = (1.f - momentum) * rmean[c] + momentum * (float)meand;
          rvar[c] = (1.f - momentum) * rvar[c] + momentum * unb;
        }
      }
    }
  }
  const long gstride = (long)gridDim.x * blockDim.x;     // a multiple of C4: a thread's channels never change
  auto one = [&](float4 v, float4 r) {
    float4 o = make_float4(fmaf(v.x, sc[0], sh[0]), fmaf(v.y, sc[1], sh[1]), fmaf(v.z, sc[2], sh[2]), fmaf(v.w, sc[3], sh[3]));
    o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
    if (relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
    return o;
  };
  const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
  long i = i0;
  if constexpr (sizeof(T) == 2) {       // 8-byte accesses: two groups per trip keep the bytes in flight of the fp32 form
    for (; i + gstride < total4; i += 2 * gstride) {
      const float4 v0 = ld4(x, i), v1 = ld4(x, i + gstride);
      const float4 r0 = res ? ld4(res, i) : zero, r1 = res ? ld4(res, i + gstride) : zero;
      st4(y, i, one(v0, r0));
      st4(y, i + gstride, one(v1, r1));
    }
  }
  for (; i < total4; i += gstride) st4(y, i, one(ld4(x, i), res ? ld4(res, i) : zero));
}

// BN backward pass 1: per channel sum(g) and sum(g * xhat), g = dy * (y > 0) when y != NULL.
// Block = 256 threads as (ROWS x C4 lanes); each thread owns 4 channels, strides over rows; LDS combine, atomics out.
// Arithmetic: the reference's CPU batch_norm backward reduces in double (at::acc_type<float, /*is_cuda=*/false>):
// sum += dy, dotp += (x - mean) * dy with every term converted first.  Same here — products and sums in fp64 per
// thread, fp64 LDS combine, fp64 atomics; invstd is applied once to the finished dot product.  (Both sums are
// differences of large terms behind a mean-subtracting layer; the kernel is HBM-bound, the fp64 VALU work is free.)
struct d4 { double x, y, z, w; };
// G float4 groups per thread = one 16-byte load of every tensor: G = 1 for fp32 storage, 2 (8 channels) for 16-bit storage —
// with 8-byte loads the 16-bit form was latency-bound at 1.9 TB/s (42 us against the fp32 form's 28 on 12544 x 1024).
template <typename T, int G>
__device__ __forceinline__ void ldg16(const T* p, long o, float4 (&out)[G]) {     // o in units of 4 * G elements
  if constexpr (G == 1) out[0] = ld4(p, o);
  else {
    typedef T h8 __attribute__((ext_vector_type(8)));
    typedef float f8 __attribute__((ext_vector_type(8)));
    const f8 f = __builtin_convertvector(reinterpret_cast<const h8*>(p)[o], f8);
    out[0] = make_float4(f[0], f[1], f[2], f[3]);
    out[1] = make_float4(f[4], f[5], f[6], f[7]);
  }
}
__device__ __forceinline__ void bn_red_acc(d4& dg, d4& dx, float4 g, const float4 yy, bool masked, float hi, const float4 xv, const d4& mu) {
  if (masked) {
    g.x = (yy.x > 0.f && yy.x < hi) ? g.x : 0.f; g.y = (yy.y > 0.f && yy.y < hi) ? g.y : 0.f;
    g.z = (yy.z > 0.f && yy.z < hi) ? g.z : 0.f; g.w = (yy.w > 0.f && yy.w < hi) ? g.w : 0.f;
  }
  dg.x += g.x; dg.y += g.y; dg.z += g.z; dg.w += g.w;
  dx.x = fma((double)g.x, (double)xv.x - mu.x, dx.x); dx.y = fma((double)g.y, (double)xv.y - mu.y, dx.y);
  dx.z = fma((double)g.z, (double)xv.z - mu.z, dx.z); dx.w = fma((double)g.w, (double)xv.w - mu.w, dx.w);
}
template <typename T, int U>
__global__ __launch_bounds__(NT) void bn_bwd_reduce_kernel(const T* __restrict__ dy, const T* __restrict__ y,
                                     const T* __restrict__ x, const float* __restrict__ mean,
                                     const float* __restrict__ inv, double* __restrict__ sum_g,
                                     double* __restrict__ sum_gx, int copies, long stride, long M, int C, float hi) {
  constexpr int G = sizeof(T) == 2 ? 2 : 1;
  const int CG = C / (4 * G);              // 16-byte channel groups per row (launcher: C % (4 G) == 0)
  const int lanes = CG < NT ? CG : NT;     // threads along channels
  const int rows = NT / lanes;             // threads along rows (threads beyond rows * lanes idle: CG need not divide 256)
  const int tc = threadIdx.x % lanes, tr = threadIdx.x / lanes;
  __shared__ d4 sh[2][G][NT];               // partial sums of the row-threads; reused as the block's final [2][lanes * 4 G] doubles
  auto& sh_g = sh[0]; auto& sh_x = sh[1];
  for (int cbase = blockIdx.x * lanes; cbase < CG; cbase += gridDim.x * lanes) {     // block-uniform trip count (barriers inside)
    const int cg = cbase + tc;
    const bool cok = cg < CG;
    d4 mu[G], dg[G], dx[G];
#pragma unroll
    for (int k = 0; k < G; ++k) {
      const float4 muf = cok ? reinterpret_cast<const float4*>(mean)[cg * G + k] : make_float4(0, 0, 0, 0);
      mu[k] = d4{muf.x, muf.y, muf.z, muf.w};
      dg[k] = d4{0, 0, 0, 0}; dx[k] = d4{0, 0, 0, 0};
    }
    if (tr < rows && cok) {
      // U rows per trip: 3 U independent 16-byte loads in flight per thread (one row per trip left the kernel
      // latency-bound at ~3.6 TB/s with 1.5 workgroups per CU)
      const long step = (long)gridDim.y * rows;
      long m = blockIdx.y * (long)rows + tr;
      for (; m + (U - 1) * step < M; m += U * step) {
        float4 g[U][G], xv[U][G], yy[U][G];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const long o = (m + u * step) * CG + cg;
          ldg16<T, G>(dy, o, g[u]);
          ldg16<T, G>(x, o, xv[u]);
          if (y) ldg16<T, G>(y, o, yy[u]);
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
          for (int k = 0; k < G; ++k) bn_red_acc(dg[k], dx[k], g[u][k], yy[u][k], y != nullptr, hi, xv[u][k], mu[k]);
      }
      for (; m < M; m += step) {
        const long o = m * CG + cg;
        float4 g[G], xv[G], yy[G];
        ldg16<T, G>(dy, o, g);
        ldg16<T, G>(x, o, xv);
        if (y) ldg16<T, G>(y, o, yy);
#pragma unroll
        for (int k = 0; k < G; ++k) bn_red_acc(dg[k], dx[k], g[k], yy[k], y != nullptr, hi, xv[k], mu[k]);
      }
    }
#pragma unroll
    for (int k = 0; k < G; ++k) { sh_g[k][threadIdx.x] = dg[k]; sh_x[k][threadIdx.x] = dx[k]; }
    __syncthreads();
    // combine the row-threads in LDS, then add the block's 2 x (lanes * 4 G) channel sums with CONSECUTIVE channels on
    // consecutive lanes: one wave-instruction = 64 doubles = 4 cache lines.  (Adding straight from the owning threads
    // strides the lanes by 32 / 64 bytes: 16 / 32 lines per instruction, and the L2 prices atomics per line touched —
    // measured on 3136 x 2048: 144 / 76 / 44 / 33 us with 800 / 400 / 200 / 100 row-blocks, i.e. the tail was the kernel.)
    double* fin = reinterpret_cast<double*>(&sh[0][0][0]);         // [2][lanes * 4 G] <= 2 * G * NT * 4 doubles, once the partials are consumed
    d4 tg[G], tx[G];
    if (tr == 0 && cok) {
#pragma unroll
      for (int k = 0; k < G; ++k) {
        tg[k] = dg[k]; tx[k] = dx[k];
        for (int r = 1; r < rows; ++r) {
          const d4 a = sh_g[k][r * lanes + tc], b = sh_x[k][r * lanes + tc];
          tg[k].x += a.x; tg[k].y += a.y; tg[k].z += a.z; tg[k].w += a.w;
          tx[k].x += b.x; tx[k].y += b.y; tx[k].z += b.z; tx[k].w += b.w;
        }
        const float4 iv = reinterpret_cast<const float4*>(inv)[cg * G + k];
        tx[k].x *= iv.x; tx[k].y *= iv.y; tx[k].z *= iv.z; tx[k].w *= iv.w;
      }
    }
    __syncthreads();                                               // every partial has been read
    const int nch = lanes * 4 * G;                                 // channels of this block's column range
    if (tr == 0) {
#pragma unroll
      for (int k = 0; k < G; ++k) {
        double* fg = fin + (tc * G + k) * 4; double* fx = fin + nch + (tc * G + k) * 4;
        const bool ok = cok;
        fg[0] = ok ? tg[k].x : 0.0; fg[1] = ok ? tg[k].y : 0.0; fg[2] = ok ? tg[k].z : 0.0; fg[3] = ok ? tg[k].w : 0.0;
        fx[0] = ok ? tx[k].x : 0.0; fx[1] = ok ? tx[k].y : 0.0; fx[2] = ok ? tx[k].z : 0.0; fx[3] = ok ? tx[k].w : 0.0;
      }
    }
    __syncthreads();
    {
      // up to 1024 row-blocks add into the same 2 x C addresses: spread them over `copies` accumulator rows (the apply
      // pass folds them) — same-address fp64 atomics serialise in L2
      const long co = (long)(blockIdx.y % copies) * stride;
      const int c0 = cbase * 4 * G;
      for (int i = threadIdx.x; i < 2 * nch; i += NT) {
        const int which = i >= nch, cl = i - which * nch;
        if (c0 + cl < C) atomicAdd((which ? sum_gx : sum_g) + co + c0 + cl, fin[i]);
      }
    }
    __syncthreads();
  }
}

// (Round 3 tried folding the copies inside the apply kernel instead — every thread folding its own four channels — to save
//  this launch: 29 launches x 5.7 us gone, but the apply kernel went from 17 to 34 us on average, 2 x 4 x copies dependent fp64
//  loads in the prologue of every thread of a 2048-block grid: +0.33 ms per step net, reverted.)
// BN backward pass 2a: fold the accumulator copies once — coef = {sum_g / M, sum_gx / M} as floats for the apply pass, and the
// parameter gradients, which are the two reductions themselves: dgamma += sum(g * xhat), dbeta += sum(g).
__global__ void bn_bwd_fold_kernel(const double* __restrict__ sum_g, const double* __restrict__ sum_gx, int copies, long stride,
                                   double invM, float* __restrict__ coef, float* __restrict__ dgamma, float* __restrict__ dbeta, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const double sg = fold_copies(sum_g, c, copies, stride), sx = fold_copies(sum_gx, c, copies, stride);
  coef[c] = (float)(sg * invM);
  coef[C + c] = (float)(sx * invM);
  if (dgamma) { dgamma[c] += (float)sx; dbeta[c] += (float)sg; }
}

// BN backward pass 2b: dx = gamma*inv*(g - mean_g - xhat*mean_gx); optionally also writes g (the gradient that
// flows on into the residual branch).  In eval mode (use_batch_stats == 0): dx = gamma*inv*g.
template <typename T>
__global__ void bn_bwd_apply_kernel(const T* __restrict__ dy, const T* __restrict__ y,
                                    const T* __restrict__ x, const float* __restrict__ mean,
                                    const float* __restrict__ inv, const float* __restrict__ gamma,
                                    const float* __restrict__ coef, T* __restrict__ dx, T* __restrict__ gout,
                                    long total4, int C4, int use_batch_stats, int hoist, float hi) {
  // hoist: the grid stride is a multiple of C4 (launcher), so a thread's 4 channels never change: per-channel terms once.
  // Otherwise (C/4 neither divides nor is a multiple of 256: MobileNetV2's 96, 144, 576 ...) they are re-read per element.
  const long i0 = blockIdx.x * (long)blockDim.x + threadIdx.x;
  int c = (int)(i0 % C4);
  float4 ga = reinterpret_cast<const float4*>(gamma)[c];
  float4 iv = reinterpret_cast<const float4*>(inv)[c];
  float4 mu = make_float4(0, 0, 0, 0), mg = mu, mx = mu;
  if (use_batch_stats) {
    mu = reinterpret_cast<const float4*>(mean)[c];
    mg = reinterpret_cast<const float4*>(coef)[c];
    mx = reinterpret_cast<const float4*>(coef)[C4 + c];
  }
  for (long i = i0; i < total4; i += (long)gridDim.x * blockDim.x) {
    if (!hoist) {
      c = (int)(i % C4);
      ga = reinterpret_cast<const float4*>(gamma)[c];
      iv = reinterpret_cast<const float4*>(inv)[c];
      if (use_batch_stats) {
        mu = reinterpret_cast<const float4*>(mean)[c];
        mg = reinterpret_cast<const float4*>(coef)[c];
        mx = reinterpret_cast<const float4*>(coef)[C4 + c];
      }
    }
    float4 g = ld4(dy, i);
    if (y) {
      const float4 yy = ld4(y, i);
      g.x = (yy.x > 0.f && yy.x < hi) ? g.x : 0.f; g.y = (yy.y > 0.f && yy.y < hi) ? g.y : 0.f;
      g.z = (yy.z > 0.f && yy.z < hi) ? g.z : 0.f; g.w = (yy.w > 0.f && yy.w < hi) ? g.w : 0.f;
    }
    if (gout) st4(gout, i, g);
    float4 o;
    if (use_batch_stats) {
      const float4 xv = ld4(x, i);
      o.x = ga.x * iv.x * (g.x - mg.x - (xv.x - mu.x) * iv.x * mx.x);
      o.y = ga.y * iv.y * (g.y - mg.y - (xv.y - mu.y) * iv.y * mx.y);
      o.z = ga.z * iv.z * (g.z - mg.z - (xv.z - mu.z) * iv.z * mx.z);
      o.w = ga.w * iv.w * (g.w - mg.w - (xv.w - mu.w) * iv.w * mx.w);
    } else {
      o = make_float4(ga.x * iv.x * g.x, ga.y * iv.y * g.y, ga.z * iv.z * g.z, ga.w * iv.w * g.w);
    }
    st4(dx, i, o);
  }
}

// ------------------------------------------------------------------ max-pool 3x3 / stride 2 / pad 1, NHWC
template <typename T>
__global__ void maxpool3x3s2_kernel(const T* __restrict__ x, T* __restrict__ y, int B, int H, int W, int C4,
                                    int Ho, int Wo) {
  const long total = (long)B * Ho * Wo * C4;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4);
    long r = i / C4;
    const int ox = (int)(r % Wo); r /= Wo;
    const int oy = (int)(r % Ho);
    const long b = r / Ho;
    float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      const int iy = oy * 2 - 1 + dy;
      if ((unsigned)iy >= (unsigned)H) continue;
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const int ix = ox * 2 - 1 + dx;
        if ((unsigned)ix >= (unsigned)W) continue;
        const float4 v = ld4(x, ((b * H + iy) * W + ix) * C4 + c);
        m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
      }
    }
    st4(y, i, m);
  }
}

// backward of the 3x3 / stride 2 / pad 1 max-pool: dx[p] = sum over the (<= 4) windows containing p of dy[window] when p is
// that window's FIRST maximum (torch's tie rule: the first element of the window scan, row-major).  Gather form: no atomics.
__global__ void maxpool3x3s2_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx,
                                        int B, int H, int W, int C, int Ho, int Wo) {
  const long total = (long)B * H * W * C;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    long r = i / C;
    const int ix = (int)(r % W); r /= W;
    const int iy = (int)(r % H);
    const long b = r / H;
    const float v = x[i];
    float g = 0.f;
    // windows (oy, ox) with 2*oy-1 <= iy <= 2*oy+1
    for (int oy = (iy >> 1); oy <= ((iy + 1) >> 1); ++oy) {
      if (oy >= Ho) continue;
      for (int ox = (ix >> 1); ox <= ((ix + 1) >> 1); ++ox) {
        if (ox >= Wo) continue;
        // is (iy, ix) the first maximum of this window?
        bool first = true;
        for (int dyy = 0; dyy < 3 && first; ++dyy) {
          const int yy = oy * 2 - 1 + dyy;
          if ((unsigned)yy >= (unsigned)H) continue;
          for (int dxx = 0; dxx < 3; ++dxx) {
            const int xx = ox * 2 - 1 + dxx;
            if ((unsigned)xx >= (unsigned)W) continue;
            const float u = x[((b * H + yy) * W + xx) * C + c];
            const bool before = yy < iy || (yy == iy && xx < ix);
            if (u > v || (before && u == v)) { first = false; break; }
          }
        }
        if (first) g += dy[((b * Ho + oy) * Wo + ox) * C + c];
      }
    }
    dx[i] = g;
  }
}

// nn.AdaptiveAvgPool2d((Ho, Wo)) on NHWC: bin i = [floor(i*H/Ho), ceil((i+1)*H/Ho))
__global__ void adaptive_avgpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int H, int W, int C4,
                                            int Ho, int Wo) {
  const long total = (long)B * Ho * Wo * C4;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4);
    long r = i / C4;
    const int ox = (int)(r % Wo); r /= Wo;
    const int oy = (int)(r % Ho);
    const long b = r / Ho;
    const int y0 = (oy * H) / Ho, y1 = ((oy + 1) * H + Ho - 1) / Ho, x0 = (ox * W) / Wo, x1 = ((ox + 1) * W + Wo - 1) / Wo;
    float4 s = make_float4(0, 0, 0, 0);
    for (int yy = y0; yy < y1; ++yy)
      for (int xx = x0; xx < x1; ++xx) {
        const float4 v = reinterpret_cast<const float4*>(x)[((b * H + yy) * W + xx) * C4 + c];
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
      }
    const float inv = 1.f / ((y1 - y0) * (x1 - x0));
    reinterpret_cast<float4*>(y)[i] = make_float4(s.x * inv, s.y * inv, s.z * inv, s.w * inv);
  }
}
__global__ void adaptive_avgpool_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int B, int H, int W, int C4,
                                            int Ho, int Wo) {
  const long total = (long)B * H * W * C4;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4);
    long r = i / C4;
    const int ix = (int)(r % W); r /= W;
    const int iy = (int)(r % H);
    const long b = r / H;
    float4 s = make_float4(0, 0, 0, 0);
    int oy_lo = (int)(((long)iy * Ho) / H) - 1; if (oy_lo < 0) oy_lo = 0;
    int ox_lo = (int)(((long)ix * Wo) / W) - 1; if (ox_lo < 0) ox_lo = 0;
    for (int oy = oy_lo; oy < Ho && (oy * H) / Ho <= iy; ++oy) {
      const int y0 = (oy * H) / Ho, y1 = ((oy + 1) * H + Ho - 1) / Ho;
      if (iy < y0 || iy >= y1) continue;
      for (int ox = ox_lo; ox < Wo && (ox * W) / Wo <= ix; ++ox) {
        const int x0 = (ox * W) / Wo, x1 = ((ox + 1) * W + Wo - 1) / Wo;
        if (ix < x0 || ix >= x1) continue;
        const float inv = 1.f / ((y1 - y0) * (x1 - x0));
        const float4 v = reinterpret_cast<const float4*>(dy)[((b * Ho + oy) * Wo + ox) * C4 + c];
        s.x += v.x * inv; s.y += v.y * inv; s.z += v.z * inv; s.w += v.w * inv;
      }
    }
    reinterpret_cast<float4*>(dx)[i] = s;
  }
}

// ------------------------------------------------------------------ LayerNorm (one wave per row)
__global__ void layernorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                     const float* __restrict__ b, float* __restrict__ y, float* __restrict__ mean_out,
                                     float* __restrict__ rstd_out, long rows, int D, float eps) {
  const int lane = threadIdx.x & 63, wpb = blockDim.x >> 6;
  const int D4 = D >> 2;
  if (D4 <= 256) {
    // rows up to 1024 wide (every LayerNorm of the step: 256 / 384 / 512): the row lives in registers — ONE global read and
    // one load latency per row instead of three dependent passes (the kernel is latency-bound: 12608 rows of 1.5 KB took
    // 13.7 us = 2.8 TB/s); the arithmetic (order of the sums) is unchanged
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    for (long r = blockIdx.x * (long)wpb + (threadIdx.x >> 6); r < rows; r += (long)gridDim.x * wpb) {
      const float4* xr = reinterpret_cast<const float4*>(x + r * D);
      float4 v[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = (lane + 64 * k) < D4 ? xr[lane + 64 * k] : z4;
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) if ((lane + 64 * k) < D4) s += (v[k].x + v[k].y) + (v[k].z + v[k].w);
      const float mu = wave_sum(s) / D;
      float q = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) if ((lane + 64 * k) < D4) {
        const float a = v[k].x - mu, c = v[k].y - mu, d = v[k].z - mu, e = v[k].w - mu;
        q += (a * a + c * c) + (d * d + e * e);
      }
      const float rs = rsqrtf(wave_sum(q) / D + eps);
      float4* yr = reinterpret_cast<float4*>(y + r * D);
#pragma unroll
      for (int k = 0; k < 4; ++k) if ((lane + 64 * k) < D4) {
        const int i = lane + 64 * k;
        const float4 gg = reinterpret_cast<const float4*>(g)[i], bb = reinterpret_cast<const float4*>(b)[i];
        yr[i] = make_float4((v[k].x - mu) * rs * gg.x + bb.x, (v[k].y - mu) * rs * gg.y + bb.y,
                            (v[k].z - mu) * rs * gg.z + bb.z, (v[k].w - mu) * rs * gg.w + bb.w);
      }
      if (lane == 0 && mean_out) { mean_out[r] = mu; rstd_out[r] = rs; }
    }
    return;
  }
  for (long r = blockIdx.x * (long)wpb + (threadIdx.x >> 6); r < rows; r += (long)gridDim.x * wpb) {
    const float4* xr = reinterpret_cast<const float4*>(x + r * D);
    float s = 0.f;
    for (int i = lane; i < D4; i += 64) { const float4 v = xr[i]; s += (v.x + v.y) + (v.z + v.w); }
    const float mu = wave_sum(s) / D;
    float q = 0.f;
    for (int i = lane; i < D4; i += 64) {
      const float4 v = xr[i];
      const float a = v.x - mu, c = v.y - mu, d = v.z - mu, e = v.w - mu;
      q += (a * a + c * c) + (d * d + e * e);
    }
    const float rs = rsqrtf(wave_sum(q) / D + eps);
    float4* yr = reinterpret_cast<float4*>(y + r * D);
    for (int i = lane; i < D4; i += 64) {
      const float4 v = xr[i];
      const float4 gg = reinterpret_cast<const float4*>(g)[i], bb = reinterpret_cast<const float4*>(b)[i];
      yr[i] = make_float4((v.x - mu) * rs * gg.x + bb.x, (v.y - mu) * rs * gg.y + bb.y,
                          (v.z - mu) * rs * gg.z + bb.z, (v.w - mu) * rs * gg.w + bb.w);
    }
    if (lane == 0 && mean_out) { mean_out[r] = mu; rstd_out[r] = rs; }
  }
}

// dx per row (one wave per row); dgamma/dbeta: per-block partial sums in LDS then atomics
__global__ void layernorm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                     const float* __restrict__ g, const float* __restrict__ mean,
                                     const float* __restrict__ rstd, float* __restrict__ dx,
                                     float* __restrict__ dgamma, float* __restrict__ dbeta, long rows, int D) {
  extern __shared__ __attribute__((aligned(16))) float sh[];  // [2][D]
  const int lane = threadIdx.x & 63, wpb = blockDim.x >> 6;
  for (int i = threadIdx.x; i < 2 * D; i += blockDim.x) sh[i] = 0.f;
  __syncthreads();
  for (long r = blockIdx.x * (long)wpb + (threadIdx.x >> 6); r < rows; r += (long)gridDim.x * wpb) {
    const float mu = mean[r], rs = rstd[r];
    const float* xr = x + r * D; const float* dr = dy + r * D;
    float s1 = 0.f, s2 = 0.f;
    for (int i = lane; i < D; i += 64) {
      const float xh = (xr[i] - mu) * rs, dg = dr[i] * g[i];
      s1 += dg; s2 += dg * xh;
    }
    s1 = wave_sum(s1) / D; s2 = wave_sum(s2) / D;
    for (int i = lane; i < D; i += 64) {
      const float xh = (xr[i] - mu) * rs, d = dr[i];
      dx[r * D + i] = rs * (d * g[i] - s1 - xh * s2);
      if (dgamma) { atomicAdd(&sh[i], d * xh); atomicAdd(&sh[D + i], d); }
    }
  }
  __syncthreads();
  if (dgamma)
    for (int i = threadIdx.x; i < D; i += blockDim.x) { atomicAdd(dgamma + i, sh[i]); atomicAdd(dbeta + i, sh[D + i]); }
}

// ------------------------------------------------------------------ row softmax (attention scores), in place
// s[row][0..L) <- softmax(scale * s) ; causal: columns > (row % Lq) are masked out
__global__ void softmax_rows_kernel(float* __restrict__ s, long rows, int L, int ld, float scale, int causal, int Lq) {
  const int lane = threadIdx.x & 63, wpb = blockDim.x >> 6;
  for (long r = blockIdx.x * (long)wpb + (threadIdx.x >> 6); r < rows; r += (long)gridDim.x * wpb) {
    float* sr = s + r * ld;
    const int lim = causal ? (int)(r % Lq) + 1 : L;
    float mx = -INFINITY;
    for (int i = lane; i < lim; i += 64) mx = fmaxf(mx, sr[i] * scale);
    mx = wave_max(mx);
    float sum = 0.f;
    for (int i = lane; i < lim; i += 64) sum += expf(sr[i] * scale - mx);
    sum = wave_sum(sum);
    const float inv = 1.f / sum;
    for (int i = lane; i < ld; i += 64) sr[i] = i < lim ? expf(sr[i] * scale - mx) * inv : 0.f;  // pad columns [L, ld) zeroed
  }
}

// dS = scale * P * (dP - sum_j dP_j P_j), in place on dP
__global__ void softmax_bwd_rows_kernel(float* __restrict__ dp, const float* __restrict__ p, long rows, int L, int ld,
                                        float scale) {
  const int lane = threadIdx.x & 63, wpb = blockDim.x >> 6;
  for (long r = blockIdx.x * (long)wpb + (threadIdx.x >> 6); r < rows; r += (long)gridDim.x * wpb) {
    float* dr = dp + r * ld; const float* pr = p + r * ld;
    float d = 0.f;
    for (int i = lane; i < L; i += 64) d += dr[i] * pr[i];
    d = wave_sum(d);
    for (int i = lane; i < ld; i += 64) dr[i] = i < L ? scale * pr[i] * (dr[i] - d) : 0.f;
  }
}

// ------------------------------------------------------------------ small utilities
// out[n] (+)= sum_m x[m][n]   (bias gradients); grid.y strides over rows, atomics combine
__global__ void colsum_kernel(const float* __restrict__ x, float* __restrict__ out, long M, int N, long ld) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  // four independent partial sums: the loads of a thread are back to back in flight instead of one dependent add per round trip
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  const long st = gridDim.y;
  long m = blockIdx.y;
  for (; m + 3 * st < M; m += 4 * st) {
    s0 += x[m * ld + n]; s1 += x[(m + st) * ld + n]; s2 += x[(m + 2 * st) * ld + n]; s3 += x[(m + 3 * st) * ld + n];
  }
  for (; m < M; m += st) s0 += x[m * ld + n];
  atomicAdd(out + n, (s0 + s1) + (s2 + s3));
}

// out[0] = max(out[0], max |x|): non-negative floats order like their bit patterns
__global__ void absmax_kernel(const float* __restrict__ x, long n4, float* __restrict__ out) {
  float m = 0.f;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const float4 v = reinterpret_cast<const float4*>(x)[i];
    m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  // ONE atomic per workgroup (first form: one per wave from ~800 workgroups = 3000 same-address atomics, 41 us per launch)
  __shared__ float wmax[NT / 64];
  if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    float b = wmax[0];
#pragma unroll
    for (int w = 1; w < NT / 64; ++w) b = fmaxf(b, wmax[w]);
    atomicMax(reinterpret_cast<unsigned*>(out), __float_as_uint(b));
  }
}

// dx = dy * (y > 0)
__global__ void relu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, float* __restrict__ dx,
                                long total4) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
    float4 g = reinterpret_cast<const float4*>(dy)[i];
    const float4 yy = reinterpret_cast<const float4*>(y)[i];
    g.x = yy.x > 0.f ? g.x : 0.f; g.y = yy.y > 0.f ? g.y : 0.f; g.z = yy.z > 0.f ? g.z : 0.f; g.w = yy.w > 0.f ? g.w : 0.f;
    reinterpret_cast<float4*>(dx)[i] = g;
  }
}

// y = a + b (float4)
__global__ void add_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ y, long total4) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
    const float4 u = reinterpret_cast<const float4*>(a)[i], v = reinterpret_cast<const float4*>(b)[i];
    reinterpret_cast<float4*>(y)[i] = make_float4(u.x + v.x, u.y + v.y, u.z + v.z, u.w + v.w);
  }
}

// embedding gather: out[i][:] = table[ids[i]][:] (+ pe[i / per_pos][:] when pe != NULL: sinusoid rows by position)
__global__ void embedding_fwd_kernel(const long* __restrict__ ids, const float* __restrict__ table,
                                     const float* __restrict__ pe, float* __restrict__ out, long n, int D4, int per_pos) {
  const long total = n * D4;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long r = i / D4; const int d = (int)(i - r * D4);
    float4 v = reinterpret_cast<const float4*>(table)[ids[r] * D4 + d];
    if (pe) {
      const long pr = per_pos > 0 ? r / per_pos : r % (-per_pos);   // position of row r (seq-first / batch-first)
      const float4 e = reinterpret_cast<const float4*>(pe)[pr * D4 + d];
      v.x += e.x; v.y += e.y; v.z += e.z; v.w += e.w;
    }
    reinterpret_cast<float4*>(out)[i] = v;
  }
}

// embedding backward: dtable[ids[i]][:] += dout[i][:]
__global__ void embedding_bwd_kernel(const long* __restrict__ ids, const float* __restrict__ dout,
                                     float* __restrict__ dtable, long n, int D) {
  const long total = n * D;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long r = i / D; const int d = (int)(i - r * D);
    atomicAdd(dtable + ids[r] * D + d, dout[i]);
  }
}

// adaptive average pooling along tokens: x (B,L,D) -> y (B,Lo,D), bin i = [floor(i*L/Lo), ceil((i+1)*L/Lo))
__global__ void token_pool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int L, int Lo, int D4) {
  const long total = (long)B * Lo * D4;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int d = (int)(i % D4); const long r = i / D4; const int o = (int)(r % Lo); const long b = r / Lo;
    const int a = (o * L) / Lo, e = ((o + 1) * L + Lo - 1) / Lo;
    float4 s = make_float4(0, 0, 0, 0);
    for (int t = a; t < e; ++t) {
      const float4 v = reinterpret_cast<const float4*>(x)[(b * L + t) * D4 + d];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    const float inv = 1.f / (e - a);
    reinterpret_cast<float4*>(y)[i] = make_float4(s.x * inv, s.y * inv, s.z * inv, s.w * inv);
  }
}

// backward: dx[b][t] = sum over bins o containing t of dy[b][o] / |bin o|
__global__ void token_pool_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int B, int L, int Lo, int D4) {
  const long total = (long)B * L * D4;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int d = (int)(i % D4); const long r = i / D4; const int t = (int)(r % L); const long b = r / L;
    float4 s = make_float4(0, 0, 0, 0);
    // candidate bins: o with floor(o*L/Lo) <= t < ceil((o+1)*L/Lo)
    int o_lo = (int)(((long)t * Lo) / L) - 1; if (o_lo < 0) o_lo = 0;
    for (int o = o_lo; o < Lo && (o * L) / Lo <= t; ++o) {
      const int a = (o * L) / Lo, e = ((o + 1) * L + Lo - 1) / Lo;
      if (t >= a && t < e) {
        const float inv = 1.f / (e - a);
        const float4 v = reinterpret_cast<const float4*>(dy)[(b * Lo + o) * D4 + d];
        s.x += v.x * inv; s.y += v.y * inv; s.z += v.z * inv; s.w += v.w * inv;
      }
    }
    reinterpret_cast<float4*>(dx)[i] = s;
  }
}


// counter-based dropout: keep(i) is a pure function of (seed, i), so the backward pass regenerates the mask
// instead of storing it.  y = x * keep / (1-p).  (The reference's nn.Dropout draws from torch's Philox stream;
// masks cannot match bit-for-bit, parity runs use p = 0 — SURVEY.md fact 7.)
__device__ __forceinline__ unsigned mix32(unsigned long long z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return (unsigned)((z ^ (z >> 31)) >> 32);
}
__global__ void dropout_kernel(const float* __restrict__ x, float* __restrict__ y, long n, float p, float inv_keep,
                               unsigned long long seed, const long* __restrict__ step) {
  if (step) seed += (unsigned long long)(*step) * 0x9E3779B97F4A7C15ull;   // device-resident step counter (graph replay)
  const unsigned thr = (unsigned)(p * 4294967296.0);
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    y[i] = mix32(seed * 0x100000001B3ull + (unsigned long long)i) >= thr ? x[i] * inv_keep : 0.f;
}

}  // namespace

#define ST static_cast<hipStream_t>(stream)

// ---- launchers of the storage-type templated kernels (fp32 and 16-bit entry points below)
template <typename T>
int bn_bwd_reduce_impl(const T* dy, const T* y, const T* x, const float* mean, const float* invstd,
                      double* sum_g, double* sum_gx, int copies, int64_t stride, long M, int C, int act, void* stream) {
  ICK_REQUIRE(copies >= 1 && (copies == 1 || stride >= C), "ick_bn_bwd_reduce: copies >= 1, stride >= C");
  ICK_REQUIRE(dy && x && mean && invstd && sum_g && sum_gx && C % 4 == 0 && M > 0, "ick_bn_bwd_reduce: bad arguments");
  constexpr int G = sizeof(T) == 2 ? 2 : 1;              // 16-byte channel groups (see the kernel)
  ICK_REQUIRE(C % (4 * G) == 0, "ick_bn_bwd_reduce: C must be a multiple of %d", 4 * G);
  const int C4 = C / (4 * G);
  const int lanes = C4 < NT ? C4 : NT;
  const int rows = NT / lanes;
  const int gx = (C4 + lanes - 1) / lanes;
  // Grid: ~200 workgroups in all.  Every workgroup ends with 2 x 4 x lanes fp64 atomics onto the same 2 x C addresses and
  // those — not the 12 B/element stream — set the time: measured (tools/bench_bn.py, M = 12544, C = 1024) 123 us with 3136
  // row-blocks, 55 us with 784, 27 us (5.7 TB/s) with 196; fewer than ~100 starves the memory system again.
  static const int target = [] { const char* e = getenv("ICK_BN_BWD_BLOCKS"); return e ? atoi(e) : 200; }();   // A/B runs
  long gy = target / gx;
  const long gmax = (M + rows * 4 - 1) / (rows * 4);          // at least one 4-row trip per thread
  if (gy > gmax) gy = gmax;
  if (gy < 1) gy = 1;
  static const int u8 = [] { const char* e = getenv("ICK_BN_BWD_U"); return e && atoi(e) == 8; }();   // A/B runs
  if (u8) ICK_LAUNCH((bn_bwd_reduce_kernel<T, 8>), dim3(gx, (int)gy), dim3(NT), 0, ST, dy, y, x, mean, invstd, sum_g, sum_gx, copies, (long)stride, M, C,
                     act == 2 ? 6.f : INFINITY);
  else ICK_LAUNCH((bn_bwd_reduce_kernel<T, 4>), dim3(gx, (int)gy), dim3(NT), 0, ST, dy, y, x, mean, invstd, sum_g, sum_gx, copies, (long)stride, M, C,
                  act == 2 ? 6.f : INFINITY);
  return ick::launch_status("bn_bwd_reduce");
}

template <typename T>
int bn_bwd_apply_impl(const T* dy, const T* y, const T* x, const float* mean, const float* invstd,
                     const float* gamma, const double* sum_g, const double* sum_gx, int copies, int64_t stride, float* coef_ws,
                     T* dx, T* g_out, long M, int C, int use_batch_stats, float* dgamma, float* dbeta, int act, void* stream) {
  if (copies < 1) copies = 1;
  ICK_REQUIRE(dy && x && mean && invstd && gamma && dx && C % 4 == 0 && M > 0, "ick_bn_bwd_apply: bad arguments");
  ICK_REQUIRE((dgamma == nullptr) == (dbeta == nullptr), "ick_bn_bwd_apply: dgamma and dbeta go together");
  ICK_REQUIRE(!use_batch_stats || (sum_g && sum_gx && coef_ws), "ick_bn_bwd_apply: batch statistics need the two sums and the 2*C workspace");
  const int C4 = C / 4;
  const int hoist = (C4 <= NT ? NT % C4 == 0 : C4 % NT == 0) ? 1 : 0;
  if (use_batch_stats)
    ICK_LAUNCH(bn_bwd_fold_kernel, dim3((C + NT - 1) / NT), dim3(NT), 0, ST, sum_g, sum_gx, copies, (long)stride, 1.0 / (double)M,
               coef_ws, dgamma, dbeta, C);
  const long total4 = M * C4;
  int grid = grid_for(total4);
  const int q = (hoist && C4 > NT) ? C4 / NT : 1;          // grid * NT must be a multiple of C4: a thread keeps its channels
  grid = (grid + q - 1) / q * q;
  ICK_LAUNCH(bn_bwd_apply_kernel<T>, dim3(grid), dim3(NT), 0, ST, dy, y, x, mean, invstd, gamma, coef_ws, dx, g_out, total4, C4,
             use_batch_stats, hoist, act == 2 ? 6.f : INFINITY);
  return ick::launch_status("bn_bwd_apply");
}

template <typename T>
int bn_train_apply_impl(const T* x, const double* sum, const double* sq, int stat_copies, int64_t stat_stride, const float* gamma, const float* beta,
                       float* running_mean, float* running_var, float momentum, float eps, const T* residual,
                       T* y, float* save_mean, float* save_invstd, long M, int C, int relu, void* stream) {
  ICK_REQUIRE(x && sum && sq && gamma && beta && y && save_mean && save_invstd && M > 0 && C % 4 == 0 && (C / 4) <= 1024 &&
              ((C / 4) & (C / 4 - 1)) == 0, "ick_bn_train_apply: C/4 must be a power of two <= 1024");
  const int C4 = C / 4;
  const long total4 = M * C4;
  int grid = grid_for(total4);
  const int q = C4 > NT ? C4 / NT : 1;          // grid * NT must be a multiple of C4
  grid = (grid + q - 1) / q * q;
  ICK_REQUIRE(stat_copies <= 1 || C <= 1024, "ick_bn_train_apply: accumulator copies are supported up to C = 1024 (C = %d)", C);
  ICK_LAUNCH(bn_train_apply_kernel<T>, dim3(grid), dim3(NT), 0, ST, x, sum, sq, stat_copies > 1 ? stat_copies : 1, (long)stat_stride, (float)M, gamma, beta, running_mean,
             running_var, momentum, eps, residual, y, save_mean, save_invstd, total4, C4, relu);
  return ick::launch_status("bn_train_apply");
}

template <typename T>
int maxpool3x3s2_impl(const T* x, T* y, int B, int H, int W, int C, void* stream) {
  ICK_REQUIRE(x && y && C % 4 == 0, "ick_maxpool3x3s2: bad arguments");
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  ICK_LAUNCH(maxpool3x3s2_kernel<T>, dim3(grid_for((long)B * Ho * Wo * (C / 4))), dim3(NT), 0, ST, x, y, B, H, W, C / 4,
                     Ho, Wo);
  return ick::launch_status("maxpool3x3s2");
}

template <typename T>
int conv_weight_dgrad_layout_impl(const T* w, T* wt, int Cout, int R, int S, int Cin, void* stream) {
  ICK_REQUIRE(w && wt && Cout > 0 && R > 0 && S > 0 && Cin > 0 && R * S <= 65535, "ick_conv_weight_dgrad_layout: bad arguments");
  ICK_LAUNCH(conv_weight_dgrad_layout_kernel<T>, dim3((Cin + 31) / 32, (Cout + 31) / 32, R * S), dim3(NT), 0, ST, w, wt, Cout,
             R * S, Cin);
  return ick::launch_status("conv_weight_dgrad_layout");
}

extern "C" {
// ---- fp32 and 16-bit entry points of the templated kernels above
int ick_bn_bwd_reduce(const float* dy, const float* y, const float* x, const float* mean, const float* invstd,
                      double* sum_g, double* sum_gx, int copies, int64_t stride, long M, int C, int act, void* stream) {
  return bn_bwd_reduce_impl<float>(dy, y, x, mean, invstd, sum_g, sum_gx, copies, stride, M, C, act, stream);
}
int ick_bn_bwd_reduce16(const void* dy, const void* y, const void* x, const float* mean, const float* invstd,
                        double* sum_g, double* sum_gx, int copies, int64_t stride, long M, int C, int act, int fp16, void* stream) {
  if (fp16) return bn_bwd_reduce_impl((const _Float16*)dy, (const _Float16*)y, (const _Float16*)x, mean, invstd, sum_g, sum_gx, copies, stride, M, C, act, stream);
  return bn_bwd_reduce_impl((const __bf16*)dy, (const __bf16*)y, (const __bf16*)x, mean, invstd, sum_g, sum_gx, copies, stride, M, C, act, stream);
}
int ick_bn_bwd_apply(const float* dy, const float* y, const float* x, const float* mean, const float* invstd,
                     const float* gamma, const double* sum_g, const double* sum_gx, int copies, int64_t stride, float* coef_ws,
                     float* dx, float* g_out, long M, int C, int use_batch_stats, float* dgamma, float* dbeta, int act, void* stream) {
  return bn_bwd_apply_impl<float>(dy, y, x, mean, invstd, gamma, sum_g, sum_gx, copies, stride, coef_ws, dx, g_out, M, C, use_batch_stats, dgamma, dbeta, act, stream);
}
int ick_bn_bwd_apply16(const void* dy, const void* y, const void* x, const float* mean, const float* invstd,
                       const float* gamma, const double* sum_g, const double* sum_gx, int copies, int64_t stride, float* coef_ws,
                       void* dx, void* g_out, long M, int C, int use_batch_stats, float* dgamma, float* dbeta, int act, int fp16, void* stream) {
  if (fp16) return bn_bwd_apply_impl((const _Float16*)dy, (const _Float16*)y, (const _Float16*)x, mean, invstd, gamma, sum_g, sum_gx, copies, stride, coef_ws, (_Float16*)dx, (_Float16*)g_out, M, C, use_batch_stats, dgamma, dbeta, act, stream);
  return bn_bwd_apply_impl((const __bf16*)dy, (const __bf16*)y, (const __bf16*)x, mean, invstd, gamma, sum_g, sum_gx, copies, stride, coef_ws, (__bf16*)dx, (__bf16*)g_out, M, C, use_batch_stats, dgamma, dbeta, act, stream);
}
int ick_bn_train_apply(const float* x, const double* sum, const double* sq, int stat_copies, int64_t stat_stride, const float* gamma, const float* beta,
                       float* running_mean, float* running_var, float momentum, float eps, const float* residual,
                       float* y, float* save_mean, float* save_invstd, long M, int C, int relu, void* stream) {
  return bn_train_apply_impl<float>(x, sum, sq, stat_copies, stat_stride, gamma, beta, running_mean, running_var, momentum, eps, residual, y, save_mean, save_invstd, M, C, relu, stream);
}
int ick_bn_train_apply16(const void* x, const double* sum, const double* sq, int stat_copies, int64_t stat_stride, const float* gamma, const float* beta,
                         float* running_mean, float* running_var, float momentum, float eps, const void* residual,
                         void* y, float* save_mean, float* save_invstd, long M, int C, int relu, int fp16, void* stream) {
  if (fp16) return bn_train_apply_impl((const _Float16*)x, sum, sq, stat_copies, stat_stride, gamma, beta, running_mean, running_var, momentum, eps, (const _Float16*)residual, (_Float16*)y, save_mean, save_invstd, M, C, relu, stream);
  return bn_train_apply_impl((const __bf16*)x, sum, sq, stat_copies, stat_stride, gamma, beta, running_mean, running_var, momentum, eps, (const __bf16*)residual, (__bf16*)y, save_mean, save_invstd, M, C, relu, stream);
}
int ick_maxpool3x3s2(const float* x, float* y, int B, int H, int W, int C, void* stream) { return maxpool3x3s2_impl<float>(x, y, B, H, W, C, stream); }
int ick_maxpool3x3s2_16(const void* x, void* y, int B, int H, int W, int C, int fp16, void* stream) {
  if (fp16) return maxpool3x3s2_impl((const _Float16*)x, (_Float16*)y, B, H, W, C, stream);
  return maxpool3x3s2_impl((const __bf16*)x, (__bf16*)y, B, H, W, C, stream);
}
int ick_conv_weight_dgrad_layout(const float* w, float* wt, int Cout, int R, int S, int Cin, void* stream) {
  return conv_weight_dgrad_layout_impl<float>(w, wt, Cout, R, S, Cin, stream);
}
int ick_conv_weight_dgrad_layout16(const void* w, void* wt, int Cout, int R, int S, int Cin, void* stream) {   /* pure data movement: bf16 and fp16 alike */
  return conv_weight_dgrad_layout_impl((const unsigned short*)w, (unsigned short*)wt, Cout, R, S, Cin, stream);
}


int ick_nchw3_to_nhwc4(const float* x, float* y, int B, int H, int W, void* stream) {
  ICK_REQUIRE(x && y && B > 0 && H > 0 && W > 0, "ick_nchw3_to_nhwc4: bad arguments");
  const long npix = (long)B * H * W;
  ICK_LAUNCH(nchw3_to_nhwc4_kernel, dim3(grid_for(npix)), dim3(NT), 0, ST, x, y, npix, (long)H * W);
  return ick::launch_status("nchw3_to_nhwc4");
}

int ick_nhwc4_to_nhwc3_add(const float* src4, float* dst3, int64_t npix, void* stream) {
  ICK_REQUIRE(src4 && dst3 && npix > 0, "ick_nhwc4_to_nhwc3_add: bad arguments");
  ICK_LAUNCH(nhwc4_to_nhwc3_add_kernel, dim3(grid_for(npix)), dim3(NT), 0, ST, src4, dst3, (long)npix);
  return ick::launch_status("nhwc4_to_nhwc3_add");
}

int ick_patchify16(const float* x, float* y, int B, int HW, void* stream) {
  ICK_REQUIRE(x && y && B > 0 && HW % 16 == 0, "ick_patchify16: bad arguments");
  const int G = HW / 16;
  ICK_LAUNCH(patchify16_kernel, dim3(grid_for((long)B * G * G * 192)), dim3(NT), 0, ST, x, y, B, HW, G);
  return ick::launch_status("patchify16");
}


int ick_vit_assemble(const float* patch, const float* cls, const float* pos, float* x, int B, int Ntok, int D, void* stream) {
  ICK_REQUIRE(patch && cls && pos && x && D % 4 == 0, "ick_vit_assemble: bad arguments");
  ICK_LAUNCH(vit_assemble_kernel, dim3(grid_for((long)B * Ntok * (D / 4))), dim3(NT), 0, ST, patch, cls, pos, x, B,
                     Ntok, D / 4);
  return ick::launch_status("vit_assemble");
}

int ick_bn_finalize(const double* sum, const double* sq, int stat_copies, int64_t stat_stride, float count, const float* gamma, const float* beta, float* rmean,
                    float* rvar, float momentum, float eps, float* scale, float* shift, float* save_mean,
                    float* save_invstd, int C, void* stream) {
  ICK_REQUIRE(sum && sq && gamma && beta && scale && shift && save_mean && save_invstd && C > 0 && count > 0,
              "ick_bn_finalize: bad arguments");
  ICK_LAUNCH(bn_finalize_kernel, dim3((C + NT - 1) / NT), dim3(NT), 0, ST, sum, sq, stat_copies > 1 ? stat_copies : 1, (long)stat_stride, count, gamma, beta, rmean, rvar,
                     momentum, eps, scale, shift, save_mean, save_invstd, C);
  return ick::launch_status("bn_finalize");
}

int ick_bn_eval_coeffs(const float* gamma, const float* beta, const float* rmean, const float* rvar, float eps,
                       float* scale, float* shift, int C, void* stream) {
  ICK_REQUIRE(gamma && beta && rmean && rvar && scale && shift && C > 0, "ick_bn_eval_coeffs: bad arguments");
  ICK_LAUNCH(bn_eval_coeffs_kernel, dim3((C + NT - 1) / NT), dim3(NT), 0, ST, gamma, beta, rmean, rvar, eps, scale,
                     shift, C);
  return ick::launch_status("bn_eval_coeffs");
}

int ick_scale_shift_act(const float* x, const float* scale, const float* shift, const float* residual, float* y, long M,
                        int C, int relu, void* stream) {
  ICK_REQUIRE(x && scale && shift && y && C % 4 == 0 && M > 0, "ick_scale_shift_act: bad arguments (C %% 4)");
  const long total4 = M * (C / 4);
  ICK_LAUNCH(scale_shift_act_kernel, dim3(grid_for(total4)), dim3(NT), 0, ST, x, scale, shift, residual, y, total4,
                     C / 4, relu);
  return ick::launch_status("scale_shift_act");
}



// train-mode BatchNorm forward in ONE pass over the raw conv output: every thread derives scale/shift of its own
// 4 channels from the fp64 batch sums the conv epilogue produced (the grid stride is a multiple of C/4, so a
// thread's channels never change), block 0 also stores mean / invstd for backward and updates the running stats.


int ick_maxpool3x3s2_bwd(const float* x, const float* dy, float* dx, int B, int H, int W, int C, void* stream) {
  ICK_REQUIRE(x && dy && dx && B > 0 && H > 0 && W > 0 && C > 0, "ick_maxpool3x3s2_bwd: bad arguments");
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  ICK_LAUNCH(maxpool3x3s2_bwd_kernel, dim3(grid_for((long)B * H * W * C)), dim3(NT), 0, ST, x, dy, dx, B, H, W, C, Ho, Wo);
  return ick::launch_status("maxpool3x3s2_bwd");
}

int ick_adaptive_avgpool_fwd(const float* x, float* y, int B, int H, int W, int C, int Ho, int Wo, void* stream) {
  ICK_REQUIRE(x && y && C % 4 == 0 && Ho > 0 && Wo > 0 && H > 0 && W > 0, "ick_adaptive_avgpool_fwd: bad arguments");   // (H < Ho: bins replicate, as torch's)
  ICK_LAUNCH(adaptive_avgpool_fwd_kernel, dim3(grid_for((long)B * Ho * Wo * (C / 4))), dim3(NT), 0, ST, x, y, B, H, W, C / 4, Ho, Wo);
  return ick::launch_status("adaptive_avgpool_fwd");
}

int ick_adaptive_avgpool_bwd(const float* dy, float* dx, int B, int H, int W, int C, int Ho, int Wo, void* stream) {
  ICK_REQUIRE(dy && dx && C % 4 == 0 && Ho > 0 && Wo > 0 && H > 0 && W > 0, "ick_adaptive_avgpool_bwd: bad arguments");
  ICK_LAUNCH(adaptive_avgpool_bwd_kernel, dim3(grid_for((long)B * H * W * (C / 4))), dim3(NT), 0, ST, dy, dx, B, H, W, C / 4, Ho, Wo);
  return ick::launch_status("adaptive_avgpool_bwd");
}

int ick_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd, long rows,
                      int D, float eps, void* stream) {
  ICK_REQUIRE(x && gamma && beta && y && D % 4 == 0 && rows > 0, "ick_layernorm_fwd: bad arguments (D %% 4)");
  ICK_LAUNCH(layernorm_fwd_kernel, dim3(grid_for(rows, 4)), dim3(NT), 0, ST, x, gamma, beta, y, mean, rstd, rows, D, eps);
  return ick::launch_status("layernorm_fwd");
}

int ick_layernorm_bwd(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd, float* dx,
                      float* dgamma, float* dbeta, long rows, int D, void* stream) {
  ICK_REQUIRE(dy && x && gamma && mean && rstd && dx && rows > 0 && D > 0 && D <= 4096, "ick_layernorm_bwd: bad arguments");
  ICK_REQUIRE((dgamma == nullptr) == (dbeta == nullptr), "ick_layernorm_bwd: dgamma and dbeta go together");
  ICK_LAUNCH(layernorm_bwd_kernel, dim3(grid_for(rows, 16, 1024)), dim3(NT), 2 * D * sizeof(float), ST, dy, x, gamma,
                     mean, rstd, dx, dgamma, dbeta, rows, D);
  return ick::launch_status("layernorm_bwd");
}

int ick_softmax_rows(float* s, long rows, int L, int ld, float scale, int causal, int Lq, void* stream) {
  ICK_REQUIRE(s && rows > 0 && L > 0 && ld >= L && (!causal || Lq > 0), "ick_softmax_rows: bad arguments");
  ICK_LAUNCH(softmax_rows_kernel, dim3(grid_for(rows, 4)), dim3(NT), 0, ST, s, rows, L, ld, scale, causal, Lq);
  return ick::launch_status("softmax_rows");
}

int ick_softmax_bwd_rows(float* dp, const float* p, long rows, int L, int ld, float scale, void* stream) {
  ICK_REQUIRE(dp && p && rows > 0 && L > 0 && ld >= L, "ick_softmax_bwd_rows: bad arguments");
  ICK_LAUNCH(softmax_bwd_rows_kernel, dim3(grid_for(rows, 4)), dim3(NT), 0, ST, dp, p, rows, L, ld, scale);
  return ick::launch_status("softmax_bwd_rows");
}

int ick_colsum(const float* x, float* out, long M, int N, long ld, void* stream) {
  ICK_REQUIRE(x && out && M > 0 && N > 0 && ld >= N, "ick_colsum: bad arguments");
  long gy = (M + 15) / 16; if (gy > 512) gy = 512;       // ~16 rows per thread (measured: 64 rows per thread left the launch latency-bound at 15 us)
  ICK_LAUNCH(colsum_kernel, dim3((N + NT - 1) / NT, (int)gy), dim3(NT), 0, ST, x, out, M, N, ld);
  return ick::launch_status("colsum");
}

int ick_absmax_f32(const float* x, int64_t n, float* out, void* stream) {
  ICK_REQUIRE(x && out && n > 0 && n % 4 == 0 && ick::aligned16(x), "ick_absmax_f32: bad arguments (n %% 4)");
  ICK_LAUNCH(absmax_kernel, dim3(grid_for(n / 4, NT * 8, 512)), dim3(NT), 0, ST, x, (long)(n / 4), out);
  return ick::launch_status("absmax_f32");
}

int ick_relu_bwd(const float* dy, const float* y, float* dx, long n, void* stream) {
  ICK_REQUIRE(dy && y && dx && n % 4 == 0, "ick_relu_bwd: n %% 4");
  ICK_LAUNCH(relu_bwd_kernel, dim3(grid_for(n / 4)), dim3(NT), 0, ST, dy, y, dx, n / 4);
  return ick::launch_status("relu_bwd");
}

int ick_add(const float* a, const float* b, float* y, long n, void* stream) {
  ICK_REQUIRE(a && b && y && n % 4 == 0, "ick_add: n %% 4");
  ICK_LAUNCH(add_kernel, dim3(grid_for(n / 4)), dim3(NT), 0, ST, a, b, y, n / 4);
  return ick::launch_status("add");
}

int ick_embedding_fwd(const int64_t* ids, const float* table, const float* pe, float* out, long n, int D, int per_pos,
                      void* stream) {
  ICK_REQUIRE(ids && table && out && D % 4 == 0 && n > 0, "ick_embedding_fwd: bad arguments");
  ICK_LAUNCH(embedding_fwd_kernel, dim3(grid_for(n * (D / 4))), dim3(NT), 0, ST, (const long*)ids, table, pe, out, n,
                     D / 4, per_pos != 0 ? per_pos : 1);
  return ick::launch_status("embedding_fwd");
}

int ick_embedding_bwd(const int64_t* ids, const float* dout, float* dtable, long n, int D, void* stream) {
  ICK_REQUIRE(ids && dout && dtable && n > 0 && D > 0, "ick_embedding_bwd: bad arguments");
  ICK_LAUNCH(embedding_bwd_kernel, dim3(grid_for(n * D)), dim3(NT), 0, ST, (const long*)ids, dout, dtable, n, D);
  return ick::launch_status("embedding_bwd");
}

int ick_token_pool_fwd(const float* x, float* y, int B, int L, int Lo, int D, void* stream) {
  ICK_REQUIRE(x && y && D % 4 == 0 && Lo > 0 && L >= Lo, "ick_token_pool_fwd: bad arguments");
  ICK_LAUNCH(token_pool_fwd_kernel, dim3(grid_for((long)B * Lo * (D / 4))), dim3(NT), 0, ST, x, y, B, L, Lo, D / 4);
  return ick::launch_status("token_pool_fwd");
}

int ick_token_pool_bwd(const float* dy, float* dx, int B, int L, int Lo, int D, void* stream) {
  ICK_REQUIRE(dy && dx && D % 4 == 0 && Lo > 0 && L >= Lo, "ick_token_pool_bwd: bad arguments");
  ICK_LAUNCH(token_pool_bwd_kernel, dim3(grid_for((long)B * L * (D / 4))), dim3(NT), 0, ST, dy, dx, B, L, Lo, D / 4);
  return ick::launch_status("token_pool_bwd");
}

int ick_dropout(const float* x, float* y, int64_t n, float p, uint64_t seed, const int64_t* step, void* stream) {
  ICK_REQUIRE(x && y && n > 0 && p >= 0.f && p < 1.f, "ick_dropout: bad arguments");
  ICK_LAUNCH(dropout_kernel, dim3(grid_for(n)), dim3(NT), 0, ST, x, y, (long)n, p, 1.0f / (1.0f - p),
                     (unsigned long long)seed, (const long*)step);
  return ick::launch_status("dropout");
}

}  // extern "C"
