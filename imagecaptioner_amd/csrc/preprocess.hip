// preprocess.hip — the training / evaluation image transform on the GPU, bit-exact with the reference's PIL pipeline
// (reference: /root/reference/src/train_student_kd.py:122-135 — torchvision Resize((224,224)) -> ColorJitter(.1,.1,.1,.05)
//  -> RandomHorizontalFlip(.3) -> ToTensor -> Normalize; torchvision applies them to PIL images, so the arithmetic is
//  Pillow's 8-bit arithmetic: ImagingResample (bilinear = triangle filter with support scaling, 22-bit fixed-point
//  coefficients, horizontal pass then vertical pass, each rounded to uint8), ImageEnhance = ImagingBlend against a
//  degenerate image in float32 with truncation, rgb2hsv / hsv2rgb of Convert.c).
// HBM-bound byte work: uint8 in, fp32 NCHW out, one workgroup per image for everything after
// the horizontal pass so that the 224x224x3 working image (147 KB) never leaves LDS between the colour operations,
// two of which need a whole-image reduction (the contrast mean).
#include "ick_common.h"

namespace {

constexpr int OUT = 224;
constexpr int PB = 22;                      // Pillow PRECISION_BITS = 32 - 8 - 2
constexpr int NPIX = OUT * OUT;

struct Item {                               // one image of a launch group (all images of a group share a coefficient table)
  long src_off;                             // byte offset of the HWC uint8 source image
  long tmp_off;                             // byte offset of its [H][224][3] horizontal-pass image in the scratch buffer
  int H, W;
  int dst;                                  // batch index of the output
  int pad;
};

struct Jitter {                             // per image; torchvision ColorJitter.get_params + RandomHorizontalFlip draws
  int order[4];                             // permutation of {0 brightness, 1 contrast, 2 saturation, 3 hue}; -1 = skip
  float brightness, contrast, saturation;
  int hue_shift;                            // uint8(hue_factor * 255), added to H modulo 256
  int flip;
  int pad[2];
};

__device__ __forceinline__ unsigned char clip8(int v) { return (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

// horizontal pass: out[y][xx][c] = clip8((2^21 + sum_x in[y][xmin+x][c] * k[xx][x]) >> 22)
__global__ void resize_h_kernel(const unsigned char* __restrict__ src, unsigned char* __restrict__ tmp,
                                const Item* __restrict__ items, const int* __restrict__ bounds,
                                const int* __restrict__ coefs, int ksize) {
  const Item it = items[blockIdx.y];
  const long total = (long)it.H * OUT;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int xx = (int)(i % OUT);
    const long y = i / OUT;
    const int xmin = bounds[2 * xx], n = bounds[2 * xx + 1];
    const unsigned char* row = src + it.src_off + (y * it.W + xmin) * 3;
    const int* k = coefs + (long)xx * ksize;
    int a0 = 1 << (PB - 1), a1 = a0, a2 = a0;
    for (int x = 0; x < n; ++x) {
      const int w = k[x];
      a0 += row[3 * x] * w; a1 += row[3 * x + 1] * w; a2 += row[3 * x + 2] * w;
    }
    unsigned char* o = tmp + it.tmp_off + i * 3;
    o[0] = clip8(a0 >> PB); o[1] = clip8(a1 >> PB); o[2] = clip8(a2 >> PB);
  }
}

__device__ __forceinline__ int to_l(int r, int g, int b) { return (r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16; }

// ImagingBlend(degenerate, image, factor): float32 arithmetic without contraction, clamp, truncate
__device__ __forceinline__ unsigned char blend(int deg, int x, float f) {
  const float t = __fadd_rn((float)deg, __fmul_rn(f, __fsub_rn((float)x, (float)deg)));
  return t <= 0.f ? 0 : (t >= 255.f ? 255 : (unsigned char)t);
}

// Convert.c rgb2hsv_row / hsv2rgb_row ("following colorsys.py"), with C's float/double promotions reproduced
__device__ __forceinline__ void rgb2hsv(int r, int g, int b, int& uh, int& us, int& uv) {
  const int maxc = max(r, max(g, b)), minc = min(r, min(g, b));
  uv = maxc;
  if (minc == maxc) { uh = 0; us = 0; return; }
  const float cr = (float)(maxc - minc);
  const float s = __fdiv_rn(cr, (float)maxc);
  const float rc = __fdiv_rn((float)(maxc - r), cr), gc = __fdiv_rn((float)(maxc - g), cr), bc = __fdiv_rn((float)(maxc - b), cr);
  float h;
  if (r == maxc) h = __fsub_rn(bc, gc);
  else if (g == maxc) h = (float)(2.0 + (double)rc - (double)bc);
  else h = (float)(4.0 + (double)gc - (double)rc);
  h = (float)fmod((double)h / 6.0 + 1.0, 1.0);
  uh = min(255, max(0, (int)((double)h * 255.0)));
  us = min(255, max(0, (int)((double)s * 255.0)));
}

__device__ __forceinline__ void hsv2rgb(int h, int s, int v, int& r, int& g, int& b) {
  if (s == 0) { r = g = b = v; return; }
  const double fs = (double)s / 255.0, fv = (double)v;
  const double h6 = (double)h * 6.0 / 255.0;
  const int i = (int)floor(h6);
  const double f = h6 - (double)i;
  const int p = min(255, max(0, (int)floor(fv * (1.0 - fs) + 0.5)));
  const int q = min(255, max(0, (int)floor(fv * (1.0 - fs * f) + 0.5)));
  const int t = min(255, max(0, (int)floor(fv * (1.0 - fs * (1.0 - f)) + 0.5)));
  switch (i % 6) {
    case 0: r = v; g = t; b = p; break;
    case 1: r = q; g = v; b = p; break;
    case 2: r = p; g = v; b = t; break;
    case 3: r = p; g = q; b = v; break;
    case 4: r = t; g = p; b = v; break;
    default: r = v; g = p; b = q; break;
  }
}

// vertical pass into LDS, colour jitter in LDS, flip + ToTensor + Normalize on the way out.  One workgroup per image.
__global__ __launch_bounds__(1024) void resize_v_jitter_norm_kernel(
    const unsigned char* __restrict__ src, const unsigned char* __restrict__ tmp, const Item* __restrict__ items,
    const int* __restrict__ bounds, const int* __restrict__ coefs, int ksize, const Jitter* __restrict__ jit,
    float* __restrict__ out, float m0, float m1, float m2, float s0, float s1, float s2) {
  extern __shared__ __attribute__((aligned(16))) unsigned char img[];   // [224][224][3]
  __shared__ int red[16];
  const Item it = items[blockIdx.x];
  const int tid = threadIdx.x, nt = blockDim.x;
  // ---- vertical pass (or plain copy when H == 224); source = horizontal-pass image, or the input itself when W == 224
  const unsigned char* in = it.W == OUT ? src + it.src_off : tmp + it.tmp_off;
  if (it.H == OUT) {
    for (int i = tid; i < NPIX * 3; i += nt) img[i] = in[i];
  } else {
    for (int i = tid; i < NPIX; i += nt) {
      const int yy = i / OUT, x = i - yy * OUT;
      const int ymin = bounds[2 * yy], n = bounds[2 * yy + 1];
      const int* k = coefs + (long)yy * ksize;
      const unsigned char* col = in + ((long)ymin * OUT + x) * 3;
      int a0 = 1 << (PB - 1), a1 = a0, a2 = a0;
      for (int y = 0; y < n; ++y) {
        const int w = k[y];
        a0 += col[(long)y * OUT * 3] * w; a1 += col[(long)y * OUT * 3 + 1] * w; a2 += col[(long)y * OUT * 3 + 2] * w;
      }
      img[3 * i] = clip8(a0 >> PB); img[3 * i + 1] = clip8(a1 >> PB); img[3 * i + 2] = clip8(a2 >> PB);
    }
  }
  __syncthreads();
  // ---- ColorJitter: the four operations in the drawn order, each on the whole uint8 image
  const Jitter* jp = jit ? jit + it.dst : nullptr;
  if (jp) {
    for (int op = 0; op < 4; ++op) {
      const int which = jp->order[op];
      if (which == 0) {                                   // brightness: blend(black, img, f)
        const float f = jp->brightness;
        for (int i = tid; i < NPIX * 3; i += nt) img[i] = blend(0, img[i], f);
      } else if (which == 1) {                            // contrast: blend(mean gray, img, f)
        int part = 0;
        for (int i = tid; i < NPIX; i += nt) part += to_l(img[3 * i], img[3 * i + 1], img[3 * i + 2]);
        for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
        if ((tid & 63) == 0) red[tid >> 6] = part;
        __syncthreads();
        int sum = 0;
        for (int w = 0; w < (nt >> 6); ++w) sum += red[w];
        const int mean = (int)((double)sum / (double)NPIX + 0.5);
        const float f = jp->contrast;
        __syncthreads();
        for (int i = tid; i < NPIX * 3; i += nt) img[i] = blend(mean, img[i], f);
      } else if (which == 2) {                            // saturation: blend(gray(img), img, f)
        const float f = jp->saturation;
        for (int i = tid; i < NPIX; i += nt) {
          const int r = img[3 * i], g = img[3 * i + 1], b = img[3 * i + 2];
          const int l = to_l(r, g, b);
          img[3 * i] = blend(l, r, f); img[3 * i + 1] = blend(l, g, f); img[3 * i + 2] = blend(l, b, f);
        }
      } else if (which == 3) {                            // hue: RGB -> HSV, H += shift (mod 256), HSV -> RGB
        const int shift = jp->hue_shift;
        for (int i = tid; i < NPIX; i += nt) {
          int h, s, v, r, g, b;
          rgb2hsv(img[3 * i], img[3 * i + 1], img[3 * i + 2], h, s, v);
          hsv2rgb((h + shift) & 255, s, v, r, g, b);
          img[3 * i] = (unsigned char)r; img[3 * i + 1] = (unsigned char)g; img[3 * i + 2] = (unsigned char)b;
        }
      }
      __syncthreads();
    }
  }
  // ---- RandomHorizontalFlip + ToTensor (x / 255) + Normalize ((x - mean) / std), CHW fp32
  const int flip = jp ? jp->flip : 0;
  float* o = out + (long)it.dst * 3 * NPIX;
  const float mean[3] = {m0, m1, m2}, sd[3] = {s0, s1, s2};
  for (int i = tid; i < NPIX * 3; i += nt) {
    const int c = i / NPIX, p = i - c * NPIX;
    const int y = p / OUT, x = p - y * OUT;
    const int xs = flip ? OUT - 1 - x : x;
    const float v = __fdiv_rn((float)img[(y * OUT + xs) * 3 + c], 255.f);
    o[i] = __fdiv_rn(__fsub_rn(v, mean[c]), sd[c]);
  }
}

}  // namespace

extern "C" {

// items: n_items records {int64 src_off, int64 tmp_off, int32 H, int32 W, int32 dst, int32 pad} (32 bytes each), all of
// one source WIDTH; bounds [224][2] / coefs [224][ksize]: Pillow's precompute_coeffs + normalize_coeffs_8bpc for
// (W -> 224), computed by the host (imagecaptioner_amd/data_pipeline.py).
int ick_resize_h_u8(const uint8_t* src, uint8_t* tmp, const void* items, int n_items, int max_h, const int32_t* bounds,
                    const int32_t* coefs, int ksize, void* stream) {
  ICK_REQUIRE(src && tmp && items && bounds && coefs && n_items > 0 && max_h > 0 && ksize > 0, "ick_resize_h_u8: bad arguments");
  long blocks = ((long)max_h * OUT + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  ICK_LAUNCH(resize_h_kernel, dim3((int)blocks, n_items), dim3(256), 0, static_cast<hipStream_t>(stream), src, tmp,
             static_cast<const Item*>(items), bounds, coefs, ksize);
  return ick::launch_status("resize_h_u8");
}

// items of one source HEIGHT; jitter: NULL (evaluation transform) or [batch] records of 44 bytes indexed by Item.dst:
// {int32 order[4], float brightness, contrast, saturation, int32 hue_shift, int32 flip, int32 pad[2]}.
// out: (batch,3,224,224) fp32 = Normalize(ToTensor(.)) with the given mean / std.
int ick_resize_v_jitter_normalize(const uint8_t* src, const uint8_t* tmp, const void* items, int n_items,
                                  const int32_t* bounds, const int32_t* coefs, int ksize, const void* jitter, float* out,
                                  const float* mean3_host, const float* std3_host, void* stream) {
  ICK_REQUIRE(src && tmp && items && out && n_items > 0 && mean3_host && std3_host, "ick_resize_v_jitter_normalize: bad arguments");
  ICK_REQUIRE((bounds && coefs && ksize > 0) || (!bounds && !coefs), "ick_resize_v_jitter_normalize: bounds and coefs go together");
  static bool attr_set = false;
  const int lds = OUT * OUT * 3;
  if (!attr_set) {   // 147 KB of dynamic LDS per workgroup (the CU has 160 KB)
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(resize_v_jitter_norm_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return ick::fail((int)e, "ick_resize_v_jitter_normalize: %s", hipGetErrorString(e));
    attr_set = true;
  }
  ICK_LAUNCH(resize_v_jitter_norm_kernel, dim3(n_items), dim3(1024), lds, static_cast<hipStream_t>(stream), src, tmp,
             static_cast<const Item*>(items), bounds, coefs, ksize, static_cast<const Jitter*>(jitter), out,
             mean3_host[0], mean3_host[1], mean3_host[2], std3_host[0], std3_host[1], std3_host[2]);
  return ick::launch_status("resize_v_jitter_normalize");
}

}  // extern "C"
