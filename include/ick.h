/*
 * ick.h — C ABI of libick.so, the MI355X (gfx950) kernel library behind the
 * image-captioning knowledge-distillation hot path.
 *
 * The reference (VeeraKarthick609/ImageCaptioner) has NO native/FFI layer: its
 * boundary is the Python nn.Module surface.  This header is the
 * build-side C boundary underneath that surface; every entry names the reference
 * computation it replaces (file:line into /root/reference).  A maintainer of the
 * reference binds it with ctypes exactly as imagecaptioner_amd/_lib.py does
 * (see INTEGRATION.md).
 *
 * Conventions
 *   - plain pointers + sizes only; all pointers are DEVICE pointers (HBM) unless named host_*
 *   - every launcher takes the HIP stream as `void* stream` (hipStream_t); nothing syncs,
 *     nothing allocates: callers own all outputs and workspaces
 *   - return 0 on success, <0 for a rejected argument, >0 = hipError_t of the launch;
 *     ick_last_error() returns a thread-local message for the last non-zero return
 *   - tensors are fp32; activations of the CNN are NHWC ("channels_last"), conv weights
 *     are [Cout][R][S][Cin] (the physical layout of a channels_last OIHW torch tensor)
 */
#ifndef ICK_H
#define ICK_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ICK_ABI_VERSION 8

const char* ick_last_error(void);
int ick_abi_version(void);

/* ------------------------------------------------------------------ implicit GEMM (fp32 MFMA)
 * C[z][m][n] = act(alpha * sum_k A(m,k) * B(n,k) + bias[n]) + residual[m][n]
 * One kernel family (v_mfma_f32_32x32x2_f32, LDS-tiled) serves every dense contraction of
 * the path: nn.Linear fwd/bwd (student_model.py:37-42,91-96,138-139,151-156; teacher_model.py:50,60-71;
 * distillation_utils.py:217-222), attention QK^T / PV batched over (batch, head)
 * (student_model.py:83-88, teacher_model.py:60-67, timm ViT blocks), and ResNet-50 convolutions
 * fwd / dgrad / wgrad as implicit GEMM over NHWC (student_model.py:16-20,57).
 */
enum {
  ICK_OP_NT = 0,        /* A [M][K] (lda), B [N][K] (ldb)             : y = x W^T               */
  ICK_OP_NN = 1,        /* A [M][K] (lda), B [K][N] (ldb)             : dx = dy W               */
  ICK_OP_TN = 2,        /* A [K][M] (lda), B [K][N] (ldb)             : dW = dy^T x             */
  ICK_OP_CONV_FWD = 3,  /* A = im2col(X NHWC), B = W [Cout][R*S*Cin]                             */
  ICK_OP_CONV_FWD_C4 = 4,/* same, Cin == 4 (padded RGB stem), K = R*S*4                          */
  ICK_OP_CONV_DGRAD = 5,/* A = gather(dY NHWC), B = W as [k=(tap,co)][n=ci]  -> dX NHWC          */
  ICK_OP_CONV_WGRAD = 6,/* A = dY as [K=B*Ho*Wo][M=Cout], B = gather(X) [K][N=(tap,ci)] -> dW    */
  ICK_OP_CONV_DGRAD_S2 = 7 /* stride-2 dgrad split into the 4 input-pixel parity classes (grid.z): M = Nb*(H/2)*(W/2) */
};
enum { ICK_ACT_NONE = 0, ICK_ACT_RELU = 1, ICK_ACT_GELU = 2, ICK_ACT_TANH = 3,
       ICK_ACT_POST_RESIDUAL = 16 /* OR-ed into act: apply the activation AFTER adding the residual */ };

typedef struct IckGemm {
  const float* A; const float* B; float* C;
  const float* bias;       /* [N] or NULL */
  const float* residual;   /* [M][ldr] or NULL, same batch strides as C */
  double* stat_sum;        /* [N] or NULL: += column sums of the raw (pre-activation) product (BatchNorm batch stats), fp64 accumulators */
  double* stat_sq;         /* [N] or NULL: += column sums of squares (fp64) */
  int32_t op, act;
  int32_t M, N, K;
  int64_t lda, ldb, ldc, ldr;
  int32_t batch_outer, batch_inner;             /* grid.z = outer*inner (1,1 for plain) */
  int64_t sAo, sAi, sBo, sBi, sCo, sCi;         /* element strides per outer / inner batch index */
  int32_t splitk;                               /* >1: K split over grid.z, C accumulated with fp32 atomics (C pre-zeroed or holding the value to add to) */
  int32_t accumulate;                           /* 1: C += result (non-atomic, splitk==1) */
  float alpha;
  /* convolution geometry (ICK_OP_CONV_*): X [Nb][H][W][Cin], Y [Nb][Ho][Wo][Cout] */
  int32_t Nb, H, W, Cin, Ho, Wo, Cout, R, S, stride, pad;
  int32_t tile;                                 /* 0 = choose by the wave-quantisation model; 1 = 128x128, 2 = 64x64, 3 = 128x64, 4 = 64x128; +16 = three LDS buffers (LDS-DMA kernel); +64 = eight waves per workgroup (LDS-DMA kernel, 128-row tiles: 65, 67, 83); +32 = M-split: 128x128 workgroups for the rows that fill whole rounds of the chip, the tile named by the low bits for the remaining rows (two launches); +256 = register-staged kernel */
  int32_t stat_copies;                          /* <= 1: one accumulator row; R > 1: stat_sum/stat_sq are [R][stat_stride] and the row-tile t of the grid adds into copy t % R (spreads the fp64 atomics of large-M convolutions over R x as many cache lines; consumers sum the copies) */
  int64_t stat_stride;                          /* elements between two copies (>= N) */
  const float* col_scale;                       /* [N] or NULL: C = act(col_scale[n] * alpha*sum + bias[n] ...) — eval-mode BatchNorm folded into the conv epilogue (scale = gamma/sqrt(var+eps), bias = shift); needs N %% 4, ldc %% 4, no split-K */
  int32_t kchunk;                               /* fp32 kernels: k elements one MFMA accumulator chain sums before it is folded into a master accumulator (bounds the rounding error of long-K products the way a K-blocked CPU GEMM does); 0 = library default (64), < 0 = one chain over all of K */
  int32_t io16;                                 /* ick_gemm_h16 only (0 elsewhere): bit 0 = C holds 16-bit elements of the operand type (also what `accumulate` reads), bit 1 = the residual does; both need the vector epilogue (N %% 4, ldc %% 4, ldr %% 4, no split-K) */
  const float* a_absmax;                        /* ick_gemm_bf16(terms = 4) only (NULL elsewhere): device pointer to max |A| (ick_absmax_f32).  The kernel multiplies A by the power of two that brings that maximum to [2^10, 2^11) before the fp16 split and divides the result by it: three-product GEMMs for operands of ANY magnitude, e.g. the data gradients of the trunk (1e-6 and below), with full 2^-22 relative precision for elements down to 3e-8 of the maximum */
} IckGemm;

int ick_gemm_f32(const IckGemm* desc, void* stream);

/* The same family on the bf16 matrix cores (v_mfma_f32_32x32x16_bf16, fp32 accumulate) for the mixed-precision regime
 * the reference trains in (torch.cuda.amp.autocast + GradScaler, train_student_kd.py:263-290).  Operands, layouts and
 * epilogues are those of ick_gemm_f32 (fp32 in HBM); values are rounded to bf16 on the way into LDS.
 * terms = 1: plain bf16 products; terms = 2: fp16 products (v_mfma_f32_32x32x16_f16 — the reference's autocast dtype: 10
 * mantissa bits, 5-bit exponent, to be run under the loss scaler); terms = 3: each operand split into hi + lo bf16 parts and
 * hi*hi + hi*lo + lo*hi accumulated (~1e-5 relative to the exact fp32 product, 3 MFMAs per k-step); terms = 4 ("f32x3",
 * round 3): fp32-GRADE products from three fp16 MFMAs — a = hi + 2^-11 lo', hi = fp16(a), lo' = fp16(2^11 (a - hi)); hi*hi in
 * one accumulator, hi*lo' + lo'*hi in a second one folded in with 2^-11 after the k-loop — error against float64 equal to
 * ick_gemm_f32's (tests/test_gemm_gpu.py::test_f32x3_is_fp32_grade) at half its time; NT and CONV_FWD descriptors the LDS-DMA
 * kernel takes run it, every other descriptor is forwarded to ick_gemm_f32.  Operand magnitudes must lie in fp16's range
 * (< 65504; values under 6e-5 keep an ABSOLUTE accuracy of 1.5e-11): meant for forward activations and weights — the host
 * layer (ops.gemm_raw x3=True) never routes gradients here.
 * IckGemm.io16 (terms 1 / 2, shapes the LDS-DMA kernel takes): C / the residual stored as bf16 / fp16 — the 4-channel stem
 * convolution of the 16-bit training regime reads fp32 images and writes 16-bit activations this way.
 * Convolutions whose channel count is not a multiple of 32 run on the exact-fp32 kernel. */
int ick_gemm_bf16(const IckGemm* desc, int terms, void* stream);
/* out[0] = max(out[0], max_i |x[i]|) — out must hold a non-negative float on entry (0 for a fresh maximum); n %% 4 == 0.  The
 * producer of IckGemm.a_absmax. */
int ick_absmax_f32(const float* x, int64_t n, float* out, void* stream);
/* NATIVE 16-bit operands: A and B hold bf16 (fp16 = 0) or fp16 (fp16 = 1) elements in HBM and in LDS — the storage the
 * reference's autocast keeps its activations and weight copies in (train_student_kd.py:271); fp32 accumulation, fp32
 * bias / statistics / epilogue arithmetic; C and the residual are fp32 or (IckGemm.io16) 16-bit.  Every op of the family;
 * all extents, leading dimensions and the conv geometry are in ELEMENTS and follow ick_gemm_bf16's rules (multiples of 4;
 * conv channel counts multiples of 32); no split-K into a 16-bit C.  Operands whose k index is contiguous (NT, CONV_FWD) with
 * K, lda, ldb multiples of 8 and Cin a multiple of 64 run on the LDS-DMA kernel (one ds_read_b128 = one MFMA operand); the
 * others on the register-staged kernel (transposed LDS reads).  ick_cast_f32_to_16 / ick_cast_16_to_f32: n %% 4 == 0. */
int ick_gemm_h16(const IckGemm* desc, int fp16, void* stream);
int ick_cast_f32_to_16(const float* x, void* y, int64_t n, int fp16, void* stream);
int ick_cast_16_to_f32(const void* x, float* y, int64_t n, int fp16, void* stream);

/* ------------------------------------------------------------------ fused attention forward (head dim 64)
 * softmax(Q K^T * scale [causal]) V per (batch, head) without materialising the scores: timm ViT-S/16 self-attention
 * (teacher_model.py:82), nn.TransformerDecoderLayer self/cross attention (teacher_model.py:60-67).  Element (b,row,h,c) of
 * an operand lives at base + b*bs + row*ld + h*64 + c (packed in_proj outputs are addressed in place; kbs = vbs = 0 lets
 * every batch entry attend to the same memory, the beam-search case). */
int ick_attention_fwd_d64(const float* q, int64_t qld, int64_t qbs, const float* k, int64_t kld, int64_t kbs,
                          const float* v, int64_t vld, int64_t vbs, float* o, int64_t old, int64_t obs,
                          int B, int H, int Lq, int Lk, int causal, float scale, void* stream);
/* The same contract with every product as three fp16 MFMAs (x = hi + 2^-11 lo'; fp32-grade: error against float64 equal to
 * the fp32-MFMA kernel's, tests/test_attention_gpu.py) — v_mfma_f32_32x32x16_f16 instead of v_mfma_f32_32x32x2_f32, K and V
 * split once per chunk on their way into LDS.  Operands inside fp16's range (the teacher's LayerNorm'd activations through a
 * Linear); selected by the host layer under ops.precision("f32x3"). */
int ick_attention_fwd_d64_x3(const float* q, int64_t qld, int64_t qbs, const float* k, int64_t kld, int64_t kbs,
                             const float* v, int64_t vld, int64_t vbs, float* o, int64_t old, int64_t obs,
                             int B, int H, int Lq, int Lk, int causal, float scale, void* stream);

/* ------------------------------------------------------------------ layout transforms
 * images arrive as the reference hands them over: (B,3,224,224) fp32 NCHW (train_student_kd.py:259). */
int ick_nchw3_to_nhwc4(const float* x, float* y, int B, int H, int W, void* stream);      /* -> (B,H,W,4), 4th channel 0 */
int ick_nhwc4_to_nhwc3_add(const float* src4, float* dst3, int64_t npix, void* stream);      /* dst3[p][c] += src4[p][c], c < 3 */
int ick_patchify16(const float* x, float* y, int B, int HW, void* stream);                /* -> [B*(HW/16)^2][768], k=(c,py,px): timm PatchEmbed as a GEMM */
int ick_conv_weight_dgrad_layout(const float* w, float* wt, int Cout, int R, int S, int Cin, void* stream); /* wt[ci][R-1-r][S-1-s][co] = w[co][r][s][ci]: with it the stride-1 data gradient of a convolution (autograd of student_model.py:57 through layer3/layer4) is a forward convolution over dY with pad R-1-pad, both GEMM operands k-contiguous */
int ick_conv_weight_dgrad_layout16(const void* w, void* wt, int Cout, int R, int S, int Cin, void* stream);   /* the same on 16-bit elements (bf16 or fp16) */
int ick_vit_assemble(const float* patch, const float* cls, const float* pos, float* x, int B, int Ntok, int D, void* stream); /* cls token + pos_embed (timm forward_features) */

/* ------------------------------------------------------------------ input transform, bit-exact with Pillow
 * torchvision Resize((224,224)) -> ColorJitter(.1,.1,.1,.05) -> RandomHorizontalFlip(.3) -> ToTensor -> Normalize on PIL
 * images (train_student_kd.py:122-135; val: :130-134 without jitter / flip).  uint8 HWC RGB images of any size, packed in
 * one device buffer, in; (B,3,224,224) fp32 out.  Coefficient tables are Pillow's precompute_coeffs (bilinear, support
 * scaling) + normalize_coeffs_8bpc, built on the host per distinct source size; random draws are made by the host in
 * torchvision's order (imagecaptioner_amd/data_pipeline.py). */
int ick_resize_h_u8(const uint8_t* src, uint8_t* tmp, const void* items, int n_items, int max_h, const int32_t* bounds,
                    const int32_t* coefs, int ksize, void* stream);
int ick_resize_v_jitter_normalize(const uint8_t* src, const uint8_t* tmp, const void* items, int n_items,
                                  const int32_t* bounds, const int32_t* coefs, int ksize, const void* jitter, float* out,
                                  const float* mean3_host, const float* std3_host, void* stream);

/* ------------------------------------------------------------------ BatchNorm2d over NHWC rows [M = B*H*W][C]
 * nn.BatchNorm2d inside torchvision resnet50 (student_model.py:16-20,57); train mode = batch statistics + running-stat
 * update, also for the "frozen" stem.  Batch sums come from the conv epilogue (IckGemm.stat_*). */
int ick_bn_finalize(const double* sum, const double* sq, int stat_copies, int64_t stat_stride, float count, const float* gamma,
                    const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                    float* scale, float* shift, float* save_mean, float* save_invstd, int C, void* stream);
int ick_bn_eval_coeffs(const float* gamma, const float* beta, const float* running_mean, const float* running_var, float eps,
                       float* scale, float* shift, int C, void* stream);
int ick_scale_shift_act(const float* x, const float* scale, const float* shift, const float* residual, float* y,
                        int64_t M, int C, int relu, void* stream);                          /* y = [relu](x*scale+shift [+ residual]) */
int ick_bn_bwd_reduce(const float* dy, const float* y, const float* x, const float* mean, const float* invstd,
                      double* sum_g, double* sum_gx, int copies, int64_t stride, int64_t M, int C, int act /* mask of y: 1 = ReLU (y > 0), 2 = ReLU6 (0 < y < 6) */, void* stream); /* += sum(g), sum(g*xhat) in fp64, spread over `copies` accumulator rows `stride` elements apart (row-block b adds into row b %% copies; ick_bn_bwd_apply folds them) (the reference's CPU batch_norm backward reduces in double); g = dy*(y>0) if y */
int ick_bn_bwd_reduce16(const void* dy, const void* y, const void* x, const float* mean, const float* invstd,
                        double* sum_g, double* sum_gx, int copies, int64_t stride, int64_t M, int C, int act, int fp16, void* stream);  /* dy, y, x stored as bf16 (fp16 = 0) / fp16 (1): the 16-bit training regime */
int ick_bn_bwd_apply(const float* dy, const float* y, const float* x, const float* mean, const float* invstd,
                     const float* gamma, const double* sum_g, const double* sum_gx, int copies, int64_t stride, float* coef_ws /* [2*C] scratch */,
                     float* dx, float* g_out, int64_t M, int C, int use_batch_stats, float* dgamma, float* dbeta, int act /* as in ick_bn_bwd_reduce */, void* stream); /* dgamma/dbeta (optional) += the two sums */
int ick_bn_bwd_apply16(const void* dy, const void* y, const void* x, const float* mean, const float* invstd,
                       const float* gamma, const double* sum_g, const double* sum_gx, int copies, int64_t stride, float* coef_ws,
                       void* dx, void* g_out, int64_t M, int C, int use_batch_stats, float* dgamma, float* dbeta, int act, int fp16, void* stream);  /* 16-bit dy, y, x, dx, g_out */
int ick_bn_train_apply(const float* x, const double* sum, const double* sq, int stat_copies, int64_t stat_stride, const float* gamma, const float* beta,
                       float* running_mean, float* running_var, float momentum, float eps, const float* residual,
                       float* y, float* save_mean, float* save_invstd, int64_t M, int C, int relu, void* stream); /* bn_finalize + scale_shift_act in one pass */
int ick_bn_train_apply16(const void* x, const double* sum, const double* sq, int stat_copies, int64_t stat_stride, const float* gamma, const float* beta,
                         float* running_mean, float* running_var, float momentum, float eps, const void* residual,
                         void* y, float* save_mean, float* save_invstd, int64_t M, int C, int relu, int fp16, void* stream);  /* 16-bit x, residual, y; statistics and parameters fp32 / fp64 as above */
int ick_maxpool3x3s2(const float* x, float* y, int B, int H, int W, int C, void* stream); /* nn.MaxPool2d(3,2,1), NHWC */
int ick_maxpool3x3s2_16(const void* x, void* y, int B, int H, int W, int C, int fp16, void* stream);   /* 16-bit activations */
int ick_maxpool3x3s2_bwd(const float* x, const float* dy, float* dx, int B, int H, int W, int C, void* stream); /* its adjoint (first maximum of a window takes the gradient): only CNNEncoder(fine_tune=False) trains below layer3 */
int ick_adaptive_avgpool_fwd(const float* x, float* y, int B, int H, int W, int C, int Ho, int Wo, void* stream); /* nn.AdaptiveAvgPool2d((7,7)) (student_model.py:34,60) on NHWC: the identity at 224x224 inputs, real pooling otherwise */
int ick_adaptive_avgpool_bwd(const float* dy, float* dx, int B, int H, int W, int C, int Ho, int Wo, void* stream);

/* ------------------------------------------------------------------ MobileNetV2 pieces of the compact student
 * torchvision mobilenet_v2().features as student_model_compact.py:19-22,51 runs it: depthwise 3x3 convolutions (weights in
 * nn.Conv2d's (C,1,3,3) layout), padding 1, stride 1 or 2, NHWC fp32; ick_colstats = the BatchNorm batch statistics of their
 * output; the 1x1 convolutions are ick_gemm_f32 NT products over pixels.  ick_scale_shift_act / ick_bn_bwd_* take act = 2 for ReLU6. */
int ick_dwconv3x3_fwd(const float* x, const float* w, float* y, int B, int H, int W, int C, int stride, void* stream);
int ick_dwconv3x3_dgrad(const float* dy, const float* w, float* dx, int B, int H, int W, int C, int stride, void* stream); /* (B,H,W) = INPUT geometry */
int ick_dwconv3x3_wgrad(const float* dy, const float* x, float* dw, int B, int H, int W, int C, int stride, void* stream); /* dw += */
int ick_colstats(const float* x, double* sum, double* sq, int64_t M, int C, void* stream);                                  /* += column sums / sums of squares (fp64) */
/* CompactLSTMDecoder.simple_attention + the additive fusion (student_model_compact.py:114-138,175): scores_j = <hp, f_j>,
 * w = softmax, x = emb + sum_j w_j f_j; adjoint: dfeats accumulated (+=), dhp stored */
int ick_dot_attn_fwd(const float* hp, const float* feats, const float* emb, float* w_out, float* x_out, int B, int L, int E, void* stream);
int ick_dot_attn_bwd(const float* dx, const float* w, const float* hp, const float* feats, float* dfeats, float* dhp, int B, int L, int E,
                     void* stream);

/* ------------------------------------------------------------------ LayerNorm / softmax / small utilities
 * nn.LayerNorm (student_model.py:41,99-100; teacher_model.py:70; timm blocks eps=1e-6; distillation_utils.py:221) */
int ick_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd,
                      int64_t rows, int D, float eps, void* stream);
int ick_layernorm_bwd(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd,
                      float* dx, float* dgamma, float* dbeta, int64_t rows, int D, void* stream); /* dgamma/dbeta accumulated (+=) */
int ick_softmax_rows(float* s, int64_t rows, int L, int ld, float scale, int causal, int Lq, void* stream);   /* in place: softmax(scale*s) */
int ick_softmax_bwd_rows(float* dp, const float* p, int64_t rows, int L, int ld, float scale, void* stream);  /* in place on dp */
int ick_colsum(const float* x, float* out, int64_t M, int N, int64_t ld, void* stream);         /* out[n] += sum_m x[m][n] (bias grads) */
int ick_relu_bwd(const float* dy, const float* y, float* dx, int64_t n, void* stream);
int ick_add(const float* a, const float* b, float* y, int64_t n, void* stream);
int ick_embedding_fwd(const int64_t* ids, const float* table, const float* pe, float* out, int64_t n, int D, int per_pos,
                      void* stream);                                                      /* nn.Embedding (+ sinusoid PE rows, teacher_model.py:25-27) */
int ick_embedding_bwd(const int64_t* ids, const float* dout, float* dtable, int64_t n, int D, void* stream);
int ick_token_pool_fwd(const float* x, float* y, int B, int L, int Lo, int D, void* stream); /* nn.AdaptiveAvgPool1d over tokens (distillation_utils.py:229,246-250) */
int ick_token_pool_bwd(const float* dy, float* dx, int B, int L, int Lo, int D, void* stream);
int ick_dropout(const float* x, float* y, int64_t n, float p, uint64_t seed, const int64_t* step, void* stream);  /* nn.Dropout; the same call on dy regenerates the mask for backward; *step (device, optional) varies the mask per replayed graph step */

/* ------------------------------------------------------------------ student decoder step (LSTM + spatial attention)
 * LSTMDecoder.attention_mechanism (student_model.py:173-203) with the time-invariant half hoisted:
 * W_a [h ; f_j] + b_a = hW[b] + Uf[b][j], Uf = feats W_f^T + b_a (one GEMM per batch), hW = h_top W_h^T (per step). */
int ick_attn_step_fwd(const float* Uf, const float* hW, const float* feats, float* w_out, float* ctx,
                      int B, int L, int E, void* stream);                                 /* scores=sum_e tanh(.), softmax_j, ctx=sum_j w_j f_j */
int ick_attn_step_bwd(const float* dctx, const float* w, const float* Uf, const float* hW, const float* feats,
                      float* dUf, float* dfeats, float* dhW, int B, int L, int E, void* stream); /* dUf, dfeats accumulated in place (+=) */
/* one nn.LSTM layer step after the two gate GEMMs (student_model.py:142-148,:244), gate order i,f,g,o */
int ick_lstm_cell_fwd(const float* G, const float* b_ih, const float* b_hh, const float* c_prev, float* gates,
                      float* c_out, float* h_out, int B, int H, void* stream);
int ick_lstm_cell_bwd(const float* dh_a, const float* dh_b, const float* dc_in, const float* gates, const float* c,
                      const float* c_prev, float* dG, float* dc_prev, int B, int H, void* stream);
int ick_argmax_rows(const float* x, int64_t* ids, int64_t rows, int V, int64_t ld, void* stream); /* greedy token (student_model.py:369) */
/* The decode step as L + 1 launches per token and direction (csrc/decoder_fused.hip) — the loop body student_model.py:232-251:
 * stage A per image: hW = W_h h_top (Wh = attention.weight[:, :H], row pitch ldwa; h_top NULL = zero state), scores / softmax /
 * context (:186-201), x = W_c2 ctx + Xe (Wc2 = attention_combine.weight[:, E:], pitch ldwc; Xe = the hoisted embedding half + bias);
 * stage L per layer: gates = [inp ; h_prev] [W_ih | W_hh]^T + b_ih + b_hh -> LSTM cell (:244), optional inter-layer dropout copy. */
int ick_dec_attn_x_fwd(const float* h_top, const float* Wh, int64_t ldwa, const float* Uf, const float* feats, const float* Wc2,
                       int64_t ldwc, const float* Xe, float* hW_out, float* w_out, float* ctx_out, float* x_out, int B, int L, int E,
                       int H, void* stream);
int ick_lstm_layer_fwd(const float* inp, int K1, const float* h_prev, const float* Wih, const float* Whh, const float* bih,
                       const float* bhh, const float* c_prev, float* gates, float* c_out, float* h_out, float* h_drop, float p_drop,
                       uint64_t seed, const int64_t* step, int B, int H, void* stream);
/* adjoints.  Stage G per layer: [carry into h(t-1) | input gradient] = dG [W_hh | W_ih], WT = that matrix transposed,
 * [(H + K1)][4H] (ick_transpose2d, once per step); carry_h_out NULL at t = 0; the input gradient is stored raw (d_inp_out: layer 0)
 * or run through the inter-layer dropout mask + the cell adjoint of the layer below (below_*; below_first = last token).
 * Stage Z per image: dctx = W_c2^T dX, attention adjoint (dUf / dfeats accumulated, dhW stored), then the TOP layer's cell adjoint for
 * token t-1 with dh = dHs_prev + top_carry_h + W_h^T dhW.  dX NULL: cell adjoint only (the last token); dHs_prev NULL: t = 0. */
int ick_lstm_layer_bwd(const float* dG, const float* WT, float* carry_h_out, float* d_inp_out, const float* below_carry_h,
                       float* below_carry_c, const float* below_gates, const float* below_c, const float* below_c_prev,
                       float* below_dG, int below_first, float p_drop, uint64_t seed, const int64_t* step, int B, int K1, int H,
                       void* stream);
int ick_dec_attn_x_bwd(const float* dX, const float* Wc2, int64_t ldwc, const float* w, const float* Uf, const float* hW,
                       const float* feats, float* dUf, float* dfeats, float* dhW_out, const float* Wh, int64_t ldwa,
                       const float* dHs_prev, const float* top_carry_h, float* top_carry_c, const float* top_gates,
                       const float* top_c, const float* top_c_prev, float* top_dG, int top_first, int B, int L, int E, int H,
                       void* stream);
int ick_transpose2d(const float* src, int64_t ld_src, float* dst, int64_t ld_dst, int rows, int cols, void* stream); /* dst[c][r] = src[r][c] */
int ick_beam_topk(const float* logits, const float* scores, int Bl, int V, int k, float* out_vals, int64_t* out_idx,
                  void* stream); /* scores[b] + log_softmax(logits[b]) -> k best flat (b*V+v) candidates (teacher_model.py:170-179) */
/* KV-cached, batched beam search (teacher_model.py:108-252 without the per-step re-run of the decoder on the whole
 * prefix): B images x W beam slots, row = b*W + j, Tcap = max_length + 1 positions.
 * ick_beam_self_attn: one new token per row; qkv [rows][3E] = packed in_proj of that token (position t): its key / value go
 *   into kcache / vcache [Tcap][rows][E], the query attends to positions 0..t where position p < t is read from row
 *   anc[p][row] (the beam's ancestor at that step: caches are never permuted).  head dim 64.
 * ick_beam_step: per image, the `width` best of score + log_softmax(logits) over its live rows (:170-179); a pair ending in
 *   end_id (< 0: none) leaves as a finished hypothesis (fin_seq [B][W][Tcap], fin_score raw, fin_len, nfin; :189-197), the
 *   others are compacted into slots 0.. in rank order with their sequence (seq_in -> seq_out [rows][Tcap]), score, ancestry
 *   (anc_in -> anc_out [Tcap][rows]) and next input token; width[b] becomes the number of survivors (:214-228). */
int ick_beam_self_attn(const float* qkv, float* kcache, float* vcache, const int32_t* anc, float* out, int rows, int E, int heads,
                       int t, int Tcap, void* stream);
int ick_beam_step(const float* logits, float* score, int32_t* width, const int32_t* seq_in, int32_t* seq_out, const int32_t* anc_in,
                  int32_t* anc_out, int64_t* next_tok, int32_t* fin_seq, float* fin_score, int32_t* fin_len, int32_t* nfin, int B,
                  int W, int V, int Tcap, int t, int end_id, void* stream);

/* ------------------------------------------------------------------ KD losses, fused forward + backward
 * DistillationLoss (distillation_utils.py:8-200).  Gradients are produced in the same pass as the loss terms,
 * already multiplied by the caller's weights; ick_kd_combine reduces deterministically into
 * out5 = {total, ce, token_kd, feature_kd, hidden_kd} (the loss_dict order, :192-198). */
int ick_count_valid(const int64_t* targets, int n, int* out, void* stream);               /* #targets != PAD(0) (CrossEntropyLoss ignore_index=0, :22) */
int ick_token_kd_ce(const float* s, const float* t, const int64_t* targets, float* ds, float* row_kl, float* row_ce,
                    const int* n_valid, int rows, int V, float tau, float g_kd, float g_ce_num, void* stream); /* :30-54 + :154 */
int ick_feature_kd(const float* s, const float* t, float* ds, float* dt, float* part, int B, int L, int E, float gscale,
                   void* stream);                                                         /* :56-94 */
int ick_hidden_kd(const float* s, const float* t, float* ds, float* part, int steps, int B, int H, float gscale,
                  void* stream);                                                          /* :96-136 */
int ick_kd_combine(const float* row_kl, const float* row_ce, int rows, const int* n_valid, const float* feat_part, int Bf,
                   int Ef, const float* hid_part, int hid_steps, int hid_B, int hid_H, float w_ce, float alpha, float beta,
                   float gamma, float tau, float* out5, void* stream);                    /* :184-189 */
int ick_scale_by_scalar(float* x, const float* scalar, int64_t n, void* stream);          /* x *= *scalar (device scalar) */
/* OptimizedDistillationLoss (train_student_kd_optimized.py:34-128): soft-target cross entropy + focal
 * loss per logits row with the gradient in the same pass (:51-56,:73-82); per-token cosine feature loss (:84-94);
 * attention-weighted hidden MSE with caller-supplied softmaxed weights (:101-110); out7 in the reference's dict order. */
int ick_token_softce_focal(const float* s, const float* t, const int64_t* targets, float* ds, float* row_kd, float* row_hard,
                           int rows, int V, float tau, float g_kd, float g_hard, float focal_alpha, float focal_gamma,
                           void* stream);
int ick_feature_cosine(const float* s, const float* t, float* ds, float* dt, float* part, int64_t rows, int E, float gscale,
                       void* stream);
int ick_weighted_hidden_mse(const float* s, const float* t, const float* w, float* ds, float* part, int T, int B, int H,
                            float gscale, void* stream);
int ick_optloss_combine(const float* row_kd, const float* row_hard, int rows, const float* cos_part, int64_t cos_rows,
                        const float* hid_part, int hid_B, int hid_H, float alpha_now, float beta_now, float gamma_now,
                        float tau, float* out7, void* stream);

/* ------------------------------------------------------------------ optimizer tail (train_student_kd.py:292-299)
 * flat fp32 buffers; clip_grad_norm_(max_norm) folded into the AdamW pass through the device-resident norm. */
int ick_grad_norm(const float* x, int64_t n, float* workspace, float* norm_out, int accumulate, void* stream);
int ick_adamw_step(float* p, float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                   float weight_decay, int step, const float* norm, float max_norm, float inv_scale, int write_clipped,
                   const float* hyper, const float* scaler, void* stream);
/* Adam's step count as device state: *applied_steps += 1 unless scaler (GradScaler state, optional) reports found_inf —
 * torch's scaler.step(optimizer) leaves state['step'] untouched on a skipped step — then hyper[g*stride + 1] = 1 - beta1^t,
 * hyper[g*stride + 2] = 1 - beta2^t for g < n_groups (the rows ick_adamw_step reads), so a replayed hipGraph needs no
 * host-computed bias correction.  lr_in (optional, n_groups contiguous floats): hyper[g*stride] = lr_in[g] — the host
 * uploads the schedule's learning rates as one contiguous pinned block and this kernel scatters them into the rows.
 * beta1_in (optional, n_groups floats; rows of >= 4 floats): this step's beta1 per group — torch's OneCycleLR cycles Adam's
 * beta1 against the learning rate (train_student_kd_optimized.py:369-378) — stored in hyper[g*stride + 3], which
 * ick_adamw_step uses instead of its beta1 argument when it is > 0, and used for 1 - beta1^t as torch.optim.AdamW does. */
int ick_adam_bias_correction(int64_t* applied_steps, const float* scaler, double beta1, double beta2, float* hyper,
                             int n_groups, int stride, const float* lr_in, const float* beta1_in, void* stream);
/* torch.amp.GradScaler (train_student_kd.py:239,288-298) as device state {scale, 1/scale, found_inf, good_steps}:
 * _check marks found_inf from the norms of the scaled gradients (unscale_), ick_adamw_step(scaler=state) unscales and
 * skips on found_inf (scaler.step), _update applies growth / backoff (scaler.update). */
int ick_loss_scale_check(const float* norms, int n_norms, float* state, void* stream);
int ick_loss_scale_update(float* state, float growth_factor, float backoff_factor, int growth_interval, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ICK_H */
