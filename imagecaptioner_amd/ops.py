"""Tensor-level launchers over the C ABI (include/ick.h).  PyTorch supplies device memory
(torch.empty), the current HIP stream and nothing else; every function here ends in a
libick.so kernel launch on that stream, so the whole step can be captured in a hipGraph.
"""
from __future__ import annotations

import ctypes
import math
import os
from typing import Optional, Tuple

import torch

from . import _lib
from ._lib import (ACT_GELU, ACT_NONE, ACT_RELU, ACT_TANH, OP_CONV_DGRAD, OP_CONV_DGRAD_S2, OP_CONV_FWD, OP_CONV_FWD_C4,
                   OP_CONV_WGRAD, OP_NN, OP_NT, OP_TN, IckGemm, check)

_F32 = torch.float32


def _st() -> int:
    if not torch.cuda.is_available():
        raise RuntimeError("the HIP path needs an MI355X (no CPU fallback exists)")
    return torch.cuda.current_stream().cuda_stream


def _chk(t: torch.Tensor, name: str = "tensor") -> torch.Tensor:
    if not t.is_cuda:
        raise RuntimeError(f"{name}: the HIP path needs a device tensor (no CPU fallback exists)")
    if t.dtype != _F32:
        raise TypeError(f"{name}: expected float32, got {t.dtype}")
    return t


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def empty(*shape, device=None) -> torch.Tensor:
    return torch.empty(*shape, dtype=_F32, device=device or "cuda")


def zeros(*shape, device=None) -> torch.Tensor:
    return torch.zeros(*shape, dtype=_F32, device=device or "cuda")


# ----------------------------------------------------------------------------- raw GEMM
_FORCE_TILE = [0]       # tools/profile_step_gemms.py sweeps tile shapes through this
_KCHUNK = [0]           # diagnostics: IckGemm.kchunk of every descriptor (0 = library default 64, -1 = one chain over K)
_TILE_OR = [0]          # diagnostics: bits OR-ed into every descriptor's tile field (512 = direct epilogue)


def _load_tuned(name: str = "tuned_tiles.json"):
    """Measured best tile per (op, M, N, K, batch, splitk) for the KD step's shapes (tools/profile_step_gemms.py on an
    MI355X); shapes not in the table use the library's wave-quantisation model."""
    import json
    import os
    path = (os.environ.get("ICK_TUNED_TABLE") if name == "tuned_tiles.json" else None) or \
        os.path.join(os.path.dirname(os.path.abspath(__file__)), name)
    if os.environ.get("ICK_NO_TUNED") == "1":      # A/B runs: the library's own tile model only
        return {}
    try:
        return {k: int(v) for k, v in json.load(open(path)).items()}
    except FileNotFoundError:
        return {}


_TUNED = _load_tuned()
_TUNED_X3 = _load_tuned("tuned_tiles_f32x3.json")       # same keys, for the f32x3 (three fp16 products) forward kernels
_TUNED_BF16 = _load_tuned("tuned_tiles_bf16.json")     # same keys, for the bf16 / bf16x3 kernel family
_TUNED_H16 = _load_tuned("tuned_tiles_h16.json")       # same keys, for the native 16-bit operand kernels (ick_gemm_h16)
_H16 = (torch.bfloat16, torch.float16)


def _h16(t: torch.Tensor) -> Optional[torch.dtype]:
    """The 16-bit storage type of an operand, None for fp32."""
    return t.dtype if t.dtype in _H16 else None


def _io16(c: torch.Tensor, residual: Optional[torch.Tensor], h16) -> int:
    bits = (1 if c.dtype in _H16 else 0) | (2 if (residual is not None and residual.dtype in _H16) else 0)
    if bits:
        assert h16 is not None and c.dtype in (h16, _F32) and (residual is None or residual.dtype in (h16, _F32)), \
            "16-bit C / residual must have the operands' 16-bit type"
    return bits

# Arithmetic of every dense contraction (Linear / attention products / convolutions): "f32" = exact fp32 MFMA (the
# parity regime), "bf16" = bf16 MFMA with fp32 accumulation (the reference's autocast regime, train_student_kd.py:263),
# "bf16x3" = split-bf16 (hi*hi + hi*lo + lo*hi, ~1e-5 of fp32).  Operands stay fp32 in HBM in all three.
# "f32x3" = fp32-grade products from three fp16 MFMAs (igemm_glds_impl.h TERMS 4: a = hi + 2^-11 lo', error against float64 equal
# to the exact-fp32 kernel's) for k-contiguous FORWARD products (Linear / convolution forward: call sites that pass x3=True to
# gemm_raw); every other launch — all gradients — runs the exact-fp32 kernel.
_PRECISIONS = {"f32": 0, "bf16": 1, "fp16": 2, "bf16x3": 3, "f32x3": 4}     # "fp16": fp16 MFMA products (the reference's autocast dtype)
_PREC = ["f32"]


def set_gemm_precision(name: str) -> None:
    if name not in _PRECISIONS:
        raise ValueError(f"unknown GEMM precision {name!r}; expected one of {sorted(_PRECISIONS)}")
    _PREC[0] = name


def gemm_precision() -> str:
    return _PREC[0]


class precision:
    """with ops.precision("bf16"): ...  — scoped set_gemm_precision."""

    def __init__(self, name: str):
        self.name, self.prev = name, None

    def __enter__(self):
        self.prev = _PREC[0]
        set_gemm_precision(self.name)
        return self

    def __exit__(self, *exc):
        _PREC[0] = self.prev
        return False


def gemm_raw(op: int, A: int, B: int, C: int, M: int, N: int, K: int, lda: int, ldb: int, ldc: int, *,
             bias: Optional[int] = None, residual: Optional[int] = None, ldr: int = 0, act: int = ACT_NONE,
             alpha: float = 1.0, batch: Tuple[int, int] = (1, 1),
             strides: Tuple[int, int, int, int, int, int] = (0, 0, 0, 0, 0, 0), splitk: int = 1,
             accumulate: bool = False, stat_sum: Optional[int] = None, stat_sq: Optional[int] = None,
             conv: Optional[Tuple[int, ...]] = None, tile: int = 0, stat_copies: int = 1, stat_stride: int = 0,
             col_scale: Optional[int] = None, kchunk: int = 0, h16: Optional[torch.dtype] = None, io16: int = 0,
             default_tile: int = 0, x3: bool = False, a_absmax: Optional[int] = None) -> None:
    """h16 = torch.bfloat16 / torch.float16: A and B hold that type in HBM (ick_gemm_h16); io16 bit 0 / 1: so do C / the
    residual.  x3: the call site states that both operands are FORWARD quantities (activations, weights: magnitudes inside
    fp16's normal range) — only such launches take the three-fp16-product kernel under precision "f32x3"; gradients, whose
    magnitudes can sit below fp16's 6e-5, take the exact-fp32 kernel there UNLESS the caller also passes a_absmax: a device
    pointer to max |A| (ops.absmax), from which the kernel takes the power-of-two scale that puts A into fp16's range (exact)."""
    d = IckGemm()
    d.A, d.B, d.C = A, B, C
    d.bias, d.residual, d.stat_sum, d.stat_sq = bias, residual, stat_sum, stat_sq
    d.stat_copies, d.stat_stride = stat_copies, stat_stride
    d.col_scale = col_scale
    d.kchunk = kchunk or _KCHUNK[0]
    d.op, d.act = op, act
    d.M, d.N, d.K = M, N, K
    d.lda, d.ldb, d.ldc, d.ldr = lda, ldb, ldc, ldr
    d.batch_outer, d.batch_inner = batch
    d.sAo, d.sAi, d.sBo, d.sBi, d.sCo, d.sCi = strides
    d.splitk, d.accumulate, d.alpha = splitk, int(accumulate), alpha
    if conv is not None:
        d.Nb, d.H, d.W, d.Cin, d.Ho, d.Wo, d.Cout, d.R, d.S, d.stride, d.pad = conv
    if h16 is not None:
        d.io16 = io16
        d.tile = tile or _FORCE_TILE[0] or _TUNED_H16.get(f"{op}:{M}:{N}:{K}:{batch[0] * batch[1]}:{splitk}", default_tile)
        check(_lib.lib().ick_gemm_h16(ctypes.byref(d), int(h16 == torch.float16), _st()), "ick_gemm_h16")
        return
    terms = _PRECISIONS[_PREC[0]]
    if terms == 4 and not (x3 and op in (OP_NT, OP_CONV_FWD, OP_TN, OP_CONV_WGRAD)):
        terms = 0
    d.a_absmax = a_absmax if terms == 4 else None
    d.io16 = io16        # (fp32 operands: only the bf16 / fp16 LDS-DMA variants can write a 16-bit C; the others refuse)
    if terms == 4:
        d.tile = tile or _FORCE_TILE[0] or _TUNED_X3.get(f"{op}:{M}:{N}:{K}:{batch[0] * batch[1]}:{splitk}", default_tile)
        check(_lib.lib().ick_gemm_bf16(ctypes.byref(d), terms, _st()), "ick_gemm_bf16")
        return
    if terms:
        d.tile = tile or _FORCE_TILE[0] or _TUNED_BF16.get(f"{op}:{M}:{N}:{K}:{batch[0] * batch[1]}:{splitk}", 0)
        check(_lib.lib().ick_gemm_bf16(ctypes.byref(d), terms, _st()), "ick_gemm_bf16")
        return
    d.tile = (tile or _FORCE_TILE[0] or _TUNED.get(f"{op}:{M}:{N}:{K}:{batch[0] * batch[1]}:{splitk}", default_tile)) | _TILE_OR[0]
    check(_lib.lib().ick_gemm_f32(ctypes.byref(d), _st()), "ick_gemm_f32")


def cast16(x: torch.Tensor, dtype=torch.bfloat16, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """fp32 -> bf16 / fp16 copy (round to nearest even) for the native 16-bit GEMM operands."""
    _chk(x, "x")
    if dtype is True or dtype is False:        # (legacy flag: fp16 yes / no)
        dtype = torch.float16 if dtype else torch.bfloat16
    assert x.is_contiguous() and x.numel() % 4 == 0 and dtype in _H16
    y = out if out is not None else torch.empty(x.shape, dtype=dtype, device=x.device)
    assert y.dtype == dtype and y.numel() == x.numel()
    check(_lib.lib().ick_cast_f32_to_16(x.data_ptr(), y.data_ptr(), x.numel(), int(dtype == torch.float16), _st()), "ick_cast_f32_to_16")
    return y


def cast32(x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """bf16 / fp16 -> fp32 copy."""
    assert x.is_cuda and x.is_contiguous() and x.numel() % 4 == 0 and x.dtype in _H16
    y = out if out is not None else torch.empty(x.shape, dtype=_F32, device=x.device)
    check(_lib.lib().ick_cast_16_to_f32(x.data_ptr(), y.data_ptr(), x.numel(), int(x.dtype == torch.float16), _st()), "ick_cast_16_to_f32")
    return y


# ----------------------------------------------------------------------------- Linear
def linear_fwd(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, act: int = ACT_NONE,
               residual: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """y = act(x @ w.T + bias) + residual ; x (..., K) contiguous, w (N, K)."""
    _chk(x, "x"); _chk(w, "w")
    K = x.shape[-1]
    N = w.shape[0]
    M = x.numel() // K
    assert x.is_contiguous() and w.is_contiguous() and w.shape[1] == K
    y = out if out is not None else empty(*x.shape[:-1], N, device=x.device)
    gemm_raw(OP_NT, x.data_ptr(), w.data_ptr(), y.data_ptr(), M, N, K, K, K, N, bias=_ptr(bias),
             residual=_ptr(residual), ldr=N, act=act, x3=True)
    return y


def linear_bwd_data(dy: torch.Tensor, w: torch.Tensor, out: Optional[torch.Tensor] = None,
                    accumulate: bool = False) -> torch.Tensor:
    """dx = dy @ w ; dy (..., N), w (N, K)."""
    N, K = w.shape
    M = dy.numel() // N
    assert dy.is_contiguous() and w.is_contiguous()
    dx = out if out is not None else empty(*dy.shape[:-1], K, device=dy.device)
    # few output tiles over a long contraction (the vocabulary head: 960 x 256 over 5000 = 60 workgroups, 140 us at 18 TF):
    # split the contraction so that the grid fills the chip; the partial products meet in fp32 atomics
    tiles = ((M + 63) // 64) * ((K + 63) // 64)
    splitk = min(16, 512 // max(tiles, 1), N // 512) if (tiles < 128 and N >= 2048) else 1
    if splitk > 1:
        if not accumulate:
            dx.zero_()
        gemm_raw(OP_NN, dy.data_ptr(), w.data_ptr(), dx.data_ptr(), M, K, N, N, K, K, splitk=splitk)
        return dx
    gemm_raw(OP_NN, dy.data_ptr(), w.data_ptr(), dx.data_ptr(), M, K, N, N, K, K, accumulate=accumulate)
    return dx


def linear_bwd_weight(dy: torch.Tensor, x: torch.Tensor, dw: torch.Tensor, splitk: int = 0) -> None:
    """dw (N, K) += dy^T @ x (accumulates into dw, like autograd's .grad)."""
    N, K = dw.shape
    M = dy.numel() // N
    assert dy.is_contiguous() and x.is_contiguous() and dw.is_contiguous()
    if splitk <= 0:
        tiles = ((N + 127) // 128) * ((K + 127) // 128)
        splitk = max(1, min(32, 512 // max(tiles, 1), M // 256))
    if splitk > 1:
        gemm_raw(OP_TN, dy.data_ptr(), x.data_ptr(), dw.data_ptr(), N, K, M, N, K, K, splitk=splitk)
    else:
        gemm_raw(OP_TN, dy.data_ptr(), x.data_ptr(), dw.data_ptr(), N, K, M, N, K, K, accumulate=True)


_dropout_step = {"ptr": None}     # optional device int64 counter mixed into every dropout seed (set by the trainer)


def set_dropout_step_counter(t: Optional[torch.Tensor]) -> None:
    _dropout_step["ptr"] = t


def dropout(x: torch.Tensor, y: torch.Tensor, p: float, seed: int) -> None:
    check(_lib.lib().ick_dropout(x.data_ptr(), y.data_ptr(), x.numel(), p, seed, _ptr(_dropout_step["ptr"]), _st()), "ick_dropout")


def colsum_into(x: torch.Tensor, out: torch.Tensor) -> None:
    """out[n] += sum over rows of x (…, N)."""
    N = x.shape[-1]
    check(_lib.lib().ick_colsum(x.data_ptr(), out.data_ptr(), x.numel() // N, N, N, _st()), "ick_colsum")


def relu_bwd(dy: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    dx = torch.empty_like(dy)
    check(_lib.lib().ick_relu_bwd(dy.data_ptr(), y.data_ptr(), dx.data_ptr(), dy.numel(), _st()), "ick_relu_bwd")
    return dx


def add(a: torch.Tensor, b: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    y = out if out is not None else torch.empty_like(a)
    check(_lib.lib().ick_add(a.data_ptr(), b.data_ptr(), y.data_ptr(), a.numel(), _st()), "ick_add")
    return y


# ----------------------------------------------------------------------------- LayerNorm
def layernorm_fwd(x: torch.Tensor, g: torch.Tensor, b: torch.Tensor, eps: float, save: bool = True):
    D = x.shape[-1]
    rows = x.numel() // D
    y = torch.empty_like(x)
    mean = empty(rows, device=x.device) if save else None
    rstd = empty(rows, device=x.device) if save else None
    check(_lib.lib().ick_layernorm_fwd(x.data_ptr(), g.data_ptr(), b.data_ptr(), y.data_ptr(), _ptr(mean), _ptr(rstd),
                                       rows, D, eps, _st()), "ick_layernorm_fwd")
    return y, mean, rstd


def layernorm_bwd(dy, x, g, mean, rstd, dg: Optional[torch.Tensor], db: Optional[torch.Tensor]) -> torch.Tensor:
    D = x.shape[-1]
    dx = torch.empty_like(x)
    check(_lib.lib().ick_layernorm_bwd(dy.data_ptr(), x.data_ptr(), g.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                       dx.data_ptr(), _ptr(dg), _ptr(db), x.numel() // D, D, _st()), "ick_layernorm_bwd")
    return dx


# ----------------------------------------------------------------------------- attention core (unfused: GEMM + softmax + GEMM)
def attention_fwd(q: torch.Tensor, qoff: int, qld: int, k: torch.Tensor, koff: int, kld: int, v: torch.Tensor, voff: int,
                  vld: int, B: int, H: int, Lq: int, Lk: int, d: int, causal: bool = False, p_drop: float = 0.0,
                  seed: int = 0, kv_batch_stride: Optional[int] = None):
    """softmax(Q K^T / sqrt(d)) V per (batch, head) [dropout on the probabilities when p_drop > 0].  q/k/v are 2-D row matrices [(B*L)][ld] in which head h of
    the operand starts at column off + h*d (packed in_proj outputs).  Returns (O [B*Lq][H*d], P [B,H,Lq,Lkp]) with
    the probability rows padded to Lkp = roundup4(Lk) (pad columns are zero)."""
    E = H * d
    Lkp = (Lk + 3) // 4 * 4
    P = empty(B, H, Lq, Lkp, device=q.device)
    O = empty(B * Lq, E, device=q.device)
    fs = 4  # bytes per float
    kbs = Lk * kld if kv_batch_stride is None else kv_batch_stride   # 0: every batch entry attends to the SAME keys
    vbs = Lk * vld if kv_batch_stride is None else kv_batch_stride
    gemm_raw(OP_NT, q.data_ptr() + qoff * fs, k.data_ptr() + koff * fs, P.data_ptr(), Lq, Lk, d, qld, kld, Lkp,
             batch=(B, H), strides=(Lq * qld, d, kbs, d, H * Lq * Lkp, Lq * Lkp))
    check(_lib.lib().ick_softmax_rows(P.data_ptr(), B * H * Lq, Lk, Lkp, 1.0 / math.sqrt(d), int(causal), Lq, _st()),
          "ick_softmax_rows")
    Pd = P
    if p_drop > 0.0:
        Pd = torch.empty_like(P)
        dropout(P, Pd, p_drop, seed)
    gemm_raw(OP_NN, Pd.data_ptr(), v.data_ptr() + voff * fs, O.data_ptr(), Lq, d, Lk, Lkp, vld, E,
             batch=(B, H), strides=(H * Lq * Lkp, Lq * Lkp, vbs, d, Lq * E, d))
    return O, P


def attention_fwd_fused(q: torch.Tensor, qoff: int, qld: int, k: torch.Tensor, koff: int, kld: int, v: torch.Tensor,
                        voff: int, vld: int, B: int, H: int, Lq: int, Lk: int, d: int, causal: bool = False,
                        kv_batch_stride: Optional[int] = None) -> torch.Tensor:
    """Forward-only attention with the score matrix kept on chip (csrc/attention.hip); same operand addressing as
    attention_fwd.  Returns O [B*Lq][H*d].  Only d == 64 is built (every attention of the teacher has d = 64)."""
    if d != 64:
        return attention_fwd(q, qoff, qld, k, koff, kld, v, voff, vld, B, H, Lq, Lk, d, causal,
                             kv_batch_stride=kv_batch_stride)[0]
    E = H * d
    O = empty(B * Lq, E, device=q.device)
    fs = 4
    kbs = Lk * kld if kv_batch_stride is None else kv_batch_stride
    vbs = Lk * vld if kv_batch_stride is None else kv_batch_stride
    # under precision "f32x3" the fp32-grade three-fp16-product kernel (forward-only callers: the teacher, eval-mode refinement)
    fn = _lib.lib().ick_attention_fwd_d64_x3 if _PREC[0] == "f32x3" else _lib.lib().ick_attention_fwd_d64
    check(fn(q.data_ptr() + qoff * fs, qld, Lq * qld, k.data_ptr() + koff * fs, kld, kbs,
             v.data_ptr() + voff * fs, vld, vbs, O.data_ptr(), E, Lq * E, B, H, Lq, Lk,
             int(causal), 1.0 / math.sqrt(d), _st()), "ick_attention_fwd_d64")
    return O


def attention_bwd(dO: torch.Tensor, P: torch.Tensor, q, qoff, qld, k, koff, kld, v, voff, vld,
                  dq, dqoff, dqld, dk, dkoff, dkld, dv, dvoff, dvld, B, H, Lq, Lk, d, p_drop: float = 0.0, seed: int = 0):
    """Backward of attention_fwd; writes dQ/dK/dV into the given row matrices at the same packed offsets."""
    E = H * d
    fs = 4
    Lkp = P.shape[-1]
    dP = torch.empty_like(P)
    sP = (H * Lq * Lkp, Lq * Lkp)
    Pd = P
    if p_drop > 0.0:                      # regenerate the dropped probabilities (counter-based mask)
        Pd = torch.empty_like(P)
        dropout(P, Pd, p_drop, seed)
    # dP = dO V^T
    gemm_raw(OP_NT, dO.data_ptr(), v.data_ptr() + voff * fs, dP.data_ptr(), Lq, Lk, d, E, vld, Lkp,
             batch=(B, H), strides=(Lq * E, d, Lk * vld, d) + sP)
    # dV = P^T dO
    gemm_raw(OP_TN, Pd.data_ptr(), dO.data_ptr(), dv.data_ptr() + dvoff * fs, Lk, d, Lq, Lkp, E, dvld,
             batch=(B, H), strides=sP + (Lq * E, d, Lk * dvld, d))
    if p_drop > 0.0:
        dropout(dP, dP, p_drop, seed)     # dP <- mask * dP / (1-p)
    # dS = scale * P * (dP - rowsum(dP*P))
    check(_lib.lib().ick_softmax_bwd_rows(dP.data_ptr(), P.data_ptr(), B * H * Lq, Lk, Lkp, 1.0 / math.sqrt(d), _st()),
          "ick_softmax_bwd_rows")
    # dQ = dS K
    gemm_raw(OP_NN, dP.data_ptr(), k.data_ptr() + koff * fs, dq.data_ptr() + dqoff * fs, Lq, d, Lk, Lkp, kld, dqld,
             batch=(B, H), strides=sP + (Lk * kld, d, Lq * dqld, d))
    # dK = dS^T Q
    gemm_raw(OP_TN, dP.data_ptr(), q.data_ptr() + qoff * fs, dk.data_ptr() + dkoff * fs, Lk, d, Lq, Lkp, qld, dkld,
             batch=(B, H), strides=sP + (Lq * qld, d, Lk * dkld, d))


# ----------------------------------------------------------------------------- convolution (NHWC implicit GEMM)
_DGRAD_AS_FWD = [True]     # A/B switch (tools): False = the CONV_DGRAD gather kernel for stride-1 data gradients too
def conv_out_hw(H: int, W: int, R: int, S: int, stride: int, pad: int) -> Tuple[int, int]:
    return (H + 2 * pad - R) // stride + 1, (W + 2 * pad - S) // stride + 1


_CONV_SPLITK = [os.environ.get("ICK_CONV_SPLITK", "1") != "0"]


def _conv_splitk(M: int, N: int, K: int) -> int:
    """Split factor for a forward-type convolution whose grid is too small for the chip (layer4: M = 3136 pixels x 512
    channels = 196 tiles of 64 x 128 on 256 CUs) over a long contraction.  Measured (tools/bench_conv_splitk.py, fp32):
    3136x512x4608 216.7 -> 166.2 us at 6 splits, 3136x512x2048 102.3 -> 88.2 us; 392 tiles (12544x256) gain < 4 %: not split."""
    if not _CONV_SPLITK[0] or K < 2048 or M < 1024:        # (tiny batches: the kernel is short anyway; keep it deterministic)
        return 1
    tiles = ((M + 63) // 64) * ((N + 127) // 128)
    return 6 if tiles <= 256 else 1


def conv_fwd(x: torch.Tensor, w: torch.Tensor, stride: int, pad: int, stats: Optional[Tuple[torch.Tensor, torch.Tensor]] = None,
             scale: Optional[torch.Tensor] = None, shift: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None,
             relu: bool = False, out_dtype: Optional[torch.dtype] = None) -> torch.Tensor:
    """x (Nb,H,W,Cin) physical NHWC contiguous; w physical (Cout,R,S,Cin); returns raw y (Nb,Ho,Wo,Cout) and
    optionally accumulates per-channel sum / sum of squares (BatchNorm batch statistics) into `stats`:
    (sum, sq) fp64 [Cout] each, or fp64 [Cout] x R copies each as rows of a (R, Cout) tensor (see stat_copies()).
    bf16 / fp16 x and w: native 16-bit operands (ick_gemm_h16), y in out_dtype (default: the operands' type)."""
    Nb, H, W, Cin = x.shape
    Cout, R, S, Cin2 = w.shape
    assert Cin == Cin2 and x.is_contiguous() and w.is_contiguous() and x.dtype == w.dtype
    Ho, Wo = conv_out_hw(H, W, R, S, stride, pad)
    h16 = _h16(x)
    y = torch.empty(Nb, Ho, Wo, Cout, dtype=out_dtype or x.dtype, device=x.device)
    op = OP_CONV_FWD_C4 if Cin == 4 else OP_CONV_FWD
    K = R * S * Cin
    if h16 is not None or y.dtype != _F32:
        # (fp32 operands + 16-bit y: the 4-channel stem under ops.precision("bf16" / "fp16") — ick_gemm_bf16 with io16)
        fused = scale is not None      # eval mode: y = relu(scale * conv + shift + residual) in the epilogue, as on fp32 storage
        assert not (fused and stats is not None)
        gemm_raw(op, x.data_ptr(), w.data_ptr(), y.data_ptr(), Nb * Ho * Wo, Cout, K, K, K, Cout,
                 residual=_ptr(residual), ldr=Cout,
                 act=((ACT_RELU if relu else ACT_NONE) | _lib.ACT_POST_RESIDUAL) if fused else (ACT_RELU if relu else ACT_NONE),
                 bias=shift.data_ptr() if fused else None, col_scale=scale.data_ptr() if fused else None,
                 stat_sum=_ptr(stats[0]) if stats is not None else None, stat_sq=_ptr(stats[1]) if stats is not None else None,
                 stat_copies=stats[0].shape[0] if (stats is not None and stats[0].dim() == 2) else 1, stat_stride=Cout,
                 conv=(Nb, H, W, Cin, Ho, Wo, Cout, R, S, stride, pad), h16=h16,
                 io16=_io16(y, residual, h16 or y.dtype))
        return y
    if scale is not None:       # eval-mode conv + BatchNorm (+ residual) (+ ReLU) in ONE kernel: y = relu(scale*conv + shift + res)
        gemm_raw(op, x.data_ptr(), w.data_ptr(), y.data_ptr(), Nb * Ho * Wo, Cout, K, K, K, Cout, bias=shift.data_ptr(),
                 col_scale=scale.data_ptr(), residual=_ptr(residual), ldr=Cout,
                 act=(ACT_RELU if relu else ACT_NONE) | _lib.ACT_POST_RESIDUAL,
                 conv=(Nb, H, W, Cin, Ho, Wo, Cout, R, S, stride, pad), x3=True)
        return y
    sk = _conv_splitk(Nb * Ho * Wo, Cout, K) if (stats is not None and op == OP_CONV_FWD and _PREC[0] in ("f32", "f32x3")) else 1
    if sk > 1:
        # train-mode forward on a small grid: split the contraction (fp32 atomics into the zeroed output), then take the
        # BatchNorm statistics from the finished output in one pass over its 6 MB (the epilogue cannot: it sees partial sums)
        y.zero_()
        gemm_raw(op, x.data_ptr(), w.data_ptr(), y.data_ptr(), Nb * Ho * Wo, Cout, K, K, K, Cout, splitk=sk,
                 default_tile=4 if K >= 4096 else 2, conv=(Nb, H, W, Cin, Ho, Wo, Cout, R, S, stride, pad), x3=True)
        s0, s1 = (stats[0][0], stats[1][0]) if stats[0].dim() == 2 else (stats[0], stats[1])
        check(_lib.lib().ick_colstats(y.data_ptr(), s0.data_ptr(), s1.data_ptr(), Nb * Ho * Wo, Cout, _st()), "ick_colstats")
        return y
    gemm_raw(op, x.data_ptr(), w.data_ptr(), y.data_ptr(), Nb * Ho * Wo, Cout, K, K, K, Cout,
             stat_sum=_ptr(stats[0]) if stats is not None else None, stat_sq=_ptr(stats[1]) if stats is not None else None,
             stat_copies=stats[0].shape[0] if (stats is not None and stats[0].dim() == 2) else 1, stat_stride=Cout,
             conv=(Nb, H, W, Cin, Ho, Wo, Cout, R, S, stride, pad), x3=True)
    return y


def stat_copies(rows: int) -> int:
    """Number of accumulator copies for the BatchNorm batch statistics of a conv with `rows` output pixels: every
    128-row tile of the grid adds 2 x Cout fp64 atomics to the SAME few cache lines; beyond a few hundred tiles that
    serialises in L2 (measured: 37-52 % of a 56x56 conv's time).  Copies spread the adds; consumers fold them."""
    tiles = rows // 128
    return 1 if tiles < 64 else min(8, max(2, tiles // 32))


def conv_dgrad(dy: torch.Tensor, w: torch.Tensor, in_hw: Tuple[int, int], stride: int, pad: int,
               residual: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None,
               accumulate: bool = False, wt_cached: Optional[torch.Tensor] = None) -> torch.Tensor:
    """dx (Nb,H,W,Cin) = conv-transpose of dy (Nb,Ho,Wo,Cout) with w (Cout,R,S,Cin) [+ residual].  bf16 / fp16 dy and w:
    native 16-bit operands, dx (and the residual) in that type."""
    Nb, Ho, Wo, Cout = dy.shape
    _, R, S, Cin = w.shape
    H, W = in_hw
    assert dy.is_contiguous() and w.is_contiguous() and dy.dtype == w.dtype
    h16 = _h16(dy)
    dx = out if out is not None else torch.empty(Nb, H, W, Cin, dtype=dy.dtype, device=dy.device)
    io = dict(h16=h16, io16=_io16(dx, residual, h16)) if h16 is not None else {}
    K = R * S * Cout
    if stride == 2 and H % 2 == 0 and W % 2 == 0:
        # 4 parity classes of input pixels, each a dense GEMM over the taps that can reach it (4x fewer MACs)
        gemm_raw(OP_CONV_DGRAD_S2, dy.data_ptr(), w.data_ptr(), dx.data_ptr(), Nb * (H // 2) * (W // 2), Cin, K, 0, 0, Cin,
                 residual=_ptr(residual), ldr=Cin, accumulate=accumulate,
                 conv=(Nb, H, W, Cin, Ho, Wo, Cout, R, S, stride, pad), **io)
        return dx
    if stride == 1 and _DGRAD_AS_FWD[0] and Cout % 32 == 0 and Cin % 4 == 0 and R - 1 - pad >= 0:
        # stride-1 data gradient = forward convolution over dY with flipped, channel-swapped weights (one small
        # transpose of the weight per call = per step): both GEMM operands are then k-contiguous, i.e. they take the
        # LDS-DMA kernel's ds_read_b128 path instead of the strided [k][n] weight view
        wt = wt_cached if wt_cached is not None else conv_weight_dgrad_layout(w)
        sk = _conv_splitk(Nb * H * W, Cin, K) if (h16 is None and _PREC[0] == "f32") else 1
        if sk > 1:        # small grid, long contraction (layer4): split-K; the residual rides on split 0, `accumulate` = add onto dx
            if not accumulate:
                dx.zero_()
            gemm_raw(OP_CONV_FWD, dy.data_ptr(), wt.data_ptr(), dx.data_ptr(), Nb * H * W, Cin, K, K, K, Cin,
                     residual=_ptr(residual), ldr=Cin, splitk=sk, default_tile=4 if K >= 4096 else 2,
                     conv=(Nb, Ho, Wo, Cout, H, W, Cin, R, S, 1, R - 1 - pad))
            return dx
        am = None
        if _PREC[0] == "f32x3" and h16 is None and _X3_DGRAD[0] and dy.numel() % 4 == 0:
            # the gradient's magnitude is anything (1e-6 and below in this trunk): its maximum, taken on the device, gives the
            # kernel the exact power-of-two scale that puts it into fp16's range — the three-product kernel then serves data
            # gradients as it serves forward activations (IckGemm.a_absmax)
            am = _grad_absmax(dy)
        gemm_raw(OP_CONV_FWD, dy.data_ptr(), wt.data_ptr(), dx.data_ptr(), Nb * H * W, Cin, K, K, K, Cin,
                 residual=_ptr(residual), ldr=Cin, accumulate=accumulate,
                 conv=(Nb, Ho, Wo, Cout, H, W, Cin, R, S, 1, R - 1 - pad), x3=am is not None, a_absmax=_ptr(am), **io)
        return dx
    gemm_raw(OP_CONV_DGRAD, dy.data_ptr(), w.data_ptr(), dx.data_ptr(), Nb * H * W, Cin, K, 0, 0, Cin,
             residual=_ptr(residual), ldr=Cin, accumulate=accumulate,
             conv=(Nb, H, W, Cin, Ho, Wo, Cout, R, S, stride, pad), **io)
    return dx


def absmax(x: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
    """out[0] = max(out[0], max |x|) on the device (out: one non-negative float, 0 for a fresh maximum); x.numel() % 4 == 0."""
    check(_lib.lib().ick_absmax_f32(_chk(x, "x").data_ptr(), x.numel(), out.data_ptr(), _st()), "ick_absmax_f32")
    return out


_ABSMAX_ARENA = {"buf": None, "next": 0, "cache": {}}


def begin_absmax_arena(n: int = 64, device="cuda") -> None:
    """one zero fill for the next n absmax slots (a backward pass opens this; without it every slot is its own torch.zeros)"""
    _ABSMAX_ARENA["buf"], _ABSMAX_ARENA["next"], _ABSMAX_ARENA["cache"] = torch.zeros(n, dtype=_F32, device=device), 0, {}


def end_absmax_arena() -> None:
    _ABSMAX_ARENA["buf"], _ABSMAX_ARENA["next"], _ABSMAX_ARENA["cache"] = None, 0, {}


def _grad_absmax(dy: torch.Tensor) -> torch.Tensor:
    """max |dy| on the device; inside an arena (one backward pass) the data gradient and the weight gradient of a convolution,
    which read the same dy, share one pass"""
    a = _ABSMAX_ARENA
    key = (dy.data_ptr(), dy.numel(), dy._version)
    if a["buf"] is not None and key in a["cache"]:
        return a["cache"][key][0]
    am = absmax(dy, _absmax_slot(dy.device))
    if a["buf"] is not None:
        a["cache"][key] = (am, dy)              # (dy kept alive: its address must not be handed to another tensor meanwhile)
    return am


def _absmax_slot(device) -> torch.Tensor:
    a = _ABSMAX_ARENA
    if a["buf"] is not None and a["next"] < a["buf"].numel() and a["buf"].device == device:
        a["next"] += 1
        return a["buf"][a["next"] - 1:a["next"]]
    return torch.zeros(1, dtype=_F32, device=device)


_X3_DGRAD = [os.environ.get("ICK_X3_DGRAD", "1") != "0"]     # precision "f32x3": stride-1 data gradients on the three-product kernel too


def conv_weight_dgrad_layout(w: torch.Tensor) -> torch.Tensor:
    """wt[ci][R-1-r][S-1-s][co] = w[co][r][s][ci] (fp32, bf16 or fp16)."""
    Cout, R, S, Cin = w.shape
    wt = torch.empty(Cin, R, S, Cout, dtype=w.dtype, device=w.device)
    if w.dtype == _F32:
        check(_lib.lib().ick_conv_weight_dgrad_layout(w.data_ptr(), wt.data_ptr(), Cout, R, S, Cin, _st()), "ick_conv_weight_dgrad_layout")
    else:
        check(_lib.lib().ick_conv_weight_dgrad_layout16(w.data_ptr(), wt.data_ptr(), Cout, R, S, Cin, _st()), "ick_conv_weight_dgrad_layout16")
    return wt


_WGRAD_1X1_AS_TN = [os.environ.get("ICK_WGRAD_1X1_AS_TN", "1") != "0"]     # A/B switch


def conv_wgrad(dy: torch.Tensor, x: torch.Tensor, dw: torch.Tensor, stride: int, pad: int, splitk: int = 0) -> None:
    """dw (Cout,R,S,Cin) += sum over output pixels of dy (x) gathered x (accumulates, fp32 atomics when split); dw is fp32,
    dy and x fp32 or both bf16 / fp16 (native 16-bit operands)."""
    Nb, Ho, Wo, Cout = dy.shape
    _, H, W, Cin = x.shape
    _, R, S, _ = dw.shape
    assert dy.is_contiguous() and x.is_contiguous() and dw.is_contiguous() and dy.dtype == x.dtype and dw.dtype == _F32
    io = dict(h16=_h16(dy), default_tile=67) if _h16(dy) is not None else {}     # 128x64, eight waves: best or within 3 % on every shape measured
    K = Nb * Ho * Wo
    N = R * S * Cin
    if R == 1 and S == 1 and stride == 1 and pad == 0 and not io and _WGRAD_1X1_AS_TN[0]:
        # a 1x1 / stride-1 convolution's im2col is the identity: dW[co][ci] = sum_pix dY[pix][co] X[pix][ci] is a plain TN GEMM,
        # which the LDS-DMA kernel runs 10-30 % faster than the register-staged CONV_WGRAD gather kernel (tools/bench_wgrad_tn.py,
        # profiles/r03_wgrad_1x1_as_tn.log: 1024x256x12544 68.7 us against 84; 128x512x50176 74 against 106).  ~800 pixels per
        # split (8 <= splits <= 48); the tile comes from the tuned table (128x64 otherwise)
        sk = splitk if splitk > 0 else (max(8, min(48, K // 784)) if K >= 3136 else max(1, K // 392))
        x3kw = {}
        if _PREC[0] == "f32x3" and _X3_DGRAD[0] and dy.numel() % 4 == 0 and sk > 1:
            # three-product kernel for the weight gradient too: A = dY carries the scale of its absmax, B = X is a forward
            # activation; the split keeps every accumulator chain to ~800 pixels
            x3kw = dict(x3=True, a_absmax=_grad_absmax(dy).data_ptr())
        if sk > 1:
            gemm_raw(OP_TN, dy.data_ptr(), x.data_ptr(), dw.data_ptr(), Cout, Cin, K, Cout, Cin, Cin, splitk=sk, default_tile=3, **x3kw)
        else:
            gemm_raw(OP_TN, dy.data_ptr(), x.data_ptr(), dw.data_ptr(), Cout, Cin, K, Cout, Cin, Cin, accumulate=True, default_tile=3)
        return
    if splitk <= 0 and io:
        # native 16-bit operands on the LDS-DMA kernel (round 3, tools/bench_wgrad16.py, profiles/r03_wgrad16_lds_dma.log): the
        # grid wants ~560 workgroups of 128 x 64 (256x2304x12544: 42 us = 350 TF at 8 splits against 53 at 16 and 75 on the
        # register-staged kernel; 512x4608x3136: 44 us at 2 splits), never fewer than ~400 pixels per split
        tiles = ((Cout + 127) // 128) * ((N + 63) // 64)
        splitk = max(1, min(64, K // 392, round(560 / max(tiles, 1))))
    if splitk <= 0:
        tiles = ((Cout + 127) // 128) * ((N + 127) // 128)
        splitk = max(1, min(64, 768 // max(tiles, 1), K // 128))
    if splitk > 1:
        if not io and _PREC[0] == "f32x3" and _X3_DGRAD[0] and dy.numel() % 4 == 0 and os.environ.get("ICK_X3_WGRAD3", "1") != "0":
            io = dict(x3=True, a_absmax=_grad_absmax(dy).data_ptr())     # 3x3 / strided weight gradients on the three-product kernel too
        gemm_raw(OP_CONV_WGRAD, dy.data_ptr(), x.data_ptr(), dw.data_ptr(), Cout, N, K, Cout, 0, N, splitk=splitk,
                 conv=(Nb, H, W, Cin, Ho, Wo, Cout, R, S, stride, pad), **io)
    else:
        gemm_raw(OP_CONV_WGRAD, dy.data_ptr(), x.data_ptr(), dw.data_ptr(), Cout, N, K, Cout, 0, N, accumulate=True,
                 conv=(Nb, H, W, Cin, Ho, Wo, Cout, R, S, stride, pad), **io)


# ----------------------------------------------------------------------------- BatchNorm pieces
def bn_finalize(ssum, ssq, count, gamma, beta, rmean, rvar, momentum, eps):
    C = gamma.numel()
    co = empty(4, C, device=gamma.device)  # scale, shift, mean, invstd
    copies = ssum.shape[0] if ssum.dim() == 2 else 1
    check(_lib.lib().ick_bn_finalize(ssum.data_ptr(), ssq.data_ptr(), copies, C, float(count), gamma.data_ptr(), beta.data_ptr(),
                                     _ptr(rmean), _ptr(rvar), momentum, eps, co[0].data_ptr(), co[1].data_ptr(),
                                     co[2].data_ptr(), co[3].data_ptr(), C, _st()), "ick_bn_finalize")
    return co


def bn_eval_coeffs(gamma, beta, rmean, rvar, eps):
    C = gamma.numel()
    co = empty(2, C, device=gamma.device)
    check(_lib.lib().ick_bn_eval_coeffs(gamma.data_ptr(), beta.data_ptr(), rmean.data_ptr(), rvar.data_ptr(), eps,
                                        co[0].data_ptr(), co[1].data_ptr(), C, _st()), "ick_bn_eval_coeffs")
    return co


def scale_shift_act(x, scale, shift, residual, relu, out=None):
    """relu: False / True, or the activation code 2 = ReLU6."""
    C = x.shape[-1]
    y = out if out is not None else torch.empty_like(x)
    check(_lib.lib().ick_scale_shift_act(x.data_ptr(), scale.data_ptr(), shift.data_ptr(), _ptr(residual), y.data_ptr(),
                                         x.numel() // C, C, int(relu), _st()), "ick_scale_shift_act")
    return y


BN_BWD_COPIES = int(__import__("os").environ.get("ICK_BN_BWD_COPIES", "8"))      # accumulator rows of the BatchNorm-backward sums (spreads same-address fp64 atomics)


def bn_bwd(dy, y_mask, x, mean, invstd, gamma, dgamma, dbeta, want_g: bool, batch_stats: bool = True, sums=None, act: int = 1):
    """Backward through [relu](bn(x)[+res]).  dy: grad wrt the block output; y_mask: that output (relu mask) or None.
    Accumulates dgamma/dbeta (+=) inside the apply kernel; returns (dx, g) with g = masked dy (gradient of the
    residual branch) if want_g.  `sums`: a zeroed (2, R, C) fp64 slice of a caller-owned arena (saves one fill per BN)."""
    C = x.shape[-1]
    M = x.numel() // C
    if sums is None:
        sums = torch.zeros(2, BN_BWD_COPIES, C, dtype=torch.float64, device=x.device)
    R = sums.shape[1] if sums.dim() == 3 else 1
    dx = torch.empty_like(x)
    g = torch.empty_like(x) if want_g else None
    coef = empty(2, C, device=x.device)
    if x.dtype in _H16:        # 16-bit activation storage (dy, y, x, dx, g all of one 16-bit type)
        assert dy.dtype == x.dtype and (y_mask is None or y_mask.dtype == x.dtype)
        f16 = int(x.dtype == torch.float16)
        check(_lib.lib().ick_bn_bwd_reduce16(dy.data_ptr(), _ptr(y_mask), x.data_ptr(), mean.data_ptr(), invstd.data_ptr(),
                                             sums[0].data_ptr(), sums[1].data_ptr(), R, C, M, C, act, f16, _st()), "ick_bn_bwd_reduce16")
        check(_lib.lib().ick_bn_bwd_apply16(dy.data_ptr(), _ptr(y_mask), x.data_ptr(), mean.data_ptr(), invstd.data_ptr(),
                                            gamma.data_ptr(), sums[0].data_ptr(), sums[1].data_ptr(), R, C, coef.data_ptr(),
                                            dx.data_ptr(), _ptr(g), M, C, int(batch_stats), _ptr(dgamma), _ptr(dbeta), act, f16, _st()),
              "ick_bn_bwd_apply16")
        return dx, g
    check(_lib.lib().ick_bn_bwd_reduce(dy.data_ptr(), _ptr(y_mask), x.data_ptr(), mean.data_ptr(), invstd.data_ptr(),
                                       sums[0].data_ptr(), sums[1].data_ptr(), R, C, M, C, act, _st()), "ick_bn_bwd_reduce")
    check(_lib.lib().ick_bn_bwd_apply(dy.data_ptr(), _ptr(y_mask), x.data_ptr(), mean.data_ptr(), invstd.data_ptr(),
                                      gamma.data_ptr(), sums[0].data_ptr(), sums[1].data_ptr(), R, C, coef.data_ptr(),
                                      dx.data_ptr(), _ptr(g),
                                      M, C, int(batch_stats), _ptr(dgamma), _ptr(dbeta), act, _st()), "ick_bn_bwd_apply")
    return dx, g


def bn_train_apply(raw, stats, gamma, beta, rmean, rvar, momentum, eps, residual, relu: bool, inplace: bool = False):
    """y = [relu](batchnorm_train(raw) [+ residual]) from the fp64 sums of the conv epilogue, one pass; returns
    (y, save_mean, save_invstd) and updates the running statistics.  inplace: y overwrites raw (layers without a backward:
    the raw convolution output is needed by the BatchNorm adjoint only; the pass is elementwise, every thread reads the
    element it writes) — no second activation-sized allocation, and the stores land on lines the read just brought in."""
    C = raw.shape[-1]
    y = raw if inplace else torch.empty_like(raw)
    sv = empty(2, C, device=raw.device)
    copies = stats[0].shape[0] if stats[0].dim() == 2 else 1
    if raw.dtype in _H16:      # 16-bit activation storage
        assert residual is None or residual.dtype == raw.dtype
        check(_lib.lib().ick_bn_train_apply16(raw.data_ptr(), stats[0].data_ptr(), stats[1].data_ptr(), copies, C, gamma.data_ptr(),
                                              beta.data_ptr(), _ptr(rmean), _ptr(rvar), momentum, eps, _ptr(residual), y.data_ptr(),
                                              sv[0].data_ptr(), sv[1].data_ptr(), raw.numel() // C, C, int(relu),
                                              int(raw.dtype == torch.float16), _st()), "ick_bn_train_apply16")
        return y, sv[0], sv[1]
    check(_lib.lib().ick_bn_train_apply(raw.data_ptr(), stats[0].data_ptr(), stats[1].data_ptr(), copies, C, gamma.data_ptr(),
                                        beta.data_ptr(), _ptr(rmean), _ptr(rvar), momentum, eps, _ptr(residual), y.data_ptr(),
                                        sv[0].data_ptr(), sv[1].data_ptr(), raw.numel() // C, C, int(relu), _st()),
          "ick_bn_train_apply")
    return y, sv[0], sv[1]


def dwconv3x3_fwd(x, w, stride: int):
    """depthwise 3x3, padding 1: x (B,H,W,C) NHWC, w (C,1,3,3) contiguous."""
    B, H, W, C = x.shape
    assert w.is_contiguous() and tuple(w.shape) == (C, 1, 3, 3)
    Ho, Wo = (H + 2 - 3) // stride + 1, (W + 2 - 3) // stride + 1
    y = empty(B, Ho, Wo, C, device=x.device)
    check(_lib.lib().ick_dwconv3x3_fwd(x.data_ptr(), w.data_ptr(), y.data_ptr(), B, H, W, C, stride, _st()), "ick_dwconv3x3_fwd")
    return y


def dwconv3x3_dgrad(dy, w, in_hw, stride: int):
    B, Ho, Wo, C = dy.shape
    H, W = in_hw
    dx = empty(B, H, W, C, device=dy.device)
    check(_lib.lib().ick_dwconv3x3_dgrad(dy.data_ptr(), w.data_ptr(), dx.data_ptr(), B, H, W, C, stride, _st()), "ick_dwconv3x3_dgrad")
    return dx


def dwconv3x3_wgrad(dy, x, dw, stride: int) -> None:
    B, H, W, C = x.shape
    assert dw.is_contiguous()
    check(_lib.lib().ick_dwconv3x3_wgrad(dy.data_ptr(), x.data_ptr(), dw.data_ptr(), B, H, W, C, stride, _st()), "ick_dwconv3x3_wgrad")


def colstats(x, stats) -> None:
    """stats (2, C) fp64 += per-channel sum / sum of squares of x (..., C)."""
    C = x.shape[-1]
    check(_lib.lib().ick_colstats(x.data_ptr(), stats[0].data_ptr(), stats[1].data_ptr(), x.numel() // C, C, _st()), "ick_colstats")


def dot_attn_fwd(hp, feats, emb_t, w_t, x_t) -> None:
    B, L, E = feats.shape
    check(_lib.lib().ick_dot_attn_fwd(hp.data_ptr(), feats.data_ptr(), emb_t.data_ptr(), w_t.data_ptr(), x_t.data_ptr(), B, L, E, _st()),
          "ick_dot_attn_fwd")


def dot_attn_bwd(dx_t, w_t, hp, feats, dfeats, dhp) -> None:
    B, L, E = feats.shape
    check(_lib.lib().ick_dot_attn_bwd(dx_t.data_ptr(), w_t.data_ptr(), hp.data_ptr(), feats.data_ptr(), dfeats.data_ptr(),
                                      dhp.data_ptr(), B, L, E, _st()), "ick_dot_attn_bwd")


def maxpool3x3s2(x):
    Nb, H, W, C = x.shape
    Ho, Wo = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
    y = torch.empty(Nb, Ho, Wo, C, dtype=x.dtype, device=x.device)
    if x.dtype in _H16:
        check(_lib.lib().ick_maxpool3x3s2_16(x.data_ptr(), y.data_ptr(), Nb, H, W, C, int(x.dtype == torch.float16), _st()), "ick_maxpool3x3s2_16")
        return y
    check(_lib.lib().ick_maxpool3x3s2(x.data_ptr(), y.data_ptr(), Nb, H, W, C, _st()), "ick_maxpool3x3s2")
    return y


def nhwc4_to_nhwc3_add(src4: torch.Tensor, dst3: torch.Tensor) -> None:
    """dst3[..., c] += src4[..., c] for c < 3 (gradient of the RGB stem weight out of its zero-padded 4-channel form);
    dst3 is the physical [Cout][R][S][3] view of a channels_last (Cout,3,R,S) tensor."""
    n = src4.numel() // 4
    assert dst3.numel() == 3 * n and src4.is_contiguous()
    check(_lib.lib().ick_nhwc4_to_nhwc3_add(src4.data_ptr(), dst3.data_ptr(), n, _st()), "ick_nhwc4_to_nhwc3_add")


def maxpool3x3s2_bwd(x, dy):
    Nb, H, W, C = x.shape
    dx = torch.empty_like(x)
    check(_lib.lib().ick_maxpool3x3s2_bwd(x.data_ptr(), dy.data_ptr(), dx.data_ptr(), Nb, H, W, C, _st()), "ick_maxpool3x3s2_bwd")
    return dx


def adaptive_avgpool_fwd(x, Ho: int, Wo: int):
    Nb, H, W, C = x.shape
    y = empty(Nb, Ho, Wo, C, device=x.device)
    check(_lib.lib().ick_adaptive_avgpool_fwd(x.data_ptr(), y.data_ptr(), Nb, H, W, C, Ho, Wo, _st()), "ick_adaptive_avgpool_fwd")
    return y


def adaptive_avgpool_bwd(dy, H: int, W: int):
    Nb, Ho, Wo, C = dy.shape
    dx = empty(Nb, H, W, C, device=dy.device)
    check(_lib.lib().ick_adaptive_avgpool_bwd(dy.data_ptr(), dx.data_ptr(), Nb, H, W, C, Ho, Wo, _st()), "ick_adaptive_avgpool_bwd")
    return dx


def nchw3_to_nhwc4(images):
    B, C, H, W = images.shape
    assert C == 3 and images.is_contiguous()
    y = empty(B, H, W, 4, device=images.device)
    check(_lib.lib().ick_nchw3_to_nhwc4(images.data_ptr(), y.data_ptr(), B, H, W, _st()), "ick_nchw3_to_nhwc4")
    return y


# ----------------------------------------------------------------------------- ViT helpers / embedding / pooling
def patchify16(images):
    B, C, H, W = images.shape
    assert C == 3 and H == W and images.is_contiguous()
    G = H // 16
    y = empty(B * G * G, 768, device=images.device)
    check(_lib.lib().ick_patchify16(images.data_ptr(), y.data_ptr(), B, H, _st()), "ick_patchify16")
    return y


def vit_assemble(patch, cls, pos, B, Ntok, D):
    x = empty(B, Ntok, D, device=patch.device)
    check(_lib.lib().ick_vit_assemble(patch.data_ptr(), cls.data_ptr(), pos.data_ptr(), x.data_ptr(), B, Ntok, D, _st()),
          "ick_vit_assemble")
    return x


def embedding_fwd(ids: torch.Tensor, table: torch.Tensor, pe: Optional[torch.Tensor] = None, per_pos: int = 1):
    assert ids.dtype == torch.int64 and ids.is_contiguous() and ids.is_cuda
    D = table.shape[1]
    out = empty(*ids.shape, D, device=table.device)
    check(_lib.lib().ick_embedding_fwd(ids.data_ptr(), table.data_ptr(), _ptr(pe), out.data_ptr(), ids.numel(), D, per_pos,
                                       _st()), "ick_embedding_fwd")
    return out


def embedding_bwd(ids, dout, dtable):
    D = dtable.shape[1]
    check(_lib.lib().ick_embedding_bwd(ids.data_ptr(), dout.data_ptr(), dtable.data_ptr(), ids.numel(), D, _st()),
          "ick_embedding_bwd")


def token_pool_fwd(x, Lo):
    B, L, D = x.shape
    y = empty(B, Lo, D, device=x.device)
    check(_lib.lib().ick_token_pool_fwd(x.data_ptr(), y.data_ptr(), B, L, Lo, D, _st()), "ick_token_pool_fwd")
    return y


def token_pool_bwd(dy, L):
    B, Lo, D = dy.shape
    dx = empty(B, L, D, device=dy.device)
    check(_lib.lib().ick_token_pool_bwd(dy.data_ptr(), dx.data_ptr(), B, L, Lo, D, _st()), "ick_token_pool_bwd")
    return dx


# ----------------------------------------------------------------------------- decoder step kernels
def attn_step_fwd(Uf, hW, feats, w_out, ctx_out):
    B, L, E = feats.shape
    check(_lib.lib().ick_attn_step_fwd(Uf.data_ptr(), hW.data_ptr(), feats.data_ptr(), w_out.data_ptr(), ctx_out.data_ptr(),
                                       B, L, E, _st()), "ick_attn_step_fwd")


def attn_step_bwd(dctx, w, Uf, hW, feats, dUf, dfeats, dhW):
    B, L, E = feats.shape
    check(_lib.lib().ick_attn_step_bwd(dctx.data_ptr(), w.data_ptr(), Uf.data_ptr(), hW.data_ptr(), feats.data_ptr(),
                                       dUf.data_ptr(), dfeats.data_ptr(), dhW.data_ptr(), B, L, E, _st()), "ick_attn_step_bwd")


def lstm_cell_fwd(G, b_ih, b_hh, c_prev, gates, c_out, h_out):
    B, H = c_out.shape
    check(_lib.lib().ick_lstm_cell_fwd(G.data_ptr(), b_ih.data_ptr(), b_hh.data_ptr(), _ptr(c_prev), _ptr(gates),
                                       c_out.data_ptr(), h_out.data_ptr(), B, H, _st()), "ick_lstm_cell_fwd")


def lstm_cell_bwd(dh_a, dh_b, dc_in, gates, c, c_prev, dG, dc_prev):
    B, H = c.shape
    check(_lib.lib().ick_lstm_cell_bwd(dh_a.data_ptr(), _ptr(dh_b), _ptr(dc_in), gates.data_ptr(), c.data_ptr(),
                                       _ptr(c_prev), dG.data_ptr(), dc_prev.data_ptr(), B, H, _st()), "ick_lstm_cell_bwd")


def dec_attn_x_fwd(h_top, Wa, Uf, feats, Wc, Xe_t, hW_t, attw_t, ctx_t, x_t):
    """stage A of the fused decode step (csrc/decoder_fused.hip).  Wa = attention.weight (E, H+E), Wc = attention_combine.weight (E, 2E)."""
    B, L, E = feats.shape
    H = Wa.shape[1] - E
    check(_lib.lib().ick_dec_attn_x_fwd(_ptr(h_top), Wa.data_ptr(), H + E, Uf.data_ptr(), feats.data_ptr(), Wc.data_ptr() + 4 * E,
                                        2 * E, Xe_t.data_ptr(), hW_t.data_ptr(), attw_t.data_ptr(), ctx_t.data_ptr(),
                                        x_t.data_ptr(), B, L, E, H, _st()), "ick_dec_attn_x_fwd")


def lstm_layer_fwd(inp, h_prev, wih, whh, bih, bhh, c_prev, gates, c_out, h_out, h_drop=None, p_drop=0.0, seed=0):
    B, H = c_out.shape
    check(_lib.lib().ick_lstm_layer_fwd(inp.data_ptr(), wih.shape[1], _ptr(h_prev), wih.data_ptr(), whh.data_ptr(), bih.data_ptr(),
                                        bhh.data_ptr(), _ptr(c_prev), _ptr(gates), c_out.data_ptr(), h_out.data_ptr(), _ptr(h_drop),
                                        p_drop if h_drop is not None else 0.0, seed, _ptr(_dropout_step["ptr"]), B, H, _st()),
          "ick_lstm_layer_fwd")


def lstm_layer_bwd(dG, WT, carry_h_out, d_inp_out, below=None, p_drop=0.0, seed=0):
    """stage G.  below = None (layer 0: raw input gradient into d_inp_out) or a dict(carry_h, carry_c, gates, c, c_prev, dG, first)."""
    B = dG.shape[0]
    H = dG.shape[1] // 4
    K1 = WT.shape[0] - H
    bl = below or {}
    check(_lib.lib().ick_lstm_layer_bwd(dG.data_ptr(), WT.data_ptr(), _ptr(carry_h_out), _ptr(d_inp_out), _ptr(bl.get("carry_h")),
                                        _ptr(bl.get("carry_c")), _ptr(bl.get("gates")), _ptr(bl.get("c")), _ptr(bl.get("c_prev")),
                                        _ptr(bl.get("dG")), int(bl.get("first", 0)), p_drop if below else 0.0, seed,
                                        _ptr(_dropout_step["ptr"]), B, K1, H, _st()), "ick_lstm_layer_bwd")


def dec_attn_x_bwd(dX_t, Wc, attw_t, Uf, hW_t, feats, dUf, dfeats, dhW_t, Wa, dHs_prev, top):
    """stage Z.  dX_t None: only the top layer's cell adjoint (top = dict(carry_h, carry_c, gates, c, c_prev, dG, first))."""
    B, L, E = feats.shape
    H = Wa.shape[1] - E
    check(_lib.lib().ick_dec_attn_x_bwd(_ptr(dX_t), Wc.data_ptr() + 4 * E, 2 * E, _ptr(attw_t), Uf.data_ptr(), _ptr(hW_t),
                                        feats.data_ptr(), dUf.data_ptr(), dfeats.data_ptr(), _ptr(dhW_t), Wa.data_ptr(), H + E,
                                        _ptr(dHs_prev), _ptr(top.get("carry_h")), _ptr(top.get("carry_c")), _ptr(top.get("gates")),
                                        _ptr(top.get("c")), _ptr(top.get("c_prev")), _ptr(top.get("dG")), int(top.get("first", 0)),
                                        B, L, E, H, _st()), "ick_dec_attn_x_bwd")


def transpose2d(src: torch.Tensor, dst: torch.Tensor) -> None:
    """dst[c][r] = src[r][c]; dst may be a row-slice of a wider matrix (its row pitch is dst.stride(0))."""
    rows, cols = src.shape
    check(_lib.lib().ick_transpose2d(src.data_ptr(), src.stride(0), dst.data_ptr(), dst.stride(0), rows, cols, _st()), "ick_transpose2d")


def argmax_rows(x: torch.Tensor) -> torch.Tensor:
    V = x.shape[-1]
    rows = x.numel() // V
    ids = torch.empty(x.shape[:-1], dtype=torch.int64, device=x.device)
    check(_lib.lib().ick_argmax_rows(x.data_ptr(), ids.data_ptr(), rows, V, V, _st()), "ick_argmax_rows")
    return ids


def _skinny_splitk(M: int, N: int, K: int) -> int:
    """Split-K factor for GEMV-sized products (M <= 64 rows, the per-token decoder step): with one 64x64 tile per
    64 output columns only N/64 workgroups exist, each serially walking all of K; splitting K spreads the weight
    read over ~256 CUs (fp32 atomics into a zeroed / accumulating output)."""
    if M > 64:
        return 1
    tiles = (N + 63) // 64
    return max(1, min(K // 32, 256 // max(tiles, 1)))


def beam_topk(logits: torch.Tensor, scores: torch.Tensor, k: int):
    """k best (value, flat index b*V+v) of scores[b] + log_softmax(logits[b]) — device-side beam expansion."""
    Bl, V = logits.shape
    vals = empty(k, device=logits.device)
    idx = torch.empty(k, dtype=torch.int64, device=logits.device)
    check(_lib.lib().ick_beam_topk(logits.data_ptr(), scores.data_ptr(), Bl, V, k, vals.data_ptr(), idx.data_ptr(), _st()),
          "ick_beam_topk")
    return vals, idx


def beam_self_attn(qkv: torch.Tensor, kcache: torch.Tensor, vcache: torch.Tensor, anc: torch.Tensor, heads: int, t: int):
    """KV-cached self-attention of ONE new token per beam row (position t); see include/ick.h ick_beam_self_attn."""
    rows, E3 = qkv.shape
    E = E3 // 3
    out = empty(rows, E, device=qkv.device)
    check(_lib.lib().ick_beam_self_attn(_chk(qkv).data_ptr(), kcache.data_ptr(), vcache.data_ptr(), anc.data_ptr(), out.data_ptr(),
                                        rows, E, heads, t, kcache.shape[0], _st()), "ick_beam_self_attn")
    return out


def beam_step(logits, score, width, seq_in, seq_out, anc_in, anc_out, next_tok, fin_seq, fin_score, fin_len, nfin, t: int,
              end_id: int) -> None:
    """one expansion of every image's beam, entirely on device; see include/ick.h ick_beam_step."""
    B, W = score.shape
    V, Tcap = logits.shape[-1], seq_in.shape[-1]
    check(_lib.lib().ick_beam_step(_chk(logits).data_ptr(), score.data_ptr(), width.data_ptr(), seq_in.data_ptr(), seq_out.data_ptr(),
                                   anc_in.data_ptr(), anc_out.data_ptr(), next_tok.data_ptr(), fin_seq.data_ptr(),
                                   fin_score.data_ptr(), fin_len.data_ptr(), nfin.data_ptr(), B, W, V, Tcap, t, end_id, _st()),
          "ick_beam_step")


def gemm_nt(x: torch.Tensor, w_ptr: int, N: int, K: int, ldb: int, out: torch.Tensor, *, bias=None, residual=None,
            accumulate=False, act=ACT_NONE, splitk: int = 0, zeroed: bool = False):
    """out (M,N) = act(x (M,K) @ W^T + bias) [+ residual]; W given by raw pointer + row pitch (column slices of a
    wider weight, e.g. the W_h / W_f halves of the decoder's attention matrix).  zeroed=True: the caller guarantees
    `out` is already zero (slice of a zero-filled arena), so a split-K launch needs no fill kernel of its own."""
    M = x.numel() // K
    if splitk <= 0:
        splitk = _skinny_splitk(M, N, K) if act == ACT_NONE else 1
    if splitk > 1:
        if not accumulate and not zeroed:
            out.zero_()
        gemm_raw(OP_NT, x.data_ptr(), w_ptr, out.data_ptr(), M, N, K, K, ldb, N, bias=_ptr(bias), residual=_ptr(residual),
                 ldr=N, splitk=splitk, x3=True)
    else:
        gemm_raw(OP_NT, x.data_ptr(), w_ptr, out.data_ptr(), M, N, K, K, ldb, N, bias=_ptr(bias), residual=_ptr(residual),
                 ldr=N, act=act, accumulate=accumulate, x3=True)
    return out


def gemm_nn(dy: torch.Tensor, w_ptr: int, N: int, K: int, ldb: int, out: torch.Tensor, *, residual=None,
            accumulate=False, splitk: int = 0, zeroed: bool = False):
    """out (M,K) = dy (M,N) @ W (N,K) [+ residual]; W by raw pointer with row pitch ldb."""
    M = dy.numel() // N
    if splitk <= 0:
        splitk = _skinny_splitk(M, K, N)
    if splitk > 1:
        if not accumulate and not zeroed:
            out.zero_()
        gemm_raw(OP_NN, dy.data_ptr(), w_ptr, out.data_ptr(), M, K, N, N, ldb, K, residual=_ptr(residual), ldr=K, splitk=splitk)
    else:
        gemm_raw(OP_NN, dy.data_ptr(), w_ptr, out.data_ptr(), M, K, N, N, ldb, K, residual=_ptr(residual), ldr=K,
                 accumulate=accumulate)
    return out


def gemm_tn_acc(dy: torch.Tensor, x: torch.Tensor, dw_ptr: int, N: int, K: int, ldc: int, splitk: int = 0):
    """dW[N][K] (row pitch ldc, raw pointer) += dy (M,N)^T @ x (M,K)."""
    M = dy.numel() // N
    if splitk <= 0:
        tiles = ((N + 127) // 128) * ((K + 127) // 128)
        splitk = max(1, min(32, 512 // max(tiles, 1), M // 256))
    if splitk > 1:
        gemm_raw(OP_TN, dy.data_ptr(), x.data_ptr(), dw_ptr, N, K, M, N, K, ldc, splitk=splitk)
    else:
        gemm_raw(OP_TN, dy.data_ptr(), x.data_ptr(), dw_ptr, N, K, M, N, K, ldc, accumulate=True)


# ----------------------------------------------------------------------------- optimizer tail
def grad_norm(flat: torch.Tensor, workspace: torch.Tensor, norm_out: torch.Tensor, accumulate: bool = False):
    check(_lib.lib().ick_grad_norm(flat.data_ptr(), flat.numel(), workspace.data_ptr(), norm_out.data_ptr(), int(accumulate),
                                   _st()), "ick_grad_norm")


def adamw_step(p, g, m, v, lr, betas, eps, wd, step, norm=None, max_norm=1.0, inv_scale=1.0, write_clipped=False,
               hyper=None, scaler=None):
    check(_lib.lib().ick_adamw_step(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), lr, betas[0], betas[1],
                                    eps, wd, step, _ptr(norm), max_norm, inv_scale, int(write_clipped), _ptr(hyper),
                                    _ptr(scaler), _st()), "ick_adamw_step")


def adam_bias_correction(applied_steps: torch.Tensor, scaler: Optional[torch.Tensor], betas, hyper: torch.Tensor,
                         lr_in: Optional[torch.Tensor] = None, beta1_in: Optional[torch.Tensor] = None) -> None:
    """t = ++applied_steps unless the GradScaler found inf/nan; hyper[g][1..2] = 1-beta1^t, 1-beta2^t for every row g;
    hyper[g][0] = lr_in[g] when lr_in (contiguous, one float per row) is given; beta1_in (one float per row): this step's
    beta1 (OneCycleLR's momentum cycling) -> hyper[g][3], read by adamw_step."""
    assert applied_steps.dtype == torch.int64 and hyper.is_contiguous()
    assert lr_in is None or (lr_in.is_contiguous() and lr_in.numel() >= hyper.shape[0])
    assert beta1_in is None or (beta1_in.is_contiguous() and beta1_in.numel() >= hyper.shape[0] and hyper.shape[1] >= 4)
    check(_lib.lib().ick_adam_bias_correction(applied_steps.data_ptr(), _ptr(scaler), betas[0], betas[1], hyper.data_ptr(),
                                              hyper.shape[0], hyper.shape[1], _ptr(lr_in), _ptr(beta1_in), _st()),
          "ick_adam_bias_correction")


def loss_scale_check(norms: torch.Tensor, state: torch.Tensor) -> None:
    check(_lib.lib().ick_loss_scale_check(norms.data_ptr(), norms.numel(), state.data_ptr(), _st()), "ick_loss_scale_check")


def loss_scale_update(state: torch.Tensor, growth_factor=2.0, backoff_factor=0.5, growth_interval=2000) -> None:
    check(_lib.lib().ick_loss_scale_update(state.data_ptr(), growth_factor, backoff_factor, growth_interval, _st()),
          "ick_loss_scale_update")
