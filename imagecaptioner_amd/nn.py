"""HIP-backed building blocks: torch.autograd.Functions with hand-written backward passes that
call libick.so through imagecaptioner_amd.ops, and thin nn.Module parameter holders with the
reference's state_dict names.  No function here computes with torch operators: torch provides
device memory (empty/zeros), the autograd graph between the large fused Functions, and streams.

Gradient convention: a Function accumulates parameter gradients IN PLACE into `param.grad`
(allocating zeros on first use) and returns None for those inputs — the same end state autograd
produces, without one extra read-modify-write pass per parameter, and it lets the trainer alias
every `.grad` into one flat fp32 buffer for a single RCCL all-reduce and a fused AdamW.
"""
from __future__ import annotations

import math
import os
import weakref
from typing import List, Optional

import torch
import torch.nn as nn
from torch.autograd import Function

from . import ops
from ._lib import ACT_GELU, ACT_NONE, ACT_RELU

_seed_state = {"seed": 0x1234ABCD, "ctr": 0}


def manual_seed(seed: int) -> None:
    """Seed of the counter-based dropout generator (csrc/norm_act.hip dropout_kernel)."""
    _seed_state["seed"] = int(seed) & 0xFFFFFFFF
    _seed_state["ctr"] = 0


def _next_seed() -> int:
    _seed_state["ctr"] += 1
    return ((_seed_state["seed"] << 20) ^ _seed_state["ctr"]) & 0xFFFFFFFFFFFF


def grad_buf(p: torch.Tensor) -> torch.Tensor:
    if p.grad is None:
        p.grad = torch.zeros_like(p)      # preserve_format keeps channels_last conv weights channels_last
    return p.grad


def _c(t: torch.Tensor) -> torch.Tensor:
    return t if t.is_contiguous() else t.contiguous()


# ----------------------------------------------------------------------------- Functions
class LinearFn(Function):
    """y = act(x W^T + b) [+ residual]; act in {none, relu}; residual only with act none."""

    @staticmethod
    def forward(ctx, x, w, b, act, residual):
        x = _c(x)
        if residual is not None:
            assert act == ACT_NONE
            residual = _c(residual)
        y = ops.linear_fwd(x, w, b, act, residual)
        ctx.x, ctx.w, ctx.b, ctx.act = x, w, b, act
        ctx.y = y if act == ACT_RELU else None
        ctx.has_res = residual is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = _c(dy)
        g = ops.relu_bwd(dy, ctx.y) if ctx.act == ACT_RELU else dy
        dx = ops.linear_bwd_data(g, ctx.w) if ctx.needs_input_grad[0] else None
        if ctx.w.requires_grad:
            ops.linear_bwd_weight(g, ctx.x, grad_buf(ctx.w))
        if ctx.b is not None and ctx.b.requires_grad:
            ops.colsum_into(g, grad_buf(ctx.b))
        return dx, None, None, None, (dy if ctx.has_res and ctx.needs_input_grad[4] else None)


def linear(x, w, b=None, act=ACT_NONE, residual=None):
    return LinearFn.apply(x, w, b, act, residual)


class LayerNormFn(Function):
    @staticmethod
    def forward(ctx, x, g, b, eps):
        x = _c(x)
        y, mean, rstd = ops.layernorm_fwd(x, g, b, eps, save=True)
        ctx.x, ctx.g, ctx.b, ctx.mean, ctx.rstd = x, g, b, mean, rstd
        return y

    @staticmethod
    def backward(ctx, dy):
        need = ctx.g.requires_grad
        dx = ops.layernorm_bwd(_c(dy), ctx.x, ctx.g, ctx.mean, ctx.rstd, grad_buf(ctx.g) if need else None,
                               grad_buf(ctx.b) if need else None)
        return dx, None, None, None


def layer_norm(x, g, b, eps=1e-5):
    return LayerNormFn.apply(x, g, b, eps)


class DropoutFn(Function):
    @staticmethod
    def forward(ctx, x, p, seed):
        x = _c(x)
        y = torch.empty_like(x)
        ops.dropout(x, y, p, seed)
        ctx.p, ctx.seed = p, seed
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = _c(dy)
        dx = torch.empty_like(dy)
        ops.dropout(dy, dx, ctx.p, ctx.seed)
        return dx, None, None


def dropout(x, p: float, training: bool):
    if not training or p <= 0.0:
        return x
    return DropoutFn.apply(x, p, _next_seed())


class EmbeddingFn(Function):
    @staticmethod
    def forward(ctx, ids, table):
        ids = _c(ids)
        ctx.ids, ctx.table = ids, table
        return ops.embedding_fwd(ids, table)

    @staticmethod
    def backward(ctx, dout):
        if ctx.table.requires_grad:
            ops.embedding_bwd(ctx.ids, _c(dout), grad_buf(ctx.table))
        return None, None


class SelfAttentionCoreFn(Function):
    """softmax(QK^T/sqrt(d))V over a packed in_proj output qkv [(B*L)][3E] (nn.MultiheadAttention core)."""

    @staticmethod
    def forward(ctx, qkv, B, H, L, d, p_drop, seed):
        qkv = _c(qkv)
        E = H * d
        O, P = ops.attention_fwd(qkv, 0, 3 * E, qkv, E, 3 * E, qkv, 2 * E, 3 * E, B, H, L, L, d, False, p_drop, seed)
        ctx.qkv, ctx.P, ctx.dims, ctx.drop = qkv, P, (B, H, L, d), (p_drop, seed)
        return O

    @staticmethod
    def backward(ctx, dO):
        B, H, L, d = ctx.dims
        E = H * d
        dqkv = ops.empty(B * L, 3 * E, device=dO.device)
        ops.attention_bwd(_c(dO), ctx.P, ctx.qkv, 0, 3 * E, ctx.qkv, E, 3 * E, ctx.qkv, 2 * E, 3 * E,
                          dqkv, 0, 3 * E, dqkv, E, 3 * E, dqkv, 2 * E, 3 * E, B, H, L, L, d, *ctx.drop)
        return dqkv, None, None, None, None, None, None


class TokenPoolFn(Function):
    @staticmethod
    def forward(ctx, x, Lo):
        x = _c(x)
        ctx.L = x.shape[1]
        return ops.token_pool_fwd(x, Lo)

    @staticmethod
    def backward(ctx, dy):
        return ops.token_pool_bwd(_c(dy), ctx.L), None


# ----------------------------------------------------------------------------- parameter holders with HIP forward
class Linear(nn.Linear):
    """nn.Linear's parameters and init (state_dict: weight, bias); forward on the MFMA GEMM."""

    def forward(self, x):
        return linear(x, self.weight, self.bias)


class LayerNorm(nn.LayerNorm):
    def forward(self, x):
        return layer_norm(x, self.weight, self.bias, self.eps)


class Embedding(nn.Embedding):
    def forward(self, ids):
        return EmbeddingFn.apply(ids, self.weight)


class Dropout(nn.Dropout):
    def forward(self, x):
        return dropout(x, self.p, self.training)


class ReLU(nn.Module):
    """Placeholder keeping nn.Sequential indices of the reference (the ReLU itself is fused into the GEMM epilogue)."""

    def forward(self, x):
        raise RuntimeError("fused into the preceding Linear; call the owning block")


class Identity(nn.Identity):
    pass


class Conv2d(nn.Module):
    """Weight holder: logical (Cout,Cin,R,S) like nn.Conv2d (state_dict compatible), physically channels_last,
    i.e. [Cout][R][S][Cin] — the layout the implicit-GEMM kernels read."""

    def __init__(self, cin, cout, k, stride=1, padding=0):
        super().__init__()
        w = torch.empty(cout, cin, k, k)
        nn.init.kaiming_normal_(w, mode="fan_out", nonlinearity="relu")
        self.weight = nn.Parameter(w.contiguous(memory_format=torch.channels_last))
        self.stride, self.padding, self.k = stride, padding, k

    def packed(self) -> torch.Tensor:
        w = self.weight
        if not w.is_contiguous(memory_format=torch.channels_last):
            # a loader replaced .data with an NCHW tensor: restore the physical layout once
            w.data = w.data.contiguous(memory_format=torch.channels_last)
        return w.permute(0, 2, 3, 1)

    def packed_grad(self) -> torch.Tensor:
        return grad_buf(self.weight).permute(0, 2, 3, 1)


class BatchNorm2d(nn.Module):
    def __init__(self, c, eps=1e-5, momentum=0.1):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))
        self.register_buffer("running_mean", torch.zeros(c))
        self.register_buffer("running_var", torch.ones(c))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))
        self.eps, self.momentum = eps, momentum


class _Arena:
    """One zero-filled buffer handed out in slices: a single fill kernel instead of one per BatchNorm."""

    def __init__(self, n: int, dtype, device):
        self.buf = torch.zeros(max(n, 1), dtype=dtype, device=device)
        self.off = 0

    def take(self, rows: int, cols: int) -> torch.Tensor:
        n = rows * cols
        v = self.buf[self.off:self.off + n].view(rows, cols)
        self.off += n
        return v


_PARAM_GENERATION = [0]


def bump_param_generation() -> None:
    """Called by whoever changes parameters / BatchNorm buffers through RAW POINTERS (graph replays of the train step,
    the flat AdamW pass, bn_train_apply): those writes do not touch torch's version counters, so caches keyed on
    (data_ptr, _version) alone would go stale — e.g. the eval-mode BatchNorm coefficients between two validations."""
    _PARAM_GENERATION[0] += 1


def _eval_coeffs(bn: BatchNorm2d) -> torch.Tensor:
    """(scale, shift) of an eval-mode BatchNorm, cached on the module until one of its four tensors changes (in-place
    version counters / storage) or a raw-pointer writer announces itself (bump_param_generation), so that repeated
    inference does not relaunch 53 tiny kernels per forward."""
    ts = (bn.weight, bn.bias, bn.running_mean, bn.running_var)
    key = (_PARAM_GENERATION[0],) + tuple((t.data_ptr(), t._version) for t in ts)
    cached = getattr(bn, "_ick_eval_co", None)
    if cached is None or cached[0] != key:
        cached = (key, ops.bn_eval_coeffs(bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps))
        bn._ick_eval_co = cached
    return cached[1]


# ---- 16-bit activation storage of the training trunk (the reference's autocast regime keeps activations and a weight copy
# in fp16, train_student_kd.py:271).  Under ops.precision("bf16" / "fp16") a TRAIN-mode trunk stores every activation,
# raw conv output and gradient as bf16 / fp16: the convolutions read them as native 16-bit MFMA operands (ick_gemm_h16),
# BatchNorm / max-pool move half the bytes; statistics (fp64), BatchNorm parameters, weight gradients and master weights
# stay fp32.  ICK_TRUNK16=0 keeps fp32 storage (the r01/r02 "fp32 image" AMP path) for A/B runs.
_TRUNK16 = [os.environ.get("ICK_TRUNK16", "1") != "0"]
_H16_OF = {"bf16": torch.bfloat16, "fp16": torch.float16}
_WEIGHT_SHADOW: dict = {}     # (id(parameter), 16-bit dtype) -> 16-bit [Cout][R][S][Cin] copy kept current by a trainer (one flat cast per step)


def install_weight_shadow(param: torch.Tensor, w16: torch.Tensor):
    """Registers a 16-bit copy of a conv weight that the caller keeps up to date (KDTrainer: one cast of the flat
    parameter buffer per step); conv weights without one are cast on use.  Keyed by parameter AND dtype (a bf16 and an fp16
    trainer on one student do not overwrite each other); the entry goes away with the parameter, or earlier through
    remove_weight_shadows(keys) — the owner's close().  Returns the key."""
    key = (id(param), w16.dtype)
    _WEIGHT_SHADOW[key] = (weakref.ref(param), w16)             # (the weak reference guards against a recycled id)
    weakref.finalize(param, _WEIGHT_SHADOW.pop, key, None)
    return key


def remove_weight_shadows(keys) -> None:
    for k in keys:
        _WEIGHT_SHADOW.pop(k, None)


def clear_weight_shadows() -> None:
    _WEIGHT_SHADOW.clear()


_SHADOWS_CURRENT = [False]


class weight_shadows_current:
    """with weight_shadows_current(): ...  — the registered shadows hold THIS moment's weights (the trainer enters it for the
    span between its per-step cast and its optimizer pass).  Outside it — validation between training steps — a shadow is one
    AdamW update behind the master weights and is not used: the weight is cast (and cached) on use instead."""

    def __enter__(self):
        self.prev, _SHADOWS_CURRENT[0] = _SHADOWS_CURRENT[0], True
        return self

    def __exit__(self, *exc):
        _SHADOWS_CURRENT[0] = self.prev
        return False


def _w16(conv: Conv2d, dt: torch.dtype) -> torch.Tensor:
    """The conv weight as [Cout][R][S][Cin] in the 16-bit storage type: the trainer's shadow, a cached copy of a frozen
    weight, or a fresh cast."""
    sh = _WEIGHT_SHADOW.get((id(conv.weight), dt)) if _SHADOWS_CURRENT[0] else None
    if sh is not None and sh[0]() is conv.weight:
        return sh[1]
    w = conv.packed()
    # cached copy: a frozen weight changes only through torch (load_state_dict -> version counter); a trainable one also by
    # raw pointer (the flat AdamW pass), which bump_param_generation announces (inference between training steps)
    key = (w.data_ptr(), conv.weight._version, dt, _PARAM_GENERATION[0] if conv.weight.requires_grad else -1)
    cached = getattr(conv, "_ick_w16", None)
    if cached is None or cached[0] != key:
        cached = (key, ops.cast16(w.contiguous(), dt))
        conv._ick_w16 = cached
    return cached[1]


def _wlike(conv: Conv2d, like: torch.Tensor) -> torch.Tensor:
    return conv.packed() if like.dtype == torch.float32 else _w16(conv, like.dtype)


def conv_bn(x, conv: Conv2d, bn: BatchNorm2d, relu: bool, residual, train: bool, w_packed=None, arena=None, counters=None,
            out_dtype=None, keep_raw: bool = True):
    """raw = conv(x); y = [relu](bn(raw) [+ residual]).  Train mode: batch statistics from the conv epilogue (fp64
    sums), normalisation + running-stat update in one pass over raw (also for frozen layers).
    Returns (y, raw, mean, invstd).  16-bit x: 16-bit weights, raw and y (out_dtype: the stem's fp32 images -> 16-bit).
    keep_raw=False (layers nobody differentiates through: the frozen stem / layer1 / layer2): y overwrites raw."""
    w = w_packed if w_packed is not None else _wlike(conv, x)
    if train:
        bn._ick_eval_co = None       # running statistics (and soon the affine parameters) change under raw kernels
        C = w.shape[0]
        Ho, Wo = ops.conv_out_hw(x.shape[1], x.shape[2], w.shape[1], w.shape[2], conv.stride, conv.padding)
        R = ops.stat_copies(x.shape[0] * Ho * Wo)
        stats = (arena.take(2 * R, C) if arena is not None else
                 torch.zeros(2 * R, C, dtype=torch.float64, device=x.device)).view(2, R, C)
        raw = ops.conv_fwd(x, w, conv.stride, conv.padding, stats=(stats[0], stats[1]), out_dtype=out_dtype)
        y, mean, inv = ops.bn_train_apply(raw, stats, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.momentum,
                                          bn.eps, residual, relu, inplace=not keep_raw)
        if counters is not None:
            counters.append(bn.num_batches_tracked)
        else:
            bn.num_batches_tracked += 1      # bookkeeping counter (int64), not part of the arithmetic
        return y, raw, mean, inv
    # eval mode: BatchNorm is a per-channel affine map -> folded into the conv epilogue (no second pass over the output);
    # same arithmetic as the two-pass form: fmaf(conv, scale, shift) + residual, then ReLU
    co = _eval_coeffs(bn)
    cin, cout = w.shape[3], w.shape[0]
    if (cin % 32 == 0 or cin == 4) and cout % 4 == 0:       # shapes the LDS-DMA kernels take (every ResNet-50 conv)
        y = ops.conv_fwd(x, w, conv.stride, conv.padding, scale=co[0], shift=co[1], residual=residual, relu=relu,
                         out_dtype=out_dtype)
    else:                                                   # two passes: conv, then the affine map + residual + ReLU
        y = ops.scale_shift_act(ops.conv_fwd(x, w, conv.stride, conv.padding), co[0], co[1], residual, relu)
    return y, None, None, None


class Bottleneck(nn.Module):
    """torchvision Bottleneck (v1.5: stride on the 3x3) — holder with torchvision's attribute names."""

    def __init__(self, inplanes, planes, stride=1, downsample=False):
        super().__init__()
        self.conv1 = Conv2d(inplanes, planes, 1)
        self.bn1 = BatchNorm2d(planes)
        self.conv2 = Conv2d(planes, planes, 3, stride=stride, padding=1)
        self.bn2 = BatchNorm2d(planes)
        self.conv3 = Conv2d(planes, planes * 4, 1)
        self.bn3 = BatchNorm2d(planes * 4)
        self.downsample = nn.Sequential(Conv2d(inplanes, planes * 4, 1, stride=stride), BatchNorm2d(planes * 4)) \
            if downsample else None


class _Placeholder(nn.Module):
    """Parameter-free child that only keeps the Sequential indices of torchvision's resnet children (relu, maxpool)."""


def build_resnet50_children() -> nn.Sequential:
    """list(resnet50.children())[:-2] as an nn.Sequential: 0 conv1, 1 bn1, 2 relu, 3 maxpool, 4-7 layer1-4
    (reference: /root/reference/src/student_model.py:16-20)."""
    mods: List[nn.Module] = [Conv2d(3, 64, 7, stride=2, padding=3), BatchNorm2d(64), _Placeholder(), _Placeholder()]
    inpl = 64
    for planes, blocks, stride in ((64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2)):
        layer = [Bottleneck(inpl, planes, stride, downsample=True)]
        inpl = planes * 4
        for _ in range(1, blocks):
            layer.append(Bottleneck(inpl, planes))
        mods.append(nn.Sequential(*layer))
    return nn.Sequential(*mods)


def _block_trainable(blk: Bottleneck) -> bool:
    return any(p.requires_grad for p in blk.parameters())


def bottleneck_forward(blk: Bottleneck, x, train: bool, arena=None, counters=None, keep_raw: bool = True):
    """One torchvision Bottleneck on NHWC activations: out = relu(bn3(conv3(relu(bn2(conv2(relu(bn1(conv1 x))))))) + identity).
    Returns (out, rec); rec holds what bottleneck_backward needs (raw conv outputs, activations, saved statistics)."""
    cb = lambda t, c, b, relu, res: conv_bn(t, c, b, relu, res, train, arena=arena, counters=counters, keep_raw=keep_raw)
    a1, r1, m1, i1 = cb(x, blk.conv1, blk.bn1, True, None)
    a2, r2, m2, i2 = cb(a1, blk.conv2, blk.bn2, True, None)
    if blk.downsample is not None:
        idt, rd, md, idv = cb(x, blk.downsample[0], blk.downsample[1], False, None)
    else:
        idt, rd, md, idv = x, None, None, None
    out, r3, m3, i3 = cb(a2, blk.conv3, blk.bn3, True, idt)
    return out, dict(x=x, a1=a1, r1=r1, m1=m1, i1=i1, a2=a2, r2=r2, m2=m2, i2=i2, out=out, r3=r3, m3=m3, i3=i3, rd=rd, md=md,
                     idv=idv)


# ---- weight gradients on a side stream.  In the trunk's backward the weight gradient of a convolution is needed only by the
# optimizer, the data gradient by the very next kernel: with a side stream installed (KDTrainer does), every conv_wgrad is
# launched there — a parallel branch of the captured graph — meant to fill the ramp-up / tail / epilogue bubbles of the
# data-gradient and BatchNorm kernels on the main stream (each launch has ~17 us of such fixed cost, profiles/r02*).
# trunk_backward_stage joins the branch before it returns (the gradient buckets leave after that).  Measured: not a win on
# this step (fp32 28.19 vs 27.58 ms, fp16 17.59 vs 17.71 ms) -> an opt-in of KDTrainer (ICK_WGRAD_STREAM=1), off by default.
_WGRAD_SIDE = {"stream": None, "keep": []}


def set_wgrad_stream(stream) -> None:
    """stream: a torch.cuda.Stream, or None to run weight gradients in program order on the current stream."""
    _WGRAD_SIDE["stream"], _WGRAD_SIDE["keep"] = stream, []


def _join_wgrad_stream() -> None:
    side = _WGRAD_SIDE["stream"]
    if side is not None and _WGRAD_SIDE["keep"]:
        torch.cuda.current_stream().wait_stream(side)
        _WGRAD_SIDE["keep"] = []          # the operands were kept alive until here: the allocator must not hand them out early


def bottleneck_backward(blk: Bottleneck, r: dict, d, need_in: bool, sums_arena=None):
    """Backward of bottleneck_forward (train-mode BatchNorm): d = dL/d out; accumulates every weight / BatchNorm gradient
    into .grad and returns dL/dx (None unless need_in)."""
    def bnb(dyv, ymask, raw, mean, inv, bn, want_g=False):
        C = bn.weight.numel()
        return ops.bn_bwd(dyv, ymask, raw, mean, inv, bn.weight, grad_buf(bn.weight) if bn.weight.requires_grad else None,
                          grad_buf(bn.bias) if bn.weight.requires_grad else None, want_g, True,
                          sums=sums_arena.take(2 * ops.BN_BWD_COPIES, C).view(2, ops.BN_BWD_COPIES, C) if sums_arena is not None else None)

    def wgrad(conv, dyv, xin):
        if not conv.weight.requires_grad:
            return
        side = _WGRAD_SIDE["stream"]
        if side is None:
            ops.conv_wgrad(dyv, xin, conv.packed_grad(), conv.stride, conv.padding)
            return
        side.wait_stream(torch.cuda.current_stream())      # dyv is ready
        with torch.cuda.stream(side):
            ops.conv_wgrad(dyv, xin, conv.packed_grad(), conv.stride, conv.padding)
        _WGRAD_SIDE["keep"].append((dyv, xin))

    dx3, g3 = bnb(d, r["out"], r["r3"], r["m3"], r["i3"], blk.bn3, want_g=True)
    wgrad(blk.conv3, dx3, r["a2"])
    da2 = ops.conv_dgrad(dx3, _wlike(blk.conv3, dx3), r["a2"].shape[1:3], 1, 0)
    dx2, _ = bnb(da2, r["a2"], r["r2"], r["m2"], r["i2"], blk.bn2)
    wgrad(blk.conv2, dx2, r["a1"])
    da1 = ops.conv_dgrad(dx2, _wlike(blk.conv2, dx2), r["a1"].shape[1:3], blk.conv2.stride, 1)
    dx1, _ = bnb(da1, r["a1"], r["r1"], r["m1"], r["i1"], blk.bn1)
    wgrad(blk.conv1, dx1, r["x"])
    hw = r["x"].shape[1:3]
    if blk.downsample is None:
        return ops.conv_dgrad(dx1, _wlike(blk.conv1, dx1), hw, 1, 0, residual=g3) if need_in else None
    dsc, dsb = blk.downsample[0], blk.downsample[1]
    dxd, _ = bnb(g3, None, r["rd"], r["md"], r["idv"], dsb)
    wgrad(dsc, dxd, r["x"])
    if not need_in:
        return None
    dxi = ops.conv_dgrad(dx1, _wlike(blk.conv1, dx1), hw, 1, 0)
    ops.conv_dgrad(dxd, _wlike(dsc, dxd), hw, dsc.stride, 0, out=dxi, accumulate=True)
    return dxi


def _bn_channels(blocks) -> int:
    return sum(b.bn1.weight.numel() + b.bn2.weight.numel() + b.bn3.weight.numel() +
               (b.downsample[1].weight.numel() if b.downsample is not None else 0) for b in blocks)


class ResNetTrunkFn(Function):
    """The whole ResNet-50 trunk as ONE autograd node with a hand-scheduled backward: conv raw outputs and block
    outputs are kept only for blocks at/after the first trainable one; residual adds, ReLU masks and BatchNorm
    reductions are fused into the kernels on both passes; weight gradients accumulate straight into .grad."""

    @staticmethod
    def forward(ctx, images, resnet: nn.Sequential, train: bool, *params):
        images = _c(images)
        x4 = ops.nchw3_to_nhwc4(images)
        stem: Conv2d = resnet[0]
        # the reference freezes the stem only under fine_tune=True (student_model.py:23-30); CNNEncoder(fine_tune=False) trains it
        stem_trainable = any(p.requires_grad for p in list(stem.parameters()) + list(resnet[1].parameters()))
        w4 = ops.nchw3_to_nhwc4(_c(stem.weight.detach()))          # (64,3,7,7) -> (64,7,7,4), zero 4th channel
        blocks: List[Bottleneck] = [b for li in (4, 5, 6, 7) for b in resnet[li]]
        arena = counters = None
        if train:
            # up to 8 accumulator copies per BatchNorm (ops.stat_copies)
            arena, counters = _Arena(2 * (64 + _bn_channels(blocks)) * 8, torch.float64, images.device), []
        # 16-bit activation storage (see _TRUNK16) under a 16-bit GEMM precision: train mode with a frozen stem (its adjoint
        # kernels are fp32), and eval mode (cfg2's bf16 inference: conv + folded BatchNorm + residual + ReLU in one epilogue,
        # 16-bit in, 16-bit out)
        dt16 = _H16_OF.get(ops.gemm_precision()) if (_TRUNK16[0] and (not train or not stem_trainable)) else None
        ctx.dt16 = dt16
        first = 0 if stem_trainable else next((i for i, b in enumerate(blocks) if _block_trainable(b)), len(blocks))
        want_bwd = any(ctx.needs_input_grad) and first < len(blocks)   # (forward itself always runs in no-grad mode)
        ctx.eval_mode_graph = want_bwd and not train     # eval forward under autograd (validation loops): fine until .backward()
        want_bwd = want_bwd and train
        ys, raw_s, mean_s, inv_s = conv_bn(x4, stem, resnet[1], True, None, train, w_packed=w4, arena=arena, counters=counters,
                                           out_dtype=dt16, keep_raw=want_bwd and stem_trainable)
        y = ops.maxpool3x3s2(ys)
        recs = []
        for i, blk in enumerate(blocks):
            y, rec = bottleneck_forward(blk, y, train, arena, counters, keep_raw=want_bwd and i >= first)
            if want_bwd and i >= first:
                recs.append(rec)
        if counters:
            torch._foreach_add_(counters, 1)       # all num_batches_tracked counters in one launch (bookkeeping)
        ctx.blocks, ctx.first, ctx.recs = blocks, first, recs
        ctx.stem = dict(x4=x4, ys=ys, raw=raw_s, mean=mean_s, inv=inv_s, conv=stem, bn=resnet[1]) if (want_bwd and stem_trainable) else None
        Nb, H, W, C = y.shape
        if dt16 is not None:
            y = ops.cast32(y)                                        # the decoder side reads fp32 features
        ctx.pool_from = None
        if (H, W) != (7, 7):                                         # nn.AdaptiveAvgPool2d((7,7)), student_model.py:34,60: real work
            ctx.pool_from = (H, W)                                   # only for inputs other than 224 x 224
            y = ops.adaptive_avgpool_fwd(y, 7, 7)
            H = W = 7
        return y.view(Nb, H * W, C)                                  # (B,49,2048): NHWC is already "permute(0,2,1)"

    @staticmethod
    def backward(ctx, dy):
        blocks, first, recs = ctx.blocks, ctx.first, ctx.recs
        if ctx.eval_mode_graph:
            raise NotImplementedError("backward through eval-mode BatchNorm is not needed by the KD step")
        if not recs:
            return (None,) * len(ctx.needs_input_grad)
        if ctx.pool_from is not None:
            Nb, C = recs[-1]["out"].shape[0], recs[-1]["out"].shape[3]
            d = ops.adaptive_avgpool_bwd(_c(dy).view(Nb, 7, 7, C), *ctx.pool_from)
        else:
            d = _c(dy).view(recs[-1]["out"].shape)
        if ctx.dt16 is not None:
            d = ops.cast16(d, ctx.dt16)                              # gradients travel through the trunk in the storage type
        if ops.gemm_precision() == "f32x3":
            ops.begin_absmax_arena(64, d.device)     # one zero fill for the absmax slots of this backward's data gradients
        state = dict(blocks=blocks, first=first, recs=recs, d=d, next=len(blocks) - 1, stem=ctx.stem,
                     sums=_Arena(2 * ops.BN_BWD_COPIES * _bn_channels(blocks[first:]), torch.float64, d.device))
        if _TRUNK_DEFER["on"]:
            # data-parallel step: the trainer runs the trunk's backward itself, stage by stage, so that the gradient
            # buckets of everything ABOVE the trunk are already on the wire while layer4 / layer3 are still computing
            _TRUNK_DEFER["pending"] = state
        else:
            trunk_backward_stage(state, first)
        return (None,) * len(ctx.needs_input_grad)


_TRUNK_DEFER = {"on": False, "pending": None}


def trunk_backward_stage(state: dict, stop: int) -> None:
    """Backward through blocks state['next'] .. stop (descending, inclusive) of the trunk; weight / BatchNorm gradients
    accumulate into .grad, the running dL/dx is carried in the state for the next stage."""
    blocks, first, recs = state["blocks"], state["first"], state["recs"]
    stop = max(stop, first)
    stem = state.get("stem")
    for i in range(state["next"], stop - 1, -1):
        state["d"] = bottleneck_backward(blocks[i], recs[i - first], state["d"], i > first or stem is not None, state["sums"])
        recs[i - first] = None                                       # free this block's activations
        ops._ABSMAX_ARENA["cache"].clear()                           # (a convolution's dgrad / wgrad pair shares its absmax pass; no longer)
    _join_wgrad_stream()
    state["next"] = stop - 1
    if state["next"] < first:
        ops.end_absmax_arena()                                       # (drops the references the absmax cache holds on gradient tensors)
    if stem is not None and state["next"] < first:                   # CNNEncoder(fine_tune=False): max-pool, bn1, conv1 adjoints
        bn, conv = stem["bn"], stem["conv"]
        d = ops.maxpool3x3s2_bwd(stem["ys"], state["d"])
        dx, _ = ops.bn_bwd(d, stem["ys"], stem["raw"], stem["mean"], stem["inv"], bn.weight, grad_buf(bn.weight), grad_buf(bn.bias),
                           False, True)
        dw4 = ops.zeros(conv.weight.shape[0], conv.k, conv.k, 4, device=dx.device)
        ops.conv_wgrad(dx, stem["x4"], dw4, conv.stride, conv.padding)
        ops.nhwc4_to_nhwc3_add(dw4, conv.packed_grad())
        state["stem"] = None


class deferred_trunk_backward:
    """with deferred_trunk_backward() as box: loss.backward() stops at the trunk boundary; box.state then drives
    trunk_backward_stage (KDTrainer's bucketed data-parallel step)."""

    def __enter__(self):
        _TRUNK_DEFER["on"], _TRUNK_DEFER["pending"] = True, None
        self.state = None
        return self

    def __exit__(self, *exc):
        self.state = _TRUNK_DEFER["pending"]
        _TRUNK_DEFER["on"], _TRUNK_DEFER["pending"] = False, None
        return False


def resnet_trunk(images, resnet: nn.Sequential, train: bool):
    params = [p for p in resnet.parameters() if p.requires_grad]
    return ResNetTrunkFn.apply(images, resnet, train, *params)
