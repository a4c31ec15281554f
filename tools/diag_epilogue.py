#!/usr/bin/env python3
"""Diagnostic: cfg5 forward+backward twice in one process (LDS-staged vs direct epilogue via ops._TILE_OR), comparing
the output region of EVERY GEMM launch in order; prints the first launches whose outputs differ beyond atomics noise."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imagecaptioner_amd import ops
from imagecaptioner_amd.student_model import CaptioningStudent
from imagecaptioner_amd.utils.seeded_init import apply_seeded_init, synthetic_batch

OPN = ["NT", "NN", "TN", "CONV_FWD", "CONV_FWD_C4", "CONV_DGRAD", "CONV_WGRAD", "CONV_DGRAD_S2"]
orig = ops.gemm_raw
log = []


def snap(op, A, Bp, C, M, N, K, lda, ldb, ldc, **kw):
    r = orig(op, A, Bp, C, M, N, K, lda, ldb, ldc, **kw)
    if op == 7:
        rows = 4 * M
    else:
        rows = M
    nb = kw.get("batch", (1, 1)); st = kw.get("strides", (0,) * 6)
    span = (nb[0] - 1) * st[4] + (nb[1] - 1) * st[5] + (rows - 1) * ldc + N
    out = _from_ptr(C, span).clone()
    log.append(((OPN[op], M, N, K, ldc, nb, kw.get("splitk", 1), kw.get("accumulate", False), kw.get("residual") is not None, kw.get("act", 0)), out))
    return r


def _from_ptr(ptr, n):
    # build a tensor aliasing device memory [ptr, ptr + 4n)
    class _Holder:
        pass
    h = _Holder()
    h.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f4", "data": (ptr, False), "version": 2}
    return torch.as_tensor(h, device="cuda")


def run(or_bits):
    log.clear()
    ops._TILE_OR[0] = or_bits
    torch.manual_seed(0)
    m = apply_seeded_init(CaptioningStudent(5000, 384, 768, 3, use_attention_refinement=True), 0).cuda().train()
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    m.attention_refinement.attention.dropout = 0.0
    m.decoder.lstm.dropout = 0.0
    images, caps = synthetic_batch(2, 5000, 16, seed=5)
    ops.gemm_raw = snap
    logits, enc, _, _ = m(images.cuda(), caps[:-1].cuda())
    (logits * torch.randn(logits.shape, generator=torch.Generator().manual_seed(1)).cuda()).sum().backward()
    ops.gemm_raw = orig
    torch.cuda.synchronize()
    return list(log)


a = run(0)
b = run(512)
print(len(a), len(b), "launches")
shown = 0
for i, ((ka, ta), (kb, tb)) in enumerate(zip(a, b)):
    assert ka == kb, (i, ka, kb)
    d = (ta - tb).abs().max().item()
    s = tb.abs().max().item()
    if d > 0.0:
        print(i, "rel diff %.3e" % (d / max(s, 1e-30)), ka)
        shown += 1
        if shown >= 12:
            break
print("done; launches over threshold shown:", shown)
