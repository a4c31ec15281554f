#!/usr/bin/env python3
"""Diagnostic: replay every distinct GEMM descriptor of a cfg5 forward+backward with the LDS-staged epilogue and with
the direct epilogue (IckGemm.tile bit 9) on identical inputs and report descriptors whose outputs differ."""
import collections, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imagecaptioner_amd import ops
from imagecaptioner_amd.student_model import CaptioningStudent
from imagecaptioner_amd.utils.seeded_init import apply_seeded_init, synthetic_batch

OPN = ["NT", "NN", "TN", "CONV_FWD", "CONV_FWD_C4", "CONV_DGRAD", "CONV_WGRAD", "CONV_DGRAD_S2"]
records = []
orig = ops.gemm_raw


def rec(op, A, Bp, C, M, N, K, lda, ldb, ldc, **kw):
    records.append((op, M, N, K, lda, ldb, ldc, kw.get("batch", (1, 1)), kw.get("strides", (0,) * 6), kw.get("splitk", 1),
                    kw.get("conv"), kw.get("act", 0), kw.get("bias") is not None, kw.get("residual") is not None,
                    kw.get("stat_sum") is not None, kw.get("accumulate", False), kw.get("ldr", 0), kw.get("alpha", 1.0)))
    return orig(op, A, Bp, C, M, N, K, lda, ldb, ldc, **kw)


m = apply_seeded_init(CaptioningStudent(5000, 384, 768, 3, use_attention_refinement=True), 0).cuda().train()
for mod in m.modules():
    if isinstance(mod, torch.nn.Dropout):
        mod.p = 0.0
m.attention_refinement.attention.dropout = 0.0
m.decoder.lstm.dropout = 0.0
images, caps = synthetic_batch(2, 5000, 16, seed=5)
ops.gemm_raw = rec
logits, enc, _, _ = m(images.cuda(), caps[:-1].cuda())
(logits.sum() + enc.sum()).backward()
ops.gemm_raw = orig
torch.cuda.synchronize()
cnt = collections.Counter(records)
print(len(records), "launches,", len(cnt), "distinct")
g = torch.Generator(device="cuda").manual_seed(0)
big = torch.empty(1 << 26, device="cuda").normal_(generator=g)
stat = torch.zeros(2, 2, 8192, dtype=torch.float64, device="cuda")
bad = 0
for r in cnt:
    op, M, N, K, lda, ldb, ldc, batch, strides, splitk, conv, act, bias, res, st, acc, ldr, alpha = r
    if splitk > 1:
        continue
    outs = []
    for variant in (0, 512):
        out = torch.full((1 << 24,), 0.5, device="cuda")
        kw = dict(batch=batch, strides=strides, splitk=splitk, conv=conv, act=act, accumulate=acc, ldr=ldr, alpha=alpha)
        if bias: kw["bias"] = big.data_ptr() + (1 << 20) * 4
        if res: kw["residual"] = big.data_ptr() + (1 << 25) * 4
        if st:
            stat.zero_()
            kw["stat_sum"], kw["stat_sq"] = stat[variant > 0, 0].data_ptr(), stat[variant > 0, 1].data_ptr()
        ops._FORCE_TILE[0] = 0
        orig(op, big.data_ptr(), big.data_ptr() + (1 << 24) * 4, out.data_ptr(), M, N, K, lda, ldb, ldc, tile=(ops._TUNED.get(f"{op}:{M}:{N}:{K}:{batch[0] * batch[1]}:{splitk}", 0) & 255) | variant or variant or 0, **kw)
        torch.cuda.synchronize()
        outs.append(out)
    d = (outs[0] - outs[1]).abs().max().item()
    if d != 0.0 or not torch.equal(outs[0], outs[1]):
        bad += 1
        print("DIFF", d, OPN[op], M, N, K, "lda/ldb/ldc", lda, ldb, ldc, "batch", batch, strides, "conv", conv, "act", act, "bias", bias, "res", res, "stat", st, "acc", acc, "ldr", ldr)
print("descriptors with different outputs:", bad)
