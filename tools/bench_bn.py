#!/usr/bin/env python3
"""Isolated timing of the BatchNorm elementwise kernels on the step's shapes (B = 64): us per launch and achieved
HBM GB/s against the algorithmic bytes (train_apply: read raw [+ residual], write y; bwd_reduce: read dy, y, x;
bwd_apply: read dy, y, x, write dx [+ g]).  usage: bench_bn.py [f32|fp16|bf16]  (storage type of the activations)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imagecaptioner_amd import ops  # noqa: E402


def timeit(f, iters=20):
    f(); f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


DT = {"f32": torch.float32, "fp16": torch.float16, "bf16": torch.bfloat16}[sys.argv[1] if len(sys.argv) > 1 else "f32"]
EB = 4 if DT == torch.float32 else 2
print(f"{'M':>7s} {'C':>5s} | {'apply us':>9s} {'GB/s':>6s} | {'reduce us':>9s} {'GB/s':>6s} | {'bwd apply us':>12s} {'GB/s':>6s}")
for (nb, h, c, bwd) in ((64, 56, 64, False), (64, 56, 256, False), (64, 28, 128, False), (64, 28, 512, False), (64, 14, 256, True),
                        (64, 14, 1024, True), (64, 7, 512, True), (64, 7, 2048, True)):
    M = nb * h * h
    raw = torch.randn(nb, h, h, c, device="cuda").to(DT)
    res = torch.randn(nb, h, h, c, device="cuda").to(DT)
    R = ops.stat_copies(M)
    stats = torch.rand(2, R, c, dtype=torch.float64, device="cuda") * M
    stats[1] += stats[0] ** 2 / M
    g, b = torch.ones(c, device="cuda"), torch.zeros(c, device="cuda")
    rm, rv = torch.zeros(c, device="cuda"), torch.ones(c, device="cuda")
    t_app = timeit(lambda: ops.bn_train_apply(raw, (stats[0], stats[1]), g, b, rm, rv, 0.1, 1e-5, res, True))
    line = f"{M:7d} {c:5d} | {t_app:9.1f} {M * c * 3 * EB / t_app / 1e3:6.0f} |"
    if bwd:
        mean, inv = torch.zeros(c, device="cuda"), torch.ones(c, device="cuda")
        sums = torch.zeros(2, ops.BN_BWD_COPIES, c, dtype=torch.float64, device="cuda")
        dg, db = torch.zeros(c, device="cuda"), torch.zeros(c, device="cuda")
        L = ops._lib.lib()
        st = ops._st
        dx = torch.empty_like(raw)
        coef = torch.empty(2, c, device="cuda")
        if DT == torch.float32:
            t_red = timeit(lambda: L.ick_bn_bwd_reduce(raw.data_ptr(), res.data_ptr(), raw.data_ptr(), mean.data_ptr(), inv.data_ptr(),
                                                       sums[0].data_ptr(), sums[1].data_ptr(), ops.BN_BWD_COPIES, c, M, c, 1, st()))
            t_ba = timeit(lambda: L.ick_bn_bwd_apply(raw.data_ptr(), res.data_ptr(), raw.data_ptr(), mean.data_ptr(), inv.data_ptr(),
                                                     g.data_ptr(), sums[0].data_ptr(), sums[1].data_ptr(), ops.BN_BWD_COPIES, c,
                                                     coef.data_ptr(), dx.data_ptr(), None, M, c, 1, dg.data_ptr(), db.data_ptr(), 1, st()))
        else:
            f16 = int(DT == torch.float16)
            t_red = timeit(lambda: L.ick_bn_bwd_reduce16(raw.data_ptr(), res.data_ptr(), raw.data_ptr(), mean.data_ptr(), inv.data_ptr(),
                                                         sums[0].data_ptr(), sums[1].data_ptr(), ops.BN_BWD_COPIES, c, M, c, 1, f16, st()))
            t_ba = timeit(lambda: L.ick_bn_bwd_apply16(raw.data_ptr(), res.data_ptr(), raw.data_ptr(), mean.data_ptr(), inv.data_ptr(),
                                                       g.data_ptr(), sums[0].data_ptr(), sums[1].data_ptr(), ops.BN_BWD_COPIES, c,
                                                       coef.data_ptr(), dx.data_ptr(), None, M, c, 1, dg.data_ptr(), db.data_ptr(), 1, f16, st()))
        line += f" {t_red:9.1f} {M * c * 3 * EB / t_red / 1e3:6.0f} | {t_ba:12.1f} {M * c * 4 * EB / t_ba / 1e3:6.0f}"
    print(line)
