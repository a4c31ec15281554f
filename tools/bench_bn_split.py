#!/usr/bin/env python3
"""bn_train_apply (statistics -> coefficients in every thread's prologue) against bn_finalize + scale_shift_act (two launches)
on the step's shapes, inside a captured graph of 20 launches each (GPU box)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imagecaptioner_amd import ops  # noqa: E402


def graph_time(f, n=20, iters=10):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        f()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            f()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters / n * 1e3


for (nb, h, c) in ((64, 56, 64), (64, 56, 256), (64, 28, 128), (64, 28, 512), (64, 14, 256), (64, 14, 1024), (64, 7, 512), (64, 7, 2048)):
    M = nb * h * h
    raw = torch.randn(nb, h, h, c, device="cuda")
    res = torch.randn_like(raw)
    R = ops.stat_copies(M)
    stats = torch.rand(2, R, c, dtype=torch.float64, device="cuda") * M
    stats[1] += stats[0] ** 2 / M
    g, b = torch.ones(c, device="cuda"), torch.zeros(c, device="cuda")
    rm, rv = torch.zeros(c, device="cuda"), torch.ones(c, device="cuda")
    y = torch.empty_like(raw)
    t1 = graph_time(lambda: ops.bn_train_apply(raw, (stats[0], stats[1]), g, b, rm, rv, 0.1, 1e-5, res, True))

    def split():
        co = ops.bn_finalize(stats[0], stats[1], M, g, b, rm, rv, 0.1, 1e-5)
        ops.scale_shift_act(raw, co[0], co[1], res, True, out=y)
    t2 = graph_time(split)
    print(f"M {M:6d} C {c:4d}: fused {t1:6.1f} us   finalize + apply {t2:6.1f} us")
