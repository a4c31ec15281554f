#!/usr/bin/env python3
"""Tile sweep of the weight-gradient launches under precision "f32x3" (three-product kernel, absmax-scaled dY) at the KD step's
B = 64 shapes (GPU box); prints the best tile per descriptor key for imagecaptioner_amd/tuned_tiles_f32x3.json."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imagecaptioner_amd import ops  # noqa: E402


def timeit(f, iters=15):
    f(); f(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


B = 64
table = {}
recs = []
orig = ops.gemm_raw


def rec(op, A, Bp, C, M, N, K, lda, ldb, ldc, **kw):
    if kw.get("x3") and op in (ops.OP_TN, ops.OP_CONV_WGRAD):
        recs.append((op, M, N, K, lda, ldb, ldc, {k: v for k, v in kw.items() if k in ("splitk", "conv", "accumulate")}))
    return orig(op, A, Bp, C, M, N, K, lda, ldb, ldc, **kw)


shapes = [(14, 256, 256, 3, 1), (7, 512, 512, 3, 1), (28, 256, 256, 3, 2), (14, 512, 512, 3, 2), (14, 256, 1024, 1, 1), (14, 1024, 256, 1, 1),
          (7, 512, 2048, 1, 1), (7, 2048, 512, 1, 1), (14, 1024, 512, 1, 1), (28, 512, 256, 1, 1), (28, 512, 1024, 1, 2), (14, 1024, 2048, 1, 2)]
with ops.precision("f32x3"):
    for (H, Cin, Cout, R, stride) in shapes:
        pad = R // 2
        Ho = (H + 2 * pad - R) // stride + 1
        x = torch.randn(B, H, H, Cin, device="cuda"); dy = torch.randn(B, Ho, Ho, Cout, device="cuda") * 1e-5
        dw = torch.zeros(Cout, R, R, Cin, device="cuda")
        recs.clear()
        ops.gemm_raw = rec
        ops.conv_wgrad(dy, x, dw, stride, pad)
        ops.gemm_raw = orig
        if not recs:
            print("no x3 launch for", (H, Cin, Cout, R, stride)); continue
        op, M, N, K, lda, ldb, ldc, kw = recs[0]
        am = ops.absmax(dy, torch.zeros(1, device="cuda"))
        res = {}
        for tile in (0, 1, 2, 3, 4, 65, 67, 19, 83):
            f = lambda: orig(op, dy.data_ptr(), x.data_ptr(), dw.data_ptr(), M, N, K, lda, ldb, ldc, tile=tile, x3=True, a_absmax=am.data_ptr(), **kw)
            try:
                res[tile] = timeit(f)
            except Exception as e:
                res[tile] = float("inf")
        best = min((t for t in res if t), key=lambda t: res[t])
        key = f"{op}:{M}:{N}:{K}:1:{kw.get('splitk', 1)}"
        print(f"{key:32s} default {res[0]:7.1f} us  best tile {best:3d} {res[best]:7.1f} us   " + " ".join(f"{t}:{v:.0f}" for t, v in res.items()))
        if res[best] < 0.97 * res[0]:
            table[key] = best
print(json.dumps(table))
