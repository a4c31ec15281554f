#!/usr/bin/env python3
"""Split-K on the ViT's N = 384 Linear shapes (fc2 12608x384x1536, proj 12608x384x384), fp32: does cutting K fill the
half-empty round of 128-row tiles?  Times include the memset of C that the atomic accumulation needs (GPU box)."""
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


for (M, N, K) in [(12608, 384, 1536), (12608, 384, 384), (12608, 1152, 384), (12608, 1536, 384)]:
    A = torch.randn(M, K, device="cuda")
    Bm = torch.randn(N, K, device="cuda") * 0.05
    bias = torch.randn(N, device="cuda")
    res = torch.randn(M, N, device="cuda")
    C = torch.empty(M, N, device="cuda")
    print(f"{M} x {N} x {K}  ({2.0 * M * N * K / 1e9:.1f} GF)")
    for tile in (1, 2, 3, 4, 65, 67):
        line = []
        for sk in (1, 2, 3, 4):
            def f():
                if sk > 1:
                    C.zero_()
                ops.gemm_raw(ops.OP_NT, A.data_ptr(), Bm.data_ptr(), C.data_ptr(), M, N, K, K, K, N, bias=bias.data_ptr(), residual=res.data_ptr(), ldr=N,
                             tile=tile, splitk=sk)
            t = timeit(f)
            line.append(f"sk{sk} {t:6.1f}us {2.0 * M * N * K / t / 1e6:4.0f}TF")
        print(f"  tile {tile:2d}: " + "  ".join(line))
