#!/usr/bin/env python3
"""Time of the ViT fc1 GEMM (N = 1536, K = 384, bias + GELU) as a function of M, i.e. of the number of ROUNDS of resident
workgroups (128x64 tile: 3 per CU -> 768 per round; 64x64: 5 per CU -> 1280): shows whether rounds run in lock-step."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imagecaptioner_amd import ops  # noqa: E402
from imagecaptioner_amd._lib import ACT_GELU, OP_NT  # noqa: E402

N, K = 1536, 384
w = torch.randn(N, K, device="cuda") * 0.05
b = torch.randn(N, device="cuda")


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


for tile, bm, bn, slots in ((3, 128, 64, 768), (2, 64, 64, 1280), (1, 128, 128, 512), (19, 128, 64, 512)):
    print(f"tile {tile} ({bm}x{bn}), {slots} resident workgroups per round")
    for M in (1024, 2048, 4096, 6144, 8192, 10240, 12288, 12608, 14336, 16384, 32768):
        x = torch.randn(M, K, device="cuda")
        y = torch.empty(M, N, device="cuda")
        t = timeit(lambda: ops.gemm_raw(OP_NT, x.data_ptr(), w.data_ptr(), y.data_ptr(), M, N, K, K, K, N, bias=b.data_ptr(),
                                        act=ACT_GELU, tile=tile))
        tiles = ((M + bm - 1) // bm) * (N // bn)
        print(f"  M {M:6d}  tiles {tiles:5d}  rounds {tiles / slots:5.2f}  {t:7.1f} us  {2.0 * M * N * K / t / 1e6:6.1f} TF")
