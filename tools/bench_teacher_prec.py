#!/usr/bin/env python3
"""ViT-teacher Linear shapes (M = 64 x 197) per arithmetic variant of the LDS-DMA kernel (GPU box): exact fp32 MFMA against
the split-bf16 variants that keep fp32 operands in HBM / LDS and split them in registers.  Error against float64."""
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


PRECS = sys.argv[1].split(",") if len(sys.argv) > 1 else ["f32", "bf16x3"]
M = 64 * 197
for name, N, K, act in (("fc1", 1536, 384, 1), ("fc2", 384, 1536, 0), ("qkv", 1152, 384, 0), ("proj", 384, 384, 0), ("square", 4096, 4096, 0)):
    Mx = 4096 if name == "square" else M
    g = torch.Generator().manual_seed(N + K)
    A = torch.randn(Mx, K, generator=g).cuda()
    W = (torch.randn(N, K, generator=g) * K ** -0.5).cuda()
    C = torch.empty(Mx, N, device="cuda")
    ref = (A[:512].double() @ W.double().t())
    fl = 2.0 * Mx * N * K
    for prec in PRECS:
        best = None
        for tile in (1, 2, 3, 4, 34, 65, 67, 129):
            with ops.precision(prec):
                def f():
                    ops.gemm_raw(ops.OP_NT, A.data_ptr(), W.data_ptr(), C.data_ptr(), Mx, N, K, K, K, N, tile=tile, x3=True)
                try:
                    f()
                except Exception as e:       # a tile the variant does not build
                    continue
                torch.cuda.synchronize()
                err = ((C[:512].double() - ref).norm() / ref.norm()).item()
                t = timeit(f)
            if best is None or t < best[0]:
                best = (t, tile, err)
            print(f"  {name:6s} {prec:7s} tile {tile:3d}: {t:7.1f} us {fl / t / 1e6:6.1f} TF  rel-L2 err vs fp64 {err:.2e}")
        print(f"{name:6s} {Mx}x{N}x{K} {prec:7s}: best {best[0]:7.1f} us ({fl / best[0] / 1e6:6.1f} TF) tile {best[1]}  err {best[2]:.2e}")
