#!/usr/bin/env python3
"""Split-K on the layer4 convolutions whose grids are too small for the chip (M = 3136: 100-200 workgroups), fp32: time per
(tile, split) including the memset the atomic accumulation needs, and the result against the unsplit kernel (GPU box)."""
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


B = 64
SHAPES = [(7, 512, 512, 3), (7, 2048, 512, 1), (7, 512, 2048, 1), (14, 256, 256, 3), (14, 1024, 256, 1)]
SPLITS = (1, 2, 3, 4, 6)
if len(sys.argv) > 1 and sys.argv[1] == "layer3":          # which split counts fill whole rounds of the chip on layer3's 196 / 392 tiles
    SHAPES, SPLITS = [(14, 256, 256, 3), (14, 1024, 256, 1), (14, 256, 1024, 1)], (1, 5, 8, 10, 13)
for (H, Cin, Cout, R) in SHAPES:
    pad = R // 2
    x = torch.randn(B, H, H, Cin, device="cuda")
    w = torch.randn(Cout, R, R, Cin, device="cuda") * 0.05
    res = torch.randn(B, H, H, Cout, device="cuda")
    y = torch.empty(B, H, H, Cout, device="cuda")
    M, N, K = B * H * H, Cout, R * R * Cin
    conv = (B, H, H, Cin, H, H, Cout, R, R, 1, pad)
    ops.gemm_raw(ops.OP_CONV_FWD, x.data_ptr(), w.data_ptr(), y.data_ptr(), M, N, K, Cin, K, N, conv=conv, residual=res.data_ptr(), ldr=N)
    ref = y.clone()
    print(f"conv M {M} N {N} K {K} ({2.0 * M * N * K / 1e9:.1f} GF)")
    for tile in (1, 2, 3, 4, 65, 67):
        line = []
        for sk in SPLITS:
            def f():
                if sk > 1:
                    y.zero_()
                ops.gemm_raw(ops.OP_CONV_FWD, x.data_ptr(), w.data_ptr(), y.data_ptr(), M, N, K, Cin, K, N, conv=conv, residual=res.data_ptr(), ldr=N,
                             tile=tile, splitk=sk)
            t = timeit(f)
            err = ((y - ref).abs().max() / ref.abs().max()).item()
            line.append(f"sk{sk} {t:6.1f}us {2.0 * M * N * K / t / 1e6:4.0f}TF{'' if err < 1e-5 else ' ERR %.1e' % err}")
        print(f"  tile {tile:2d}: " + "  ".join(line))
