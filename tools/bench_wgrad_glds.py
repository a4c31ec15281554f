#!/usr/bin/env python3
"""Weight-gradient convolutions that are NOT plain TN GEMMs (3x3, stride 2) on the LDS-DMA kernel with the incremental pixel
index (run with ICK_WGRAD_GLDS=1) against the register-staged kernel (tile | 256), fp32, per (tile, split-K); results are
checked against the register-staged kernel's (GPU box)."""
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
for (H, Cin, Cout, R, stride) in [(14, 256, 256, 3, 1), (7, 512, 512, 3, 1), (28, 128, 128, 3, 1), (28, 256, 256, 3, 2), (14, 512, 512, 3, 2),
                                  (28, 512, 1024, 1, 2), (14, 1024, 2048, 1, 2)]:
    pad = R // 2
    Ho = (H + 2 * pad - R) // stride + 1
    x = torch.randn(B, H, H, Cin, device="cuda")
    dy = torch.randn(B, Ho, Ho, Cout, device="cuda")
    dw = torch.zeros(Cout, R, R, Cin, device="cuda")
    K, N = B * Ho * Ho, R * R * Cin
    conv = (B, H, H, Cin, Ho, Ho, Cout, R, R, stride, pad)
    fl = 2.0 * Cout * N * K
    ops.gemm_raw(ops.OP_CONV_WGRAD, dy.data_ptr(), x.data_ptr(), dw.data_ptr(), Cout, N, K, Cout, 0, N, conv=conv, tile=259, accumulate=True)
    ref = dw.clone()
    print(f"wgrad {R}x{R} s{stride} {H}x{H} Cout {Cout} N {N} K {K}  ({fl / 1e9:.1f} GF)")
    sks = sorted({max(1, K // 784 // 2), max(1, K // 784), max(2, K // 392), max(1, min(64, 768 // (((Cout + 127) // 128) * ((N + 127) // 128))))})
    for tile in (259, 275, 1, 2, 3, 4, 65, 67, 19):
        line = []
        for sk in sks:
            kw = dict(splitk=sk) if sk > 1 else dict(accumulate=True)
            f = lambda: ops.gemm_raw(ops.OP_CONV_WGRAD, dy.data_ptr(), x.data_ptr(), dw.data_ptr(), Cout, N, K, Cout, 0, N, conv=conv, tile=tile, **kw)
            dw.zero_(); f(); torch.cuda.synchronize()
            err = ((dw - ref).abs().max() / ref.abs().max()).item()
            t = timeit(f)
            line.append(f"sk{sk:2d} {t:6.1f}us {fl / t / 1e6:4.0f}TF{'' if err < 1e-4 else ' ERR %.1e' % err}")
        print(f"  tile {tile:3d}{' (regs)' if tile & 256 else '       '}: " + "  ".join(line))
