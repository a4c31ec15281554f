#!/usr/bin/env python3
"""Weight-gradient convolutions of layer3 / layer4 on native fp16 operands (or fp32: argument f32): time per (tile, split-K)
(GPU box)."""
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
dt = None if (len(sys.argv) > 1 and sys.argv[1] == "f32") else torch.float16
for (H, Cin, Cout, R) in [(14, 256, 256, 3), (14, 256, 1024, 1), (14, 1024, 256, 1), (7, 512, 512, 3), (7, 512, 2048, 1), (7, 2048, 512, 1)]:
    pad = R // 2
    x = torch.randn(B, H, H, Cin, device="cuda")
    dy = torch.randn(B, H, H, Cout, device="cuda")
    if dt is not None:
        x, dy = ops.cast16(x, dt), ops.cast16(dy, dt)
    dw = torch.zeros(Cout, R, R, Cin, device="cuda")
    K, N = B * H * H, R * R * Cin
    conv = (B, H, H, Cin, H, H, Cout, R, R, 1, pad)
    fl = 2.0 * Cout * N * K
    print(f"wgrad Cout {Cout} N {N} K {K}  ({fl / 1e9:.1f} GF)")
    # 16-bit operands: 259 = the register-staged IN16 kernel, the others the LDS-DMA kernel with transposed reads (round 3)
    ops.gemm_raw(ops.OP_CONV_WGRAD, dy.data_ptr(), x.data_ptr(), dw.data_ptr(), Cout, N, K, Cout, 0, N, conv=conv, tile=259, h16=dt, accumulate=True)
    ref = dw.clone()
    for tile in ((259, 1, 2, 3, 4, 65, 67) if dt is not None else (3, 19, 67, 83, 259)):
        line = []
        for sk in (1, 2, 4, 8, 16, 24, 48):
            kw = dict(splitk=sk) if sk > 1 else dict(accumulate=True)
            f = lambda: ops.gemm_raw(ops.OP_CONV_WGRAD, dy.data_ptr(), x.data_ptr(), dw.data_ptr(), Cout, N, K, Cout, 0, N, conv=conv, tile=tile, h16=dt, **kw)
            dw.zero_(); f(); torch.cuda.synchronize()
            err = ((dw - ref).abs().max() / ref.abs().max()).item()
            t = timeit(f)
            line.append(f"sk{sk:2d} {t:6.1f}us {fl / t / 1e6:4.0f}TF{'' if err < 1e-4 else ' ERR %.1e' % err}")
        print(f"  tile {tile:3d}: " + "  ".join(line))
