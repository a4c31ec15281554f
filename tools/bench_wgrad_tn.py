#!/usr/bin/env python3
"""1x1 / stride-1 weight-gradient convolutions two ways (fp32, GPU box): the register-staged CONV_WGRAD kernel the step uses
today against the SAME contraction as a plain TN GEMM on the LDS-DMA kernel (dW[co][ci] = sum_pix dY[pix][co] X[pix][ci]:
for a 1x1 / stride-1 / pad-0 convolution the im2col gather is the identity)."""
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
for (H, Cin, Cout) in [(14, 256, 1024), (14, 1024, 256), (14, 1024, 512), (7, 512, 2048), (7, 2048, 512), (28, 512, 256), (28, 512, 128), (28, 128, 512)]:
    x = torch.randn(B, H, H, Cin, device="cuda")
    dy = torch.randn(B, H, H, Cout, device="cuda")
    K = B * H * H
    conv = (B, H, H, Cin, H, H, Cout, 1, 1, 1, 0)
    fl = 2.0 * Cout * Cin * K
    dw = torch.zeros(Cout, 1, 1, Cin, device="cuda")
    ops.gemm_raw(ops.OP_CONV_WGRAD, dy.data_ptr(), x.data_ptr(), dw.data_ptr(), Cout, Cin, K, Cout, 0, Cin, conv=conv, accumulate=True)
    ref = dw.clone()
    print(f"wgrad 1x1 Cout {Cout} Cin {Cin} K {K}  ({fl / 1e9:.1f} GF)")
    for name, op, tiles in (("CONV_WGRAD regs", ops.OP_CONV_WGRAD, (3, 19, 67, 259)), ("TN LDS-DMA    ", ops.OP_TN, (1, 2, 3, 4, 65, 67, 19))):
        for tile in tiles:
            line = []
            for sk in (8, 16, 24, 48):
                def f():
                    if op == ops.OP_TN:
                        ops.gemm_raw(op, dy.data_ptr(), x.data_ptr(), dw.data_ptr(), Cout, Cin, K, Cout, Cin, Cin, tile=tile, splitk=sk)
                    else:
                        ops.gemm_raw(op, dy.data_ptr(), x.data_ptr(), dw.data_ptr(), Cout, Cin, K, Cout, 0, Cin, conv=conv, tile=tile, splitk=sk)
                dw.zero_(); f(); torch.cuda.synchronize()
                err = ((dw - ref).abs().max() / ref.abs().max()).item()
                t = timeit(f)
                line.append(f"sk{sk:2d} {t:6.1f}us {fl / t / 1e6:4.0f}TF{'' if err < 1e-4 else ' ERR %.1e' % err}")
            print(f"  {name} tile {tile:3d}: " + "  ".join(line))
