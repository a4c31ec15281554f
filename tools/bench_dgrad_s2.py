#!/usr/bin/env python3
"""Stride-2 data gradients of the trunk (layer3.0 / layer4.0 conv2 3x3 and the 1x1 downsample), fp32 and fp16 (GPU box)."""
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
for (H, Cin, Cout, R) in [(14, 512, 512, 3), (28, 256, 256, 3), (14, 1024, 2048, 1)]:
    pad = R // 2
    Ho = H // 2
    w = torch.randn(Cout, R, R, Cin, device="cuda") * 0.05
    dy = torch.randn(B, Ho, Ho, Cout, device="cuda")
    for dt in (None, torch.float16):
        ww, dd = (w, dy) if dt is None else (ops.cast16(w, dt), ops.cast16(dy, dt))
        t = timeit(lambda: ops.conv_dgrad(dd, ww, (H, H), 2, pad))
        print(f"dgrad s2 H {H} Cin {Cin} Cout {Cout} R {R} {'fp16' if dt is not None else 'f32 '}: {t:7.1f} us")
