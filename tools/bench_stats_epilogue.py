#!/usr/bin/env python3
"""What the BatchNorm-statistics epilogue costs on the large-M convolutions of layer1 / layer2 (GPU box): each shape with
and without statistics, stat copies 1 / 8 / 32, fp32 and native fp16 operands."""
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
for (H, Cin, Cout, R) in [(56, 64, 256, 1), (56, 64, 64, 3), (56, 256, 64, 1), (56, 64, 64, 1), (28, 128, 512, 1), (28, 128, 128, 3), (14, 256, 1024, 1)]:
    pad = R // 2
    x = torch.randn(B, H, H, Cin, device="cuda")
    w = torch.randn(Cout, R, R, Cin, device="cuda") * 0.05
    M, N, K = B * H * H, Cout, R * R * Cin
    conv = (B, H, H, Cin, H, H, Cout, R, R, 1, pad)
    for dt in (None, torch.float16):
        xx, ww = (x, w) if dt is None else (ops.cast16(x, dt), ops.cast16(w, dt))
        y = torch.empty(B, H, H, Cout, device="cuda", dtype=dt or torch.float32)
        line = []
        for copies in (0, 1, 8, 32):
            st = torch.zeros(2, max(copies, 1), Cout, dtype=torch.float64, device="cuda")
            kw = dict(stat_sum=st[0].data_ptr(), stat_sq=st[1].data_ptr(), stat_copies=copies, stat_stride=Cout) if copies else {}
            best = 1e9
            for tile in (1, 3, 19, 65, 67):
                t = timeit(lambda: ops.gemm_raw(ops.OP_CONV_FWD, xx.data_ptr(), ww.data_ptr(), y.data_ptr(), M, N, K, Cin, K, N, conv=conv, tile=tile,
                                                h16=dt, io16=1 if dt is not None else 0, **kw))
                best = min(best, t)
            line.append(f"copies {copies:2d}: {best:7.1f} us")
        byts = (M * K / (R * R) + M * N) * (2 if dt is not None else 4)
        print(f"M {M:6d} N {N:4d} K {K:4d} {'fp16' if dt is not None else 'f32 '}  " + "   ".join(line) + f"   (activation bytes {byts / 1e6:.0f} MB = {byts / 5e6:.0f} us at 5 TB/s)")
