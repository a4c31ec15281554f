#!/usr/bin/env python3
"""layer1 1x1 convolution 64 -> 256 channels at 56x56, B = 64 (M = 200704, K = 64): cost of the BatchNorm-statistics epilogue."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imagecaptioner_amd import ops
x = torch.randn(64, 56, 56, 64, device="cuda")
w = torch.randn(256, 1, 1, 64, device="cuda") * 0.1
def timeit(f, iters=20):
    f(); f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for tile in (0, 19, 3, 1, 2):
    ops._FORCE_TILE[0] = tile
    t0 = timeit(lambda: ops.conv_fwd(x, w, 1, 0))
    line = f"tile {tile:3d}: no stats {t0:6.1f} us |"
    for R in (1, 8, 32, 128):
        stats = torch.zeros(2, R, 256, dtype=torch.float64, device="cuda")
        t = timeit(lambda: ops.conv_fwd(x, w, 1, 0, stats=(stats[0], stats[1])))
        line += f" R={R}: {t:6.1f}"
    print(line)
