#!/usr/bin/env python3
"""BatchNorm train-mode apply on 16-bit storage at the trunk's B = 64 shapes (GPU box): us and achieved HBM GB/s
(read raw [+ residual], write y).  Eager launches: below ~20 us per call the figure is the host's launch path, not the kernel
(inside the step's hipGraph the small layers take 6-10 us).  A 16-byte-access variant (eight channels per thread) was built and
measured with this tool: no faster on the large layers (5.2-5.7 TB/s either way), slower on the small ones; not kept."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imagecaptioner_amd import ops  # noqa: E402


def timeit(f, iters=30):
    f(); f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


dt = torch.float16
tot = 0.0
for name, M, C, res, n in (("stem", 802816, 64, False, 1), ("layer1 64", 200704, 64, False, 6), ("layer1 256 (+res)", 200704, 256, True, 3),
                           ("layer1 256 (downsample)", 200704, 256, False, 1), ("layer2 128 @56", 200704, 128, False, 1),
                           ("layer2 128", 50176, 128, False, 7), ("layer2 512 (+res)", 50176, 512, True, 4), ("layer3 256", 12544, 256, False, 11),
                           ("layer3 1024 (+res)", 12544, 1024, True, 6), ("layer4 512", 3136, 512, False, 5), ("layer4 2048 (+res)", 3136, 2048, True, 3)):
    x = torch.randn(M, C, device="cuda").to(dt)
    r = torch.randn(M, C, device="cuda").to(dt) if res else None
    xf = x.float()
    R = ops.stat_copies(M)
    stats = torch.zeros(2, R, C, dtype=torch.float64, device="cuda")
    stats[0, 0] = xf.double().sum(0); stats[1, 0] = (xf.double() ** 2).sum(0)
    g, b = torch.rand(C, device="cuda") + 0.5, torch.randn(C, device="cuda")
    f = lambda: ops.bn_train_apply(x.view(1, 1, M, C), stats, g, b, None, None, 0.1, 1e-5, r.view(1, 1, M, C) if res else None, True)
    y, mean, inv = f()
    ref = torch.relu((xf - xf.mean(0)) / torch.sqrt(xf.var(0, unbiased=False) + 1e-5) * g + b + (r.float() if res else 0))
    err = (y.view(M, C).float() - ref).abs().max().item()
    t = timeit(f)
    nbytes = M * C * 2 * (3 if res else 2)
    tot += t * n
    print(f"{name:26s} M {M:7d} C {C:5d}: {t:7.1f} us  {nbytes / t / 1e3:7.0f} GB/s  max err {err:.1e}  x{n}")
print(f"sum over the step's 53 launches (approx.): {tot / 1e3:.3f} ms")
