#!/usr/bin/env python3
"""What the epilogue of the ViT fc1 GEMM (12608 x 1536 x 384) costs: no bias / bias / bias+ReLU / bias+GELU, two tiles."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imagecaptioner_amd import ops
from imagecaptioner_amd._lib import ACT_GELU, ACT_NONE, ACT_RELU, OP_NT
M, N, K = 12608, 1536, 384
x, w, b = torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda") * 0.05, torch.randn(N, device="cuda")
y = torch.empty(M, N, device="cuda")
def timeit(f, iters=30):
    f(); f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for tile in (3, 2):
    for name, kw in (("none", {}), ("bias", dict(bias=b.data_ptr())), ("bias+relu", dict(bias=b.data_ptr(), act=ACT_RELU)),
                     ("bias+gelu", dict(bias=b.data_ptr(), act=ACT_GELU))):
        t = timeit(lambda: ops.gemm_raw(OP_NT, x.data_ptr(), w.data_ptr(), y.data_ptr(), M, N, K, K, K, N, tile=tile, **kw))
        print(f"tile {tile} {name:10s} {t:7.1f} us")
