#!/usr/bin/env python3
"""Representative native 16-bit launches (fp16) for rocprofv3 --pmc runs: layer3's 3x3 convolution forward with statistics
(12544 x 256 x 2304, the tuned tile), its weight gradient, and the 4096^3 NT product on the 256 x 256 tile; 20 launches each."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imagecaptioner_amd import ops  # noqa: E402

dt = torch.float16
B = 64
x = ops.cast16(torch.randn(B, 14, 14, 256, device="cuda"), dt)
w = ops.cast16(torch.randn(256, 3, 3, 256, device="cuda") * 0.05, dt)
dy = ops.cast16(torch.randn(B, 14, 14, 256, device="cuda"), dt)
dw = torch.zeros(256, 3, 3, 256, device="cuda")
a = ops.cast16(torch.randn(4096, 4096, device="cuda"), dt)
b = ops.cast16(torch.randn(4096, 4096, device="cuda"), dt)
c = torch.empty(4096, 4096, device="cuda")
R = ops.stat_copies(B * 14 * 14)
for _ in range(20):
    stats = torch.zeros(2, R, 256, dtype=torch.float64, device="cuda")
    ops.conv_fwd(x, w, 1, 1, stats=(stats[0], stats[1]))
    ops.conv_wgrad(dy, x, dw, 1, 1)
    ops.gemm_raw(ops.OP_NT, a.data_ptr(), b.data_ptr(), c.data_ptr(), 4096, 4096, 4096, 4096, 4096, 4096, h16=dt, tile=69)
torch.cuda.synchronize()
