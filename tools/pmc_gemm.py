#!/usr/bin/env python3
"""A handful of representative igemm shapes, 3 launches each, for rocprofv3 --pmc runs."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imagecaptioner_amd import ops
tile = int(sys.argv[1]) if len(sys.argv) > 1 else 0
ops._FORCE_TILE[0] = tile
B = 64
x = torch.randn(12608, 384, device="cuda"); w = torch.randn(1536, 384, device="cuda"); y = torch.empty(12608, 1536, device="cuda")
x2 = torch.randn(12608, 1536, device="cuda"); w2 = torch.randn(384, 1536, device="cuda"); y2 = torch.empty(12608, 384, device="cuda")
xa = torch.randn(4096, 4096, device="cuda"); wa = torch.randn(4096, 4096, device="cuda"); ya = torch.empty(4096, 4096, device="cuda")
cx = torch.randn(B, 14, 14, 256, device="cuda"); cw = torch.randn(256, 3, 3, 256, device="cuda") * 0.05
stats = torch.zeros(2, 256, dtype=torch.float64, device="cuda")
for _ in range(3):
    ops.linear_fwd(x, w, out=y)
    ops.linear_fwd(x2, w2, out=y2)
    ops.linear_fwd(xa, wa, out=ya)
    ops.conv_fwd(cx, cw, 1, 1, stats=(stats[0], stats[1]))
torch.cuda.synchronize()
