#!/usr/bin/env python3
"""Native 16-bit operand GEMM (ick_gemm_h16) against the fp32-image variants on 4096^3 and on the trunk's convolution
shapes at B=64 (GPU box): correctness against torch on the rounded operands, then time / TFLOP/s per tile."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imagecaptioner_amd import ops  # noqa: E402

torch.manual_seed(0)
dev = "cuda"


def timeit(f, iters=20):
    f(); f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


TILES = [1, 2, 3, 4, 18, 19, 65, 67, 83]
NT_TILES = TILES + [69]          # 69: the 256 x 256 tile (native 16-bit NT only)
print("== NT  C = A @ B^T")
for M, N, K in [(4096, 4096, 4096), (8192, 8192, 8192), (1024, 5000, 512), (12544, 1024, 256), (12544, 256, 1024), (3136, 2048, 512)]:
    A = torch.randn(M, K, device=dev)
    Bm = torch.randn(N, K, device=dev)
    C = torch.empty(M, N, device=dev)
    for fp16 in (False, True):
        A16, B16 = ops.cast16(A, fp16), ops.cast16(Bm, fp16)
        assert torch.equal(A16, A.to(A16.dtype)), "cast kernel differs from torch's rounding"
        ref = A16.double() @ B16.double().t() if M * N <= 4096 * 5000 else None
        line = []
        for tile in NT_TILES:
            f = lambda: ops.gemm_raw(ops.OP_NT, A16.data_ptr(), B16.data_ptr(), C.data_ptr(), M, N, K, K, K, N, h16=A16.dtype, tile=tile)
            f()
            if ref is not None:
                err = ((C.double() - ref).norm() / ref.norm()).item()
                assert err < 2e-6, (M, N, K, fp16, tile, err)
            t = timeit(f)
            line.append(f"{tile}:{t * 1e6:7.1f}us {2.0 * M * N * K / t / 1e12:6.0f}TF")
        print(f"{M:6d} {N:6d} {K:6d} {'fp16' if fp16 else 'bf16'} h16   " + "  ".join(line))
    # the fp32-image variants on the same shape (operands fp32 in HBM, rounded in registers) and plain fp32
    for prec in ("bf16", "f32"):
        line = []
        with ops.precision(prec):
            for tile in TILES:
                t = timeit(lambda: ops.gemm_raw(ops.OP_NT, A.data_ptr(), Bm.data_ptr(), C.data_ptr(), M, N, K, K, K, N, tile=tile))
                line.append(f"{tile}:{t * 1e6:7.1f}us {2.0 * M * N * K / t / 1e12:6.0f}TF")
        print(f"{M:6d} {N:6d} {K:6d} {prec:4s} image " + "  ".join(line))

print("== CONV_FWD (NHWC, B=64)")
B = 64
for (H, Cin, Cout, R, stride) in [(56, 64, 64, 3, 1), (28, 128, 128, 3, 1), (14, 256, 256, 3, 1), (7, 512, 512, 3, 1), (14, 1024, 256, 1, 1), (14, 256, 1024, 1, 1),
                                   (28, 512, 128, 1, 1), (56, 256, 64, 1, 1), (7, 2048, 512, 1, 1)]:
    pad = R // 2
    Ho = (H + 2 * pad - R) // stride + 1
    x = torch.randn(B, H, H, Cin, device=dev)
    w = torch.randn(Cout, R, R, Cin, device=dev) * 0.05
    y = torch.empty(B, Ho, Ho, Cout, device=dev)
    M, N, K = B * Ho * Ho, Cout, R * R * Cin
    conv = (B, H, H, Cin, Ho, Ho, Cout, R, R, stride, pad)
    x16, w16 = ops.cast16(x), ops.cast16(w)
    ref = torch.nn.functional.conv2d(x16.float().permute(0, 3, 1, 2).double(), w16.float().permute(0, 3, 1, 2).double(), stride=stride, padding=pad).permute(0, 2, 3, 1)
    best = {}
    for name, f in (("h16", lambda tile: ops.gemm_raw(ops.OP_CONV_FWD, x16.data_ptr(), w16.data_ptr(), y.data_ptr(), M, N, K, Cin, K, N, conv=conv, tile=tile, h16=x16.dtype)),
                    ("bf16 image", lambda tile: ops.gemm_raw(ops.OP_CONV_FWD, x.data_ptr(), w.data_ptr(), y.data_ptr(), M, N, K, Cin, K, N, conv=conv, tile=tile)),
                    ("f32", lambda tile: ops.gemm_raw(ops.OP_CONV_FWD, x.data_ptr(), w.data_ptr(), y.data_ptr(), M, N, K, Cin, K, N, conv=conv, tile=tile))):
        ctx = ops.precision("bf16" if name == "bf16 image" else "f32")
        with ctx:
            ts = {}
            for tile in TILES:
                f(tile)
                if name == "h16":
                    err = ((y.double() - ref).norm() / ref.norm()).item()
                    assert err < 2e-6, (H, Cin, Cout, R, tile, err)
                ts[tile] = timeit(lambda: f(tile))
        bt = min(ts, key=ts.get)
        best[name] = ts[bt]
        print(f"H{H:3d} Cin{Cin:5d} Cout{Cout:5d} R{R} {name:10s} best tile {bt:3d} {ts[bt] * 1e6:7.1f}us {2.0 * M * N * K / ts[bt] / 1e12:6.0f}TF")
    print(f"    h16 / bf16-image speedup {best['bf16 image'] / best['h16']:.2f}x   h16 / f32 {best['f32'] / best['h16']:.2f}x")
