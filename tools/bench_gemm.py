#!/usr/bin/env python3
"""Micro-benchmark of the igemm_f32 family on the shapes of the KD step (runs on the GPU box).
Prints TFLOP/s per shape against the 157.3 TF fp32-MFMA peak."""
import sys
import os
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imagecaptioner_amd import ops  # noqa: E402

PEAK = 157.3


def timeit(fn, iters=20):
    fn(); fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def main():
    prec = sys.argv[1] if len(sys.argv) > 1 else "f32"
    ops.set_gemm_precision(prec)
    print(f"precision {prec}")
    B = 64
    rows = []
    for (M, N, K) in [(4096, 4096, 4096), (12608, 1152, 384), (12608, 384, 384), (12608, 1536, 384), (12608, 384, 1536),
                      (12608, 1024, 512), (960, 5000, 256), (960, 5000, 512), (3136, 256, 2048), (64, 2048, 512)]:
        x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda"); y = torch.empty(M, N, device="cuda")
        t = timeit(lambda: ops.linear_fwd(x, w, out=y))
        rows.append((f"NT {M}x{N}x{K}", 2 * M * N * K / t / 1e12, t))
        dy = torch.randn(M, N, device="cuda")
        t = timeit(lambda: ops.linear_bwd_data(dy, w))
        rows.append((f"NN {M}x{K}x{N}", 2 * M * N * K / t / 1e12, t))
        dw = torch.zeros(N, K, device="cuda")
        t = timeit(lambda: ops.linear_bwd_weight(dy, x, dw))
        rows.append((f"TN {N}x{K}x{M}", 2 * M * N * K / t / 1e12, t))
    convs = [(B, 224, 224, 4, 64, 7, 2, 3), (B, 56, 56, 64, 64, 1, 1, 0), (B, 56, 56, 64, 64, 3, 1, 1), (B, 56, 56, 64, 256, 1, 1, 0),
             (B, 56, 56, 256, 64, 1, 1, 0), (B, 56, 56, 128, 128, 3, 2, 1), (B, 28, 28, 128, 128, 3, 1, 1),
             (B, 28, 28, 512, 128, 1, 1, 0), (B, 28, 28, 128, 512, 1, 1, 0), (B, 14, 14, 256, 256, 3, 1, 1),
             (B, 14, 14, 1024, 256, 1, 1, 0), (B, 14, 14, 256, 1024, 1, 1, 0), (B, 7, 7, 512, 512, 3, 1, 1),
             (B, 7, 7, 2048, 512, 1, 1, 0), (B, 7, 7, 512, 2048, 1, 1, 0)]
    for (Nb, H, W, Cin, Cout, R, st, pad) in convs:
        x = torch.randn(Nb, H, W, Cin, device="cuda"); w = torch.randn(Cout, R, R, Cin, device="cuda") * 0.05
        stats = torch.zeros(2, Cout, device="cuda", dtype=torch.float64)
        Ho, Wo = ops.conv_out_hw(H, W, R, R, st, pad)
        fl = 2.0 * Nb * Ho * Wo * Cout * R * R * (3 if Cin == 4 else Cin)
        t = timeit(lambda: ops.conv_fwd(x, w, st, pad, stats=(stats[0], stats[1])))
        rows.append((f"conv fwd {H}x{W} {Cin}->{Cout} k{R}s{st}", fl / t / 1e12, t))
        if Cin >= 256 or (Cin == 128 and H <= 28):
            y = torch.randn(Nb, Ho, Wo, Cout, device="cuda")
            t = timeit(lambda: ops.conv_dgrad(y, w, (H, W), st, pad))
            rows.append((f"conv dgrad {H}x{W} {Cin}->{Cout} k{R}s{st}", fl / t / 1e12, t))
            dw = torch.zeros_like(w)
            t = timeit(lambda: ops.conv_wgrad(y, x, dw, st, pad))
            rows.append((f"conv wgrad {H}x{W} {Cin}->{Cout} k{R}s{st}", fl / t / 1e12, t))
    for name, tf, t in rows:
        print(f"{name:44s} {tf:7.1f} TF/s  {100 * tf / PEAK:5.1f}% of fp32 MFMA peak   {t * 1e6:9.1f} us")


if __name__ == "__main__":
    main()
