"""A/B of the LDS-DMA kernel with parts compiled out (tools/ablate/libglds{0..3}.so built with -DICK_ABL=n):
0 full, 1 no BN-stat atomics, 2 + no C stores, 3 + no DMA after the prologue.  Wrong results by construction; times only."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from imagecaptioner_amd._lib import IckGemm
here = os.path.dirname(os.path.abspath(__file__))
x = torch.randn(1 << 27, device="cuda"); w = torch.randn(1 << 26, device="cuda"); y = torch.empty(1 << 27, device="cuda")
stat = torch.zeros(2, 1 << 20, dtype=torch.float64, device="cuda")
B = 64
cases = [("NT 4096^3", 0, 4096, 4096, 4096, None, 1), ("NT 12608x1536x384", 0, 12608, 1536, 384, None, 2), ("NT 12608x384x1536", 0, 12608, 384, 1536, None, 2),
         ("conv1x1 56x56 64->256", 3, B * 56 * 56, 256, 64, (B, 56, 56, 64, 56, 56, 256, 1, 1, 1, 0), 3),
         ("conv1x1 56x56 256->64", 3, B * 56 * 56, 64, 256, (B, 56, 56, 256, 56, 56, 64, 1, 1, 1, 0), 3),
         ("conv3x3 14x14 256->256", 3, B * 14 * 14, 256, 2304, (B, 14, 14, 256, 14, 14, 256, 3, 3, 1, 1), 4),
         ("conv1x1 28x28 128->512", 3, B * 28 * 28, 512, 128, (B, 28, 28, 128, 28, 28, 512, 1, 1, 1, 0), 3),
         ("dgrad3x3 14x14 256->256", 5, B * 14 * 14, 256, 2304, (B, 14, 14, 256, 14, 14, 256, 3, 3, 1, 1), 4)]
libs = []
for n in (0, 1, 2, 3, 4, 5):
    L = ctypes.CDLL(os.path.join(here, f"libglds{n}.so")); L.ick_gemm_f32.argtypes = [ctypes.c_void_p, ctypes.c_void_p]; libs.append(L)
for name, op, M, N, K, conv, tile in cases:
    line = f"{name:26s}"
    for n, L in zip((0, 1, 2, 3, 4, 5), libs):
        d = IckGemm(); d.A, d.B, d.C = x.data_ptr(), w.data_ptr(), y.data_ptr()
        d.op = op; d.M, d.N, d.K = M, N, K; d.lda, d.ldb, d.ldc = K, K, N; d.batch_outer = d.batch_inner = 1; d.splitk = 1; d.alpha = 1.0; d.tile = tile
        if conv:
            d.Nb, d.H, d.W, d.Cin, d.Ho, d.Wo, d.Cout, d.R, d.S, d.stride, d.pad = conv
            d.stat_sum, d.stat_sq = stat[0].data_ptr(), stat[1].data_ptr()
            if op == 5: d.stat_sum = d.stat_sq = None; d.lda = d.ldb = 0
        st = torch.cuda.current_stream().cuda_stream
        f = lambda: L.ick_gemm_f32(ctypes.byref(d), st)
        rc = f(); assert rc == 0, (name, n, rc); f(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10): f()
        b.record(); torch.cuda.synchronize()
        t = a.elapsed_time(b) / 10 * 1e-3
        line += f"  V{n} {t * 1e6:7.1f}us {2.0 * M * N * K / t / 1e12:6.1f}TF"
    if conv and op == 3:
        for copies in (8, 64, 512):
            L = libs[0]
            d = IckGemm(); d.A, d.B, d.C = x.data_ptr(), w.data_ptr(), y.data_ptr()
            d.op = op; d.M, d.N, d.K = M, N, K; d.lda, d.ldb, d.ldc = K, K, N; d.batch_outer = d.batch_inner = 1; d.splitk = 1; d.alpha = 1.0; d.tile = tile
            d.Nb, d.H, d.W, d.Cin, d.Ho, d.Wo, d.Cout, d.R, d.S, d.stride, d.pad = conv
            d.stat_sum, d.stat_sq = stat[0].data_ptr(), stat[1].data_ptr(); d.stat_copies = copies; d.stat_stride = N
            st = torch.cuda.current_stream().cuda_stream
            f = lambda: L.ick_gemm_f32(ctypes.byref(d), st)
            assert f() == 0; f(); torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(10): f()
            b.record(); torch.cuda.synchronize()
            line += f"  copies{copies} {a.elapsed_time(b) / 10 * 1e3:7.1f}us"
    print(line, flush=True)
