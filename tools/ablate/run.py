import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from imagecaptioner_amd._lib import IckGemm
here = os.path.dirname(os.path.abspath(__file__))
shapes = [(4096, 4096, 4096), (12608, 1536, 384), (12608, 384, 1536), (12544, 256, 2304)]
x = torch.randn(1 << 26, device="cuda"); w = torch.randn(1 << 26, device="cuda"); y = torch.empty(1 << 26, device="cuda")
for tile in (1, 2):
  for (M, N, K) in shapes:
    line = f"tile {'128x128' if tile == 1 else '64x64'} {M}x{N}x{K}: "
    for n in range(5):
        L = ctypes.CDLL(os.path.join(here, f"libabl{n}.so"))
        L.ick_gemm_f32.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        d = IckGemm(); d.A, d.B, d.C = x.data_ptr(), w.data_ptr(), y.data_ptr()
        d.op = 0; d.M, d.N, d.K = M, N, K; d.lda, d.ldb, d.ldc = K, K, N; d.batch_outer = d.batch_inner = 1; d.splitk = 1; d.alpha = 1.0; d.tile = tile
        st = torch.cuda.current_stream().cuda_stream
        f = lambda: L.ick_gemm_f32(ctypes.byref(d), st)
        assert f() == 0; f(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10): f()
        b.record(); torch.cuda.synchronize()
        t = a.elapsed_time(b) / 10 * 1e-3
        line += f" V{n} {2.0 * M * N * K / t / 1e12:6.1f}TF"
    print(line, flush=True)
