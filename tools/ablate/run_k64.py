"""tools/ablate/k64_stream.hip on the GPU box: what bounds the K = 64 write-bound product (see the .hip header)."""
import ctypes, os, sys
import torch
here = os.path.dirname(os.path.abspath(__file__))
L = ctypes.CDLL(os.path.join(here, "libk64.so"))
L.k64_run.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 3 + [ctypes.c_int] * 2 + [ctypes.c_void_p] * 2 + [ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
M = 200704
names = {0: "full", 1: "no statistics", 2: "no stores", 3: "non-temporal stores", 4: "stores, no MFMAs"}
for N in (256, 64):
    x = torch.randn(M, 64, device="cuda"); w = torch.randn(N, 64, device="cuda") * 0.1
    y = torch.empty(M, N, device="cuda"); st = torch.zeros(2, 8, N, dtype=torch.float64, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    for cap in (512, 1024, 2048):
        for mode in range(5):
            f = lambda: L.k64_run(mode, x.data_ptr(), w.data_ptr(), y.data_ptr(), M, N, st[0].data_ptr(), st[1].data_ptr(), 8, s, cap)
            f(); f(); torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(30): f()
            b.record(); torch.cuda.synchronize()
            t = a.elapsed_time(b) / 30 * 1e3
            print(f"N {N:3d} grid cap {cap:4d} mode {mode} ({names[mode]:20s}): {t:7.1f} us  {(M * N * 4) / t / 1e3:6.0f} GB/s of output")
    # reference points: a pure fill and a pure copy of the output's size
    for nm, f in (("fill (torch zero_)", lambda: y.zero_()), ("copy (torch copy_)", lambda: y.copy_(y2))):
        y2 = torch.empty_like(y)
        f(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(30): f()
        b.record(); torch.cuda.synchronize()
        t = a.elapsed_time(b) / 30 * 1e3
        print(f"N {N:3d} {nm}: {t:7.1f} us  {(M * N * 4) / t / 1e3:6.0f} GB/s written")
