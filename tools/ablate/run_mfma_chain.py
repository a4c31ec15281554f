"""tools/ablate/mfma_chain.hip on the GPU box: matrix-pipe rate against the dependency pattern of the MFMA stream."""
import ctypes, os
import torch
here = os.path.dirname(os.path.abspath(__file__))
L = ctypes.CDLL(os.path.join(here, "libmfma.so"))
L.chain_run.restype = ctypes.c_long
L.chain_run.argtypes = [ctypes.c_int] * 3 + [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
out = torch.zeros(1024, device="cuda")
st = torch.cuda.current_stream().cuda_stream
iters = 4000
for f16 in (0, 1):
    flop = 32 * 32 * (16 if f16 else 2) * 2
    for blocks, label in ((256, "1 wave / SIMD"), (512, "2 waves / SIMD"), (1024, "4 waves / SIMD")):
        for nacc, run in ((1, 1), (2, 1), (4, 1), (8, 1), (1, 4), (2, 2), (2, 4), (4, 4)):
            f = lambda: L.chain_run(nacc, run, f16, out.data_ptr(), blocks, iters, st)
            per = f(); torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); f(); f(); b.record(); torch.cuda.synchronize()
            t = a.elapsed_time(b) / 2 * 1e-3
            tf = blocks * 4 * iters * per * flop / t / 1e12
            print(f"{'f16 32x32x16' if f16 else 'f32 32x32x2 '} {label:15s} accumulators {nacc} run {run}: {tf:7.1f} TF")
