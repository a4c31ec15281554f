"""k-loop experiments on the LDS-DMA fp32 kernel (igemm_glds_impl.h, -DICK_EXP=n; see the knob list there).

    python tools/ablate/run_kloop.py build 16 17 19 23      # here (hipcc cross-compiles): tools/ablate/libkl<n>.so
    gpurun -- python tools/ablate/run_kloop.py run 16 17 19 23
    gpurun -- python tools/ablate/run_kloop.py stamps 25 27  # builds with knob 8: per-workgroup phase times

Every variant is checked bit for bit against the first library of the list before it is timed."""
import ctypes, os, subprocess, sys
here = os.path.dirname(os.path.abspath(__file__))
root = os.path.dirname(os.path.dirname(here))
sys.path.insert(0, root)
C = os.path.join(root, "imagecaptioner_amd", "csrc")


def build(ns):
    procs = []
    for n in ns:
        cmd = ["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-shared", f"-DICK_EXP={n}",
               f"{C}/igemm_f32_glds.hip", f"{C}/igemm_f32.hip", f"{C}/ick_api.hip", "-o", f"{here}/libkl{n}.so"]
        procs.append((n, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        if len(procs) % 4 == 0:
            for m, p in procs[-4:]:
                out, _ = p.communicate(); print(m, "rc", p.returncode, out[-2000:] if p.returncode else "")
    for m, p in procs:
        if p.returncode is None:
            out, _ = p.communicate(); print(m, "rc", p.returncode, out[-2000:] if p.returncode else "")


B = 64
CASES = [("NT 4096^3", 0, 4096, 4096, 4096, None, 0), ("fc1 12608x1536x384 gelu", 0, 12608, 1536, 384, None, 2),
         ("fc2 12608x384x1536", 0, 12608, 384, 1536, None, 0), ("qkv 12608x1152x384", 0, 12608, 1152, 384, None, 0),
         ("proj 12608x384x384", 0, 12608, 384, 384, None, 0),
         ("conv3x3 14x14 256->256", 3, B * 14 * 14, 256, 2304, (B, 14, 14, 256, 14, 14, 256, 3, 3, 1, 1), 0),
         ("conv1x1 14x14 1024->256", 3, B * 14 * 14, 256, 1024, (B, 14, 14, 1024, 14, 14, 256, 1, 1, 1, 0), 0),
         ("conv1x1 14x14 256->1024", 3, B * 14 * 14, 1024, 256, (B, 14, 14, 256, 14, 14, 1024, 1, 1, 1, 0), 0),
         ("conv3x3 28x28 128->128", 3, B * 28 * 28, 128, 1152, (B, 28, 28, 128, 28, 28, 128, 3, 3, 1, 1), 0),
         ("conv1x1 56x56 64->256", 3, B * 56 * 56, 256, 64, (B, 56, 56, 64, 56, 56, 256, 1, 1, 1, 0), 0)]
TILES = [1, 65, 129, 3, 4, 2]


def desc(torch, IckGemm, bufs, op, M, N, K, conv, act, tile, out):
    x, w, bias, stat = bufs
    d = IckGemm(); d.A, d.B, d.C = x.data_ptr(), w.data_ptr(), out.data_ptr()
    d.op = op; d.M, d.N, d.K = M, N, K; d.lda, d.ldb, d.ldc = K, K, N; d.batch_outer = d.batch_inner = 1; d.splitk = 1
    d.alpha = 1.0; d.tile = tile; d.act = act
    if act: d.bias = bias.data_ptr()
    if conv:
        d.Nb, d.H, d.W, d.Cin, d.Ho, d.Wo, d.Cout, d.R, d.S, d.stride, d.pad = conv
        d.stat_sum, d.stat_sq = stat[0].data_ptr(), stat[1].data_ptr(); d.stat_copies = 8; d.stat_stride = 4096
    return d


def main():
    mode, ns = sys.argv[1], [int(a) for a in sys.argv[2:]]
    if mode == "build":
        return build(ns)
    import torch
    from imagecaptioner_amd._lib import IckGemm
    torch.manual_seed(0)
    x = torch.randn(1 << 26, device="cuda"); w = torch.randn(1 << 25, device="cuda") * 0.05; bias = torch.randn(8192, device="cuda")
    y = torch.empty(1 << 26, device="cuda"); yref = torch.empty(1 << 26, device="cuda")
    stat = torch.zeros(2, 8 * 4096, dtype=torch.float64, device="cuda")
    bufs = (x, w, bias, stat)
    libs = {}
    for n in ns:
        L = ctypes.CDLL(os.path.join(here, f"libkl{n}.so")); L.ick_gemm_f32.argtypes = [ctypes.c_void_p, ctypes.c_void_p]; libs[n] = L
    st = torch.cuda.current_stream().cuda_stream

    def timeit(f, iters=20):
        f(); f(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters): f()
        b.record(); torch.cuda.synchronize()
        return a.elapsed_time(b) / iters * 1e-3

    if mode == "run":
        print("times in us (TF/s); columns = ICK_EXP", ns, flush=True)
        ytile = torch.empty(1 << 26, device="cuda")
        for name, op, M, N, K, conv, act in CASES:
            for ti, tile in enumerate(TILES):
                line = f"{name:26s} tile {tile:3d}"
                if len(ns) == 1:      # one library: every tile's result against the first tile's (fp32 reorder noise only)
                    d0 = desc(torch, IckGemm, bufs, op, M, N, K, conv, act, tile, ytile if ti == 0 else y)
                    assert libs[ns[0]].ick_gemm_f32(ctypes.byref(d0), st) == 0
                    torch.cuda.synchronize()
                    if ti > 0:
                        e = ((y[:M * N] - ytile[:M * N]).abs().max() / ytile[:M * N].abs().max()).item()
                        line += f" [vs tile {TILES[0]}: {e:.1e}{'' if e < 1e-5 else ' ERR'}]"
                for j, n in enumerate(ns):
                    L = libs[n]
                    d = desc(torch, IckGemm, bufs, op, M, N, K, conv, act, tile, yref if j == 0 else y)
                    rc = L.ick_gemm_f32(ctypes.byref(d), st); assert rc == 0, (name, n, rc)
                    torch.cuda.synchronize()
                    same = ""
                    if j > 0:
                        same = " ==" if torch.equal(y[:M * N], yref[:M * N]) else f" !!{(y[:M * N] - yref[:M * N]).abs().max().item():.2e}"
                    t = timeit(lambda: L.ick_gemm_f32(ctypes.byref(d), st))
                    line += f"  {t * 1e6:7.1f} ({2.0 * M * N * K / t / 1e12:5.1f}){same}"
                print(line, flush=True)
        return
    if mode == "stagger":          # libraries built with knob 32: initial delay of the odd-wave-slot workgroups, in microseconds
        L = libs[ns[0]]
        print("stagger sweep, ICK_EXP", ns[0], "; columns = delay in us of the odd-slot workgroups of the first round", flush=True)
        delays = [0, 3, 6, 9, 12, 16, 20, 26]
        for name, op, M, N, K, conv, act in CASES:
            for tile in (1, 65, 4, 2):
                d = desc(torch, IckGemm, bufs, op, M, N, K, conv, act, tile, y)
                line = f"{name:26s} tile {tile:3d}"
                for us_ in delays:
                    assert L.ick_exp_set_stagger(int(us_ * 2100 / 64)) == 0      # ~2.1 GHz under load
                    t = timeit(lambda: L.ick_gemm_f32(ctypes.byref(d), st))
                    line += f"  {us_:2d}us {t * 1e6:6.1f}"
                print(line, flush=True)
        return
    # stamps
    import numpy as np
    for n in ns:
        L = libs[n]
        for stag_us in ((0, 10) if hasattr(L, "ick_exp_set_stagger") else (0,)):
          if hasattr(L, "ick_exp_set_stagger"):
              assert L.ick_exp_set_stagger(int(stag_us * 2100 / 64)) == 0
              print(f"==== stagger {stag_us} us", flush=True)
          for name, op, M, N, K, conv, act in CASES[:3] + CASES[5:6]:
            for tile in (1, 65, 129):
                  dbg = torch.zeros(16 * 65536, dtype=torch.int64, device="cuda")
                  assert L.ick_exp_set_dbg(ctypes.c_void_p(dbg.data_ptr())) == 0
                  d = desc(torch, IckGemm, bufs, op, M, N, K, conv, act, tile, y)
                  for _ in range(3):
                      L.ick_gemm_f32(ctypes.byref(d), st)
                  torch.cuda.synchronize()
                  a = dbg.cpu().numpy().reshape(-1, 16)
                  a = a[a[:, 0] != 0]
                  t0 = a[:, 0].min()
                  clk = 100e6                                           # s_memrealtime: 100 MHz
                  us = lambda v: v / clk * 1e6
                  pro, kl, ep = us(a[:, 1] - a[:, 0]), us(a[:, 2] - a[:, 1]), us(a[:, 3] - a[:, 2])
                  start, end = us(a[:, 0] - t0), us(a[:, 3] - t0)
                  if (a[:, 4] != 0).all():
                      e1, e2, e3 = us(a[:, 4] - a[:, 2]), us(a[:, 5] - a[:, 4]), us(a[:, 3] - a[:, 5])
                      print(f"      epilogue split (thread 0): barrier + registers->LDS {np.median(e1):.2f} (p90 {np.percentile(e1, 90):.2f})  "
                            f"second barrier {np.median(e2):.2f} (p90 {np.percentile(e2, 90):.2f})  LDS->global stores issued "
                            f"{np.median(e3):.2f} (p90 {np.percentile(e3, 90):.2f}) us", flush=True)
                  if (a[:, 13] != 0).any():
                      kk = a[a[:, 13] != 0]
                      seg = [np.median(kk[:, 9 + i] - kk[:, 8 + i]) for i in range(5)]
                      print("      one k-tile, wave 0, shader cycles: wait own DMA %d | barrier %d | DMA issue %d | LDS reads + first MFMA group "
                            "%d | remaining MFMA groups %d  (sum %d; 64 MFMAs = 4096 at one wave per SIMD)" % (*seg, sum(seg)), flush=True)
                  hw, xcc = a[:, 6], a[:, 7]
                  cu = ((xcc & 15) << 8) | (((hw >> 13) & 7) << 5) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 15)
                  print(f"EXP {n} {name} tile {tile}: {len(a)} workgroups on {len(np.unique(cu))} CUs, span {end.max():.1f} us; per-workgroup "
                        f"prologue {np.median(pro):.2f} (p90 {np.percentile(pro, 90):.2f})  k-loop {np.median(kl):.2f} (p10 {np.percentile(kl, 10):.2f} "
                        f"p90 {np.percentile(kl, 90):.2f})  epilogue {np.median(ep):.2f} (p90 {np.percentile(ep, 90):.2f}) us", flush=True)
                  # per CU: how much of the span has 0 / 1 / 2+ workgroups inside their k-loops
                  grid = np.linspace(0, end.max(), 2001)
                  k0, k1 = us(a[:, 1] - t0), us(a[:, 2] - t0)
                  occ = np.zeros((3,))
                  for c in np.unique(cu):
                      m = cu == c
                      cnt = ((k0[m][:, None] <= grid[None, :]) & (grid[None, :] < k1[m][:, None])).sum(0)
                      for v in range(3):
                          occ[v] += (np.minimum(cnt, 2) == v).mean()
                  occ /= len(np.unique(cu))
                  rounds = np.sort(start)
                  print(f"      share of the span with 0 / 1 / 2 workgroups of a CU in their k-loop: {occ[0]:.2f} / {occ[1]:.2f} / {occ[2]:.2f};"
                        f" starts at {np.percentile(start, [0, 25, 50, 75, 100]).round(1)} us", flush=True)


if __name__ == "__main__":
    main()
