"""CPU diagnostic: rounding error of a length-K fp32 dot product accumulated (a) the way torch's CPU GEMM does, (b) as one
sequential chain (what one MFMA accumulator does over the whole K loop), (c) as chains of `chunk` k folded into a master sum."""
import torch, sys
torch.manual_seed(0)
for K in (256, 1152, 2048, 4608):
    M = N = 128
    a = torch.randn(M, K).abs() * torch.randn(M, K).sign()   # relu-ish operands: mostly same-sign products make sums grow
    a = torch.relu(torch.randn(M, K))
    b = torch.randn(K, N) * (2.0 / K) ** 0.5
    ref = a.double() @ b.double()
    err = lambda x: ((x.double() - ref).norm() / ref.norm()).item()
    seq = torch.zeros(M, N)
    for k in range(K):
        seq = torch.addcmul(seq, a[:, k:k + 1], b[k:k + 1, :])     # rounded product + rounded add (fma would be slightly better)
    def chunked(c):
        tot = torch.zeros(M, N)
        for k0 in range(0, K, c):
            acc = torch.zeros(M, N)
            for k in range(k0, min(K, k0 + c)):
                acc = torch.addcmul(acc, a[:, k:k + 1], b[k:k + 1, :])
            tot = tot + acc
        return tot
    print(f"K={K:5d} torch.mm {err(a @ b):.2e}  sequential {err(seq):.2e}  chunk32 {err(chunked(32)):.2e} chunk128 {err(chunked(128)):.2e} chunk256 {err(chunked(256)):.2e}")
