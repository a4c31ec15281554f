"""GPU parity of the MFMA implicit-GEMM families (csrc/igemm_f32.hip, csrc/igemm_bf16.hip) through the C ABI,
against float64 CPU references built from torch primitives.  Tolerances, relative to the output scale
(max |err| / max |ref|):
  f32     2e-5  exact-fp32 MFMA = fmaf chain; the reference path is fp32 too, so only summation order differs
  bf16x3  2e-4  split-bf16 (hi*hi + hi*lo + lo*hi): ~2^-16 per product
  bf16    1.5e-2  operands rounded to 8 significant bits, fp32 accumulate
  fp16    2e-3    operands rounded to 11 significant bits (v_mfma_f32_32x32x16_f16: the reference's autocast dtype), fp32 accumulate
  f32x3   2e-5  fp32 by three fp16 products (forward Linear / convolution; every other op IS the exact-fp32 kernel): held to the
                exact-fp32 tolerance, and to the exact-fp32 kernel's own error against float64 in test_f32x3_is_fp32_grade"""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

TOLS = {"f32": 2e-5, "bf16x3": 2e-4, "bf16": 1.5e-2, "fp16": 2e-3, "f32x3": 2e-5}
TOL = 2e-5          # rebound per test by the `ops` fixture


def rel_err(a, b):
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


@pytest.fixture(params=["f32", "bf16", "bf16x3", "fp16", "f32x3"])
def ops(request):
    from imagecaptioner_amd import ops as o
    global TOL
    TOL = TOLS[request.param]
    with o.precision(request.param):
        yield o
    TOL = TOLS["f32"]


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed + sum(shape))
    return (torch.randn(*shape, generator=g) * scale)


@pytest.mark.parametrize("M,N,K", [(64, 64, 16), (128, 128, 32), (197, 5000, 72), (960, 256, 512), (3136, 256, 2048),
                                   (33, 7, 20), (1, 1, 4), (300, 384, 1536)])
def test_linear_fwd_bwd(ops, M, N, K):
    x, w, b, r = rnd(M, K, seed=1), rnd(N, K, seed=2), rnd(N, seed=3), rnd(M, N, seed=4)
    xd, wd, bd, rd = (t.cuda() for t in (x, w, b, r))
    for act, fn in ((0, lambda v: v), (1, torch.relu), (2, F.gelu), (3, torch.tanh)):
        y = ops.linear_fwd(xd, wd, bd, act=act, residual=rd)
        pre = x.double() @ w.double().T + b.double()
        ref = fn(pre) + r.double()
        # the rounding error lives on the pre-activation's scale (every activation here is 1-Lipschitz)
        scale = max(1.0, (pre.abs().max() / ref.abs().max()).item()) if ops.gemm_precision() != "f32" else 1.0
        assert rel_err(y, ref) < TOL * scale, f"fwd act={act}"
    dy = rnd(M, N, seed=5)
    dyd = dy.cuda()
    if K % 4 == 0 and N % 4 == 0:
        dx = ops.linear_bwd_data(dyd, wd)
        assert rel_err(dx, dy.double() @ w.double()) < TOL
        for sk in (1, 3):
            dw = torch.ones(N, K, device="cuda")
            ops.linear_bwd_weight(dyd, xd, dw, splitk=sk)
            assert rel_err(dw, 1.0 + dy.double().T @ x.double()) < TOL, f"wgrad splitk={sk}"
        db = torch.zeros(N, device="cuda")
        ops.colsum_into(dyd, db)
        assert rel_err(db, dy.double().sum(0)) < TOL


def test_mfma_layout_asymmetric(ops):
    """A = I against an asymmetric B: catches a transposed C write (guide §3)."""
    n = 128
    A = torch.eye(n)
    Bm = torch.arange(n * n, dtype=torch.float32).reshape(n, n) / 997.0   # B[i][j] != B[j][i]
    if ops.gemm_precision() != "f32":
        Bm = (torch.arange(n * n) % 251).float().reshape(n, n)            # exactly representable in bf16
    y = ops.linear_fwd(A.cuda(), Bm.cuda())         # y = A @ B^T = B^T
    assert torch.equal(y.cpu(), Bm.T.contiguous())


@pytest.mark.parametrize("B,H,Lq,Lk,d,causal", [(2, 6, 197, 197, 64, False), (3, 4, 49, 49, 64, False),
                                                (2, 8, 15, 15, 64, True), (2, 8, 15, 197, 64, False)])
def test_attention_core(ops, B, H, Lq, Lk, d, causal):
    E = H * d
    q = rnd(B * Lq, 3 * E, seed=1)
    kv = rnd(B * Lk, 3 * E, seed=2) if Lk != Lq else q
    qd, kvd = q.cuda(), kv.cuda()
    O, P = ops.attention_fwd(qd, 0, 3 * E, kvd, E, 3 * E, kvd, 2 * E, 3 * E, B, H, Lq, Lk, d, causal)
    qq = q[:, :E].double().view(B, Lq, H, d).transpose(1, 2).requires_grad_(True)
    kk = kv[:, E:2 * E].double().view(B, Lk, H, d).transpose(1, 2).requires_grad_(True)
    vv = kv[:, 2 * E:].double().view(B, Lk, H, d).transpose(1, 2).requires_grad_(True)
    s = qq @ kk.transpose(-1, -2) / math.sqrt(d)
    if causal:
        s = s.masked_fill(torch.ones(Lq, Lk, dtype=torch.bool).triu(1), float("-inf"))
    p = torch.softmax(s, -1)
    o = (p @ vv).transpose(1, 2).reshape(B * Lq, E)
    assert rel_err(P[..., :Lk], p) < TOL
    assert P[..., Lk:].abs().max().item() == 0 if P.shape[-1] > Lk else True
    assert rel_err(O, o) < TOL
    dO = rnd(B * Lq, E, seed=3)
    (o * dO.double()).sum().backward()
    dq = torch.zeros(B * Lq, 3 * E, device="cuda")
    dkv = torch.zeros(B * Lk, 3 * E, device="cuda") if Lk != Lq else dq
    ops.attention_bwd(dO.cuda(), P, qd, 0, 3 * E, kvd, E, 3 * E, kvd, 2 * E, 3 * E,
                      dq, 0, 3 * E, dkv, E, 3 * E, dkv, 2 * E, 3 * E, B, H, Lq, Lk, d)
    assert rel_err(dq[:, :E], qq.grad.transpose(1, 2).reshape(B * Lq, E)) < 2.5 * TOL
    assert rel_err(dkv[:, E:2 * E], kk.grad.transpose(1, 2).reshape(B * Lk, E)) < 2.5 * TOL
    assert rel_err(dkv[:, 2 * E:], vv.grad.transpose(1, 2).reshape(B * Lk, E)) < 2.5 * TOL


CONVS = [  # (Nb, H, W, Cin, Cout, R, stride, pad) — every distinct ResNet-50 conv geometry at reduced batch
    (2, 56, 56, 64, 64, 1, 1, 0), (2, 56, 56, 64, 64, 3, 1, 1), (2, 56, 56, 64, 256, 1, 1, 0),
    (2, 56, 56, 256, 128, 1, 1, 0), (2, 56, 56, 128, 128, 3, 2, 1), (2, 56, 56, 256, 512, 1, 2, 0),
    (2, 28, 28, 512, 256, 1, 1, 0), (2, 28, 28, 256, 256, 3, 2, 1), (2, 14, 14, 256, 256, 3, 1, 1),
    (2, 14, 14, 1024, 2048, 1, 2, 0), (3, 14, 14, 512, 512, 3, 2, 1), (3, 7, 7, 512, 512, 3, 1, 1),
    (3, 7, 7, 2048, 512, 1, 1, 0),
]


@pytest.mark.parametrize("Nb,H,W,Cin,Cout,R,stride,pad", CONVS)
def test_conv_fwd_dgrad_wgrad(ops, Nb, H, W, Cin, Cout, R, stride, pad):
    x = rnd(Nb, Cin, H, W, seed=1)
    w = rnd(Cout, Cin, R, R, seed=2, scale=1.0 / math.sqrt(Cin * R * R))
    xd = x.cuda().permute(0, 2, 3, 1).contiguous()          # NHWC
    wd = w.cuda().permute(0, 2, 3, 1).contiguous()          # (Cout,R,S,Cin)
    stats = torch.zeros(2, Cout, device="cuda", dtype=torch.float64)
    y = ops.conv_fwd(xd, wd, stride, pad, stats=(stats[0], stats[1]))
    ref = F.conv2d(x.double(), w.double(), None, stride, pad)
    assert rel_err(y.permute(0, 3, 1, 2), ref) < TOL
    assert rel_err(stats[0], ref.sum((0, 2, 3))) < max(1e-4, 5 * TOL) * max(1.0, ref.abs().sum((0, 2, 3)).max().item() / ref.sum((0, 2, 3)).abs().max().item())
    assert rel_err(stats[1], (ref * ref).sum((0, 2, 3))) < max(1e-4, 5 * TOL)
    # eval-mode conv + BatchNorm + residual + ReLU folded into the epilogue: relu(scale*conv + shift + res)
    sc, sh = rnd(Cout, seed=7).abs() + 0.5, rnd(Cout, seed=8)
    resy = rnd(*ref.shape, seed=9)
    yf = ops.conv_fwd(xd, wd, stride, pad, scale=sc.cuda(), shift=sh.cuda(), residual=resy.cuda().permute(0, 2, 3, 1).contiguous(), relu=True)
    reff = torch.relu(ref * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1) + resy.double())
    assert rel_err(yf.permute(0, 3, 1, 2), reff) < TOL * max(1.0, (ref.abs().max() * sc.max() / reff.abs().max()).item())
    dy = rnd(*ref.shape, seed=3)
    dyd = dy.cuda().permute(0, 2, 3, 1).contiguous()
    res = rnd(Nb, H, W, Cin, seed=4).cuda()
    dx = ops.conv_dgrad(dyd, wd, (H, W), stride, pad, residual=res)
    dx_ref = torch.nn.grad.conv2d_input(x.shape, w.double(), dy.double(), stride, pad) + res.cpu().double().permute(0, 3, 1, 2)
    assert rel_err(dx.permute(0, 3, 1, 2), dx_ref) < TOL
    dw_ref = torch.nn.grad.conv2d_weight(x.double(), w.shape, dy.double(), stride, pad)
    for sk in (1, 0):
        dw = torch.zeros_like(wd)
        ops.conv_wgrad(dyd, xd, dw, stride, pad, splitk=sk)
        assert rel_err(dw.permute(0, 3, 1, 2), dw_ref) < 2.5 * TOL, f"wgrad splitk={sk}"


def test_conv_stem_c4(ops):
    x = rnd(2, 3, 224, 224, seed=1)
    w = rnd(64, 3, 7, 7, seed=2, scale=0.1)
    x4 = ops.nchw3_to_nhwc4(x.cuda())
    assert torch.equal(x4[..., :3].cpu(), x.permute(0, 2, 3, 1)) and x4[..., 3].abs().max().item() == 0
    w4 = torch.zeros(64, 7, 7, 4, device="cuda")
    w4[..., :3] = w.cuda().permute(0, 2, 3, 1)
    stats = torch.zeros(2, 64, device="cuda", dtype=torch.float64)
    y = ops.conv_fwd(x4, w4, 2, 3, stats=(stats[0], stats[1]))
    ref = F.conv2d(x.double(), w.double(), None, 2, 3)
    assert rel_err(y.permute(0, 3, 1, 2), ref) < TOL
    assert rel_err(stats[1], (ref * ref).sum((0, 2, 3))) < max(1e-4, 5 * TOL)
    mp = ops.maxpool3x3s2(y)
    assert torch.equal(mp.permute(0, 3, 1, 2).cpu(), F.max_pool2d(y.permute(0, 3, 1, 2).cpu(), 3, 2, 1))


@pytest.mark.parametrize("tile", [34, 35, 36, 50, 0, 65, 67, 83, 65 + 32, 129])
def test_msplit_dispatch_linear_and_conv_vs_fp64(tile):
    """IckGemm.tile +32: rows that fill whole rounds of the chip run on 128x128 workgroups, the remaining rows in a second
    launch with a smaller tile (tile 0 lets the library's cost model decide).  Both launches must cover every row exactly
    once — bias / GELU / residual, BatchNorm statistics and a ragged M included."""
    from imagecaptioner_amd import ops as o
    from imagecaptioner_amd._lib import ACT_GELU, OP_NT
    M, N, K = 64 * 197, 1536, 96                      # the ViT fc1 geometry with a short K: 99 x 12 tiles = 2.32 rounds
    x, w, b, r = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.1), rnd(N, seed=3), rnd(M, N, seed=4)
    xd, wd, bd, rd = (t.cuda() for t in (x, w, b, r))
    y = torch.zeros(M, N, device="cuda")
    o.gemm_raw(OP_NT, xd.data_ptr(), wd.data_ptr(), y.data_ptr(), M, N, K, K, K, N, bias=bd.data_ptr(), act=ACT_GELU,
               residual=rd.data_ptr(), ldr=N, tile=tile)
    ref = F.gelu(x.double() @ w.double().T + b.double()) + r.double()
    assert rel_err(y, ref) < 2e-5
    # convolution + statistics: 64 x 14 x 14 pixels, 256 -> 1024 channels: 98 x 8 tiles = 1.53 rounds
    xc, wc = rnd(64, 14, 14, 256, seed=5), rnd(1024, 1, 1, 256, seed=6, scale=0.05)
    stats = torch.zeros(2, 1024, dtype=torch.float64, device="cuda")
    o._FORCE_TILE[0] = tile
    try:
        raw = o.conv_fwd(xc.cuda(), wc.cuda(), 1, 0, stats=(stats[0], stats[1]))
    finally:
        o._FORCE_TILE[0] = 0
    refc = xc.double().view(-1, 256) @ wc.double().view(1024, 256).T
    assert rel_err(raw.view(-1, 1024), refc) < 2e-5
    assert rel_err(stats[0], refc.sum(0)) < 1e-5 and rel_err(stats[1], (refc * refc).sum(0)) < 1e-5


def test_full_size_step_shapes_with_their_tuned_tiles_vs_fp64():
    """The bench's own instantiations (B = 64 shapes whose tile ids come from tuned_tiles.json, BatchNorm statistics spread
    over 8 accumulator copies, chunked accumulation over K = 2304) against float64 CPU references — VERDICT r01 weak item 2:
    the small-batch parity tests exercise other tile templates than the bench does."""
    from imagecaptioner_amd import ops as o
    from imagecaptioner_amd._lib import ACT_GELU
    B = 64
    # (1) layer1 conv3: 1x1, 64 -> 256 channels at 56x56 (M = 200704, K = 64), statistics in 8 copies
    x = rnd(B, 56, 56, 64, seed=1)
    w = rnd(256, 1, 1, 64, seed=2, scale=0.1)
    M = B * 56 * 56
    R = o.stat_copies(M)
    assert R == 8 and f"3:{M}:256:64:1:1" in o._TUNED
    stats = torch.zeros(2, R, 256, dtype=torch.float64, device="cuda")
    raw = o.conv_fwd(x.cuda(), w.cuda(), 1, 0, stats=(stats[0], stats[1]))
    ref = x.double().view(M, 64) @ w.double().view(256, 64).T
    assert rel_err(raw.view(M, 256), ref) < 2e-5
    assert rel_err(stats[0].sum(0), ref.sum(0)) < 1e-5 and rel_err(stats[1].sum(0), (ref * ref).sum(0)) < 1e-5
    # (2) layer3 conv2: 3x3, 256 -> 256 at 14x14 (M = 12544, K = 2304): forward and data gradient
    x3 = rnd(B, 14, 14, 256, seed=3)
    w3 = rnd(256, 3, 3, 256, seed=4, scale=0.05)
    y3 = o.conv_fwd(x3.cuda(), w3.cuda(), 1, 1)
    xr = x3.double().permute(0, 3, 1, 2)
    wr = w3.double().permute(0, 3, 1, 2)
    ref3 = F.conv2d(xr, wr, padding=1)
    assert rel_err(y3.permute(0, 3, 1, 2), ref3) < 2e-5
    dy = rnd(B, 14, 14, 256, seed=5)
    dx = o.conv_dgrad(dy.cuda(), w3.cuda(), (14, 14), 1, 1)
    refdx = torch.nn.grad.conv2d_input((B, 256, 14, 14), wr, dy.double().permute(0, 3, 1, 2), padding=1)
    assert rel_err(dx.permute(0, 3, 1, 2), refdx) < 2e-5
    # (3) its weight gradient (split-K over M = 12544 rows, fp32 atomics)
    dw = torch.zeros(256, 3, 3, 256, device="cuda")
    o.conv_wgrad(dy.cuda(), x3.cuda(), dw, 1, 1)
    refdw = torch.nn.grad.conv2d_weight(xr, (256, 256, 3, 3), dy.double().permute(0, 3, 1, 2), padding=1)
    assert rel_err(dw.permute(0, 3, 1, 2), refdw) < 2e-5
    # (4) the dominant kernel: ViT fc1 Linear + GELU, 12608 x 1536 x 384
    xa, wa, ba = rnd(B * 197, 384, seed=6), rnd(1536, 384, seed=7, scale=0.05), rnd(1536, seed=8)
    ya = o.linear_fwd(xa.cuda(), wa.cuda(), ba.cuda(), act=ACT_GELU)
    assert rel_err(ya, F.gelu(xa.double() @ wa.double().T + ba.double())) < 2e-5
    # (5) ViT fc2 (K = 1536: the long-K chunked accumulation) with the residual add
    xb, wb, bb, rb = rnd(B * 197, 1536, seed=9), rnd(384, 1536, seed=10, scale=0.03), rnd(384, seed=11), rnd(B * 197, 384, seed=12)
    yb = o.linear_fwd(xb.cuda(), wb.cuda(), bb.cuda(), residual=rb.cuda())
    assert rel_err(yb, xb.double() @ wb.double().T + bb.double() + rb.double()) < 2e-5


@pytest.mark.parametrize("tile", [65, 67, 83, 129])
def test_eight_wave_tiles_all_ops_vs_fp64(tile):
    """IckGemm.tile +64: the 128-row tiles with eight waves per workgroup (4 x 2 wave grid) — every operand-fetch pattern of
    the LDS-DMA kernel (k-contiguous / x-contiguous A and B, conv gathers, stride-2 parity classes) against float64.
    129: four compute + four loader waves, three buffers (NT and CONV_FWD; the other ops fall back to the 128 x 128 tile)."""
    from imagecaptioner_amd import ops as o
    o._FORCE_TILE[0] = tile
    try:
        M, N, K = 300, 200, 136
        x, w, dy = rnd(M, K, seed=1), rnd(N, K, seed=2), rnd(M, N, seed=3)
        assert rel_err(o.linear_fwd(x.cuda(), w.cuda()), x.double() @ w.double().T) < 2e-5                 # NT
        assert rel_err(o.linear_bwd_data(dy.cuda(), w.cuda()), dy.double() @ w.double()) < 2e-5             # NN
        dw = torch.zeros(N, K, device="cuda")
        o.linear_bwd_weight(dy.cuda(), x.cuda(), dw, splitk=1)
        assert rel_err(dw, dy.double().T @ x.double()) < 2e-5                                                # TN
        xc, wc = rnd(3, 14, 14, 64, seed=4), rnd(96, 3, 3, 64, seed=5, scale=0.1)
        xr, wr = xc.double().permute(0, 3, 1, 2), wc.double().permute(0, 3, 1, 2)
        for stride in (1, 2):
            ref = F.conv2d(xr, wr, stride=stride, padding=1)
            y = o.conv_fwd(xc.cuda(), wc.cuda(), stride, 1)
            assert rel_err(y.permute(0, 3, 1, 2), ref) < 2e-5                                                # CONV_FWD
            dyc = rnd(*y.shape, seed=6)
            dx = o.conv_dgrad(dyc.cuda(), wc.cuda(), (14, 14), stride, 1)
            refdx = torch.nn.grad.conv2d_input(xr.shape, wr, dyc.double().permute(0, 3, 1, 2), stride=stride, padding=1)
            assert rel_err(dx.permute(0, 3, 1, 2), refdx) < 2e-5                                             # dgrad (as fwd / S2 classes)
        o._DGRAD_AS_FWD[0] = False
        dyc = rnd(3, 14, 14, 96, seed=7)
        dx = o.conv_dgrad(dyc.cuda(), wc.cuda(), (14, 14), 1, 1)
        refdx = torch.nn.grad.conv2d_input(xr.shape, wr, dyc.double().permute(0, 3, 1, 2), padding=1)
        assert rel_err(dx.permute(0, 3, 1, 2), refdx) < 2e-5                                                 # CONV_DGRAD gather kernel
    finally:
        o._FORCE_TILE[0] = 0
        o._DGRAD_AS_FWD[0] = True


def test_small_grid_convolutions_split_the_contraction():
    """layer4 geometry at a batch where the forward-type convolutions run split-K (ops._conv_splitk: M >= 1024 rows, <= 256
    tiles, K >= 2048): train-mode forward = fp32 atomics into the zeroed output + BatchNorm statistics from ick_colstats,
    stride-1 data gradient with the residual on split 0 and `accumulate` adding onto dx — all against fp64."""
    from imagecaptioner_amd import ops as o
    Nb, H, Cin, Cout, R = 32, 7, 512, 512, 3
    assert o._conv_splitk(Nb * H * H, Cout, R * R * Cin) > 1
    x = rnd(Nb, Cin, H, H, seed=1)
    w = rnd(Cout, Cin, R, R, seed=2, scale=1.0 / math.sqrt(Cin * R * R))
    xd = x.cuda().permute(0, 2, 3, 1).contiguous()
    wd = w.cuda().permute(0, 2, 3, 1).contiguous()
    stats = torch.zeros(2, 3, Cout, device="cuda", dtype=torch.float64)          # three accumulator copies: colstats fills copy 0
    y = o.conv_fwd(xd, wd, 1, 1, stats=(stats[0], stats[1]))
    ref = F.conv2d(x.double(), w.double(), None, 1, 1)
    assert rel_err(y.permute(0, 3, 1, 2), ref) < 2e-5
    assert rel_err(stats[0].sum(0), ref.sum((0, 2, 3))) < 1e-4 * max(1.0, ref.abs().sum((0, 2, 3)).max().item() / ref.sum((0, 2, 3)).abs().max().item())
    assert rel_err(stats[1].sum(0), (ref * ref).sum((0, 2, 3))) < 1e-5
    dy = rnd(*ref.shape, seed=3)
    dyd = dy.cuda().permute(0, 2, 3, 1).contiguous()
    res = rnd(Nb, H, H, Cin, seed=4).cuda()
    dx_ref = torch.nn.grad.conv2d_input(x.shape, w.double(), dy.double(), 1, 1) + res.cpu().double().permute(0, 3, 1, 2)
    dx = o.conv_dgrad(dyd, wd, (H, H), 1, 1, residual=res)
    assert rel_err(dx.permute(0, 3, 1, 2), dx_ref) < 2e-5
    dx2 = res.clone()
    o.conv_dgrad(dyd, wd, (H, H), 1, 1, out=dx2, accumulate=True)
    assert rel_err(dx2.permute(0, 3, 1, 2), dx_ref) < 2e-5


@pytest.mark.parametrize("M,N,K,act", [(12608, 1536, 384, 2), (12608, 384, 1536, 0), (12608, 1152, 384, 0), (1000, 512, 4096, 0)])
def test_f32x3_is_fp32_grade(M, N, K, act):
    """precision "f32x3" (igemm_glds_impl.h TERMS 4: a = hi + 2^-11 lo' in fp16, three MFMAs per product, two accumulators) on
    the ViT teacher's Linear shapes and one long-K product: its error against float64 (relative L2 over 512 rows) is at most
    1.25 x the exact-fp32 MFMA kernel's on the same operands — fp32-grade, not a reduced precision — with operands spanning
    five decades of magnitude (column scales 1e-3 .. 1e2) so that the scaled low part is exercised off fp16's sweet spot."""
    from imagecaptioner_amd import ops as o
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g) * torch.logspace(-3, 2, K).roll(7)
    w = torch.randn(N, K, generator=g) * K ** -0.5 * torch.logspace(-2, 1, N).unsqueeze(1)
    b = torch.randn(N, generator=g)
    pre = x[:512].double() @ w.double().T + b.double()
    ref = F.gelu(pre) if act == 2 else pre
    errs = {}
    for prec in ("f32", "f32x3"):
        with o.precision(prec):
            y = o.linear_fwd(x.cuda(), w.cuda(), b.cuda(), act=act)
        errs[prec] = ((y[:512].double().cpu() - ref).norm() / ref.norm()).item()
    print(M, N, K, errs)
    assert errs["f32x3"] <= 1.25 * errs["f32"] and errs["f32x3"] < 5e-7, errs


@pytest.mark.parametrize("scale", [1.0, 1e-7, 3e-12, 4e4])
def test_f32x3_with_absmax_scale_is_fp32_grade_for_any_magnitude(scale):
    """IckGemm.a_absmax (round 3): the three-fp16-product kernel takes max |A| from device memory and multiplies A by the power
    of two that puts it into fp16's range — so data gradients (1e-6 and far below) get fp32-grade products too.  A stride-1
    3x3 data gradient (forward convolution over dY with rotated weights) at layer3's geometry with dY scaled over sixteen
    decades, elements spread over four more.  Bound: 4e-7 relative L2 against float64 at EVERY scale — the level of a K-blocked
    CPU fp32 convolution on this K = 2304 contraction (the exact-fp32 kernel, which folds its accumulator chain every 64 k, reaches
    1.1e-7 here; the three-product kernel's chain of 144 accumulations 2.9e-7; an unfolded fp32 MFMA chain 1.8e-6) — and
    the SAME error at every scale; without the scale the same launch loses every low part at 1e-7 (asserted too)."""
    from imagecaptioner_amd import ops as o
    g = torch.Generator().manual_seed(11)
    Nb, H, C = 8, 14, 256
    dy = torch.randn(Nb, H, H, C, generator=g) * torch.logspace(-4, 0, C) * scale
    w = torch.randn(C, 3, 3, C, generator=g) * 0.03
    ref = torch.nn.grad.conv2d_input((Nb, C, H, H), w.double().permute(0, 3, 1, 2), dy.double().permute(0, 3, 1, 2), padding=1)
    errs = {}
    for prec in ("f32", "f32x3"):
        with o.precision(prec):
            dx = o.conv_dgrad(dy.cuda(), w.cuda(), (H, H), 1, 1)
        errs[prec] = ((dx.double().cpu().permute(0, 3, 1, 2) - ref).norm() / ref.norm()).item()
    o._X3_DGRAD[0] = False
    try:
        with o.precision("f32x3"):
            dx = o.conv_dgrad(dy.cuda(), w.cuda(), (H, H), 1, 1)      # exact kernel (gradients never take x3 unscaled)
        errs["f32x3, no scale (exact kernel)"] = ((dx.double().cpu().permute(0, 3, 1, 2) - ref).norm() / ref.norm()).item()
    finally:
        o._X3_DGRAD[0] = True
    print(scale, errs)
    assert errs["f32x3"] < 4e-7 and errs["f32x3"] <= 3.0 * errs["f32"], errs
    # the raw three-product kernel WITHOUT the scale on the same operands: fine at O(1), useless at 1e-7
    wt = o.conv_weight_dgrad_layout(w.cuda())
    dxr = torch.empty(Nb, H, H, C, device="cuda")
    with o.precision("f32x3"):
        o.gemm_raw(o.OP_CONV_FWD, dy.cuda().data_ptr(), wt.data_ptr(), dxr.data_ptr(), Nb * H * H, C, 9 * C, 9 * C, 9 * C, C,
                   conv=(Nb, H, H, C, H, H, C, 3, 3, 1, 1), x3=True)
    raw = ((dxr.double().cpu().permute(0, 3, 1, 2) - ref).norm() / ref.norm()).item()
    if scale == 1e-7:
        assert raw > 50 * errs["f32x3"], (raw, errs)


def test_absmax_kernel():
    from imagecaptioner_amd import ops as o
    x = torch.randn(1 << 20) * 1e-5
    x[123457] = -3.25e-2
    out = torch.zeros(1, device="cuda")
    o.absmax(x.cuda(), out)
    assert out.item() == float(torch.tensor(3.25e-2, dtype=torch.float32))          # exact: a maximum, not a sum
    o.absmax(torch.full((8,), 7.0).cuda(), out)            # accumulates: max with what is there
    assert out.item() == 7.0
    o.absmax(torch.zeros(8).cuda(), out)
    assert out.item() == 7.0


def test_f32x3_weight_gradient_as_tn_with_absmax_scale():
    """1x1 / stride-1 weight gradient under precision "f32x3": the TN product dW = dY^T X on the three-product kernel, dY scaled
    by its device-side absmax (gradient magnitudes 1e-7 here), split-K with fp32 atomics: against float64, beside the exact path."""
    from imagecaptioner_amd import ops as o
    g = torch.Generator().manual_seed(3)
    Nb, H, Cin, Cout = 16, 14, 1024, 256
    x = torch.randn(Nb, H, H, Cin, generator=g)
    dy = torch.randn(Nb, H, H, Cout, generator=g) * torch.logspace(-3, 0, Cout) * 1e-7
    ref = dy.double().view(-1, Cout).T @ x.double().view(-1, Cin)
    errs = {}
    for prec in ("f32", "f32x3"):
        dw = torch.zeros(Cout, 1, 1, Cin, device="cuda")
        with o.precision(prec):
            o.conv_wgrad(dy.cuda(), x.cuda(), dw, 1, 0)
        errs[prec] = ((dw.view(Cout, Cin).double().cpu() - ref).norm() / ref.norm()).item()
    print(errs)
    assert errs["f32x3"] < 4e-7 and errs["f32x3"] <= 3.0 * errs["f32"], errs


@pytest.mark.parametrize("stride", [1, 2])
def test_f32x3_3x3_weight_gradient_with_absmax_scale(stride):
    """3x3 weight gradient (CONV_WGRAD gather, stride 1 and 2) under precision "f32x3": the three-product kernel with dY scaled by
    its device-side absmax (magnitudes 1e-7), split-K, against float64 beside the exact path."""
    from imagecaptioner_amd import ops as o
    g = torch.Generator().manual_seed(5 + stride)
    Nb, H, Cin, Cout = 16, 14, 256, 256
    Ho = (H + 2 - 3) // stride + 1
    x = torch.randn(Nb, H, H, Cin, generator=g)
    dy = torch.randn(Nb, Ho, Ho, Cout, generator=g) * torch.logspace(-3, 0, Cout) * 1e-7
    ref = torch.nn.grad.conv2d_weight(x.double().permute(0, 3, 1, 2), (Cout, Cin, 3, 3), dy.double().permute(0, 3, 1, 2), stride=stride, padding=1)
    errs = {}
    for prec in ("f32", "f32x3"):
        dw = torch.zeros(Cout, 3, 3, Cin, device="cuda")
        with o.precision(prec):
            o.conv_wgrad(dy.cuda(), x.cuda(), dw, stride, 1)
        errs[prec] = ((dw.permute(0, 3, 1, 2).double().cpu() - ref).norm() / ref.norm()).item()
    print(stride, errs)
    assert errs["f32x3"] < 4e-7 and errs["f32x3"] <= 3.0 * errs["f32"], errs
