"""GPU parity of the native 16-bit storage path (ick_gemm_h16, ick_bn_*16, ick_maxpool3x3s2_16, the casts) through the
C ABI.  The reference's autocast regime keeps activations and a weight copy in fp16 (train_student_kd.py:271); these
kernels consume and produce that storage directly.  References are float64 torch computations on the SAME rounded
operands, so the only error left is fp32 accumulation order (2e-5 of scale) — plus, where the output is stored in 16 bits,
one rounding of the result (2^-8 relative for bf16, 2^-11 for fp16)."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

ULP = {torch.bfloat16: 2.0 ** -8, torch.float16: 2.0 ** -11}
DTYPES = [torch.bfloat16, torch.float16]


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed + sum(shape))
    return torch.randn(*shape, generator=g) * scale


def rel_err(a, b):
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def close16(out, ref, dt, acc_tol=2e-5):
    """out (stored in dt) against the fp64 reference: elementwise |out - ref| <= ulp * |ref| + acc_tol * scale."""
    o, r = out.detach().double().cpu(), ref.detach().double().cpu()
    bound = ULP[dt] * r.abs() + acc_tol * r.abs().max()
    return bool(((o - r).abs() <= bound).all())


@pytest.mark.parametrize("dt", DTYPES)
def test_casts_match_torch_rounding(dt):
    from imagecaptioner_amd import ops
    x = (rnd(1000, 64, seed=5) * torch.logspace(-6, 4, 64)).cuda()
    h = ops.cast16(x, dt)
    assert h.dtype == dt and torch.equal(h, x.to(dt))
    assert torch.equal(ops.cast32(h), h.float())


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (197, 5000, 72), (960, 256, 512), (3136, 256, 2048), (1000, 384, 1536)])
def test_linear_nt_native16(dt, M, N, K):
    """K % 8 == 0 -> LDS-DMA kernel; K = 72 (K % 8 == 0 as well) and tile bit 8 force the register-staged one."""
    from imagecaptioner_amd import ops
    a, b = rnd(M, K, seed=1).cuda(), rnd(N, K, seed=2, scale=1 / math.sqrt(K)).cuda()
    bias, res = rnd(N, seed=3).cuda(), rnd(M, N, seed=4).cuda()
    a16, b16 = ops.cast16(a, dt), ops.cast16(b, dt)
    ref = a16.double() @ b16.double().t()
    for tile in (0, 2, 65, 256, 258):
        c = torch.empty(M, N, device="cuda")
        ops.gemm_raw(ops.OP_NT, a16.data_ptr(), b16.data_ptr(), c.data_ptr(), M, N, K, K, K, N, h16=dt, tile=tile)
        assert rel_err(c, ref) < 2e-5, f"tile {tile}"
    # bias + ReLU + 16-bit residual, 16-bit C
    res16 = ops.cast16(res, dt)
    c16 = torch.empty(M, N, device="cuda", dtype=dt)
    for tile in (0, 256):
        ops.gemm_raw(ops.OP_NT, a16.data_ptr(), b16.data_ptr(), c16.data_ptr(), M, N, K, K, K, N, h16=dt, tile=tile, bias=bias.data_ptr(),
                     act=ops.ACT_RELU, residual=res16.data_ptr(), ldr=N, io16=3)
        assert close16(c16, torch.relu(ref + bias.double()) + res16.double(), dt), f"tile {tile}"


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("M,N,K", [(512, 512, 256), (1000, 776, 328), (2048, 1024, 4096)])
def test_linear_nt_native16_256x256_tile(dt, M, N, K):
    """IckGemm.tile 69: the 256 x 256 x (64 halves) workgroup tile (128 KB of LDS, two-slab epilogue), ragged M / N / K
    tails included; bias + ReLU + residual through the same staged epilogue."""
    from imagecaptioner_amd import ops
    a16, b16 = ops.cast16(rnd(M, K, seed=1).cuda(), dt), ops.cast16(rnd(N, K, seed=2, scale=1 / math.sqrt(K)).cuda(), dt)
    bias, res = rnd(N, seed=3).cuda(), rnd(M, N, seed=4).cuda()
    ref = a16.double() @ b16.double().t()
    c = torch.empty(M, N, device="cuda")
    ops.gemm_raw(ops.OP_NT, a16.data_ptr(), b16.data_ptr(), c.data_ptr(), M, N, K, K, K, N, h16=dt, tile=69)
    assert rel_err(c, ref) < 2e-5
    ops.gemm_raw(ops.OP_NT, a16.data_ptr(), b16.data_ptr(), c.data_ptr(), M, N, K, K, K, N, h16=dt, tile=69, bias=bias.data_ptr(),
                 act=ops.ACT_RELU, residual=res.data_ptr(), ldr=N)
    assert rel_err(c, torch.relu(ref + bias.double()) + res.double()) < 2e-5


CONVS = [  # (Nb, H, W, Cin, Cout, R, stride, pad): trunk geometries of layer1-4 (Cin % 32 == 0; % 64 takes the LDS-DMA kernel)
    (2, 56, 56, 64, 64, 3, 1, 1), (2, 56, 56, 64, 256, 1, 1, 0), (2, 56, 56, 256, 128, 1, 1, 0), (2, 56, 56, 128, 128, 3, 2, 1),
    (2, 56, 56, 256, 512, 1, 2, 0), (2, 28, 28, 256, 256, 3, 2, 1), (2, 14, 14, 256, 256, 3, 1, 1), (3, 14, 14, 512, 512, 3, 2, 1),
    (3, 7, 7, 512, 512, 3, 1, 1), (3, 7, 7, 2048, 512, 1, 1, 0), (2, 14, 14, 96, 160, 3, 1, 1),
]


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("Nb,H,W,Cin,Cout,R,stride,pad", CONVS)
def test_conv_fwd_dgrad_wgrad_native16(dt, Nb, H, W, Cin, Cout, R, stride, pad):
    from imagecaptioner_amd import ops
    x = rnd(Nb, Cin, H, W, seed=1)
    w = rnd(Cout, Cin, R, R, seed=2, scale=1.0 / math.sqrt(Cin * R * R))
    x16 = ops.cast16(x.cuda().permute(0, 2, 3, 1).contiguous(), dt)
    w16 = ops.cast16(w.cuda().permute(0, 2, 3, 1).contiguous(), dt)
    xr, wr = x16.double().cpu().permute(0, 3, 1, 2), w16.double().cpu().permute(0, 3, 1, 2)
    ref = F.conv2d(xr, wr, None, stride, pad)
    # forward: fp32 raw output + BatchNorm statistics from the fp32 accumulators
    stats = torch.zeros(2, Cout, device="cuda", dtype=torch.float64)
    y = ops.conv_fwd(x16, w16, stride, pad, stats=(stats[0], stats[1]), out_dtype=torch.float32)
    assert y.dtype == torch.float32 and rel_err(y.permute(0, 3, 1, 2), ref) < 2e-5
    assert rel_err(stats[1], (ref * ref).sum((0, 2, 3))) < 1e-4
    # forward: 16-bit output (the storage the next layer reads), statistics unchanged by the output rounding
    stats2 = torch.zeros(2, Cout, device="cuda", dtype=torch.float64)
    y16 = ops.conv_fwd(x16, w16, stride, pad, stats=(stats2[0], stats2[1]))
    assert y16.dtype == dt and close16(y16.permute(0, 3, 1, 2), ref, dt)
    assert torch.equal(stats, stats2)
    # data gradient (stride 1: forward conv over dY with the 16-bit rotated weight; stride 2: parity classes), 16-bit residual
    dy = rnd(*ref.shape, seed=3)
    dy16 = ops.cast16(dy.cuda().permute(0, 2, 3, 1).contiguous(), dt)
    res16 = ops.cast16(rnd(Nb, H, W, Cin, seed=4).cuda(), dt)
    dyr = dy16.double().cpu().permute(0, 3, 1, 2)
    dx_ref = torch.nn.grad.conv2d_input(x.shape, wr, dyr, stride, pad) + res16.double().cpu().permute(0, 3, 1, 2)
    dx = ops.conv_dgrad(dy16, w16, (H, W), stride, pad, residual=res16)
    assert dx.dtype == dt and close16(dx.permute(0, 3, 1, 2), dx_ref, dt)
    if stride == 1:     # the direct CONV_DGRAD form (register-staged kernel, transposed LDS reads of the weight)
        prev = ops._DGRAD_AS_FWD[0]
        ops._DGRAD_AS_FWD[0] = False
        try:
            dx2 = ops.conv_dgrad(dy16, w16, (H, W), stride, pad, residual=res16)
        finally:
            ops._DGRAD_AS_FWD[0] = prev
        assert close16(dx2.permute(0, 3, 1, 2), dx_ref, dt)
    # accumulate into a 16-bit dx (layer entry: conv1 + downsample gradients)
    dx3 = res16.clone()
    ops.conv_dgrad(dy16, w16, (H, W), stride, pad, out=dx3, accumulate=True)
    assert close16(dx3.permute(0, 3, 1, 2), dx_ref, dt)
    # weight gradient: fp32 accumulate into .grad
    dw_ref = torch.nn.grad.conv2d_weight(xr, w.shape, dyr, stride, pad)
    for sk in (1, 0):
        dw = torch.zeros(Cout, R, R, Cin, device="cuda")
        ops.conv_wgrad(dy16, x16, dw, stride, pad, splitk=sk)
        assert rel_err(dw.permute(0, 3, 1, 2), dw_ref) < 5e-5, f"wgrad splitk={sk}"


@pytest.mark.parametrize("dt", DTYPES)
def test_weight_dgrad_layout16(dt):
    from imagecaptioner_amd import ops
    w = rnd(96, 3, 3, 160, seed=1).cuda()
    assert torch.equal(ops.conv_weight_dgrad_layout(w.to(dt)), ops.conv_weight_dgrad_layout(w).to(dt))


@pytest.mark.parametrize("dt", DTYPES)
def test_stem_conv_writes_16bit_activations(dt):
    """fp32 NHWC4 images in, 16-bit raw activations out (ick_gemm_bf16 with io16): statistics from the fp32 accumulators."""
    from imagecaptioner_amd import ops
    x = rnd(2, 3, 224, 224, seed=1)
    w = rnd(64, 3, 7, 7, seed=2, scale=0.1)
    x4 = ops.nchw3_to_nhwc4(x.cuda())
    w4 = ops.nchw3_to_nhwc4(w.cuda())
    with ops.precision("fp16" if dt == torch.float16 else "bf16"):
        s32 = torch.zeros(2, 64, device="cuda", dtype=torch.float64)
        y32 = ops.conv_fwd(x4, w4, 2, 3, stats=(s32[0], s32[1]))
        s16 = torch.zeros(2, 64, device="cuda", dtype=torch.float64)
        y16 = ops.conv_fwd(x4, w4, 2, 3, stats=(s16[0], s16[1]), out_dtype=dt)
    assert y16.dtype == dt and torch.equal(y16, y32.to(dt)) and torch.equal(s16, s32)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("M,C", [(98, 2048), (3136, 256), (12544, 64), (25088, 1024)])
def test_batchnorm_train_and_backward_on_16bit_storage(dt, M, C):
    """The templated BatchNorm kernels on 16-bit storage against THE SAME kernels on the fp32 copies of the same rounded
    inputs (those are pinned against fp64 in test_bn_trunk_gpu.py): identical arithmetic, outputs rounded once."""
    from imagecaptioner_amd import ops
    raw = ops.cast16((rnd(M, C, seed=1) * 2 + 0.3).cuda(), dt)
    res = ops.cast16(rnd(M, C, seed=2).cuda(), dt)
    gamma, beta = (rnd(C, seed=3).abs() + 0.5).cuda(), rnd(C, seed=4).cuda()
    r32 = raw.float()
    stats = torch.stack([r32.double().sum(0), (r32.double() ** 2).sum(0)]).contiguous()
    rm, rv = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
    y32, m32, i32 = ops.bn_train_apply(r32, (stats[0], stats[1]), gamma, beta, rm.clone(), rv.clone(), 0.1, 1e-5, res.float(), True)
    y16, m16, i16 = ops.bn_train_apply(raw, (stats[0], stats[1]), gamma, beta, rm, rv, 0.1, 1e-5, res, True)
    assert y16.dtype == dt and torch.equal(m16, m32) and torch.equal(i16, i32)
    assert torch.equal(y16, y32.to(dt))
    dy = ops.cast16(rnd(M, C, seed=5).cuda(), dt)
    dg32, db32 = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
    dg16, db16 = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
    # the mask comes from the STORED activation in both runs
    dx32, g32 = ops.bn_bwd(dy.float(), y16.float(), r32, m32, i32, gamma, dg32, db32, True)
    dx16, g16 = ops.bn_bwd(dy, y16, raw, m16, i16, gamma, dg16, db16, True)
    assert dx16.dtype == dt and g16.dtype == dt
    assert torch.equal(g16, g32.to(dt))
    assert torch.allclose(dg16, dg32, rtol=1e-6, atol=1e-6 * dg32.abs().max().item())
    assert torch.allclose(db16, db32, rtol=1e-6, atol=1e-6 * db32.abs().max().item())
    assert close16(dx16, dx32, dt, acc_tol=1e-6)


@pytest.mark.parametrize("dt", DTYPES)
def test_maxpool16_is_exact(dt):
    from imagecaptioner_amd import ops
    x = ops.cast16(rnd(2, 112, 112, 64, seed=1).cuda(), dt)
    assert torch.equal(ops.maxpool3x3s2(x), ops.maxpool3x3s2(x.float()).to(dt))


def test_h16_argument_checks():
    from imagecaptioner_amd import ops
    from imagecaptioner_amd._lib import IckError
    a = torch.zeros(64, 64, device="cuda", dtype=torch.bfloat16)
    c = torch.zeros(64, 64, device="cuda")
    with pytest.raises(IckError):        # split-K into a 16-bit C
        ops.gemm_raw(ops.OP_NT, a.data_ptr(), a.data_ptr(), a.data_ptr(), 64, 64, 64, 64, 64, 64, h16=torch.bfloat16, io16=1, splitk=2)
    with pytest.raises(IckError):        # the exact-fp32 family never writes 16-bit
        ops.gemm_raw(ops.OP_NT, c.data_ptr(), c.data_ptr(), c.data_ptr(), 64, 64, 64, 64, 64, 64, io16=1)


def _rel_l2(a, b):
    a, b = a.detach().double().cpu().flatten(), b.detach().double().cpu().flatten()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


# (inplanes, planes, stride, downsample, H): layer3.0, layer3.1, layer4.0, layer4.2 geometries of the trunk
@pytest.mark.parametrize("dt,tol,tol_g", [(torch.float16, 6e-3, 4e-2), (torch.bfloat16, 5e-2, 1.2e-1)])
@pytest.mark.parametrize("inpl,planes,stride,down,H", [(512, 256, 2, True, 28), (1024, 256, 1, False, 14),
                                                        (1024, 512, 2, True, 14), (2048, 512, 1, False, 7)])
def test_bottleneck_on_16bit_storage_vs_fp32_storage(dt, tol, tol_g, inpl, planes, stride, down, H):
    """One train-mode Bottleneck, forward and backward, with every activation / gradient stored in 16 bits (nn._TRUNK16's
    regime) against the exact-fp32 HIP path (pinned against fp64 in test_bn_trunk_gpu.py) on the SAME 16-bit-representable
    input, weights and output gradient.  What differs: three intermediate activations, three raw conv outputs and the
    gradients between the kernels are rounded to 11 (fp16) / 8 (bf16) significant bits.  Relative L2 bounds — output:
    6e-3 / 5e-2.  Gradients: 4e-2 / 1.2e-1 (measured fp16 2.3-2.7e-2 on all four geometries): a pre-activation within one
    rounding of zero takes the other side of the ReLU in the two runs, a fraction ~ulp of the elements, each an O(1) error
    in its gradient entry -> relative L2 ~ sqrt(ulp) (2.2e-2 for fp16), the mask effect test_bn_trunk_gpu.py removes by
    sharing masks; the per-kernel tests above are the tight ones."""
    from imagecaptioner_amd import nn as hnn
    from imagecaptioner_amd import ops
    B = 8
    torch.manual_seed(inpl + planes + stride)
    blk = hnn.Bottleneck(inpl, planes, stride, downsample=down)
    with torch.no_grad():
        for name, p in blk.named_parameters():
            if p.dim() == 1:
                p.copy_(torch.rand_like(p) * 0.5 + 0.75 if name.endswith("weight") else torch.randn_like(p) * 0.1)
            else:
                p.copy_(p.to(dt).float())                     # weights representable in the storage type
    blk = blk.cuda()
    x = (torch.relu(rnd(B, H, H, inpl, seed=21)) * 0.7).to(dt).cuda()
    dout = rnd(B, H // stride, H // stride, planes * 4, seed=22, scale=0.1).to(dt).cuda()

    def run(x_in, d_in):
        for p in blk.parameters():
            p.grad = None
        out, rec = hnn.bottleneck_forward(blk, x_in, True)
        dx = hnn.bottleneck_backward(blk, rec, d_in, True)
        return out, dx, {k: p.grad.clone() for k, p in blk.named_parameters()}

    hnn.clear_weight_shadows()
    out32, dx32, g32 = run(x.float(), dout.float())
    out16, dx16, g16 = run(x, dout)
    assert out16.dtype == dt and dx16.dtype == dt
    assert _rel_l2(out16, out32) < tol
    assert _rel_l2(dx16, dx32) < tol_g
    bad = {k: _rel_l2(g16[k], g32[k]) for k in g32 if _rel_l2(g16[k], g32[k]) >= tol_g}
    assert not bad, bad


def test_trunk16_kd_step_runs_on_16bit_kernels_and_matches_fp32_storage_losses():
    """The whole KD step under precision fp16 with 16-bit trunk storage (default) against fp32 storage between the kernels
    (nn._TRUNK16 off): same loss terms within 1 %, the native 16-bit entry points actually run, weights' shadow current."""
    from imagecaptioner_amd import nn as hnn
    from imagecaptioner_amd import ops
    from imagecaptioner_amd.train_student_kd import KDTrainer, build_kd_models
    from imagecaptioner_amd.utils.seeded_init import synthetic_batch
    images, caps = synthetic_batch(4, 5000, 16, seed=5)
    losses, calls = {}, {"h16": 0}
    orig = ops.gemm_raw

    def counting(*a, **kw):
        calls["h16"] += kw.get("h16") is not None
        return orig(*a, **kw)

    prev = hnn._TRUNK16[0]
    try:
        for on in (False, True):
            hnn._TRUNK16[0] = on
            hnn.clear_weight_shadows()
            s, t, p = build_kd_models(device="cuda")
            for mod in list(s.modules()) + list(p["encoder"].modules()):      # same arithmetic in both runs: no dropout draws
                if isinstance(mod, torch.nn.Dropout):
                    mod.p = 0.0
            s.attention_refinement.attention.dropout = 0.0
            s.decoder.lstm.dropout = 0.0
            tr = KDTrainer(s, t, p, vocab_size=5000, batch_size=4, use_graph=False, precision="fp16")
            assert (tr.flat16 is not None) == on
            ops.gemm_raw = counting
            calls["h16"] = 0
            tr.train_step(images.cuda(), caps.cuda())
            ops.gemm_raw = orig
            assert (calls["h16"] > 100) == on, calls
            losses[on] = tr.loss_dict()
            if on:
                conv = s.encoder.resnet[7][2].conv3
                with hnn.weight_shadows_current():          # inside the trainer's step: the flat 16-bit shadow, cast at its top
                    sh = hnn._w16(conv, torch.float16)
                lo, hi = tr.flat16.data_ptr(), tr.flat16.data_ptr() + tr.flat16.numel() * 2
                assert lo <= sh.data_ptr() < hi and sh.dtype == torch.float16
                # outside it (validation between steps) the shadow is one AdamW update behind: the weight is cast on use
                now = hnn._w16(conv, torch.float16)
                assert not (lo <= now.data_ptr() < hi)
                assert torch.equal(now, conv.packed().contiguous().to(torch.float16))
                assert not torch.equal(now, sh)
            del tr, s, t, p
    finally:
        ops.gemm_raw = orig
        hnn._TRUNK16[0] = prev
        hnn.clear_weight_shadows()
    for k in ("total_loss", "token_kd_loss", "feature_kd_loss"):
        assert abs(losses[True][k] - losses[False][k]) <= 1e-2 * max(abs(losses[False][k]), 1e-3), (k, losses)
