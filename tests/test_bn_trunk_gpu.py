"""GPU parity of the train-mode BatchNorm kernels in isolation and of a single ResNet Bottleneck, forward AND backward,
against float64 torch references.  Reference computation: nn.BatchNorm2d in train mode inside
torchvision's resnet50 as the reference uses it (/root/reference/src/student_model.py:16-30,57 — batch statistics in the
"frozen" and the trainable blocks alike).

Tolerances (max |err| / max |ref| unless stated):
  statistics of the conv epilogue     1e-6   (fp64 accumulators over fp32 partials of <= 64 values)
  bn_train_apply / bn_bwd_*           1e-5   (pure fp32 elementwise arithmetic on top of fp64-reduced sums)
  Bottleneck fwd + all gradients      1e-4   (three fp32 convolutions + three BatchNorms deep, B = 8)
"""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def rnd(*shape, seed=0, scale=1.0, shift=0.0):
    g = torch.Generator().manual_seed(seed + sum(shape))
    return torch.randn(*shape, generator=g) * scale + shift


# (Nb, H, W): M = 98, 3136, 12544, 200704 rows — one accumulator copy up to eight (ops.stat_copies)
GEOMS = [(2, 7, 7), (64, 7, 7), (64, 14, 14), (64, 56, 56)]


@pytest.mark.parametrize("Nb,H,W", GEOMS)
@pytest.mark.parametrize("C", [64, 256])
def test_conv_epilogue_statistics(Nb, H, W, C):
    """sum / sum of squares produced by the conv epilogue == fp64 sums of the raw output the same launch stored."""
    from imagecaptioner_amd import ops
    M = Nb * H * W
    x = rnd(Nb, H, W, 64, seed=1, shift=0.5).cuda()               # non-zero mean: E[x^2] - E[x]^2 has something to cancel
    w = rnd(C, 1, 1, 64, seed=2, scale=0.2).cuda()
    R = ops.stat_copies(M)
    stats = torch.zeros(2, R, C, dtype=torch.float64, device="cuda")
    raw = ops.conv_fwd(x, w, 1, 0, stats=(stats[0], stats[1]))
    r64 = raw.double().view(M, C)
    s, q = stats[0].sum(0), stats[1].sum(0)
    assert rel(s, r64.sum(0)) < 1e-6
    assert rel(q, (r64 * r64).sum(0)) < 1e-6
    var = q / M - (s / M) ** 2
    assert rel(var, r64.var(0, unbiased=False)) < 1e-5


@pytest.mark.parametrize("Nb,H,W", GEOMS)
@pytest.mark.parametrize("C,relu,res", [(64, True, False), (256, True, True), (1024, False, False)])
def test_bn_train_apply_vs_fp64(Nb, H, W, C, relu, res):
    from imagecaptioner_amd import ops
    M = Nb * H * W
    if M * C > 64 * 56 * 56 * 256:
        pytest.skip("larger than any activation of the trunk")
    raw = rnd(M, C, seed=3, scale=1.7, shift=0.8)
    raw = raw * (1.0 + torch.arange(C) % 7).float() * 0.3          # per-channel spread of scales
    g, b = rnd(C, seed=4, scale=0.3, shift=1.0), rnd(C, seed=5, scale=0.2)
    rm, rv = rnd(C, seed=6, scale=0.1), rnd(C, seed=7).abs() + 0.5
    resid = rnd(M, C, seed=8) if res else None
    R = ops.stat_copies(M)
    rd = raw.cuda()
    stats = torch.zeros(2, R, C, dtype=torch.float64, device="cuda")
    r64 = rd.double()
    # spread the exact sums over the copies the way the epilogue would (copy t gets every R-th 128-row tile)
    tiles = r64.split(128)
    for t_i, blk in enumerate(tiles):
        stats[0, t_i % R] += blk.sum(0)
        stats[1, t_i % R] += (blk * blk).sum(0)
    rmd, rvd = rm.clone().cuda(), rv.clone().cuda()
    y, mean, inv = ops.bn_train_apply(rd.view(Nb, H, W, C), (stats[0], stats[1]), g.cuda(), b.cuda(), rmd, rvd, 0.1, 1e-5,
                                      resid.cuda().view(Nb, H, W, C) if res else None, relu)
    rm64, rv64 = rm.double().clone(), rv.double().clone()
    x64 = raw.double().view(Nb, H, W, C).permute(0, 3, 1, 2)
    ref = F.batch_norm(x64, rm64, rv64, g.double(), b.double(), training=True, momentum=0.1, eps=1e-5).permute(0, 2, 3, 1)
    if res:
        ref = ref + resid.double().view(Nb, H, W, C)
    if relu:
        ref = torch.relu(ref)
    assert rel(y, ref) < 1e-5
    assert rel(mean, raw.double().mean(0)) < 1e-6
    assert rel(inv, 1.0 / torch.sqrt(raw.double().var(0, unbiased=False) + 1e-5)) < 1e-6
    assert rel(rmd, rm64) < 1e-6 and rel(rvd, rv64) < 1e-6


@pytest.mark.parametrize("Nb,H,W", GEOMS)
@pytest.mark.parametrize("C,mask,want_g", [(64, True, False), (256, True, True), (1024, False, False)])
def test_bn_backward_vs_fp64(Nb, H, W, C, mask, want_g):
    """ick_bn_bwd_reduce + ick_bn_bwd_apply == fp64 autograd of [relu](batch_norm(x) [+ res])."""
    from imagecaptioner_amd import ops
    M = Nb * H * W
    if M * C > 64 * 56 * 56 * 256:
        pytest.skip("larger than any activation of the trunk")
    x = rnd(M, C, seed=11, scale=1.3, shift=0.4)
    g, b = rnd(C, seed=12, scale=0.3, shift=1.0), rnd(C, seed=13, scale=0.2)
    # upstream gradient with a large per-channel constant part: sum(g) is then a sum of same-sign terms and
    # g - mean(g) cancels most of it (the regime behind a loss that averages over positions)
    dy = rnd(M, C, seed=14, scale=0.05) + rnd(1, C, seed=15)
    x64 = x.double().requires_grad_(True)
    g64, b64 = g.double().requires_grad_(True), b.double().requires_grad_(True)
    xn = x64.view(Nb, H, W, C).permute(0, 3, 1, 2)
    y64 = F.batch_norm(xn, None, None, g64, b64, training=True, eps=1e-5).permute(0, 2, 3, 1).reshape(M, C)
    out64 = torch.relu(y64) if mask else y64
    out64.backward(dy.double())
    mean = x.double().mean(0)
    inv = 1.0 / torch.sqrt(x.double().var(0, unbiased=False) + 1e-5)
    dg, db = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
    xd = x.cuda().view(Nb, H, W, C)
    dx, gout = ops.bn_bwd(dy.cuda().view(Nb, H, W, C), out64.detach().float().cuda().view(Nb, H, W, C) if mask else None, xd,
                          mean.float().cuda(), inv.float().cuda(), g.cuda(), dg, db, want_g, True)
    assert rel(dx.view(M, C), x64.grad) < 1e-5
    # dgamma = sum g*xhat with xhat built from the SAVED fp32 mean / invstd (torch saves them in the input dtype too):
    # against the same sum in fp64 the kernel must agree to 1e-6; against autograd's exact-statistics value the rounding
    # of the saved mean shows up as M * mean(g) * (mean32 - mean) * invstd (here up to ~3e-5 of scale)
    gm = dy.double() * (out64.detach() > 0) if mask else dy.double()
    dg_saved = (gm * (x.double() - mean.float().double()) * inv.float().double()).sum(0)
    assert rel(dg, dg_saved) < 1e-6
    assert rel(dg, g64.grad) < 1e-4
    assert rel(db, b64.grad) < 1e-6
    if want_g:
        gm = dy.double() * (out64.detach() > 0) if mask else dy.double()
        assert rel(gout.view(M, C), gm) < 1e-6


class _MaskedReLU(torch.autograd.Function):
    """relu with a caller-supplied 0/1 mask.  A pre-activation within fp32 rounding of zero (|y| ~ 1e-7; expected a few
    times per million elements) may land on either side in two correct fp32 evaluations, and BOTH 0 and 1 are valid
    subgradients there — but one flipped element moves dx by ~1e-1 of its scale in a 3x3 neighbourhood.  The reference
    therefore differentiates with the masks the HIP forward produced (they must agree with its own everywhere except at
    such ambiguous points: checked)."""

    @staticmethod
    def forward(ctx, y, mask):
        ctx.save_for_backward(mask)
        return y * mask

    @staticmethod
    def backward(ctx, g):
        (mask,) = ctx.saved_tensors
        return g * mask, None


def _ref_bottleneck(x, blk_sd, stride, down, masks):
    """torchvision Bottleneck v1.5 (stride on the 3x3), train-mode BatchNorm, NCHW float64; masks = (a1>0, a2>0, out>0)."""
    bn = lambda t, p: F.batch_norm(t, None, None, blk_sd[p + ".weight"], blk_sd[p + ".bias"], training=True, eps=1e-5)

    def relu(y, mask):
        own = y.detach() > 0
        differ = own != mask
        assert float(y.detach()[differ].abs().max() if differ.any() else 0.0) < 1e-5 * float(y.detach().abs().max()), \
            "ReLU masks differ away from zero"
        return _MaskedReLU.apply(y, mask.double())

    idt = x
    if down:
        idt = bn(F.conv2d(x, blk_sd["downsample.0.weight"], None, stride=stride), "downsample.1")
    y = relu(bn(F.conv2d(x, blk_sd["conv1.weight"]), "bn1"), masks[0])
    y = relu(bn(F.conv2d(y, blk_sd["conv2.weight"], None, stride=stride, padding=1), "bn2"), masks[1])
    y = bn(F.conv2d(y, blk_sd["conv3.weight"]), "bn3")
    return relu(y + idt, masks[2])


# (inplanes, planes, stride, downsample, H): layer3.0, layer3.1, layer4.0, layer4.2 geometries of the trunk
@pytest.mark.parametrize("inpl,planes,stride,down,H", [(512, 256, 2, True, 28), (1024, 256, 1, False, 14),
                                                        (1024, 512, 2, True, 14), (2048, 512, 1, False, 7)])
def test_bottleneck_train_fwd_bwd_vs_fp64(inpl, planes, stride, down, H):
    from imagecaptioner_amd import nn as hnn
    B = 8
    torch.manual_seed(inpl + planes + stride)
    blk = hnn.Bottleneck(inpl, planes, stride, downsample=down)
    with torch.no_grad():
        for name, p in blk.named_parameters():
            if p.dim() == 1:
                p.copy_(torch.rand_like(p) * 0.5 + 0.75 if name.endswith("weight") else torch.randn_like(p) * 0.1)
    x = torch.relu(rnd(B, inpl, H, H, seed=21)) * 0.7            # post-ReLU input, like every block of the trunk sees
    dout = rnd(B, planes * 4, H // stride, H // stride, seed=22, scale=0.1)
    sd64 = {k: v.detach().double().contiguous().requires_grad_(True) for k, v in blk.state_dict().items()
            if v.dtype.is_floating_point and "running" not in k}
    x64 = x.double().requires_grad_(True)

    blk = blk.cuda()
    xd = x.cuda().permute(0, 2, 3, 1).contiguous()
    out, rec = hnn.bottleneck_forward(blk, xd, True)
    masks = tuple((rec[k].permute(0, 3, 1, 2) > 0).cpu() for k in ("a1", "a2", "out"))
    out64 = _ref_bottleneck(x64, sd64, stride, down, masks)
    out64.backward(dout.double())
    assert rel(out.permute(0, 3, 1, 2), out64) < 1e-5
    dx = hnn.bottleneck_backward(blk, rec, dout.cuda().permute(0, 2, 3, 1).contiguous(), True)
    assert rel(dx.permute(0, 3, 1, 2), x64.grad) < 1e-4
    worst = {}
    for name, p in blk.named_parameters():
        worst[name] = rel(p.grad, sd64[name].grad)
    bad = {k: v for k, v in worst.items() if v >= 1e-4}
    assert not bad, bad
