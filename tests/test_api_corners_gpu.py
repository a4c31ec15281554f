"""The three API corners VERDICT r01 listed as raising instead of working (missing item 5), now built on the HIP path:
  * LSTMDecoder.forward(..., hidden=(h0, c0))      reference /root/reference/src/student_model.py:205,220
  * inputs other than 224x224 (AdaptiveAvgPool2d)   reference :34,60
  * CNNEncoder(fine_tune=False): everything trains  reference :12-30  (stem conv / bn1 / max-pool adjoints)
each against the CPU oracle (float64 where gradients through the train-mode trunk are involved)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def l2(a, b):
    a, b = a.detach().double().cpu().flatten(), b.detach().double().cpu().flatten()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


@pytest.mark.parametrize("E,H,NL", [(256, 512, 2), (128, 256, 1)])
def test_decoder_with_caller_supplied_initial_state(E, H, NL):
    from imagecaptioner_amd.student_model import LSTMDecoder
    from imagecaptioner_amd.utils.seeded_init import apply_seeded_init
    from oracle import restatement as R
    V, T, B = 500, 6, 3
    dec = apply_seeded_init(LSTMDecoder(V, E, H, NL, 0.0), 3).cuda().eval()
    g = torch.Generator().manual_seed(1)
    feats = torch.randn(B, 49, E, generator=g)
    caps = torch.randint(4, V, (T, B), generator=g)
    h0, c0 = torch.randn(NL, B, H, generator=g) * 0.5, torch.randn(NL, B, H, generator=g) * 0.5
    sd = {"decoder." + k: v.detach().double().cpu().requires_grad_(v.dtype.is_floating_point) for k, v in dec.state_dict().items()}
    f64 = feats.double().requires_grad_(True)
    h64, c64 = h0.double().requires_grad_(True), c0.double().requires_grad_(True)
    want, hids, attw = R.lstm_decoder(sd, f64, caps, NL, H, init=(h64, c64))
    dl = torch.randn(T, B, V, generator=g) * 1e-2
    want.backward(dl.double())
    fd = feats.cuda().requires_grad_(True)
    dec.train()                                                     # dropout p = 0: train mode keeps the backward records
    hd, cd = h0.cuda().requires_grad_(True), c0.cuda().requires_grad_(True)
    got, ghids, gattw = dec(fd, caps.cuda(), hidden=(hd, cd))
    assert rel(got, want) < 2e-5 and rel(ghids[-1], hids[-1]) < 2e-5 and rel(gattw[0], attw[0]) < 2e-5
    got.backward(dl.cuda())
    assert rel(fd.grad, f64.grad) < 2e-4
    # the gradient INTO the caller's state (the reference lets autograd flow into `hidden`, :205-222)
    assert hd.grad is not None and cd.grad is not None
    assert rel(hd.grad, h64.grad) < 2e-4 and rel(cd.grad, c64.grad) < 2e-4
    for k in ("lstm.weight_hh_l0", "attention.weight", "lstm.weight_ih_l0", "embedding.weight"):
        assert rel(dict(dec.named_parameters())[k].grad, sd["decoder." + k].grad) < 2e-4, k
    # zero state given explicitly == no state given
    z = torch.zeros(NL, B, H, device="cuda")
    a = dec(feats.cuda(), caps.cuda(), hidden=(z, z))[0]
    b = dec(feats.cuda(), caps.cuda())[0]
    assert rel(a, b) < 1e-6      # (not bit-equal: with an explicit state the zero recurrent half takes part in the K split)


def test_adaptive_avgpool_kernels_vs_torch():
    from imagecaptioner_amd import ops
    for H, W in ((8, 8), (10, 9), (13, 7), (5, 5), (3, 6), (1, 1)):      # smaller than 7x7: bins overlap / replicate
        x = torch.randn(3, H, W, 64)
        want = F.adaptive_avg_pool2d(x.double().permute(0, 3, 1, 2).requires_grad_(True), (7, 7))
        got = ops.adaptive_avgpool_fwd(x.cuda(), 7, 7)
        assert rel(got.permute(0, 3, 1, 2), want) < 1e-6
        xr = x.double().permute(0, 3, 1, 2).requires_grad_(True)
        dy = torch.randn(3, 7, 7, 64)
        F.adaptive_avg_pool2d(xr, (7, 7)).backward(dy.double().permute(0, 3, 1, 2))
        dx = ops.adaptive_avgpool_bwd(dy.cuda(), H, W)
        assert rel(dx.permute(0, 3, 1, 2), xr.grad) < 1e-6


def test_maxpool_backward_vs_torch():
    from imagecaptioner_amd import ops
    x = torch.randn(2, 14, 14, 32)
    x[0, 3:5, 3:5, :4] = 1.5                                         # ties inside a window: the first maximum takes the gradient
    xr = x.double().permute(0, 3, 1, 2).requires_grad_(True)
    y = F.max_pool2d(xr, 3, 2, 1)
    dy = torch.randn(2, 7, 7, 32)
    y.backward(dy.double().permute(0, 3, 1, 2))
    dx = ops.maxpool3x3s2_bwd(x.cuda(), dy.cuda())
    assert rel(dx.permute(0, 3, 1, 2), xr.grad) < 1e-6


def _oracle_encoder(sd, images, dt):
    from oracle import restatement as R
    sd = {k: (v.to(dt) if v.dtype.is_floating_point else v) for k, v in sd.items()}
    return R.cnn_encoder(sd, images.to(dt), train=False)


def test_encoder_accepts_other_input_sizes():
    """256x288 input: the trunk ends at 8x9 positions and AdaptiveAvgPool2d((7,7)) does real pooling (eval mode)."""
    from imagecaptioner_amd.student_model import CNNEncoder
    from imagecaptioner_amd.utils.seeded_init import apply_seeded_init
    enc = apply_seeded_init(CNNEncoder(256), 4).cuda().eval()
    images = torch.randn(2, 3, 256, 288, generator=torch.Generator().manual_seed(2))
    sd = {"encoder." + k: v.detach().cpu() for k, v in enc.state_dict().items()}
    want = _oracle_encoder(sd, images, torch.float32)
    with torch.no_grad():
        got = enc(images.cuda())
    assert got.shape == (2, 49, 256)
    assert rel(got, want) < 2e-4


def test_encoder_accepts_inputs_smaller_than_193():
    """160x160 input: the trunk ends at 5x5 positions and AdaptiveAvgPool2d((7,7)) REPLICATES bins (the reference accepts any
    size; round 2 raised ValueError here) — eval forward against the oracle, train-mode forward + backward runs."""
    from imagecaptioner_amd.student_model import CNNEncoder
    from imagecaptioner_amd.utils.seeded_init import apply_seeded_init
    enc = apply_seeded_init(CNNEncoder(256), 4).cuda().eval()
    images = torch.randn(2, 3, 160, 160, generator=torch.Generator().manual_seed(3))
    sd = {"encoder." + k: v.detach().cpu() for k, v in enc.state_dict().items()}
    want = _oracle_encoder(sd, images, torch.float32)
    with torch.no_grad():
        got = enc(images.cuda())
    assert got.shape == (2, 49, 256) and rel(got, want) < 2e-4
    enc.train()
    out = enc(images.cuda())
    out.square().sum().backward()
    g = enc.resnet[7][2].conv3.weight.grad
    assert g is not None and torch.isfinite(g).all() and float(g.abs().max()) > 0


def test_encoder_fine_tune_false_trains_the_stem():
    """CNNEncoder(fine_tune=False): gradients reach conv1 / bn1 / layer1 through the max-pool; vs the float64 oracle with
    the HIP path's own ReLU decisions being those of an fp32 evaluation (ill-conditioned through 50 train-mode layers at
    B = 2: relative L2 against fp64 within 3x what the fp32 oracle itself shows, and both small in absolute terms)."""
    from imagecaptioner_amd.student_model import CNNEncoder
    from imagecaptioner_amd.utils.seeded_init import apply_seeded_init
    from oracle import restatement as R
    enc = apply_seeded_init(CNNEncoder(128, fine_tune=False), 5).cuda().train()
    enc.projection[2].p = 0.0
    assert all(p.requires_grad for p in enc.parameters())
    images = torch.randn(2, 3, 224, 224, generator=torch.Generator().manual_seed(3))
    dy = torch.randn(2, 49, 128, generator=torch.Generator().manual_seed(4)) * 1e-2
    grads = {}
    for tag, dt in (("f64", torch.float64), ("f32", torch.float32)):
        sd = {"encoder." + k: (v.detach().cpu().to(dt).clone().requires_grad_(True) if (v.dtype.is_floating_point and "running" not in k)
                               else v.detach().cpu().clone().to(dt) if v.dtype.is_floating_point else v.detach().cpu().clone())
              for k, v in enc.state_dict().items()}
        out = R.cnn_encoder(sd, images.to(dt), train=True)
        out.backward(dy.to(dt))
        grads[tag] = {k: v.grad for k, v in sd.items() if v.requires_grad}
    got = enc(images.cuda())
    assert rel(got, out) < 1e-3                                       # out: the fp32 oracle run (last loop iteration)
    got.backward(dy.cuda())
    named = dict(enc.named_parameters())
    for k in ("resnet.0.weight", "resnet.1.weight", "resnet.4.0.conv1.weight", "resnet.5.3.conv3.weight", "resnet.7.2.conv3.weight"):
        g = named[k].grad
        assert g is not None and float(g.abs().max()) > 0, k
        e_hip, e_32 = l2(g, grads["f64"]["encoder." + k]), l2(grads["f32"]["encoder." + k], grads["f64"]["encoder." + k])
        assert e_hip < max(3.0 * e_32, 5e-3), (k, e_hip, e_32)
