"""Debug: intermediates of one Bottleneck backward (HIP) vs fp64 autograd."""
import sys, os, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imagecaptioner_amd import nn as hnn, ops

def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).abs().max() / b.abs().max()).item()
def rnd(*shape, seed=0, scale=1.0, shift=0.0):
    g = torch.Generator().manual_seed(seed + sum(shape))
    return torch.randn(*shape, generator=g) * scale + shift
nchw = lambda t: t.permute(0, 3, 1, 2)

inpl, planes, stride, down, H = [int(a) for a in sys.argv[1:6]] if len(sys.argv) > 5 else (1024, 256, 1, 0, 14)
B = int(sys.argv[6]) if len(sys.argv) > 6 else 8
torch.manual_seed(inpl + planes + stride)
blk = hnn.Bottleneck(inpl, planes, stride, downsample=bool(down))
with torch.no_grad():
    for name, p in blk.named_parameters():
        if p.dim() == 1:
            p.copy_(torch.rand_like(p) * 0.5 + 0.75 if name.endswith("weight") else torch.randn_like(p) * 0.1)
x = torch.relu(rnd(B, inpl, H, H, seed=21)) * 0.7
dout = rnd(B, planes * 4, H // stride, H // stride, seed=22, scale=0.1)
sd = {k: v.detach().double().contiguous().requires_grad_(True) for k, v in blk.state_dict().items() if v.dtype.is_floating_point and "running" not in k}
x64 = x.double().requires_grad_(True)
bn = lambda t, p: F.batch_norm(t, None, None, sd[p + ".weight"], sd[p + ".bias"], training=True, eps=1e-5)
keep = {}
def k(name, t):
    t.retain_grad(); keep[name] = t; return t
r1 = k("r1", F.conv2d(x64, sd["conv1.weight"])); a1 = k("a1", torch.relu(bn(r1, "bn1")))
r2 = k("r2", F.conv2d(a1, sd["conv2.weight"], None, stride=stride, padding=1)); a2 = k("a2", torch.relu(bn(r2, "bn2")))
r3 = k("r3", F.conv2d(a2, sd["conv3.weight"]))
idt = x64
if down:
    rd_ = k("rd", F.conv2d(x64, sd["downsample.0.weight"], None, stride=stride)); idt = bn(rd_, "downsample.1")
pre = k("pre", bn(r3, "bn3") + idt)
out = torch.relu(pre)
out.backward(dout.double())

blk = blk.cuda()
xd = x.cuda().permute(0, 2, 3, 1).contiguous()
o, r = hnn.bottleneck_forward(blk, xd, True)
print("fwd out", rel(nchw(o), out), "r1", rel(nchw(r["r1"]), r1), "a1", rel(nchw(r["a1"]), a1), "r2", rel(nchw(r["r2"]), r2), "a2", rel(nchw(r["a2"]), a2), "r3", rel(nchw(r["r3"]), r3))
d = dout.cuda().permute(0, 2, 3, 1).contiguous()
dx3, g3 = ops.bn_bwd(d, r["out"], r["r3"], r["m3"], r["i3"], blk.bn3.weight, None, None, True, True)
print("g3 (=dpre)", rel(nchw(g3), pre.grad), " dx3 (=dr3)", rel(nchw(dx3), r3.grad))
da2 = ops.conv_dgrad(dx3, blk.conv3.packed(), r["a2"].shape[1:3], 1, 0)
print("da2", rel(nchw(da2), a2.grad))
dx2, _ = ops.bn_bwd(da2, r["a2"], r["r2"], r["m2"], r["i2"], blk.bn2.weight, None, None, False, True)
print("dx2 (=dr2)", rel(nchw(dx2), r2.grad))
da1 = ops.conv_dgrad(dx2, blk.conv2.packed(), r["a1"].shape[1:3], stride, 1)
print("da1", rel(nchw(da1), a1.grad))
dx1, _ = ops.bn_bwd(da1, r["a1"], r["r1"], r["m1"], r["i1"], blk.bn1.weight, None, None, False, True)
print("dx1 (=dr1)", rel(nchw(dx1), r1.grad))
# feed the EXACT upstream gradient to each stage: isolates the stage's own error from inherited error
ex = lambda t: t.grad.float().permute(0, 2, 3, 1).contiguous().cuda()
dx2e, _ = ops.bn_bwd(ex(a2), r["a2"], r["r2"], r["m2"], r["i2"], blk.bn2.weight, None, None, False, True)
print("dx2 from exact da2", rel(nchw(dx2e), r2.grad))
dx1e, _ = ops.bn_bwd(ex(a1), r["a1"], r["r1"], r["m1"], r["i1"], blk.bn1.weight, None, None, False, True)
print("dx1 from exact da1", rel(nchw(dx1e), r1.grad))
dx3e, g3e = ops.bn_bwd(d, r["out"], r["r3"], r["m3"], r["i3"], blk.bn3.weight, None, None, True, True)
print("mask agreement (out>0):", ((nchw(r["out"]).cpu() > 0) != (out.detach() > 0)).sum().item(), "of", out.numel())
for nm, t in (("a1", a1), ("a2", a2)):
    print(nm, "mask flips:", ((nchw(r[nm]).cpu() > 0) != (t.detach() > 0)).sum().item(), "of", t.numel(),
          " min var-ish invstd max:", float(r["i1" if nm == "a1" else "i2"].max()))
