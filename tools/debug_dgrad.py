"""Debug: stride-1 data gradients of the layer3 bottleneck at B=8 vs fp64, per tile choice."""
import sys, os, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imagecaptioner_amd import ops

def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).abs().max() / b.abs().max()).item()

torch.manual_seed(0)
B, H = 8, 14
for (cin, cout, k, pad, with_res) in ((1024, 256, 1, 0, True), (256, 256, 3, 1, False), (256, 1024, 1, 0, False)):
    w = torch.randn(cout, cin, k, k) * 0.05
    dy = torch.randn(B, cout, H, H)
    res = torch.randn(B, cin, H, H) if with_res else None
    ref = torch.nn.grad.conv2d_input((B, cin, H, H), w.double(), dy.double(), stride=1, padding=pad)
    if with_res:
        ref = ref + res.double()
    wp = w.permute(0, 2, 3, 1).contiguous().cuda()
    dyd = dy.permute(0, 2, 3, 1).contiguous().cuda()
    rd = res.permute(0, 2, 3, 1).contiguous().cuda() if with_res else None
    for tile in (0, 1, 2, 3, 4, 18, 19, 20, 256 + 1, 256 + 2):
        for kc in (0, -1):
            ops._FORCE_TILE[0] = tile
            ops._KCHUNK[0] = kc
            try:
                dx = ops.conv_dgrad(dyd, wp, (H, H), 1, pad, residual=rd)
                e = rel(dx.permute(0, 3, 1, 2), ref)
            except Exception as ex:
                e = str(ex)[:60]
            print(f"cin {cin} cout {cout} k{k} res={with_res} tile {tile:3d} kchunk {kc:2d}: {e}")
    ops._FORCE_TILE[0] = 0
    ops._KCHUNK[0] = 0
    for dgf in (True, False):
        ops._DGRAD_AS_FWD[0] = dgf
        dx = ops.conv_dgrad(dyd, wp, (H, H), 1, pad, residual=rd)
        print(f"   dgrad_as_fwd={dgf}: {rel(dx.permute(0, 3, 1, 2), ref)}")
    ops._DGRAD_AS_FWD[0] = True
