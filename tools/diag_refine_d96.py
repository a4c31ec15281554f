#!/usr/bin/env python3
"""Gradient error of AttentionRefinement (train mode, dropout 0) at E = 384 / 4 heads (head dim 96: the unfused attention
path) and E = 256 (head dim 64) against the oracle in float64, beside the oracle's own float32 error (GPU box)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imagecaptioner_amd.student_model import AttentionRefinement
from oracle import restatement as R

torch.manual_seed(0)
for E in (256, 384):
    m = AttentionRefinement(E, 4)
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    m.attention.dropout = 0.0
    m.train()
    for scale in (1.0, 4.0):
        x = torch.randn(16, 49, E) * scale
        g = torch.randn(16, 49, E)
        sd = {"attention_refinement." + k: v.detach().clone() for k, v in m.state_dict().items()}
        res = {}
        for dt in (torch.float32, torch.float64):
            s2 = {k: v.to(dt).clone().detach().requires_grad_(True) for k, v in sd.items()}
            xx = x.to(dt).clone().detach().requires_grad_(True)
            y = R.attention_refinement(s2, xx, train=True)
            (y * g.to(dt)).sum().backward()
            res[dt] = ({k: v.grad.double() for k, v in s2.items()}, xx.grad.double())
        mc = m.cuda()
        for p in mc.parameters():
            p.grad = None
        xc = x.cuda().requires_grad_(True)
        (mc(xc) * g.cuda()).sum().backward()
        print(f"E={E} input scale {scale}")
        for k, p in mc.named_parameters():
            ref = res[torch.float64][0]["attention_refinement." + k]
            e_hip = ((p.grad.double().cpu() - ref).norm() / ref.norm()).item()
            e_f32 = ((res[torch.float32][0]["attention_refinement." + k] - ref).norm() / ref.norm()).item()
            print(f"  {k:32s} hip {e_hip:.2e}  cpu-f32 {e_f32:.2e}  ratio {e_hip / max(e_f32, 1e-30):.2f}")
        ref = res[torch.float64][1]
        print(f"  {'dx':32s} hip {((xc.grad.double().cpu() - ref).norm() / ref.norm()).item():.2e}  cpu-f32 "
              f"{((res[torch.float32][1] - ref).norm() / ref.norm()).item():.2e}")
        m = mc.cpu()
