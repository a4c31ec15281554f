#!/usr/bin/env python3
"""cfg5 (embed 384, 4 heads -> head dim 96): is the refinement block's own gradient arithmetic accurate ON THE STEP'S ACTUAL
INPUTS?  Runs the B = 16 KD step on the HIP path, captures the block's input features and the gradient arriving at its output,
and re-evaluates the block alone on the CPU oracle in float64 and float32 on exactly those tensors (GPU box)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imagecaptioner_amd.distillation_utils import DistillationLoss, TeacherWrapper
from imagecaptioner_amd.train_student_kd import build_kd_models
from imagecaptioner_amd.utils.seeded_init import synthetic_batch
from oracle import restatement as R

E = int(sys.argv[1]) if len(sys.argv) > 1 else 384
dims = dict(embed_size=384, hidden_size=768, num_layers=3) if E == 384 else {}
s, t, p = build_kd_models(device="cuda", **dims)
for m in list(s.modules()) + list(p["encoder"].modules()):
    if isinstance(m, torch.nn.Dropout):
        m.p = 0.0
s.attention_refinement.attention.dropout = 0.0
s.decoder.lstm.dropout = 0.0
s.train()
cap = {}
ref_mod = s.attention_refinement
orig = ref_mod.forward


def fwd(x):
    cap["x"] = x.detach().clone()
    y = orig(x)
    y.register_hook(lambda g: cap.__setitem__("gy", g.detach().clone()))
    return y


ref_mod.forward = fwd
images, caps = synthetic_batch(16, 5000, 16, seed=1234)
images, caps = images.cuda(), caps.cuda()
t_out = TeacherWrapper(t)(images, caps[:-1])
logits, enc, hids, _ = s(images, caps[:-1])
t_out["encoder_features"] = p["encoder"](t_out["encoder_features"])
loss, _ = DistillationLoss(0.7, 0.2, 0.1, 4.0, 5000)({"logits": logits, "encoder_features": enc, "hidden_states": hids}, t_out, caps[1:])
loss.backward()
x, gy = cap["x"].cpu(), cap["gy"].cpu()
print(f"E={E}: |x| rms {x.pow(2).mean().sqrt():.3f} max {x.abs().max():.2f}; |gy| rms {gy.pow(2).mean().sqrt():.3e}")
sd = {"attention_refinement." + k: v.detach().cpu() for k, v in ref_mod.state_dict().items()}
res = {}
for dt in (torch.float32, torch.float64):
    s2 = {k: v.to(dt).clone().requires_grad_(True) for k, v in sd.items()}
    xx = x.to(dt).clone().requires_grad_(True)
    y = R.attention_refinement(s2, xx, train=True)
    (y * gy.to(dt)).sum().backward()
    res[dt] = {k: v.grad.double() for k, v in s2.items()}
for k, prm in ref_mod.named_parameters():
    ref = res[torch.float64]["attention_refinement." + k]
    e_hip = ((prm.grad.double().cpu() - ref).norm() / ref.norm()).item()
    e_f32 = ((res[torch.float32]["attention_refinement." + k] - ref).norm() / ref.norm()).item()
    extra = ""
    if k == "attention.in_proj_weight":
        for nm, sl in (("q", slice(0, E)), ("k", slice(E, 2 * E)), ("v", slice(2 * E, 3 * E))):
            r = ref[sl]; h = prm.grad.double().cpu()[sl]; c = res[torch.float32]["attention_refinement." + k][sl]
            extra += f"  [{nm}: |ref| {r.norm():.2e} hip {((h - r).norm() / r.norm()).item():.1e} cpu32 {((c - r).norm() / r.norm()).item():.1e}]"
    print(f"  {k:32s} hip {e_hip:.2e}  cpu-f32 {e_f32:.2e}  ratio {e_hip / max(e_f32, 1e-30):.2f}{extra}")
