#!/usr/bin/env python3
"""The B = 16 f32x3 step run repeatedly: where do the two outcomes of attention_refinement.in_proj_weight.grad differ?"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imagecaptioner_amd import ops  # noqa: E402
from imagecaptioner_amd.distillation_utils import DistillationLoss, TeacherWrapper  # noqa: E402
from imagecaptioner_amd.train_student_kd import build_kd_models  # noqa: E402
from imagecaptioner_amd.utils.seeded_init import synthetic_batch  # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else "f32x3"
runs = []
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 10):
    student, teacher, projectors = build_kd_models(device="cuda")
    for m in list(student.modules()) + list(projectors["encoder"].modules()):
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    student.attention_refinement.attention.dropout = 0.0
    student.decoder.lstm.dropout = 0.0
    student.train()
    images, caps = synthetic_batch(16, 5000, 16, seed=1234)
    images, caps = images.cuda(), caps.cuda()
    with ops.precision(prec):
        t_out = TeacherWrapper(teacher)(images, caps[:-1])
        logits, enc, hids, _ = student(images, caps[:-1])
        enc.retain_grad()
        t_out["encoder_features"] = projectors["encoder"](t_out["encoder_features"])
        loss, parts = DistillationLoss(0.7, 0.2, 0.1, 4.0, 5000)({"logits": logits, "encoder_features": enc, "hidden_states": hids}, t_out, caps[1:])
        loss.backward()
    sd = dict(student.named_parameters())
    keep = {k: sd[k].grad.detach().clone() for k in sd if (k.startswith("attention_refinement.") or k.startswith("encoder.projection.") or k == "decoder.attention.weight")}
    keep["enc"] = enc.detach().clone()
    keep["enc.grad"] = enc.grad.detach().clone()
    keep["logits"] = logits.detach().clone()
    keep["t_logits"] = t_out["logits"].detach().clone()
    runs.append(keep)
ref = runs[0]
for i, r in enumerate(runs[1:], 1):
    line = []
    for k in ref:
        d = (r[k].double() - ref[k].double())
        rel = float(d.norm() / ref[k].double().norm())
        nbig = int((d.abs() > 1e-4 * ref[k].abs().max()).sum())
        line.append(f"{'.'.join(k.split('.')[-3:])}:{rel:.1e}/{nbig}")
    print(f"run {i} vs run 0:", "  ".join(line))
