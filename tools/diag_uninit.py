#!/usr/bin/env python3
"""Does any kernel of the KD step read memory it (or a predecessor) did not write?  Every torch.empty / empty_like / new_empty
(and ops.empty) of one B = 16 forward + backward is filled with NaN first (GPU box); a NaN in the loss or in any gradient names
a consumer of uninitialised memory.  Usage: python tools/diag_uninit.py [f32|f32x3|fp16]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imagecaptioner_amd import ops  # noqa: E402
from imagecaptioner_amd.distillation_utils import DistillationLoss, TeacherWrapper  # noqa: E402
from imagecaptioner_amd.train_student_kd import build_kd_models  # noqa: E402
from imagecaptioner_amd.utils.seeded_init import synthetic_batch  # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else "f32x3"
student, teacher, projectors = build_kd_models(device="cuda")
student.train()
images, caps = synthetic_batch(16, 5000, 16, seed=1234)
images, caps = images.cuda(), caps.cuda()

_e, _el = torch.empty, torch.empty_like


def poison(t):
    if t.is_cuda and t.is_floating_point():
        t.fill_(float("nan"))
    elif t.is_cuda:
        t.fill_(-(1 << 20))
    return t


torch.empty = lambda *a, **k: poison(_e(*a, **k))
torch.empty_like = lambda *a, **k: poison(_el(*a, **k))
ops.empty = lambda *shape, dtype=torch.float32, device=None: poison(_e(*shape, dtype=dtype, device=device or "cuda"))

with ops.precision(prec):
    t_out = TeacherWrapper(teacher)(images, caps[:-1])
    logits, enc, hids, _ = student(images, caps[:-1])
    t_out["encoder_features"] = projectors["encoder"](t_out["encoder_features"])
    loss, parts = DistillationLoss(0.7, 0.2, 0.1, 4.0, 5000)({"logits": logits, "encoder_features": enc, "hidden_states": hids}, t_out, caps[1:])
    loss.backward()
torch.cuda.synchronize()
print("precision", prec, "loss", float(loss), parts)
bad = [k for k, p in list(student.named_parameters()) + list(projectors["encoder"].named_parameters()) if p.grad is not None and not torch.isfinite(p.grad).all()]
print("non-finite in: logits", not torch.isfinite(logits).all().item(), "enc", not torch.isfinite(enc).all().item(),
      "teacher logits", not torch.isfinite(t_out["logits"]).all().item(), "gradients:", bad[:20], len(bad))
