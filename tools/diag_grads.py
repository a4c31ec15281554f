#!/usr/bin/env python3
"""Diagnostic (GPU box): per-tensor gradient error of the HIP KD step vs an fp64 CPU oracle, next to the error of
the fp32 CPU oracle (the reference's own arithmetic class) vs the same fp64 run.

    python tools/diag_grads.py B [kchunk ...]      # one HIP run per kchunk value (ICK_KCHUNK; 0 = library default, -1 = off)

The CPU runs happen once; every HIP variant runs in a child process (the env switch is read once per process)."""
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from imagecaptioner_amd.utils.seeded_init import seeded_state_dict, synthetic_batch  # noqa: E402


def hip_child(B, out):
    from imagecaptioner_amd.train_student_kd import build_kd_models
    from imagecaptioner_amd.distillation_utils import DistillationLoss, TeacherWrapper
    student, teacher, projectors = build_kd_models(device="cuda")
    for m in list(student.modules()) + list(projectors["encoder"].modules()):
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    student.attention_refinement.attention.dropout = 0.0
    student.decoder.lstm.dropout = 0.0
    student.train()
    images, caps = synthetic_batch(B, 5000, 16, seed=1234)
    images, caps = images.cuda(), caps.cuda()
    t_out = TeacherWrapper(teacher)(images, caps[:-1])
    logits, enc, hids, _ = student(images, caps[:-1])
    t_out["encoder_features"] = projectors["encoder"](t_out["encoder_features"])
    loss, parts = DistillationLoss(0.7, 0.2, 0.1, 4.0, 5000)({"logits": logits, "encoder_features": enc, "hidden_states": hids},
                                                             t_out, caps[1:])
    loss.backward()
    res = {k: p.grad.detach().double().cpu() for k, p in student.named_parameters() if p.grad is not None}
    res["__enc__"] = enc.detach().double().cpu()
    res["__logits__"] = logits.detach().double().cpu()
    torch.save(res, out)


def run_cpu(B, dtype):
    from oracle import restatement as R
    torch.set_default_dtype(dtype)
    trainable = lambda k: not any(k.startswith(f"encoder.resnet.{i}.") for i in (0, 1, 4, 5)) and "running_" not in k
    conv = lambda sd: {k: (v.to(dtype).clone().requires_grad_(True) if (v.dtype.is_floating_point and trainable(k))
                           else v.to(dtype).clone()) for k, v in sd.items()}
    ssd = conv(seeded_state_dict(R.student_state_shapes(5000, 256, 512, 2, True), seed=0))
    tsd = {k: v.to(dtype) for k, v in seeded_state_dict(R.teacher_state_shapes(5000, 512, 4), seed=1).items()}
    psd = conv(seeded_state_dict(R.projector_state_shapes(512, 256), seed=2))
    images, caps = synthetic_batch(B, 5000, 16, seed=1234)
    _, _, logits = R.kd_forward_backward(ssd, tsd, psd, images.to(dtype), caps, hidden=512, layers=2, refine=True, t_heads=8,
                                         t_layers=4)
    torch.set_default_dtype(torch.float32)
    res = {k: v.grad.double() for k, v in ssd.items() if v.grad is not None}
    res["__logits__"] = logits.double()
    return res


if __name__ == "__main__":
    if sys.argv[1] == "--child":
        hip_child(int(sys.argv[2]), sys.argv[3])
        sys.exit(0)
    B = int(sys.argv[1])
    chunks = [int(a) for a in sys.argv[2:]] or [0]
    outdir = os.path.join(ROOT, "gpurun_out")
    os.makedirs(outdir, exist_ok=True)
    g64 = run_cpu(B, torch.float64)
    g32 = run_cpu(B, torch.float32)
    print(f"B={B}: CPU fp64 and fp32 oracle runs done", flush=True)
    l2 = lambda x, y: ((x - y).norm() / y.norm().clamp_min(1e-30)).item()
    hips = {}
    for kc in chunks:
        path = os.path.join(outdir, f"diag_hip_B{B}_k{kc}.pt")
        env = dict(os.environ, ICK_KCHUNK=str(kc))
        subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(B), path], check=True, env=env)
        hips[kc] = torch.load(path)
        os.remove(path)
    groups = {"layer3": "encoder.resnet.6.", "layer4": "encoder.resnet.7.", "projection": "encoder.projection.",
              "refinement": "attention_refinement.", "decoder": "decoder."}
    print(f"{'group (mean rel-L2 vs fp64)':32s} {'cpu32':>10s} " + " ".join(f"{'hip k=' + str(kc):>16s}" for kc in chunks))
    for gname, pre in groups.items():
        keys = [k for k in g64 if k.startswith(pre)]
        mean = lambda g: sum(l2(g[k], g64[k]) for k in keys) / len(keys)
        c = mean(g32)
        print(f"{gname:32s} {c:10.3e} " + " ".join(f"{mean(hips[kc]):9.2e}({mean(hips[kc]) / c:4.2f})".rjust(16) for kc in chunks))
    print(f"{'logits':32s} {l2(g32['__logits__'], g64['__logits__']):10.3e} " +
          " ".join(f"{l2(hips[kc]['__logits__'], g64['__logits__']):12.3e}" for kc in chunks))
    kc0 = chunks[0]
    print(f"\nper tensor (hip k={kc0}):")
    print(f"{'tensor':60s} {'hip L2':>9s} {'cpu32 L2':>9s} {'ratio':>6s}")
    for k in g64:
        if k.startswith("__"):
            continue
        a, c = l2(hips[kc0][k], g64[k]), l2(g32[k], g64[k])
        print(f"{k:60s} {a:9.2e} {c:9.2e} {a / max(c, 1e-30):6.2f}")
