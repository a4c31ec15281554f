"""Reference-anchored checks of the 16-bit (AMP) regime and of the metric's own batch (VERDICT r02 "next round" item 3).

(a) AMP yardstick.  tests/golden/kd_step_cfg3_B16_autocast_bf16.npz and kd_step_cfg5_B16.npz hold gradient slices of the KD
    step at B = 16 produced by the REFERENCE's own modules (oracle/make_goldens.py::_kd_step_reference) in float32, in
    float64 and under torch.autocast('cpu', bfloat16) opened exactly where train_student_kd.py:271-285 opens
    autocast('cuda') (student forward + projector + loss inside, teacher outside in fp32).  CPU autocast's op lists are not
    CUDA's (and the reference trains in fp16, which CPU autocast does not offer for every op): the autocast run is a
    YARDSTICK for how far a legitimate 16-bit evaluation of this ill-conditioned step lies from the fp32 one — not a pin.
        err(hip fp16, ref fp32) <= 1.25 x err(ref autocast-bf16, ref fp32)     per tensor group, median of per-tensor ratios
        err(hip bf16, ref fp32) <= 1.5  x err(ref autocast-bf16, ref fp32)
    fp16 is the reference's actual regime (train_student_kd.py:239,271) and carries 3 more significant bits than the
    yardstick: it must be clearly inside it (measured medians 0.28-0.62).  bf16 against a bf16 yardstick compares two
    evaluations of the SAME precision class whose errors are saturated — 0.3 to 1.4 relative L2 in every group, i.e. both
    gradients are mostly rounding noise at random init and B = 16 — so their ratio scatters around 1 (measured medians 0.81
    - 1.32 over cfg3 / cfg5; the op lists of CPU autocast and of this path differ, e.g. which normalisations stay fp32)
    and the bound is 1.5.
(b) cfg5's dimensions (384 / 768 / 3) get the fp64 yardstick of tests/test_kd_step_b16_gpu.py once:
        err(hip fp32, ref fp64) <= 1.25 x err(ref fp32, ref fp64).
(c) cfg3 at the metric's batch (B = 64): train-mode forward logits / loss terms <= 1e-3 against oracle.restatement run on the
    GPU box's CPU (the oracle itself is pinned to the reference by tests/test_oracle_vs_golden.py).
(d) the per-rank workloads of cfg4 (fp16, B = 64) and cfg5 (fp16, 384 / 768 / 3, B = 32) run under -m gpu as property tests."""
import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu

# tensor groups for the median-of-ratios criterion.  A median needs a few tensors: the fixtures hold one projection and two
# refinement tensors, so the two form ONE group ("head").  Measured at cfg5 in fp32: projection 0.95, refinement ffn 0.46,
# refinement in_proj 6.3 (8.5e-4 against the reference-fp32's 1.35e-4 on that tensor, both far below the 2e-3 ceiling) — the
# block's own arithmetic was then checked in isolation ON THE STEP'S ACTUAL INPUTS (tools/diag_refine_cfg5.py: captured
# features + incoming gradient, block re-evaluated in float64 on the CPU): HIP 1.4e-7, CPU float32 2.3e-7 on in_proj_weight.
# The 6.3 is therefore the sensitivity of that one gradient to the (valid, different) fp32 roundings upstream of the block,
# not an error of the attention kernels; single-tensor ratios are noisy for exactly that reason (tests/test_kd_step_b16_gpu.py).
GROUPS = {"layer3": ("encoder.resnet.6.",), "layer4": ("encoder.resnet.7.",), "head": ("encoder.projection.", "attention_refinement."),
          "decoder": ("decoder.",)}
CEIL = {"layer3": 2.5e-2, "layer4": 2.5e-2, "head": 2e-3, "decoder": 2e-3}     # fp32: absolute relative-L2 ceilings (group mean)
CFG = {"cfg3": dict(embed_size=256, hidden_size=512, num_layers=2), "cfg5": dict(embed_size=384, hidden_size=768, num_layers=3)}


def l2(a, b):
    a, b = torch.as_tensor(np.asarray(a)).double().flatten(), torch.as_tensor(np.asarray(b)).double().flatten()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def _no_dropout(s, p):
    for mod in list(s.modules()) + list(p["encoder"].modules()):
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    s.attention_refinement.attention.dropout = 0.0
    s.decoder.lstm.dropout = 0.0


def _hip_step(cfg, prec, B=16, seed=1234):
    """one eager KD step through KDTrainer (the path bench.py runs); returns ({name: unscaled grad}, loss_dict)"""
    from imagecaptioner_amd.train_student_kd import KDTrainer, build_kd_models
    from imagecaptioner_amd.utils.seeded_init import synthetic_batch
    s, t, p = build_kd_models(device="cuda", **CFG[cfg])
    _no_dropout(s, p)
    tr = KDTrainer(s, t, p, vocab_size=5000, batch_size=B, use_graph=False, precision=prec)
    images, caps = synthetic_batch(B, 5000, 16, seed=seed)
    tr.train_step(images.cuda(), caps.cuda())
    inv = 1.0
    if prec == "fp16":
        st = tr.scaler.tolist()
        assert st[2] == 0.0, st                  # no overflow at the default 2^16 scale
        inv = 1.0 / tr.loss_scale0
    grads = {k: v.grad.detach().float().cpu() * inv for k, v in s.named_parameters() if v.grad is not None}
    out = tr.loss_dict()
    del tr, s, t, p
    torch.cuda.empty_cache()
    return grads, out


def _goldens(cfg):
    from oracle.make_golden_keys import B16_KEYS, CFG5_KEYS
    if cfg == "cfg3":
        g, ac = load_golden("kd_step_cfg3_B16.npz"), load_golden("kd_step_cfg3_B16_autocast_bf16.npz")
        return B16_KEYS, g, ac
    g = load_golden("kd_step_cfg5_B16.npz")
    return CFG5_KEYS, g, g


def _report(rows, what):
    txt = "\n".join(f"{k:55s} hip {a:.2e}  yardstick {c:.2e}  ratio {a / max(c, 1e-30):.2f}" for k, a, c in rows)
    print(f"--- {what}\n{txt}")
    return txt


@pytest.mark.parametrize("cfg", ["cfg3", "cfg5"])
@pytest.mark.parametrize("prec", ["bf16", "fp16"])
def test_amp_step_b16_against_the_reference_autocast_yardstick(cfg, prec):
    keys, g, ac = _goldens(cfg)
    grads, losses = _hip_step(cfg, prec)
    rows = []
    for k, sl in keys.items():
        ref32 = g[f"g_f32:{k}"]
        rows.append((k, l2(grads[k][sl], ref32), l2(ac[f"g_ac:{k}"], ref32)))
    report = _report(rows, f"{cfg} {prec}: error vs the reference's fp32 gradients; yardstick = reference under CPU autocast bf16")
    # loss terms: the 16-bit step must sit as close to the reference's fp32 loss as the reference's own autocast run does (x2)
    kd32, kdac = float(g["kd_f32"]), float(ac["kd_ac"])
    assert abs(losses["token_kd_loss"] - kd32) <= max(2.0 * abs(kdac - kd32), 5e-3 * abs(kd32)), (losses, kd32, kdac)
    for name, pre in GROUPS.items():
        sel = [(a, c) for k, a, c in rows if k.startswith(pre)]
        med = float(np.median([a / max(c, 1e-30) for a, c in sel]))
        print(f"{cfg} {prec} {name}: median hip / yardstick error ratio {med:.2f} (mean hip error {np.mean([a for a, _ in sel]):.2e})")
        assert med <= (1.25 if prec == "fp16" else 1.5), f"{cfg} {prec} {name}: median ratio {med:.2f}\n{report}"


def test_cfg5_fp32_step_b16_against_the_fp64_yardstick():
    keys, g, _ = _goldens("cfg5")
    grads, losses = _hip_step("cfg5", "f32")
    assert abs(losses["token_kd_loss"] - float(g["kd_f32"])) < 1e-3 * abs(float(g["kd_f32"]))
    assert abs(losses["feature_kd_loss"] - float(g["feat_f32"])) < 1e-3 * max(1.0, abs(float(g["feat_f32"])))
    assert abs(losses["total_loss"] - float(g["loss_f32"])) < 1e-3 * abs(float(g["loss_f32"]))
    rows = [(k, l2(grads[k][sl], g[f"g_f64:{k}"]), l2(g[f"g_f32:{k}"], g[f"g_f64:{k}"])) for k, sl in keys.items()]
    report = _report(rows, "cfg5 f32: error vs the reference's fp64 gradients; yardstick = the reference's own fp32")
    ratios = np.array([a / max(c, 1e-30) for _, a, c in rows])
    for name, pre in GROUPS.items():
        sel = [(a, c) for k, a, c in rows if k.startswith(pre)]
        med = float(np.median([a / max(c, 1e-30) for a, c in sel]))
        hip_m = float(np.mean([a for a, _ in sel]))
        print(f"cfg5 f32 {name}: median ratio {med:.2f}, mean hip error {hip_m:.2e}")
        assert med <= 1.25, f"{name}: {med:.2f}\n{report}"
        assert hip_m <= CEIL[name], f"{name}: {hip_m:.3e} > {CEIL[name]}\n{report}"
    # (no single-tensor cap here: see the note at GROUPS on attention_refinement.attention.in_proj_weight)
    assert float(np.exp(np.log(ratios).mean())) <= 1.1, report


def test_cfg3_full_batch_forward_and_loss_vs_oracle():
    """B = 64, the batch BASELINE.json's metric is quoted on: eval logits and the train-mode forward's logits / loss terms
    against oracle.restatement on this box's CPU (about 5 s), north_star's 1e-3."""
    from imagecaptioner_amd.distillation_utils import DistillationLoss, TeacherWrapper
    from imagecaptioner_amd.train_student_kd import build_kd_models
    from imagecaptioner_amd.utils.seeded_init import seeded_state_dict, synthetic_batch
    from oracle import restatement as R
    B = 64
    student, teacher, projectors = build_kd_models(device="cuda")
    _no_dropout(student, projectors)
    images, caps = synthetic_batch(B, 5000, 16, seed=1234)
    cin, ctg = caps[:-1], caps[1:]
    ssd = seeded_state_dict(R.student_state_shapes(5000, 256, 512, 2, True), seed=0)
    tsd = seeded_state_dict(R.teacher_state_shapes(5000, 512, 4), seed=1)
    psd = seeded_state_dict(R.projector_state_shapes(512, 256), seed=2)
    torch.set_num_threads(max(1, min(16, torch.get_num_threads())))
    with torch.no_grad():
        want_eval, want_enc, _, _ = R.student_forward(ssd, images, cin, hidden=512, layers=2, refine=True, train=False)
        t_logits, t_feats = R.teacher_forward(tsd, images, cin, heads=8, layers=4)
        w_logits, w_enc, w_hids, _ = R.student_forward({k: v.clone() for k, v in ssd.items()}, images, cin, hidden=512, layers=2,
                                                       refine=True, train=True)
        t_proj = R.feature_projector(psd, t_feats, w_enc.shape[1])
        _, want = R.distillation_loss({"logits": w_logits, "encoder_features": w_enc, "hidden_states": w_hids},
                                      {"logits": t_logits, "encoder_features": t_proj, "hidden_states": None}, ctg)
    ic, cc = images.cuda(), cin.cuda()
    student.eval()
    with torch.no_grad():
        got_eval, got_enc, _, _ = student(ic, cc)
    assert (got_eval.cpu() - want_eval).abs().max().item() < 1e-3
    assert (got_enc.cpu() - want_enc).abs().max().item() < 1e-3
    assert torch.equal(got_eval.argmax(-1).cpu(), want_eval.argmax(-1)) or \
        float((got_eval.argmax(-1).cpu() != want_eval.argmax(-1)).float().mean()) < 1e-3   # top-2 ties at random init
    student.train()
    t_out = TeacherWrapper(teacher)(ic, cc)
    assert (t_out["logits"].cpu() - t_logits).abs().max().item() < 1e-3
    logits, enc, hids, _ = student(ic, cc)
    assert (logits.detach().cpu() - w_logits).abs().max().item() < 1e-3
    assert (enc.detach().cpu() - w_enc).abs().max().item() < 1e-3
    t_out["encoder_features"] = projectors["encoder"](t_out["encoder_features"])
    _, parts = DistillationLoss(0.7, 0.2, 0.1, 4.0, 5000)({"logits": logits, "encoder_features": enc, "hidden_states": hids},
                                                          t_out, ctg.cuda())
    for k in ("total_loss", "token_kd_loss", "feature_kd_loss", "ce_loss"):
        assert abs(parts[k] - want[k]) <= 1e-3 * max(1.0, abs(want[k])), (k, parts[k], want[k])


@pytest.mark.parametrize("cfg,B", [("cfg3", 64), ("cfg5", 32)])
def test_fp16_step_at_the_per_rank_batch_of_cfg4_and_cfg5(cfg, B):
    """cfg4's per-rank workload (cfg3 models, fp16 + GradScaler, B = 64) and cfg5's (384 / 768 / 3, fp16, B = 32) as captured
    steps: finite loss terms that agree with the fp32 step's first loss to 1 %, no overflow at the default scale, the loss
    falls over replayed steps on a fixed batch, captured == eager on the first step."""
    from imagecaptioner_amd.train_student_kd import KDTrainer, build_kd_models
    from imagecaptioner_amd.utils.seeded_init import synthetic_batch
    images, caps = synthetic_batch(B, 5000, 16, seed=9)
    first = {}
    for prec, use_graph in (("f32", False), ("fp16", False), ("fp16", True)):
        s, t, p = build_kd_models(device="cuda", **CFG[cfg])
        _no_dropout(s, p)
        tr = KDTrainer(s, t, p, vocab_size=5000, batch_size=B, use_graph=use_graph, precision=prec, learning_rate=1e-3)
        ls = []
        for i in range(1 if prec == "f32" else 4):
            tr.train_step(images.cuda() if i == 0 else None, caps.cuda() if i == 0 else None)
            ls.append(tr.loss_dict())
        if prec == "fp16":
            st = tr.scaler.tolist()
            assert st[0] == 65536.0 and st[2] == 0.0 and tr.applied_steps() == 4, st
            assert all(np.isfinite(v) for d in ls for v in d.values())
            assert ls[-1]["total_loss"] < ls[0]["total_loss"], [d["total_loss"] for d in ls]
        first[(prec, use_graph)] = ls[0]
        del tr, s, t, p
        torch.cuda.empty_cache()
    ref = first[("f32", False)]
    for key in (("fp16", False), ("fp16", True)):
        for k in ("total_loss", "token_kd_loss", "feature_kd_loss"):
            assert abs(first[key][k] - ref[k]) <= 1e-2 * max(abs(ref[k]), 1e-3), (key, k, first[key][k], ref[k])
    a, b = first[("fp16", False)], first[("fp16", True)]
    assert abs(a["total_loss"] - b["total_loss"]) <= 2e-3 * abs(a["total_loss"])
