#!/usr/bin/env python3
"""TEST INFRASTRUCTURE.  Generates tests/golden/*.npz by IMPORTING THE REFERENCE'S OWN
MODULES from /root/reference/src (read-only) in the build container — the reference
cannot travel to the GPU box, its outputs can.

    python oracle/make_goldens.py            # writes tests/golden/*.npz

What runs is the reference's code (CaptioningStudent, LSTMDecoder, AttentionRefinement,
CaptioningTeacher, DistillationLoss, FeatureProjector, TeacherWrapper,
create_feature_projectors); the only build-owned pieces are the torch-only stand-ins
for torchvision/timm (oracle/standins.py), the key-seeded weight generator and the
synthetic inputs (imagecaptioner_amd/utils/seeded_init.py).  Fixtures hold data only:
seeds, inputs that cannot be regenerated from a seed, and the reference's outputs.
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import standins  # noqa: E402
from imagecaptioner_amd.utils.seeded_init import apply_seeded_init, synthetic_batch  # noqa: E402

standins.install()
sys.path.insert(0, "/root/reference/src")
import student_model as ref_student  # noqa: E402
import teacher_model as ref_teacher  # noqa: E402
import distillation_utils as ref_kd  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
V = 5000
T1 = 16


class _Vocab:
    """Minimal .stoi/.itos object (the only vocabulary API caption_image uses,
    /root/reference/src/student_model.py:344,372,376)."""

    def __init__(self, n):
        self.itos = {i: f"w{i}" for i in range(n)}
        self.itos.update({0: "<PAD>", 1: "<START>", 2: "<END>", 3: "<UNK>"})
        self.stoi = {w: i for i, w in self.itos.items()}


def zero_dropout(m: torch.nn.Module):
    """p=0 everywhere so that train-mode runs are deterministic."""
    for sub in m.modules():
        if isinstance(sub, torch.nn.Dropout):
            sub.p = 0.0
        if isinstance(sub, (torch.nn.MultiheadAttention, torch.nn.LSTM)):
            sub.dropout = 0.0
    return m


def npz(name, **arrs):
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **{k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v))
                                 for k, v in arrs.items()})
    print(f"wrote {path}  ({os.path.getsize(path) / 1024:.0f} KiB)")


def top2_margin(logits):
    t2 = logits.topk(2, dim=-1).values
    return (t2[..., 0] - t2[..., 1])


# ------------------------------------------------------------------ (1) cfg1 student eval + greedy
def golden_cfg1():
    torch.manual_seed(0)
    m = ref_student.CaptioningStudent(V, 128, 256, 1, use_attention_refinement=False)
    apply_seeded_init(m, seed=0)
    m.eval()
    images, caps = synthetic_batch(2, V, T1, seed=1234)
    with torch.no_grad():
        logits, enc, hids, attw = m(images, caps[:-1])
    vocab = _Vocab(V)
    words = [m.caption_image(images[b], vocab, max_length=20) for b in range(2)]
    ids = np.full((20, 2), -1, dtype=np.int64)
    for b, ws in enumerate(words):
        for t, w in enumerate(ws):
            ids[t, b] = vocab.stoi[w]
    npz("cfg1_student_eval.npz", seed=0, batch_seed=1234, logits=logits, argmax=logits.argmax(-1),
        margin=top2_margin(logits), enc=enc, hid0=hids[0], hid7=hids[7], hid14=hids[14],
        attw0=attw[0], attw14=attw[14], greedy_ids=ids)


# ------------------------------------------------------------------ (2) decoder-only
def golden_decoders():
    for (E, H, L) in ((128, 256, 1), (256, 512, 2), (384, 768, 3)):
        torch.manual_seed(0)
        Vd = 1000
        dec = ref_student.LSTMDecoder(Vd, E, H, L, dropout=0.2)
        apply_seeded_init(dec, seed=3, prefix="decoder.")
        dec.eval()
        g = torch.Generator().manual_seed(77)
        feats = torch.randn(3, 49, E, generator=g)
        caps = torch.randint(0, Vd, (7, 3), generator=g)
        feats_g = feats.clone().requires_grad_(True)
        logits, hids, attw = dec(feats_g, caps)
        # scalar whose gradient exercises every path incl. hidden states
        gl = torch.randn(logits.shape, generator=g)
        gh = torch.randn(len(hids), *hids[0].shape, generator=g)
        obj = (logits * gl).sum() + (torch.stack(hids) * gh).sum()
        dec.zero_grad()
        obj.backward()
        npz(f"decoder_E{E}_H{H}_L{L}.npz", feats=feats, caps=caps, logits=logits, hids=torch.stack(hids),
            attw=torch.stack(attw), gl=gl, gh=gh, dfeats=feats_g.grad,
            d_whh0=dec.lstm.weight_hh_l0.grad[:8], d_wih0=dec.lstm.weight_ih_l0.grad[:8],
            d_att_w=dec.attention.weight.grad[:8], d_att_b=dec.attention.bias.grad,
            d_comb_w=dec.attention_combine.weight.grad[:8],
            d_emb_rows=dec.embedding.weight.grad[caps.flatten()[:6]], emb_row_ids=caps.flatten()[:6],
            d_out0_w=dec.output_projection[0].weight.grad[:8], d_out3_b=dec.output_projection[3].bias.grad,
            d_bih_last=getattr(dec.lstm, f"bias_ih_l{L-1}").grad)


# ------------------------------------------------------------------ (3) refinement
def golden_refinement():
    for E in (256, 384):
        torch.manual_seed(0)
        r = ref_student.AttentionRefinement(E)
        apply_seeded_init(r, seed=4, prefix="attention_refinement.")
        r.eval()
        g = torch.Generator().manual_seed(5)
        x = torch.randn(3, 49, E, generator=g)
        xg = x.clone().requires_grad_(True)
        y = r(xg)
        gy = torch.randn(y.shape, generator=g)
        (y * gy).sum().backward()
        npz(f"refinement_E{E}.npz", x=x, y=y, gy=gy, dx=xg.grad, d_inproj=r.attention.in_proj_weight.grad[::37],
            d_inproj_b=r.attention.in_proj_bias.grad, d_outproj=r.attention.out_proj.weight.grad[::29],
            d_ffn0=r.ffn[0].weight.grad[::41], d_norm1_w=r.norm1.weight.grad, d_norm2_b=r.norm2.bias.grad)


# ------------------------------------------------------------------ (4) losses
def golden_losses():
    g = torch.Generator().manual_seed(11)
    T, B, Vl, E, H = 6, 5, 257, 64, 48
    for tau in (3.0, 4.0):
        s = (torch.randn(T, B, Vl, generator=g) * 2).requires_grad_(True)
        t = torch.randn(T, B, Vl, generator=g) * 3
        tg = torch.randint(1, Vl, (T, B), generator=g)
        tg[4:, 1] = 0
        tg[2:, 3] = 0
        sf = torch.randn(B, 49, E, generator=g).requires_grad_(True)
        tf = torch.randn(B, 49, E, generator=g).requires_grad_(True)
        sh = [torch.randn(B, H, generator=g).requires_grad_(True) for _ in range(T)]
        th = [torch.randn(B, H, generator=g) for _ in range(T - 1)]      # shorter: exercises truncation (:109-113)
        L = ref_kd.DistillationLoss(alpha=0.5, beta=0.2, gamma=0.1, temperature=tau, vocab_size=Vl)
        kl = L.token_level_distillation(s, t)
        fe = L.encoder_feature_distillation(sf, tf)
        hi = L.decoder_hidden_state_distillation(sh, th)
        total, parts = L({"logits": s, "encoder_features": sf, "hidden_states": sh},
                         {"logits": t, "encoder_features": tf, "hidden_states": th}, tg)
        total.backward()
        # default weights (CE weight 2.78e-17) and hidden_states None (fact 2)
        L2 = ref_kd.DistillationLoss(vocab_size=Vl, temperature=tau)
        total2, parts2 = L2({"logits": s.detach(), "encoder_features": sf.detach(), "hidden_states": sh},
                            {"logits": t, "encoder_features": tf.detach(), "hidden_states": None}, tg)
        npz(f"losses_tau{int(tau)}.npz", s=s, t=t, targets=tg, sf=sf, tf=tf, sh=torch.stack(sh), th=torch.stack(th),
            kl=kl, feat=fe, hid=hi, total=total, ce=parts["ce_loss"], ds=s.grad, dsf=sf.grad, dtf=tf.grad,
            dsh=torch.stack([x.grad if x.grad is not None else torch.zeros_like(x) for x in sh]), total_default=total2, hid_default=parts2["hidden_kd_loss"],
            ce_weight_default=(1 - L2.alpha - L2.beta - L2.gamma))


# ------------------------------------------------------------------ (5) projector
def golden_projector():
    for E in (128, 256, 384):
        torch.manual_seed(0)
        pr = ref_kd.FeatureProjector(512, E, 197, 49)
        apply_seeded_init(pr, seed=2)
        pr.eval()
        g = torch.Generator().manual_seed(6)
        x = torch.randn(2, 197, 512, generator=g)
        xg = x.clone().requires_grad_(True)
        y = pr(xg)
        gy = torch.randn(y.shape, generator=g)
        (y * gy).sum().backward()
        npz(f"projector_E{E}.npz", x=x[:, ::8, ::8], x_seed=6, y=y, gy=gy, dx=xg.grad[:, ::8, ::8],
            dw=pr.feature_projection[0].weight.grad[::16], db=pr.feature_projection[0].bias.grad,
            dlnw=pr.feature_projection[3].weight.grad)
    # the reference's own test shape (test_dimension_fix.py:16-43): (2,197,384) -> (2,64,256)
    pr = ref_kd.FeatureProjector(384, 256, 197, 64)
    apply_seeded_init(pr, seed=2)
    pr.eval()
    g = torch.Generator().manual_seed(6)
    x = torch.randn(2, 197, 384, generator=g)
    with torch.no_grad():
        y = pr(x)
    npz("projector_ref_test_shape.npz", y=y, shape=np.array(y.shape))


# ------------------------------------------------------------------ (6) teacher eval
def build_teacher():
    torch.manual_seed(0)
    t = ref_teacher.CaptioningTeacher(V, embed_size=512, num_heads=8, num_decoder_layers=4, dropout=0.15)
    apply_seeded_init(t, seed=1)
    t.eval()
    return t


def golden_teacher():
    t = build_teacher()
    images, caps = synthetic_batch(2, V, T1, seed=1234)
    out = ref_kd.TeacherWrapper(t)(images, caps[:-1])
    with torch.no_grad():
        vit = t.encoder.forward_features(images)
    npz("teacher_eval.npz", logits=out["logits"], argmax=out["logits"].argmax(-1), margin=top2_margin(out["logits"]),
        enc_feats=out["encoder_features"][:, ::4], vit_tokens=vit[:, ::4],
        hidden_is_none=np.array(out["hidden_states"] is None))


# ------------------------------------------------------------------ (7)+(8) train-mode step
def golden_kd_step():
    torch.manual_seed(0)
    t = build_teacher()
    s = ref_student.CaptioningStudent(V, 256, 512, 2, dropout=0.3, use_attention_refinement=True)
    apply_seeded_init(s, seed=0)
    zero_dropout(s)
    s.train()
    projectors = ref_kd.create_feature_projectors(t, s)
    apply_seeded_init(projectors["encoder"], seed=2)
    zero_dropout(projectors["encoder"])
    images, caps = synthetic_batch(2, V, T1, seed=1234)
    cin, ctg = caps[:-1], caps[1:]
    tw = ref_kd.TeacherWrapper(t)
    L = ref_kd.DistillationLoss(alpha=0.7, beta=0.2, gamma=0.1, temperature=4.0, vocab_size=V)
    # optimizer exactly as /root/reference/src/train_student_kd.py:219-236
    lr = 2e-4
    other = list(s.attention_refinement.parameters())
    for pr in projectors.values():
        other.extend(list(pr.parameters()))
    opt = torch.optim.AdamW([{"params": list(s.encoder.parameters()), "lr": lr * 0.1},
                             {"params": list(s.decoder.parameters()), "lr": lr},
                             {"params": other, "lr": lr}], weight_decay=0.01)
    t_out = tw(images, cin)
    logits, enc, hids, attw = s(images, cin)
    s_out = {"logits": logits, "encoder_features": enc, "hidden_states": hids}
    t_out["encoder_features"] = projectors["encoder"](t_out["encoder_features"])
    loss, parts = L(s_out, t_out, ctg)
    loss.backward()
    sd = dict(s.named_parameters())
    frozen_none = all(sd[k].grad is None for k in sd if k.startswith("encoder.resnet.5.") or k.startswith("encoder.resnet.4."))
    grads = {k: p.grad.clone() for k, p in sd.items() if p.grad is not None}
    gn_student = torch.sqrt(sum((g ** 2).sum() for g in grads.values()))
    pj = dict(projectors["encoder"].named_parameters())
    gn_proj = torch.sqrt(sum((p.grad ** 2).sum() for p in pj.values()))
    rm_after = s.encoder.resnet[1].running_mean.clone()
    rv_after = s.encoder.resnet[7][2].bn3.running_var.clone()
    torch.nn.utils.clip_grad_norm_(s.parameters(), max_norm=1.0)
    for pr in projectors.values():
        torch.nn.utils.clip_grad_norm_(pr.parameters(), max_norm=1.0)
    before = {k: sd[k].detach().clone() for k in ("decoder.lstm.weight_hh_l0", "encoder.resnet.7.2.conv3.weight",
                                                  "encoder.projection.0.weight", "attention_refinement.ffn.0.weight")}
    pj_before = pj["feature_projection.0.weight"].detach().clone()
    opt.step()
    npz("kd_step_cfg3_B2.npz", loss=loss, ce=parts["ce_loss"], kd=parts["token_kd_loss"], feat=parts["feature_kd_loss"],
        hid=parts["hidden_kd_loss"], logits_slice=logits[:, :, ::50], enc=enc, t_proj=t_out["encoder_features"],
        hid14=hids[14], frozen_none=np.array(frozen_none), gn_student=gn_student, gn_proj=gn_proj,
        g_whh0=grads["decoder.lstm.weight_hh_l0"][::64, ::16],
        g_l4conv3=grads["encoder.resnet.7.2.conv3.weight"][::64, ::16, 0, 0],
        g_l3conv2=grads["encoder.resnet.6.0.conv2.weight"][::16, ::16],
        g_l4ds=grads["encoder.resnet.7.0.downsample.0.weight"][::64, ::32, 0, 0],
        g_l3bn1_w=grads["encoder.resnet.6.0.bn1.weight"], g_l4bn3_b=grads["encoder.resnet.7.2.bn3.bias"],
        g_encproj=grads["encoder.projection.0.weight"][::16, ::64],
        g_ref_inproj=grads["attention_refinement.attention.in_proj_weight"][::32, ::16],
        g_emb_row1=grads["decoder.embedding.weight"][1], g_out3_b=grads["decoder.output_projection.3.bias"][::10],
        g_att_w=grads["decoder.attention.weight"][::16, ::16],
        g_proj_w=pj["feature_projection.0.weight"].grad[::16, ::16] if False else np.zeros(1),
        bn1_running_mean=rm_after, l4_bn3_running_var=rv_after,
        d_whh0=(sd["decoder.lstm.weight_hh_l0"].detach() - before["decoder.lstm.weight_hh_l0"])[::64, ::16],
        d_l4conv3=(sd["encoder.resnet.7.2.conv3.weight"].detach() - before["encoder.resnet.7.2.conv3.weight"])[::64, ::16, 0, 0],
        d_encproj=(sd["encoder.projection.0.weight"].detach() - before["encoder.projection.0.weight"])[::16, ::64],
        d_ffn0=(sd["attention_refinement.ffn.0.weight"].detach() - before["attention_refinement.ffn.0.weight"])[::16, ::16],
        d_projw=(pj["feature_projection.0.weight"].detach() - pj_before)[::16, ::16])


# ------------------------------------------------------------------ (7b) the same step at a larger batch, fp32 AND fp64
from oracle.make_golden_keys import B16_KEYS  # noqa: E402


def golden_kd_step_b16(B=16):
    """Forward + backward of the cfg3 KD step through the reference's own modules at B = 16, once in float32 (the
    reference's arithmetic) and once in float64 (the yardstick): the GPU test holds the HIP gradients to
    err(hip, fp64) <= 1.2 x err(reference fp32, fp64) per tensor group."""
    out = {}
    for tag, dt in (("f32", torch.float32), ("f64", torch.float64)):
        torch.manual_seed(0)
        torch.set_default_dtype(dt)      # the reference creates its LSTM state with a bare torch.zeros (student_model.py:167-171)
        t = build_teacher().to(dt)
        s = ref_student.CaptioningStudent(V, 256, 512, 2, dropout=0.3, use_attention_refinement=True)
        apply_seeded_init(s, seed=0)
        zero_dropout(s)
        s.to(dt).train()
        projectors = ref_kd.create_feature_projectors(t, s)
        apply_seeded_init(projectors["encoder"], seed=2)
        zero_dropout(projectors["encoder"])
        projectors["encoder"].to(dt)
        images, caps = synthetic_batch(B, V, T1, seed=1234)
        cin, ctg = caps[:-1], caps[1:]
        # TeacherWrapper casts to .float() (distillation_utils.py:273): for the fp64 yardstick call the teacher's own
        # forward pieces instead (same computation, no cast)
        with torch.no_grad():
            t_logits = t(images.to(dt), cin)
            t_feats = t.encoder_projection(t.encoder.forward_features(images.to(dt)))
        L = ref_kd.DistillationLoss(alpha=0.7, beta=0.2, gamma=0.1, temperature=4.0, vocab_size=V)
        logits, enc, hids, attw = s(images.to(dt), cin)
        s_out = {"logits": logits, "encoder_features": enc, "hidden_states": hids}
        t_out = {"logits": t_logits, "encoder_features": projectors["encoder"](t_feats), "hidden_states": None}
        loss, parts = L(s_out, t_out, ctg)
        loss.backward()
        sd = dict(s.named_parameters())
        out[f"loss_{tag}"] = loss.detach()
        out[f"kd_{tag}"] = np.float64(parts["token_kd_loss"])
        out[f"feat_{tag}"] = np.float64(parts["feature_kd_loss"])
        out[f"logits_{tag}"] = logits.detach()[::2, :, ::25]
        out[f"enc_{tag}"] = enc.detach()[:, ::4, ::4]
        for k, sl in B16_KEYS.items():
            out[f"g_{tag}:{k}"] = sd[k].grad[sl]
        print(f"  {tag}: loss {float(loss):.6f}")
    torch.set_default_dtype(torch.float32)
    out["keys"] = np.array(sorted(B16_KEYS))
    npz(f"kd_step_cfg3_B{B}.npz", **out)


def _kd_step_reference(dt, B, dims, autocast=None):
    """One forward + backward of the KD step through the reference's own modules (train_student_kd.py:262-288, dropout p = 0).
    dims = (embed, hidden, layers); autocast = None or a torch dtype: the student forward, the projector and the loss then
    run inside torch.autocast('cpu', dtype=...) exactly where the reference opens autocast('cuda') (:271-285), the teacher
    outside of it in fp32 (:265-268)."""
    torch.manual_seed(0)
    torch.set_default_dtype(dt)
    E, H, Lyr = dims
    t = build_teacher().to(dt)
    s = ref_student.CaptioningStudent(V, E, H, Lyr, dropout=0.3, use_attention_refinement=True)
    apply_seeded_init(s, seed=0)
    zero_dropout(s)
    s.to(dt).train()
    projectors = ref_kd.create_feature_projectors(t, s)
    apply_seeded_init(projectors["encoder"], seed=2)
    zero_dropout(projectors["encoder"])
    projectors["encoder"].to(dt)
    images, caps = synthetic_batch(B, V, T1, seed=1234)
    cin, ctg = caps[:-1], caps[1:]
    with torch.no_grad():
        t_logits = t(images.to(dt), cin)
        t_feats = t.encoder_projection(t.encoder.forward_features(images.to(dt)))
    L = ref_kd.DistillationLoss(alpha=0.7, beta=0.2, gamma=0.1, temperature=4.0, vocab_size=V)
    import contextlib
    ctx = torch.autocast("cpu", dtype=autocast) if autocast is not None else contextlib.nullcontext()
    with ctx:
        logits, enc, hids, attw = s(images.to(dt), cin)
        s_out = {"logits": logits, "encoder_features": enc, "hidden_states": hids}
        t_out = {"logits": t_logits, "encoder_features": projectors["encoder"](t_feats), "hidden_states": None}
        loss, parts = L(s_out, t_out, ctg)
    loss.backward()
    torch.set_default_dtype(torch.float32)
    return s, loss, parts, logits, enc


def golden_kd_step_b16_autocast(B=16):
    """The AMP yardstick: the cfg3 step at B = 16 through the reference's own modules under
    torch.autocast('cpu', bfloat16), placed where train_student_kd.py:271-285 places autocast('cuda').  CPU autocast's op
    lists differ from CUDA's (and CPU has no fp16 autocast for every op): this is a YARDSTICK for how far a 16-bit
    evaluation of this step lies from the fp32 one, not a pin.  Stored: the same gradient slices as kd_step_cfg3_B16.npz."""
    out = {}
    s, loss, parts, logits, enc = _kd_step_reference(torch.float32, B, (256, 512, 2), autocast=torch.bfloat16)
    sd = dict(s.named_parameters())
    out["loss_ac"] = loss.detach().float()
    out["kd_ac"] = np.float64(parts["token_kd_loss"])
    out["feat_ac"] = np.float64(parts["feature_kd_loss"])
    out["logits_ac"] = logits.detach().float()[::2, :, ::25]
    out["enc_ac"] = enc.detach().float()[:, ::4, ::4]
    for k, sl in B16_KEYS.items():
        out[f"g_ac:{k}"] = sd[k].grad.float()[sl]
    print(f"  autocast bf16: loss {float(loss):.6f}")
    npz(f"kd_step_cfg3_B{B}_autocast_bf16.npz", **out)


from oracle.make_golden_keys import CFG5_KEYS, COMPACT_B16_KEYS  # noqa: E402


def golden_kd_step_cfg5_b16(B=16):
    """cfg5's student (embed 384 / hidden 768 / 3-layer LSTM, refinement on) + the teacher at B = 16: reference float32,
    float64 (yardstick) and CPU-autocast bf16 gradients ('repeat the fp64 yardstick once at cfg5')."""
    out = {}
    for tag, dt, ac in (("f32", torch.float32, None), ("f64", torch.float64, None), ("ac", torch.float32, torch.bfloat16)):
        s, loss, parts, logits, enc = _kd_step_reference(dt, B, (384, 768, 3), autocast=ac)
        sd = dict(s.named_parameters())
        out[f"loss_{tag}"] = loss.detach().double()
        out[f"kd_{tag}"] = np.float64(parts["token_kd_loss"])
        out[f"feat_{tag}"] = np.float64(parts["feature_kd_loss"])
        out[f"logits_{tag}"] = logits.detach()[::2, :, ::25].to(dt)
        out[f"enc_{tag}"] = enc.detach()[:, ::4, ::4].to(dt)
        for k, sl in CFG5_KEYS.items():
            out[f"g_{tag}:{k}"] = sd[k].grad.to(dt)[sl]
        print(f"  cfg5 {tag}: loss {float(loss):.6f}")
    npz(f"kd_step_cfg5_B{B}.npz", **out)


# ------------------------------------------------------------------ (N4) compact student
def golden_compact():
    """CompactCaptioningStudent (reference src/student_model_compact.py) with the MobileNetV2 stand-in: eval forward + greedy
    caption ids, and a train-mode (batch-statistics BatchNorm, dropout 0) forward + backward with gradient slices."""
    import student_model_compact as ref_compact
    torch.manual_seed(0)
    m = ref_compact.CompactCaptioningStudent(V, 256, 256, 1, use_attention_refinement=False)
    apply_seeded_init(m, seed=7)
    zero_dropout(m)
    images, caps = synthetic_batch(2, V, T1, seed=4321)
    m.eval()
    with torch.no_grad():
        logits, enc, hids, attw = m(images, caps[:-1])
    vocab = _Vocab(V)
    ids = []
    for b in range(2):
        words = m.caption_image(images[b], vocab, max_length=12)
        ids.append([vocab.stoi[w] for w in words] + [-1] * (12 - len(words)))
    out = dict(eval_logits=logits[:, :, ::10], eval_argmax=logits.argmax(-1), eval_margin=top2_margin(logits), eval_enc=enc,
               eval_hid7=hids[7], eval_attw0=attw[0], greedy_ids=np.array(ids))
    m.train()
    logits, enc, hids, attw = m(images, caps[:-1])
    g = torch.Generator().manual_seed(77)
    dl = torch.randn(logits.shape, generator=g) * 1e-2
    de = torch.randn(enc.shape, generator=g) * 1e-2
    (logits * dl).sum().add((enc * de).sum()).backward()
    sd = dict(m.named_parameters())
    frozen = all(sd[k].grad is None for k in sd if any(k.startswith(f"encoder.backbone.{i}.") for i in range(10)))
    out.update(train_logits=logits[:, :, ::10], train_enc=enc, frozen_none=np.array(frozen),
               bn0_running_mean=m.encoder.backbone[0][1].running_mean, bn17_running_var=m.encoder.backbone[17].conv[3].running_var)
    for k, sl in (("encoder.backbone.18.0.weight", np.s_[::8, ::4, 0, 0]), ("encoder.backbone.17.conv.1.0.weight", np.s_[::4, 0]),
                  ("encoder.backbone.14.conv.0.0.weight", np.s_[::8, ::4, 0, 0]), ("encoder.backbone.10.conv.2.weight", np.s_[::2, ::8, 0, 0]),
                  ("encoder.backbone.10.conv.1.1.weight", np.s_[:]), ("encoder.backbone.12.conv.3.weight", np.s_[:]),
                  ("encoder.projection.0.weight", np.s_[::4, ::16]), ("decoder.attention.weight", np.s_[::4, ::4]),
                  ("decoder.lstm.weight_hh_l0", np.s_[::16, ::4]), ("decoder.embedding.weight", np.s_[::40, ::4]),
                  ("decoder.output_projection.weight", np.s_[::40, ::4])):
        out["g:" + k] = sd[k].grad[sl]
    total, trainable = ref_compact.count_parameters(m)
    out.update(total_params=total, trainable_params=trainable, keys=np.array(sorted(m.state_dict().keys())))
    npz("compact_student.npz", **out)


def golden_compact_b16(B=16):
    """CompactCaptioningStudent train-mode forward + backward at B = 16 through the reference class in float32 AND float64
    (the B = 2 comparison uses 2-6 % tolerances; here the HIP gradients are held to the error-ratio criterion of
    tests/test_kd_step_b16_gpu.py against the fp64 yardstick)."""
    import student_model_compact as ref_compact
    out = {}
    for tag, dt in (("f32", torch.float32), ("f64", torch.float64)):
        torch.manual_seed(0)
        m = ref_compact.CompactCaptioningStudent(V, 256, 256, 1, use_attention_refinement=False)
        apply_seeded_init(m, seed=7)                    # float32 draws in both passes; the float64 pass widens the same values
        zero_dropout(m)
        m.to(dt).train()
        images, caps = synthetic_batch(B, V, T1, seed=4321)
        torch.set_default_dtype(dt)                     # the reference's forward allocates its zero state in the default dtype
        try:
            logits, enc, hids, attw = m(images.to(dt), caps[:-1])
        finally:
            torch.set_default_dtype(torch.float32)
        g = torch.Generator().manual_seed(77)
        dl = torch.randn(logits.shape, generator=g) * 1e-2
        de = torch.randn(enc.shape, generator=g) * 1e-2
        (logits * dl.to(dt)).sum().add((enc * de.to(dt)).sum()).backward()
        sd = dict(m.named_parameters())
        out[f"logits_{tag}"] = logits.detach()[::2, :, ::25]
        out[f"enc_{tag}"] = enc.detach()[:, ::4, ::4]
        for k, sl in COMPACT_B16_KEYS.items():
            out[f"g_{tag}:{k}"] = sd[k].grad[sl]
    npz(f"compact_student_B{B}.npz", **out)


def golden_compact_decoder_layers():
    """The reference's CompactLSTMDecoder with num_layers = 2 and a caller-supplied (h0, c0) (student_model_compact.py:72-111,
    :140-190): forward outputs and gradients (parameters, features, the state) on fixed features — VERDICT r02 item 10."""
    import student_model_compact as ref_compact
    torch.manual_seed(0)
    Vd, E, H, NL, T, B = 500, 128, 256, 2, 6, 3
    dec = ref_compact.CompactLSTMDecoder(Vd, E, H, NL, 0.0)
    apply_seeded_init(dec, seed=9)
    dec.train()
    g = torch.Generator().manual_seed(4)
    feats = torch.randn(B, 49, E, generator=g).requires_grad_(True)
    caps = torch.randint(4, Vd, (T, B), generator=g)
    h0 = (torch.randn(NL, B, H, generator=g) * 0.5).requires_grad_(True)
    c0 = (torch.randn(NL, B, H, generator=g) * 0.5).requires_grad_(True)
    dl = torch.randn(T, B, Vd, generator=g) * 1e-2
    out, hids, attw = dec(feats, caps, hidden=(h0, c0))
    (out * dl).sum().backward()
    sd = dict(dec.named_parameters())
    res = dict(feats=feats.detach(), caps=caps, h0=h0.detach(), c0=c0.detach(), dl=dl, logits=out.detach(), hid_last=hids[-1].detach(),
               attw0=attw[0].detach(), dfeats=feats.grad, dh0=h0.grad, dc0=c0.grad, dims=np.array([Vd, E, H, NL, T, B]))
    for k in ("lstm.weight_ih_l0", "lstm.weight_hh_l0", "lstm.weight_ih_l1", "lstm.weight_hh_l1", "lstm.bias_ih_l1", "attention.weight",
              "embedding.weight", "output_projection.weight"):
        res["g:" + k] = sd[k].grad[::4, ::4] if sd[k].dim() == 2 else sd[k].grad
    out0, _, _ = dec(feats.detach(), caps)                      # no state given = zero state
    res["logits_zero_state"] = out0.detach()
    npz("compact_decoder_2layer.npz", **res)


# ------------------------------------------------------------------ (9) teacher beam search
def golden_beam():
    t = build_teacher()
    vocab = _Vocab(V)
    images, _ = synthetic_batch(3, V, T1, seed=4321)
    outs = {}
    for b in range(3):
        for (beam, lp, nret, tag) in ((5, 0.6, 3, "b5"), (3, 0.0, 1, "b3"), (1, 0.6, 1, "b1")):
            caps = t.caption_image(images[b], vocab, max_length=12, beam_size=beam, length_penalty=lp,
                                   num_return_sequences=nret)
            outs[f"img{b}_{tag}"] = np.array(caps)
    # with key-seeded random weights <END> never wins, so the finishing / length-penalty / shrinking-beam logic
    # (teacher_model.py:198-229) is exercised by raising the <END> bias of fc_out (same shift on both sides in the test)
    for bias in (12.0, 10.5):
        with torch.no_grad():
            t.fc_out.bias[2] += bias
        for b in range(3):
            caps = t.caption_image(images[b], vocab, max_length=12, beam_size=5, length_penalty=0.6, num_return_sequences=5)
            outs[f"img{b}_b5_end{bias}"] = np.array(caps)
        with torch.no_grad():
            t.fc_out.bias[2] -= bias
    npz("teacher_beam.npz", batch_seed=4321, **outs)


def golden_optloss():
    """N4: the reference's OptimizedDistillationLoss (/root/reference/src/train_student_kd_optimized.py:34-128), imported
    with empty stand-ins for the two modules its file imports but the class does not use (torchvision.transforms,
    data_loader).  Teacher hiddens are None as in the reference's train path (its hidden term draws torch.randn
    attention weights, which no fixture can pin)."""
    import types
    import torchvision
    tr = types.ModuleType("torchvision.transforms")
    sys.modules["torchvision.transforms"] = tr
    torchvision.transforms = tr
    dl = types.ModuleType("data_loader")
    dl.get_loader = None
    sys.modules["data_loader"] = dl
    import train_student_kd_optimized as ref_opt
    g = torch.Generator().manual_seed(23)
    T, B, Vl, E = 6, 5, 257, 64
    out = {}
    s_logits0 = torch.randn(T, B, Vl, generator=g) * 2
    t_logits = torch.randn(T, B, Vl, generator=g) * 3
    s_feat0 = torch.randn(B, 49, E, generator=g)
    t_feat0 = torch.randn(B, 49, E, generator=g)
    targets = torch.randint(0, Vl, (T, B), generator=g)
    targets[4:, 1] = 0                                      # PAD tails (not ignored by this loss: plain CrossEntropyLoss)
    out.update(s_logits=s_logits0, t_logits=t_logits, s_feat=s_feat0, t_feat=t_feat0, targets=targets)
    for epoch in (0, 1, 3, 7):
        s_logits = s_logits0.clone().requires_grad_(True)
        s_feat = s_feat0.clone().requires_grad_(True)
        t_feat = t_feat0.clone().requires_grad_(True)
        L = ref_opt.OptimizedDistillationLoss(alpha=0.7, beta=0.2, gamma=0.1, temperature=4.0, vocab_size=Vl)
        L.epoch = epoch
        total, d = L({"logits": s_logits, "encoder_features": s_feat, "hidden_states": None},
                     {"logits": t_logits, "encoder_features": t_feat, "hidden_states": None}, targets)
        total.backward()
        out[f"e{epoch}_values"] = np.array([d[k] for k in ("total_loss", "token_kd_loss", "feature_kd_loss", "hidden_kd_loss",
                                                            "kd_loss", "hard_loss", "ce_loss")], np.float64)
        out[f"e{epoch}_dlogits"] = s_logits.grad
        out[f"e{epoch}_dsfeat"] = s_feat.grad if s_feat.grad is not None else torch.zeros_like(s_feat0)
        out[f"e{epoch}_dtfeat"] = t_feat.grad if t_feat.grad is not None else torch.zeros_like(t_feat0)
    npz("optloss.npz", **out)


from oracle.make_golden_keys import OPT_RECIPE_KEYS  # noqa: E402


def golden_optimized_recipe(B=8, steps=5, total_steps=20, lr=1e-3):
    """N4: the reference's "optimized" training recipe (/root/reference/src/train_student_kd_optimized.py:338-378, :400-452) run
    for `steps` optimizer steps on its own classes — CompactCaptioningStudent + OptimizedDistillationLoss, AdamW with the three
    parameter groups (lr x0.1 / x1 / x1.5, weight decay 0.01 / 0.01 / 0.005), OneCycleLR (pct_start 0.1, cos, div 10, final div
    100; torch's default cycle_momentum cycles Adam's beta1 0.95 <-> 0.85), clip_grad_norm_ 1.0 on the student and per projector.
    fp32 on the CPU (no autocast / scaler: as in the other goldens), dropout p = 0, the loss's epoch attribute 0 for the first
    two steps and 2 afterwards.  Recorded: the schedule (lr and beta1 per group per step), the loss terms per step, the gradient
    norms, and slices of the parameter CHANGE after the last step."""
    import types
    import torchvision
    tr = types.ModuleType("torchvision.transforms")
    sys.modules["torchvision.transforms"] = tr
    torchvision.transforms = tr
    dl = types.ModuleType("data_loader")
    dl.get_loader = None
    sys.modules["data_loader"] = dl
    import student_model_compact as ref_compact
    import train_student_kd_optimized as ref_opt
    from torch.optim.lr_scheduler import OneCycleLR
    torch.manual_seed(0)
    t = build_teacher()
    s = ref_compact.CompactCaptioningStudent(V, 256, 256, 1, use_attention_refinement=False)
    apply_seeded_init(s, seed=7)
    zero_dropout(s)
    s.train()
    projectors = ref_kd.create_feature_projectors(t, s)
    apply_seeded_init(projectors["encoder"], seed=2)
    zero_dropout(projectors["encoder"])
    L = ref_opt.OptimizedDistillationLoss(alpha=0.7, beta=0.2, gamma=0.1, temperature=4.0, vocab_size=V)
    other = [q for pr in projectors.values() for q in pr.parameters()]
    opt = torch.optim.AdamW([{"params": list(s.encoder.parameters()), "lr": lr * 0.1, "weight_decay": 0.01},
                             {"params": list(s.decoder.parameters()), "lr": lr, "weight_decay": 0.01},
                             {"params": other, "lr": lr * 1.5, "weight_decay": 0.005}], betas=(0.9, 0.999), eps=1e-8)
    sched = OneCycleLR(opt, max_lr=[lr * 0.1, lr, lr * 1.5], total_steps=total_steps, pct_start=0.1, anneal_strategy="cos",
                       div_factor=10, final_div_factor=100)
    named = dict(s.named_parameters())
    named.update({f"projector.{k}": q for k, q in projectors["encoder"].named_parameters()})
    init = {k: named[k].detach().clone() for k in OPT_RECIPE_KEYS}
    images, caps = synthetic_batch(B, V, T1, seed=4321)
    cin, ctg = caps[:-1], caps[1:]
    tw = ref_kd.TeacherWrapper(t)
    out = dict(B=B, total_steps=total_steps, lr=lr)
    lrs, b1s, vals, norms = [], [], [], []
    for k in range(steps):
        L.epoch = 0 if k < 2 else 2
        lrs.append([g["lr"] for g in opt.param_groups])
        b1s.append([g["betas"][0] for g in opt.param_groups])
        with torch.no_grad():
            t_out = tw(images.float(), cin.long())
        logits, enc, hids, _ = s(images, cin)
        t_out["encoder_features"] = projectors["encoder"](t_out["encoder_features"])
        loss, d = L({"logits": logits, "encoder_features": enc, "hidden_states": hids}, t_out, ctg)
        loss.backward()
        n1 = torch.nn.utils.clip_grad_norm_(s.parameters(), max_norm=1.0)
        n2 = torch.nn.utils.clip_grad_norm_(projectors["encoder"].parameters(), max_norm=1.0)
        opt.step()
        opt.zero_grad()
        sched.step()
        vals.append([d[q] for q in ("total_loss", "token_kd_loss", "feature_kd_loss", "hidden_kd_loss", "kd_loss", "hard_loss", "ce_loss")])
        norms.append([float(n1), float(n2)])
        print("step", k, L.epoch, vals[-1][0], norms[-1], lrs[-1][1], b1s[-1][1])
    out.update(lrs=np.array(lrs, np.float64), beta1s=np.array(b1s, np.float64), values=np.array(vals, np.float64), norms=np.array(norms, np.float64))
    for k, sl in OPT_RECIPE_KEYS.items():
        out["delta:" + k] = (named[k].detach() - init[k])[sl]
    npz("optimized_recipe.npz", **out)


def golden_param_counts():
    """SURVEY.md §0 fact 10 — pins the architecture sizes."""
    torch.manual_seed(0)
    s = ref_student.CaptioningStudent(3000)
    tot, tr = ref_student.count_parameters(s)
    t = ref_teacher.CaptioningTeacher(3000, embed_size=512, num_heads=8, num_decoder_layers=4, dropout=0.15)
    tt = sum(p.numel() for p in t.parameters())
    keys = sorted(s.state_dict().keys())
    tkeys = sorted(t.state_dict().keys())
    npz("param_counts.npz", student_total=tot, student_trainable=tr, teacher_total=tt,
        student_keys=np.array(keys), teacher_keys=np.array(tkeys),
        student_shapes=np.array([str(tuple(s.state_dict()[k].shape)) for k in keys]),
        teacher_shapes=np.array([str(tuple(t.state_dict()[k].shape)) for k in tkeys]))


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    which = sys.argv[1:] or ["counts", "losses", "projector", "refinement", "decoders", "cfg1", "teacher", "kd_step", "kd_step_b16", "compact", "beam", "optloss"]
    fns = {"counts": golden_param_counts, "losses": golden_losses, "projector": golden_projector,
           "refinement": golden_refinement, "decoders": golden_decoders, "cfg1": golden_cfg1,
           "teacher": golden_teacher, "kd_step": golden_kd_step, "kd_step_b16": golden_kd_step_b16, "kd_step_b16_autocast": golden_kd_step_b16_autocast, "kd_step_cfg5_b16": golden_kd_step_cfg5_b16, "compact": golden_compact, "compact_layers": golden_compact_decoder_layers, "compact_b16": golden_compact_b16, "beam": golden_beam, "optloss": golden_optloss, "optimized_recipe": golden_optimized_recipe}
    for w in which:
        fns[w]()
