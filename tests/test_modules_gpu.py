"""GPU parity of the HIP-backed modules (through the C ABI) against golden vectors captured from the
reference's own modules (tests/golden, made by oracle/make_goldens.py) and against the oracle
(oracle/restatement.py) on the same seeded inputs.

Tolerances (north_star): logits and loss terms within 1e-3 fp32 — tests use 2e-4 relative to the tensor's
scale (the kernels are exact-fp32 MFMA, measured error is ~1e-5); argmax / greedy token ids bit-exact.
"""
import numpy as np
import pytest
import torch

from conftest import load_golden, t

pytestmark = pytest.mark.gpu

TOL = 2e-4


def close(a, b, tol=TOL, name=""):
    a = torch.as_tensor(a).detach().double().cpu()
    b = torch.as_tensor(np.asarray(b)).double()
    assert a.shape == b.shape, f"{name}: shape {tuple(a.shape)} vs {tuple(b.shape)}"
    err = (a - b).abs().max().item()
    scale = max(1e-6, b.abs().max().item())
    assert err <= tol * scale, f"{name}: max abs err {err:.3e} vs scale {scale:.3e} (rel {err / scale:.2e})"


def close_l2(a, b, tol, name=""):
    """relative L2 error over the (sliced) tensor — the metric for end-to-end gradients through the train-mode
    ResNet trunk.  Those gradients are ill-conditioned at B=2 (98 samples per BatchNorm channel + ReLU masks): the
    REFERENCE's own fp32 CPU arithmetic sits 1-2e-2 (rel. L2) away from an fp64 evaluation of the same graph
    (tests/diag_grads.py prints both columns; profiles/diag_grads_r01.log), so tighter bounds would test noise."""
    a = torch.as_tensor(a).detach().double().cpu().flatten()
    b = torch.as_tensor(np.asarray(b)).double().flatten()
    assert a.shape == b.shape, name
    err = ((a - b).norm() / b.norm().clamp_min(1e-30)).item()
    assert err <= tol, f"{name}: relative L2 error {err:.3e} > {tol:.1e}"


def seeded(module, seed, prefix=""):
    from imagecaptioner_amd.utils.seeded_init import apply_seeded_init
    return apply_seeded_init(module, seed, prefix)


# ----------------------------------------------------------------------------- losses
@pytest.mark.parametrize("tau", [3, 4])
def test_losses_vs_reference_golden(tau):
    from imagecaptioner_amd.distillation_utils import DistillationLoss
    g = load_golden(f"losses_tau{tau}.npz")
    V = g["s"].shape[-1]
    s = t(g["s"]).cuda().requires_grad_(True)
    tt = t(g["t"]).cuda()
    sf = t(g["sf"]).cuda().requires_grad_(True)
    tf = t(g["tf"]).cuda().requires_grad_(True)
    sh = [x.cuda().requires_grad_(True) for x in t(g["sh"])]
    th = [x.cuda() for x in t(g["th"])]
    L = DistillationLoss(alpha=0.5, beta=0.2, gamma=0.1, temperature=float(tau), vocab_size=V)
    close(L.token_level_distillation(s, tt), g["kl"], name="kl")
    close(L.encoder_feature_distillation(sf, tf), g["feat"], name="feat")
    close(L.decoder_hidden_state_distillation(sh, th), g["hid"], name="hid")
    total, parts = L({"logits": s, "encoder_features": sf, "hidden_states": sh},
                     {"logits": tt, "encoder_features": tf, "hidden_states": th}, t(g["targets"]).cuda())
    close(total, g["total"], name="total")
    assert abs(parts["ce_loss"] - float(g["ce"])) < 1e-4 * abs(float(g["ce"]))
    assert set(parts) == {"total_loss", "ce_loss", "token_kd_loss", "feature_kd_loss", "hidden_kd_loss"}
    (total * 3.0).backward()      # non-unit upstream gradient exercises the scale-by-device-scalar path
    close(s.grad / 3.0, g["ds"], name="ds")
    close(sf.grad / 3.0, g["dsf"], name="dsf")
    close(tf.grad / 3.0, g["dtf"], name="dtf")
    close(torch.stack([x.grad if x.grad is not None else torch.zeros_like(x) for x in sh]) / 3.0, g["dsh"], name="dsh")
    L2 = DistillationLoss(vocab_size=V, temperature=float(tau))
    total2, parts2 = L2({"logits": s.detach(), "encoder_features": sf.detach(), "hidden_states": sh},
                        {"logits": tt, "encoder_features": tf.detach(), "hidden_states": None}, t(g["targets"]).cuda())
    close(total2, g["total_default"], name="total_default")
    assert parts2["hidden_kd_loss"] == 0.0


def test_loss_error_behaviour():
    from imagecaptioner_amd.distillation_utils import DistillationLoss
    L = DistillationLoss(vocab_size=8)
    with pytest.raises(ValueError):
        L.encoder_feature_distillation(torch.zeros(2, 49, 8, device="cuda"), torch.zeros(2, 49, 12, device="cuda"))
    with pytest.raises(ValueError):
        L.decoder_hidden_state_distillation([torch.zeros(2, 8, device="cuda")], [torch.zeros(2, 12, device="cuda")])
    with pytest.raises(KeyError):
        L({"encoder_features": None}, {"logits": torch.zeros(2, 2, 8, device="cuda")}, torch.zeros(2, 2, dtype=torch.long, device="cuda"))
    nc = torch.zeros(2, 8, 3, device="cuda").permute(0, 2, 1)            # (2,3,8) non-contiguous
    with pytest.raises(RuntimeError):
        L({"logits": nc}, {"logits": nc.contiguous()}, torch.zeros(2, 3, dtype=torch.long, device="cuda"))


# ----------------------------------------------------------------------------- projector / refinement / decoder
@pytest.mark.parametrize("E", [128, 256, 384])
def test_projector(E):
    from imagecaptioner_amd.distillation_utils import FeatureProjector
    g = load_golden(f"projector_E{E}.npz")
    pr = seeded(FeatureProjector(512, E, 197, 49), 2).cuda().eval()
    x = torch.randn(2, 197, 512, generator=torch.Generator().manual_seed(int(g["x_seed"]))).cuda().requires_grad_(True)
    y = pr(x)
    close(y, g["y"], name="y")
    (y * t(g["gy"]).cuda()).sum().backward()
    close(x.grad[:, ::8, ::8], g["dx"], name="dx")
    close(pr.feature_projection[0].weight.grad[::16], g["dw"], name="dw")
    close(pr.feature_projection[0].bias.grad, g["db"], name="db")
    close(pr.feature_projection[3].weight.grad, g["dlnw"], name="dlnw")


@pytest.mark.parametrize("E", [256, 384])
def test_refinement(E):
    from imagecaptioner_amd.student_model import AttentionRefinement
    g = load_golden(f"refinement_E{E}.npz")
    r = seeded(AttentionRefinement(E), 4, "attention_refinement.").cuda().eval()
    x = t(g["x"]).cuda().requires_grad_(True)
    y = r(x)
    close(y, g["y"], name="y")
    (y * t(g["gy"]).cuda()).sum().backward()
    close(x.grad, g["dx"], name="dx")
    close(r.attention.in_proj_weight.grad[::37], g["d_inproj"], name="d_inproj")
    close(r.attention.in_proj_bias.grad, g["d_inproj_b"], name="d_inproj_b")
    close(r.attention.out_proj.weight.grad[::29], g["d_outproj"], name="d_outproj")
    close(r.ffn[0].weight.grad[::41], g["d_ffn0"], name="d_ffn0")
    close(r.norm1.weight.grad, g["d_norm1_w"], name="d_norm1_w")
    close(r.norm2.bias.grad, g["d_norm2_b"], name="d_norm2_b")


@pytest.mark.parametrize("E,H,L", [(128, 256, 1), (256, 512, 2), (384, 768, 3)])
def test_decoder_fwd_bwd(E, H, L):
    from imagecaptioner_amd.student_model import LSTMDecoder
    g = load_golden(f"decoder_E{E}_H{H}_L{L}.npz")
    dec = seeded(LSTMDecoder(1000, E, H, L, dropout=0.2), 3, "decoder.").cuda().eval()
    feats = t(g["feats"]).cuda().requires_grad_(True)
    caps = t(g["caps"]).cuda()
    logits, hids, attw = dec(feats, caps)
    assert isinstance(hids, list) and len(hids) == caps.shape[0] and tuple(hids[0].shape) == (3, H)
    close(logits, g["logits"], name="logits")
    close(torch.stack(hids), g["hids"], name="hids")
    close(torch.stack(attw), g["attw"], name="attw")
    ((logits * t(g["gl"]).cuda()).sum() + (torch.stack(hids) * t(g["gh"]).cuda()).sum()).backward()
    close(feats.grad, g["dfeats"], name="dfeats")
    close(dec.lstm.weight_hh_l0.grad[:8], g["d_whh0"], name="d_whh0")
    close(dec.lstm.weight_ih_l0.grad[:8], g["d_wih0"], name="d_wih0")
    close(dec.attention.weight.grad[:8], g["d_att_w"], name="d_att_w")
    close(dec.attention.bias.grad, g["d_att_b"], name="d_att_b")
    close(dec.attention_combine.weight.grad[:8], g["d_comb_w"], name="d_comb_w")
    close(dec.embedding.weight.grad[t(g["emb_row_ids"]).cuda()], g["d_emb_rows"], name="d_emb")
    close(dec.output_projection[0].weight.grad[:8], g["d_out0_w"], name="d_out0_w")
    close(dec.output_projection[3].bias.grad, g["d_out3_b"], name="d_out3_b")
    close(getattr(dec.lstm, f"bias_ih_l{L - 1}").grad, g["d_bih_last"], name="d_bih_last")
    # the single-step entries the reference's caption_image reaches into
    ctx, w = dec.attention_mechanism(torch.zeros(3, H, device="cuda"), feats.detach())
    close(w, g["attw"][0], name="attention_mechanism weights at h=0")


# ----------------------------------------------------------------------------- full models vs reference goldens
def test_state_dict_keys_match_reference():
    from imagecaptioner_amd.student_model import CaptioningStudent, count_parameters
    from imagecaptioner_amd.teacher_model import CaptioningTeacher
    g = load_golden("param_counts.npz")
    s = CaptioningStudent(3000)
    sd = s.state_dict()
    assert sorted(sd.keys()) == sorted(g["student_keys"].tolist())
    shapes = dict(zip(g["student_keys"].tolist(), g["student_shapes"].tolist()))
    for k, v in sd.items():
        assert str(tuple(v.shape)) == shapes[k], k
    assert count_parameters(s) == (int(g["student_total"]), int(g["student_trainable"]))
    tm = CaptioningTeacher(3000, embed_size=512, num_heads=8, num_decoder_layers=4, dropout=0.15)
    assert sorted(tm.state_dict().keys()) == sorted(g["teacher_keys"].tolist())
    assert sum(p.numel() for p in tm.parameters()) == int(g["teacher_total"])


def test_cfg1_student_eval_and_greedy():
    from imagecaptioner_amd.student_model import CaptioningStudent
    from imagecaptioner_amd.utils.seeded_init import synthetic_batch
    g = load_golden("cfg1_student_eval.npz")
    m = seeded(CaptioningStudent(5000, 128, 256, 1, use_attention_refinement=False), 0).cuda().eval()
    images, caps = synthetic_batch(2, 5000, 16, seed=1234)
    with torch.no_grad():
        logits, enc, hids, attw = m(images.cuda(), caps[:-1].cuda())
    close(enc, g["enc"], name="enc")
    close(logits, g["logits"], name="logits")
    assert (logits.cpu() - t(g["logits"])).abs().max().item() < 1e-3          # north_star: within 1e-3 fp32
    close(hids[0], g["hid0"], name="hid0")
    close(hids[14], g["hid14"], name="hid14")
    close(attw[14], g["attw14"], name="attw14")
    assert torch.equal(logits.argmax(-1).cpu(), t(g["argmax"])), "argmax token ids must be bit-exact"
    ids, _ = m.generate(images.cuda(), max_length=20)
    ids = ids.cpu()
    ref = t(g["greedy_ids"])
    for b in range(2):                       # rows stop at <END> in the reference (B=1 caption_image)
        n = int((ref[:, b] >= 0).sum())
        assert torch.equal(ids[:n, b], ref[:n, b]), f"greedy ids differ for image {b}"
        if n < 20:
            assert int(ids[n, b]) == 2

    class V:
        itos = {i: f"w{i}" for i in range(5000)}
        itos.update({0: "<PAD>", 1: "<START>", 2: "<END>", 3: "<UNK>"})
        stoi = {w: i for i, w in itos.items()}
    words = m.caption_image(images[0].cuda(), V(), max_length=20)
    n0 = int((ref[:, 0] >= 0).sum())
    assert words == [V.itos[int(i)] for i in ref[:n0, 0]]


@pytest.mark.parametrize("teacher_precision", ["f32", "f32x3"])
def test_teacher_eval(teacher_precision):
    """"f32x3": the teacher's Linears as fp32-grade three-fp16-product GEMMs (igemm_glds_impl.h TERMS 4) — held to the SAME
    tolerances against the reference's golden as the exact-fp32 path, argmax ids bit-exact included."""
    from imagecaptioner_amd import ops
    from imagecaptioner_amd.distillation_utils import TeacherWrapper
    from imagecaptioner_amd.teacher_model import CaptioningTeacher
    from imagecaptioner_amd.utils.seeded_init import synthetic_batch
    g = load_golden("teacher_eval.npz")
    tm = seeded(CaptioningTeacher(5000, embed_size=512, num_heads=8, num_decoder_layers=4, dropout=0.15), 1).cuda()
    images, caps = synthetic_batch(2, 5000, 16, seed=1234)
    with ops.precision(teacher_precision):
        out = TeacherWrapper(tm)(images.cuda(), caps[:-1].cuda())
        vit = tm.encoder.forward_features(images.cuda())
    assert out["hidden_states"] is None and not any(p.requires_grad for p in tm.parameters())
    close(vit[:, ::4], g["vit_tokens"], name="vit tokens")
    close(out["encoder_features"][:, ::4], g["enc_feats"], name="enc_feats")
    close(out["logits"], g["logits"], name="logits")
    assert (out["logits"].cpu() - t(g["logits"])).abs().max().item() < 1e-3
    assert torch.equal(out["logits"].argmax(-1).cpu(), t(g["argmax"]))


def _kd_models():
    from imagecaptioner_amd.train_student_kd import build_kd_models
    return build_kd_models(device="cuda")


def test_kd_step_train_mode_vs_reference_golden():
    """cfg3 models, B=2, train mode (BN batch stats, dropout p=0): loss terms, logits, gradients, BN running stats."""
    from imagecaptioner_amd.distillation_utils import DistillationLoss, TeacherWrapper
    from imagecaptioner_amd.utils.seeded_init import synthetic_batch
    g = load_golden("kd_step_cfg3_B2.npz")
    student, teacher, projectors = _kd_models()
    for m in list(student.modules()) + list(projectors["encoder"].modules()):
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    student.attention_refinement.attention.dropout = 0.0
    student.decoder.lstm.dropout = 0.0
    student.train()
    images, caps = synthetic_batch(2, 5000, 16, seed=1234)
    images, caps = images.cuda(), caps.cuda()
    cin, ctg = caps[:-1], caps[1:]
    t_out = TeacherWrapper(teacher)(images, cin)
    logits, enc, hids, _ = student(images, cin)
    t_out["encoder_features"] = projectors["encoder"](t_out["encoder_features"])
    L = DistillationLoss(0.7, 0.2, 0.1, 4.0, 5000)
    loss, parts = L({"logits": logits, "encoder_features": enc, "hidden_states": hids}, t_out, ctg)
    TM = 1e-3    # train-mode BatchNorm over 98 samples/channel amplifies fp32 reorder noise; north_star bound is 1e-3
    close(enc, g["enc"], TM, name="enc (train-mode BN)")
    close(t_out["encoder_features"], g["t_proj"], name="projected teacher feats")
    close(logits[:, :, ::50], g["logits_slice"], TM, name="logits")
    assert (logits[:, :, ::50].cpu() - t(g["logits_slice"])).abs().max().item() < 1e-3
    close(hids[14], g["hid14"], TM, name="hid14")
    close(loss, g["loss"], name="loss")
    for k, gk in (("ce_loss", "ce"), ("token_kd_loss", "kd"), ("feature_kd_loss", "feat"), ("hidden_kd_loss", "hid")):
        assert abs(parts[k] - float(g[gk])) <= 1e-3 * max(1.0, abs(float(g[gk]))), (k, parts[k], float(g[gk]))
    loss.backward()
    sd = dict(student.named_parameters())
    assert all(sd[k].grad is None for k in sd if k.startswith("encoder.resnet.4.") or k.startswith("encoder.resnet.5."))
    gn = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in sd.values() if p.grad is not None))
    close(gn, g["gn_student"], 1e-3, "student grad norm")
    # Gradients downstream of the trunk still depend on the trunk's train-mode FORWARD output (BatchNorm over 98
    # samples per channel at B=2).  Measured: rounding ONE BatchNorm sum of squares with fma instead of mul+add (both
    # valid fp32) moves g_whh0 by 6.9e-3 and the first Adam delta by 1.0e-2 — so 2e-2 here; the same kernels are held
    # to 2e-4 on fixed inputs in test_decoder_fwd_bwd / test_refinement / test_projector.
    GT = 2e-2
    TRUNK = 6e-2    # through the train-mode trunk: the fp32 reference itself is 1-2e-2 from exact (see close_l2)
    close(sd["decoder.lstm.weight_hh_l0"].grad[::64, ::16], g["g_whh0"], GT, "g_whh0")
    close_l2(sd["encoder.resnet.7.2.conv3.weight"].grad[::64, ::16, 0, 0], g["g_l4conv3"], TRUNK, "g_l4conv3")
    close_l2(sd["encoder.resnet.6.0.conv2.weight"].grad[::16, ::16], g["g_l3conv2"], TRUNK, "g_l3conv2")
    close_l2(sd["encoder.resnet.7.0.downsample.0.weight"].grad[::64, ::32, 0, 0], g["g_l4ds"], TRUNK, "g_l4ds")
    close_l2(sd["encoder.resnet.6.0.bn1.weight"].grad, g["g_l3bn1_w"], TRUNK, "g_l3bn1_w")
    close_l2(sd["encoder.resnet.7.2.bn3.bias"].grad, g["g_l4bn3_b"], TRUNK, "g_l4bn3_b")
    close(sd["encoder.projection.0.weight"].grad[::16, ::64], g["g_encproj"], GT, "g_encproj")
    close(sd["attention_refinement.attention.in_proj_weight"].grad[::32, ::16], g["g_ref_inproj"], GT, "g_ref_inproj")
    close(sd["decoder.embedding.weight"].grad[1], g["g_emb_row1"], GT, "g_emb_row1")
    close(sd["decoder.output_projection.3.bias"].grad[::10], g["g_out3_b"], GT, "g_out3_b")
    # the spatial-attention scores are sums of 256 tanh values (|score| up to ~10^2) -> a peaky softmax whose adjoint
    # w_j (dw_j - sum_i w_i dw_i) cancels; measured 1.7e-3 vs the fp32 reference here, 2e-4 in the isolated decoder test
    close(sd["decoder.attention.weight"].grad[::16, ::16], g["g_att_w"], 5e-3, "g_att_w")
    gp = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in projectors["encoder"].parameters()))
    close(gp, g["gn_proj"], 1e-3, "projector grad norm")
    close(student.encoder.resnet[1].running_mean, g["bn1_running_mean"], name="bn1 running_mean (frozen stem still updates)")
    close(student.encoder.resnet[7][2].bn3.running_var, g["l4_bn3_running_var"], name="l4 bn3 running_var")
    assert int(student.encoder.resnet[1].num_batches_tracked) == 1


@pytest.mark.parametrize("use_graph", [False, True])
def test_kd_trainer_full_step_vs_reference_golden(use_graph):
    """clip_grad_norm_ + AdamW(3 LR groups) parameter deltas after ONE step, eager and hipGraph-replayed."""
    from imagecaptioner_amd.train_student_kd import KDTrainer
    from imagecaptioner_amd.utils.seeded_init import synthetic_batch
    g = load_golden("kd_step_cfg3_B2.npz")
    student, teacher, projectors = _kd_models()
    for m in list(student.modules()) + list(projectors["encoder"].modules()):
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    student.attention_refinement.attention.dropout = 0.0
    student.decoder.lstm.dropout = 0.0
    sd = dict(student.named_parameters())
    keys = ("decoder.lstm.weight_hh_l0", "encoder.resnet.7.2.conv3.weight", "encoder.projection.0.weight",
            "attention_refinement.ffn.0.weight")
    before = {k: sd[k].detach().clone() for k in keys}
    pw = projectors["encoder"].feature_projection[0].weight
    pbefore = pw.detach().clone()
    tr = KDTrainer(student, teacher, projectors, vocab_size=5000, batch_size=2, t_plus_1=16, use_graph=use_graph)
    images, caps = synthetic_batch(2, 5000, 16, seed=1234)
    tr.train_step(images.cuda(), caps.cuda())
    d = tr.loss_dict()
    assert abs(d["total_loss"] - float(g["loss"])) < 1e-3 * abs(float(g["loss"]))
    DT = 3e-2       # (see GT in the test above for the conditioning) first Adam step: delta = -lr*g/(|g|+1e-8) - lr*wd*p; entries with |g| ~ eps amplify gradient noise, so
                    # deltas are compared in relative L2 over the slice (max-abs would test those few entries)
    close_l2(sd["decoder.lstm.weight_hh_l0"].detach()[::64, ::16] - before["decoder.lstm.weight_hh_l0"][::64, ::16], g["d_whh0"], DT, "d_whh0")
    close_l2((sd["encoder.resnet.7.2.conv3.weight"].detach() - before["encoder.resnet.7.2.conv3.weight"])[::64, ::16, 0, 0], g["d_l4conv3"], 6e-2, "d_l4conv3")
    close_l2((sd["encoder.projection.0.weight"].detach() - before["encoder.projection.0.weight"])[::16, ::64], g["d_encproj"], DT, "d_encproj")
    close_l2((sd["attention_refinement.ffn.0.weight"].detach() - before["attention_refinement.ffn.0.weight"])[::16, ::16], g["d_ffn0"], DT, "d_ffn0")
    close_l2((pw.detach() - pbefore)[::16, ::16], g["d_projw"], DT, "d_projw")
    assert int(student.encoder.resnet[1].num_batches_tracked) == 1
    # a second step runs (graph replay) and changes the parameters again
    snap = sd["decoder.lstm.weight_hh_l0"].detach().clone()
    tr.train_step()
    torch.cuda.synchronize()
    assert not torch.equal(snap, sd["decoder.lstm.weight_hh_l0"].detach())
    assert tr.step_count == 2


@pytest.mark.parametrize("epoch", [0, 1, 3, 7])
def test_optimized_distillation_loss_vs_reference_golden(epoch):
    """OptimizedDistillationLoss (reference train_student_kd_optimized.py:34-128) on the HIP path vs values
    and gradients captured from the reference class (tests/golden/optloss.npz), for the warm-up epochs 0/1 and the
    steady state; plus the hidden term with injected attention scores against the oracle restatement."""
    from imagecaptioner_amd.train_student_kd_optimized import KEYS, OptimizedDistillationLoss
    from oracle import restatement as R
    g = load_golden("optloss.npz")

    def rel_err(a, b):
        a, b = torch.as_tensor(a).detach().double().cpu(), torch.as_tensor(np.asarray(b) if not torch.is_tensor(b) else b.detach().cpu()).double()
        return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()

    s_logits, s_feat, t_feat = (t(g[k]).cuda().requires_grad_(True) for k in ("s_logits", "s_feat", "t_feat"))
    L = OptimizedDistillationLoss(alpha=0.7, beta=0.2, gamma=0.1, temperature=4.0, vocab_size=257)
    L.epoch = epoch
    total, d = L({"logits": s_logits, "encoder_features": s_feat, "hidden_states": None},
                 {"logits": t(g["t_logits"]).cuda(), "encoder_features": t_feat, "hidden_states": None}, t(g["targets"]).cuda())
    total.backward()
    want = g[f"e{epoch}_values"]
    got = np.array([d[k] for k in KEYS])
    assert np.allclose(got, want, rtol=2e-5, atol=1e-6), (got, want)
    assert rel_err(s_logits.grad, g[f"e{epoch}_dlogits"]) < 2e-5
    if epoch > 0:
        assert rel_err(s_feat.grad, g[f"e{epoch}_dsfeat"]) < 2e-5 and rel_err(t_feat.grad, g[f"e{epoch}_dtfeat"]) < 2e-5
    else:                                             # beta_now = 0 during the first warm-up epoch
        assert float(s_feat.grad.abs().max()) == 0.0
    with pytest.raises(RuntimeError):
        OptimizedDistillationLoss(vocab_size=300)({"logits": s_logits}, {"logits": s_logits.detach()}, t(g["targets"]).cuda())
    # hidden term: injected scores, oracle = restatement (the reference draws them with torch.randn -> unpinnable)
    gen = torch.Generator().manual_seed(epoch)
    sh = [torch.randn(5, 48, generator=gen) for _ in range(6)]
    th = [torch.randn(5, 48, generator=gen) for _ in range(6)]
    aw = torch.randn(6, 5, generator=gen)
    sh_r = [x.clone().requires_grad_(True) for x in sh]
    ref_total, ref_d = R.optimized_distillation_loss({"logits": t(g["s_logits"]), "hidden_states": sh_r},
                                                      {"logits": t(g["t_logits"]), "hidden_states": th}, t(g["targets"]),
                                                      epoch=epoch, attention_weights=aw)
    ref_total.backward()
    sh_d = [x.cuda().requires_grad_(True) for x in sh]
    tot2, d2 = L({"logits": s_logits.detach(), "hidden_states": sh_d}, {"logits": t(g["t_logits"]).cuda(), "hidden_states": [x.cuda() for x in th]},
                 t(g["targets"]).cuda(), attention_weights=aw.cuda())
    assert abs(d2["hidden_kd_loss"] - ref_d["hidden_kd_loss"]) < 1e-5 and abs(d2["total_loss"] - ref_d["total_loss"]) < 2e-5 * abs(ref_d["total_loss"])
    if epoch > 0:
        tot2.backward()
        assert rel_err(torch.stack([x.grad for x in sh_d]), torch.stack([x.grad for x in sh_r])) < 2e-5
