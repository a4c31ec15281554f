"""CPU-only checks: the C-ABI library loads and exports every symbol include/ick.h declares, argument
validation happens before any HIP call, the host logic (schedule, flat buffers, seeded init, API surface)
behaves like the reference's, and the product path refuses to run without the GPU (no silent fallback)."""
import ctypes
import io
import contextlib
import os

import numpy as np
import pytest
import torch

from conftest import ROOT, load_golden


def test_header_symbols_are_exported():
    from imagecaptioner_amd import _lib
    protos = _lib.parse_header()
    assert len(protos) >= 35 and "ick_gemm_f32" in protos and "ick_token_kd_ce" in protos
    L = ctypes.CDLL(_lib.LIB_PATH)
    for name in protos:
        assert hasattr(L, name), f"libick.so does not export {name} declared in include/ick.h"
    assert _lib.lib().ick_abi_version() == _lib.ABI_VERSION == 8
    # every extern "C" entry of the sources is declared in the header (no undeclared ABI)
    import glob
    import re
    defined = set()
    for f in glob.glob(os.path.join(ROOT, "imagecaptioner_amd", "csrc", "*.hip")):
        defined |= set(re.findall(r"^(?:extern \"C\" )?(?:int|const char\*) (ick_\w+)\(", open(f).read(), flags=re.M))
    assert defined == set(protos), (defined ^ set(protos))


def test_argument_validation_needs_no_gpu():
    from imagecaptioner_amd import _lib
    L = _lib.lib()
    assert L.ick_gemm_f32(None, None) < 0
    assert b"null descriptor" in L.ick_last_error()
    d = _lib.IckGemm()
    d.A, d.B, d.C = 16, 16, 16
    d.M, d.N, d.K = 4, 4, 0
    assert L.ick_gemm_f32(ctypes.byref(d), None) < 0 and b"empty problem" in L.ick_last_error()
    d.K, d.op, d.lda, d.ldb = 6, _lib.OP_NT, 6, 8
    assert L.ick_gemm_f32(ctypes.byref(d), None) < 0 and b"multiples of 4" in L.ick_last_error()
    assert L.ick_layernorm_fwd(16, 16, 16, 16, None, None, 4, 6, 1e-5, None) < 0      # D % 4
    assert L.ick_token_kd_ce(16, 16, None, 16, 16, 16, 16, 4, 30000, 4.0, 1.0, 0.0, None) < 0
    assert b"too large" in L.ick_last_error()
    with pytest.raises(_lib.IckError):
        _lib.check(L.ick_adamw_step(None, None, None, None, 0, 0.0, 0.9, 0.999, 1e-8, 0.0, 1, None, 1.0, 1.0, 0, None, None, None))


def test_gemm_struct_layout_matches_header():
    from imagecaptioner_amd import _lib
    # field order of the ctypes mirror == field order in the header's struct
    import re
    src = open(_lib.HEADER).read()
    body = src[src.index("typedef struct IckGemm {"):src.index("} IckGemm;")]
    body = re.sub(r"/\*.*?\*/", " ", body, flags=re.S)
    names = []
    for decl in body.split("{", 1)[1].split(";"):
        decl = decl.strip()
        if not decl:
            continue
        parts = decl.replace("*", " ").split()
        rest = decl.split(None, 2 if parts[0] == "const" else 1)[-1]
        for nm in rest.split(","):
            names.append(nm.replace("*", "").strip().split()[-1])
    assert names == [f[0] for f in _lib.IckGemm._fields_]


def test_product_path_refuses_cpu_tensors():
    from imagecaptioner_amd import ops
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.linear_fwd(torch.zeros(4, 8), torch.zeros(4, 8))


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from imagecaptioner_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="There is no CPU fallback"):
        _lib.lib()


def test_cosine_warm_restarts_matches_torch():
    from imagecaptioner_amd.train_student_kd import cosine_warm_restarts_factor as f
    p = torch.nn.Parameter(torch.zeros(1))
    o = torch.optim.SGD([p], lr=2e-4)
    s = torch.optim.lr_scheduler.CosineAnnealingWarmRestarts(o, T_0=5, T_mult=2, eta_min=1e-6)
    for ep in np.concatenate([np.linspace(0, 36, 97), [4.999, 5.0, 14.999, 15.0, 35.0]]):
        s.step(float(ep))
        assert abs(o.param_groups[0]["lr"] - (1e-6 + (2e-4 - 1e-6) * f(float(ep)))) < 1e-12


def test_state_dict_surface_matches_reference_on_cpu():
    from imagecaptioner_amd.student_model import CaptioningStudent, count_parameters
    from imagecaptioner_amd.teacher_model import CaptioningTeacher
    g = load_golden("param_counts.npz")
    s = CaptioningStudent(3000)
    assert sorted(s.state_dict().keys()) == sorted(g["student_keys"].tolist())
    assert count_parameters(s) == (int(g["student_total"]), int(g["student_trainable"]))
    assert s.encoder.adaptive_pool.output_size == (7, 7) and s.embed_size == 256 and s.hidden_size == 512
    assert not any(p.requires_grad for n, p in s.named_parameters() if n.startswith("encoder.resnet.5."))
    assert all(p.requires_grad for n, p in s.named_parameters() if n.startswith("encoder.resnet.6."))
    t = CaptioningTeacher(3000, embed_size=512, num_heads=8, num_decoder_layers=4, dropout=0.15)
    assert sorted(t.state_dict().keys()) == sorted(g["teacher_keys"].tolist())
    assert t.encoder.num_features == 384 and t.encoder_projection.out_features == 512
    assert getattr(t, "embed_size", 512) == 512          # the reference teacher does not set it either
    # conv weights keep logical OIHW shapes (state_dict compatible) but live channels_last for the kernels
    w = s.encoder.resnet[6][0].conv2.weight
    assert tuple(w.shape) == (256, 256, 3, 3) and w.is_contiguous(memory_format=torch.channels_last)
    sd = {k: v.clone() for k, v in s.state_dict().items()}
    s.load_state_dict(sd)
    assert s.encoder.resnet[6][0].conv2.weight.is_contiguous(memory_format=torch.channels_last)


def test_compact_student_surface_matches_reference_on_cpu():
    """N4: CompactCaptioningStudent keeps the reference's state_dict keys, parameter counts and frozen set (captured from the
    reference class in tests/golden/compact_student.npz) — parameter holders only, no compute on the CPU."""
    import numpy as np
    from conftest import load_golden
    from imagecaptioner_amd.student_model_compact import CompactCaptioningStudent, count_parameters
    g = load_golden("compact_student.npz")
    m = CompactCaptioningStudent(5000, 256, 256, 1)
    assert sorted(m.state_dict().keys()) == sorted(g["keys"].tolist())
    assert count_parameters(m) == (int(g["total_params"]), int(g["trainable_params"]))
    frozen = [k for k, p in m.named_parameters() if not p.requires_grad]
    assert frozen and all(int(k.split(".")[2]) < 10 for k in frozen)          # features[0:10] (student_model_compact.py:26-30)
    assert m.encoder.adaptive_pool.output_size == (7, 7) and m.decoder.lstm.weight_hh_l0.shape == (1024, 256)
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 3, 224, 224), torch.zeros(3, 1, dtype=torch.long))   # no CPU fallback


def test_create_feature_projectors_and_helpers():
    from imagecaptioner_amd import distillation_utils as D
    from imagecaptioner_amd.student_model import CaptioningStudent
    from imagecaptioner_amd.teacher_model import CaptioningTeacher
    s, t = CaptioningStudent(100, 128, 256, 1, use_attention_refinement=False), CaptioningTeacher(100, 512, 8, 1)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        pr = D.create_feature_projectors(t, s)
    assert "512 -> 128, seq_len: 197 -> 49" in buf.getvalue()
    enc = pr["encoder"]
    assert (enc.teacher_dim, enc.student_dim, enc.teacher_seq_len, enc.student_seq_len) == (512, 128, 197, 49)
    assert sorted(enc.state_dict()) == ["feature_projection.0.bias", "feature_projection.0.weight",
                                        "feature_projection.3.bias", "feature_projection.3.weight"]
    assert isinstance(pr["hidden"].seq_projection, torch.nn.AdaptiveAvgPool1d)      # 197 -> 64 default, never called
    L = D.DistillationLoss()
    assert (L.alpha, L.beta, L.gamma, L.temperature, L.vocab_size) == (0.7, 0.2, 0.1, 4.0, None)
    assert abs((1 - L.alpha - L.beta - L.gamma) - 2.7755575615628914e-17) < 1e-30

    class V:
        itos = {0: "<PAD>", 1: "<START>", 2: "<END>", 3: "<UNK>", 4: "a", 5: "dog", 6: "runs"}
        stoi = {w: i for i, w in itos.items()}
    assert D.compute_bleu_score([1, 4, 5, 2, 0], [1, 4, 5, 6, 2], V) == pytest.approx(2 / 3)
    assert D.compute_bleu_score([4], [0, 1, 2], V) == 0.0
    out = io.StringIO()
    with contextlib.redirect_stdout(out):
        d = dict(total_loss=1.0, ce_loss=2.0, token_kd_loss=3.0, feature_kd_loss=4.0, hidden_kd_loss=0.0)
        D.log_training_progress(1, 50, d, 1e-4, 100)
        D.log_training_progress(1, 51, d, 1e-4, 100)
    assert out.getvalue().count("Token KD: 3.0000") == 1
    with pytest.raises(RuntimeError, match="no CPU fallback"):       # beam search exists, but only on the HIP path
        t.caption_image(torch.zeros(3, 224, 224), V)


def test_seeded_init_is_order_independent_and_batch_layout():
    from imagecaptioner_amd.utils.seeded_init import seeded_state_dict, seeded_tensor, synthetic_batch
    a = seeded_state_dict({"x.weight": (4, 3), "y.bias": (5,)}, seed=7)
    b = seeded_state_dict({"y.bias": (5,), "zzz.weight": (2, 2), "x.weight": (4, 3)}, seed=7)
    assert torch.equal(a["x.weight"], b["x.weight"]) and torch.equal(a["y.bias"], b["y.bias"])
    assert seeded_tensor("bn.num_batches_tracked", (), 0) is None and seeded_tensor("pos_encoder.pe", (5000, 1, 8), 0) is None
    assert (seeded_tensor("bn1.running_var", (64,), 0) >= 0.5).all()
    images, caps = synthetic_batch(5, 5000, 16, seed=1234, rank=3)
    assert images.shape == (5, 3, 224, 224) and caps.shape == (16, 5) and caps.dtype == torch.int64
    assert (caps[0] == 1).all()
    for b_ in range(5):
        col = caps[:, b_]
        n = int((col != 0).sum())
        assert 8 <= n <= 16 and col[n - 1] == 2 and (col[n:] == 0).all() and (col[1:n - 1] >= 4).all()
    i2, c2 = synthetic_batch(5, 5000, 16, seed=1234, rank=4)
    assert not torch.equal(images, i2)


def test_flat_params_alias_parameters_and_grads():
    from imagecaptioner_amd.train_student_kd import FlatParams
    lin = torch.nn.Linear(6, 5)
    conv_w = torch.nn.Parameter(torch.randn(8, 4, 3, 3).contiguous(memory_format=torch.channels_last))
    frozen = torch.nn.Parameter(torch.randn(3), requires_grad=False)
    w0, c0 = lin.weight.detach().clone(), conv_w.detach().clone()
    fp = FlatParams([("a", [lin.weight, lin.bias, frozen]), ("b", [conv_w, lin.weight])], torch.device("cpu"))
    assert fp.segments == [("a", 0, 32 + 8), ("b", 40, 40 + 288)] and fp.total == 328
    assert torch.equal(lin.weight, w0) and torch.equal(conv_w, c0)
    assert conv_w.is_contiguous(memory_format=torch.channels_last) and conv_w.grad.is_contiguous(memory_format=torch.channels_last)
    fp.param[0] = 42.0
    assert lin.weight[0, 0].item() == 42.0
    conv_w.grad[1, 2, 0, 1] = 3.0        # logical (co=1, ci=2, r=0, s=1) -> physical [co][r][s][ci]
    assert fp.grad[40 + ((1 * 3 + 0) * 3 + 1) * 4 + 2].item() == 3.0
    fp.grad.zero_()
    assert conv_w.grad.abs().sum().item() == 0.0 and frozen.grad is None


def test_checkpoint_has_the_reference_layout(tmp_path):
    """KDTrainer.checkpoint() == the dict train_student_kd.py:359-380 saves; torch's AdamW accepts the optimizer part."""
    import contextlib
    import io
    from imagecaptioner_amd import distillation_utils as D
    from imagecaptioner_amd.student_model import CaptioningStudent
    from imagecaptioner_amd.teacher_model import CaptioningTeacher
    from imagecaptioner_amd.train_student_kd import KDTrainer
    s, t = CaptioningStudent(60, 128, 256, 2), CaptioningTeacher(60, 512, 8, 1)
    with contextlib.redirect_stdout(io.StringIO()):
        pr = D.create_feature_projectors(t, s)
    tr = KDTrainer(s, t, pr, vocab_size=60, batch_size=2, t_plus_1=8, use_graph=False)
    tr.step_count = 3
    tr.applied_steps_dev.fill_(3)
    tr.flat.exp_avg.uniform_(-1, 1)
    tr.flat.exp_avg_sq.uniform_(0, 1)
    ck = tr.checkpoint(epoch=4, val_loss=1.25, val_bleu=0.5)
    assert set(ck) == {"epoch", "student_state_dict", "projectors_state_dict", "optimizer_state_dict", "scheduler_state_dict",
                       "val_loss", "val_bleu", "vocab_size", "model_config", "distillation_config"}
    assert ck["model_config"] == {"embed_size": 128, "hidden_size": 256, "num_layers": 2, "dropout": 0.2}
    assert ck["distillation_config"] == {"alpha": 0.7, "beta": 0.2, "gamma": 0.1, "temperature": 4.0}
    assert set(ck["projectors_state_dict"]) == {"encoder", "hidden"}
    assert sorted(ck["student_state_dict"]) == sorted(s.state_dict())
    w = ck["student_state_dict"]["encoder.resnet.6.0.conv2.weight"]
    assert tuple(w.shape) == (256, 256, 3, 3) and w.is_contiguous()           # plain NCHW tensors on disk
    # the reference's optimizer construction (train_student_kd.py:219-234) loads our optimizer state
    other = list(s.attention_refinement.parameters()) + [p for m in pr.values() for p in m.parameters()]
    opt = torch.optim.AdamW([{"params": list(s.encoder.parameters()), "lr": 2e-5}, {"params": list(s.decoder.parameters()), "lr": 2e-4},
                             {"params": other, "lr": 2e-4}], weight_decay=0.01)
    opt.load_state_dict(ck["optimizer_state_dict"])
    conv = s.encoder.resnet[6][0].conv2.weight
    assert torch.equal(opt.state[conv]["exp_avg"], conv.grad * 0 + [v for p_, o, n in tr.flat.metas if p_ is conv
                                                                     for v in [tr.flat.exp_avg[o:o + n].view(256, 3, 3, 256).permute(0, 3, 1, 2)]][0])
    assert s.encoder.resnet[0].weight not in opt.state                         # frozen stem: no moments
    path = tmp_path / "best_student_model.pth"
    tr.save_checkpoint(str(path), epoch=4, val_loss=1.25, val_bleu=0.5)
    back = torch.load(str(path), weights_only=True)
    s2 = CaptioningStudent(60, 128, 256, 2)
    s2.load_state_dict(back["student_state_dict"])
    assert torch.equal(s2.decoder.lstm.weight_hh_l1, s.decoder.lstm.weight_hh_l1)
    assert s2.encoder.resnet[7][0].conv1.weight.is_contiguous(memory_format=torch.channels_last)


def test_evaluator_metrics_known_answers():
    """StudentEvaluator.bleu_score / meteor_score_simple (reference evaluate_student.py:30-69): clipped n-gram precision
    against one reference without brevity penalty, and the F1 of the two word sets — known answers worked by hand."""
    from imagecaptioner_amd.evaluate_student import StudentEvaluator

    class _M:
        def eval(self):
            return self

    class _V:
        itos = {0: "<PAD>", 1: "<START>", 2: "<END>", 3: "<UNK>", 4: "a", 5: "dog", 6: "runs"}
        stoi = {v: k for k, v in itos.items()}

    ev = StudentEvaluator(_M(), None, _V(), "cpu")
    assert ev.bleu_score("a dog runs on grass", "a dog runs fast", 1) == pytest.approx(3 / 5)
    assert ev.bleu_score("a dog runs on grass", "a dog runs fast", 2) == pytest.approx(2 / 4)
    assert ev.bleu_score("a a a", "a dog", 1) == pytest.approx(1 / 3)          # clipping by the reference count
    assert ev.bleu_score("dog", "a dog runs", 2) == 0.0                         # shorter than n
    assert ev.bleu_score("A Dog", "a dog", 1) == 1.0                            # case-insensitive
    p, r = 3 / 5, 3 / 4
    assert ev.meteor_score_simple("a dog runs on grass", "a dog runs fast") == pytest.approx(2 * p * r / (p + r))
    assert ev.meteor_score_simple("", "a dog") == 0.0 and ev.meteor_score_simple("cat", "") == 0.0
    assert ev.reference_caption([1, 4, 5, 6, 2, 0, 0]) == "a dog runs"
    assert ev._words([4, 5, 2, 6]) == ["a", "dog"]


def test_optimized_loss_warmup_schedule_host_logic():
    """OptimizedDistillationLoss.current_weights = the reference's adaptive weights (train_student_kd_optimized.py:63-66)."""
    from imagecaptioner_amd.train_student_kd_optimized import KEYS, OptimizedDistillationLoss
    L = OptimizedDistillationLoss(alpha=0.7, beta=0.2, gamma=0.1)
    assert KEYS == ("total_loss", "token_kd_loss", "feature_kd_loss", "hidden_kd_loss", "kd_loss", "hard_loss", "ce_loss")
    L.epoch = 0
    assert L.current_weights() == pytest.approx((0.9, 0.0, 0.0))
    L.epoch = 1.5
    assert L.current_weights() == pytest.approx((0.7 * 0.5 + 0.45, 0.1, 0.05))
    L.epoch = 30
    assert L.current_weights() == pytest.approx((0.7, 0.2, 0.1))
    with pytest.raises(RuntimeError):           # no GPU here: the loss has no CPU fallback
        import torch
        z = torch.zeros(2, 2, 5000)
        L({"logits": z}, {"logits": z}, torch.zeros(2, 2, dtype=torch.long))


def test_gpu_transform_refuses_cpu():
    import numpy as np
    import torch
    from imagecaptioner_amd.data_pipeline import GpuImageTransform
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError):
        GpuImageTransform(train=False, device="cpu")([np.zeros((10, 10, 3), np.uint8)])


def test_bench_roofline_denominators():
    """bench.py's roofline denominators: fp32 MFMA peak for the exact path; FLOP-weighted blend (teacher on the fp32
    MFMA, student on the bf16 MFMA or 3 bf16 MFMAs per product) for the mixed-precision modes."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("bench", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    assert b.GFLOP_PER_IMAGE == pytest.approx(b.GFLOP_TEACHER + b.GFLOP_STUDENT, abs=0.02)
    assert b.mfma_peak("f32") == pytest.approx(157.3, rel=1e-3)
    t = 9.80 / 157.3 + 18.61 / 2516.0
    assert b.mfma_peak("bf16") == pytest.approx(28.41 / t, rel=1e-3)
    assert b.mfma_peak("bf16") > b.mfma_peak("bf16x3") > b.mfma_peak("f32")
    assert b.mfma_peak("f32", 20.95) == pytest.approx(157.3, rel=1e-3)      # cfg5 split, same peak for pure fp32


def test_one_cycle_schedule_equals_torch_onecyclelr():
    """OptimizedKDTrainer's host-side schedule against torch.optim.lr_scheduler.OneCycleLR with the reference's settings
    (train_student_kd_optimized.py:369-378; cycle_momentum left at torch's default True): learning rate and Adam's beta1."""
    import torch
    from torch.optim.lr_scheduler import OneCycleLR
    from imagecaptioner_amd.train_student_kd_optimized import one_cycle
    for total, max_lr in ((37, 3e-3), (20, 1e-3), (1000, 5e-4)):
        p = torch.nn.Parameter(torch.zeros(1))
        o = torch.optim.AdamW([p], lr=1e-3)
        s = OneCycleLR(o, max_lr=max_lr, total_steps=total, pct_start=0.1, anneal_strategy="cos", div_factor=10, final_div_factor=100)
        for k in range(total):
            lr, b1 = one_cycle(k, total, max_lr)
            assert abs(lr - o.param_groups[0]["lr"]) < 1e-12 and abs(b1 - o.param_groups[0]["betas"][0]) < 1e-12, (total, k)
            o.step()
            if k < total - 1:
                s.step()
