"""teacher beam-search captioning on the HIP path vs captions produced by the reference's own
CaptioningTeacher.caption_image (tests/golden/teacher_beam.npz, made by oracle/make_goldens.py `beam`)."""
import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu


class Vocab:
    def __init__(self, n):
        self.itos = {i: f"w{i}" for i in range(n)}
        self.itos.update({0: "<PAD>", 1: "<START>", 2: "<END>", 3: "<UNK>"})
        self.stoi = {w: i for i, w in self.itos.items()}


@pytest.fixture(scope="module")
def teacher():
    from imagecaptioner_amd.teacher_model import CaptioningTeacher
    from imagecaptioner_amd.utils.seeded_init import apply_seeded_init
    t = apply_seeded_init(CaptioningTeacher(5000, embed_size=512, num_heads=8, num_decoder_layers=4, dropout=0.15), 1)
    return t.cuda().eval()


def test_beam_topk_kernel():
    from imagecaptioner_amd import ops
    g = torch.Generator().manual_seed(0)
    logits = torch.randn(5, 5000, generator=g) * 3
    scores = torch.tensor([0.0, -1.5, float("-inf"), -0.2, -7.0])
    vals, idx = ops.beam_topk(logits.cuda(), scores.cuda(), 5)
    cand = (scores[:, None] + torch.log_softmax(logits, -1)).view(-1)
    rv, ri = torch.topk(cand, 5)
    assert torch.equal(idx.cpu(), ri)
    assert (vals.cpu() - rv).abs().max().item() < 1e-5


def test_decode_last_matches_full_decode(teacher):
    from imagecaptioner_amd.utils.seeded_init import synthetic_batch
    images, caps = synthetic_batch(3, 5000, 16, seed=4321)
    mem = teacher.project_memory(teacher.encoder.forward_features(images.cuda()))
    cin = caps[:7].cuda()
    full = teacher.decode(mem, cin)
    last = teacher.decode_last(mem, cin)
    assert (full[-1] - last).abs().max().item() < 1e-5
    shared = teacher.decode_last(mem[1:2].contiguous(), cin[:, 1:2].expand(7, 4).contiguous())     # one image, 4 "beams"
    assert (shared - last[1:2]).abs().max().item() < 1e-5


@pytest.mark.parametrize("precision", ["f32", "f32x3"])
def test_beam_search_captions_vs_reference(teacher, precision):
    """"f32x3" (three fp16 MFMAs per product, fp32-grade): the same captions as the reference, token for token."""
    from imagecaptioner_amd import ops
    from imagecaptioner_amd.utils.seeded_init import synthetic_batch
    g = load_golden("teacher_beam.npz")
    vocab = Vocab(5000)
    images, _ = synthetic_batch(3, 5000, 16, seed=int(g["batch_seed"]))
    with ops.precision(precision):
        for b in range(3):
            for beam, lp, nret, tag in ((5, 0.6, 3, "b5"), (3, 0.0, 1, "b3"), (1, 0.6, 1, "b1")):
                got = teacher.caption_image(images[b].cuda(), vocab, max_length=12, beam_size=beam, length_penalty=lp,
                                            num_return_sequences=nret)
                assert got == g[f"img{b}_{tag}"].tolist(), (b, tag, got)


@pytest.mark.parametrize("bias", [12.0, 10.5])
def test_beam_search_finishing_logic_vs_reference(teacher, bias):
    """<END> made likely (bias on fc_out, as in the golden run): hypotheses finish at different lengths, the beam shrinks,
    the GNMT length penalty orders the results."""
    from imagecaptioner_amd.utils.seeded_init import synthetic_batch
    g = load_golden("teacher_beam.npz")
    vocab = Vocab(5000)
    images, _ = synthetic_batch(3, 5000, 16, seed=int(g["batch_seed"]))
    with torch.no_grad():
        teacher.fc_out.bias[2] += bias
    try:
        for b in range(3):
            got = teacher.caption_image(images[b].cuda(), vocab, max_length=12, beam_size=5, length_penalty=0.6,
                                        num_return_sequences=5)
            assert got == g[f"img{b}_b5_end{bias}"].tolist(), (b, got)
    finally:
        with torch.no_grad():
            teacher.fc_out.bias[2] -= bias


def test_caption_image_api_errors(teacher):
    class NoStart:
        stoi, itos = {"<END>": 2}, {2: "<END>"}
    with pytest.raises(AssertionError):
        teacher.caption_image(torch.zeros(3, 224, 224), NoStart())


def test_student_evaluator_on_synthetic_loader(teacher):
    """StudentEvaluator.compare_models_on_dataset (reference evaluate_student.py:99-201) over a synthetic loader: the
    batched student decode must give, per image, exactly the caption of the reference-style B=1 caption_image, and the
    result dict has the reference's layout."""
    from imagecaptioner_amd.evaluate_student import StudentEvaluator
    from imagecaptioner_amd.student_model import CaptioningStudent
    from imagecaptioner_amd.utils.seeded_init import apply_seeded_init, synthetic_batch
    vocab = Vocab(5000)
    student = apply_seeded_init(CaptioningStudent(5000, 256, 512, 2), 7).cuda().eval()
    with torch.no_grad():   # margins large enough for tie-free argmax (see test_configs_gpu.py)
        student.decoder.output_projection[3].weight.mul_(16.0)
        student.decoder.embedding.weight.mul_(10.0)
    ev = StudentEvaluator(student, teacher, vocab, "cuda")
    loader = [synthetic_batch(4, 5000, 16, seed=300 + i) for i in range(2)]
    res = ev.compare_models_on_dataset(loader, num_samples=5, per_batch=3)
    assert res["total_samples"] == 5 and len(res["reference_captions"]) == 5
    for side in ("student", "teacher"):
        assert set(res[side]) == {"bleu1_scores", "bleu2_scores", "meteor_scores", "generated_captions", "inference_times",
                                  "successful_generations"}
        assert len(res[side]["generated_captions"]) == 5
        assert len(res[side]["bleu1_scores"]) == res[side]["successful_generations"]
        assert all(0.0 <= v <= 1.0 for v in res[side]["bleu1_scores"] + res[side]["meteor_scores"])
    imgs = loader[0][0]
    for j in range(3):
        single = " ".join(student.caption_image(imgs[j], vocab, max_length=25)).strip()
        batched = res["student"]["generated_captions"][j]
        assert batched in (single, "")          # "" only if the reference's "> 2 words" filter dropped it
        if len(single.split()) > 2:
            assert batched == single


def test_beam_self_attn_kernel_follows_the_ancestry_table():
    """csrc/beam.hip beam_self_attn: the new token's query attends to positions 0..t, position p < t read from row
    anc[p][row]; the kernel also stores this step's key / value at position t.  Checked against torch on gathered caches."""
    from imagecaptioner_amd import ops
    g = torch.Generator().manual_seed(5)
    rows, H, E, Tcap, t = 10, 8, 512, 13, 6
    qkv = torch.randn(rows, 3 * E, generator=g)
    kc, vc = torch.randn(Tcap, rows, E, generator=g), torch.randn(Tcap, rows, E, generator=g)
    anc = torch.randint(0, rows, (Tcap, rows), generator=g, dtype=torch.int32)
    kd, vd = kc.cuda(), vc.cuda()
    out = ops.beam_self_attn(qkv.cuda(), kd, vd, anc.cuda(), H, t).cpu()
    assert torch.equal(kd[t].cpu(), qkv[:, E:2 * E]) and torch.equal(vd[t].cpu(), qkv[:, 2 * E:])
    assert torch.equal(kd[:t].cpu(), kc[:t])                    # older positions untouched
    want = torch.empty(rows, E)
    for r in range(rows):
        K = torch.stack([kc[p, anc[p, r]] for p in range(t)] + [qkv[r, E:2 * E]]).view(t + 1, H, 64)
        Vv = torch.stack([vc[p, anc[p, r]] for p in range(t)] + [qkv[r, 2 * E:]]).view(t + 1, H, 64)
        q = qkv[r, :E].view(H, 64)
        w = torch.softmax(torch.einsum("hd,thd->ht", q.double(), K.double()) / 8.0, -1)
        want[r] = torch.einsum("ht,thd->hd", w, Vv.double()).reshape(E).float()
    assert (out - want).abs().max().item() < 1e-5


def _beam_step_reference(logits, score, width, seq, t, end_id, V):
    """the reference's expansion for one image (teacher_model.py:170-228) on host tensors"""
    w = int(width)
    cand = (score[:w, None] + torch.log_softmax(logits[:w], -1)).view(-1)
    vals, flat = torch.topk(cand, w)
    live, fin = [], []
    for v, f in zip(vals.tolist(), flat.tolist()):
        s = seq[f // V][:t + 1] + [f % V]
        (fin if (end_id is not None and f % V == end_id) else live).append((s, v, f // V))
    return live, fin


def test_beam_step_kernel_vs_reference_expansion():
    from imagecaptioner_amd import ops
    g = torch.Generator().manual_seed(11)
    B, W, V, Tcap, t, end_id = 4, 5, 5000, 9, 3, 2
    logits = torch.randn(B * W, V, generator=g) * 2
    logits[:, end_id] += 6.0                                     # some candidates finish
    score = -torch.rand(B, W, generator=g) * 3
    width = torch.tensor([5, 3, 0, 1], dtype=torch.int32)
    seq = torch.randint(4, V, (B * W, Tcap), generator=g, dtype=torch.int32)
    anc = torch.randint(0, B * W, (Tcap, B * W), generator=g, dtype=torch.int32)
    d = lambda x: x.clone().cuda()
    sc, wd, seq_in, anc_in = d(score), d(width), d(seq), d(anc)
    seq_out, anc_out = torch.zeros_like(seq_in), torch.zeros_like(anc_in)
    tok = torch.full((B * W,), -1, dtype=torch.int64, device="cuda")
    fin_seq = torch.zeros(B, W, Tcap, dtype=torch.int32, device="cuda"); fin_score = torch.zeros(B, W, device="cuda")
    fin_len = torch.zeros(B, W, dtype=torch.int32, device="cuda"); nfin = torch.tensor([0, 1, 0, 0], dtype=torch.int32, device="cuda")
    ops.beam_step(logits.cuda(), sc, wd, seq_in, seq_out, anc_in, anc_out, tok, fin_seq, fin_score, fin_len, nfin, t, end_id)
    torch.cuda.synchronize()
    for b in range(B):
        rows = slice(b * W, (b + 1) * W)
        if int(width[b]) == 0:
            assert int(wd[b]) == 0 and tok[rows].tolist() == [0] * W
            continue
        live, fin = _beam_step_reference(logits[rows], score[b], width[b], seq[rows].tolist(), t, end_id, V)
        assert int(wd[b]) == len(live)
        n0 = 1 if b == 1 else 0
        assert int(nfin[b]) == n0 + len(fin)
        for j, (s, v, origin) in enumerate(live):
            assert seq_out[b * W + j, :t + 2].tolist() == s
            assert abs(float(sc[b, j]) - v) < 1e-5 and int(tok[b * W + j]) == s[-1]
            assert anc_out[:t, b * W + j].tolist() == anc[:t, b * W + origin].tolist() and int(anc_out[t, b * W + j]) == b * W + origin
        for j, (s, v, _) in enumerate(fin):
            assert fin_seq[b, n0 + j, :t + 2].tolist() == s and int(fin_len[b, n0 + j]) == t + 2
            assert abs(float(fin_score[b, n0 + j]) - v) < 1e-5
        assert tok[b * W + len(live):(b + 1) * W].tolist() == [0] * (W - len(live))


@pytest.mark.parametrize("bias", [0.0, 10.5])
def test_batched_cached_beam_search_equals_the_prefix_rerun_search(teacher, bias):
    """caption_images (KV cache + ancestry table, B images x W beams per launch sequence, one device->host copy) against
    caption_image_recompute (the reference's own procedure: the decoder re-run on the growing prefixes, one image at a
    time): identical captions for every image, with and without early finishes."""
    from imagecaptioner_amd.utils.seeded_init import synthetic_batch
    vocab = Vocab(5000)
    images, _ = synthetic_batch(6, 5000, 16, seed=777)
    images = images.cuda()
    with torch.no_grad():
        teacher.fc_out.bias[2] += bias
    try:
        for beam, lp, nret in ((5, 0.6, 5), (3, 0.0, 2)):
            got = teacher.caption_images(images, vocab, max_length=14, beam_size=beam, length_penalty=lp, num_return_sequences=nret)
            for b in range(images.shape[0]):
                want = teacher.caption_image_recompute(images[b], vocab, max_length=14, beam_size=beam, length_penalty=lp,
                                                       num_return_sequences=nret)
                assert got[b] == want, (b, beam, got[b], want)
    finally:
        with torch.no_grad():
            teacher.fc_out.bias[2] -= bias
