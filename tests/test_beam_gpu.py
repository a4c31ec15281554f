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


def test_beam_search_captions_vs_reference(teacher):
    from imagecaptioner_amd.utils.seeded_init import synthetic_batch
    g = load_golden("teacher_beam.npz")
    vocab = Vocab(5000)
    images, _ = synthetic_batch(3, 5000, 16, seed=int(g["batch_seed"]))
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
