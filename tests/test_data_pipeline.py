"""SURVEY.md §8(f) row N3 — the on-GPU input pipeline (imagecaptioner_amd/data_pipeline.py, csrc/preprocess.hip).
The oracle is Pillow itself (oracle/preprocess_ref.py): bit-exact equality, not a tolerance."""
import numpy as np
import pytest
import torch

SIZES = [(375, 500), (500, 375), (224, 224), (224, 300), (300, 224), (100, 80), (640, 480), (33, 1000)]


def _images(seed=0, sizes=SIZES):
    rng = np.random.default_rng(seed)
    out = []
    for i, (h, w) in enumerate(sizes):
        a = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        if i % 3 == 1:                      # smooth content as well as noise
            yy, xx = np.mgrid[0:h, 0:w]
            a = np.stack([(yy * 255 // max(h - 1, 1)), (xx * 255 // max(w - 1, 1)), ((yy + xx) % 256)], -1).astype(np.uint8)
        out.append(a)
    return out


def test_pillow_coefficient_tables_reproduce_pil_resize():
    """host logic: integer two-pass resampling with pillow_bilinear_coeffs == PIL.Image.resize(BILINEAR), bit for bit"""
    from PIL import Image
    from imagecaptioner_amd.data_pipeline import PRECISION_BITS, pillow_bilinear_coeffs

    def axis(img, out):        # resample along axis 1
        b, k = pillow_bilinear_coeffs(img.shape[1], out)
        res = np.zeros((img.shape[0], out, 3), np.uint8)
        for xx in range(out):
            x0, n = b[xx]
            acc = (1 << (PRECISION_BITS - 1)) + (img[:, x0:x0 + n].astype(np.int64) * k[xx, :n][None, :, None]).sum(1)
            res[:, xx] = np.clip(acc >> PRECISION_BITS, 0, 255)
        return res

    for a in _images(3, [(375, 500), (100, 80), (224, 300), (17, 23)]):
        t = a if a.shape[1] == 224 else axis(a, 224)
        t = t if a.shape[0] == 224 else axis(t.transpose(1, 0, 2), 224).transpose(1, 0, 2)
        assert np.array_equal(t, np.asarray(Image.fromarray(a).resize((224, 224), Image.BILINEAR)))


def test_draw_order_and_caption_side():
    from imagecaptioner_amd.data_pipeline import Vocabulary, collate_captions, draw_train_params, hue_shift_u8
    g = torch.Generator().manual_seed(5)
    p = draw_train_params(3, g)
    g2 = torch.Generator().manual_seed(5)     # torchvision's order: randperm(4), 4 uniforms, rand(1) — per image
    for q in p:
        assert q["order"] == torch.randperm(4, generator=g2).tolist()
        for key, lo, hi in (("brightness", .9, 1.1), ("contrast", .9, 1.1), ("saturation", .9, 1.1), ("hue", -.05, .05)):
            assert q[key] == float(torch.empty(1).uniform_(lo, hi, generator=g2))
        assert q["flip"] == bool(torch.rand(1, generator=g2) < 0.3)
    assert hue_shift_u8(0.05) == 12 and hue_shift_u8(-0.05) == 244 and hue_shift_u8(0.0) == 0
    v = Vocabulary(freq_threshold=2)
    v.build_vocabulary(["A dog runs.", "a dog sits", "the cat"])
    assert v.stoi["a"] == 4 and v.stoi["dog"] == 5 and len(v) == 6
    assert v.caption_ids("A dog flies") == [1, 4, 5, 3, 2]
    c = collate_captions([[1, 4, 2], [1, 4, 5, 3, 2]], pad_idx=0, device="cpu")
    assert c.shape == (5, 2) and c[:, 0].tolist() == [1, 4, 2, 0, 0] and c.dtype == torch.int64


@pytest.mark.gpu
@pytest.mark.parametrize("train", [False, True])
def test_gpu_transform_bit_exact_vs_pillow(train):
    from imagecaptioner_amd.data_pipeline import GpuImageTransform, draw_train_params
    from oracle.preprocess_ref import pil_transform
    imgs = _images(11)
    params = draw_train_params(len(imgs), torch.Generator().manual_seed(17)) if train else None
    if train:                                  # make sure every operation order / flip value occurs
        params[0]["order"], params[1]["order"], params[2]["order"] = [3, 2, 1, 0], [1, 0, 3, 2], [2, 3, 0, 1]
        params[0]["flip"], params[1]["flip"] = True, False
        params[3]["hue"], params[4]["hue"] = -0.05, 0.05
    tf = GpuImageTransform(train=train)
    got = tf(imgs, params).cpu()
    assert got.shape == (len(imgs), 3, 224, 224)
    for i, a in enumerate(imgs):
        want = pil_transform(a, params[i] if train else None)
        assert torch.equal(got[i], want), (i, a.shape, float((got[i] - want).abs().max()))
    again = tf(imgs, params).cpu()             # cached coefficient tables, same result
    assert torch.equal(got, again)
