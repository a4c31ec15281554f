"""On-GPU input pipeline — SURVEY.md §8(f) row N3.

Replaces, for the image side, what the reference does per sample on the CPU inside DataLoader workers
(/root/reference/src/train_student_kd.py:122-135, src/data_loader.py:71-103):

    train:  Resize((224,224)) -> ColorJitter(.1,.1,.1,.05) -> RandomHorizontalFlip(.3) -> ToTensor -> Normalize
    val:    Resize((224,224)) -> ToTensor -> Normalize

torchvision applies these to PIL images, so the arithmetic is Pillow's 8-bit arithmetic; csrc/preprocess.hip reproduces
it bit for bit (Pillow 12.2 is the pin: tests/test_data_pipeline_*.py compare against PIL itself).  The host side here
  * builds Pillow's resampling coefficient tables per distinct source size (precompute_coeffs for the bilinear filter
    with support scaling + normalize_coeffs_8bpc), cached on the device,
  * draws the random parameters IN TORCHVISION'S ORDER from a torch generator — per image: randperm(4) (operation
    order), uniform brightness, contrast, saturation, hue factors, rand(1) < p for the flip — so a run seeded like the
    reference's sees the same augmentations,
  * packs the decoded uint8 HWC images of a batch into one buffer and launches one horizontal-pass kernel per distinct
    width and one "vertical pass + jitter + flip + normalize" kernel per distinct height.
Decoding JPEG stays on the host (no decoder in this image); captions: `collate_captions` is the (T,B) PAD-filled
pad_sequence of the reference's Collate (:91-103); `Vocabulary` mirrors data_loader.py:20-48 with a pluggable tokenizer
(the reference's spaCy tokenizer is not installable offline: parity of tokenisation is unpinned, numericalisation given
tokens is exact).
"""
from __future__ import annotations

import ctypes
import math
import re
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib
from ._lib import check

OUT = 224
PRECISION_BITS = 32 - 8 - 2
IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


def pillow_bilinear_coeffs(in_size: int, out_size: int = OUT) -> Tuple[np.ndarray, np.ndarray]:
    """Pillow ImagingResample: precompute_coeffs (triangle filter, support = max(scale, 1)) and
    normalize_coeffs_8bpc.  Returns bounds (out,2) int32 = (first source index, tap count) and integer weights
    (out, ksize) int32 with 22 fractional bits."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.float64)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        x = np.arange(xmax)
        w = 1.0 - np.abs((x + xmin - center + 0.5) * ss)
        w = np.where(w > 0.0, w, 0.0)
        tot = w.sum()
        kk[xx, :xmax] = w / tot if tot != 0.0 else w
        bounds[xx] = (xmin, xmax)
    ik = np.where(kk < 0, -0.5 + kk * (1 << PRECISION_BITS), 0.5 + kk * (1 << PRECISION_BITS)).astype(np.int64).astype(np.int32)
    return bounds, ik


def draw_train_params(n: int, generator: Optional[torch.Generator] = None, brightness=0.1, contrast=0.1, saturation=0.1,
                      hue=0.05, p_flip=0.3) -> List[dict]:
    """The random draws of ColorJitter.get_params followed by RandomHorizontalFlip.forward, per image, in torchvision's
    order (torch CPU generator; `None` = the global one, as torchvision uses)."""
    out = []
    for _ in range(n):
        order = torch.randperm(4, generator=generator).tolist()
        b = float(torch.empty(1).uniform_(max(0.0, 1 - brightness), 1 + brightness, generator=generator))
        c = float(torch.empty(1).uniform_(max(0.0, 1 - contrast), 1 + contrast, generator=generator))
        s = float(torch.empty(1).uniform_(max(0.0, 1 - saturation), 1 + saturation, generator=generator))
        h = float(torch.empty(1).uniform_(-hue, hue, generator=generator))
        flip = bool(torch.rand(1, generator=generator) < p_flip)
        out.append({"order": order, "brightness": b, "contrast": c, "saturation": s, "hue": h, "flip": flip})
    return out


def hue_shift_u8(hue_factor: float) -> int:
    """torchvision F_pil.adjust_hue: `np_h += np.array(hue_factor * 255).astype(np.uint8)` — a C cast of a (possibly
    negative) float to uint8, i.e. truncation toward zero, then modulo 256."""
    return int(hue_factor * 255) & 255


class GpuImageTransform:
    """callable: list of uint8 HWC RGB images (numpy arrays or torch tensors, any sizes) -> (B,3,224,224) fp32 on device."""

    def __init__(self, train: bool, device="cuda", generator: Optional[torch.Generator] = None,
                 mean: Sequence[float] = IMAGENET_MEAN, std: Sequence[float] = IMAGENET_STD):
        self.train, self.device, self.generator = train, torch.device(device), generator
        # torch.as_tensor(mean, dtype=float32): the fp32 roundings of the constants
        self._mean = (ctypes.c_float * 3)(*[float(np.float32(m)) for m in mean])
        self._std = (ctypes.c_float * 3)(*[float(np.float32(s)) for s in std])
        self._tables: Dict[int, Tuple[torch.Tensor, torch.Tensor, int]] = {}

    def _table(self, in_size: int):
        if in_size not in self._tables:
            b, k = pillow_bilinear_coeffs(in_size)
            self._tables[in_size] = (torch.from_numpy(b).to(self.device), torch.from_numpy(np.ascontiguousarray(k)).to(self.device),
                                     k.shape[1])
        return self._tables[in_size]

    def __call__(self, images: Sequence, params: Optional[List[dict]] = None) -> torch.Tensor:
        if not torch.cuda.is_available():
            raise RuntimeError("GpuImageTransform needs an MI355X (the transform has no CPU fallback; use PIL on the host)")
        B = len(images)
        imgs = [torch.as_tensor(np.ascontiguousarray(im) if isinstance(im, np.ndarray) else im) for im in images]
        for im in imgs:
            if im.dtype != torch.uint8 or im.dim() != 3 or im.shape[2] != 3:
                raise ValueError(f"expected uint8 HWC RGB images, got {tuple(im.shape)} {im.dtype}")
        # ---- pack
        sizes = [(int(im.shape[0]), int(im.shape[1])) for im in imgs]
        src_off, tmp_off, so, to = [], [], 0, 0
        for h, w in sizes:
            src_off.append(so); tmp_off.append(to)
            so += (h * w * 3 + 15) // 16 * 16
            to += (h * OUT * 3 + 15) // 16 * 16 if w != OUT else 0
        src = torch.empty(max(so, 16), dtype=torch.uint8, device=self.device)
        for im, o, (h, w) in zip(imgs, src_off, sizes):
            src[o:o + h * w * 3].copy_(im.reshape(-1), non_blocking=True)
        tmp = torch.empty(max(to, 16), dtype=torch.uint8, device=self.device)
        out = torch.empty(B, 3, OUT, OUT, dtype=torch.float32, device=self.device)
        item = lambda i: (src_off[i], tmp_off[i], sizes[i][0], sizes[i][1], i, 0)
        rec = np.dtype([("src", "<i8"), ("tmp", "<i8"), ("H", "<i4"), ("W", "<i4"), ("dst", "<i4"), ("pad", "<i4")])
        # ---- random parameters (train) in torchvision's draw order
        jit_dev = None
        if self.train:
            params = params if params is not None else draw_train_params(B, self.generator)
            jrec = np.dtype([("order", "<i4", 4), ("b", "<f4"), ("c", "<f4"), ("s", "<f4"), ("hue", "<i4"), ("flip", "<i4"),
                             ("pad", "<i4", 2)])
            j = np.zeros(B, jrec)
            for i, p in enumerate(params):
                j[i] = (p["order"], p["brightness"], p["contrast"], p["saturation"], hue_shift_u8(p["hue"]), int(p["flip"]), (0, 0))
            jit_dev = torch.from_numpy(j.view(np.uint8).reshape(B, -1).copy()).to(self.device)
        L = _lib.lib()
        st = torch.cuda.current_stream().cuda_stream
        keep = []
        # ---- horizontal pass, one launch per distinct source width
        for w in sorted({w for _, w in sizes if w != OUT}):
            idx = [i for i, (_, ww) in enumerate(sizes) if ww == w]
            items = torch.from_numpy(np.array([item(i) for i in idx], rec).view(np.uint8).reshape(len(idx), -1).copy()).to(self.device)
            bnd, coef, ks = self._table(w)
            keep.append(items)
            check(L.ick_resize_h_u8(src.data_ptr(), tmp.data_ptr(), items.data_ptr(), len(idx), max(sizes[i][0] for i in idx),
                                    bnd.data_ptr(), coef.data_ptr(), ks, st), "ick_resize_h_u8")
        # ---- vertical pass + jitter + flip + normalize, one launch per distinct source height
        for h in sorted({h for h, _ in sizes}):
            idx = [i for i, (hh, _) in enumerate(sizes) if hh == h]
            items = torch.from_numpy(np.array([item(i) for i in idx], rec).view(np.uint8).reshape(len(idx), -1).copy()).to(self.device)
            keep.append(items)
            bnd, coef, ks = self._table(h) if h != OUT else (None, None, 0)
            check(L.ick_resize_v_jitter_normalize(src.data_ptr(), tmp.data_ptr(), items.data_ptr(), len(idx),
                                                  bnd.data_ptr() if bnd is not None else None,
                                                  coef.data_ptr() if coef is not None else None, ks,
                                                  jit_dev.data_ptr() if jit_dev is not None else None, out.data_ptr(),
                                                  self._mean, self._std, st), "ick_resize_v_jitter_normalize")
        for t in keep + [src, tmp] + ([jit_dev] if jit_dev is not None else []):
            t.record_stream(torch.cuda.current_stream())
        return out


# ----------------------------------------------------------------------------------------------------- captions
class Vocabulary:
    """reference: data_loader.py:20-48 (ids 0..3 = <PAD>, <START>, <END>, <UNK>; words enter at freq_threshold)."""

    def __init__(self, freq_threshold: int, tokenizer: Optional[Callable[[str], List[str]]] = None):
        self.itos = {0: "<PAD>", 1: "<START>", 2: "<END>", 3: "<UNK>"}
        self.stoi = {"<PAD>": 0, "<START>": 1, "<END>": 2, "<UNK>": 3}
        self.freq_threshold = freq_threshold
        self._tok = tokenizer or self.tokenizer_eng

    def __len__(self):
        return len(self.itos)

    @staticmethod
    def tokenizer_eng(text) -> List[str]:
        """stand-in for the reference's spaCy tokenizer (not installable offline): lower-cased words and punctuation."""
        return re.findall(r"[a-z0-9]+(?:'[a-z]+)?|[^\sa-z0-9]", str(text).lower())

    def build_vocabulary(self, sentence_list):
        frequencies, idx = {}, 4
        for sentence in sentence_list:
            for word in self._tok(sentence):
                frequencies[word] = frequencies.get(word, 0) + 1
                if frequencies[word] == self.freq_threshold:
                    self.stoi[word] = idx
                    self.itos[idx] = word
                    idx += 1

    def numericalize(self, text) -> List[int]:
        return [self.stoi.get(t, self.stoi["<UNK>"]) for t in self._tok(text)]

    def caption_ids(self, text) -> List[int]:
        """<START> + numericalized caption + <END> (data_loader.py:84-86)."""
        return [self.stoi["<START>"]] + self.numericalize(text) + [self.stoi["<END>"]]


def collate_captions(captions: Sequence[Sequence[int]], pad_idx: int = 0, device="cuda") -> torch.Tensor:
    """Collate.__call__ of the reference (data_loader.py:96-102): pad_sequence(batch_first=False) -> (T,B) int64."""
    T = max(len(c) for c in captions)
    out = torch.full((T, len(captions)), pad_idx, dtype=torch.int64)
    for b, c in enumerate(captions):
        out[:len(c), b] = torch.as_tensor(list(c), dtype=torch.int64)
    return out.to(device, non_blocking=True)
