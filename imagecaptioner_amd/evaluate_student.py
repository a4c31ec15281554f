"""Evaluation after training — the "next" row N1 of SURVEY.md §8(f): the reference's StudentEvaluator
(/root/reference/src/evaluate_student.py:21-201) on top of the device-side decoders of this package
(student: batched greedy `CaptioningStudent.generate`; teacher: `CaptioningTeacher.caption_images`, the KV-cached
beam search batched over images on the HIP path).  Same class / method names and the same result-dict layout; what differs is the execution: a whole batch is
decoded per launch sequence with ONE device->host copy of the token ids instead of one host sync per token per image.

The metrics are host-side string arithmetic exactly as the reference defines them (they are not BLEU / METEOR proper):
  bleu_score(pred, ref, n)      clipped n-gram precision of `pred` against ONE reference, no brevity penalty (:30-51)
  meteor_score_simple(pred, ref) F1 of the two WORD SETS (:53-69)
"""
from __future__ import annotations

import time
from collections import Counter
from typing import Dict, List, Sequence

import torch


class StudentEvaluator:
    def __init__(self, student_model, teacher_model, vocab, device):
        self.student_model, self.teacher_model, self.vocab, self.device = student_model, teacher_model, vocab, device
        self.student_model.eval()
        if self.teacher_model is not None:
            self.teacher_model.eval()

    # ------------------------------------------------------------------ metrics (reference :30-69)
    def bleu_score(self, predicted: str, reference: str, n: int = 1) -> float:
        pred_words, ref_words = predicted.lower().split(), reference.lower().split()
        if len(pred_words) < n or len(ref_words) < n:
            return 0.0
        pred_ngrams = [tuple(pred_words[i:i + n]) for i in range(len(pred_words) - n + 1)]
        ref_counter = Counter(tuple(ref_words[i:i + n]) for i in range(len(ref_words) - n + 1))
        overlap = sum(min(c, ref_counter[g]) for g, c in Counter(pred_ngrams).items())
        return overlap / len(pred_ngrams) if pred_ngrams else 0

    def meteor_score_simple(self, predicted: str, reference: str) -> float:
        pred_words, ref_words = set(predicted.lower().split()), set(reference.lower().split())
        if not ref_words:
            return 0.0
        overlap = len(pred_words & ref_words)
        recall = overlap / len(ref_words)
        precision = overlap / len(pred_words) if pred_words else 0
        return 0.0 if precision + recall == 0 else 2 * precision * recall / (precision + recall)

    # ------------------------------------------------------------------ decoding helpers
    def _words(self, ids: Sequence[int]) -> List[str]:
        out = []
        for i in ids:
            w = self.vocab.itos[int(i)]
            if w == "<END>":
                break
            out.append(w)
        return out

    def reference_caption(self, caption_column) -> str:
        """(T,) token ids of one reference caption -> text without <START>/<END>/<PAD> (reference :141-146)."""
        skip = {self.vocab.stoi["<START>"], self.vocab.stoi["<END>"], self.vocab.stoi["<PAD>"]}
        return " ".join(self.vocab.itos[int(t)] for t in caption_column if int(t) not in skip).strip()

    @torch.no_grad()
    def student_captions(self, images: torch.Tensor, max_length: int = 25) -> List[str]:
        """Greedy captions of a whole batch: one captured-able launch sequence, one device->host copy."""
        start = self.vocab.stoi.get("<START>", self.vocab.stoi["<UNK>"])
        ids, _ = self.student_model.generate(images.to(self.device), max_length, 1.0, start)
        ids = ids.cpu()
        return [" ".join(self._words(ids[:, b].tolist())).strip() for b in range(ids.shape[1])]

    @torch.no_grad()
    def teacher_caption(self, image: torch.Tensor, max_length: int = 25) -> str:
        toks = self.teacher_model.caption_image(image.to(self.device), self.vocab, max_length=max_length)
        if isinstance(toks, list) and toks and not isinstance(toks[0], str):
            toks = toks[0]
        return " ".join(toks).strip() if toks else ""

    @torch.no_grad()
    def teacher_captions(self, images: torch.Tensor, max_length: int = 25) -> List[str]:
        """Beam-search captions of a whole batch: the KV-cached batched search (CaptioningTeacher.caption_images), one
        device->host copy per batch instead of one decoder re-run per step per image."""
        if hasattr(self.teacher_model, "caption_images"):
            return [" ".join(c).strip() if c and not isinstance(c, str) else (c or "")
                    for c in self.teacher_model.caption_images(images.to(self.device), self.vocab, max_length=max_length)]
        return [self.teacher_caption(images[j], max_length) for j in range(images.shape[0])]

    def measure_inference_time(self, image: torch.Tensor, num_runs: int = 10):
        """(student seconds, teacher seconds) per caption, synchronised around the timed loops (reference :71-97)."""
        image = image.to(self.device)
        for _ in range(3):
            self.student_model.caption_image(image, self.vocab, max_length=20)
            if self.teacher_model is not None:
                self.teacher_model.caption_image(image, self.vocab, max_length=20)
        times = []
        for model in (self.student_model, self.teacher_model):
            if model is None:
                times.append(float("nan"))
                continue
            if torch.cuda.is_available():
                torch.cuda.synchronize()
            t0 = time.time()
            for _ in range(num_runs):
                model.caption_image(image, self.vocab, max_length=20)
            if torch.cuda.is_available():
                torch.cuda.synchronize()
            times.append((time.time() - t0) / num_runs)
        return times[0], times[1]

    def _score(self, bucket: Dict, caption: str, ref: str) -> None:
        if caption and len(caption.split()) > 2:         # the reference counts a generation only if it has > 2 words
            bucket["bleu1_scores"].append(self.bleu_score(caption, ref, 1))
            bucket["bleu2_scores"].append(self.bleu_score(caption, ref, 2))
            bucket["meteor_scores"].append(self.meteor_score_simple(caption, ref))
            bucket["generated_captions"].append(caption)
            bucket["successful_generations"] += 1
        else:
            bucket["generated_captions"].append("")

    def compare_models_on_dataset(self, data_loader, num_samples: int = 200, per_batch: int = 3, time_runs: int = 0):
        """Reference :99-201 (up to `per_batch` images of each batch, same result dict).  The student decodes the
        selected images of a batch together; per-image timing is optional (time_runs > 0) because it re-runs the
        decoders `time_runs` times per image like the reference does."""
        mk = lambda: {"bleu1_scores": [], "bleu2_scores": [], "meteor_scores": [], "generated_captions": [],
                      "inference_times": [], "successful_generations": 0}
        results = {"student": mk(), "teacher": mk(), "reference_captions": [], "total_samples": 0}
        for imgs, captions in data_loader:
            if results["total_samples"] >= num_samples:
                break
            k = min(imgs.size(0), per_batch, num_samples - results["total_samples"])
            s_caps = self.student_captions(imgs[:k], max_length=25)
            t_caps = self.teacher_captions(imgs[:k], max_length=25) if self.teacher_model is not None else None
            for j in range(k):
                ref = self.reference_caption(captions[:, j].cpu().tolist())
                results["reference_captions"].append(ref)
                if time_runs > 0:
                    st, tt = self.measure_inference_time(imgs[j], num_runs=time_runs)
                    results["student"]["inference_times"].append(st)
                    results["teacher"]["inference_times"].append(tt)
                self._score(results["student"], s_caps[j], ref)
                if self.teacher_model is not None:
                    self._score(results["teacher"], t_caps[j], ref)
                results["total_samples"] += 1
        return results
