#!/usr/bin/env python3
"""Which lines of the package launch ATen copies / fills / adds inside one KD step (GPU box).  torch.profiler gives no Python
stacks on this build, so the tensor methods that launch such kernels are wrapped at the Python level for ONE eager step and
every call from inside imagecaptioner_amd/ is attributed to its innermost package frame.  (autograd's own gradient
accumulation adds are invisible here: they are the `CUDAFunctor_add<float>` launches of the kernel statistics.)
VERDICT r02 item 9: the step should launch nothing but libick.so kernels."""
import collections
import os
import sys
import traceback

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imagecaptioner_amd.train_student_kd import KDTrainer, build_kd_models  # noqa: E402
from imagecaptioner_amd.utils.seeded_init import synthetic_batch  # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else "f32"
B = 64
s, t, p = build_kd_models(device="cuda")
tr = KDTrainer(s, t, p, vocab_size=5000, batch_size=B, use_graph=False, precision=prec)
images, caps = synthetic_batch(B, 5000, 16, seed=1)
tr.train_step(images.cuda(), caps.cuda())
tr.train_step()
torch.cuda.synchronize()

agg = collections.Counter()
ACTIVE = [False]


def where():
    fr = [f for f in traceback.extract_stack()[:-2] if "imagecaptioner_amd" in f.filename]
    if not fr:
        return None
    f = fr[-1]
    return f"{os.path.basename(f.filename)}:{f.lineno} {f.line.strip()[:90]}"


def wrap(owner, name, label):
    orig = getattr(owner, name)

    def inner(*a, **k):
        if ACTIVE[0]:
            ACTIVE[0] = False
            try:
                w = where()
                big = next((x for x in a if torch.is_tensor(x) and x.is_cuda), None)
                if w and (big is not None or label in ("zeros", "cat", "stack", "empty_like->fill")):
                    n = big.numel() if big is not None else 0
                    agg[(label, w)] += 1
                    agg[("bytes", label, w)] += n * (big.element_size() if big is not None else 0)
            finally:
                ACTIVE[0] = True
        return orig(*a, **k)
    setattr(owner, name, inner)


for nm, lab in (("copy_", "copy_"), ("clone", "clone"), ("contiguous", "contiguous"), ("zero_", "zero_"), ("fill_", "fill_"),
                ("add_", "add_"), ("mul_", "mul_"), ("__add__", "add"), ("__mul__", "mul"), ("to", "to"), ("float", "float")):
    wrap(torch.Tensor, nm, lab)
for nm in ("zeros", "cat", "stack", "zeros_like", "full"):
    wrap(torch, nm, nm)

ACTIVE[0] = True
tr.train_step()
torch.cuda.synchronize()
ACTIVE[0] = False
rows = [(k, v) for k, v in agg.items() if k[0] != "bytes"]
for (label, w), n in sorted(rows, key=lambda kv: -kv[1]):
    mb = agg[("bytes", label, w)] / 1e6
    print(f"{n:4d}  {label:11s} {mb:9.2f} MB  {w}")
