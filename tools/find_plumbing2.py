#!/usr/bin/env python3
"""Every ATen op that reaches the dispatcher during ONE eager KD step (GPU box), including the ones autograd's engine issues
from C++ (AccumulateGrad, gradient summation, view materialisation): a TorchDispatchMode sees them all, where the Python-level
wrappers of tools/find_plumbing.py cannot.  Printed: op, shapes, count, bytes, and the innermost package frame (forward ops) or
the autograd Function whose backward was running.  VERDICT r02 item 9."""
import collections
import os
import sys
import traceback

import torch
from torch.utils._python_dispatch import TorchDispatchMode

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
agg, nbytes = collections.Counter(), collections.Counter()
LAUNCHING = ("copy_", "clone", "fill_", "zero_", "add", "add_", "mul", "mul_", "cat", "stack", "_to_copy", "contiguous", "sum", "index_select",
             "zeros", "zeros_like", "ones_like", "div", "sub", "neg", "masked_fill", "where", "select_backward", "slice_backward", "index_put_")


class Log(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.__name__.split(".")[0]
        if name in LAUNCHING:
            tens = [a for a in args if torch.is_tensor(a) and a.is_cuda]
            if tens or name in ("zeros",):
                fr = [f for f in traceback.extract_stack()[:-1] if "imagecaptioner_amd" in f.filename]
                w = f"{os.path.basename(fr[-1].filename)}:{fr[-1].lineno} {fr[-1].line.strip()[:80]}" if fr else "<autograd engine>"
                shp = tuple(tens[0].shape) if tens else ()
                agg[(name, shp, w)] += 1
                nbytes[(name, shp, w)] += tens[0].numel() * tens[0].element_size() if tens else 0
        return func(*args, **(kwargs or {}))


with Log():
    tr.train_step()
torch.cuda.synchronize()
print(f"ATen ops that launch kernels / copies in one eager KD step (precision {prec}, B = {B}): {sum(agg.values())} calls")
for k, n in sorted(agg.items(), key=lambda kv: -kv[1])[:60]:
    print(f"{n:4d} x {k[0]:14s} {str(k[1]):28s} {nbytes[k] / 1e6:9.2f} MB  {k[2]}")
