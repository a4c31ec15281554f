#!/usr/bin/env python3
"""Which lines of the package launch ATen / runtime kernels (copies, fills, adds) inside one KD step (GPU box): one eager step
under torch.profiler with Python stacks; prints, per ATen op that launches a kernel, the innermost frames inside
imagecaptioner_amd/ with their call counts.  VERDICT r02 item 9: the step should launch nothing but libick.so kernels."""
import collections
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imagecaptioner_amd.train_student_kd import KDTrainer, build_kd_models  # noqa: E402
from imagecaptioner_amd.utils.seeded_init import synthetic_batch  # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else "f32"
s, t, p = build_kd_models(device="cuda")
tr = KDTrainer(s, t, p, vocab_size=5000, batch_size=8, use_graph=False, precision=prec)
images, caps = synthetic_batch(8, 5000, 16, seed=1)
tr.train_step(images.cuda(), caps.cuda())
tr.train_step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    tr.train_step()
    torch.cuda.synchronize()
WATCH = ("aten::copy_", "aten::fill_", "aten::zero_", "aten::add", "aten::add_", "aten::mul", "aten::mul_", "aten::cat", "aten::clone",
         "aten::contiguous", "aten::_to_copy", "aten::sum", "aten::stack", "aten::index_select", "aten::zeros", "aten::div", "aten::sub")
agg = collections.Counter()
for ev in prof.key_averages(group_by_stack_n=12):
    if ev.key in WATCH:
        frames = [f for f in (ev.stack or []) if "imagecaptioner_amd" in f]
        where = " <- ".join(fr.split("imagecaptioner_amd/")[-1].split(" ")[0] for fr in frames[:2]) if frames else ((ev.stack or ["?"])[0])
        agg[(ev.key, where)] += ev.count
for (name, where), n in sorted(agg.items(), key=lambda kv: -kv[1]):
    print(f"{n:4d}  {name:18s} {where}")
