#!/usr/bin/env python3
"""Where the device-to-device memcpys of one eager KD step come from (torch profiler, grouped by Python stack)."""
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
tr = KDTrainer(s, t, p, vocab_size=5000, batch_size=64, use_graph=False, precision=prec)
images, caps = synthetic_batch(64, 5000, 16)
tr.train_step(images.cuda(), caps.cuda())
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    tr.train_step()
    torch.cuda.synchronize()
cnt = collections.Counter()
for e in prof.events():
    if e.name in ("aten::copy_", "aten::clone", "aten::contiguous", "aten::cat", "aten::stack", "aten::zero_", "aten::fill_", "aten::zeros", "aten::add_", "aten::mul_"):
        st = [f for f in (e.stack or []) if "imagecaptioner_amd" in f or "oracle" in f]
        cnt[(e.name, st[0] if st else "?")] += 1
for (name, where), n in cnt.most_common(40):
    print(f"{n:4d} {name:18s} {where}")
