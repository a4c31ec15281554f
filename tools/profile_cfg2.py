#!/usr/bin/env python3
"""cfg2 (student forward + 20-token batched greedy decode, bf16, batch 128) a few times for rocprofv3 --kernel-trace --stats."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imagecaptioner_amd import ops
from imagecaptioner_amd.student_model import CaptioningStudent
from imagecaptioner_amd.utils.seeded_init import apply_seeded_init, synthetic_batch
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
m = apply_seeded_init(CaptioningStudent(5000, 256, 512, 2), 0).cuda().eval()
images, _ = synthetic_batch(128, 5000, 16, seed=4321)
images = images.cuda()
with ops.precision(prec):
    for _ in range(5):
        m.generate(images, 20)
torch.cuda.synchronize()
