#!/usr/bin/env python3
"""The student decoder alone (teacher-forced T = 15 forward + BPTT, cfg3 sizes, B = 64) as one captured hipGraph:
ms per forward+backward, for `rocprofv3 --kernel-trace --stats -- python3 tools/profile_decoder.py`."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imagecaptioner_amd.student_model import LSTMDecoder  # noqa: E402
from imagecaptioner_amd.utils.seeded_init import apply_seeded_init  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
E, H, NL, V, T = 256, 512, 2, 5000, 15
dec = apply_seeded_init(LSTMDecoder(V, E, H, NL, 0.3), 0).cuda().train()
feats = torch.randn(B, 49, E, device="cuda", requires_grad=True)
caps = torch.randint(4, V, (T, B), device="cuda")
dl = torch.randn(T, B, V, device="cuda") * 1e-3


def step():
    logits, hs, _ = dec(feats, caps)
    logits.backward(dl)


s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    step(); step()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    step()
g.replay(); torch.cuda.synchronize()
t0 = time.perf_counter()
N = 20
for _ in range(N):
    g.replay()
torch.cuda.synchronize()
print(f"decoder fwd+bwd (B={B}, T={T}, {E}/{H}/{NL}): {(time.perf_counter() - t0) / N * 1e3:.3f} ms per replay")
