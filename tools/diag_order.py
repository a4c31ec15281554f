#!/usr/bin/env python3
"""Order dependence of tests/test_kd_step_b16_gpu.py inside one process (GPU box): python tools/diag_order.py f32x3 f32x3 | f32 f32x3 ..."""
import gc
import os
import sys

import torch

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
sys.path.insert(0, os.path.join(root, "tests"))
import test_kd_step_b16_gpu as T  # noqa: E402

for i, prec in enumerate(sys.argv[1:]):
    if prec == "gc":
        gc.collect(); torch.cuda.empty_cache(); print("-- gc + empty_cache"); continue
    print(f"==== run {i}: {prec}", flush=True)
    try:
        T.test_kd_step_b16_gradients_vs_fp64_yardstick(prec)
        print("PASS")
    except AssertionError as e:
        print("FAIL", str(e).splitlines()[0][:100])
