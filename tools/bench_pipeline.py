#!/usr/bin/env python3
"""Throughput of the on-GPU input transform (imagecaptioner_amd/data_pipeline.py) on Flickr8k-shaped images
(500x375 / 375x500 uint8), batch 64, images already resident on the device; algorithmic bytes per image =
562.5 KB read + 602 KB fp32 written."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imagecaptioner_amd.data_pipeline import GpuImageTransform

rng = np.random.default_rng(0)
imgs = [torch.from_numpy(rng.integers(0, 256, ((375, 500) if i % 3 else (500, 375)) + (3,), dtype=np.uint8)).cuda() for i in range(64)]
for train in (False, True):
    tf = GpuImageTransform(train=train, generator=torch.Generator().manual_seed(0))
    tf(imgs); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        tf(imgs)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    print(f"{'train' if train else 'val'} transform: {dt * 1e3:.3f} ms / batch of 64 = {64 / dt:,.0f} images/s "
          f"({64 * (562.5e3 + 602.1e3) / dt / 1e9:.1f} GB/s algorithmic; includes host-side packing and parameter draws)")
