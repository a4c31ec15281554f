#!/usr/bin/env python3
"""BASELINE configs[1] alone (bench.py's cfg2 leg): student eval forward + 20-token batched greedy decode, bf16, B=128."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402

print(json.dumps(bench.run_cfg2(torch.device("cuda:0"), print)))
