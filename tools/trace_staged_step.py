#!/usr/bin/env python3
"""The staged (bucketed) data-parallel step under an RCCL group of world 1, for rocprofv3 --kernel-trace:
   cd /tmp && export TMPDIR=/tmp
   rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace_staged -o st -- python3 tools/trace_staged_step.py
   python tools/trace_staged_step.py analyse gpurun_out/trace_staged > profiles/r03_staged_step_overlap.json
The analysis reports, for every RCCL kernel of the last steps, how much of its interval overlaps kernels of the compute stream
(the three bucket all-reduces are meant to run beside layer4's / layer3's backward)."""
import csv
import glob
import json
import os
import sys

if len(sys.argv) > 2 and sys.argv[1] == "analyse":
    rows = []
    for f in glob.glob(sys.argv[2] + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Stream_Id", r.get("Queue_Id", ""))))
    rows.sort()
    nccl = [r for r in rows if "nccl" in r[2].lower() or "rccl" in r[2].lower()]
    comp = [r for r in rows if not ("nccl" in r[2].lower() or "rccl" in r[2].lower())]
    out = []
    for s, e, name, q in nccl[-9:]:
        ov = sum(max(0, min(e, ce) - max(s, cs)) for cs, ce, _, _ in comp if ce > s and cs < e)
        beside = sorted({n.split("(")[0].replace("void ", "").replace("(anonymous namespace)::", "")[:48] for cs, ce, n, _ in comp if ce > s and cs < e})
        out.append({"kernel": name[:60], "duration_us": round((e - s) / 1e3, 1), "overlapped_by_compute_us": round(min(ov, e - s) / 1e3, 1),
                    "compute_kernels_beside_it": beside[:6]})
    print(json.dumps({"workload": "staged KD step (3 graphs + 3 bucket all-reduces on a communication stream), RCCL group of world 1, B=64, fp32",
                      "note": "world 1: the all-reduce moves no data over xGMI; what the trace shows is that the collective's kernel runs on "
                              "its own stream beside the next stage's backward kernels", "rccl_kernels_total": len(nccl), "last_nine": out}, indent=1))
    raise SystemExit(0)

import socket

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imagecaptioner_amd.train_student_kd import KDTrainer, build_kd_models  # noqa: E402
from imagecaptioner_amd.utils.seeded_init import synthetic_batch  # noqa: E402

with socket.socket() as sk:
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
torch.cuda.set_device(0)
dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda", 0))
s, t, p = build_kd_models(device="cuda")
tr = KDTrainer(s, t, p, vocab_size=5000, batch_size=64, use_graph=True, bucketed=True)
images, caps = synthetic_batch(64, 5000, 16, seed=1)
tr.train_step(images.cuda(), caps.cuda())
for _ in range(5):
    tr.train_step()
torch.cuda.synchronize()
print("staged steps done; loss", tr.loss_dict()["total_loss"])
dist.destroy_process_group()
