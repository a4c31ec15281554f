#!/usr/bin/env python3
"""Per-shape timing of every igemm_f32 launch of one KD train step (GPU box).
Records the descriptors of one eager step at B=64, then replays each distinct descriptor in isolation
(L2-warm, 20 iterations) and prints time, count, TFLOP/s, sorted by total time in the step."""
import collections
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imagecaptioner_amd import _lib, ops  # noqa: E402
from imagecaptioner_amd.train_student_kd import KDTrainer, build_kd_models  # noqa: E402
from imagecaptioner_amd.utils.seeded_init import synthetic_batch  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
PREC = sys.argv[2] if len(sys.argv) > 2 else "f32"     # "bf16": sweep the bf16 family (student side of the AMP step)
H16 = None                                              # "h16" / "h16-bf16": the native 16-bit launches of the fp16 / bf16 step
if PREC.startswith("h16"):
    H16 = torch.bfloat16 if PREC.endswith("bf16") else torch.float16
    PREC = "bf16" if PREC.endswith("bf16") else "fp16"
OPN = ["NT", "NN", "TN", "CONV_FWD", "CONV_FWD_C4", "CONV_DGRAD", "CONV_WGRAD", "CONV_DGRAD_S2"]
records = []
orig = ops.gemm_raw


def rec(op, A, Bp, C, M, N, K, lda, ldb, ldc, **kw):
    if (ops.gemm_precision() == "f32") != (PREC == "f32") or (kw.get("h16") is not None) != (H16 is not None) or \
            (PREC == "f32x3" and not (kw.get("x3") and op in (0, 3))):
        return orig(op, A, Bp, C, M, N, K, lda, ldb, ldc, **kw)          # AMP: only the student's 16-bit launches are swept
    records.append((op, M, N, K, lda, ldb, ldc, kw.get("batch", (1, 1)), kw.get("strides", (0,) * 6), kw.get("splitk", 1),
                    kw.get("conv"), kw.get("act", 0), kw.get("bias") is not None, kw.get("residual") is not None,
                    kw.get("stat_sum") is not None, kw.get("accumulate", False), kw.get("ldr", 0), kw.get("io16", 0),
                    kw.get("stat_copies", 1), kw.get("stat_stride", 0)))
    return orig(op, A, Bp, C, M, N, K, lda, ldb, ldc, **kw)


WORKLOAD = sys.argv[3] if len(sys.argv) > 3 else "kd"    # "cfg2": student eval forward + 20-token greedy decode at batch B
if WORKLOAD == "cfg2":
    from imagecaptioner_amd.student_model import CaptioningStudent
    from imagecaptioner_amd.utils.seeded_init import apply_seeded_init
    m = apply_seeded_init(CaptioningStudent(5000, 256, 512, 2), 0).cuda().eval()
    images, _ = synthetic_batch(B, 5000, 16, seed=4321)
    images = images.cuda()
    if PREC != "f32":
        ops._TUNED_BF16.clear(); ops._TUNED_H16.clear()
    with ops.precision(PREC):
        m.generate(images, 20)
        ops.gemm_raw = rec
        m.generate(images, 20)
        ops.gemm_raw = orig
else:
    dims = dict(embed_size=384, hidden_size=768, num_layers=3) if WORKLOAD == "cfg5" else {}
    student, teacher, projectors = build_kd_models(device="cuda", **dims)
    tr = KDTrainer(student, teacher, projectors, vocab_size=5000, batch_size=B, use_graph=False, overlap_teacher=False,
                   precision=PREC, teacher_precision="f32x3" if PREC == "f32x3" else "f32")   # "f32x3": the forward launches of teacher and student
    if PREC != "f32":
        ops._TUNED_BF16.clear(); ops._TUNED_H16.clear(); ops._TUNED_X3.clear()
    images, caps = synthetic_batch(B, 5000, 16)
    tr.train_step(images.cuda(), caps.cuda())
    ops.gemm_raw = rec
    tr.train_step()
    ops.gemm_raw = orig
torch.cuda.synchronize()
print(f"{len(records)} igemm launches per step at B={B}")
cnt = collections.Counter(records)
big = torch.empty(1 << 28, device="cuda").normal_()         # 1 GiB scratch for operands
if H16 is not None:
    big = big.view(H16)                                     # (the bit patterns of fp32 normals: finite 16-bit values and zeros-ish, no NaN halves matter for timing)
    big.normal_()
out = torch.empty(1 << 27, device="cuda")
stat = torch.zeros(2, 8 * 4096, dtype=torch.float64, device="cuda")
rows = []
TN_ = ["model", "128x128", "64x64", "128x64", "64x128", "3b64x64", "3b128x64", "3b64x128", "r128x128", "r64x64", "r128x64", "r64x128",
       "s+64x64", "s+128x64", "s+64x128", "s+3b64x64", "8w128x128", "8w128x64", "8w3b128x64", "ld128x128"]
TILES = [0, 1, 2, 3, 4, 18, 19, 20, 257, 258, 259, 260, 34, 35, 36, 50, 65, 67, 83, 129]   # +16: three LDS buffers; +32: M-split (128x128 body + that tail tile); +256: the register-staged kernel; 129: 4 compute + 4 loader waves, three buffers
if PREC != "f32":
    ops.set_gemm_precision(PREC)


def timeit(f, iters=10):
    f(); f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


for r, n in cnt.items():
    op, M, N, K, lda, ldb, ldc, batch, strides, splitk, conv, act, bias, res, st, acc, ldr, io16, scopies, sstride = r
    kw = dict(batch=batch, strides=strides, splitk=splitk, conv=conv, act=act, accumulate=acc, ldr=ldr)
    if PREC == "f32x3":
        kw["x3"] = True
    if H16 is not None:
        kw.update(h16=H16, io16=io16)
    if bias:
        kw["bias"] = big.data_ptr()
    if res:
        kw["residual"] = big.data_ptr() + (1 << 29)
    if st:
        kw["stat_sum"], kw["stat_sq"] = stat[0].data_ptr(), stat[1].data_ptr()
        kw["stat_copies"], kw["stat_stride"] = scopies, sstride      # (as the step launches it: the atomics spread over the copies)
    a, b = big.data_ptr(), big.data_ptr() + (1 << 29)
    ts = []
    kc = op in (0, 3)                                        # NT / CONV_FWD: the LDS-DMA kernel exists for 16-bit operands
    for tile in TILES:
        if tile == 129 and (PREC not in ("f32", "f32x3") or op not in (0, 3)):
            ts.append(float("inf"))          # loader-wave variant: exact-fp32 NT / CONV_FWD only
            continue
        if PREC == "f32x3" and tile > 255:
            ts.append(float("inf"))          # three-product kernel: the LDS-DMA forward tiles only (everything else IS the exact-fp32 launch)
            continue
        if H16 is not None and ((not kc and (tile > 4)) or (tile & 32)):
            ts.append(float("inf"))          # register-staged kernel only: the four plain tiles
            continue
        if tile & 32 and (nb_ := batch[0] * batch[1]) * splitk != 1:
            ts.append(float("inf"))          # M-split needs a single un-split grid
            continue
        ts.append(timeit(lambda: orig(op, a, b, out.data_ptr(), M, N, K, lda, ldb, ldc, tile=tile, **kw)))
    nb = batch[0] * batch[1]
    fl = 2.0 * M * N * K * nb * (0.75 if op == 4 else 1.0)
    best = min(range(1, len(TILES)), key=lambda i: ts[i])
    rows.append((ts[0] * n, n, ts[0], fl / ts[0] / 1e12, OPN[op], M, N, K, nb, splitk, conv, ts, best))
rows.sort(reverse=True, key=lambda r: r[0])
tot = sum(r[0] for r in rows)
totb = sum(r[11][r[12]] * r[1] for r in rows)
print(f"sum of isolated igemm time per step: model {tot * 1e3:.2f} ms; best tile per shape {totb * 1e3:.2f} ms")
print(f"{'total us':>9s} {'n':>4s} {'model us':>8s} {'TF/s':>6s}  " + " ".join(f"{x:>8s}" for x in TN_[1:]) + " best     op          M      N      K  batch splitk conv")
for tt, n, t, tf, opn, M, N, K, nb, sk, conv, ts, best in rows[:90]:
    print(f"{tt * 1e6:9.0f} {n:4d} {t * 1e6:8.1f} {tf:6.1f}  " + " ".join(f"{x * 1e6:8.1f}" for x in ts[1:]) +
          f" {TN_[best]:9s} {opn:11s} {M:6d} {N:6d} {K:6d} {nb:5d} {sk:5d}  {conv if conv else ''}")

# ---- autotune table: best tile per descriptor where it beats the model's choice by > 3 %
import json
table = {}
for tt, n, t, tf, opn, M, N, K, nb, sk, conv, ts, best in rows:
    if ts[best] < 0.97 * ts[0]:
        table[f"{OPN.index(opn)}:{M}:{N}:{K}:{nb}:{sk}"] = TILES[best]
out_path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out",
                        (f"tuned_tiles_B{B}.json" if PREC == "f32" else f"tuned_tiles_{'h16_' if H16 is not None else ''}{PREC}_B{B}.json") if WORKLOAD == "kd"
                        else f"tuned_tiles_{PREC}_{WORKLOAD}_B{B}.json")
os.makedirs(os.path.dirname(out_path), exist_ok=True)
json.dump(table, open(out_path, "w"), indent=0, sort_keys=True)
print(f"wrote {len(table)} tuned entries to {out_path}")
