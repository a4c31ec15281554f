#!/usr/bin/env python3
"""bench.py — images/sec of the KD train step (BASELINE.json metric) on N MI355X of one node.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

Workload (config.workload): BASELINE.json configs[2] "Full KD step" (cfg3 of SURVEY.md §8(d)) — ViT-S/16 +
4-layer decoder teacher forward (fp32, no grad), ResNet50+refinement+2-layer-LSTM student forward in TRAIN mode
(BatchNorm batch statistics, reference dropout rates), feature projector, fused KD loss (alpha .7 / beta .2 /
gamma .1 / T 4), backward through decoder / refinement / projection / layer4 / layer3, gradient all-reduce
(N>1, RCCL), clip_grad_norm_(1.0) and AdamW — batch 64 per GPU, V=5000, T=15, synthetic 224x224 inputs resident in
HBM, key-seeded random-init weights (no datasets/checkpoints offline).  One "step" = one optimizer step on one
batch per rank (accumulation_steps=1: strictly more work per image than the reference's accumulate-2 loop).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

GFLOP_PER_IMAGE = 28.4          # algorithmic, SURVEY.md §8(d): teacher 9.80 + student fwd+loss+bwd 18.61 (2 FLOP per MAC)
PEAK_F32_MFMA_TF = 157.3        # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_BF16_MFMA_TF = 2516.0      # same guide: ~2.5 PF dense bf16 (v_mfma_f32_32x32x16_bf16 = 16x the fp32 MFMA rate)
GFLOP_TEACHER, GFLOP_STUDENT = 9.80, 18.61


def mfma_peak(precision: str, gflop_student: float = GFLOP_STUDENT, teacher_precision: str = "f32"):
    """Roofline denominator for the step: the teacher always runs on the fp32 MFMA; the student's contractions run on
    the fp32 MFMA ("f32"), on the bf16 MFMA ("bf16") or as 3 bf16 MFMAs per product ("bf16x3").  The blended peak
    is total FLOPs / (time of each part at its own MFMA peak)."""
    ps = {"f32": PEAK_F32_MFMA_TF, "bf16": PEAK_BF16_MFMA_TF, "fp16": PEAK_BF16_MFMA_TF, "bf16x3": PEAK_BF16_MFMA_TF / 3, "f32x3": None}[precision]
    pt = PEAK_F32_MFMA_TF if teacher_precision == "f32" else PEAK_BF16_MFMA_TF / 3     # "f32x3": three fp16 MFMAs per product
    if precision == "f32x3":      # forward products (8.73 GFLOP/image), the stride-1 data gradients of layer3 / layer4 (~4.0)
        x3 = 8.73 + 4.0 + 4.5     # and their weight gradients (~4.5) as three fp16 MFMAs; stride-2 data gradients + the head's backward exact fp32
        return (GFLOP_TEACHER + gflop_student) / (GFLOP_TEACHER / pt + x3 / (PEAK_BF16_MFMA_TF / 3) + (gflop_student - x3) / PEAK_F32_MFMA_TF)
    return (GFLOP_TEACHER + gflop_student) / (GFLOP_TEACHER / pt + gflop_student / ps)
BATCH = 64
VOCAB, T1 = 5000, 16


GFLOP_HOISTED = 0.85            # of the 28.4: the reference's 49x-redundant W_h.h score GEMM (0.29 fwd + 0.56 bwd per image)
                                # that this build hoists out of the per-position loop — counted by the contract, not executed


def measured_traffic(batch: int, precision: str = "f32"):
    """(HBM bytes per step, source file) from the newest committed PMC run of THIS precision (profiles/r0N_traffic*.json:
    separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this same command, gfx950 x2 fetch correction;
    tools/measure_traffic.py).  PMC counters cannot be collected inside the timed run itself, so the line names its
    source; (None, None) for other batch sizes / precisions without a committed run."""
    names = {"f32": ("r03_traffic.json", "r02k_traffic.json", "r02_traffic.json", "r01_traffic.json"),
             "fp16": ("r03_traffic_fp16.json", "r02k_traffic_fp16.json")}.get(precision, ())
    for name in names:
        try:
            t = json.load(open(os.path.join(ROOT, "profiles", name)))
        except (OSError, ValueError):
            continue
        if batch == BATCH:
            return t["hbm_bytes_per_step_corrected"], f"profiles/{name}"
    return None, None


def host_cpu():
    """(threads this process may use, CPU model string) of the box."""
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    try:                                    # cgroup CPU quota, when the box sets one
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            cores = max(1, min(cores, int(int(q) / int(p))))
    except Exception:
        pass
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return cores, model


def cpu_baseline(batch: int = BATCH, budget_s: float = 20.0):
    """The oracle (CPU restatement, plain PyTorch fp32) timed on ALL of this box's host cores on a bounded sample of the
    same workload at the metric's batch: teacher fwd + student fwd (train-mode BN, dropout off) + KD
    loss + backward with a single ViT pass (`value`, the algorithmic step), the same step as the reference executes it
    (a second, bit-identical ViT pass: distillation_utils.py:278-282), and clip + AdamW timed separately."""
    from oracle import restatement as R
    from imagecaptioner_amd.utils.seeded_init import seeded_state_dict, synthetic_batch
    cores, model = host_cpu()
    torch.set_num_threads(cores)
    trainable = lambda k: not any(k.startswith(f"encoder.resnet.{i}.") for i in (0, 1, 4, 5)) and "running_" not in k
    mk = lambda sd: {k: (v.clone().requires_grad_(True) if (v.dtype.is_floating_point and trainable(k)) else v.clone())
                     for k, v in sd.items()}
    ssd = mk(seeded_state_dict(R.student_state_shapes(VOCAB, 256, 512, 2, True), seed=0))
    tsd = seeded_state_dict(R.teacher_state_shapes(VOCAB, 512, 4), seed=1)
    psd = mk(seeded_state_dict(R.projector_state_shapes(512, 256), seed=2))
    images, caps = synthetic_batch(batch, VOCAB, T1, seed=1234)
    run = lambda: R.kd_forward_backward(ssd, tsd, psd, images, caps, hidden=512, layers=2, refine=True, t_heads=8, t_layers=4)
    leaves = [v for v in list(ssd.values()) + list(psd.values()) if v.requires_grad]

    def timed(fn, min_steps, budget):
        """per-call seconds of `fn`, at least min_steps calls, stopping after ~budget seconds"""
        ts, t_start = [], time.perf_counter()
        while len(ts) < min_steps or (time.perf_counter() - t_start < budget and len(ts) < 40):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        return ts

    def step():
        for v in leaves:
            v.grad = None
        run()

    def vit_again():                        # the reference's redundant second ViT pass (no grad), distillation_utils.py:278-282
        with torch.no_grad():
            R.vit_small_features(tsd, "encoder", images)

    step()                                  # warm-up (thread pools, allocator)
    t_step = timed(step, 2, budget_s * 0.7)
    t_vit = timed(vit_again, 1, budget_s * 0.15)
    opt = torch.optim.AdamW([{"params": [v for k, v in ssd.items() if v.requires_grad and k.startswith("encoder.")], "lr": 2e-5},
                             {"params": [v for k, v in ssd.items() if v.requires_grad and not k.startswith("encoder.")] +
                                        [v for v in psd.values() if v.requires_grad], "lr": 2e-4}], weight_decay=0.01)

    def opt_step():
        torch.nn.utils.clip_grad_norm_([v for v in ssd.values() if v.requires_grad], 1.0)
        torch.nn.utils.clip_grad_norm_([v for v in psd.values() if v.requires_grad], 1.0)
        opt.step()

    opt_step()
    t_opt = timed(opt_step, 3, 1.0)
    mean = lambda xs: sum(xs) / len(xs)
    ts, tv = mean(t_step), mean(t_vit)
    return {"value": round(batch / ts, 3), "unit": "images/s", "cores": cores, "cpu_model": model, "kind": "port",
            "value_as_reference_executes": round(batch / (ts + tv), 3), "optimizer_ms_per_step": round(mean(t_opt) * 1e3, 2),
            "sample": f"{len(t_step)} steps of batch {batch} (teacher fwd + student fwd + KD loss + bwd, fp32, single ViT pass, no "
                      f"optimizer) on {cores} threads, mean {ts:.2f} s/step; value_as_reference_executes adds the reference's "
                      f"duplicate ViT pass ({tv:.2f} s, {len(t_vit)} timed) to the same step time; optimizer (2x clip_grad_norm_ + "
                      f"AdamW) timed separately over {len(t_opt)} steps; torch {torch.__version__} CPU eager"}


def run_kd(args, precision, dev, rank, world, log, student_cfg=None, batch=None, teacher_precision=None):
    """W untimed + K timed KD train steps (hipGraph replays) at the given student precision.  Returns
    (images/s whole job, wall seconds for K steps [max over ranks], device ms for K steps on this rank, loss dict).
    student_cfg / batch default to the command line's (the extras pass cfg5 / 32)."""
    from imagecaptioner_amd.train_student_kd import KDTrainer, build_kd_models
    from imagecaptioner_amd.utils.seeded_init import synthetic_batch

    student_cfg = student_cfg or args.student
    if batch is not None:
        args = argparse.Namespace(**{**vars(args), "batch": batch})
    dims = dict(embed_size=384, hidden_size=768, num_layers=3) if student_cfg == "cfg5" else {}
    student, teacher, projectors = build_kd_models(vocab_size=VOCAB, device=dev, **dims)   # identical init on every rank
    trainer = KDTrainer(student, teacher, projectors, vocab_size=VOCAB, batch_size=args.batch, t_plus_1=T1,
                        use_graph=not args.no_graph, precision=precision, overlap_teacher=not args.no_overlap,
                        teacher_precision=teacher_precision or args.teacher_precision or ("f32" if precision == "f32" else "f32x3"))
    images, caps = synthetic_batch(args.batch, VOCAB, T1, seed=1234, rank=rank)     # rank-specific shard of the global batch
    log(f"[{precision}] models built; first step (hipGraph capture) ...")
    trainer.train_step(images.to(dev), caps.to(dev))                                # inputs resident in HBM from here on
    torch.cuda.synchronize()
    for _ in range(max(0, args.warmup - 1)):
        trainer.train_step()
    torch.cuda.synchronize()
    log(f"[{precision}] timing {args.steps} steps ...")
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        trainer.train_step()
    ev1.record()
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    dev_ms = ev0.elapsed_time(ev1)                      # HIP events on the stream the step's graphs are launched on
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        dt = float(tt.item())
    loss = trainer.loss_dict()
    del trainer, student, teacher, projectors
    torch.cuda.empty_cache()
    return world * args.batch * args.steps / dt, dt, dev_ms, loss


def dominant_kernel(dev, iters=20):
    """The kernel shape with the largest total time in the step, in isolation.  Candidates = the teacher ViT's three big
    Linear shapes (12 launches per step each; profiles/r02k_step_gemm_shapes_tile_sweep.log: fc2 1983, fc1 1916, qkv 1365 us
    per step — every other shape is below 900): each is timed with HIP events around `iters` back-to-back launches on the
    launch stream and the one with the largest time is reported (not a hard-wired fc1).  FLOPs = 2 M N K."""
    from imagecaptioner_amd import ops
    from imagecaptioner_amd._lib import ACT_GELU, ACT_NONE
    M = BATCH * 197
    best = None
    for name, N, K, act, resid in (("ViT fc1 Linear+GELU", 1536, 384, ACT_GELU, False), ("ViT fc2 Linear+residual", 384, 1536, ACT_NONE, True),
                                   ("ViT qkv Linear", 1152, 384, ACT_NONE, False)):
        x = torch.randn(M, K, device=dev)
        w = torch.randn(N, K, device=dev) * 0.05
        b = torch.randn(N, device=dev)
        r = torch.randn(M, N, device=dev) if resid else None
        y = torch.empty(M, N, device=dev)
        f = lambda: ops.linear_fwd(x, w, b, act=act, out=y, residual=r)
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            f()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / iters * 1e3
        if best is None or us > best[0]:
            best = (us, name, N, K)
    us, name, N, K = best
    tf = 2.0 * M * N * K / us / 1e6
    tid = ops._TUNED.get(f"0:{M}:{N}:{K}:1:1", 0)
    tile = {1: "128,128", 2: "64,64", 3: "128,64", 4: "64,128", 18: "64,64,3buf", 19: "128,64,3buf", 20: "64,128,3buf"}.get(
        tid & 31, "cost-model tile") + (",8 waves" if tid & 64 else "") + (",M-split" if tid & 32 else "")
    return {"kernel": f"igemm_glds_kernel<NT,{tile}> {name} {M}x{N}x{K} (fp32 MFMA, LDS-DMA staging, chunked accumulation), 12 launches per step",
            "avg_us": round(us, 1), "achieved": round(tf, 1), "peak": PEAK_F32_MFMA_TF, "unit": "TFLOP/s",
            "frac": round(tf / PEAK_F32_MFMA_TF, 4)}


def run_beam_eval(dev, log, batch=64, beam=5, max_length=20, iters=3):
    """cfg5's "beam=5 eval" (SURVEY 8(f) N1): the teacher's beam search (reference teacher_model.py:108-252) over a batch of
    images with the KV-cached, batched, sync-free search (CaptioningTeacher.beam_search): ViT encode + max_length decode
    steps of B x beam rows + one device->host copy.  Also times the reference's procedure (decoder re-run on the growing
    prefix, one image at a time: caption_image_recompute) on 4 images for the A/B."""
    from imagecaptioner_amd.teacher_model import CaptioningTeacher
    from imagecaptioner_amd.utils.seeded_init import apply_seeded_init, synthetic_batch

    class _V:
        def __init__(self, n):
            self.itos = {i: f"w{i}" for i in range(n)}
            self.itos.update({0: "<PAD>", 1: "<START>", 2: "<END>", 3: "<UNK>"})
            self.stoi = {w: i for i, w in self.itos.items()}
    vocab = _V(VOCAB)
    t = apply_seeded_init(CaptioningTeacher(VOCAB, embed_size=512, num_heads=8, num_decoder_layers=4, dropout=0.15), 1).to(dev).eval()
    images, _ = synthetic_batch(batch, VOCAB, T1, seed=4321)
    images = images.to(dev)
    t.caption_images(images, vocab, max_length=max_length, beam_size=beam)
    torch.cuda.synchronize()
    log("[beam] timing the batched KV-cached beam search ...")
    t0 = time.perf_counter()
    for _ in range(iters):
        caps = t.caption_images(images, vocab, max_length=max_length, beam_size=beam)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    from imagecaptioner_amd import ops
    with ops.precision("f32x3"):           # the same search with every Linear as three fp16 MFMAs per product (fp32-grade)
        caps3 = t.caption_images(images, vocab, max_length=max_length, beam_size=beam)
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        for _ in range(iters):
            caps3 = t.caption_images(images, vocab, max_length=max_length, beam_size=beam)
        torch.cuda.synchronize()
        dt3 = (time.perf_counter() - t3) / iters
    t.caption_image_recompute(images[0], vocab, max_length=max_length, beam_size=beam)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for b in range(4):
        t.caption_image_recompute(images[b], vocab, max_length=max_length, beam_size=beam)
    torch.cuda.synchronize()
    dt_ref = (time.perf_counter() - t1) / 4
    out = {"workload": f"teacher beam search, beam {beam}, max_length {max_length}, batch {batch}: ViT-S/16 encode + KV-cached decode of "
                       f"{batch * beam} beam rows per step, on-device top-k / finishing / length bookkeeping, one D2H copy per batch",
           "dtype": "f32", "images_per_s": round(batch / dt, 1), "beam_tokens_per_s": round(batch * beam * max_length / dt, 1),
           "ms_per_batch": round(dt * 1e3, 2),
           "prefix_rerun_one_image_at_a_time_images_per_s": round(1.0 / dt_ref, 2), "speedup_vs_prefix_rerun": round(dt_ref * batch / dt, 1),
           "mean_caption_words": round(sum(len(c[0].split()) for c in caps) / len(caps), 2),
           "f32x3": {"images_per_s": round(batch / dt3, 1), "ms_per_batch": round(dt3 * 1e3, 2),
                     "captions_identical_to_f32": sum(a == b for a, b in zip(caps, caps3)), "of": len(caps)}}
    del t
    torch.cuda.empty_cache()
    return out


def run_cfg2(dev, log, batch=128, max_length=20, iters=10):
    """BASELINE.json configs[1]: student ResNet50+LSTM (256/512/2-layer) forward + batched greedy decode in bf16 at
    batch 128 on one GPU (captured once into a hipGraph, no per-token host sync).  8.87 algorithmic GFLOP/image
    (encoder + refinement 8.28 + 20 tokens x 0.0297)."""
    from imagecaptioner_amd import ops
    from imagecaptioner_amd.student_model import CaptioningStudent
    from imagecaptioner_amd.utils.seeded_init import apply_seeded_init, synthetic_batch
    m = apply_seeded_init(CaptioningStudent(VOCAB, 256, 512, 2), 0).to(dev).eval()
    images, _ = synthetic_batch(batch, VOCAB, T1, seed=4321)
    images = images.to(dev)
    with ops.precision("bf16"):
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            m.generate(images, max_length)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            ids, _ = m.generate(images, max_length)
    g.replay()
    torch.cuda.synchronize()
    log("[cfg2] timing the captured forward + greedy decode ...")
    t0 = time.perf_counter()
    for _ in range(iters):
        g.replay()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    out = {"workload": "cfg2: student ResNet50+LSTM (256/512/2-layer, refinement on) eval forward + 20-step batched greedy decode, "
                       f"batch {batch}, 1 GPU, hipGraph", "dtype": "bf16 (fp32 accumulate)", "images_per_s": round(batch / dt, 1),
           "decode_tokens_per_s": round(batch * max_length / dt, 1), "ms_per_batch": round(dt * 1e3, 3),
           "achieved_TFLOPs": round(8.87 * batch / dt / 1e3, 2)}
    del g, m
    torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=BATCH, help="per-GPU batch (BASELINE: 64)")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--student", default="cfg3", choices=["cfg3", "cfg5"],
                    help="cfg3 (default, the metric's config): student 256/512/2-layer; cfg5: the large student 384/768/3-layer "
                         "(BASELINE configs[4]; use --batch 32 for its per-GPU batch)")
    ap.add_argument("--dominant-kernel-only", action="store_true",
                    help="only time the dominant kernel in isolation (for the matching rocprofv3 --kernel-trace --stats run)")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary measurements (bf16 step, cfg2 decode) at N=1")
    ap.add_argument("--no-overlap", action="store_true", help="teacher forward on the main stream instead of a parallel graph branch")
    ap.add_argument("--teacher-precision", default=None, choices=["f32", "f32x3"],
                    help="teacher Linear arithmetic: exact fp32 MFMA (default with --precision f32, the headline's regime) or fp32-grade "
                         "three-fp16-product GEMMs (default otherwise)")
    ap.add_argument("--precision", default="f32", choices=["f32", "bf16", "fp16", "bf16x3", "f32x3"],
                    help="student/projector GEMM arithmetic (teacher stays fp32 as in the reference); f32 = parity regime")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` with no launcher around it: start the N ranks ourselves — fresh child processes,
        # BEFORE this process makes any GPU call — and relay their output (rank 0 prints the JSON line).
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.run(cmd, env=env).returncode)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.distributed.init_process_group("nccl", device_id=dev)      # "nccl" is RCCL on ROCm
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU "
                         f"(python -m torch.distributed.run --nproc-per-node {args.gpus} ...), or run without a launcher")
    n_seen = 1
    if world > 1:      # the line's n_gpus is what an RCCL all-reduce of a rank counter saw, not what the flags claim
        one = torch.ones(1, device=dev)
        torch.distributed.all_reduce(one)
        n_seen = int(one.item())
        assert n_seen == world, (n_seen, world)

    log = (lambda m: print(f"[bench rank {rank}] {m}", file=sys.stderr, flush=True))
    if args.dominant_kernel_only:
        print(json.dumps({"dominant_kernel": dominant_kernel(dev, iters=50)}), flush=True)
        return
    ips, dt, dev_ms, loss = run_kd(args, args.precision, dev, rank, world, log)
    if rank == 0:
        ips = world * args.batch * args.steps / dt
        step_ms_dev = dev_ms / args.steps
        gflop_img = GFLOP_PER_IMAGE if args.student == "cfg3" else 30.7   # cfg5 = 30.7 algorithmic GFLOP/image
        achieved = gflop_img * args.batch / step_ms_dev                   # GFLOP / ms = TFLOP/s, this rank's GPU
        traffic, traffic_src = measured_traffic(args.batch, args.precision)
        tprec = args.teacher_precision or ("f32" if args.precision == "f32" else "f32x3")
        peak = mfma_peak(args.precision, gflop_img - GFLOP_TEACHER, teacher_precision=tprec)
        dtype = {"f32": "f32", "bf16": "bf16 student (16-bit activation + weight-shadow storage in the trunk, fp32 accumulate, fp32 master weights) + f32 teacher",
                 "fp16": "fp16 student (16-bit activation + weight-shadow storage in the trunk, fp32 accumulate, fp32 master weights, device GradScaler) + f32 teacher",
                 "bf16x3": "split-bf16x3 student + f32 teacher",
                 "f32x3": "f32-grade: forward Linears / convolutions + the trunk's stride-1 data gradients and weight gradients as three fp16 MFMAs per product, the rest exact fp32 MFMA"}[args.precision]
        out = {
            "metric": "images/sec KD train step (teacher+student fwd + KD loss + bwd)", "value": round(ips, 2),
            "unit": "images/s", "n_gpus": n_seen, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": dtype, "data": "synthetic",
            "config": {"workload": f"{args.student} full KD step: vit_small_patch16_224 teacher (embed 512/8 heads/4 layers) + "
                                   f"ResNet50-LSTM student ({'256/512/2' if args.student == 'cfg3' else '384/768/3'}-layer, refinement on), alpha=0.7 beta=0.2 gamma=0.1 T=4, "
                                   "V=5000, T=15, 224x224, clip 1.0 + AdamW every step",
                       "per_gpu_batch": args.batch, "global_batch": world * args.batch, "parallelism": f"dp{world}",
                       "hipgraph": not args.no_graph, "final_loss": round(loss["total_loss"], 5)},
            "roofline": {"bound": "mfma", "achieved": round(achieved, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
                         "frac": round(achieved / peak, 4), "traffic": traffic, "traffic_source": traffic_src,
                         "frac_executed": round(achieved * (gflop_img - GFLOP_HOISTED) / gflop_img / peak, 4),
                         "note": f"whole step: {gflop_img} algorithmic GFLOP/image (SURVEY 8d) x batch / device time per step "
                                 "(HIP events on the launch stream); denominator = fp32 MFMA peak for the exact-fp32 path, else the "
                                 "FLOP-weighted blend of the fp32 (teacher) and bf16 (student) MFMA peaks; frac_executed discounts the "
                                 f"{GFLOP_HOISTED} GFLOP/image of redundant score GEMM the contract counts and this build hoists away"},
        }
        if world == 1 and not args.no_extras and args.student == "cfg3":
            out["roofline"]["dominant_kernel"] = dominant_kernel(dev)
            # secondary measurements, same process: the reference's mixed-precision regime and BASELINE configs[1]
            if args.precision == "f32":
                # the reference's AMP regime (train_student_kd.py:239,271: fp16 autocast + GradScaler), then its bf16 twin
                out["mixed_precision"] = {}
                X3 = "teacher Linears fp32-GRADE from three fp16 MFMAs per product (igemm_glds_impl.h TERMS 4: error vs float64 = the exact-fp32 kernel's)"
                S16 = "student on 16-bit activation + weight-shadow storage in the trunk, fp32 accumulate, fp32 master weights"
                for key2, prec2, tprec2, label in (
                        ("fp16", "fp16", "f32x3", f"fp16 {S16}, device GradScaler 2^16; {X3} — KDTrainer's default teacher in this regime"),
                        ("bf16", "bf16", "f32x3", f"bf16 {S16}; {X3}"),
                        ("fp16_exact_fp32_teacher", "fp16", "f32", f"fp16 {S16}, device GradScaler 2^16; teacher on the exact fp32 MFMA (rounds 1-2's fp16 line)"),
                        ("f32x3", "f32x3", "f32x3", "fp32-grade step: every FORWARD Linear / convolution of teacher and student and the trunk's stride-1 data gradients + weight gradients (dY scaled by "
                                                    "its device-side absmax) as three fp16 MFMAs per product, stride-2 data gradients and the head's backward exact fp32 MFMA "
                                                    "(tests/test_kd_step_b16_gpu.py[f32x3]: same fp64 yardstick as the exact path)"),
                        ("f32_teacher_f32x3", "f32", "f32x3", f"student exact fp32 MFMA; {X3}")):
                    ips2, dt2, dev2, loss2 = run_kd(args, prec2, dev, rank, world, log, teacher_precision=tprec2)
                    ach2 = GFLOP_PER_IMAGE * args.batch / (dev2 / args.steps)
                    out["mixed_precision"][key2] = {"dtype": label, "value": round(ips2, 2), "unit": "images/s",
                                                     "ms_per_step": round(dt2 / args.steps * 1e3, 3),
                                                     "final_loss": round(loss2["total_loss"], 5), "achieved_TFLOPs": round(ach2, 2),
                                                     "blended_peak_TFLOPs": round(mfma_peak(prec2, teacher_precision=tprec2), 1)}
            out["cfg2"] = run_cfg2(dev, log)
            # cfg5's per-rank workload (large student 384/768/3 + teacher, per-GPU batch 32 of the 8-GPU global batch 256),
            # 30.7 algorithmic GFLOP/image, and its "beam=5 eval"
            out["cfg5_per_rank"] = {}
            for prec5 in ("f32", "fp16", "f32x3"):
                ips5, dt5, dev5, loss5 = run_kd(args, prec5, dev, rank, world, log, student_cfg="cfg5", batch=32)
                ach5 = 30.7 * 32 / (dev5 / args.steps)
                pk5 = mfma_peak(prec5, 30.7 - GFLOP_TEACHER, teacher_precision="f32" if prec5 == "f32" else "f32x3")
                out["cfg5_per_rank"][prec5] = {"workload": "cfg5 KD step: student 384/768/3-layer + ViT teacher" + ("" if prec5 == "f32" else " (fp32-grade f32x3 Linears)") + ", per-GPU batch 32", "value": round(ips5, 2),
                                               "unit": "images/s", "ms_per_step": round(dt5 / args.steps * 1e3, 3),
                                               "final_loss": round(loss5["total_loss"], 5), "achieved_TFLOPs": round(ach5, 2),
                                               "peak_TFLOPs": round(pk5, 1), "frac": round(ach5 / pk5, 4)}
            out["beam5_eval"] = run_beam_eval(dev, log)
        if world == 1 and not args.no_cpu_baseline:
            log(f"GPU: {ips:.1f} images/s; timing the CPU baseline on a bounded sample ...")
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
