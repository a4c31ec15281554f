"""The KD train step, MI355X-native — the loop body of the reference's train_student_with_kd()
(/root/reference/src/train_student_kd.py:258-303) as a reusable object, plus what the reference lacks:
data-parallel replication over one 8xMI355X node with ONE RCCL all-reduce of the flat gradient buffer.

Design (MI355X-first):
  * every trainable tensor (student layer3/4 + projection + refinement + decoder, encoder projector) is a
    VIEW into one flat fp32 parameter buffer; `.grad`s are views into one flat gradient buffer; Adam moments
    are flat too.  Gradient norm = one streaming reduction, AdamW = one fused pass per LR group (28 B/param),
    gradient exchange = one all-reduce of 120 MB over xGMI — no per-tensor launches.
  * forward + loss + backward is captured once into a hipGraph (torch.cuda.CUDAGraph) and replayed: the ~1.5k
    kernel launches of a step cost one graph launch on the host; so is the clip+AdamW tail.  The all-reduce
    runs between the two graphs on the same stream.
  * all-reduce happens BEFORE clipping, so every rank derives the same norm and takes the same step with no
    further communication; equal per-rank batches make SUM/world exact for the KL / MSE / cosine terms.
  * the LR schedule (CosineAnnealingWarmRestarts T_0=5, T_mult=2, eta_min=1e-6, fractional-epoch stepping,
    reference :236,:303) and Adam bias corrections live in a small device buffer updated per step, so the
    captured optimizer graph follows the schedule.
"""
from __future__ import annotations

import math
import os
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn as nn

from . import dp
from . import nn as hnn
from . import ops
from .distillation_utils import DistillationLoss, TeacherWrapper


def cosine_warm_restarts_state(epoch: float, T_0: int = 5, T_mult: int = 2) -> Tuple[float, float]:
    """(T_cur, T_i) of torch.optim.lr_scheduler.CosineAnnealingWarmRestarts after .step(epoch), closed form for
    fractional epochs (reference: stepped with epoch + batch_idx / len(loader), train_student_kd.py:236,:303)."""
    if epoch >= T_0:
        if T_mult == 1:
            return epoch % T_0, T_0
        n = int(math.log(epoch / T_0 * (T_mult - 1) + 1, T_mult))
        return epoch - T_0 * (T_mult ** n - 1) / (T_mult - 1), T_0 * T_mult ** n
    return epoch, T_0


def cosine_warm_restarts_factor(epoch: float, T_0: int = 5, T_mult: int = 2) -> float:
    """(1 + cos(pi * T_cur / T_i)) / 2; a group's LR is eta_min + (base_lr - eta_min) * factor."""
    t_cur, t_i = cosine_warm_restarts_state(epoch, T_0, T_mult)
    return (1 + math.cos(math.pi * t_cur / t_i)) / 2


class FlatParams:
    """Re-homes a list of parameter groups into flat fp32 buffers (params, grads) + Adam moments."""

    def __init__(self, groups: List[Tuple[str, List[nn.Parameter]]], device):
        self.segments: List[Tuple[str, int, int]] = []           # (name, start, end) in elements
        metas = []
        off = 0
        seen = set()
        for name, plist in groups:
            start = off
            for p in plist:
                if not p.requires_grad or id(p) in seen:
                    continue
                seen.add(id(p))
                n = p.numel()
                metas.append((p, off, n))
                off += (n + 3) // 4 * 4                           # keep every tensor 16-byte aligned
            self.segments.append((name, start, off))
        self.total = off
        self.param = torch.zeros(off, dtype=torch.float32, device=device)
        self.grad = torch.zeros(off, dtype=torch.float32, device=device)
        self.exp_avg = torch.zeros(off, dtype=torch.float32, device=device)
        self.exp_avg_sq = torch.zeros(off, dtype=torch.float32, device=device)
        with torch.no_grad():
            for p, o, n in metas:
                if p.dim() == 4:                                  # conv weight: physical [Cout][R][S][Cin]
                    co, ci, r, s = p.shape
                    view = self.param[o:o + n].view(co, r, s, ci).permute(0, 3, 1, 2)
                    gview = self.grad[o:o + n].view(co, r, s, ci).permute(0, 3, 1, 2)
                else:
                    view = self.param[o:o + n].view(p.shape)
                    gview = self.grad[o:o + n].view(p.shape)
                view.copy_(p.data)
                p.data = view
                p.grad = gview
        self.metas = metas

    def segment(self, name: str) -> Tuple[int, int]:
        for n, a, b in self.segments:
            if n == name:
                return a, b
        raise KeyError(name)


class KDTrainer:
    """One object = the reference's training-loop state: models, loss, optimizer state, schedule."""

    def __init__(self, student, teacher, projectors: Dict[str, nn.Module], *, vocab_size: int, alpha=0.7, beta=0.2,
                 gamma=0.1, temperature=4.0, learning_rate=2e-4, weight_decay=0.01, max_norm=1.0, batches_per_epoch=1000,
                 batch_size: int = 64, t_plus_1: int = 16, use_graph: bool = True, process_group=None,
                 precision: str = "f32", teacher_precision: Optional[str] = None, overlap_teacher: bool = True,
                 accumulation_steps: int = 1, loss_scale=None, growth_interval: int = 2000, bucketed: Optional[bool] = None):
        """precision: arithmetic of the student + projector contractions, forward and backward ("f32" exact; "fp16" = the
        reference's autocast regime :271-285 — fp16 MFMA products, fp32 accumulation and master weights, GradScaler on
        the device; "bf16" the same with bf16 products and no scaler; "bf16x3" split-bf16); the teacher runs
        outside autocast in fp32 in the reference (:265-268) -> teacher_precision is "f32" (exact fp32 MFMA)
        under precision "f32" and "f32x3" otherwise: fp32-GRADE Linears from three fp16 MFMAs per product (igemm_glds_impl.h
        TERMS 4; error against float64 equal to the exact kernel's, tests/test_gemm_gpu.py::test_f32x3_is_fp32_grade), 2x the
        exact kernel — in the mixed-precision regimes the fp32 teacher is otherwise more than half of the step.
        precision "f32x3": every forward Linear / convolution of the student that way too, and the trunk's stride-1 data gradients
        and weight gradients with the power-of-two scale of dY's device-side absmax (IckGemm.a_absmax); stride-2 data gradients
        and the head's backward on the exact fp32 MFMA."""
        if teacher_precision is None:
            teacher_precision = "f32" if precision == "f32" else "f32x3"
        self.precision, self.teacher_precision = precision, teacher_precision
        if precision == "fp16" and loss_scale is None:
            loss_scale = 65536.0         # torch.amp.GradScaler's init_scale: fp16's 5-bit exponent needs it (reference :239)
        # gradient accumulation (reference :229,:285,:290): `loss / accumulation_steps` per micro-batch and one
        # optimizer step per window.  Here the flat gradient buffer simply accumulates the un-divided gradients and
        # 1/accumulation_steps is folded, with 1/world, into the fused clip+AdamW pass; the all-reduce and the LR
        # schedule run on window boundaries only.
        self.accumulation_steps = max(1, int(accumulation_steps))
        self.micro_idx = 0
        # loss_scale: None = no scaling (fp32 / bf16 need none); a float = GradScaler(init_scale=that) with the torch
        # defaults growth 2.0 / backoff 0.5 / growth_interval (reference :239,:288-298).  The state lives on the device
        # ({scale, 1/scale, found_inf, good_steps}) so the captured graphs carry scale(), unscale_(), step(), update().
        self.loss_scale0, self.growth_interval = loss_scale, int(growth_interval)
        self.side_stream = torch.cuda.Stream() if (overlap_teacher and torch.cuda.is_available()) else None
        self.student, self.teacher, self.projectors = student, teacher, projectors
        self.device = next(student.parameters()).device
        self.teacher_wrapper = TeacherWrapper(teacher)
        self.loss = DistillationLoss(alpha, beta, gamma, temperature, vocab_size)
        self.loss.unit_grad_fastpath = True
        self.lr, self.wd, self.max_norm = learning_rate, weight_decay, max_norm
        self.betas, self.eps = (0.9, 0.999), 1e-8
        self.batches_per_epoch = batches_per_epoch
        self.pg = process_group
        self.world = torch.distributed.get_world_size(process_group) if (process_group is not None or
                                                                          torch.distributed.is_initialized()) else 1
        # LR groups exactly as train_student_kd.py:219-234 (encoder lr x0.1; decoder; refinement + projectors)
        other = list(student.attention_refinement.parameters()) if student.use_attention_refinement else []
        proj_params = [p for pr in projectors.values() for p in pr.parameters()]
        self.flat = FlatParams([("encoder", list(student.encoder.parameters())), ("decoder", list(student.decoder.parameters())),
                                ("refine", other), ("projector", proj_params)], self.device)
        self.group_lr_mult = {"encoder": 0.1, "decoder": 1.0, "refine": 1.0, "projector": 1.0}
        self.hyper = torch.zeros(4, 4, dtype=torch.float32, device=self.device)     # per group: lr, 1-b1^t, 1-b2^t, -
        # the LR column is uploaded per optimizer step from a RING of pinned buffers, each guarded by an event: the host
        # runs ahead of the stream (train_step never synchronises), so one buffer would be overwritten while earlier
        # copies are still queued.  The bias corrections are computed ON DEVICE from a device-side count of APPLIED
        # steps (ick_adam_bias_correction inside the optimizer graph): a step the GradScaler skips does not advance t,
        # exactly like torch's scaler.step(optimizer).
        cuda = self.device.type == "cuda"
        self._hyper_ring = [torch.zeros(4, dtype=torch.float32).pin_memory() if cuda else torch.zeros(4) for _ in range(8)]
        self.lr_dev = torch.zeros(4, dtype=torch.float32, device=self.device)      # this step's LR per group, contiguous
        self._hyper_events = [None] * len(self._hyper_ring)
        self._hyper_slot = 0
        self.applied_steps_dev = torch.zeros(1, dtype=torch.int64, device=self.device)
        self.norms = torch.zeros(2, dtype=torch.float32, device=self.device)        # [student norm, projector norm]
        self.ws = torch.zeros(1024, dtype=torch.float32, device=self.device)
        self.step_count = 0          # optimizer windows completed on the host side (applied_steps() excludes skipped ones)
        self.batch_idx = 0
        self.epoch = 0
        self.drop_step = torch.zeros(1, dtype=torch.int64, device=self.device)
        ops.set_dropout_step_counter(self.drop_step)
        if self.world > 1:           # every rank draws its own dropout masks on its shard
            rank = torch.distributed.get_rank(process_group)
            hnn.manual_seed(hnn._seed_state["seed"] ^ (0x9E3779B1 * (rank + 1) & 0xFFFFFFFF))
        self.images = torch.zeros(batch_size, 3, 224, 224, dtype=torch.float32, device=self.device)
        self.captions = torch.zeros(t_plus_1, batch_size, dtype=torch.int64, device=self.device)
        self.scaler = None
        if self.loss_scale0 is not None:
            s0 = float(self.loss_scale0)
            self.scaler = torch.tensor([s0, 1.0 / s0, 0.0, 0.0], dtype=torch.float32, device=self.device)
            self.loss.unit_grad_fastpath = False          # the seed gradient of total_loss is the device-resident scale
        self.out5 = None
        self.use_graph = use_graph and self.device.type == "cuda"
        self.g_fb: Optional[torch.cuda.CUDAGraph] = None
        self.g_opt: Optional[torch.cuda.CUDAGraph] = None
        # Data-parallel step = three stages with a gradient bucket leaving after each (dp.gradient_buckets): (0) forward +
        # loss + backward down to the trunk boundary, (1) layer4 backward, (2) layer3 backward.  bucketed=None: on for
        # world > 1; True forces the staged step on one rank too (tests: it must equal the single-graph step).
        names = {id(p): k for k, p in student.named_parameters()}
        names.update({id(p): f"projector.{k}.{n}" for k, pr in projectors.items() for n, p in pr.named_parameters()})
        self.buckets = dp.gradient_buckets(self.flat.metas, self.flat.total, names)
        # bucketed=None: OFF unless ICK_DP_BUCKETED=1.  The staged step has never run on more than one GPU (no 2+ GPU box was
        # available to this build; SCALE_r01/r02 are skipped records), so the default for world > 1 is the flat single
        # all-reduce between the two graphs; opt in with bucketed=True / the environment variable once a multi-GPU run of
        # tests/test_configs_gpu.py::test_two_rank_rccl_step_matches_serial_average is on record.
        self.bucketed = (os.environ.get("ICK_DP_BUCKETED", "0") == "1" and self.world > 1) if bucketed is None else bool(bucketed)
        if len(self.buckets) != 3:
            self.bucketed = False
        # test hook (one-GPU rehearsal of the staged step): a float s makes _reduce_bucket multiply its bucket by s on the
        # communication stream instead of the (identity at world 1) all-reduce, so that a bucket launched before its gradients
        # are complete, a missed or doubled range or a missing stream join changes the result.
        self._test_bucket_scale: Optional[float] = None
        self.comm_stream = torch.cuda.Stream() if (self.bucketed and self.device.type == "cuda") else None
        # ICK_WGRAD_STREAM=1: weight gradients of the trunk on their own stream / graph branch (hnn.set_wgrad_stream).  Off by
        # default — measured: the fp32 step gets SLOWER (28.19 vs 27.58 ms: two matrix-pipe-bound GEMMs share the LDS and L2 of
        # every CU), the fp16 step gains 0.7 % (17.59 vs 17.71 ms), inside the box-to-box noise.
        self.wgrad_stream = torch.cuda.Stream() if (cuda and os.environ.get("ICK_WGRAD_STREAM", "0") == "1") else None
        # 16-bit training regime (hnn._TRUNK16): the trunk reads bf16 / fp16 weights.  One cast of the encoder segment of the
        # flat parameter buffer per step keeps a flat 16-bit shadow current; every trainable conv weight gets a view of it.
        self.flat16 = None
        self._shadow_keys = []
        dt16 = hnn._H16_OF.get(precision)
        if dt16 is not None and hnn._TRUNK16[0] and cuda:
            a, b = self.flat.segment("encoder")
            self.flat16 = torch.zeros(b, dtype=dt16, device=self.device)
            for p, o, n in self.flat.metas:
                if p.dim() == 4 and o + n <= b:
                    co, ci, r, s_ = p.shape
                    self._shadow_keys.append(hnn.install_weight_shadow(p, self.flat16[o:o + n].view(co, r, s_, ci)))
        self.g_stage: List[Optional[torch.cuda.CUDAGraph]] = [None, None, None]
        self._trunk_state = None
        self._l4_first = None

    def close(self) -> None:
        """Drops what this trainer registered outside itself (the 16-bit weight shadows: nn._WEIGHT_SHADOW would otherwise keep
        flat16 alive after the trainer is gone).  Called by __del__; idempotent."""
        hnn.remove_weight_shadows(getattr(self, "_shadow_keys", ()))
        self._shadow_keys = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ the step body
    def _stage(self, k: int):
        """stage k of the staged (data-parallel) step; see __init__."""
        hnn.set_wgrad_stream(self.wgrad_stream)
        if k == 0:
            with hnn.deferred_trunk_backward() as box:
                out = self._forward_backward()
            self._trunk_state = box.state
            if self._l4_first is None and box.state is not None:
                blocks = box.state["blocks"]
                l4 = set(id(b) for b in self.student.encoder.resnet[7])
                self._l4_first = next(i for i, b in enumerate(blocks) if id(b) in l4)
            return out
        if self._trunk_state is not None:
            with ops.precision(self.precision), hnn.weight_shadows_current():
                hnn.trunk_backward_stage(self._trunk_state, self._l4_first if k == 1 else 0)
            if k == 2:
                self._trunk_state = None
        hnn.set_wgrad_stream(None)

    def _reduce_bucket(self, k: int, boundary: bool):
        """all-reduce(SUM) of bucket k on the communication stream, behind everything the launch stream has enqueued so far."""
        if not boundary or (self._test_bucket_scale is None and self.world <= 1 and self.pg is None and
                            not torch.distributed.is_initialized()):
            return
        a, b = self.buckets[k]
        if self.comm_stream is None:
            dp.allreduce_gradients(self.flat.grad[a:b], self.pg)
            return
        self.comm_stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.comm_stream):
            if self._test_bucket_scale is not None:
                self.flat.grad[a:b].mul_(self._test_bucket_scale)
            else:
                dp.allreduce_gradients(self.flat.grad[a:b], self.pg, force=True)

    def _forward_backward(self):
        """teacher fwd (no grad, fp32) -> student fwd -> projector -> KD loss -> backward  (reference :262-288)."""
        if self.accumulation_steps == 1:
            self.flat.grad.zero_()   # gradients stay inspectable after the step; with accumulation the buffer is
        self.drop_step += 1          # zeroed at the end of the optimizer pass instead, so a window accumulates
        if self.flat16 is not None:  # this step's 16-bit copy of the trunk's master weights (after the previous AdamW pass)
            ops.cast16(self.flat.param[:self.flat16.numel()], self.flat16.dtype, out=self.flat16)
        cin, ctg = self.captions[:-1], self.captions[1:]
        # The frozen teacher's forward does not depend on the student: it runs on a side HIP stream (a parallel
        # branch of the captured graph), so its large GEMMs fill the CUs that the student's latency-bound
        # per-token decoder kernels leave idle.  Joined before the loss.
        side = self.side_stream
        if side is not None:
            cur = torch.cuda.current_stream()
            side.wait_stream(cur)
            with torch.cuda.stream(side), ops.precision(self.teacher_precision):
                t_out = self.teacher_wrapper(self.images, cin)
        else:
            with ops.precision(self.teacher_precision):
                t_out = self.teacher_wrapper(self.images, cin)
        with ops.precision(self.precision), hnn.weight_shadows_current():     # (flat16 was cast at the top of this step)
            logits, enc, hids, _ = self.student(self.images, cin)
            if side is not None:
                cur.wait_stream(side)
                for v in t_out.values():
                    if torch.is_tensor(v):
                        v.record_stream(cur)
            s_out = {"logits": logits, "encoder_features": enc, "hidden_states": hids}
            t_out["encoder_features"] = self.projectors["encoder"](t_out["encoder_features"])
            out5 = self.loss.forward_device(s_out, t_out, ctg)
            hnn.set_wgrad_stream(self.wgrad_stream)
            try:
                if self.scaler is None:
                    out5[0].backward()
                else:
                    out5[0].backward(self.scaler[0])          # scaler.scale(loss).backward()
            finally:
                hnn.set_wgrad_stream(None)
        self.out5 = out5
        return out5

    def _optimizer(self):
        """clip_grad_norm_(student, 1.0); clip per projector; AdamW (reference :292-299), fused over the flat buffers.
        Gradients hold the SUM over ranks at this point; 1/world is folded into the kernels."""
        f = self.flat
        inv = 1.0 / (self.world * self.accumulation_steps)
        a0, _ = f.segment("encoder")
        _, b2 = f.segment("refine")
        pa, pb = f.segment("projector")
        ops.grad_norm(f.grad[a0:b2], self.ws, self.norms[0:1])
        if pb > pa:
            ops.grad_norm(f.grad[pa:pb], self.ws, self.norms[1:2])
        if self.scaler is not None:
            ops.loss_scale_check(self.norms, self.scaler)
        ops.adam_bias_correction(self.applied_steps_dev, self.scaler, self.betas, self.hyper, self.lr_dev)
        for gi, name in enumerate(("encoder", "decoder", "refine", "projector")):
            a, b = f.segment(name)
            if b <= a:
                continue
            norm = self.norms[1:2] if name == "projector" else self.norms[0:1]
            ops.adamw_step(f.param[a:b], f.grad[a:b], f.exp_avg[a:b], f.exp_avg_sq[a:b], 0.0, self.betas, self.eps, self.wd, 0,
                           norm=norm, max_norm=self.max_norm, inv_scale=inv, hyper=self.hyper[gi], scaler=self.scaler)
        if self.scaler is not None:
            ops.loss_scale_update(self.scaler, 2.0, 0.5, self.growth_interval)
        if self.accumulation_steps > 1:
            f.grad.zero_()                               # optimizer.zero_grad() (reference :299)

    def _update_hyper(self):
        ep = self.epoch + self.batch_idx / max(1, self.batches_per_epoch)
        slot = self._hyper_slot
        self._hyper_slot = (slot + 1) % len(self._hyper_ring)
        if self._hyper_events[slot] is not None:
            self._hyper_events[slot].synchronize()       # the copy that last read this pinned buffer has executed
        host = self._hyper_ring[slot]
        for gi, name in enumerate(("encoder", "decoder", "refine", "projector")):
            # the scheduler is stepped AFTER optimizer.step() in the reference (:299-303): step k uses the LR set at k-1
            base = self.lr * self.group_lr_mult[name]
            host[gi] = self.eta_min + (base - self.eta_min) * self._f_now
        # ONE contiguous pinned -> device copy (a strided copy would be staged through a pageable temporary and block the
        # host); ick_adam_bias_correction scatters lr_dev into the hyper rows inside the optimizer graph
        self.lr_dev.copy_(host, non_blocking=True)
        if self.device.type == "cuda":
            ev = torch.cuda.Event()
            ev.record()
            self._hyper_events[slot] = ev
        self._f_next = cosine_warm_restarts_factor(ep)

    def applied_steps(self) -> int:
        """optimizer steps actually applied (synchronises): excludes the ones the GradScaler skipped on inf / nan."""
        return int(self.applied_steps_dev.item())

    _f_now = 1.0
    _f_next = 1.0
    eta_min = 1e-6

    def _capture(self):
        bufs = [(b, b.clone()) for b in self.student.buffers()]       # warm-up must not count as training steps
        if self.scaler is not None:
            bufs.append((self.scaler, self.scaler.clone()))
        bufs.append((self.applied_steps_dev, self.applied_steps_dev.clone()))
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(2):                                    # warm-up: allocator, lazy inits, BN counters
                self._forward_backward()
                self._optimizer_dry()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        self.g_fb = torch.cuda.CUDAGraph()
        # thread_local: under torch.distributed the RCCL watchdog thread polls events while this thread captures
        if self.bucketed:
            # three graphs replayed back to back share one memory pool (stage k reads what stage k-1 left behind)
            with torch.cuda.graph(self.g_fb, capture_error_mode="thread_local"):
                self._stage(0)
            self.g_stage[0] = self.g_fb
            for k in (1, 2):
                self.g_stage[k] = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.g_stage[k], pool=self.g_fb.pool(), capture_error_mode="thread_local"):
                    self._stage(k)
        else:
            with torch.cuda.graph(self.g_fb, capture_error_mode="thread_local"):
                self._forward_backward()
        self.g_opt = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.g_opt, capture_error_mode="thread_local"):
            self._optimizer()
        with torch.no_grad():
            for b, saved in bufs:
                b.copy_(saved)
            self.drop_step.zero_()

    def _optimizer_dry(self):
        """warm-up call of the optimizer kernels that leaves parameters untouched (lr = 0, moments restored)."""
        self.hyper.zero_()           # lr = 0 (the bias-correction columns are rewritten by the device-side counter)
        self.lr_dev.zero_()
        m, v = self.flat.exp_avg.clone(), self.flat.exp_avg_sq.clone()
        wd, self.wd = self.wd, 0.0
        self._optimizer()
        self.wd = wd
        self.flat.exp_avg.copy_(m)
        self.flat.exp_avg_sq.copy_(v)

    # ------------------------------------------------------------------ public
    def train_step(self, images: Optional[torch.Tensor] = None, captions: Optional[torch.Tensor] = None):
        """One KD step on this rank's batch.  Returns out5 (device tensor: total, ce, token_kd, feature_kd, hidden_kd)
        without synchronising.  images (B,3,224,224) fp32, captions (T+1,B) int64; None = reuse the resident batch."""
        if images is not None:
            self.images.copy_(images, non_blocking=True)
        if captions is not None:
            self.captions.copy_(captions, non_blocking=True)
        self.student.train()
        hnn.bump_param_generation()      # this step rewrites weights and BatchNorm running statistics by raw pointer
        if self.use_graph and self.g_fb is None:
            self._capture()
        # window boundary by the per-epoch batch index, as the reference counts it (:290): a window cut short by the end
        # of an epoch keeps its gradients and completes in the next epoch
        boundary = (self.batch_idx + 1) % self.accumulation_steps == 0
        if boundary:
            self._update_hyper()
        if self.bucketed:
            for k in range(3):
                if self.use_graph:
                    self.g_stage[k].replay()
                else:
                    self._stage(k)
                self._reduce_bucket(k, boundary)
            if self.comm_stream is not None:
                torch.cuda.current_stream().wait_stream(self.comm_stream)
        elif self.use_graph:
            self.g_fb.replay()
        else:
            self._forward_backward()
        self.micro_idx += 1
        self.batch_idx += 1
        if boundary:
            if self.world > 1 and not self.bucketed:
                dp.allreduce_gradients(self.flat.grad, self.pg)
            if self.use_graph:
                self.g_opt.replay()
            else:
                self._optimizer()
            self.step_count += 1
            self._f_now = self._f_next
        if self.batch_idx >= self.batches_per_epoch:
            self.batch_idx = 0
            self.epoch += 1
        return self.out5

    def loss_dict(self) -> Dict[str, float]:
        v = self.out5.detach().cpu().tolist()
        return {"total_loss": v[0], "ce_loss": v[1], "token_kd_loss": v[2], "feature_kd_loss": v[3], "hidden_kd_loss": v[4]}

    # ------------------------------------------------------------------ checkpoint (reference .pth layout)
    def _optimizer_param_groups(self):
        """parameter lists in the reference optimizer's order (train_student_kd.py:219-234): all encoder params
        (frozen ones included, they simply never get state), decoder params, refinement + every projector's params."""
        other = list(self.student.attention_refinement.parameters()) if self.student.use_attention_refinement else []
        for pr in self.projectors.values():
            other.extend(list(pr.parameters()))
        return [list(self.student.encoder.parameters()), list(self.student.decoder.parameters()), other]

    def optimizer_state_dict(self) -> dict:
        """torch.optim.AdamW-compatible state dict rebuilt from the flat moment buffers, so that
        `AdamW(groups).load_state_dict(...)` of a reference-side tool accepts it."""
        where = {id(p): (o, n) for p, o, n in self.flat.metas}
        applied = self.applied_steps()
        state, groups, idx = {}, [], 0
        lr_mult = (0.1, 1.0, 1.0)
        for gi, plist in enumerate(self._optimizer_param_groups()):
            ids = []
            for p in plist:
                if id(p) in where and applied > 0:
                    o, n = where[id(p)]
                    shape4 = p.dim() == 4
                    view = (lambda flat: flat[o:o + n].view(p.shape[0], p.shape[2], p.shape[3], p.shape[1]).permute(0, 3, 1, 2)
                            if shape4 else flat[o:o + n].view(p.shape))
                    state[idx] = {"step": torch.tensor(float(applied)), "exp_avg": view(self.flat.exp_avg).clone(),
                                  "exp_avg_sq": view(self.flat.exp_avg_sq).clone()}
                ids.append(idx)
                idx += 1
            base = self.lr * lr_mult[gi]
            groups.append({"lr": self.eta_min + (base - self.eta_min) * self._f_now, "betas": self.betas, "eps": self.eps,
                           "weight_decay": self.wd, "amsgrad": False, "maximize": False, "foreach": None, "capturable": False,
                           "differentiable": False, "fused": None, "decoupled_weight_decay": True, "initial_lr": base,
                           "params": ids})
        return {"state": state, "param_groups": groups}

    def scheduler_state_dict(self) -> dict:
        ep = self.epoch + self.batch_idx / max(1, self.batches_per_epoch)
        t_cur, t_i = cosine_warm_restarts_state(ep)
        return {"T_0": 5, "T_i": t_i, "T_mult": 2, "eta_min": self.eta_min, "T_cur": t_cur, "last_epoch": ep,
                "base_lrs": [self.lr * 0.1, self.lr, self.lr]}

    def checkpoint(self, epoch: int, val_loss: float = float("nan"), val_bleu: float = float("nan"),
                   model_config: Optional[dict] = None) -> dict:
        """The dict the reference writes with torch.save (train_student_kd.py:359-380): same keys, so
        evaluate_student.py / a resumed reference run can read a checkpoint produced here."""
        s, L = self.student, self.loss
        mc = model_config or {"embed_size": s.embed_size, "hidden_size": s.hidden_size, "num_layers": s.decoder.num_layers,
                              "dropout": s.decoder.output_projection[2].p}
        contig = lambda sd: {k: (v.detach().contiguous().cpu() if torch.is_tensor(v) else v) for k, v in sd.items()}
        return {"epoch": epoch, "student_state_dict": contig(s.state_dict()),
                "projectors_state_dict": {k: contig(v.state_dict()) for k, v in self.projectors.items()},
                "optimizer_state_dict": self.optimizer_state_dict(), "scheduler_state_dict": self.scheduler_state_dict(),
                "val_loss": val_loss, "val_bleu": val_bleu, "vocab_size": s.vocab_size, "model_config": mc,
                "distillation_config": {"alpha": L.alpha, "beta": L.beta, "gamma": L.gamma, "temperature": L.temperature}}

    def save_checkpoint(self, path: str, epoch: int, **kw) -> None:
        torch.save(self.checkpoint(epoch, **kw), path)


def build_kd_models(vocab_size=5000, embed_size=256, hidden_size=512, num_layers=2, dropout=0.3, refine=True,
                    teacher_embed=512, teacher_heads=8, teacher_layers=4, teacher_dropout=0.15, device="cuda", seeds=(0, 1, 2)):
    """cfg3 of BASELINE.json: the student/teacher/projector triple with the values hard-coded at
    /root/reference/src/train_student_kd.py:98-101,161-167, key-seeded weights (no pretrained weights offline)."""
    from .distillation_utils import create_feature_projectors
    from .student_model import CaptioningStudent
    from .teacher_model import CaptioningTeacher
    from .utils.seeded_init import apply_seeded_init
    import contextlib
    import io
    student = CaptioningStudent(vocab_size, embed_size, hidden_size, num_layers, dropout, refine)
    teacher = CaptioningTeacher(vocab_size, teacher_embed, teacher_heads, teacher_layers, teacher_dropout)
    apply_seeded_init(student, seeds[0])
    apply_seeded_init(teacher, seeds[1])
    teacher.eval()
    with contextlib.redirect_stdout(io.StringIO()):
        projectors = create_feature_projectors(teacher, student)
    apply_seeded_init(projectors["encoder"], seeds[2])
    student.to(device)
    teacher.to(device)
    for k in projectors:
        projectors[k].to(device)
    return student, teacher, projectors
