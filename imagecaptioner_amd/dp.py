"""Data-parallel gradient exchange for the KD step: ONE all-reduce(SUM) of the flat fp32 gradient buffer per
optimizer step, one process per GPU, `torch.distributed` backend "nccl" (= RCCL over xGMI on ROCm; "gloo" in the
CPU tests).  The reference has no distributed code at all — this is the build-side addition.

The flat gradient buffer is laid out [layer3 | layer4 | projection | decoder | refinement | projector]; backward
produces it right to left.  KDTrainer cuts it into three buckets — everything above the trunk (~44 MB), layer4
(~60 MB), layer3 (~28 MB) — and issues each bucket's all-reduce on a side HIP stream as soon as its gradients
are complete, while the next stage of the trunk's backward runs; the payload is 120 MB: a direct
reduce-scatter + all-gather over the fully connected xGMI mesh moves S/8 per peer link per phase (~0.2 ms), a ring
~1.4 ms, against a ~28 ms step, so two of the three transfers hide completely.  1/world is NOT applied
here: it is folded into the fused clip+AdamW kernels (csrc/optim.hip inv_scale), which saves one pass over the buffer.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.distributed as dist


def world_size(group=None) -> int:
    return dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1


def allreduce_gradients(flat_grad: torch.Tensor, group=None, force: bool = False) -> torch.Tensor:
    """In-place SUM over ranks of (a bucket of) the flat gradient buffer (no-op for a single process unless `force`: an
    initialised world-1 group then still issues the collective — the single-GPU rehearsal of the staged step).  Every
    rank then holds the same sum, so the norm, the clip coefficient and the AdamW update are identical everywhere with
    no further communication (all-reduce BEFORE clipping)."""
    if world_size(group) > 1 or (force and dist.is_available() and dist.is_initialized()):
        dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=group)
    return flat_grad


def shard_batch(global_batch: int, rank: int, world: int):
    """Contiguous, equal shards of the global batch (512 -> 8 x 64): equal sizes are what make
    mean-over-global-batch == mean of per-rank means for the KL / MSE / cosine terms."""
    if global_batch % world != 0:
        raise ValueError(f"global batch {global_batch} is not divisible by world size {world}")
    per = global_batch // world
    return rank * per, (rank + 1) * per


def gradient_buckets(metas, total: int, names):
    """[(start, end)] element ranges of the flat gradient buffer in the order backward COMPLETES them: everything from
    the first non-trunk tensor to the end of the buffer, then layer4, then layer3.  `metas` = FlatParams.metas
    [(param, offset, numel)], `names` = {id(param): state_dict key}."""
    l4 = next((o for p, o, n in metas if names[id(p)].startswith("encoder.resnet.7.")), None)
    top = next((o for p, o, n in metas if not names[id(p)].startswith("encoder.resnet.")), total)
    if l4 is None or l4 == 0 or top <= l4:          # no trainable trunk (or an unexpected order): one bucket
        return [(0, total)]
    return [(top, total), (l4, top), (0, l4)]
