"""Key-seeded deterministic weights (SURVEY.md §8(c) "Weights on both sides").

No pretrained weights exist offline (the reference fetches them at
/root/reference/src/student_model.py:16 and /root/reference/src/teacher_model.py:36),
so parity is judged on synthetic weights that can be regenerated bit-identically
in the build container (loaded into the reference's modules) and on the GPU box
(loaded into this package's modules) from nothing but the tensor NAME, SHAPE and
a seed.  Every tensor gets its own generator seeded with crc32(name) ^ seed, so
the values do not depend on the order or the set of other tensors.
"""
from __future__ import annotations

import math
import zlib
from typing import Dict, Iterable, Tuple

import torch


def _gen(name: str, seed: int) -> torch.Generator:
    g = torch.Generator(device="cpu")
    g.manual_seed((zlib.crc32(name.encode("utf-8")) ^ (seed & 0xFFFFFFFF)) & 0x7FFFFFFF)
    return g


def _is_norm_scale(name: str, shape: Tuple[int, ...]) -> bool:
    if len(shape) != 1 or not name.endswith("weight"):
        return False
    # BatchNorm (bnN / downsample.1 / resnet.1) and LayerNorm (norm*, *.3 of Sequential(Linear,ReLU,Dropout,LN))
    return True


def seeded_tensor(name: str, shape: Tuple[int, ...], seed: int = 0, dtype=torch.float32) -> torch.Tensor | None:
    """Deterministic value for one state_dict entry; None = keep what the module has
    (integer counters and the sinusoid table, which are not random)."""
    leaf = name.rsplit(".", 1)[-1]
    if leaf == "num_batches_tracked" or leaf == "pe":
        return None
    g = _gen(name, seed)
    shape = tuple(shape)
    if leaf == "running_var":
        return torch.rand(shape, generator=g, dtype=dtype) + 0.5
    if leaf == "running_mean":
        return torch.randn(shape, generator=g, dtype=dtype) * 0.1
    if leaf in ("cls_token", "pos_embed"):
        return torch.randn(shape, generator=g, dtype=dtype) * 0.02
    if "embedding" in name and len(shape) == 2 and leaf == "weight":
        return (torch.rand(shape, generator=g, dtype=dtype) * 2 - 1) * 0.1
    if len(shape) == 1:
        if leaf == "weight":  # BN / LN scale
            return torch.rand(shape, generator=g, dtype=dtype) * 0.5 + 0.75
        return torch.randn(shape, generator=g, dtype=dtype) * 0.05  # every bias
    # matrices / conv kernels: fan-in scaled; He gain so ReLU stacks keep their scale
    fan_in = 1
    for d in shape[1:]:
        fan_in *= d
    gain = math.sqrt(2.0) if len(shape) == 4 else 1.0
    if "lstm" in name:
        bound = 1.0 / math.sqrt(shape[1] if "weight_hh" in name else shape[0] // 4)
        return (torch.rand(shape, generator=g, dtype=dtype) * 2 - 1) * bound
    # vocabulary heads get a larger scale so that argmax margins are far above
    # fp32 reorder noise (SURVEY.md §7 "Argmax bit-exactness")
    if name.endswith("output_projection.3.weight") or name.endswith("fc_out.weight"):
        gain = 4.0
    return torch.randn(shape, generator=g, dtype=dtype) * (gain / math.sqrt(fan_in))


def seeded_state_dict(shapes: Dict[str, Tuple[int, ...]], seed: int = 0) -> Dict[str, torch.Tensor]:
    out = {}
    for name in sorted(shapes):
        t = seeded_tensor(name, shapes[name], seed)
        if t is not None:
            out[name] = t
    return out


@torch.no_grad()
def apply_seeded_init(module: torch.nn.Module, seed: int = 0, prefix: str = "") -> torch.nn.Module:
    """Overwrite every parameter/buffer of `module` in place with its key-seeded value."""
    sd = module.state_dict()
    for name in sorted(sd):
        t = seeded_tensor(prefix + name, tuple(sd[name].shape), seed)
        if t is None:
            continue
        sd[name].copy_(t.to(sd[name].dtype))
    return module


def synthetic_batch(batch: int, vocab: int = 5000, t_plus_1: int = 16, seed: int = 1234,
                    rank: int = 0) -> Tuple[torch.Tensor, torch.Tensor]:
    """Synthetic inputs of SURVEY.md §8(d): images ~ N(0,1) (B,3,224,224) f32 and captions
    (T+1,B) int64 = <START>=1, random ids in [4,V), <END>=2, <PAD>=0 tail, len ~ U{8..16}."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed + rank)
    images = torch.randn(batch, 3, 224, 224, generator=g, dtype=torch.float32)
    lens = torch.randint(8, t_plus_1 + 1, (batch,), generator=g)
    body = torch.randint(4, vocab, (t_plus_1, batch), generator=g, dtype=torch.int64)
    caps = torch.zeros(t_plus_1, batch, dtype=torch.int64)
    for b in range(batch):
        n = int(lens[b])
        caps[:n, b] = body[:n, b]
        caps[0, b] = 1
        caps[n - 1, b] = 2
    return images, caps


def named_shapes(module: torch.nn.Module) -> Iterable[Tuple[str, Tuple[int, ...]]]:
    for k, v in module.state_dict().items():
        yield k, tuple(v.shape)
