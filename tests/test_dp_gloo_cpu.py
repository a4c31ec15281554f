"""world_size-2 `gloo` test of the data-parallel gradient exchange (imagecaptioner_amd/dp.py + FlatParams):
each rank runs the KD loss + backward of the ORACLE on its own shard (per-shard train-mode BatchNorm), packs the
gradients into the flat buffer exactly as the trainer does, all-reduces, and the result / world must equal the
average of the per-shard gradients computed serially in one process — the equivalence the 8-GPU step relies on
(SUM/world of per-rank gradients is exact for the KL / MSE terms with equal shards)."""
import os
import socket
import tempfile

import pytest
import torch
import torch.multiprocessing as mp

from imagecaptioner_amd.utils.seeded_init import seeded_state_dict, synthetic_batch
from oracle import restatement as R

V, E, H = 300, 128, 256


def _shard_grads(rank: int):
    """oracle KD gradients of a small student on shard `rank` (B=2 per shard, synthetic teacher outputs)."""
    torch.manual_seed(0)
    torch.set_num_threads(2)
    shapes = R.student_state_shapes(V, E, H, 1, False)
    trainable = lambda k: not any(k.startswith(f"encoder.resnet.{i}.") for i in (0, 1, 4, 5)) and "running_" not in k
    sd = {k: (v.clone().requires_grad_(True) if trainable(k) else v.clone()) for k, v in seeded_state_dict(shapes, seed=0).items()}
    images, caps = synthetic_batch(2, V, 8, seed=99, rank=rank)
    g = torch.Generator().manual_seed(1000 + rank)
    t_logits = torch.randn(7, 2, V, generator=g) * 2
    t_feats = torch.randn(2, 49, E, generator=g)
    logits, enc, hids, _ = R.student_forward(sd, images, caps[:-1], hidden=H, layers=1, refine=False, train=True)
    loss, _ = R.distillation_loss({"logits": logits, "encoder_features": enc, "hidden_states": hids},
                                  {"logits": t_logits, "encoder_features": t_feats, "hidden_states": None}, caps[1:],
                                  alpha=0.7, beta=0.3, gamma=0.0)        # CE weight exactly 0: see test docstring
    loss.backward()
    return {k: v.grad for k, v in sd.items() if v.requires_grad}, {k: tuple(v.shape) for k, v in sd.items() if v.requires_grad}


def _flat_from(grads, shapes):
    from imagecaptioner_amd.train_student_kd import FlatParams
    params = [(k, torch.nn.Parameter(torch.zeros(shapes[k]).contiguous(memory_format=torch.channels_last)
                                     if len(shapes[k]) == 4 else torch.zeros(shapes[k])))
              # module order of CNNEncoder.parameters(): the ResNet children first, then the projection
              for k in sorted(shapes, key=lambda k: (not k.startswith("encoder.resnet."), k))]
    enc = [p for k, p in params if k.startswith("encoder.")]
    dec = [p for k, p in params if k.startswith("decoder.")]
    fp = FlatParams([("encoder", enc), ("decoder", dec), ("refine", []), ("projector", [])], torch.device("cpu"))
    with torch.no_grad():
        for k, p in params:
            p.grad.copy_(grads[k])
    return fp, dict(params)


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    from imagecaptioner_amd import dp
    grads, shapes = _shard_grads(rank)
    fp, _ = _flat_from(grads, shapes)
    assert dp.world_size() == world
    # the trainer's three buckets (above-the-trunk, layer4, layer3), reduced one by one in completion order, must cover the
    # buffer exactly once: reducing them == reducing the whole buffer
    fp, params = _flat_from(grads, shapes)
    names = {id(p): k for k, p in params.items()}
    buckets = dp.gradient_buckets(fp.metas, fp.total, names)
    assert len(buckets) == 3 and buckets[0][1] == fp.total and buckets[2][0] == 0
    assert buckets[0][0] == buckets[1][1] and buckets[1][0] == buckets[2][1]
    l4 = [o for p_, o, n in fp.metas if names[id(p_)].startswith("encoder.resnet.7.")]
    l3 = [o for p_, o, n in fp.metas if names[id(p_)].startswith("encoder.resnet.6.")]
    assert min(l4) == buckets[1][0] and max(l3) < buckets[1][0] and max(l4) < buckets[0][0]
    for a, b in buckets:
        dp.allreduce_gradients(fp.grad[a:b])
    if rank == 0:
        torch.save(fp.grad / world, out)
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_allreduce_of_flat_gradients_equals_average_of_shards():
    world = 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = os.path.join(tempfile.mkdtemp(), "avg.pt")
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    got = torch.load(out, weights_only=True)
    g0, shapes = _shard_grads(0)
    g1, _ = _shard_grads(1)
    fp, _ = _flat_from({k: (g0[k] + g1[k]) / 2 for k in g0}, shapes)
    assert got.shape == fp.grad.shape
    assert torch.allclose(got, fp.grad, rtol=1e-6, atol=1e-9)
    assert float(got.abs().sum()) > 0


def test_shard_batch():
    from imagecaptioner_amd.dp import shard_batch
    assert [shard_batch(512, r, 8) for r in (0, 7)] == [(0, 64), (448, 512)]
    with pytest.raises(ValueError):
        shard_batch(100, 0, 8)
