"""TEST INFRASTRUCTURE — the oracle.  Never imported by the product path
(only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it).

CPU restatement (plain PyTorch fp32 primitives, functional over a state_dict)
of the reference's knowledge-distillation hot path, SURVEY.md §8(a) rows A1-A14.
It exists because the reference's Python cannot travel to the GPU box; it is
PINNED in the build container against golden vectors captured by importing the
reference's own modules (oracle/make_goldens.py -> tests/golden/*.npz,
tests/test_oracle_vs_golden.py).  Each function cites the reference lines it
restates.  Third-party arithmetic (torchvision resnet50, timm ViT-S/16; both
absent from the image and un-pinned by the reference) is restated from the
published architectures: parity at that boundary is "unpinned" by the reference
itself and anchored on torch CPU primitives.

All tensors are torch CPU fp32; `sd` maps the reference's state_dict key names
to tensors (leaf tensors with requires_grad where a gradient is wanted).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]

RESNET_STAGES = ((4, 64, 3, 1), (5, 128, 4, 2), (6, 256, 6, 2), (7, 512, 3, 2))  # (child idx, planes, blocks, stride)


# ----------------------------------------------------------------------------- helpers
def _bn(sd: SD, p: str, x: torch.Tensor, train: bool) -> torch.Tensor:
    # nn.BatchNorm2d defaults eps=1e-5, momentum=0.1; train mode => batch statistics and a
    # running-stat update even for the "frozen" stem
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"],
                        training=train, momentum=0.1, eps=1e-5)


def _drop(x: torch.Tensor, p: float, train: bool) -> torch.Tensor:
    return F.dropout(x, p, training=True) if (train and p > 0.0) else x


def _lin(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    return F.linear(x, sd[p + ".weight"], sd.get(p + ".bias"))


def _ln(sd: SD, p: str, x: torch.Tensor, eps: float = 1e-5) -> torch.Tensor:
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], eps)


def _mha(sd: SD, p: str, q_in: torch.Tensor, kv_in: torch.Tensor, heads: int,
         causal: bool = False) -> torch.Tensor:
    """nn.MultiheadAttention forward (packed in_proj), batch-first tensors (B,L,E)."""
    E = q_in.shape[-1]
    w, b = sd[p + ".in_proj_weight"], sd[p + ".in_proj_bias"]
    q = F.linear(q_in, w[:E], b[:E])
    k = F.linear(kv_in, w[E:2 * E], b[E:2 * E])
    v = F.linear(kv_in, w[2 * E:], b[2 * E:])
    B, Lq, _ = q.shape
    Lk = k.shape[1]
    d = E // heads
    q = q.view(B, Lq, heads, d).transpose(1, 2)
    k = k.view(B, Lk, heads, d).transpose(1, 2)
    v = v.view(B, Lk, heads, d).transpose(1, 2)
    s = (q @ k.transpose(-1, -2)) / math.sqrt(d)
    if causal:
        s = s.masked_fill(torch.ones(Lq, Lk, dtype=torch.bool).triu(1), float("-inf"))
    o = torch.softmax(s, dim=-1) @ v
    o = o.transpose(1, 2).reshape(B, Lq, E)
    return F.linear(o, sd[p + ".out_proj.weight"], sd[p + ".out_proj.bias"])


# ----------------------------------------------------------------------------- A1: CNN encoder
def resnet50_trunk(sd: SD, p: str, x: torch.Tensor, train: bool) -> torch.Tensor:
    """torchvision resnet50 children [:-2] as the reference slices them
    (/root/reference/src/student_model.py:16-20,57): indices 0 conv1, 1 bn1, 2 relu,
    3 maxpool, 4-7 layer1-4."""
    x = F.conv2d(x, sd[p + ".0.weight"], None, stride=2, padding=3)
    x = F.relu(_bn(sd, p + ".1", x, train))
    x = F.max_pool2d(x, 3, 2, 1)
    for child, planes, blocks, stride in RESNET_STAGES:
        for i in range(blocks):
            q = f"{p}.{child}.{i}"
            s = stride if i == 0 else 1
            idt = x
            if (q + ".downsample.0.weight") in sd:
                idt = _bn(sd, q + ".downsample.1", F.conv2d(x, sd[q + ".downsample.0.weight"], None, stride=s), train)
            y = F.relu(_bn(sd, q + ".bn1", F.conv2d(x, sd[q + ".conv1.weight"]), train))
            y = F.relu(_bn(sd, q + ".bn2", F.conv2d(y, sd[q + ".conv2.weight"], None, stride=s, padding=1), train))
            y = _bn(sd, q + ".bn3", F.conv2d(y, sd[q + ".conv3.weight"]), train)
            x = F.relu(y + idt)
    return x


def cnn_encoder(sd: SD, images: torch.Tensor, train: bool, p_drop: float = 0.0, p: str = "encoder") -> torch.Tensor:
    """CNNEncoder.forward, /root/reference/src/student_model.py:46-69."""
    f = resnet50_trunk(sd, p + ".resnet", images, train)            # (B,2048,7,7)
    f = F.adaptive_avg_pool2d(f, (7, 7))                            # identity at 224 input (:60)
    B = f.shape[0]
    f = f.view(B, 2048, -1).permute(0, 2, 1)                        # (B,49,2048) (:63-64)
    f = F.relu(_lin(sd, p + ".projection.0", f))
    f = _drop(f, p_drop, train)
    return _ln(sd, p + ".projection.3", f)


# ----------------------------------------------------------------------------- A2: refinement
def attention_refinement(sd: SD, x: torch.Tensor, train: bool = False, p_drop: float = 0.0,
                         p: str = "attention_refinement", heads: int = 4) -> torch.Tensor:
    """AttentionRefinement.forward, /root/reference/src/student_model.py:102-118 (post-LN).
    p_drop restates only the FFN dropout; MHA-internal dropout is exercised with p=0."""
    x = _ln(sd, p + ".norm1", x + _mha(sd, p + ".attention", x, x, heads))
    h = _drop(F.relu(_lin(sd, p + ".ffn.0", x)), p_drop, train)
    return _ln(sd, p + ".norm2", x + _lin(sd, p + ".ffn.3", h))


# ----------------------------------------------------------------------------- A3/A4: decoder
def spatial_attention(sd: SD, h_top: torch.Tensor, feats: torch.Tensor, p: str = "decoder"):
    """LSTMDecoder.attention_mechanism, /root/reference/src/student_model.py:173-203:
    scores_j = sum_e tanh(W_a [h ; f_j] + b_a)_e  (no learned v), softmax over j, ctx = sum_j w_j f_j."""
    B, L, E = feats.shape
    H = h_top.shape[1]
    cat = torch.cat([h_top.unsqueeze(1).expand(B, L, H), feats], dim=2)
    s = torch.tanh(_lin(sd, p + ".attention", cat)).sum(dim=2)
    w = torch.softmax(s, dim=1)
    ctx = torch.bmm(w.unsqueeze(1), feats).squeeze(1)
    return ctx, w


def lstm_step(sd: SD, x: torch.Tensor, h: List[torch.Tensor], c: List[torch.Tensor], layers: int,
              train: bool = False, p_drop: float = 0.0, p: str = "decoder.lstm"):
    """One nn.LSTM time step, gate order i,f,g,o, two biases, inter-layer dropout
    (/root/reference/src/student_model.py:142-148, :244)."""
    nh, nc = [], []
    inp = x
    for l in range(layers):
        g = (F.linear(inp, sd[f"{p}.weight_ih_l{l}"], sd[f"{p}.bias_ih_l{l}"])
             + F.linear(h[l], sd[f"{p}.weight_hh_l{l}"], sd[f"{p}.bias_hh_l{l}"]))
        i, f, gg, o = g.chunk(4, dim=1)
        cl = torch.sigmoid(f) * c[l] + torch.sigmoid(i) * torch.tanh(gg)
        hl = torch.sigmoid(o) * torch.tanh(cl)
        nh.append(hl)
        nc.append(cl)
        inp = hl if l == layers - 1 else _drop(hl, p_drop, train)
    return nh, nc


def output_projection(sd: SD, h_top: torch.Tensor, train: bool = False, p_drop: float = 0.0,
                      p: str = "decoder.output_projection") -> torch.Tensor:
    """Linear(H,E)+ReLU+Dropout+Linear(E,V), /root/reference/src/student_model.py:151-156."""
    return _lin(sd, p + ".3", _drop(F.relu(_lin(sd, p + ".0", h_top)), p_drop, train))


def lstm_decoder(sd: SD, feats: torch.Tensor, captions: torch.Tensor, layers: int, hidden: int,
                 train: bool = False, p_drop: float = 0.0, p: str = "decoder", init=None):
    """LSTMDecoder.forward (teacher forcing), /root/reference/src/student_model.py:205-256.  init = (h0, c0), each
    (layers, B, H): the caller-supplied initial state of :205,:220 (None = init_hidden's zeros)."""
    T, B = captions.shape
    h = [feats.new_zeros(B, hidden) for _ in range(layers)] if init is None else list(init[0].unbind(0))
    c = [feats.new_zeros(B, hidden) for _ in range(layers)] if init is None else list(init[1].unbind(0))
    emb = F.embedding(captions, sd[p + ".embedding.weight"])        # (T,B,E)
    outs, hids, attw = [], [], []
    for t in range(T):
        ctx, w = spatial_attention(sd, h[-1], feats, p)             # top-layer h of previous step (:237)
        x = _lin(sd, p + ".attention_combine", torch.cat([emb[t], ctx], dim=1))
        h, c = lstm_step(sd, x, h, c, layers, train, p_drop, p + ".lstm")
        outs.append(output_projection(sd, h[-1], train, p_drop, p + ".output_projection"))
        hids.append(h[-1])
        attw.append(w)
    return torch.stack(outs, 0), hids, attw


# ----------------------------------------------------------------------------- N4: compact student (MobileNetV2 + dot attention)
MBV2_SETTINGS = ((1, 16, 1, 1), (6, 24, 2, 2), (6, 32, 3, 2), (6, 64, 4, 2), (6, 96, 3, 1), (6, 160, 3, 2), (6, 320, 1, 1))


def mobilenet_v2_features(sd: SD, p: str, x: torch.Tensor, train: bool) -> torch.Tensor:
    """torchvision mobilenet_v2().features as the reference uses it (/root/reference/src/student_model_compact.py:19-22,51):
    19 modules, ReLU6, train-mode BatchNorm in frozen and trainable modules alike (the reference never calls .eval() on them)."""
    def cna(q, t, stride=1, groups=1, pad=0):
        t = F.conv2d(t, sd[q + ".0.weight"], None, stride=stride, padding=pad, groups=groups)
        return F.relu6(_bn(sd, q + ".1", t, train))
    x = cna(p + ".0", x, 2, 1, 1)
    idx, cin = 1, 32
    for t, c, n, s in MBV2_SETTINGS:
        for i in range(n):
            q = f"{p}.{idx}.conv"
            stride = s if i == 0 else 1
            y, j = x, 0
            if t != 1:
                y = cna(f"{q}.0", y)
                j = 1
            y = cna(f"{q}.{j}", y, stride, cin * t, 1)
            y = _bn(sd, f"{q}.{j + 2}", F.conv2d(y, sd[f"{q}.{j + 1}.weight"]), train)
            x = x + y if (stride == 1 and cin == c) else y
            cin, idx = c, idx + 1
    return cna(f"{p}.18", x)


def compact_encoder(sd: SD, images: torch.Tensor, train: bool, p: str = "encoder") -> torch.Tensor:
    """CompactCNNEncoder.forward, /root/reference/src/student_model_compact.py:41-64 (dropout p = 0 / eval)."""
    f = mobilenet_v2_features(sd, p + ".backbone", images, train)
    f = F.adaptive_avg_pool2d(f, (7, 7))
    B = f.shape[0]
    f = f.view(B, 1280, -1).permute(0, 2, 1)
    return F.relu(_lin(sd, p + ".projection.0", f))


def compact_decoder(sd: SD, feats: torch.Tensor, captions: torch.Tensor, hidden: int, p: str = "decoder"):
    """CompactLSTMDecoder.forward, /root/reference/src/student_model_compact.py:114-195: dot-product attention
    scores_j = <W_a h + b_a, f_j>, context added to the word embedding, one LSTM layer, Linear(H, V)."""
    T, B = captions.shape
    h = [feats.new_zeros(B, hidden)]
    c = [feats.new_zeros(B, hidden)]
    emb = F.embedding(captions, sd[p + ".embedding.weight"])
    outs, hids, attw = [], [], []
    for t in range(T):
        hp = _lin(sd, p + ".attention", h[-1])
        w = torch.softmax(torch.bmm(hp.unsqueeze(1), feats.transpose(1, 2)).squeeze(1), dim=1)
        ctx = torch.bmm(w.unsqueeze(1), feats).squeeze(1)
        h, c = lstm_step(sd, emb[t] + ctx, h, c, 1, False, 0.0, p + ".lstm")
        outs.append(_lin(sd, p + ".output_projection", h[-1]))
        hids.append(h[-1])
        attw.append(w)
    return torch.stack(outs, 0), hids, attw


def compact_student_forward(sd: SD, images: torch.Tensor, captions: torch.Tensor, *, hidden: int, train: bool = False):
    """CompactCaptioningStudent.forward without the optional refinement, /root/reference/src/student_model_compact.py:233-262."""
    enc = compact_encoder(sd, images, train)
    logits, hids, attw = compact_decoder(sd, enc, captions, hidden)
    return logits, enc, hids, attw


def compact_state_shapes(vocab: int, embed: int, hidden: int) -> Dict[str, Tuple[int, ...]]:
    s: Dict[str, Tuple[int, ...]] = {}

    def bn(q, c):
        s[q + ".weight"] = (c,); s[q + ".bias"] = (c,); s[q + ".running_mean"] = (c,); s[q + ".running_var"] = (c,)
    p = "encoder.backbone"
    s[p + ".0.0.weight"] = (32, 3, 3, 3); bn(p + ".0.1", 32)
    idx, cin = 1, 32
    for t, c, n, st in MBV2_SETTINGS:
        for i in range(n):
            q, j, hid = f"{p}.{idx}.conv", 0, cin * t
            if t != 1:
                s[f"{q}.0.0.weight"] = (hid, cin, 1, 1); bn(f"{q}.0.1", hid)
                j = 1
            s[f"{q}.{j}.0.weight"] = (hid, 1, 3, 3); bn(f"{q}.{j}.1", hid)
            s[f"{q}.{j + 1}.weight"] = (c, hid, 1, 1); bn(f"{q}.{j + 2}", c)
            cin, idx = c, idx + 1
    s[p + ".18.0.weight"] = (1280, 320, 1, 1); bn(p + ".18.1", 1280)
    s["encoder.projection.0.weight"] = (embed, 1280); s["encoder.projection.0.bias"] = (embed,)
    s["decoder.embedding.weight"] = (vocab, embed)
    s["decoder.attention.weight"] = (embed, hidden); s["decoder.attention.bias"] = (embed,)
    s["decoder.lstm.weight_ih_l0"] = (4 * hidden, embed); s["decoder.lstm.weight_hh_l0"] = (4 * hidden, hidden)
    s["decoder.lstm.bias_ih_l0"] = (4 * hidden,); s["decoder.lstm.bias_hh_l0"] = (4 * hidden,)
    s["decoder.output_projection.weight"] = (vocab, hidden); s["decoder.output_projection.bias"] = (vocab,)
    return s


# ----------------------------------------------------------------------------- A5: student
def student_forward(sd: SD, images: torch.Tensor, captions: torch.Tensor, *, hidden: int, layers: int,
                    refine: bool, train: bool = False, p_drop: float = 0.0):
    """CaptioningStudent.forward, /root/reference/src/student_model.py:288-312.  Returns the
    UN-refined encoder features for KD while the decoder consumes the refined ones (:301,:305,:312).
    `p_drop` is the single knob for parity runs (0.0); the reference's distinct rates only
    matter for throughput runs, where masks cannot match torch's Philox stream anyway."""
    enc = cnn_encoder(sd, images, train, 0.2 if p_drop > 0 else 0.0)
    ref = attention_refinement(sd, enc, train, 0.1 if p_drop > 0 else 0.0) if refine else enc
    logits, hids, attw = lstm_decoder(sd, ref, captions, layers, hidden, train, p_drop)
    return logits, enc, hids, attw


def greedy_decode(sd: SD, images: torch.Tensor, *, hidden: int, layers: int, refine: bool,
                  max_length: int = 20, temperature: float = 1.0, start_id: int = 1, end_id: int = 2):
    """Batched equivalent of CaptioningStudent.caption_image (greedy),
    /root/reference/src/student_model.py:314-381: per row the ids the B=1 reference emits;
    rows continue independently after another row hits <END>.
    Returns (ids (max_length,B) with -1 after a row's <END>, logits (max_length,B,V))."""
    with torch.no_grad():
        enc = cnn_encoder(sd, images, False)
        feats = attention_refinement(sd, enc) if refine else enc
        B = images.shape[0]
        h = [feats.new_zeros(B, hidden) for _ in range(layers)]
        c = [feats.new_zeros(B, hidden) for _ in range(layers)]
        tok = torch.full((B,), start_id, dtype=torch.long)
        alive = torch.ones(B, dtype=torch.bool)
        ids = torch.full((max_length, B), -1, dtype=torch.long)
        all_logits = []
        for t in range(max_length):
            emb = F.embedding(tok, sd["decoder.embedding.weight"])
            ctx, _ = spatial_attention(sd, h[-1], feats)
            x = _lin(sd, "decoder.attention_combine", torch.cat([emb, ctx], dim=1))
            h, c = lstm_step(sd, x, h, c, layers)
            lg = output_projection(sd, h[-1])
            if temperature != 1.0:
                lg = lg / temperature
            all_logits.append(lg)
            nxt = lg.argmax(dim=1)
            alive = alive & (nxt != end_id)
            ids[t] = torch.where(alive, nxt, torch.full_like(nxt, -1))
            tok = nxt
        return ids, torch.stack(all_logits, 0)


# ----------------------------------------------------------------------------- A6/A7: teacher
def vit_small_features(sd: SD, p: str, images: torch.Tensor, heads: int = 6, depth: int = 12) -> torch.Tensor:
    """timm vit_small_patch16_224.forward_features (all 197 normed tokens); call sites
    /root/reference/src/teacher_model.py:82 and /root/reference/src/distillation_utils.py:281."""
    x = F.conv2d(images, sd[p + ".patch_embed.proj.weight"], sd[p + ".patch_embed.proj.bias"], stride=16)
    x = x.flatten(2).transpose(1, 2)
    x = torch.cat([sd[p + ".cls_token"].expand(x.shape[0], -1, -1), x], dim=1) + sd[p + ".pos_embed"]
    B, N, C = x.shape
    d = C // heads
    for i in range(depth):
        q = f"{p}.blocks.{i}"
        y = _ln(sd, q + ".norm1", x, 1e-6)
        qkv = _lin(sd, q + ".attn.qkv", y).view(B, N, 3, heads, d).permute(2, 0, 3, 1, 4)
        s = (qkv[0] @ qkv[1].transpose(-1, -2)) / math.sqrt(d)
        o = (torch.softmax(s, -1) @ qkv[2]).transpose(1, 2).reshape(B, N, C)
        x = x + _lin(sd, q + ".attn.proj", o)
        y = _ln(sd, q + ".norm2", x, 1e-6)
        x = x + _lin(sd, q + ".mlp.fc2", F.gelu(_lin(sd, q + ".mlp.fc1", y)))
    return _ln(sd, p + ".norm", x, 1e-6)


def sinusoid_pe(T: int, d: int) -> torch.Tensor:
    """PositionalEncoding buffer rows [0,T), /root/reference/src/teacher_model.py:17-23."""
    pe = torch.zeros(T, d)
    pos = torch.arange(0, T, dtype=torch.float).unsqueeze(1)
    div = torch.exp(torch.arange(0, d, 2).float() * (-math.log(10000.0) / d))
    pe[:, 0::2] = torch.sin(pos * div)
    pe[:, 1::2] = torch.cos(pos * div)
    return pe


def teacher_decoder(sd: SD, memory: torch.Tensor, captions: torch.Tensor, heads: int, layers: int) -> torch.Tensor:
    """Embedding + sinusoid PE + N post-norm nn.TransformerDecoderLayer (ReLU, FFN 2E) with a
    causal mask + pre_output_norm + fc_out, eval mode (dropout off),
    /root/reference/src/teacher_model.py:86-104.  memory (B,197,E) batch-first here."""
    T, B = captions.shape
    E = memory.shape[-1]
    x = F.embedding(captions, sd["embedding.weight"]) + sinusoid_pe(T, E).unsqueeze(1)   # (T,B,E)
    x = x.transpose(0, 1)                                                                 # (B,T,E)
    for i in range(layers):
        q = f"decoder.layers.{i}"
        x = _ln(sd, q + ".norm1", x + _mha(sd, q + ".self_attn", x, x, heads, causal=True))
        x = _ln(sd, q + ".norm2", x + _mha(sd, q + ".multihead_attn", x, memory, heads))
        x = _ln(sd, q + ".norm3", x + _lin(sd, q + ".linear2", F.relu(_lin(sd, q + ".linear1", x))))
    x = _ln(sd, "pre_output_norm", x)
    return _lin(sd, "fc_out", x).transpose(0, 1).contiguous()                            # (T,B,V)


def teacher_forward(sd: SD, images: torch.Tensor, captions: torch.Tensor, *, heads: int, layers: int):
    """TeacherWrapper.forward, /root/reference/src/distillation_utils.py:268-292, with the
    duplicate ViT pass removed (both passes are bit-identical in eval/no_grad).
    Returns (logits (T,B,V), encoder_features (B,197,E))."""
    with torch.no_grad():
        f = vit_small_features(sd, "encoder", images)
        if "encoder_projection.weight" in sd:
            f = _lin(sd, "encoder_projection", f)
        return teacher_decoder(sd, f, captions, heads, layers), f


# ----------------------------------------------------------------------------- A8: projector
def adaptive_avg_pool_tokens(x: torch.Tensor, out_len: int) -> torch.Tensor:
    """nn.AdaptiveAvgPool1d along the token axis of (B,L,E): bin i = [floor(i*L/out), ceil((i+1)*L/out))."""
    B, L, E = x.shape
    cols = []
    for i in range(out_len):
        a = (i * L) // out_len
        b = -((-(i + 1) * L) // out_len)
        cols.append(x[:, a:b].mean(dim=1))
    return torch.stack(cols, dim=1)


def feature_projector(sd: SD, feats: torch.Tensor, out_len: int, train: bool = False, p_drop: float = 0.0,
                      p: str = "feature_projection") -> torch.Tensor:
    """FeatureProjector.forward, /root/reference/src/distillation_utils.py:233-252."""
    y = feats
    if (p + ".0.weight") in sd:
        y = _ln(sd, p + ".3", _drop(F.relu(_lin(sd, p + ".0", feats)), p_drop, train))
    if y.shape[1] != out_len:
        y = adaptive_avg_pool_tokens(y, out_len)
    return y


# ----------------------------------------------------------------------------- A9-A12: losses
def token_kd(s_logits: torch.Tensor, t_logits: torch.Tensor, tau: float) -> torch.Tensor:
    """/root/reference/src/distillation_utils.py:30-54: KL(batchmean over T*B rows) * tau^2."""
    V = s_logits.shape[-1]
    ls = F.log_softmax(s_logits.reshape(-1, V) / tau, dim=1)
    lt = F.log_softmax(t_logits.reshape(-1, V) / tau, dim=1)
    pt = lt.exp()
    return (pt * (lt - ls)).sum() / ls.shape[0] * (tau * tau)


def feature_kd(s: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
    """/root/reference/src/distillation_utils.py:56-94."""
    if s.shape[-1] != t.shape[-1]:
        raise ValueError(f"Feature dimensions don't match: student {s.shape[-1]}, teacher {t.shape[-1]}")
    g = F.mse_loss(s.mean(dim=1), t.mean(dim=1))
    ws = torch.softmax(s.sum(dim=-1), dim=1).unsqueeze(-1)
    wt = torch.softmax(t.sum(dim=-1), dim=1).unsqueeze(-1)
    a = F.mse_loss((s * ws).sum(dim=1), (t * wt).sum(dim=1))
    return 0.6 * g + 0.4 * a


def hidden_kd(s_list: Optional[List[torch.Tensor]], t_list: Optional[List[torch.Tensor]]) -> torch.Tensor:
    """/root/reference/src/distillation_utils.py:96-136: per step 0.7*MSE + 0.3*mean_b(1-cos), mean over steps
    (nn.CosineEmbeddingLoss target=+1: 1 - s.t / sqrt((|s|^2+eps)(|t|^2+eps)), eps=1e-8 (ATen EPSILON))."""
    if s_list is None or t_list is None:
        return torch.tensor(0.0)
    n = min(len(s_list), len(t_list))
    terms = []
    for s, t in zip(s_list[:n], t_list[:n]):
        if s.shape[-1] != t.shape[-1]:
            raise ValueError(f"Hidden dimensions don't match: student {s.shape[-1]}, teacher {t.shape[-1]}")
        cos = (s * t).sum(1) / torch.sqrt(((s * s).sum(1) + 1e-8) * ((t * t).sum(1) + 1e-8))
        terms.append(0.7 * F.mse_loss(s, t) + 0.3 * (1.0 - cos).mean())
    return torch.stack(terms).mean()


def distillation_loss(s_out: Dict, t_out: Dict, targets: torch.Tensor, *, alpha=0.7, beta=0.2, gamma=0.1,
                      tau=4.0) -> Tuple[torch.Tensor, Dict[str, float]]:
    """DistillationLoss.forward, /root/reference/src/distillation_utils.py:138-200."""
    s_logits, t_logits = s_out["logits"], t_out["logits"]
    V = s_logits.shape[-1]
    ce = F.cross_entropy(s_logits.reshape(-1, V), targets.reshape(-1), ignore_index=0)
    kd = token_kd(s_logits, t_logits, tau)
    feat = torch.tensor(0.0)
    if "encoder_features" in s_out and "encoder_features" in t_out:
        feat = feature_kd(s_out["encoder_features"], t_out["encoder_features"])
    hid = torch.tensor(0.0)
    if "hidden_states" in s_out and "hidden_states" in t_out:
        hid = hidden_kd(s_out["hidden_states"], t_out["hidden_states"])
    total = (1 - alpha - beta - gamma) * ce + alpha * kd + beta * feat + gamma * hid
    return total, {"total_loss": float(total.detach()), "ce_loss": float(ce.detach()), "token_kd_loss": float(kd.detach()),
                   "feature_kd_loss": float(feat.detach()), "hidden_kd_loss": float(hid.detach())}


def optimized_distillation_loss(s_out: Dict, t_out: Dict, targets: torch.Tensor, *, epoch: float = 0, alpha=0.7, beta=0.2,
                                gamma=0.1, tau=4.0, focal_alpha=0.25, focal_gamma=2.0, warmup_epochs=3,
                                attention_weights: Optional[torch.Tensor] = None):
    """OptimizedDistillationLoss.forward, /root/reference/src/train_student_kd_optimized.py:58-128:
    warm-up-scheduled weights (:63-66), soft-target cross entropy x tau^2 (:73-77), focal loss over ALL rows (plain
    CrossEntropyLoss, PAD included, :51-56,:80), per-token cosine feature loss (:84-94), hidden term = MSE of
    attention-weighted sums over time (:101-110; the reference draws the weights with torch.randn — here they are an
    argument, softmaxed over time exactly as :106)."""
    V = s_out["logits"].shape[-1]
    wf = min(1.0, epoch / warmup_epochs)
    ca, cb, cg = alpha * wf + (1 - wf) * 0.9, beta * wf, gamma * wf
    s_logits, t_logits, tg = s_out["logits"].reshape(-1, V), t_out["logits"].reshape(-1, V), targets.reshape(-1)
    kd = -(torch.softmax(t_logits / tau, -1) * torch.log_softmax(s_logits / tau, -1)).sum(-1).mean() * tau ** 2
    ce = F.cross_entropy(s_logits, tg, reduction="none")
    hard = (focal_alpha * (1 - torch.exp(-ce)) ** focal_gamma * ce).mean()
    token = ca * kd + (1 - ca) * hard
    feat = torch.tensor(0.0)
    if "encoder_features" in s_out and "encoder_features" in t_out:
        cos = (F.normalize(s_out["encoder_features"], p=2, dim=-1) * F.normalize(t_out["encoder_features"], p=2, dim=-1)).sum(-1)
        feat = 1 - cos.mean()
    hid = torch.tensor(0.0)
    if s_out.get("hidden_states") is not None and t_out.get("hidden_states") is not None and attention_weights is not None:
        sh, th = torch.stack(list(s_out["hidden_states"])), torch.stack(list(t_out["hidden_states"]))
        w = torch.softmax(attention_weights, dim=0).unsqueeze(-1)
        hid = F.mse_loss((sh * w).sum(0), (th * w).sum(0))
    total = token + cb * feat + cg * hid
    return total, {"total_loss": float(total.detach()), "token_kd_loss": float(token.detach()), "feature_kd_loss": float(feat.detach()),
                   "hidden_kd_loss": float(hid.detach()), "kd_loss": float(kd.detach()), "hard_loss": float(hard.detach()),
                   "ce_loss": float(hard.detach())}


# ----------------------------------------------------------------------------- A14: one KD step
def kd_forward_backward(student_sd: SD, teacher_sd: SD, proj_sd: SD, images: torch.Tensor, captions: torch.Tensor, *,
                        hidden: int, layers: int, refine: bool, t_heads: int, t_layers: int,
                        alpha=0.7, beta=0.2, gamma=0.1, tau=4.0, train: bool = True,
                        teacher_hiddens: Optional[List[torch.Tensor]] = None):
    """Loop body /root/reference/src/train_student_kd.py:258-288 without AMP/accumulation
    (fp32, dropout p=0): caption shift, teacher (no grad), student, projector, loss, backward.
    Leaves .grad on every leaf in student_sd / proj_sd that requires grad."""
    cin, ctg = captions[:-1], captions[1:]
    t_logits, t_feats = teacher_forward(teacher_sd, images, cin, heads=t_heads, layers=t_layers)
    logits, enc, hids, _ = student_forward(student_sd, images, cin, hidden=hidden, layers=layers, refine=refine,
                                           train=train)
    t_proj = feature_projector(proj_sd, t_feats, enc.shape[1])
    s_out = {"logits": logits, "encoder_features": enc, "hidden_states": hids}
    t_out = {"logits": t_logits, "encoder_features": t_proj, "hidden_states": teacher_hiddens}
    loss, parts = distillation_loss(s_out, t_out, ctg, alpha=alpha, beta=beta, gamma=gamma, tau=tau)
    loss.backward()
    return loss.detach(), parts, logits.detach()


def student_state_shapes(vocab: int, embed: int, hidden: int, layers: int, refine: bool) -> Dict[str, Tuple[int, ...]]:
    """state_dict key -> shape of CaptioningStudent; lets tests build key-seeded
    weights without instantiating any module."""
    s: Dict[str, Tuple[int, ...]] = {}

    def bn(p, c):
        s[p + ".weight"] = (c,); s[p + ".bias"] = (c,); s[p + ".running_mean"] = (c,); s[p + ".running_var"] = (c,)

    r = "encoder.resnet"
    s[r + ".0.weight"] = (64, 3, 7, 7); bn(r + ".1", 64)
    inpl = 64
    for child, planes, blocks, stride in RESNET_STAGES:
        for i in range(blocks):
            q = f"{r}.{child}.{i}"
            s[q + ".conv1.weight"] = (planes, inpl, 1, 1); bn(q + ".bn1", planes)
            s[q + ".conv2.weight"] = (planes, planes, 3, 3); bn(q + ".bn2", planes)
            s[q + ".conv3.weight"] = (planes * 4, planes, 1, 1); bn(q + ".bn3", planes * 4)
            if i == 0:
                s[q + ".downsample.0.weight"] = (planes * 4, inpl, 1, 1); bn(q + ".downsample.1", planes * 4)
            inpl = planes * 4
    E, H, V = embed, hidden, vocab
    s["encoder.projection.0.weight"] = (E, 2048); s["encoder.projection.0.bias"] = (E,)
    s["encoder.projection.3.weight"] = (E,); s["encoder.projection.3.bias"] = (E,)
    if refine:
        a = "attention_refinement"
        s[a + ".attention.in_proj_weight"] = (3 * E, E); s[a + ".attention.in_proj_bias"] = (3 * E,)
        s[a + ".attention.out_proj.weight"] = (E, E); s[a + ".attention.out_proj.bias"] = (E,)
        s[a + ".ffn.0.weight"] = (2 * E, E); s[a + ".ffn.0.bias"] = (2 * E,)
        s[a + ".ffn.3.weight"] = (E, 2 * E); s[a + ".ffn.3.bias"] = (E,)
        for n in ("norm1", "norm2"):
            s[f"{a}.{n}.weight"] = (E,); s[f"{a}.{n}.bias"] = (E,)
    d = "decoder"
    s[d + ".embedding.weight"] = (V, E)
    s[d + ".attention.weight"] = (E, H + E); s[d + ".attention.bias"] = (E,)
    s[d + ".attention_combine.weight"] = (E, 2 * E); s[d + ".attention_combine.bias"] = (E,)
    for l in range(layers):
        s[f"{d}.lstm.weight_ih_l{l}"] = (4 * H, E if l == 0 else H); s[f"{d}.lstm.weight_hh_l{l}"] = (4 * H, H)
        s[f"{d}.lstm.bias_ih_l{l}"] = (4 * H,); s[f"{d}.lstm.bias_hh_l{l}"] = (4 * H,)
    s[d + ".output_projection.0.weight"] = (E, H); s[d + ".output_projection.0.bias"] = (E,)
    s[d + ".output_projection.3.weight"] = (V, E); s[d + ".output_projection.3.bias"] = (V,)
    return s


def teacher_state_shapes(vocab: int, embed: int, layers: int) -> Dict[str, Tuple[int, ...]]:
    s: Dict[str, Tuple[int, ...]] = {}
    e = "encoder"
    s[e + ".cls_token"] = (1, 1, 384); s[e + ".pos_embed"] = (1, 197, 384)
    s[e + ".patch_embed.proj.weight"] = (384, 3, 16, 16); s[e + ".patch_embed.proj.bias"] = (384,)
    for i in range(12):
        q = f"{e}.blocks.{i}"
        for n in ("norm1", "norm2"):
            s[f"{q}.{n}.weight"] = (384,); s[f"{q}.{n}.bias"] = (384,)
        s[q + ".attn.qkv.weight"] = (1152, 384); s[q + ".attn.qkv.bias"] = (1152,)
        s[q + ".attn.proj.weight"] = (384, 384); s[q + ".attn.proj.bias"] = (384,)
        s[q + ".mlp.fc1.weight"] = (1536, 384); s[q + ".mlp.fc1.bias"] = (1536,)
        s[q + ".mlp.fc2.weight"] = (384, 1536); s[q + ".mlp.fc2.bias"] = (384,)
    s[e + ".norm.weight"] = (384,); s[e + ".norm.bias"] = (384,)
    E = embed
    if E != 384:
        s["encoder_projection.weight"] = (E, 384); s["encoder_projection.bias"] = (E,)
    s["embedding.weight"] = (vocab, E)
    for i in range(layers):
        q = f"decoder.layers.{i}"
        for a in ("self_attn", "multihead_attn"):
            s[f"{q}.{a}.in_proj_weight"] = (3 * E, E); s[f"{q}.{a}.in_proj_bias"] = (3 * E,)
            s[f"{q}.{a}.out_proj.weight"] = (E, E); s[f"{q}.{a}.out_proj.bias"] = (E,)
        s[q + ".linear1.weight"] = (2 * E, E); s[q + ".linear1.bias"] = (2 * E,)
        s[q + ".linear2.weight"] = (E, 2 * E); s[q + ".linear2.bias"] = (E,)
        for n in ("norm1", "norm2", "norm3"):
            s[f"{q}.{n}.weight"] = (E,); s[f"{q}.{n}.bias"] = (E,)
    s["pre_output_norm.weight"] = (E,); s["pre_output_norm.bias"] = (E,)
    s["fc_out.weight"] = (vocab, E); s["fc_out.bias"] = (vocab,)
    return s


def projector_state_shapes(teacher_dim: int, student_dim: int) -> Dict[str, Tuple[int, ...]]:
    if teacher_dim == student_dim:
        return {}
    return {"feature_projection.0.weight": (student_dim, teacher_dim), "feature_projection.0.bias": (student_dim,),
            "feature_projection.3.weight": (student_dim,), "feature_projection.3.bias": (student_dim,)}
