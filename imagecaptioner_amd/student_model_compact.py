"""MI355X-native compact student — drop-in for the reference's src/student_model_compact.py
(/root/reference/src/student_model_compact.py:9-336): same class names, constructor arguments,
forward signatures / return arity, attribute names and state_dict keys (torchvision's `mobilenet_v2().features` names under
`encoder.backbone.*`).  Every arithmetic step runs in libick.so:

  * MobileNetV2 features as ONE autograd node (MobileNetTrunkFn): NHWC activations; the 1x1 convolutions are fp32-MFMA GEMMs
    over pixels with the BatchNorm statistics in their epilogue, the 3x3 depthwise convolutions are HBM-bound float4 kernels
    (csrc/depthwise.hip) followed by a column-statistics pass; BatchNorm + ReLU6 (+ residual) is one elementwise pass;
    train-mode batch statistics in frozen modules too (the reference never puts them in eval mode); hand-scheduled backward
    through modules 10-18 (the reference freezes features[0:10], :26-30);
  * decoder: the dot-product attention + additive fusion step is one kernel per token (ick_dot_attn_fwd), the LSTM layer
    the fused MFMA GEMM + cell kernel of the main student (ick_lstm_layer_fwd), the vocabulary projection one GEMM over all
    T*B rows after the loop; hand-written BPTT with the weight gradients deferred to batched GEMMs.
"""
from __future__ import annotations

from typing import List

import torch
import torch.nn as nn
from torch.autograd import Function

from . import nn as hnn
from . import ops
from ._lib import ACT_NONE, ACT_RELU, OP_NT
from .student_model import _LSTMParams

MBV2_SETTINGS = ((1, 16, 1, 1), (6, 24, 2, 2), (6, 32, 3, 2), (6, 64, 4, 2), (6, 96, 3, 1), (6, 160, 3, 2), (6, 320, 1, 1))
RELU6 = 2


class _DWConv(nn.Module):
    """depthwise 3x3 weight holder with nn.Conv2d's (C,1,3,3) shape (state_dict compatible)."""

    def __init__(self, c, stride):
        super().__init__()
        w = torch.empty(c, 1, 3, 3)
        nn.init.kaiming_normal_(w, mode="fan_out")
        self.weight = nn.Parameter(w)
        self.stride = stride


class _PWConv(nn.Module):
    """1x1 convolution weight holder, logical (Cout,Cin,1,1): a [Cout][Cin] GEMM operand as it stands."""

    def __init__(self, cin, cout):
        super().__init__()
        w = torch.empty(cout, cin, 1, 1)
        nn.init.kaiming_normal_(w, mode="fan_out")
        self.weight = nn.Parameter(w)


def _cna(conv):
    return nn.Sequential(conv, hnn.BatchNorm2d(conv.weight.shape[0]), hnn._Placeholder())


class InvertedResidual(nn.Module):
    def __init__(self, cin, cout, stride, t):
        super().__init__()
        hid = cin * t
        self.use_res_connect = stride == 1 and cin == cout
        self.expand = t != 1
        layers = []
        if t != 1:
            layers.append(_cna(_PWConv(cin, hid)))
        layers += [_cna(_DWConv(hid, stride)), _PWConv(hid, cout), hnn.BatchNorm2d(cout)]
        self.conv = nn.Sequential(*layers)


def build_mobilenet_v2_features() -> nn.Sequential:
    first = nn.Sequential(hnn.Conv2d(3, 32, 3, stride=2, padding=1), hnn.BatchNorm2d(32), hnn._Placeholder())
    mods: List[nn.Module] = [first]
    cin = 32
    for t, c, n, s in MBV2_SETTINGS:
        for i in range(n):
            mods.append(InvertedResidual(cin, c, s if i == 0 else 1, t))
            cin = c
    mods.append(_cna(_PWConv(cin, 1280)))
    return nn.Sequential(*mods)


# ----------------------------------------------------------------------------- conv + BN (+ ReLU6) building blocks
def _bn_apply(raw, stats, bn, act, residual, train):
    """y = act(bn(raw) [+ residual]); train: batch statistics from `stats` (2, R, C) fp64 + running-stat update."""
    if train:
        bn._ick_eval_co = None
        M = raw.numel() // raw.shape[-1]
        co = ops.bn_finalize(stats[0], stats[1], M, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.momentum, bn.eps)
        y = ops.scale_shift_act(raw, co[0], co[1], residual, act)
        return y, co[2], co[3]
    co = hnn._eval_coeffs(bn)
    return ops.scale_shift_act(raw, co[0], co[1], residual, act), None, None


def _pw(x, conv: _PWConv, bn, act, residual, train, counters):
    """1x1 convolution (GEMM over pixels, BatchNorm statistics in the epilogue) + BN + activation."""
    B, H, W, Cin = x.shape
    w = conv.weight.view(conv.weight.shape[0], Cin)
    Cout, M = w.shape[0], B * H * W
    raw = ops.empty(B, H, W, Cout, device=x.device)
    stats = None
    if train:
        R = ops.stat_copies(M)
        stats = torch.zeros(2, R, Cout, dtype=torch.float64, device=x.device)
        counters.append(bn.num_batches_tracked)
    ops.gemm_raw(OP_NT, x.data_ptr(), w.data_ptr(), raw.data_ptr(), M, Cout, Cin, Cin, Cin, Cout,
                 stat_sum=stats[0].data_ptr() if train else None, stat_sq=stats[1].data_ptr() if train else None,
                 stat_copies=stats.shape[1] if train else 1, stat_stride=Cout)
    y, mean, inv = _bn_apply(raw, stats, bn, act, residual, train)
    return y, raw, mean, inv


def _dw(x, conv: _DWConv, bn, train, counters):
    raw = ops.dwconv3x3_fwd(x, conv.weight, conv.stride)
    stats = None
    if train:
        stats = torch.zeros(2, 1, raw.shape[-1], dtype=torch.float64, device=x.device)
        ops.colstats(raw, stats[:, 0])
        counters.append(bn.num_batches_tracked)
    y, mean, inv = _bn_apply(raw, stats, bn, RELU6, None, train)
    return y, raw, mean, inv


class MobileNetTrunkFn(Function):
    @staticmethod
    def forward(ctx, images, feats: nn.Sequential, train: bool, *params):
        images = hnn._c(images)
        x4 = ops.nchw3_to_nhwc4(images)
        stem, bn0 = feats[0][0], feats[0][1]
        if any(p.requires_grad for p in feats[0].parameters()):
            raise NotImplementedError("features[0] is frozen in the reference (student_model_compact.py:26-30)")
        counters: List[torch.Tensor] = []
        w4 = ops.nchw3_to_nhwc4(hnn._c(stem.weight.detach()))          # (32,3,3,3) -> (32,3,3,4)
        Ho, Wo = ops.conv_out_hw(x4.shape[1], x4.shape[2], 3, 3, 2, 1)
        stats = None
        if train:
            stats = torch.zeros(2, ops.stat_copies(x4.shape[0] * Ho * Wo), 32, dtype=torch.float64, device=images.device)
            counters.append(bn0.num_batches_tracked)
        raw = ops.conv_fwd(x4, w4, 2, 1, stats=(stats[0], stats[1]) if train else None)
        y, _, _ = _bn_apply(raw, stats, bn0, RELU6, None, train)
        blocks = list(feats)[1:-1]
        first = next((i for i, b in enumerate(blocks) if any(p.requires_grad for p in b.parameters())), len(blocks))
        head_trainable = any(p.requires_grad for p in feats[-1].parameters())
        want_bwd = any(ctx.needs_input_grad) and (first < len(blocks) or head_trainable)
        ctx.eval_mode_graph = want_bwd and not train
        want_bwd = want_bwd and train
        recs = []
        for i, blk in enumerate(blocks):
            keep = want_bwd and i >= first
            x_in = y
            j = 0
            r = dict(x=x_in)
            if blk.expand:
                y, r["r0"], r["m0"], r["i0"] = _pw(y, blk.conv[0][0], blk.conv[0][1], RELU6, None, train, counters)
                r["a0"] = y
                j = 1
            y, r["r1"], r["m1"], r["i1"] = _dw(y, blk.conv[j][0], blk.conv[j][1], train, counters)
            r["a1"] = y
            y, r["r2"], r["m2"], r["i2"] = _pw(y, blk.conv[j + 1], blk.conv[j + 2], 0, x_in if blk.use_res_connect else None, train,
                                                counters)
            if keep:
                recs.append(r)
        head_in = y
        y, rh, mh, ih = _pw(y, feats[-1][0], feats[-1][1], RELU6, None, train, counters)
        if counters:
            torch._foreach_add_(counters, 1)
        ctx.blocks, ctx.first, ctx.recs, ctx.feats = blocks, first, recs, feats
        ctx.head = dict(x=head_in, raw=rh, mean=mh, inv=ih, out=y) if (want_bwd and head_trainable) else None
        Nb, H, W, C = y.shape
        ctx.pool_from = None
        if (H, W) != (7, 7):
            ctx.pool_from = (H, W)
            y = ops.adaptive_avgpool_fwd(y, 7, 7)
        return y.view(Nb, 49, C)

    @staticmethod
    def backward(ctx, dy):
        if ctx.eval_mode_graph:
            raise NotImplementedError("backward through eval-mode BatchNorm is not built")
        blocks, first, recs, feats = ctx.blocks, ctx.first, ctx.recs, ctx.feats
        none = (None,) * len(ctx.needs_input_grad)
        if ctx.head is None:
            return none
        gb = hnn.grad_buf
        hd = ctx.head
        Nb, H, W, C = hd["out"].shape
        d = ops.adaptive_avgpool_bwd(hnn._c(dy).view(Nb, 7, 7, C), H, W) if ctx.pool_from else hnn._c(dy).view(Nb, H, W, C)

        def pw_bwd(d_out, y_mask, raw, mean, inv, conv, bn, x_in, act, need_in, want_g=False):
            """adjoint of _pw: returns (dx or None, g = masked upstream gradient if want_g)."""
            tr = bn.weight.requires_grad
            draw, g = ops.bn_bwd(d_out, y_mask, raw, mean, inv, bn.weight, gb(bn.weight) if tr else None, gb(bn.bias) if tr else None,
                                 want_g, True, act=act)
            Cout, Cin = conv.weight.shape[0], conv.weight.shape[1]
            w = conv.weight.view(Cout, Cin)
            if conv.weight.requires_grad:
                ops.linear_bwd_weight(draw.view(-1, Cout), x_in.view(-1, Cin), gb(conv.weight).view(Cout, Cin))
            dx = ops.linear_bwd_data(draw.view(-1, Cout), w).view(x_in.shape) if need_in else None
            return dx, g

        bnh, convh = feats[-1][1], feats[-1][0]
        d, _ = pw_bwd(d, hd["out"], hd["raw"], hd["mean"], hd["inv"], convh, bnh, hd["x"], RELU6, len(recs) > 0)
        for i in range(len(blocks) - 1, first - 1, -1):
            blk, r = blocks[i], recs[i - first]
            j = 1 if blk.expand else 0
            need_in = i > first
            # project 1x1 (+ residual): no activation -> no mask; the residual branch receives d unchanged
            dres = d if blk.use_res_connect else None
            da1, _ = pw_bwd(d, None, r["r2"], r["m2"], r["i2"], blk.conv[j + 1], blk.conv[j + 2], r["a1"], 0, True)
            dwc, dwbn = blk.conv[j][0], blk.conv[j][1]
            tr = dwbn.weight.requires_grad
            draw, _ = ops.bn_bwd(da1, r["a1"], r["r1"], r["m1"], r["i1"], dwbn.weight, gb(dwbn.weight) if tr else None,
                                 gb(dwbn.bias) if tr else None, False, True, act=RELU6)
            x_dw = r["a0"] if blk.expand else r["x"]
            if dwc.weight.requires_grad:
                ops.dwconv3x3_wgrad(draw, x_dw, gb(dwc.weight), dwc.stride)
            need_dw_in = blk.expand or need_in
            dxdw = ops.dwconv3x3_dgrad(draw, dwc.weight, x_dw.shape[1:3], dwc.stride) if need_dw_in else None
            if blk.expand:
                dx, _ = pw_bwd(dxdw, r["a0"], r["r0"], r["m0"], r["i0"], blk.conv[0][0], blk.conv[0][1], r["x"], RELU6, need_in)
            else:
                dx = dxdw
            if need_in:
                d = ops.add(dx, dres) if dres is not None else dx
            recs[i - first] = None
        return none


class _CompactProjection(nn.Sequential):
    """nn.Sequential(Linear, ReLU, Dropout(0.1)) with the reference's child indices; fused forward."""

    def forward(self, x):
        lin, drop = self[0], self[2]
        return hnn.dropout(hnn.linear(x, lin.weight, lin.bias, ACT_RELU), drop.p, self.training)


class CompactCNNEncoder(nn.Module):
    """reference: CompactCNNEncoder, /root/reference/src/student_model_compact.py:9-64."""

    def __init__(self, embed_size=256, fine_tune=True):
        super().__init__()
        self.embed_size = embed_size
        self.backbone = build_mobilenet_v2_features()      # random init: pretrained weights cannot be fetched offline
        if fine_tune:
            for i, layer in enumerate(self.backbone):
                if i < 10:
                    for p in layer.parameters():
                        p.requires_grad = False
        self.adaptive_pool = nn.AdaptiveAvgPool2d((7, 7))
        self.projection = _CompactProjection(hnn.Linear(1280, embed_size), nn.ReLU(), nn.Dropout(0.1))

    def forward(self, images):
        params = [p for p in self.backbone.parameters() if p.requires_grad]
        f = MobileNetTrunkFn.apply(images, self.backbone, self.training, *params)     # (B,49,1280)
        return self.projection(f)


class CompactDecoderFn(Function):
    """teacher-forced decode of all T steps + hand-written BPTT (reference loop: student_model_compact.py:163-190), any number
    of LSTM layers, optional caller-supplied initial state (h0, c0), each (layers, B, H) (reference :140,:155-156): the
    attention query is the TOP layer's state of the previous step (`hidden[0][-1]`, :172)."""

    @staticmethod
    def forward(ctx, feats, captions, dec: "CompactLSTMDecoder", h0, c0, *params):
        feats, captions = hnn._c(feats), hnn._c(captions)
        T, B = captions.shape
        _, P, E = feats.shape
        H, V, NL = dec.hidden_size, dec.vocab_size, dec.num_layers
        dev = feats.device
        if h0 is not None:
            h0, c0 = hnn._c(h0.detach()), hnn._c(c0.detach())
        Wa, ba = dec.attention.weight, dec.attention.bias
        emb = ops.embedding_fwd(captions, dec.embedding.weight)            # (T,B,E)
        keep = any(ctx.needs_input_grad)
        Hall, Call = ops.empty(NL, T, B, H, device=dev), ops.empty(NL, T, B, H, device=dev)
        Gates = ops.empty(NL, T, B, 4 * H, device=dev) if keep else None
        HP = ops.empty(T, B, E, device=dev)
        attw, X = ops.empty(T, B, P, device=dev), ops.empty(T, B, E, device=dev)
        zero_h = ops.zeros(B, H, device=dev)
        hp = lambda l, t: Hall[l, t - 1] if t > 0 else (h0[l] if h0 is not None else None)
        cp = lambda l, t: Call[l, t - 1] if t > 0 else (c0[l] if c0 is not None else None)
        for t in range(T):
            h_top = hp(NL - 1, t)
            ops.gemm_nt(h_top if h_top is not None else zero_h, Wa.data_ptr(), E, H, H, HP[t], bias=ba)
            ops.dot_attn_fwd(HP[t], feats, emb[t], attw[t], X[t])
            inp = X[t]
            for l in range(NL):
                wi, wh, bi, bh = dec.lstm.layer(l)
                ops.lstm_layer_fwd(inp, hp(l, t), wi, wh, bi, bh, cp(l, t), Gates[l, t] if keep else None, Call[l, t], Hall[l, t])
                inp = Hall[l, t]
        Wo, bo = dec.output_projection.weight, dec.output_projection.bias
        Htop = Hall[NL - 1]
        logits = ops.linear_fwd(Htop.view(T * B, H), Wo, bo).view(T, B, V)
        if keep:
            ctx.dec, ctx.dims = dec, (T, B, P, E, H, V, NL)
            ctx.saved = dict(feats=feats, captions=captions, Hall=Hall, Call=Call, Gates=Gates, HP=HP, attw=attw, X=X, h0=h0, c0=c0)
        ctx.mark_non_differentiable(attw)
        return logits, Htop, attw

    @staticmethod
    def backward(ctx, dlogits, dH_ext, _dattw):
        dec, s = ctx.dec, ctx.saved
        T, B, P, E, H, V, NL = ctx.dims
        dev = s["feats"].device
        gb = hnn.grad_buf
        Wa, Wo = dec.attention.weight, dec.output_projection.weight
        Hall, Call, Gates, h0, c0 = s["Hall"], s["Call"], s["Gates"], s["h0"], s["c0"]
        Htop = Hall[NL - 1]
        if dlogits is not None:
            dl = hnn._c(dlogits).view(T * B, V)
            dHs = ops.empty(T, B, H, device=dev)
            ops.gemm_nn(dl, Wo.data_ptr(), V, H, H, dHs, residual=hnn._c(dH_ext) if dH_ext is not None else None)
            if Wo.requires_grad:
                ops.linear_bwd_weight(dl, Htop.view(T * B, H), gb(Wo))
                ops.colsum_into(dl, gb(dec.output_projection.bias))
        else:
            dHs = hnn._c(dH_ext) if dH_ext is not None else ops.zeros(T, B, H, device=dev)
        DG = ops.empty(NL, T, B, 4 * H, device=dev)
        dX = ops.empty(T, B, E, device=dev)
        dHP = ops.empty(T, B, E, device=dev)
        dfeats = ops.zeros(B, P, E, device=dev)
        carry_c = [ops.zeros(B, H, device=dev) for _ in range(NL)]
        carry_h = ops.zeros(NL, T, B, H, device=dev)        # carry_h[l, t]: dL/dh_l(t) arriving from step t+1 (split-K arena)
        d_inp = ops.zeros(NL, B, H, device=dev) if NL > 1 else None     # gradient into layer l's input at the current step
        want_state = h0 is not None and (ctx.needs_input_grad[3] or ctx.needs_input_grad[4])
        dh0 = ops.zeros(NL, B, H, device=dev) if want_state else None
        for t in range(T - 1, -1, -1):
            for l in range(NL - 1, -1, -1):
                wi, wh, _, _ = dec.lstm.layer(l)
                dh_a = dHs[t] if l == NL - 1 else d_inp[l + 1]
                c_prev = Call[l, t - 1] if t > 0 else (c0[l] if c0 is not None else None)
                ops.lstm_cell_bwd(dh_a, carry_h[l, t] if t < T - 1 else None, carry_c[l] if t < T - 1 else None, Gates[l, t],
                                  Call[l, t], c_prev, DG[l, t], carry_c[l])
                if l > 0:
                    ops.gemm_nn(DG[l, t], wi.data_ptr(), 4 * H, H, H, d_inp[l])
                else:
                    ops.gemm_nn(DG[0, t], wi.data_ptr(), 4 * H, E, E, dX[t])              # d(emb + ctx)
                if t > 0:
                    ops.gemm_nn(DG[l, t], wh.data_ptr(), 4 * H, H, H, carry_h[l, t - 1], zeroed=True)
                elif want_state:
                    ops.gemm_nn(DG[l, 0], wh.data_ptr(), 4 * H, H, H, dh0[l], zeroed=True)
            ops.dot_attn_bwd(dX[t], s["attw"][t], s["HP"][t], s["feats"], dfeats, dHP[t])
            if t > 0:
                ops.gemm_nn(dHP[t], Wa.data_ptr(), E, H, H, carry_h[NL - 1, t - 1], accumulate=True)
            elif want_state:
                ops.gemm_nn(dHP[0], Wa.data_ptr(), E, H, H, dh0[NL - 1], accumulate=True)
        for l in range(NL):
            wi, wh, bi, bh = dec.lstm.layer(l)
            if not wi.requires_grad:
                continue
            DG2 = DG[l].view(T * B, 4 * H)
            inp_all = s["X"] if l == 0 else Hall[l - 1]
            k_in = wi.shape[1]
            ops.gemm_tn_acc(DG2, inp_all.view(T * B, k_in), gb(wi).data_ptr(), 4 * H, k_in, k_in)
            if T > 1:
                ops.gemm_tn_acc(DG[l, 1:].reshape((T - 1) * B, 4 * H), Hall[l, :-1].reshape((T - 1) * B, H), gb(wh).data_ptr(), 4 * H, H, H)
            else:
                gb(wh)
            if h0 is not None:
                ops.gemm_tn_acc(DG[l, 0], h0[l], gb(wh).data_ptr(), 4 * H, H, H)
            ops.colsum_into(DG2, gb(bi))
            ops.colsum_into(DG2, gb(bh))
        if Wa.requires_grad:
            if T > 1:
                ops.gemm_tn_acc(dHP[1:].reshape((T - 1) * B, E), Htop[:-1].reshape((T - 1) * B, H), gb(Wa).data_ptr(), E, H, H)
            else:
                gb(Wa)
            if h0 is not None:
                ops.gemm_tn_acc(dHP[0], h0[NL - 1], gb(Wa).data_ptr(), E, H, H)
            ops.colsum_into(dHP.view(T * B, E), gb(dec.attention.bias))
        if dec.embedding.weight.requires_grad:
            ops.embedding_bwd(s["captions"], dX.view(T * B, E), gb(dec.embedding.weight))
        dc0 = torch.stack([hnn._c(c) for c in carry_c], 0) if want_state else None
        ctx.saved = None
        return (dfeats, None, None, dh0, dc0) + (None,) * (len(ctx.needs_input_grad) - 5)


class CompactLSTMDecoder(nn.Module):
    """reference: CompactLSTMDecoder, /root/reference/src/student_model_compact.py:67-195."""

    def __init__(self, vocab_size, embed_size=256, hidden_size=256, num_layers=1, dropout=0.1):
        super().__init__()
        self.embed_size, self.hidden_size, self.num_layers, self.vocab_size = embed_size, hidden_size, num_layers, vocab_size
        self.embedding = hnn.Embedding(vocab_size, embed_size)
        nn.init.uniform_(self.embedding.weight, -0.1, 0.1)
        self.attention = hnn.Linear(hidden_size, embed_size)
        self.lstm = _LSTMParams(embed_size, hidden_size, num_layers, 0)
        self.output_projection = hnn.Linear(hidden_size, vocab_size)

    def init_hidden(self, batch_size, device):
        return (ops.zeros(self.num_layers, batch_size, self.hidden_size, device=device),
                ops.zeros(self.num_layers, batch_size, self.hidden_size, device=device))

    @torch.no_grad()
    def simple_attention(self, hidden, image_features):
        """(context (B,E), attention_weights (B,L)) — reference :114-138 (inference entry)."""
        feats = hnn._c(image_features)
        B, L, E = feats.shape
        hp = ops.gemm_nt(hnn._c(hidden), self.attention.weight.data_ptr(), E, self.hidden_size, self.hidden_size,
                         ops.empty(B, E, device=feats.device), bias=self.attention.bias)
        w, x = ops.empty(B, L, device=feats.device), ops.empty(B, E, device=feats.device)
        ops.dot_attn_fwd(hp, feats, ops.zeros(B, E, device=feats.device), w, x)
        return x, w

    def forward(self, image_features, captions, hidden=None):
        """hidden: optional (h0, c0), each (num_layers, B, H) (reference :140,:155-156); None = zero state."""
        params = [p for p in self.parameters() if p.requires_grad]
        h0, c0 = hidden if hidden is not None else (None, None)
        logits, hs, attw = CompactDecoderFn.apply(image_features, captions, self, h0, c0, *params)
        return logits, list(hs.unbind(0)), list(attw.unbind(0))

    @torch.no_grad()
    def greedy(self, image_features, max_length=20, start_id=1):
        feats = hnn._c(image_features)
        B, P, E = feats.shape
        H, V, NL = self.hidden_size, self.vocab_size, self.num_layers
        dev = feats.device
        h = [ops.zeros(B, H, device=dev) for _ in range(NL)]
        c = [ops.zeros(B, H, device=dev) for _ in range(NL)]
        hn = [ops.empty(B, H, device=dev) for _ in range(NL)]
        cn = [ops.empty(B, H, device=dev) for _ in range(NL)]
        hp, w, x = ops.empty(B, E, device=dev), ops.empty(B, P, device=dev), ops.empty(B, E, device=dev)
        tok = torch.full((B,), start_id, dtype=torch.int64, device=dev)
        ids = torch.empty(max_length, B, dtype=torch.int64, device=dev)
        logits = ops.empty(max_length, B, V, device=dev)
        for t in range(max_length):
            emb = ops.embedding_fwd(tok, self.embedding.weight)
            ops.gemm_nt(h[NL - 1], self.attention.weight.data_ptr(), E, H, H, hp, bias=self.attention.bias)
            ops.dot_attn_fwd(hp, feats, emb, w, x)
            inp = x
            for l in range(NL):
                wi, wh, bi, bh = self.lstm.layer(l)
                ops.lstm_layer_fwd(inp, h[l], wi, wh, bi, bh, c[l], None, cn[l], hn[l])
                h[l], hn[l], c[l], cn[l] = hn[l], h[l], cn[l], c[l]
                inp = h[l]
            ops.linear_fwd(h[NL - 1], self.output_projection.weight, self.output_projection.bias, out=logits[t])
            tok = ops.argmax_rows(logits[t])
            ids[t] = tok
        return ids, logits


class CompactCaptioningStudent(nn.Module):
    """reference: CompactCaptioningStudent, /root/reference/src/student_model_compact.py:198-330."""

    def __init__(self, vocab_size, embed_size=256, hidden_size=256, num_layers=1, dropout=0.1, use_attention_refinement=False):
        super().__init__()
        self.vocab_size, self.embed_size, self.hidden_size = vocab_size, embed_size, hidden_size
        self.use_attention_refinement = use_attention_refinement
        self.encoder = CompactCNNEncoder(embed_size=embed_size, fine_tune=True)
        if use_attention_refinement:
            # parameter holders with the reference's names (:214-221); forward = packed in_proj GEMM + attention core + out_proj + LN
            self.attention_refinement = nn.MultiheadAttention(embed_dim=embed_size, num_heads=4, dropout=0.1, batch_first=True)
            self.norm = hnn.LayerNorm(embed_size)
        self.decoder = CompactLSTMDecoder(vocab_size, embed_size, hidden_size, num_layers, dropout)

    def _refine(self, f):
        B, L, E = f.shape
        a = self.attention_refinement
        x2 = f.reshape(B * L, E)
        qkv = hnn.linear(x2, a.in_proj_weight, a.in_proj_bias)
        p = a.dropout if self.training else 0.0
        o = hnn.SelfAttentionCoreFn.apply(qkv, B, 4, L, E // 4, p, hnn._next_seed() if p > 0 else 0)
        y = hnn.linear(o, a.out_proj.weight, a.out_proj.bias, ACT_NONE, x2)
        return self.norm(y).view(B, L, E)

    def forward(self, images, captions):
        encoder_features = self.encoder(images)
        refined = self._refine(encoder_features) if self.use_attention_refinement else encoder_features
        outputs, hidden_states, attention_weights = self.decoder(refined, captions)
        return outputs, encoder_features, hidden_states, attention_weights

    @torch.no_grad()
    def generate(self, images, max_length=20, start_id=1):
        was = self.training
        self.eval()
        try:
            f = self.encoder(images)
            if self.use_attention_refinement:
                f = self._refine(f)
            return self.decoder.greedy(f, max_length, start_id)
        finally:
            self.train(was)

    def caption_image(self, image, vocabulary, max_length=20, temperature=1.0):
        """greedy caption of one image as a list of words (reference :264-330; temperature rescales logits: argmax unchanged)."""
        self.eval()
        device = next(self.parameters()).device
        if image.dim() == 3:
            image = image.unsqueeze(0)
        start = vocabulary.stoi.get("<START>", vocabulary.stoi["<UNK>"])
        ids, _ = self.generate(image.to(device), max_length, start)
        words = []
        for i in ids[:, 0].tolist():
            if vocabulary.itos[i] == "<END>":
                break
            words.append(vocabulary.itos[i])
        return words


def count_parameters(model):
    total = sum(p.numel() for p in model.parameters())
    trainable = sum(p.numel() for p in model.parameters() if p.requires_grad)
    return total, trainable
