"""MI355X-native CNN-LSTM student — drop-in for the reference's src/student_model.py
(/root/reference/src/student_model.py): same class names, constructor arguments, forward
signatures / return arity, attribute names callers reach into and state_dict keys
(SURVEY.md §8(b)).  Every arithmetic step runs in libick.so (hand-written HIP for gfx950).

Structure differs from the reference on purpose (MI355X-first, not a port):
  * the ResNet-50 trunk is ONE autograd node (nn.ResNetTrunkFn): NHWC activations, fp32-MFMA
    implicit-GEMM convolutions with BatchNorm statistics in the epilogue, hand-scheduled backward;
  * the decoder's per-token Python loop of ~12 small ops becomes: one hoisted GEMM for the
    time-invariant half of the attention scores (W_f f + b_a), one for the embedding half of
    attention_combine, then per step 1 + 1 + 2L GEMV-sized GEMMs and 1 + L wavefront kernels;
    the vocabulary projection runs once over all T*B rows after the loop;
  * BPTT is written by hand (DecoderFn.backward): all weight gradients are deferred to batched
    GEMMs over T*B rows after the time loop.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
import torch.nn as nn
from torch.autograd import Function

from . import nn as hnn
from . import ops
from ._lib import ACT_NONE, ACT_RELU


class _FusedSeq(nn.Sequential):
    """nn.Sequential(Linear, ReLU, Dropout, X) with the reference's child indices (state_dict keys `0.*`, `3.*`)
    and a fused forward: GEMM with ReLU epilogue -> dropout -> X (Linear or LayerNorm)."""

    def forward(self, x):
        lin, drop, last = self[0], self[2], self[3]
        h = hnn.linear(x, lin.weight, lin.bias, ACT_RELU)
        h = hnn.dropout(h, drop.p, self.training)
        return last(h)


class CNNEncoder(nn.Module):
    """reference: CNNEncoder, /root/reference/src/student_model.py:8-69."""

    def __init__(self, embed_size=256, fine_tune=True):
        super().__init__()
        # random init here: pretrained ImageNet weights cannot be fetched offline; load_state_dict brings them in
        self.resnet = hnn.build_resnet50_children()
        if fine_tune:  # freeze children 0-5 (conv1, bn1, relu, maxpool, layer1, layer2), train layer3/4 (:23-30)
            for i, child in enumerate(self.resnet.children()):
                for p in child.parameters():
                    p.requires_grad = i >= 6
        self.adaptive_pool = nn.AdaptiveAvgPool2d((7, 7))   # attribute read by create_feature_projectors (output_size)
        self.projection = _FusedSeq(hnn.Linear(2048, embed_size), nn.ReLU(), nn.Dropout(0.2), hnn.LayerNorm(embed_size))
        self.embed_size = embed_size

    def forward(self, images):
        # (any size the five stride-2 stages leave non-empty: nn.AdaptiveAvgPool2d((7,7)) replicates bins when the feature map
        #  is smaller than 7x7, reference student_model.py:34,60, and so do ick_adaptive_avgpool_fwd / _bwd)
        f = hnn.resnet_trunk(images, self.resnet, self.training)     # (B,49,2048), NHWC rows (adaptive-pooled if not 224x224)
        return self.projection(f)                                    # (B,49,E)


class AttentionRefinement(nn.Module):
    """reference: AttentionRefinement, /root/reference/src/student_model.py:72-118 (post-LN MHA + FFN)."""

    def __init__(self, embed_size, num_heads=4):
        super().__init__()
        self.embed_size, self.num_heads = embed_size, num_heads
        # parameter holder with nn.MultiheadAttention's names/init (in_proj_weight, in_proj_bias, out_proj.*)
        self.attention = nn.MultiheadAttention(embed_dim=embed_size, num_heads=num_heads, dropout=0.1, batch_first=True)
        self.ffn = _FusedSeq(hnn.Linear(embed_size, embed_size * 2), nn.ReLU(), nn.Dropout(0.1),
                             hnn.Linear(embed_size * 2, embed_size))
        self.norm1 = hnn.LayerNorm(embed_size)
        self.norm2 = hnn.LayerNorm(embed_size)

    def forward(self, features):
        B, L, E = features.shape
        H = self.num_heads
        a = self.attention
        x2 = features.reshape(B * L, E)
        qkv = hnn.linear(x2, a.in_proj_weight, a.in_proj_bias)
        p = a.dropout if self.training else 0.0
        o = hnn.SelfAttentionCoreFn.apply(qkv, B, H, L, E // H, p, hnn._next_seed() if p > 0 else 0)
        y = hnn.linear(o, a.out_proj.weight, a.out_proj.bias, ACT_NONE, x2)          # + residual
        y = self.norm1(y)
        lin0, drop, lin3 = self.ffn[0], self.ffn[2], self.ffn[3]
        h = hnn.dropout(hnn.linear(y, lin0.weight, lin0.bias, ACT_RELU), drop.p, self.training)
        z = hnn.linear(h, lin3.weight, lin3.bias, ACT_NONE, y)                        # + residual
        return self.norm2(z).view(B, L, E)


class _LSTMParams(nn.Module):
    """Holder with nn.LSTM's parameter names (weight_ih_l{k}, weight_hh_l{k}, bias_ih_l{k}, bias_hh_l{k}) and the
    reference's init (xavier / orthogonal / zeros, student_model.py:159-165); forward = HIP single/multi step."""

    def __init__(self, input_size, hidden_size, num_layers, dropout=0.0):
        super().__init__()
        self.input_size, self.hidden_size, self.num_layers, self.dropout = input_size, hidden_size, num_layers, dropout
        self.batch_first = True
        for l in range(num_layers):
            i = input_size if l == 0 else hidden_size
            wi = torch.empty(4 * hidden_size, i); nn.init.xavier_uniform_(wi)
            wh = torch.empty(4 * hidden_size, hidden_size); nn.init.orthogonal_(wh)
            self.register_parameter(f"weight_ih_l{l}", nn.Parameter(wi))
            self.register_parameter(f"weight_hh_l{l}", nn.Parameter(wh))
            self.register_parameter(f"bias_ih_l{l}", nn.Parameter(torch.zeros(4 * hidden_size)))
            self.register_parameter(f"bias_hh_l{l}", nn.Parameter(torch.zeros(4 * hidden_size)))

    def layer(self, l):
        return (getattr(self, f"weight_ih_l{l}"), getattr(self, f"weight_hh_l{l}"), getattr(self, f"bias_ih_l{l}"),
                getattr(self, f"bias_hh_l{l}"))

    @torch.no_grad()
    def forward(self, x, hidden):
        """nn.LSTM call shape used by the reference's caption_image (student_model.py:360): x (B,S,in) batch_first,
        hidden = (h0,c0) each (layers,B,H).  Inference-only HIP path (no autograd through this entry)."""
        h0, c0 = hidden
        B, S, _ = x.shape
        H, NL = self.hidden_size, self.num_layers
        h = [h0[l].contiguous() for l in range(NL)]
        c = [c0[l].contiguous() for l in range(NL)]
        outs = ops.empty(B, S, H, device=x.device)
        G = ops.empty(B, 4 * H, device=x.device)
        for s in range(S):
            inp = x[:, s].contiguous()
            for l in range(NL):
                wi, wh, bi, bh = self.layer(l)
                hn, cn = ops.empty(B, H, device=x.device), ops.empty(B, H, device=x.device)
                if wi.shape[1] % 16 == 0 and H % 16 == 0:
                    ops.lstm_layer_fwd(inp, h[l], wi, wh, bi, bh, c[l], None, cn, hn)
                else:
                    ops.gemm_nt(inp, wi.data_ptr(), 4 * H, wi.shape[1], wi.shape[1], G)
                    ops.gemm_nt(h[l], wh.data_ptr(), 4 * H, H, H, G, accumulate=True)
                    ops.lstm_cell_fwd(G, bi, bh, c[l], None, cn, hn)
                h[l], c[l], inp = hn, cn, hn
            outs[:, s] = h[-1]
        return outs, (torch.stack(h, 0), torch.stack(c, 0))


def _fused_step_ok(E: int, H: int) -> bool:
    """the L + 1-launch decode step (csrc/decoder_fused.hip) needs MFMA-sized widths; every configuration of the
    reference (128/256, 256/512, 384/768) qualifies — other widths take the generic per-op path."""
    return E % 16 == 0 and H % 16 == 0 and E <= 1024


class DecoderFn(Function):
    """Teacher-forced decode of all T steps + hand-written BPTT (see module docstring)."""

    @staticmethod
    def forward(ctx, feats, captions, dec: "LSTMDecoder", train: bool, h0, c0, *params):
        """h0 / c0: optional caller-supplied initial state (layers, B, H) — LSTMDecoder.forward(hidden=...), reference :205,220.
        A state that requires grad receives its gradient (the reference lets autograd flow into it, :205-222): dL/dh0[l] =
        dG[l, 0] W_hh[l] (+ dhW[0] W_h for the top layer, whose state feeds step 0's attention), dL/dc0[l] = the cell
        adjoint's carry after step 0."""
        feats = hnn._c(feats)
        captions = hnn._c(captions)
        if h0 is not None:
            h0, c0 = hnn._c(h0.detach()), hnn._c(c0.detach())
        hp = lambda l, t: Hall[l, t - 1] if t > 0 else (h0[l] if h0 is not None else None)
        cp = lambda l, t: Call[l, t - 1] if t > 0 else (c0[l] if c0 is not None else None)
        T, B = captions.shape
        _, P, E = feats.shape
        H, NL, V = dec.hidden_size, dec.num_layers, dec.vocab_size
        dev = feats.device
        Wa, ba = dec.attention.weight, dec.attention.bias
        Wc, bc = dec.attention_combine.weight, dec.attention_combine.bias
        W1, b1 = dec.output_projection[0].weight, dec.output_projection[0].bias
        W2, b2 = dec.output_projection[3].weight, dec.output_projection[3].bias
        p_drop = dec.output_projection[2].p if train else 0.0
        p_lstm = dec.lstm.dropout if (train and NL > 1) else 0.0
        fs = 4
        keep = any(ctx.needs_input_grad)        # (forward itself always runs in no-grad mode)
        feats2 = feats.view(B * P, E)
        Uf = ops.gemm_nt(feats2, Wa.data_ptr() + H * fs, E, E, H + E, ops.empty(B * P, E, device=dev), bias=ba).view(B, P, E)
        emb = ops.embedding_fwd(captions, dec.embedding.weight)                               # (T,B,E)
        Xe = ops.gemm_nt(emb.view(T * B, E), Wc.data_ptr(), E, E, 2 * E, ops.empty(T * B, E, device=dev), bias=bc).view(T, B, E)
        Hall = ops.empty(NL, T, B, H, device=dev)
        Call = ops.empty(NL, T, B, H, device=dev)
        Gates = ops.empty(NL, T, B, 4 * H, device=dev) if keep else None
        Hd = ops.empty(NL - 1, T, B, H, device=dev) if p_lstm > 0 else None                   # dropped inter-layer inputs
        attw = ops.empty(T, B, P, device=dev)
        ctxs = ops.empty(T, B, E, device=dev)
        seeds = [[hnn._next_seed() for _ in range(T)] for _ in range(NL - 1)] if p_lstm > 0 else None
        fused = _fused_step_ok(E, H)
        if fused:
            # L + 1 launches per token (csrc/decoder_fused.hip): stage A per image, then one MFMA GEMM + cell per layer
            hW = ops.empty(T, B, E, device=dev)
            X = ops.empty(T, B, E, device=dev)
            for t in range(T):
                ops.dec_attn_x_fwd(hp(NL - 1, t), Wa, Uf, feats, Wc, Xe[t], hW[t], attw[t], ctxs[t], X[t])
                inp = X[t]
                for l in range(NL):
                    wi, wh, bi, bh = dec.lstm.layer(l)
                    hd = Hd[l, t] if (p_lstm > 0 and l < NL - 1) else None
                    ops.lstm_layer_fwd(inp, hp(l, t), wi, wh, bi, bh, cp(l, t),
                                       Gates[l, t] if keep else None, Call[l, t], Hall[l, t], hd, p_lstm,
                                       seeds[l][t] if hd is not None else 0)
                    inp = hd if hd is not None else Hall[l, t]
        else:
            # generic path (widths that are not multiples of 16): GEMV-sized split-K GEMMs (fp32 atomics) need zeroed
            # outputs; one fill per arena instead of one per launch
            hW = ops.zeros(T, B, E, device=dev)
            X = ops.zeros(T, B, E, device=dev)
            zero_h = ops.zeros(B, H, device=dev)
            Gall = ops.zeros(T, NL, B, 4 * H, device=dev)
            for t in range(T):
                h_top = hp(NL - 1, t)
                ops.gemm_nt(h_top if h_top is not None else zero_h, Wa.data_ptr(), E, H, H + E, hW[t], zeroed=True)
                ops.attn_step_fwd(Uf, hW[t], feats, attw[t], ctxs[t])
                ops.gemm_nt(ctxs[t], Wc.data_ptr() + E * fs, E, E, 2 * E, X[t], residual=Xe[t], zeroed=True)
                inp = X[t]
                for l in range(NL):
                    wi, wh, bi, bh = dec.lstm.layer(l)
                    G = Gall[t, l]
                    ops.gemm_nt(inp, wi.data_ptr(), 4 * H, wi.shape[1], wi.shape[1], G, zeroed=True)
                    if hp(l, t) is not None:
                        ops.gemm_nt(hp(l, t), wh.data_ptr(), 4 * H, H, H, G, accumulate=True)
                    ops.lstm_cell_fwd(G, bi, bh, cp(l, t), Gates[l, t] if keep else None, Call[l, t], Hall[l, t])
                    inp = Hall[l, t]
                    if p_lstm > 0 and l < NL - 1:
                        ops.dropout(Hall[l, t], Hd[l, t], p_lstm, seeds[l][t])
                        inp = Hd[l, t]
        Hs = Hall[NL - 1]                                                                       # (T,B,H) contiguous
        Z = ops.linear_fwd(Hs.view(T * B, H), W1, b1, act=ACT_RELU)
        Zd, seed_z = Z, 0
        if p_drop > 0:
            seed_z = hnn._next_seed()
            Zd = torch.empty_like(Z)
            ops.dropout(Z, Zd, p_drop, seed_z)
        logits = ops.linear_fwd(Zd, W2, b2).view(T, B, V)
        if keep:
            ctx.dec, ctx.dims = dec, (T, B, P, E, H, NL, V)
            ctx.saved = dict(feats=feats, captions=captions, Uf=Uf, emb=emb, Hall=Hall, Call=Call, Gates=Gates, Hd=Hd, hW=hW,
                             attw=attw, ctxs=ctxs, X=X, Z=Z, Zd=Zd, seed_z=seed_z, p_drop=p_drop, p_lstm=p_lstm, seeds=seeds,
                             fused=fused, h0=h0, c0=c0)
        hs_out = Hs
        ctx.mark_non_differentiable(attw)
        return logits, hs_out, attw

    @staticmethod
    def backward(ctx, dlogits, dHs_ext, _dattw):
        dec = ctx.dec
        T, B, P, E, H, NL, V = ctx.dims
        s = ctx.saved
        dev = s["feats"].device
        fs = 4
        h0, c0 = s["h0"], s["c0"]
        Wa, Wc = dec.attention.weight, dec.attention_combine.weight
        W1, W2 = dec.output_projection[0].weight, dec.output_projection[3].weight
        b1, b2 = dec.output_projection[0].bias, dec.output_projection[3].bias
        gb = hnn.grad_buf
        Hall, Call, Gates, Hd = s["Hall"], s["Call"], s["Gates"], s["Hd"]
        Hs = Hall[NL - 1]
        # ---- vocabulary projection, batched over T*B rows
        dHs = None
        if dlogits is not None:
            dl = hnn._c(dlogits).view(T * B, V)
            dZd = ops.linear_bwd_data(dl, W2)
            if W2.requires_grad:
                ops.linear_bwd_weight(dl, s["Zd"], gb(W2))
                ops.colsum_into(dl, gb(b2))
            if s["p_drop"] > 0:
                ops.dropout(dZd, dZd, s["p_drop"], s["seed_z"])
            dZ = ops.relu_bwd(dZd, s["Z"])
            if W1.requires_grad:
                ops.linear_bwd_weight(dZ, Hs.view(T * B, H), gb(W1))
                ops.colsum_into(dZ, gb(b1))
            dHs = ops.empty(T, B, H, device=dev)
            ops.gemm_nn(dZ, W1.data_ptr(), E, H, H, dHs, residual=hnn._c(dHs_ext) if dHs_ext is not None else None)
        elif dHs_ext is not None:
            dHs = hnn._c(dHs_ext)
        else:
            dHs = ops.zeros(T, B, H, device=dev)
        # ---- BPTT
        DG = ops.empty(NL, T, B, 4 * H, device=dev)
        dUf = ops.zeros(B, P, E, device=dev)
        dfeats = ops.zeros(B, P, E, device=dev)
        dhW = ops.empty(T, B, E, device=dev)
        if s["fused"]:
            # L + 1 launches per token: stage G per layer (top first), stage Z per image (csrc/decoder_fused.hip)
            dX = ops.empty(T, B, E, device=dev)
            WT = []
            for l in range(NL):                               # [W_hh | W_ih]^T, once per step
                wi, wh, _, _ = dec.lstm.layer(l)
                wt = ops.empty(H + wi.shape[1], 4 * H, device=dev)
                ops.transpose2d(wh, wt[:H])
                ops.transpose2d(wi, wt[H:])
                WT.append(wt)
            carry_h = [ops.empty(B, H, device=dev) for _ in range(NL)]
            carry_c = [ops.empty(B, H, device=dev) for _ in range(NL)]
            top = NL - 1

            def cell(l, t, first):
                return dict(carry_h=None if first else carry_h[l], carry_c=carry_c[l], gates=Gates[l, t], c=Call[l, t],
                            c_prev=Call[l, t - 1] if t > 0 else (c0[l] if c0 is not None else None), dG=DG[l, t], first=first)

            ops.dec_attn_x_bwd(None, Wc, None, s["Uf"], None, s["feats"], dUf, dfeats, None, Wa, dHs[T - 1], cell(top, T - 1, True))
            for t in range(T - 1, -1, -1):
                for l in range(top, -1, -1):
                    ops.lstm_layer_bwd(DG[l, t], WT[l], carry_h[l] if t > 0 else None, dX[t] if l == 0 else None,
                                       cell(l - 1, t, t == T - 1) if l > 0 else None, s["p_lstm"],
                                       s["seeds"][l - 1][t] if (l > 0 and s["p_lstm"] > 0) else 0)
                ops.dec_attn_x_bwd(dX[t], Wc, s["attw"][t], s["Uf"], s["hW"][t], s["feats"], dUf, dfeats, dhW[t], Wa,
                                   dHs[t - 1] if t > 0 else None, cell(top, t - 1, False) if t > 0 else {})
        else:
            dX = ops.zeros(T, B, E, device=dev)             # zero-filled arenas for the split-K GEMV-sized products
            carry = ops.zeros(T, NL, B, H, device=dev)      # carry[t, l]: dL/dh_l flowing INTO step t from step t+1
            carry_c = [ops.zeros(B, H, device=dev) for _ in range(NL)]
            d_inp = ops.zeros(T, NL, B, H, device=dev)
            dctx = ops.zeros(T, B, E, device=dev)
            for t in range(T - 1, -1, -1):
                for l in range(NL - 1, -1, -1):
                    wi, wh, _, _ = dec.lstm.layer(l)
                    dh_a = dHs[t] if l == NL - 1 else d_inp[t, l + 1]
                    ops.lstm_cell_bwd(dh_a, carry[t, l] if t < T - 1 else None, carry_c[l] if t < T - 1 else None, Gates[l, t],
                                      Call[l, t], Call[l, t - 1] if t > 0 else (c0[l] if c0 is not None else None), DG[l, t],
                                      carry_c[l])
                    if t > 0:
                        ops.gemm_nn(DG[l, t], wh.data_ptr(), 4 * H, H, H, carry[t - 1, l], zeroed=True)
                    if l > 0:
                        ops.gemm_nn(DG[l, t], wi.data_ptr(), 4 * H, H, H, d_inp[t, l], zeroed=True)
                        if s["p_lstm"] > 0:
                            ops.dropout(d_inp[t, l], d_inp[t, l], s["p_lstm"], s["seeds"][l - 1][t])
                    else:
                        ops.gemm_nn(DG[0, t], wi.data_ptr(), 4 * H, E, E, dX[t], zeroed=True)
                ops.gemm_nn(dX[t], Wc.data_ptr() + E * fs, E, E, 2 * E, dctx[t], zeroed=True)
                ops.attn_step_bwd(dctx[t], s["attw"][t], s["Uf"], s["hW"][t], s["feats"], dUf, dfeats, dhW[t])
                if t > 0:
                    ops.gemm_nn(dhW[t], Wa.data_ptr(), E, H, H + E, carry[t - 1, NL - 1], accumulate=True)
        # ---- deferred weight gradients, batched over all steps
        for l in range(NL):
            wi, wh, bi, bh = dec.lstm.layer(l)
            if not wi.requires_grad:
                continue
            inp_all = s["X"] if l == 0 else (Hd[l - 1] if Hd is not None else Hall[l - 1])
            k_in = wi.shape[1]
            ops.gemm_tn_acc(DG[l].view(T * B, 4 * H), inp_all.view(T * B, k_in), gb(wi).data_ptr(), 4 * H, k_in, k_in)
            if T > 1:
                ops.gemm_tn_acc(DG[l, 1:].reshape((T - 1) * B, 4 * H), Hall[l, :-1].reshape((T - 1) * B, H), gb(wh).data_ptr(),
                                4 * H, H, H)
            else:
                gb(wh)
            if h0 is not None:                                  # the caller's initial state feeds step 0's recurrent product
                ops.gemm_tn_acc(DG[l, 0], h0[l], gb(wh).data_ptr(), 4 * H, H, H)
            ops.colsum_into(DG[l].view(T * B, 4 * H), gb(bi))
            ops.colsum_into(DG[l].view(T * B, 4 * H), gb(bh))
        dX2 = dX.view(T * B, E)
        if Wc.requires_grad:
            gWc = gb(Wc)
            ops.gemm_tn_acc(dX2, s["emb"].view(T * B, E), gWc.data_ptr(), E, E, 2 * E)
            ops.gemm_tn_acc(dX2, s["ctxs"].view(T * B, E), gWc.data_ptr() + E * fs, E, E, 2 * E)
            ops.colsum_into(dX2, gb(dec.attention_combine.bias))
        if dec.embedding.weight.requires_grad:
            demb = ops.gemm_nn(dX2, Wc.data_ptr(), E, E, 2 * E, ops.empty(T * B, E, device=dev))
            ops.embedding_bwd(s["captions"], demb, gb(dec.embedding.weight))
        dUf2 = dUf.view(B * P, E)
        feats2 = s["feats"].view(B * P, E)
        if Wa.requires_grad:
            gWa = gb(Wa)
            if T > 1:
                ops.gemm_tn_acc(dhW[1:].reshape((T - 1) * B, E), Hs[:-1].reshape((T - 1) * B, H), gWa.data_ptr(), E, H, H + E)
            if h0 is not None:
                ops.gemm_tn_acc(dhW[0], h0[NL - 1], gWa.data_ptr(), E, H, H + E)
            ops.gemm_tn_acc(dUf2, feats2, gWa.data_ptr() + H * fs, E, E, H + E)
            ops.colsum_into(dUf2, gb(dec.attention.bias))
        ops.gemm_nn(dUf2, Wa.data_ptr() + H * fs, E, E, H + E, dfeats.view(B * P, E), accumulate=True)
        dh0 = dc0 = None
        if h0 is not None and (ctx.needs_input_grad[4] or ctx.needs_input_grad[5]):
            dh0 = ops.zeros(NL, B, H, device=dev)
            for l in range(NL):
                _, wh, _, _ = dec.lstm.layer(l)
                ops.gemm_nn(DG[l, 0], wh.data_ptr(), 4 * H, H, H, dh0[l], zeroed=True)
            ops.gemm_nn(dhW[0], Wa.data_ptr(), E, H, H + E, dh0[NL - 1], accumulate=True)
            carries = carry_c                                       # after step 0 each holds dL/dc(-1) = dL/dc0
            dc0 = torch.stack([hnn._c(cc) for cc in carries], 0)
        ctx.saved = None
        return (dfeats, None, None, None, dh0, dc0) + (None,) * (len(ctx.needs_input_grad) - 6)


class LSTMDecoder(nn.Module):
    """reference: LSTMDecoder, /root/reference/src/student_model.py:121-256."""

    def __init__(self, vocab_size, embed_size=256, hidden_size=512, num_layers=2, dropout=0.2):
        super().__init__()
        self.embed_size, self.hidden_size, self.num_layers, self.vocab_size = embed_size, hidden_size, num_layers, vocab_size
        self.embedding = hnn.Embedding(vocab_size, embed_size)
        nn.init.uniform_(self.embedding.weight, -0.1, 0.1)
        self.attention = hnn.Linear(hidden_size + embed_size, embed_size)
        self.attention_combine = hnn.Linear(embed_size * 2, embed_size)
        self.lstm = _LSTMParams(embed_size, hidden_size, num_layers, dropout if num_layers > 1 else 0)
        self.output_projection = _FusedSeq(hnn.Linear(hidden_size, embed_size), nn.ReLU(), nn.Dropout(dropout),
                                           hnn.Linear(embed_size, vocab_size))

    def init_hidden(self, batch_size, device):
        return (ops.zeros(self.num_layers, batch_size, self.hidden_size, device=device),
                ops.zeros(self.num_layers, batch_size, self.hidden_size, device=device))

    @torch.no_grad()
    def attention_mechanism(self, hidden, image_features):
        """(context (B,E), attention_weights (B,L)) — reference :173-203; inference entry (used by caption_image-style
        callers); the training path goes through DecoderFn."""
        B, L, E = image_features.shape
        H = self.hidden_size
        feats = hnn._c(image_features)
        Wa = self.attention.weight
        Uf = ops.gemm_nt(feats.view(B * L, E), Wa.data_ptr() + H * 4, E, E, H + E, ops.empty(B * L, E, device=feats.device),
                         bias=self.attention.bias)
        hW = ops.gemm_nt(hnn._c(hidden), Wa.data_ptr(), E, H, H + E, ops.empty(B, E, device=feats.device))
        w = ops.empty(B, L, device=feats.device)
        ctx = ops.empty(B, E, device=feats.device)
        ops.attn_step_fwd(Uf, hW, feats, w, ctx)
        return ctx, w

    def forward(self, image_features, captions, hidden=None):
        """hidden: optional (h0, c0), each (num_layers, B, H), as in the reference (:205,:220); None = zero state."""
        params = [p for p in self.parameters() if p.requires_grad]
        h0, c0 = hidden if hidden is not None else (None, None)
        logits, hs, attw = DecoderFn.apply(image_features, captions, self, self.training, h0, c0, *params)
        return logits, list(hs.unbind(0)), list(attw.unbind(0))

    @torch.no_grad()
    def greedy(self, image_features, max_length=20, temperature=1.0, start_id=1):
        """Batched greedy decode on device, no per-token host sync: ids (max_length,B) int64 and logits
        (max_length,B,V).  Row b equals what the reference's B=1 caption_image (:339-379) computes for image b;
        truncation at <END> is the caller's (rows keep decoding independently)."""
        feats = hnn._c(image_features)
        B, P, E = feats.shape
        H, NL, V = self.hidden_size, self.num_layers, self.vocab_size
        dev = feats.device
        Wa, Wc = self.attention.weight, self.attention_combine.weight
        Uf = ops.gemm_nt(feats.view(B * P, E), Wa.data_ptr() + H * 4, E, E, H + E, ops.empty(B * P, E, device=dev),
                         bias=self.attention.bias).view(B, P, E)
        h = [ops.zeros(B, H, device=dev) for _ in range(NL)]
        c = [ops.zeros(B, H, device=dev) for _ in range(NL)]
        hn = [ops.empty(B, H, device=dev) for _ in range(NL)]
        cn = [ops.empty(B, H, device=dev) for _ in range(NL)]
        tok = torch.full((B,), start_id, dtype=torch.int64, device=dev)
        ids = torch.empty(max_length, B, dtype=torch.int64, device=dev)
        logits = ops.empty(max_length, B, V, device=dev)
        hW, w, cx = ops.empty(B, E, device=dev), ops.empty(B, P, device=dev), ops.empty(B, E, device=dev)
        xe, x, G, z = ops.empty(B, E, device=dev), ops.empty(B, E, device=dev), ops.empty(B, 4 * H, device=dev), ops.empty(B, E, device=dev)
        W1, W2 = self.output_projection[0], self.output_projection[3]
        fused = _fused_step_ok(E, H)
        for t in range(max_length):
            emb = ops.embedding_fwd(tok, self.embedding.weight)
            ops.gemm_nt(emb, Wc.data_ptr(), E, E, 2 * E, xe, bias=self.attention_combine.bias)
            if fused:
                ops.dec_attn_x_fwd(h[NL - 1], Wa, Uf, feats, Wc, xe, hW, w, cx, x)
            else:
                ops.gemm_nt(h[NL - 1], Wa.data_ptr(), E, H, H + E, hW)
                ops.attn_step_fwd(Uf, hW, feats, w, cx)
                ops.gemm_nt(cx, Wc.data_ptr() + E * 4, E, E, 2 * E, x, residual=xe)
            inp = x
            for l in range(NL):
                wi, wh, bi, bh = self.lstm.layer(l)
                if fused:
                    ops.lstm_layer_fwd(inp, h[l], wi, wh, bi, bh, c[l], None, cn[l], hn[l])
                else:
                    ops.gemm_nt(inp, wi.data_ptr(), 4 * H, wi.shape[1], wi.shape[1], G)
                    ops.gemm_nt(h[l], wh.data_ptr(), 4 * H, H, H, G, accumulate=True)
                    ops.lstm_cell_fwd(G, bi, bh, c[l], None, cn[l], hn[l])
                h[l], hn[l] = hn[l], h[l]
                c[l], cn[l] = cn[l], c[l]
                inp = h[l]
            ops.linear_fwd(h[NL - 1], W1.weight, W1.bias, act=ACT_RELU, out=z)
            ops.linear_fwd(z, W2.weight, W2.bias, out=logits[t])
            # temperature > 0 rescales logits uniformly: argmax is unchanged (reference :366-369)
            tok = ops.argmax_rows(logits[t])
            ids[t] = tok
        return ids, logits


class CaptioningStudent(nn.Module):
    """reference: CaptioningStudent, /root/reference/src/student_model.py:259-381."""

    def __init__(self, vocab_size, embed_size=256, hidden_size=512, num_layers=2, dropout=0.2,
                 use_attention_refinement=True):
        super().__init__()
        self.vocab_size, self.embed_size, self.hidden_size = vocab_size, embed_size, hidden_size
        self.encoder = CNNEncoder(embed_size=embed_size, fine_tune=True)
        self.use_attention_refinement = use_attention_refinement
        if use_attention_refinement:
            self.attention_refinement = AttentionRefinement(embed_size=embed_size)
        self.decoder = LSTMDecoder(vocab_size=vocab_size, embed_size=embed_size, hidden_size=hidden_size,
                                   num_layers=num_layers, dropout=dropout)

    def forward(self, images, captions):
        """-> (outputs (T,B,V), encoder_features (B,49,E) UN-refined (:301,:312), hidden_states list[T] of (B,H),
        attention_weights list[T] of (B,49))."""
        encoder_features = self.encoder(images)
        refined = self.attention_refinement(encoder_features) if self.use_attention_refinement else encoder_features
        outputs, hidden_states, attention_weights = self.decoder(refined, captions)
        return outputs, encoder_features, hidden_states, attention_weights

    @torch.no_grad()
    def generate(self, images, max_length=20, temperature=1.0, start_id=1):
        """Batched greedy decode (the build's batched equivalent of caption_image): ids (max_length,B), logits."""
        was = self.training
        self.eval()
        try:
            f = self.encoder(images)
            if self.use_attention_refinement:
                f = self.attention_refinement(f)
            return self.decoder.greedy(f, max_length, temperature, start_id)
        finally:
            self.train(was)

    def caption_image(self, image, vocabulary, max_length=20, temperature=1.0):
        """Greedy caption of ONE image as a list of words (reference :314-381; sets eval mode like the reference)."""
        self.eval()
        device = next(self.parameters()).device
        if image.dim() == 3:
            image = image.unsqueeze(0)
        start = vocabulary.stoi.get("<START>", vocabulary.stoi["<UNK>"])
        ids, _ = self.generate(image.to(device), max_length, temperature, start)
        words = []
        for i in ids[:, 0].tolist():       # one device->host copy for the whole caption
            if vocabulary.itos[i] == "<END>":
                break
            words.append(vocabulary.itos[i])
        return words


def count_parameters(model):
    total = sum(p.numel() for p in model.parameters())
    trainable = sum(p.numel() for p in model.parameters() if p.requires_grad)
    return total, trainable
