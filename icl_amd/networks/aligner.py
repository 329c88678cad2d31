"""The two train-only aligner heads of ICL (``sspa`` / ``uscl``): multi-scale prototype cross-attention.

Mirror of ``InherentConsistent`` / ``Class_Decoder`` / ``Query_Attention`` / ``MLP`` / ``SeparableConv3d``
(/root/reference/code/networks/unet_3D_icl.py:155-345) with identical parameter names, built on
icl_amd.ops.  Quirks reproduced on purpose: reshape-based head split, pre-softmax
logits returned as the "attention", ``q + drop_path(q)`` doubling, token-axis ``mlp2``.
"""
from __future__ import annotations

import contextlib
import math
from typing import Sequence

import torch
import torch.nn as nn

from .. import ops
from .layers import BatchNormAct, Conv2d, Conv3d


class Linear(nn.Module):
    def __init__(self, fin, fout, bias=True, device=None):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(fout, fin, device=device))
        self.bias = nn.Parameter(torch.empty(fout, device=device)) if bias else None
        b = 1.0 / math.sqrt(fin)
        with torch.no_grad():
            self.weight.uniform_(-b, b)
            if self.bias is not None:
                self.bias.uniform_(-b, b)

    def forward(self, x):
        return ops.linear(x, self.weight, self.bias, owner=self)


class LayerNorm(nn.Module):
    def __init__(self, dim, device=None):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(dim, device=device))
        self.bias = nn.Parameter(torch.zeros(dim, device=device))

    def forward(self, x):
        return ops.layer_norm(x, self.weight, self.bias, 1e-5)


class BatchNorm3d(BatchNormAct):
    """nn.BatchNorm3d parameters/buffers; forward fused with the following ReLU.  A ``BatchNormAct`` (act = ReLU), so the
    trainer's deferred step counters (one multi-tensor add per step instead of one launch per layer) cover the aligners too."""

    def __init__(self, c, device=None):
        super().__init__(c, 1, device)


# which lane the guided call's level i takes: 1 + (i + rot) % 3 (rot 0: the lane of the own-query call's level i)
USCL_LANE_ROT = int(__import__("os").environ.get("ICL_USCL_LANE_ROT", "0"))


def _open_tail_gate(grad):
    opt = ops.FactoredGrads.fused_optimizer
    if opt is not None and getattr(opt, "update_placement", "") in ("tail", "deep"):
        opt.flush_deferred(gate=True, only="tail")      # ("deep" entries wait for the backbone's gate, unet_3D._open_update_gate)
    return None


class DropPath(nn.Module):
    """Per-sample stochastic depth (MONAI DropPath semantics)."""

    def __init__(self, p=0.0):
        super().__init__()
        self.drop_prob = p

    def forward(self, x):
        if self.drop_prob == 0.0 or not self.training:
            return x
        return ops.drop_path(x, self.drop_prob)

    def add(self, res, x):
        """res + self(x) as one kernel."""
        return ops.drop_path_add(res, x, self.drop_prob, self.training)


class MLP(nn.Module):
    """fc2(GELU_erf(fc1(x))) — unet_3D_icl.py:299-315."""

    def __init__(self, fin, hidden, device=None):
        super().__init__()
        self.fc1 = Linear(fin, hidden, device=device)
        self.fc2 = Linear(hidden, fin, device=device)

    def forward(self, x):
        return self.fc2(ops.gelu(self.fc1(x)))


class Query_Attention(nn.Module):  # noqa: N801
    """unet_3D_icl.py:270-297."""

    def __init__(self, dim, num_heads, device=None):
        super().__init__()
        self.dim, self.num_heads = dim, num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.fc_q = Linear(dim, dim, device=device)
        self.fc_kv = Linear(dim, dim * 2, device=device)
        self.proj = Linear(dim, dim, device=device)

    def forward(self, q, x):
        B, N, C = x.shape
        nc, h = q.shape[1], self.num_heads
        qh = self.fc_q(q).reshape(B, h, nc, C // h)          # reshape quirk (:287)
        kv = self.fc_kv(x)                                    # [B, N, 2C] = [B, N, (k|v), h, d]
        out, logits = ops.prototype_attention(qh, kv, h, self.scale)   # out [B,h,nc,d], logits [B,nc,h,N] (= attn1.permute(0,2,1,3), :296)
        out = self.proj(out.reshape(B, nc, C))                # reshape quirk (:293)
        return out, logits


class Class_Decoder(nn.Module):  # noqa: N801
    """unet_3D_icl.py:244-268."""

    def __init__(self, dim, n_tokens, num_heads, drop_path=0.0, device=None):
        super().__init__()
        self.norm1 = LayerNorm(dim, device)
        self.norm1_query = LayerNorm(dim, device)
        self.attn = Query_Attention(dim, num_heads, device)
        self.drop_path = DropPath(drop_path)
        self.norm2 = LayerNorm(dim, device)
        self.mlp = MLP(dim, int(dim * 4.0), device)
        self.norm3 = LayerNorm(n_tokens, device)
        self.mlp2 = MLP(n_tokens, n_tokens, device)

    def attend(self, query, feat):
        """The query half (:258-264): attention, then the query's own residual MLP.  The returned logits feed ``refine``."""
        query, attn = self.attn(self.norm1_query(query), self.norm1(feat))
        query = self.drop_path.add(query, query)
        query = self.drop_path.add(query, self.mlp(self.norm2(query)))
        return query, attn

    def fused_ok(self, query, feat) -> bool:
        """The query half as fused stages (ops.query_attend, csrc/kernels/qchain.h): <= 32 query rows, C <= 256 (hidden layer <= 1024);
        anything attached to the modules it replaces (forward hooks, swapped children) keeps the operator-by-operator path."""
        a, m = self.attn, self.mlp
        mods = (self, self.norm1_query, self.norm2, a, a.fc_q, a.proj, m, m.fc1, m.fc2, self.drop_path)
        if any(x._forward_hooks or x._forward_pre_hooks for x in mods):
            return False
        if not (type(a) is Query_Attention and type(m) is MLP and type(self.norm1_query) is LayerNorm and type(self.norm2) is LayerNorm):
            return False
        return query.dim() == 3 and self.fused_ok_dims(feat.shape[0], query.shape[1], query.shape[0], feat)

    def fused_ok_dims(self, batch: int, nc: int, query_batch: int, like) -> bool:
        """The shape half of ``fused_ok`` (no hooks checked): ``batch`` samples x ``nc`` classes query rows, a query of ``query_batch``
        (1 = broadcast) samples."""
        return (query_batch in (1, batch)
                and ops.query_chain_ok(batch * nc, self.attn.dim, self.mlp.fc1.weight.shape[0], nc, like))

    def attend_fused(self, query, feat, qconv, ba, full=True, kv=None):
        """``attend`` + ``qconv`` on the fused stages.  ``query`` [B or 1, nc, C]; returns (logits, q[:ba], q[ba:], qconv(q)) — or
        (logits, None, None, None) with ``full`` False, for a caller that reads only the attention maps.  ``kv``: fc_kv(norm1(feat)) when
        the caller has computed it already (the token lane of forward_labeled_pair)."""
        a = self.attn
        if kv is None:
            kv = a.fc_kv(self.norm1(feat))
        p, training = self.drop_path.drop_prob, self.drop_path.training
        if full:
            s0, d0 = ops.drop_path_site(kv, p, training)
            s1, d1 = ops.drop_path_site(kv, p, training)
        else:
            s0 = s1 = (0, 0, 1.0)
            d0 = d1 = None
        params = (self.norm1_query.weight, self.norm1_query.bias, a.fc_q.weight, a.fc_q.bias, a.proj.weight, a.proj.bias,
                  self.norm2.weight, self.norm2.bias, self.mlp.fc1.weight, self.mlp.fc1.bias, self.mlp.fc2.weight, self.mlp.fc2.bias,
                  qconv.weight.squeeze(-1), qconv.bias)
        out = ops.query_attend(query, kv, a.num_heads, a.scale, ba, 1e-5, (s0, s1, d0 if d0 is not None else d1), full, params)
        if not full:
            return out[0], None, None, None
        return out

    def refine(self, attn):
        """The attention-map half (:265-267): residual token-axis MLP.  Needs nothing of the query half but the logits."""
        attn = self.drop_path.add(attn, attn)
        return self.drop_path.add(attn, self.mlp2(self.norm3(attn)))

    def forward(self, query, feat):
        query, attn = self.attend(query, feat)
        return query, self.refine(attn)


class _SepBlock(nn.Module):
    """The ``block`` Sequential of SeparableConv3d(relu_first=False) (unet_3D_icl.py:336-343)."""

    def __init__(self, planes, device=None, dims=3):
        super().__init__()
        conv = Conv3d if dims == 3 else Conv2d   # 2-D: SeparableConv2d, networks/unet_icl.py:98-126
        self.depthwise = conv(planes, planes, 3, bias=False, groups=planes, device=device)
        self.bn_depth = BatchNorm3d(planes, device)
        self.pointwise = conv(planes, planes, 1, bias=False, device=device)
        self.bn_point = BatchNorm3d(planes, device)

    def forward(self, x):
        x = self.bn_depth(self.depthwise(x))      # BN + ReLU fused
        return self.bn_point(self.pointwise(x))   # BN + ReLU fused


class SeparableConv3d(nn.Module):
    def __init__(self, planes, device=None, dims=3):
        super().__init__()
        self.block = _SepBlock(planes, device, dims)

    def forward(self, x):
        return self.block(x)


class Conv1d(nn.Module):
    """nn.Conv1d(cin, cout, 1) == per-token Linear (query_convs, unet_3D_icl.py:197)."""

    def __init__(self, cin, cout, device=None):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin, 1, device=device))
        self.bias = nn.Parameter(torch.empty(cout, device=device))
        b = 1.0 / math.sqrt(cin)
        with torch.no_grad():
            self.weight.uniform_(-b, b)
            self.bias.uniform_(-b, b)

    def forward(self, q):  # q [B, nc, C] -> [B, nc, C/2]; the reference permutes to [B,C,nc] and back
        return ops.linear(q, self.weight.squeeze(-1), self.bias)


def _batch_mean(t):
    """t.mean(dim=0, keepdim=True); the mean over a single sample is the sample itself (sum / 1, exact)."""
    return t if t.shape[0] == 1 else t.mean(dim=0, keepdim=True)


class InherentConsistent(nn.Module):
    """unet_3D_icl.py:155-242.  ``forward(feats, guided_Q=None, modal='labeled')`` keeps the reference
    call convention, including ``sspa(feats, 'labeled')`` passing the string positionally (:144-145)."""

    def __init__(self, in_chans: Sequence[int], depths=(2, 2, 2), patch_size=(2, 2, 2),
                 input_resolution: Sequence[int] = (6, 12, 24), num_classes: int = 2,
                 num_heads: Sequence[int] = (16, 8, 4), norm_layer=None, patch_norm=False,
                 spatial_dims: int = 3, drop_path_rate: float = 0.1, device=None, query_name: str = "guided_Q",
                 tokenized_input: bool = False):
        super().__init__()
        self._qname = query_name   # "guided_Q" in the U-Net files, "guide_Q" in swinunetr_icl.py:403
        # 2-D Swin-UNet (networks/vision_transformer.py:247-248): the decoder hands over TOKENS [B, N, C]; proj_layers and
        # norm_layers exist as parameters (state_dict parity, grad None) but are never called
        self.tokenized_input = tokenized_input
        self.in_chans, self.depth = tuple(in_chans), tuple(depths)
        self.resolutions = tuple(input_resolution)
        self.dims = spatial_dims   # 3: unet_3D_icl.py:155-242; 2: unet_icl.py:253-340 (r^2 tokens, Conv2d/BatchNorm2d)
        conv = Conv3d if spatial_dims == 3 else Conv2d
        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, sum(depths))]
        self.proj_layers = nn.ModuleList()
        self.norm_layers = nn.ModuleList()
        self.class_decoders = nn.ModuleList()
        self.attn_convs0 = nn.ModuleList()
        self.attn_convs1 = nn.ModuleList()
        self.query_convs = nn.ModuleList()
        for i in range(len(depths)):
            c, r, h = in_chans[i], input_resolution[i], num_heads[i]
            self.proj_layers.append(conv(c, c, 1, device=device))
            self.norm_layers.append(LayerNorm(c, device))
            self.class_decoders.append(Class_Decoder(c, r ** spatial_dims, h, drop_path=dpr[1], device=device))
            self.attn_convs0.append(SeparableConv3d(h, device, spatial_dims))
            self.attn_convs1.append(conv(h, 1, 1, device=device))
            self.query_convs.append(Conv1d(c, c // 2, device))
        self.register_parameter(query_name, nn.Parameter(torch.zeros(1, num_classes, in_chans[0], device=device)))

    def _tokens(self, i, feat):
        """``norm_layers[i](proj_layers[i](feat).flatten(2).transpose(1, 2))`` (:212): a 1x1x1 convolution followed by the token
        transpose is one GEMM on the channel axis, ``feat^T W^T + b`` -> [B, N, C] — one batched product of the tiled fp32-MFMA kernel
        (csrc/kernels/gemm.h) that reads the channel-major map in place (k-strided A operand), no transpose copy."""
        if self.tokenized_input:
            return feat
        p = self.proj_layers[i]
        return self.norm_layers[i](ops.linear(feat.flatten(2).transpose(1, 2), p.weight.flatten(1), p.bias))

    def forward_labeled_pair(self, feats, ba):
        """``self(feats[:ba], 'labeled')`` and ``self(feats[ba:], 'labeled')`` in lock step (the two ``sspa`` calls of
        unet_3D_icl.py:144-145 share their weights); ``feats`` hold both inputs as one batch, first ``ba`` samples = input a.
        Everything up to the attention maps is per-sample, so both inputs go through the token projection / Class_Decoder
        as ONE batch: the 13,824^2 ``mlp2`` weights (1.5 GB per call) are streamed twice per step instead of three times
        and their gradient is produced by one GEMM instead of two plus an accumulation.  The BatchNorm layers of
        ``attn_convs0`` (batch statistics) and the per-call batch means ``updated_Qs`` are still evaluated per input, in
        the reference order (a, then b)."""
        bs = feats[0].shape[0]
        maps_a, maps_b, qs_a, qs_b, branches = [], [], [], [], []
        nxt = getattr(self, self._qname)      # [1, nc, C]: broadcast over the batch (an expand, or inside the fused first stage)
        # the token side of every level (projection, LayerNorms, fc_kv: a function of the feature maps alone) on the token lane, beside
        # the query chain that links the levels (ops.SideStream.token_lane_on)
        kvs = [None] * len(self.depth)
        tok_s = ops.SideStream.token_stream() if feats[0].is_cuda else None
        if (tok_s is not None and all(not (m._forward_hooks or m._forward_pre_hooks) for cd in self.class_decoders for m in (cd.norm1, cd.attn.fc_kv))
                and all(cd.fused_ok(nxt[:, :, :1].expand(-1, -1, cd.attn.dim), feats[0]) for cd in self.class_decoders)
                and not any(q._forward_hooks or q._forward_pre_hooks for q in self.query_convs)):
            cur = torch.cuda.current_stream(feats[0].device)
            with torch.cuda.stream(tok_s):
                for i in range(len(self.depth)):
                    feats[i].record_stream(tok_s)
                    cd = self.class_decoders[i]
                    kvs[i] = cd.attn.fc_kv(cd.norm1(self._tokens(i, feats[i])))
            cur.wait_stream(tok_s)
            for kv in kvs:
                kv.record_stream(cur)
        for i in range(len(self.depth)):
            cd = self.class_decoders[i]
            tok = self._tokens(i, feats[i]) if kvs[i] is None else None
            like = tok if tok is not None else kvs[i]
            fused = cd.fused_ok(nxt, like) and not (self.query_convs[i]._forward_hooks or self.query_convs[i]._forward_pre_hooks)
            if fused:
                attn, qa, qb, nxt_i = cd.attend_fused(nxt, tok, self.query_convs[i], ba, kv=kvs[i])
            elif tok is None:
                raise RuntimeError("aligner: the token lane computed k / v for a level that does not take the fused query chain")
            else:
                q_out, attn = cd.attend(nxt if nxt.shape[0] == bs else nxt.expand(bs, -1, -1), tok)
            if i == len(self.depth) - 1 and attn.requires_grad:
                # backward: when the logits' gradient of the LAST level arrives, that level's map chain (the two big weight streams) is done
                # and what remains of this aligner is the serial query chain — the gate of FusedSGD's "tail" update placement
                attn.register_hook(_open_tail_gate)
            # the map chain of this level (token-axis MLP + separable convolutions) does not feed the next level: its own lane
            with ops.SideStream([attn], lane=(1 + i) if ops.SideStream.lane_mask & 1 else 99) as lane:
                attn = self.class_decoders[i].refine(attn)
                b, nc, h, n = attn.shape
                r = self.resolutions[i]
                sp = (r,) * self.dims
                a = attn.contiguous().view(b, nc, h, *sp)
                for part, maps in zip(ops.split_batch(a, ba), (maps_a, maps_b)):
                    pb = part.shape[0]
                    m = self.attn_convs1[i](self.attn_convs0[i](part.reshape(pb * nc, h, *sp)))
                    maps.append(m.reshape(pb, nc, *sp))
            branches.append((lane, [maps_a[-1], maps_b[-1]]))
            if fused:
                nxt = nxt_i
            else:
                nxt = self.query_convs[i](q_out)
                qa, qb = ops.split_batch(q_out, ba)
            qs_a.append(_batch_mean(qa))
            qs_b.append(_batch_mean(qb))
        for lane, outs in branches:
            lane.join(outs)
        return (maps_a, qs_a), (maps_b, qs_b)

    def forward(self, feats, guided_Q=None, modal="labeled", need_queries: bool = True):
        """``need_queries`` False (not a reference argument; the 3-D ICL models pass it for the guided call, whose updated queries they
        discard, unet_3D_icl.py:147): the second return value is a list of None and the query half behind the attention maps — proj, the
        query MLP, query_convs, none of which reaches an output — is not evaluated."""
        bs = feats[0].shape[0]
        feat_maps, updated_qs, branches = [], [], []
        labeled = modal == "labeled"
        nxt = getattr(self, self._qname) if labeled else None
        for i in range(len(self.depth)):
            # guided queries (unet_3D_icl.py:224-239; its unused next_guided_Q is not computed): the levels do not depend on each other at all — a whole level per lane; own queries:
            # only the map chain leaves the current stream (see forward_labeled_pair)
            whole = None if labeled else ops.SideStream([feats[i], guided_Q[i]], lane=(1 + (i + USCL_LANE_ROT) % 3) if ops.SideStream.lane_mask & 2 else 99)
            with (whole if whole is not None else contextlib.nullcontext()):
                tok = self._tokens(i, feats[i])
                q_in = nxt if labeled else guided_Q[i]
                cd = self.class_decoders[i]
                fused = cd.fused_ok(q_in, tok) and not (self.query_convs[i]._forward_hooks or self.query_convs[i]._forward_pre_hooks)
                if fused:
                    attn, q_out, _, nxt_i = cd.attend_fused(q_in, tok, self.query_convs[i], bs, full=labeled or need_queries)
                else:
                    q_out, attn = cd.attend(q_in if q_in.shape[0] == bs else q_in.expand(bs, -1, -1), tok)
                part = ops.SideStream([attn], lane=1 + i) if labeled else None
                with (part if part is not None else contextlib.nullcontext()):
                    attn = self.class_decoders[i].refine(attn)
                    b, nc, h, n = attn.shape
                    r = self.resolutions[i]
                    sp = (r,) * self.dims
                    a = attn.contiguous().view(b * nc, h, *sp)
                    a = self.attn_convs1[i](self.attn_convs0[i](a))
                    feat_maps.append(a.reshape(b, nc, *sp))
                if labeled:
                    nxt = nxt_i if fused else self.query_convs[i](q_out)
                updated_qs.append(_batch_mean(q_out) if q_out is not None else None)
            branches.append((whole or part, [feat_maps[-1]] + ([updated_qs[-1]] if whole is not None and updated_qs[-1] is not None else [])))
        for lane, outs in branches:
            lane.join(outs)
        return feat_maps, updated_qs
