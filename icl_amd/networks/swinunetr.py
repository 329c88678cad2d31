"""SwinUNETR on the HIP kernels — drop-in for the reference's ``SwinUNETR``
(/root/reference/code/networks/swinunetr.py:29-293; the same classes are vendored again in swinunetr_icl.py, whose line
numbers are cited below because the ICL model is the one on the hot path).

Module and attribute names reproduce the reference/MONAI-1.0.1 ``state_dict`` keys (``swinViT.layers1.0.blocks.0.attn.qkv.weight``,
``encoder1.layer.conv1.conv.weight``, ``decoder5.transp_conv.conv.weight``, ``out.conv.conv.weight`` ...), so reference
checkpoints and the self-supervised ``load_from`` weights load unchanged.

Data layout: the transformer stages keep tokens channels-last ``[B, D, H, W, C]`` (every Linear / LayerNorm is then a
row-major GEMM / row kernel with no transposes); the CNN encoder/decoder blocks are channels-first volumes for the MFMA
convolution kernels.  The five hidden states are converted once each (LayerNorm over C fused in front of the transpose).

Kept quirks (all exercised by tests against the reference goldens): the zero padding to a multiple of the window is
applied AFTER norm1 (:825-832); clipped windows index the 7^3 relative-position table with its top-left corner (:733-735);
PatchMerging concatenates slices 0,1,2,3,4,2,3,7 rather than the eight octants (:953-961).
"""
from __future__ import annotations

import itertools
from typing import Sequence

import numpy as np
import torch
import torch.nn as nn

from .. import ops
from .aligner import LayerNorm, Linear
from .layers import Conv3d

WINDOW = (7, 7, 7)


# ------------------------------------------------------------------------------------------------- windows
def get_window_size(x_size, window_size, shift_size=None):
    """swinunetr_icl.py:617-641."""
    ws = list(window_size)
    ss = list(shift_size) if shift_size is not None else None
    for i in range(len(x_size)):
        if x_size[i] <= window_size[i]:
            ws[i] = x_size[i]
            if ss is not None:
                ss[i] = 0
    return tuple(ws) if ss is None else (tuple(ws), tuple(ss))


def window_partition(x, ws):
    """[b,d,h,w,c] -> [b*nW, n, c]  (swinunetr_icl.py:552-582)."""
    b, d, h, w, c = x.shape
    x = x.view(b, d // ws[0], ws[0], h // ws[1], ws[1], w // ws[2], ws[2], c)
    return x.permute(0, 1, 3, 5, 2, 4, 6, 7).reshape(-1, ws[0] * ws[1] * ws[2], c)


def window_reverse(win, ws, dims):
    """swinunetr_icl.py:585-614."""
    b, d, h, w = dims
    x = win.view(b, d // ws[0], h // ws[1], w // ws[2], ws[0], ws[1], ws[2], -1)
    return x.permute(0, 1, 4, 2, 5, 3, 6, 7).reshape(b, d, h, w, -1)


_REGION_CACHE = {}


def window_regions(dims, ws, ss, device):
    """Region id of every token of every window of the rolled, padded volume: int32 [nW, n].  Two tokens of a window may
    attend to each other iff their ids are equal — the dense 0/-100 ``attn_mask`` of compute_mask
    (swinunetr_icl.py:979-1016) is ``-100 * (id_i != id_j)``; the ids are 343x smaller and are what the kernel reads.
    ``slice(-0, None)`` covering a whole unshifted axis is kept as written."""
    key = (tuple(dims), tuple(ws), tuple(ss), str(device))
    if key not in _REGION_CACHE:
        d, h, w = dims
        img = torch.zeros((1, d, h, w, 1), dtype=torch.int32)
        cnt = 0
        for sd, sh, sw in itertools.product(*[(slice(-ws[a]), slice(-ws[a], -ss[a]), slice(-ss[a], None)) for a in range(3)]):
            img[:, sd, sh, sw, :] = cnt
            cnt += 1
        _REGION_CACHE[key] = window_partition(img, ws).squeeze(-1).contiguous().to(device)
    return _REGION_CACHE[key]


_WINDOW_INDEX_CACHE = {}


def window_index(dims, ws, ss, device):
    """(to_windows [nW*n], to_tokens [D*H*W]) int32: slot m of the (zero-padded, rolled by -ss, window-partitioned) order holds
    token to_windows[m] of the volume (-1 for padding), and token s sits in slot to_tokens[s] — the composition of F.pad,
    torch.roll and window_partition of swinunetr_icl.py:825-845 (and its inverse :849-866) as one index each."""
    key = (tuple(dims), tuple(ws), tuple(ss), str(device))
    if key not in _WINDOW_INDEX_CACHE:
        d, h, w = dims
        pd, ph, pw = [(ws[a] - s % ws[a]) % ws[a] for a, s in enumerate(dims)]
        ids = torch.arange(d * h * w, dtype=torch.int32).view(1, d, h, w, 1)
        ids = torch.nn.functional.pad(ids, (0, 0, 0, pw, 0, ph, 0, pd), value=-1)
        if any(s > 0 for s in ss):
            ids = torch.roll(ids, shifts=(-ss[0], -ss[1], -ss[2]), dims=(1, 2, 3))
        to_win = window_partition(ids, ws).reshape(-1).contiguous()
        to_tok = torch.empty(d * h * w, dtype=torch.int32)
        valid = to_win >= 0
        to_tok[to_win[valid].long()] = torch.nonzero(valid).squeeze(1).to(torch.int32)
        _WINDOW_INDEX_CACHE[key] = (to_win.to(device), to_tok.to(device))
    return _WINDOW_INDEX_CACHE[key]


_MERGE_INDEX_CACHE = {}


def merge_index(dims, slices, device):
    """(idx [S], back [S, 2]) int32 for PatchMerging on an even (d, h, w) grid: output row ((d2*H2 + h2)*W2 + w2)*8 + slot reads
    token (2*d2 + i, 2*h2 + j, 2*w2 + k) with (i, j, k) = slices[slot] — the channel concatenation of swinunetr_icl.py:953-961
    written as a row gather.  ``back`` lists the (at most two) output rows reading each token, -1 where none does."""
    key = (tuple(dims), tuple(slices), str(device))
    if key not in _MERGE_INDEX_CACHE:
        d, h, w = dims
        ids = torch.arange(d * h * w, dtype=torch.int64).view(d, h, w)
        idx = torch.stack([ids[i::2, j::2, k::2] for i, j, k in slices], -1).reshape(-1)
        back = torch.full((d * h * w, 2), -1, dtype=torch.int64)
        order = torch.argsort(idx, stable=True)
        src = idx[order]
        first = torch.ones_like(src, dtype=torch.bool)
        first[1:] = src[1:] != src[:-1]
        back[src[first], 0] = order[first]
        second = ~first
        assert not (second[1:] & second[:-1]).any(), "a token is read by more than two merge slots"
        back[src[second], 1] = order[second]
        _MERGE_INDEX_CACHE[key] = (idx.to(torch.int32).to(device), back.to(torch.int32).contiguous().to(device))
    return _MERGE_INDEX_CACHE[key]


class WindowAttention(nn.Module):
    """swinunetr_icl.py:644-750."""

    def __init__(self, dim, num_heads, window_size, device=None):
        super().__init__()
        self.dim, self.num_heads, self.window_size = dim, num_heads, tuple(window_size)
        self.scale = (dim // num_heads) ** -0.5
        ws = self.window_size
        self.relative_position_bias_table = nn.Parameter(
            torch.zeros((2 * ws[0] - 1) * (2 * ws[1] - 1) * (2 * ws[2] - 1), num_heads, device=device))
        nn.init.trunc_normal_(self.relative_position_bias_table, std=0.02)
        coords = torch.stack(torch.meshgrid(*[torch.arange(w) for w in ws], indexing="ij")).flatten(1)
        rel = (coords[:, :, None] - coords[:, None, :]).permute(1, 2, 0).contiguous()
        for a in range(3):
            rel[:, :, a] += ws[a] - 1
        rel[:, :, 0] *= (2 * ws[1] - 1) * (2 * ws[2] - 1)
        rel[:, :, 1] *= 2 * ws[2] - 1
        self.register_buffer("relative_position_index", rel.sum(-1).to(device))
        self.qkv = Linear(dim, dim * 3, device=device)
        self.proj = Linear(dim, dim, device=device)

    def forward(self, x, regions):
        """x [B*nW, n, C]; regions int32 [nW, n] or None (unshifted block)."""
        out = ops.window_attention(self.qkv(x), self.relative_position_bias_table, self.relative_position_index, regions,
                                   self.num_heads, self.scale)
        return self.proj(out)


class MLPBlock(nn.Module):
    """MONAI MLPBlock(dropout 0): linear2(GELU(linear1(x)))."""

    def __init__(self, dim, hidden, device=None):
        super().__init__()
        self.linear1 = Linear(dim, hidden, device=device)
        self.linear2 = Linear(hidden, dim, device=device)

    def forward(self, x):
        return self.linear2(ops.gelu(self.linear1(x)))


class SwinTransformerBlock(nn.Module):
    """swinunetr_icl.py:753-916 (drop_path == identity: net_factory_3d.py passes dropout_path_rate 0.0)."""

    def __init__(self, dim, num_heads, window_size, shift_size, device=None):
        super().__init__()
        self.window_size, self.shift_size = tuple(window_size), tuple(shift_size)
        self.norm1 = LayerNorm(dim, device)
        self.attn = WindowAttention(dim, num_heads, window_size, device)
        self.norm2 = LayerNorm(dim, device)
        self.mlp = MLPBlock(dim, int(dim * 4.0), device)

    def forward_part1(self, x, regions):
        b, d, h, w, c = x.shape
        ws, ss = get_window_size((d, h, w), self.window_size, self.shift_size)
        y = self.norm1(x)
        shifted = any(s > 0 for s in ss)
        n = ws[0] * ws[1] * ws[2]
        # pad -> roll -> window_partition as ONE row gather (and window_reverse -> roll -> crop as its inverse)
        to_win, to_tok = window_index((d, h, w), ws, ss if shifted else (0, 0, 0), x.device)
        win = ops.gather_rows(y.view(b, d * h * w, c), to_win, to_tok)               # [b, nW*n, c]
        win = self.attn(win.view(-1, n, c), regions if shifted else None)
        return ops.gather_rows(win.view(b, -1, c), to_tok, to_win).view(b, d, h, w, c)

    def forward(self, x, regions):
        x = x + self.forward_part1(x, regions)
        return x + self.mlp(self.norm2(x))

    def load_from(self, weights, n_block, layer):
        """swinunetr_icl.py:871-903: self-supervised checkpoints name the MLP ``fc1``/``fc2``."""
        root = f"module.{layer}.0.blocks.{n_block}."
        sd = weights["state_dict"]
        with torch.no_grad():
            for k, t in itertools.chain(self.named_parameters(), self.named_buffers()):
                src = k.replace("mlp.linear1", "mlp.fc1").replace("mlp.linear2", "mlp.fc2")
                t.copy_(sd[root + src])


class PatchMerging(nn.Module):
    """swinunetr_icl.py:919-976."""

    SLICES = ((0, 0, 0), (1, 0, 0), (0, 1, 0), (0, 0, 1), (1, 0, 1), (0, 1, 0), (0, 0, 1), (1, 1, 1))

    def __init__(self, dim, device=None):
        super().__init__()
        self.reduction = Linear(8 * dim, 2 * dim, bias=False, device=device)
        self.norm = LayerNorm(8 * dim, device)

    def forward(self, x):
        b, d, h, w, c = x.shape
        if d % 2 or h % 2 or w % 2:   # :950-951, argument order as written (W padded by d%2, H by w%2, D by h%2)
            x = torch.nn.functional.pad(x, (0, 0, 0, d % 2, 0, w % 2, 0, h % 2))
        b, d, h, w, c = x.shape
        if c % 4:
            x = torch.cat([x[:, i::2, j::2, k::2, :] for i, j, k in self.SLICES], -1)
        else:   # the same concatenation as one row gather (and one gather-sum for its gradient)
            idx, back = merge_index((d, h, w), self.SLICES, x.device)
            x = ops.gather_rows_dup(x.reshape(b, d * h * w, c), idx, back).view(b, d // 2, h // 2, w // 2, 8 * c)
        return self.reduction(self.norm(x))


class BasicLayer(nn.Module):
    """swinunetr_icl.py:1019-1116.  Input and output are channels-last here; the reference's two ``rearrange`` calls
    per stage are folded into SwinTransformer.forward."""

    def __init__(self, dim, depth, num_heads, window_size, device=None):
        super().__init__()
        self.window_size = tuple(window_size)
        self.shift_size = tuple(i // 2 for i in window_size)
        no_shift = tuple(0 for _ in window_size)
        self.blocks = nn.ModuleList([
            SwinTransformerBlock(dim, num_heads, self.window_size, no_shift if i % 2 == 0 else self.shift_size, device)
            for i in range(depth)])
        self.downsample = PatchMerging(dim, device)

    def forward(self, x):
        b, d, h, w, c = x.shape
        ws, ss = get_window_size((d, h, w), self.window_size, self.shift_size)
        padded = [int(np.ceil(s / ws[a])) * ws[a] for a, s in enumerate((d, h, w))]
        regions = window_regions(padded, ws, ss, x.device) if any(s > 0 for s in ss) else None
        for blk in self.blocks:
            x = blk(x, regions)
        return self.downsample(x)


class PatchEmbed(nn.Module):
    """MONAI PatchEmbed: Conv3d(k=2, s=2) — non-overlapping patches, i.e. one GEMM on the gathered 2x2x2 patches."""

    def __init__(self, in_chans, embed_dim, device=None):
        super().__init__()
        self.proj = Conv3d(in_chans, embed_dim, 2, device=device)   # parameter holder (weight [E, Cin, 2,2,2], bias)

    def forward(self, x):
        b, c, d, h, w = x.shape
        assert d % 2 == 0 and h % 2 == 0 and w % 2 == 0
        p = x.view(b, c, d // 2, 2, h // 2, 2, w // 2, 2).permute(0, 2, 4, 6, 1, 3, 5, 7).reshape(b, d // 2, h // 2, w // 2, c * 8)
        return ops.linear(p, self.proj.weight.flatten(1), self.proj.bias)       # channels-last tokens


class SwinTransformer(nn.Module):
    """swinunetr_icl.py:1119-1235."""

    def __init__(self, in_chans, embed_dim, window_size, depths, num_heads, device=None):
        super().__init__()
        self.patch_embed = PatchEmbed(in_chans, embed_dim, device)
        for i in range(4):
            layer = BasicLayer(embed_dim * 2 ** i, depths[i], num_heads[i], window_size, device)
            setattr(self, f"layers{i + 1}", nn.ModuleList([layer]))

    @staticmethod
    def proj_out(x, normalize=False):
        """:1208-1221: channel LayerNorm without affine, returned channels-first for the convolution blocks."""
        if normalize:
            x = ops.layer_norm(x, None, None, 1e-5)
        return x.permute(0, 4, 1, 2, 3).contiguous()

    def forward(self, x, normalize=True):
        x0 = self.patch_embed(x)
        outs = [self.proj_out(x0, normalize)]
        cur = x0
        for i in range(4):
            cur = getattr(self, f"layers{i + 1}")[0](cur)
            outs.append(self.proj_out(cur, normalize))
        return outs


# ------------------------------------------------------------------------------------------------- MONAI CNN blocks
class _ConvOnly(nn.Module):
    """monai Convolution(conv_only=True): a Sequential with one child named ``conv``."""

    def __init__(self, cin, cout, ks, bias=False, device=None):
        super().__init__()
        self.conv = Conv3d(cin, cout, ks, bias=bias, device=device, feeds_instance_norm=not bias)

    def forward(self, x):
        return self.conv(x)


class _TransposedConvOnly(nn.Module):
    def __init__(self, cin, cout, device=None):
        super().__init__()
        self.conv = nn.Module()
        self.conv.weight = nn.Parameter(torch.empty(cin, cout, 2, 2, 2, device=device))
        bound = 1.0 / np.sqrt(cout * 8)     # nn.ConvTranspose3d default init: fan_in is taken from dim 1
        with torch.no_grad():
            self.conv.weight.uniform_(-bound, bound)

    def forward(self, x, skip=None):
        return ops.conv_transpose3d_k2s2(x, self.conv.weight, skip)


class UnetResBlock(nn.Module):
    """MONAI UnetResBlock(norm 'instance', LeakyReLU 0.01): conv3-IN-lrelu-conv3-IN (+ 1x1 conv-IN shortcut when the
    channel count changes) -> add -> lrelu.  Convolutions carry no bias."""

    def __init__(self, cin, cout, device=None):
        super().__init__()
        self.conv1 = _ConvOnly(cin, cout, 3, device=device)
        self.conv2 = _ConvOnly(cout, cout, 3, device=device)
        self.downsample = cin != cout
        if self.downsample:
            self.conv3 = _ConvOnly(cin, cout, 1, device=device)

    def forward(self, x):
        hooked = any(m._forward_hooks or m._forward_pre_hooks for m in (self.conv1, self.conv2, self.conv1.conv, self.conv2.conv))
        if hooked or type(self.conv1) is not _ConvOnly or type(self.conv2) is not _ConvOnly:
            # children with hooks or swapped children: the unfused composition, module by module
            out = ops.instance_norm_act(self.conv1(x), act=2)
            res = ops.instance_norm_act(self.conv3(x), act=0) if self.downsample else x
            return ops.instance_norm_add_act(self.conv2(out), res, act=2)
        # conv -> InstanceNorm -> act as one operator each: the convolution's epilogue hands the normalisation its statistics (ops.py)
        out = ops.conv3d_instance_norm_act(x, self.conv1.conv.weight, None, act=2)
        res = ops.instance_norm_act(self.conv3(x), act=0) if self.downsample else x
        return ops.conv3d_instance_norm_add_act(out, self.conv2.conv.weight, res, act=2)


class UnetrBasicBlock(nn.Module):
    def __init__(self, cin, cout, device=None):
        super().__init__()
        self.layer = UnetResBlock(cin, cout, device)

    def forward(self, x):
        return self.layer(x)


class UnetrUpBlock(nn.Module):
    """MONAI UnetrUpBlock: ConvTranspose3d(k2, s2) -> cat((up, skip), 1) -> UnetResBlock(2*cout -> cout)."""

    def __init__(self, cin, cout, device=None):
        super().__init__()
        self.transp_conv = _TransposedConvOnly(cin, cout, device)
        self.conv_block = UnetResBlock(2 * cout, cout, device)

    def forward(self, x, skip):
        return self.conv_block(self.transp_conv(x, skip))      # [up | skip] built in one buffer


class UnetOutBlock(nn.Module):
    def __init__(self, cin, cout, device=None):
        super().__init__()
        self.conv = _ConvOnly(cin, cout, 1, bias=True, device=device)

    def forward(self, x):
        return self.conv(x)


class SwinUNETRBackbone(nn.Module):
    """Parameters and one-stream forward shared by SwinUNETR and SwinUNETR_icl (swinunetr_icl.py:37-232,313-327)."""

    def __init__(self, img_size, in_channels, out_channels, depths=(2, 2, 2, 2), num_heads=(3, 6, 12, 24),
                 feature_size=24, norm_name="instance", drop_rate=0.0, attn_drop_rate=0.0, dropout_path_rate=0.0,
                 normalize=True, use_checkpoint=False, spatial_dims=3, device=None):
        super().__init__()
        if spatial_dims != 3:
            raise ValueError("icl_amd SwinUNETR: only spatial_dims=3 (the trainers' configuration) is built")
        if isinstance(img_size, int):
            img_size = (img_size,) * 3
        for m in img_size:
            if m % 32 != 0:
                raise ValueError("input image size (img_size) should be divisible by stage-wise image resolution.")
        if feature_size % 12 != 0:
            raise ValueError("feature_size should be divisible by 12.")
        if drop_rate or attn_drop_rate or dropout_path_rate:
            raise NotImplementedError("SwinUNETR dropout / stochastic depth: the reference factory passes 0.0 for all three")
        self.img_size = tuple(img_size)
        self.normalize = normalize
        f = feature_size
        self.swinViT = SwinTransformer(in_channels, f, WINDOW, depths, num_heads, device)
        self.encoder1 = UnetrBasicBlock(in_channels, f, device)
        self.encoder2 = UnetrBasicBlock(f, f, device)
        self.encoder3 = UnetrBasicBlock(2 * f, 2 * f, device)
        self.encoder4 = UnetrBasicBlock(4 * f, 4 * f, device)
        self.encoder10 = UnetrBasicBlock(16 * f, 16 * f, device)
        self.decoder5 = UnetrUpBlock(16 * f, 8 * f, device)
        self.decoder4 = UnetrUpBlock(8 * f, 4 * f, device)
        self.decoder3 = UnetrUpBlock(4 * f, 2 * f, device)
        self.decoder2 = UnetrUpBlock(2 * f, f, device)
        self.decoder1 = UnetrUpBlock(f, f, device)
        self.out = UnetOutBlock(f, out_channels, device)

    def run_backbone(self, x, heads=None):
        """``heads`` (the ICL model's aligner calls) is invoked on [dec3, dec2, dec1] as soon as they exist; its result is returned as
        a third value."""
        hs = self.swinViT(x, self.normalize)
        enc0 = self.encoder1(x)
        enc1 = self.encoder2(hs[0])
        enc2 = self.encoder3(hs[1])
        enc3 = self.encoder4(hs[2])
        dec4 = self.encoder10(hs[4])
        dec3 = self.decoder5(dec4, hs[3])
        dec2 = self.decoder4(dec3, enc3)
        dec1 = self.decoder3(dec2, enc2)
        extra = heads([dec3, dec2, dec1]) if heads is not None else None
        dec0 = self.decoder2(dec1, enc1)
        out = self.decoder1(dec0, enc0)
        if heads is not None:
            return self.out(out), [dec3, dec2, dec1, dec0], extra
        return self.out(out), [dec3, dec2, dec1, dec0]

    def load_from(self, weights):
        """swinunetr_icl.py:260-308: copy a self-supervised Swin encoder checkpoint (keys ``module.<...>``)."""
        sd = weights["state_dict"]
        with torch.no_grad():
            self.swinViT.patch_embed.proj.weight.copy_(sd["module.patch_embed.proj.weight"])
            self.swinViT.patch_embed.proj.bias.copy_(sd["module.patch_embed.proj.bias"])
            for i in range(1, 5):
                layer = getattr(self.swinViT, f"layers{i}")[0]
                for bname, block in layer.blocks.named_children():
                    block.load_from(weights, n_block=bname, layer=f"layers{i}")
                for k in ("reduction.weight", "norm.weight", "norm.bias"):
                    mod, attr = k.split(".")
                    getattr(getattr(layer.downsample, mod), attr).copy_(sd[f"module.layers{i}.0.downsample.{k}"])


class SwinUNETR(SwinUNETRBackbone):
    def forward(self, x_in):
        return self.run_backbone(x_in)[0]
