"""2-D Swin-UNet backbone on the HIP kernels — drop-in for the reference's ``SwinTransformerSys``
(/root/reference/code/networks/swinunet_icl.py:605-809) with the same ``state_dict`` keys
(``patch_embed.proj.weight``, ``layers.0.blocks.1.attn_mask``, ``layers_up.1.upsample.expand.weight``, ``concat_back_dim.2.bias``,
``up.expand.weight``, ``output.weight`` ...).

Tokens stay ``[B, L, C]`` throughout (every Linear / LayerNorm is a row-major GEMM / row kernel); window attention runs on the
fused MFMA kernels with head dim 32 (csrc/kernels/winattn.h); the 4x4-stride-4 patch embedding and the 1x1 output convolution
are GEMMs on the token axis.  The dense 0/-100 ``attn_mask`` buffers are kept for checkpoint parity, the kernels read the
equivalent region ids.  Stochastic depth: timm ``DropPath`` semantics with the reference's linear schedule (drop_path_rate 0.2
in configs/swin_tiny_patch4_window7_224_lite.yaml).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import ops
from .aligner import DropPath, LayerNorm, Linear
from .layers import Conv2d


def window_partition(x, ws):
    """[B,H,W,C] -> [B*nW, ws*ws, C]  (swinunet_icl.py:33-47)."""
    b, h, w, c = x.shape
    return x.view(b, h // ws, ws, w // ws, ws, c).permute(0, 1, 3, 2, 4, 5).reshape(-1, ws * ws, c)


def window_reverse(win, ws, h, w):
    """swinunet_icl.py:50-65."""
    b = win.shape[0] // ((h // ws) * (w // ws))
    return win.view(b, h // ws, w // ws, ws, ws, -1).permute(0, 1, 3, 2, 4, 5).reshape(b, h, w, -1)


def _region_image(res, ws, shift):
    img = torch.zeros((1, res, res, 1))
    cnt = 0
    for hs in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
        for wsl in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
            img[:, hs, wsl, :] = cnt
            cnt += 1
    return img


class Mlp(nn.Module):
    def __init__(self, dim, hidden, device=None):
        super().__init__()
        self.fc1 = Linear(dim, hidden, device=device)
        self.fc2 = Linear(hidden, dim, device=device)

    def forward(self, x):
        return self.fc2(ops.gelu(self.fc1(x)))


class WindowAttention(nn.Module):
    """swinunet_icl.py:68-155."""

    def __init__(self, dim, window_size, num_heads, device=None):
        super().__init__()
        self.dim, self.window_size, self.num_heads = dim, window_size, num_heads
        self.scale = (dim // num_heads) ** -0.5
        ws = window_size
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * ws - 1) * (2 * ws - 1), num_heads, device=device))
        nn.init.trunc_normal_(self.relative_position_bias_table, std=0.02)
        coords = torch.stack(torch.meshgrid(torch.arange(ws), torch.arange(ws), indexing="ij")).flatten(1)
        rel = (coords[:, :, None] - coords[:, None, :]).permute(1, 2, 0).contiguous()
        rel[:, :, 0] += ws - 1
        rel[:, :, 1] += ws - 1
        rel[:, :, 0] *= 2 * ws - 1
        self.register_buffer("relative_position_index", rel.sum(-1).to(device))
        self.qkv = Linear(dim, dim * 3, device=device)
        self.proj = Linear(dim, dim, device=device)

    def forward(self, x, regions):
        out = ops.window_attention(self.qkv(x), self.relative_position_bias_table, self.relative_position_index, regions,
                                   self.num_heads, self.scale)
        return self.proj(out)


class SwinTransformerBlock(nn.Module):
    """swinunet_icl.py:174-293."""

    def __init__(self, dim, input_resolution, num_heads, window_size=7, shift_size=0, drop_path=0.0, device=None):
        super().__init__()
        self.input_resolution = tuple(input_resolution)
        self.window_size, self.shift_size = window_size, shift_size
        if min(self.input_resolution) <= self.window_size:      # :203-206
            self.shift_size = 0
            self.window_size = min(self.input_resolution)
        h0 = self.input_resolution[0]
        ids = torch.arange(h0 * h0, dtype=torch.int32).view(1, h0, h0, 1)
        if self.shift_size > 0:
            ids = torch.roll(ids, shifts=(-self.shift_size, -self.shift_size), dims=(1, 2))
        to_win = window_partition(ids, self.window_size).reshape(-1).contiguous()
        to_tok = torch.empty_like(to_win)
        to_tok[to_win.long()] = torch.arange(to_win.numel(), dtype=torch.int32)
        self.register_buffer("_to_win", to_win.to(device), persistent=False)   # roll + window_partition as one row gather
        self.register_buffer("_to_tok", to_tok.to(device), persistent=False)   # window_reverse + roll back
        attn_mask, regions = None, None
        if self.shift_size > 0:                                  # :222-245
            h = self.input_resolution[0]
            ids = window_partition(_region_image(h, self.window_size, self.shift_size), self.window_size).squeeze(-1)
            m = ids.unsqueeze(1) - ids.unsqueeze(2)
            attn_mask = m.masked_fill(m != 0, -100.0).masked_fill(m == 0, 0.0).to(device)
            regions = ids.to(torch.int32).contiguous().to(device)
        self.register_buffer("attn_mask", attn_mask)
        self.register_buffer("_regions", regions, persistent=False)
        self.norm1 = LayerNorm(dim, device)
        self.attn = WindowAttention(dim, self.window_size, num_heads, device)
        self.drop_path = DropPath(drop_path)
        self.norm2 = LayerNorm(dim, device)
        self.mlp = Mlp(dim, int(dim * 4.0), device)

    def forward(self, x):
        h, w = self.input_resolution
        b, l, c = x.shape
        ws, ss = self.window_size, self.shift_size
        win = ops.gather_rows(self.norm1(x), self._to_win, self._to_tok)             # [b, nW*n, c] in window order
        win = self.attn(win.view(-1, ws * ws, c), self._regions if ss > 0 else None)
        y = ops.gather_rows(win.view(b, l, c), self._to_tok, self._to_win)
        x = self.drop_path.add(x, y)
        return self.drop_path.add(x, self.mlp(self.norm2(x)))


class PatchMerging(nn.Module):
    """swinunet_icl.py:314-351."""

    def __init__(self, input_resolution, dim, device=None):
        super().__init__()
        self.input_resolution = tuple(input_resolution)
        self.reduction = Linear(4 * dim, 2 * dim, bias=False, device=device)
        self.norm = LayerNorm(4 * dim, device)

    def forward(self, x):
        h, w = self.input_resolution
        b, l, c = x.shape
        x = x.view(b, h, w, c)
        x = torch.cat([x[:, 0::2, 0::2], x[:, 1::2, 0::2], x[:, 0::2, 1::2], x[:, 1::2, 1::2]], -1).view(b, -1, 4 * c)
        return self.reduction(self.norm(x))


class PatchExpand(nn.Module):
    """swinunet_icl.py:363-387 (dim_scale 2) and FinalPatchExpand_X4 :390-414 (dim_scale 4)."""

    def __init__(self, input_resolution, dim, dim_scale=2, device=None):
        super().__init__()
        self.input_resolution, self.dim_scale = tuple(input_resolution), dim_scale
        self.expand = Linear(dim, (2 if dim_scale == 2 else 16) * dim, bias=False, device=device)
        self.norm = LayerNorm(dim // 2 if dim_scale == 2 else dim, device)

    def forward(self, x):
        h, w = self.input_resolution
        s = self.dim_scale
        x = self.expand(x)
        b, l, c = x.shape
        co = c // (s * s)
        x = x.view(b, h, w, s, s, co).permute(0, 1, 3, 2, 4, 5).reshape(b, h * s * w * s, co)
        return self.norm(x)


class BasicLayer(nn.Module):
    """swinunet_icl.py:418-477."""

    def __init__(self, dim, input_resolution, depth, num_heads, window_size, drop_path, downsample, device=None):
        super().__init__()
        self.blocks = nn.ModuleList([
            SwinTransformerBlock(dim, input_resolution, num_heads, window_size, 0 if i % 2 == 0 else window_size // 2,
                                 drop_path[i], device) for i in range(depth)])
        self.downsample = PatchMerging(input_resolution, dim, device) if downsample else None

    def forward(self, x):
        for blk in self.blocks:
            x = blk(x)
        return self.downsample(x) if self.downsample is not None else x


class BasicLayer_up(nn.Module):  # noqa: N801
    """swinunet_icl.py:491-551: returns (x after the optional PatchExpand, block output before it)."""

    def __init__(self, dim, input_resolution, depth, num_heads, window_size, drop_path, upsample, device=None):
        super().__init__()
        self.blocks = nn.ModuleList([
            SwinTransformerBlock(dim, input_resolution, num_heads, window_size, 0 if i % 2 == 0 else window_size // 2,
                                 drop_path[i], device) for i in range(depth)])
        self.upsample = PatchExpand(input_resolution, dim, 2, device) if upsample else None

    def forward(self, x):
        for blk in self.blocks:
            x = blk(x)
        inter = x
        return (self.upsample(x) if self.upsample is not None else x), inter


class PatchEmbed(nn.Module):
    """swinunet_icl.py:554-594: Conv2d(k = s = patch) == one GEMM on the gathered patches, then LayerNorm."""

    def __init__(self, img_size=224, patch_size=4, in_chans=3, embed_dim=96, device=None):
        super().__init__()
        self.img_size, self.patch_size = (img_size, img_size), patch_size
        self.patches_resolution = [img_size // patch_size, img_size // patch_size]
        self.proj = Conv2d(in_chans, embed_dim, patch_size, device=device)     # parameter holder: weight [E, Cin, p, p], bias
        self.norm = LayerNorm(embed_dim, device)

    def forward(self, x):
        b, c, h, w = x.shape
        assert (h, w) == self.img_size, f"Input image size ({h}*{w}) doesn't match model ({self.img_size[0]}*{self.img_size[1]})."
        p = self.patch_size
        t = x.view(b, c, h // p, p, w // p, p).permute(0, 2, 4, 1, 3, 5).reshape(b, (h // p) * (w // p), c * p * p)
        return self.norm(ops.linear(t, self.proj.weight.flatten(1), self.proj.bias))


class SwinTransformerSys(nn.Module):
    def __init__(self, img_size=224, patch_size=4, in_chans=3, num_classes=1000, embed_dim=96, depths=(2, 2, 2, 2),
                 depths_decoder=(1, 2, 2, 2), num_heads=(3, 6, 12, 24), window_size=7, mlp_ratio=4.0, qkv_bias=True, qk_scale=None,
                 drop_rate=0.0, attn_drop_rate=0.0, drop_path_rate=0.1, ape=False, patch_norm=True, use_checkpoint=False,
                 final_upsample="expand_first", device=None, **kwargs):
        super().__init__()
        if ape or not patch_norm or drop_rate or attn_drop_rate or mlp_ratio != 4.0 or not qkv_bias or qk_scale:
            raise NotImplementedError("icl_amd SwinTransformerSys: only the configuration of swin_tiny_patch4_window7_224_lite.yaml")
        self.num_classes, self.num_layers, self.embed_dim = num_classes, len(depths), embed_dim
        self.patch_embed = PatchEmbed(img_size, patch_size, in_chans, embed_dim, device)
        pr = self.patch_embed.patches_resolution
        self.patches_resolution = pr
        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, sum(depths))]
        nl = self.num_layers
        self.layers = nn.ModuleList()
        for i in range(nl):
            self.layers.append(BasicLayer(embed_dim * 2 ** i, (pr[0] // 2 ** i, pr[1] // 2 ** i), depths[i], num_heads[i], window_size,
                                          dpr[sum(depths[:i]):sum(depths[:i + 1])], i < nl - 1, device))
        self.layers_up = nn.ModuleList()
        self.concat_back_dim = nn.ModuleList()
        for i in range(nl):
            j = nl - 1 - i
            dim, res = embed_dim * 2 ** j, (pr[0] // 2 ** j, pr[1] // 2 ** j)
            self.concat_back_dim.append(Linear(2 * dim, dim, device=device) if i > 0 else nn.Identity())
            if i == 0:
                self.layers_up.append(PatchExpand(res, dim, 2, device))
            else:
                self.layers_up.append(BasicLayer_up(dim, res, depths[j], num_heads[j], window_size,
                                                    dpr[sum(depths[:j]):sum(depths[:j + 1])], i < nl - 1, device))
        self.norm = LayerNorm(embed_dim * 2 ** (nl - 1), device)
        self.norm_up = LayerNorm(embed_dim, device)
        self.up = PatchExpand((img_size // patch_size, img_size // patch_size), embed_dim, 4, device)
        self.output = Conv2d(embed_dim, num_classes, 1, bias=False, device=device)     # 1x1: a GEMM on the token axis

    def forward_features(self, x):
        x = self.patch_embed(x)
        skips = []
        for layer in self.layers:
            skips.append(x)
            x = layer(x)
        return self.norm(x), skips

    def forward_up_features(self, x, skips):
        feats = []
        for inx, layer_up in enumerate(self.layers_up):
            if inx == 0:
                x = layer_up(x)
            else:
                x = self.concat_back_dim[inx](torch.cat([x, skips[3 - inx]], -1))
                x, feat = layer_up(x)
                feats.append(feat)
        return self.norm_up(x), feats

    def up_x4(self, x):
        h, w = self.patches_resolution
        b = x.shape[0]
        x = self.up(x)                                                              # [B, 16 H W, C]
        y = ops.linear(x, self.output.weight.flatten(1), None)                      # 1x1 conv on tokens
        return y.view(b, 4 * h, 4 * w, self.num_classes).permute(0, 3, 1, 2).contiguous()

    def run(self, x):
        xe, skips = self.forward_features(x)
        xl, feats = self.forward_up_features(xe, skips)
        return self.up_x4(xl), feats

    def forward(self, x_lab, x_unlab=None, inference=False):
        out_lab, feats_lab = self.run(x_lab)
        if inference:
            return out_lab
        out_unlab, feats_unlab = self.run(x_unlab)
        return out_lab, out_unlab, feats_lab, feats_unlab
