"""CPU oracle for the 2-D Swin-UNet ICL model — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Functional torch-CPU restatement over a flat ``{state_dict key: tensor}`` dict of ``networks/swinunet_icl.py``
(SwinTransformerSys and its blocks) and ``networks/vision_transformer.py`` (the SwinUnet wrapper with the aligners);
citations are lines of those files under /root/reference/code.  Same usage rule as oracle/icl_oracle.py.

Pinned: tests/golden/make_golden.py --only swinunet2d imports the real reference (stubbing timm's DropPath / to_2tuple /
trunc_normal_, the accidental ``from turtle import back`` and the yacs config object — none of which carries arithmetic of
the path) and stores outputs in tests/golden/model_swinunet2d_icl_nc4.npz; tests/test_oracle_golden.py checks this file
against them.  Configuration: configs/swin_tiny_patch4_window7_224_lite.yaml (embed 96, depths 2-2-2-2, heads 3-6-12-24,
window 7, 224^2 input, patch 4).
"""
from __future__ import annotations

from typing import Dict, List

import torch
import torch.nn.functional as F

from . import icl_oracle as O

P = Dict[str, torch.Tensor]

EMBED, DEPTHS, HEADS, WINDOW, IMG, PATCH = 96, (2, 2, 2, 2), (3, 6, 12, 24), 7, 224, 4
ICL_CH, ICL_RES, ICL_HEADS = (384, 192, 96), (14, 28, 56), (24, 12, 6)     # vision_transformer.py:55-58


def relative_position_index(ws: int = WINDOW) -> torch.Tensor:
    """WindowAttention.__init__, swinunet_icl.py:95-108."""
    coords = torch.stack(torch.meshgrid(torch.arange(ws), torch.arange(ws), indexing="ij")).flatten(1)
    rel = (coords[:, :, None] - coords[:, None, :]).permute(1, 2, 0).contiguous()
    rel[:, :, 0] += ws - 1
    rel[:, :, 1] += ws - 1
    rel[:, :, 0] *= 2 * ws - 1
    return rel.sum(-1)


def window_partition(x, ws):
    """swinunet_icl.py:33-47: [B,H,W,C] -> [B*nW, ws, ws, C]."""
    b, h, w, c = x.shape
    return x.view(b, h // ws, ws, w // ws, ws, c).permute(0, 1, 3, 2, 4, 5).contiguous().view(-1, ws, ws, c)


def window_reverse(win, ws, h, w):
    """swinunet_icl.py:50-65."""
    b = int(win.shape[0] / (h * w / ws / ws))
    return win.view(b, h // ws, w // ws, ws, ws, -1).permute(0, 1, 3, 2, 4, 5).contiguous().view(b, h, w, -1)


def attn_mask(res: int, ws: int, shift: int) -> torch.Tensor:
    """SwinTransformerBlock.__init__, swinunet_icl.py:222-245: 0 / -100 mask per window of the rolled image."""
    img = torch.zeros((1, res, res, 1))
    cnt = 0
    for hs in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
        for wsl in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
            img[:, hs, wsl, :] = cnt
            cnt += 1
    mw = window_partition(img, ws).view(-1, ws * ws)
    m = mw.unsqueeze(1) - mw.unsqueeze(2)
    return m.masked_fill(m != 0, -100.0).masked_fill(m == 0, 0.0)


def window_attention(p: P, pre: str, x, mask, heads: int):
    """WindowAttention.forward, swinunet_icl.py:120-155."""
    b_, n, c = x.shape
    d = c // heads
    qkv = F.linear(x, p[f"{pre}.qkv.weight"], p[f"{pre}.qkv.bias"]).reshape(b_, n, 3, heads, d).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0] * d ** -0.5, qkv[1], qkv[2]
    attn = q @ k.transpose(-2, -1)
    bias = p[f"{pre}.relative_position_bias_table"][p[f"{pre}.relative_position_index"].view(-1)].view(n, n, -1).permute(2, 0, 1)
    attn = attn + bias.unsqueeze(0)
    if mask is not None:
        nw = mask.shape[0]
        attn = (attn.view(b_ // nw, nw, heads, n, n) + mask.unsqueeze(1).unsqueeze(0)).view(-1, heads, n, n)
    attn = attn.softmax(-1)
    x = (attn @ v).transpose(1, 2).reshape(b_, n, c)
    return F.linear(x, p[f"{pre}.proj.weight"], p[f"{pre}.proj.bias"])


def swin_block(p: P, pre: str, x, res: int, heads: int, shift: int):
    """SwinTransformerBlock.forward, swinunet_icl.py:249-293 (drop_path identity in parity mode).  A resolution <= the window
    makes the block unshifted with window = resolution (:203-206)."""
    ws = WINDOW
    if res <= ws:
        shift, ws = 0, res
    b, l, c = x.shape
    y = O._ln(p, f"{pre}.norm1", x).view(b, res, res, c)
    if shift > 0:
        y = torch.roll(y, shifts=(-shift, -shift), dims=(1, 2))
    win = window_partition(y, ws).view(-1, ws * ws, c)
    win = window_attention(p, f"{pre}.attn", win, p.get(f"{pre}.attn_mask") if shift > 0 else None, heads)
    y = window_reverse(win.view(-1, ws, ws, c), ws, res, res)
    if shift > 0:
        y = torch.roll(y, shifts=(shift, shift), dims=(1, 2))
    x = x + y.view(b, l, c)
    z = O._ln(p, f"{pre}.norm2", x)
    z = F.linear(F.gelu(F.linear(z, p[f"{pre}.mlp.fc1.weight"], p[f"{pre}.mlp.fc1.bias"])), p[f"{pre}.mlp.fc2.weight"], p[f"{pre}.mlp.fc2.bias"])
    return x + z


def patch_merging(p: P, pre: str, x, res: int):
    """PatchMerging.forward, swinunet_icl.py:330-351: the four 2x2 phases (0,0),(1,0),(0,1),(1,1) -> LN(4C) -> Linear(4C->2C)."""
    b, l, c = x.shape
    x = x.view(b, res, res, c)
    x = torch.cat([x[:, 0::2, 0::2], x[:, 1::2, 0::2], x[:, 0::2, 1::2], x[:, 1::2, 1::2]], -1).view(b, -1, 4 * c)
    return F.linear(O._ln(p, f"{pre}.norm", x), p[f"{pre}.reduction.weight"])


def patch_expand(p: P, pre: str, x, res: int, scale: int = 2):
    """PatchExpand / FinalPatchExpand_X4, swinunet_icl.py:372-387, 400-414: Linear (no bias) -> 'b h w (p1 p2 c) -> b (h p1) (w p2) c'
    -> LN."""
    x = F.linear(x, p[f"{pre}.expand.weight"])
    b, l, c = x.shape
    co = c // (scale * scale)
    x = x.view(b, res, res, scale, scale, co).permute(0, 1, 3, 2, 4, 5).reshape(b, res * scale * res * scale, co)
    return O._ln(p, f"{pre}.norm", x)


def forward_features(p: P, x, pre: str = "swin_unet"):
    """SwinTransformerSys.forward_features, swinunet_icl.py:752-766."""
    x = F.conv2d(x, p[f"{pre}.patch_embed.proj.weight"], p[f"{pre}.patch_embed.proj.bias"], stride=PATCH).flatten(2).transpose(1, 2)
    x = O._ln(p, f"{pre}.patch_embed.norm", x)
    skips = []
    res = IMG // PATCH
    for i in range(4):
        skips.append(x)
        for b in range(DEPTHS[i]):
            x = swin_block(p, f"{pre}.layers.{i}.blocks.{b}", x, res, HEADS[i], 0 if b % 2 == 0 else WINDOW // 2)
        if i < 3:
            x = patch_merging(p, f"{pre}.layers.{i}.downsample", x, res)
            res //= 2
    return O._ln(p, f"{pre}.norm", x), skips


def forward_up_features(p: P, x, skips, pre: str = "swin_unet"):
    """:768-781: PatchExpand, then three BasicLayer_up stages on cat([x, skip]) -> Linear; the aligner features are the
    block outputs BEFORE the stage's own PatchExpand (inter_feat, :546-551)."""
    feats = []
    res = IMG // PATCH // 8
    x = patch_expand(p, f"{pre}.layers_up.0", x, res)
    res *= 2
    for inx in (1, 2, 3):
        x = torch.cat([x, skips[3 - inx]], -1)
        x = F.linear(x, p[f"{pre}.concat_back_dim.{inx}.weight"], p[f"{pre}.concat_back_dim.{inx}.bias"])
        stage = 3 - inx
        for b in range(DEPTHS[stage]):
            x = swin_block(p, f"{pre}.layers_up.{inx}.blocks.{b}", x, res, HEADS[stage], 0 if b % 2 == 0 else WINDOW // 2)
        feats.append(x)
        if inx < 3:
            x = patch_expand(p, f"{pre}.layers_up.{inx}.upsample", x, res)
            res *= 2
    return O._ln(p, f"{pre}.norm_up", x), feats


def up_x4(p: P, x, pre: str = "swin_unet"):
    """:783-794."""
    res = IMG // PATCH
    x = patch_expand(p, f"{pre}.up", x, res, 4)
    x = x.view(x.shape[0], 4 * res, 4 * res, -1).permute(0, 3, 1, 2)
    return F.conv2d(x, p[f"{pre}.output.weight"])


def backbone(p: P, x):
    xe, skips = forward_features(p, x)
    xl, feats = forward_up_features(p, xe, skips)
    return up_x4(p, xl), feats


def swinunet_icl_forward(p: P, x_lab, x_unlab=None, inference=False, training=True):
    """SwinUnet.forward, vision_transformer.py:88-108 (grey images are repeated to three channels)."""
    if x_lab.shape[1] == 1:
        x_lab = x_lab.repeat(1, 3, 1, 1)
    out_lab, feats_lab = backbone(p, x_lab)
    if inference:
        return out_lab
    if x_unlab.shape[1] == 1:
        x_unlab = x_unlab.repeat(1, 3, 1, 1)
    out_unlab, feats_unlab = backbone(p, x_unlab)
    kw = dict(training=training, token_dims=2)
    maps_lab, qs = O.inherent_consistent(p, "sspa", feats_lab, ICL_HEADS, None, "labeled", **kw)
    maps_con, _ = O.inherent_consistent(p, "sspa", feats_unlab, ICL_HEADS, None, "labeled", **kw)
    maps_unlab, _ = O.inherent_consistent(p, "uscl", feats_unlab, ICL_HEADS, qs, "unlabeled", **kw)
    return out_lab, out_unlab, maps_lab, maps_unlab, maps_con


def icl_losses(outputs, labels, n_classes: int):
    """train_inherent_consistent_swinunet_2D.py:148-155: ce + dice(softmax) + aux + pse + 50 * consistency, maps resized to 224^2."""
    return O.icl_losses_2d(outputs, labels, n_classes, size=(IMG, IMG), w_con=50.0)


# ---------------------------------------------------------------------------------------------- parameter specs
def _block_shapes(q: str, c: int, h: int):
    return [(q + "norm1.weight", (c,)), (q + "norm1.bias", (c,)),
            (q + "attn.relative_position_bias_table", ((2 * WINDOW - 1) ** 2, h)),
            (q + "attn.qkv.weight", (3 * c, c)), (q + "attn.qkv.bias", (3 * c,)),
            (q + "attn.proj.weight", (c, c)), (q + "attn.proj.bias", (c,)),
            (q + "norm2.weight", (c,)), (q + "norm2.bias", (c,)),
            (q + "mlp.fc1.weight", (4 * c, c)), (q + "mlp.fc1.bias", (4 * c,)),
            (q + "mlp.fc2.weight", (c, 4 * c)), (q + "mlp.fc2.bias", (c,))]


def swin_unet_shapes(nc: int, pre: str = "swin_unet."):
    e = EMBED
    out = [(pre + "patch_embed.proj.weight", (e, 3, PATCH, PATCH)), (pre + "patch_embed.proj.bias", (e,)),
           (pre + "patch_embed.norm.weight", (e,)), (pre + "patch_embed.norm.bias", (e,))]
    for i in range(4):
        c = e * 2 ** i
        for b in range(DEPTHS[i]):
            out += _block_shapes(f"{pre}layers.{i}.blocks.{b}.", c, HEADS[i])
        if i < 3:
            q = f"{pre}layers.{i}.downsample."
            out += [(q + "reduction.weight", (2 * c, 4 * c)), (q + "norm.weight", (4 * c,)), (q + "norm.bias", (4 * c,))]
    out += [(pre + "layers_up.0.expand.weight", (16 * e, 8 * e)), (pre + "layers_up.0.norm.weight", (4 * e,)),
            (pre + "layers_up.0.norm.bias", (4 * e,))]
    for inx in (1, 2, 3):
        stage = 3 - inx
        c = e * 2 ** stage
        for b in range(DEPTHS[stage]):
            out += _block_shapes(f"{pre}layers_up.{inx}.blocks.{b}.", c, HEADS[stage])
        if inx < 3:
            q = f"{pre}layers_up.{inx}.upsample."
            out += [(q + "expand.weight", (2 * c, c)), (q + "norm.weight", (c // 2,)), (q + "norm.bias", (c // 2,))]
    for inx in (1, 2, 3):
        c = e * 2 ** (3 - inx)
        out += [(f"{pre}concat_back_dim.{inx}.weight", (c, 2 * c)), (f"{pre}concat_back_dim.{inx}.bias", (c,))]
    out += [(pre + "norm.weight", (8 * e,)), (pre + "norm.bias", (8 * e,)), (pre + "norm_up.weight", (e,)), (pre + "norm_up.bias", (e,)),
            (pre + "up.expand.weight", (16 * e, e)), (pre + "up.norm.weight", (e,)), (pre + "up.norm.bias", (e,)),
            (pre + "output.weight", (nc, e, 1, 1))]
    return out


def swinunet_icl_shapes(nc: int):
    return (swin_unet_shapes(nc) + O.aligner_shapes_nd("sspa.", ICL_CH, ICL_RES, nc, ICL_HEADS, 2)
            + O.aligner_shapes_nd("uscl.", ICL_CH, ICL_RES, nc, ICL_HEADS, 2))


def swin_buffers(pre: str = "swin_unet."):
    out = {}
    idx = relative_position_index()
    stages = [(f"{pre}layers.{i}", IMG // PATCH // 2 ** i) for i in range(4)] + \
             [(f"{pre}layers_up.{inx}", IMG // PATCH // 2 ** (3 - inx)) for inx in (1, 2, 3)]
    for q, res in stages:
        for b in range(2):
            out[f"{q}.blocks.{b}.attn.relative_position_index"] = idx
            if b == 1 and res > WINDOW:
                out[f"{q}.blocks.{b}.attn_mask"] = attn_mask(res, WINDOW, WINDOW // 2)
    return out


def make_params(nc: int, requires_grad: bool = False) -> P:
    p = O.make_params(swinunet_icl_shapes(nc), requires_grad=requires_grad)
    p.update(swin_buffers())
    p.update(O.aligner_buffers("sspa.", ICL_HEADS))
    p.update(O.aligner_buffers("uscl.", ICL_HEADS))
    return p
