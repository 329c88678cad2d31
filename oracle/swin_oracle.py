"""CPU oracle for SwinUNETR-ICL — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Functional torch-CPU restatement over a flat ``{state_dict key: tensor}`` dict of
``networks/swinunetr_icl.py`` (citations below are lines of that file under /root/reference/code).
Same usage rule as oracle/icl_oracle.py: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import it.

Pinning
  * S1-S4 (window attention, shifted-window block, patch merging, stage wiring), the model wiring and the aligners (S6)
    are the reference's OWN vendored code; tests/golden/make_golden.py --only swin imports it and stores outputs in
    tests/golden/model_swinunetr_icl_64_nc2.npz (64^3 volumes, the vector tests/test_oracle_golden.py checks this file
    against) and model_swinunetr_icl_nc2.npz (96^3, the BASELINE shape, used by the GPU parity test).
  * S5 — the MONAI 1.0.1 blocks the reference imports (:22-23: MLPBlock, PatchEmbed, UnetrBasicBlock, UnetrUpBlock,
    UnetOutBlock) are a third-party dependency that is NOT in /root/reference and not installed here.  They are
    restated from MONAI 1.0.1's published definitions (monai/networks/blocks/{mlp,patchembedding,dynunet_block,
    unetr_block}.py); the golden above was made with the same restatement standing in for them, so for those five
    blocks the status is "PARITY UNPINNED" (no MONAI output was available to check against).
"""
from __future__ import annotations

import itertools
from typing import Dict, List, Sequence

import numpy as np
import torch
import torch.nn.functional as F

from . import icl_oracle as O

P = Dict[str, torch.Tensor]

WINDOW = (7, 7, 7)               # :83 ensure_tuple_rep(7, 3)
DEPTHS = (2, 2, 2, 2)            # :42
SWIN_HEADS = (3, 6, 12, 24)      # :43
FEATURE = 48                     # net_factory_3d.py:58
ICL_HEADS = (24, 12, 6)          # :236 num_heads[::-1][:3]
ICL_RES = (6, 12, 24)            # :235 img/16, img/8, img/4 for 96^3


# ---------------------------------------------------------------------------------------------- S1  window attention
def relative_position_index(ws: Sequence[int] = WINDOW) -> torch.Tensor:
    """WindowAttention.__init__ :684-699: index into the (2w-1)^3 bias table for every (query, key) pair of a window."""
    coords = torch.stack(torch.meshgrid(*[torch.arange(w) for w in ws], indexing="ij")).flatten(1)
    rel = (coords[:, :, None] - coords[:, None, :]).permute(1, 2, 0).contiguous()
    for a in range(3):
        rel[:, :, a] += ws[a] - 1
    rel[:, :, 0] *= (2 * ws[1] - 1) * (2 * ws[2] - 1)
    rel[:, :, 1] *= 2 * ws[2] - 1
    return rel.sum(-1)


def window_attention(p: P, pre: str, x: torch.Tensor, mask, heads: int) -> torch.Tensor:
    """WindowAttention.forward :727-750.  x [nW*B, n, C].  Quirk kept: for clipped windows (n < 343) the bias is the
    top-left [:n, :n] corner of the 7^3 index, not the index of the clipped window (:733-735)."""
    b, n, c = x.shape
    d = c // heads
    qkv = F.linear(x, p[f"{pre}.qkv.weight"], p[f"{pre}.qkv.bias"]).reshape(b, n, 3, heads, d).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0] * d ** -0.5, qkv[1], qkv[2]
    attn = q @ k.transpose(-2, -1)
    idx = p[f"{pre}.relative_position_index"][:n, :n].reshape(-1)
    bias = p[f"{pre}.relative_position_bias_table"][idx].reshape(n, n, -1).permute(2, 0, 1)
    attn = attn + bias.unsqueeze(0)
    if mask is not None:
        nw = mask.shape[0]
        attn = (attn.view(b // nw, nw, heads, n, n) + mask.unsqueeze(1).unsqueeze(0)).view(-1, heads, n, n)
    attn = attn.softmax(-1)
    x = (attn @ v).transpose(1, 2).reshape(b, n, c)
    return F.linear(x, p[f"{pre}.proj.weight"], p[f"{pre}.proj.bias"])


# ---------------------------------------------------------------------------------------------- S2  shifted-window block
def get_window_size(x_size, window_size, shift_size):
    """:617-641: clip the window to the volume, and do not shift an axis that fits in one window."""
    ws, ss = list(window_size), list(shift_size)
    for i in range(len(x_size)):
        if x_size[i] <= window_size[i]:
            ws[i] = x_size[i]
            ss[i] = 0
    return tuple(ws), tuple(ss)


def window_partition(x: torch.Tensor, ws) -> torch.Tensor:
    """:552-582: [b,d,h,w,c] -> [b*nW, prod(ws), c], windows ordered (b, zw, yw, xw), tokens (z, y, x) inside."""
    b, d, h, w, c = x.shape
    x = x.view(b, d // ws[0], ws[0], h // ws[1], ws[1], w // ws[2], ws[2], c)
    return x.permute(0, 1, 3, 5, 2, 4, 6, 7).contiguous().view(-1, ws[0] * ws[1] * ws[2], c)


def window_reverse(win: torch.Tensor, ws, dims) -> torch.Tensor:
    """:585-614."""
    b, d, h, w = dims
    x = win.view(b, d // ws[0], h // ws[1], w // ws[2], ws[0], ws[1], ws[2], -1)
    return x.permute(0, 1, 4, 2, 5, 3, 6, 7).contiguous().view(b, d, h, w, -1)


def compute_mask(dims, ws, ss) -> torch.Tensor:
    """:979-1016: region ids of the rolled volume -> additive mask (0 / -100) per window [nW, n, n].
    (``slice(-0, None)`` is the whole axis when an axis is not shifted — kept as written.)"""
    d, h, w = dims
    img = torch.zeros((1, d, h, w, 1))
    cnt = 0
    for sd, sh, sw in itertools.product(*[(slice(-ws[a]), slice(-ws[a], -ss[a]), slice(-ss[a], None)) for a in range(3)]):
        img[:, sd, sh, sw, :] = cnt
        cnt += 1
    mw = window_partition(img, ws).squeeze(-1)
    m = mw.unsqueeze(1) - mw.unsqueeze(2)
    return m.masked_fill(m != 0, -100.0).masked_fill(m == 0, 0.0)


def swin_block(p: P, pre: str, x: torch.Tensor, mask, heads: int, window, shift) -> torch.Tensor:
    """SwinTransformerBlock.forward :905-916 (drop_path == identity: dropout_path_rate 0.0, net_factory_3d.py:34).
    part1 :814-866: LN -> zero-pad to a multiple of the window (AFTER the norm) -> roll(-shift) -> windows -> attention
    -> reverse -> roll(+shift) -> crop;  part2 :868: MLPBlock(LN2(x))  [MONAI MLPBlock: linear2(GELU(linear1(x)))]."""
    b, d, h, w, c = x.shape
    ws, ss = get_window_size((d, h, w), window, shift)
    y = O._ln(p, f"{pre}.norm1", x)
    pd, ph, pw = [(ws[a] - s % ws[a]) % ws[a] for a, s in enumerate((d, h, w))]
    y = F.pad(y, (0, 0, 0, pw, 0, ph, 0, pd))
    dims = [b, d + pd, h + ph, w + pw]
    shifted = any(s > 0 for s in ss)
    if shifted:
        y = torch.roll(y, shifts=(-ss[0], -ss[1], -ss[2]), dims=(1, 2, 3))
    win = window_attention(p, f"{pre}.attn", window_partition(y, ws), mask if shifted else None, heads)
    y = window_reverse(win.view(-1, *ws, c), ws, dims)
    if shifted:
        y = torch.roll(y, shifts=ss, dims=(1, 2, 3))
    y = y[:, :d, :h, :w, :]
    x = x + y
    z = O._ln(p, f"{pre}.norm2", x)
    z = F.linear(F.gelu(F.linear(z, p[f"{pre}.mlp.linear1.weight"], p[f"{pre}.mlp.linear1.bias"])),
                 p[f"{pre}.mlp.linear2.weight"], p[f"{pre}.mlp.linear2.bias"])
    return x + z


# ---------------------------------------------------------------------------------------------- S3  patch merging
def patch_merging(p: P, pre: str, x: torch.Tensor) -> torch.Tensor:
    """PatchMerging.forward :946-966 (MONAI 1.0.1 variant): the eight slices are NOT the eight octants — x5 repeats x2
    and x6 repeats x3 (:958-959), kept as written; LN(8C) -> Linear(8C -> 2C, no bias).  Even sizes only here."""
    assert all(s % 2 == 0 for s in x.shape[1:4])
    sl = [(0, 0, 0), (1, 0, 0), (0, 1, 0), (0, 0, 1), (1, 0, 1), (0, 1, 0), (0, 0, 1), (1, 1, 1)]
    x = torch.cat([x[:, a::2, b::2, c::2, :] for a, b, c in sl], -1)
    x = O._ln(p, f"{pre}.norm", x)
    return F.linear(x, p[f"{pre}.reduction.weight"])


# ---------------------------------------------------------------------------------------------- S4  stages
def basic_layer(p: P, pre: str, x: torch.Tensor, heads: int) -> torch.Tensor:
    """BasicLayer.forward :1086-1101: channels-last, one mask per stage, block 0 unshifted / block 1 shifted by 3."""
    b, c, d, h, w = x.shape
    shift = tuple(i // 2 for i in WINDOW)
    ws, ss = get_window_size((d, h, w), WINDOW, shift)
    x = x.permute(0, 2, 3, 4, 1)
    dp, hp, wp = [int(np.ceil(s / ws[a])) * ws[a] for a, s in enumerate((d, h, w))]
    mask = compute_mask([dp, hp, wp], ws, ss)
    for i in range(2):
        x = swin_block(p, f"{pre}.blocks.{i}", x, mask, heads, WINDOW, (0, 0, 0) if i % 2 == 0 else shift)
    x = patch_merging(p, f"{pre}.downsample", x.reshape(b, d, h, w, -1))
    return x.permute(0, 4, 1, 2, 3)


def proj_out(x: torch.Tensor) -> torch.Tensor:
    """:1208-1221 with normalize=True: LayerNorm over channels, no affine."""
    return F.layer_norm(x.permute(0, 2, 3, 4, 1), [x.shape[1]]).permute(0, 4, 1, 2, 3)


def swin_vit(p: P, x: torch.Tensor, pre: str = "swinViT") -> List[torch.Tensor]:
    """SwinTransformer.forward :1223-1235.  PatchEmbed (MONAI, S5): Conv3d(k=2, s=2) with bias."""
    x0 = F.conv3d(x, p[f"{pre}.patch_embed.proj.weight"], p[f"{pre}.patch_embed.proj.bias"], stride=2)
    outs = [proj_out(x0)]
    cur = x0
    for i, heads in enumerate(SWIN_HEADS):
        cur = basic_layer(p, f"{pre}.layers{i + 1}.0", cur.contiguous(), heads)
        outs.append(proj_out(cur))
    return outs


# ---------------------------------------------------------------------------------------------- S5  MONAI CNN blocks
def _inorm(x):
    return F.instance_norm(x, eps=1e-5)      # ("instance", affine False) -> nn.InstanceNorm3d defaults


def unet_res_block(p: P, pre: str, x: torch.Tensor) -> torch.Tensor:
    """MONAI 1.0.1 UnetResBlock.forward (dynunet_block.py): conv3-IN-LeakyReLU(0.01)-conv3-IN, + (1x1 conv-IN of the input
    when channels change), LeakyReLU.  Convolutions have no bias.  PARITY UNPINNED (see header)."""
    out = F.leaky_relu(_inorm(F.conv3d(x, p[f"{pre}.conv1.conv.weight"], padding=1)), 0.01)
    out = _inorm(F.conv3d(out, p[f"{pre}.conv2.conv.weight"], padding=1))
    res = x
    if f"{pre}.conv3.conv.weight" in p:
        res = _inorm(F.conv3d(x, p[f"{pre}.conv3.conv.weight"]))
    return F.leaky_relu(out + res, 0.01)


def unetr_up_block(p: P, pre: str, x: torch.Tensor, skip: torch.Tensor) -> torch.Tensor:
    """MONAI 1.0.1 UnetrUpBlock.forward (unetr_block.py): ConvTranspose3d(k=2, s=2, no bias) -> cat((up, skip)) -> res block."""
    up = F.conv_transpose3d(x, p[f"{pre}.transp_conv.conv.weight"], stride=2)
    return unet_res_block(p, f"{pre}.conv_block", torch.cat((up, skip), 1))


def swin_backbone(p: P, x: torch.Tensor):
    """One stream of SwinUNETR_icl.forward :313-327.  Returns (logits, [dec3, dec2, dec1])."""
    hs = swin_vit(p, x)
    enc0 = unet_res_block(p, "encoder1.layer", x)
    enc1 = unet_res_block(p, "encoder2.layer", hs[0])
    enc2 = unet_res_block(p, "encoder3.layer", hs[1])
    enc3 = unet_res_block(p, "encoder4.layer", hs[2])
    dec4 = unet_res_block(p, "encoder10.layer", hs[4])
    dec3 = unetr_up_block(p, "decoder5", dec4, hs[3])
    dec2 = unetr_up_block(p, "decoder4", dec3, enc3)
    dec1 = unetr_up_block(p, "decoder3", dec2, enc2)
    dec0 = unetr_up_block(p, "decoder2", dec1, enc1)
    out = unetr_up_block(p, "decoder1", dec0, enc0)
    logits = F.conv3d(out, p["out.conv.conv.weight"], p["out.conv.conv.bias"])
    return logits, [dec3, dec2, dec1]


def swinunetr_icl_forward(p: P, x_lab, x_unlab=None, inference=False, training=True):
    """SwinUNETR_icl.forward :310-357.  The aligners get four feature maps but use the first three (depths[:3])."""
    final_lab, feats_lab = swin_backbone(p, x_lab)
    if inference:
        return final_lab
    final_unlab, feats_unlab = swin_backbone(p, x_unlab)
    kw = dict(training=training, q_name="guide_Q")
    maps_lab, qs_lab = O.inherent_consistent(p, "sspa", feats_lab, ICL_HEADS, None, "labeled", **kw)
    maps_con, _ = O.inherent_consistent(p, "sspa", feats_unlab, ICL_HEADS, None, "labeled", **kw)
    maps_unlab, _ = O.inherent_consistent(p, "uscl", feats_unlab, ICL_HEADS, qs_lab, "unlabeled", **kw)
    return final_lab, final_unlab, maps_lab, maps_unlab, maps_con


# ---------------------------------------------------------------------------------------------- parameter specs
def swin_vit_shapes(pre: str = "swinViT.", in_ch: int = 1, f: int = FEATURE):
    out = [(pre + "patch_embed.proj.weight", (f, in_ch, 2, 2, 2)), (pre + "patch_embed.proj.bias", (f,))]
    nt = (2 * WINDOW[0] - 1) * (2 * WINDOW[1] - 1) * (2 * WINDOW[2] - 1)
    for i, h in enumerate(SWIN_HEADS):
        c = f * 2 ** i
        for b in range(DEPTHS[i]):
            q = f"{pre}layers{i + 1}.0.blocks.{b}."
            out += [(q + "norm1.weight", (c,)), (q + "norm1.bias", (c,)),
                    (q + "attn.relative_position_bias_table", (nt, h)),
                    (q + "attn.qkv.weight", (3 * c, c)), (q + "attn.qkv.bias", (3 * c,)),
                    (q + "attn.proj.weight", (c, c)), (q + "attn.proj.bias", (c,)),
                    (q + "norm2.weight", (c,)), (q + "norm2.bias", (c,)),
                    (q + "mlp.linear1.weight", (4 * c, c)), (q + "mlp.linear1.bias", (4 * c,)),
                    (q + "mlp.linear2.weight", (c, 4 * c)), (q + "mlp.linear2.bias", (c,))]
        q = f"{pre}layers{i + 1}.0.downsample."
        out += [(q + "reduction.weight", (2 * c, 8 * c)), (q + "norm.weight", (8 * c,)), (q + "norm.bias", (8 * c,))]
    return out


def res_block_shapes(pre: str, cin: int, cout: int):
    out = [(pre + "conv1.conv.weight", (cout, cin, 3, 3, 3)), (pre + "conv2.conv.weight", (cout, cout, 3, 3, 3))]
    if cin != cout:
        out.append((pre + "conv3.conv.weight", (cout, cin, 1, 1, 1)))
    return out


def swinunetr_shapes(nc: int, in_ch: int = 1, f: int = FEATURE):
    """Parameters of the plain SwinUNETR backbone in the reference's registration order (:123-232)."""
    out = swin_vit_shapes("swinViT.", in_ch, f)
    out += res_block_shapes("encoder1.layer.", in_ch, f)
    out += res_block_shapes("encoder2.layer.", f, f)
    out += res_block_shapes("encoder3.layer.", 2 * f, 2 * f)
    out += res_block_shapes("encoder4.layer.", 4 * f, 4 * f)
    out += res_block_shapes("encoder10.layer.", 16 * f, 16 * f)
    for name, cin, cout in (("decoder5", 16 * f, 8 * f), ("decoder4", 8 * f, 4 * f), ("decoder3", 4 * f, 2 * f),
                            ("decoder2", 2 * f, f), ("decoder1", f, f)):
        out.append((f"{name}.transp_conv.conv.weight", (cin, cout, 2, 2, 2)))
        out += res_block_shapes(f"{name}.conv_block.", 2 * cout, cout)
    out += [("out.conv.conv.weight", (nc, f, 1, 1, 1)), ("out.conv.conv.bias", (nc,))]
    return out


def swinunetr_icl_shapes(nc: int, in_ch: int = 1, f: int = FEATURE, res=ICL_RES):
    """``res``: aligner token grids = (img/16, img/8, img/4) (swinunetr_icl.py:235): (6, 12, 24) for 96^3 volumes."""
    ch = (8 * f, 4 * f, 2 * f)
    return (swinunetr_shapes(nc, in_ch, f)
            + O.aligner_shapes("sspa.", ch, res, nc, ICL_HEADS, q_name="guide_Q")
            + O.aligner_shapes("uscl.", ch, res, nc, ICL_HEADS, q_name="guide_Q"))


def swin_buffers(pre: str = "swinViT."):
    idx = relative_position_index()
    return {f"{pre}layers{i + 1}.0.blocks.{b}.attn.relative_position_index": idx
            for i in range(4) for b in range(DEPTHS[i])}


def make_swin_params(nc: int, requires_grad: bool = False, icl: bool = True, res=ICL_RES) -> P:
    p = O.make_params(swinunetr_icl_shapes(nc, res=res) if icl else swinunetr_shapes(nc), requires_grad=requires_grad)
    p.update(swin_buffers())
    if icl:
        p.update(O.aligner_buffers("sspa.", ICL_HEADS))
        p.update(O.aligner_buffers("uscl.", ICL_HEADS))
    return p
