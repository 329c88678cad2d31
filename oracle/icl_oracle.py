"""CPU oracle for the ICL hot path — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A functional restatement (plain torch-CPU fp32 ops over a flat ``{name: tensor}``
parameter dict, no nn.Module, no MONAI) of the reference algorithm for
SURVEY.md §8(a) rows A1–A7, B1–B6, L1–L5, L7 and T1.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module; the product path (``icl_amd``) never does and fails loudly when its
HIP library is missing.

Pinning: ``tests/golden/make_golden.py`` imports the real reference from
``/root/reference/code`` (with the five trivial MONAI symbols of SURVEY.md §8c
stubbed) in the build container, runs it on hash-filled weights/inputs and
stores inputs' seeds + outputs under ``tests/golden/*.npz``;
``tests/test_oracle_golden.py`` checks this oracle against those vectors.
The reference has no tests or golden vectors of its own.

Parameter names are the reference's ``state_dict`` keys, so the same dict drives
the reference, this oracle and the HIP-backed modules.

All file:line citations are relative to ``/root/reference/code``.
"""
from __future__ import annotations

import math
from typing import Dict, List, Sequence, Tuple

import torch
import torch.nn.functional as F

P = Dict[str, torch.Tensor]

# ----------------------------------------------------------------------------
# Backbone (A1-A7)
# ----------------------------------------------------------------------------


def unet_conv3(p: P, pre: str, x: torch.Tensor) -> torch.Tensor:
    """UnetConv3, networks/utils.py:99-123: 2 x [Conv3d 3^3 pad 1 -> InstanceNorm3d -> ReLU].

    InstanceNorm3d defaults: eps 1e-5, affine=False, no running stats (utils.py:105,108).
    """
    for blk in ("conv1", "conv2"):
        x = F.conv3d(x, p[f"{pre}.{blk}.0.weight"], p[f"{pre}.{blk}.0.bias"], stride=1, padding=1)
        x = F.instance_norm(x, eps=1e-5)
        x = F.relu(x)
    return x


def unet_up3_ct(p: P, pre: str, skip: torch.Tensor, deep: torch.Tensor) -> torch.Tensor:
    """UnetUp3_CT.forward, networks/utils.py:271-276: trilinear x2 (align_corners=False),
    zero-offset pad (no-op for matching sizes), cat([skip, up], 1), UnetConv3."""
    up = F.interpolate(deep, scale_factor=(2, 2, 2), mode="trilinear", align_corners=False)
    assert up.shape[2:] == skip.shape[2:]
    return unet_conv3(p, f"{pre}.conv", torch.cat([skip, up], 1))


def backbone(p: P, x: torch.Tensor, drop_p: float = 0.0, training: bool = False):
    """One stream of unet_3D_icl.forward, networks/unet_3D_icl.py:100-117
    (== unet_3D.forward, networks/unet_3D.py:72-94).  Returns (logits, [center, up4, up3])."""
    c1 = unet_conv3(p, "conv1", x)
    c2 = unet_conv3(p, "conv2", F.max_pool3d(c1, 2))
    c3 = unet_conv3(p, "conv3", F.max_pool3d(c2, 2))
    c4 = unet_conv3(p, "conv4", F.max_pool3d(c3, 2))
    center = unet_conv3(p, "center", F.max_pool3d(c4, 2))
    center = F.dropout(center, drop_p, training)  # dropout1, :110
    up4 = unet_up3_ct(p, "up_concat4", c4, center)
    up3 = unet_up3_ct(p, "up_concat3", c3, up4)
    up2 = unet_up3_ct(p, "up_concat2", c2, up3)
    up1 = unet_up3_ct(p, "up_concat1", c1, up2)
    up1 = F.dropout(up1, drop_p, training)  # dropout2, :116
    logits = F.conv3d(up1, p["final.weight"], p["final.bias"])  # :117
    return logits, [center, up4, up3]


# ----------------------------------------------------------------------------
# Aligners (B1-B6)
# ----------------------------------------------------------------------------


def _ln(p: P, pre: str, x: torch.Tensor) -> torch.Tensor:
    w = p[f"{pre}.weight"]
    return F.layer_norm(x, (w.shape[0],), w, p[f"{pre}.bias"], 1e-5)


def _lin(p: P, pre: str, x: torch.Tensor) -> torch.Tensor:
    return F.linear(x, p[f"{pre}.weight"], p[f"{pre}.bias"])


def _mlp(p: P, pre: str, x: torch.Tensor) -> torch.Tensor:
    """MLP.forward, networks/unet_3D_icl.py:309-315: fc2(GELU_erf(fc1(x))), dropout 0."""
    return _lin(p, f"{pre}.fc2", F.gelu(_lin(p, f"{pre}.fc1", x)))


def query_attention(p: P, pre: str, q: torch.Tensor, x: torch.Tensor, heads: int):
    """Query_Attention.forward, networks/unet_3D_icl.py:283-297.

    Quirks kept on purpose: q and the attention output use a
    plain ``reshape`` for the head split (no transpose); the returned map is the scaled
    PRE-softmax logits permuted to [B, nc, h, N]."""
    B, N, C = x.shape
    nc = q.shape[1]
    d = C // heads
    qh = _lin(p, f"{pre}.fc_q", q).reshape(B, heads, nc, d)
    kv = _lin(p, f"{pre}.fc_kv", x).reshape(B, N, 2, heads, d).permute(2, 0, 3, 1, 4)
    k, v = kv[0], kv[1]
    logits = (qh @ k.transpose(-2, -1)) * (d ** -0.5)
    out = (logits.softmax(dim=-1) @ v).reshape(B, nc, C)
    out = _lin(p, f"{pre}.proj", out)
    return out, logits.permute(0, 2, 1, 3)


def class_decoder(p: P, pre: str, query: torch.Tensor, feat: torch.Tensor, heads: int):
    """Class_Decoder.forward, networks/unet_3D_icl.py:260-268 with DropPath == identity
    (parity mode: drop prob forced to 0): `q + dp(q)` doubles."""
    query, attn = query_attention(p, f"{pre}.attn", _ln(p, f"{pre}.norm1_query", query),
                                  _ln(p, f"{pre}.norm1", feat), heads)
    query = query + query
    query = query + _mlp(p, f"{pre}.mlp", _ln(p, f"{pre}.norm2", query))
    attn = attn + attn
    attn = attn + _mlp(p, f"{pre}.mlp2", _ln(p, f"{pre}.norm3", attn))
    return query, attn


def _bn_train(p: P, pre: str, x: torch.Tensor, training: bool) -> torch.Tensor:
    if training:  # batch statistics, biased variance; running stats updated in place when the buffers are present
        return F.batch_norm(x, p.get(f"{pre}.running_mean"), p.get(f"{pre}.running_var"), p[f"{pre}.weight"],
                            p[f"{pre}.bias"], True, 0.1, 1e-5)
    return F.batch_norm(x, p[f"{pre}.running_mean"], p[f"{pre}.running_var"],
                        p[f"{pre}.weight"], p[f"{pre}.bias"], False, 0.1, 1e-5)


def _convnd(x, w, b=None, padding=0, groups=1):
    return (F.conv3d if w.dim() == 5 else F.conv2d)(x, w, b, padding=padding, groups=groups)


def separable_conv3d(p: P, pre: str, x: torch.Tensor, training: bool) -> torch.Tensor:
    """SeparableConv3d(relu_first=False), networks/unet_3D_icl.py:317-345 (2-D twin: unet_icl.py:98-126): depthwise 3^d
    (groups=C, no bias) -> BN -> ReLU -> pointwise 1^d (no bias) -> BN -> ReLU."""
    c = x.shape[1]
    x = _convnd(x, p[f"{pre}.block.depthwise.weight"], None, padding=1, groups=c)
    x = F.relu(_bn_train(p, f"{pre}.block.bn_depth", x, training))
    x = _convnd(x, p[f"{pre}.block.pointwise.weight"], None)
    x = F.relu(_bn_train(p, f"{pre}.block.bn_point", x, training))
    return x


def inherent_consistent(p: P, pre: str, feats: Sequence[torch.Tensor], heads: Sequence[int],
                        guided_q=None, modal: str = "labeled", training: bool = True, q_name: str = "guided_Q",
                        token_dims: int = 0):
    """InherentConsistent.forward, networks/unet_3D_icl.py:202-242 (swinunetr_icl.py:406-446 is the same code with the
    learnable query named ``guide_Q``: pass ``q_name``).

    labeled: queries start from the learnable ``guided_Q`` and are handed down through
    ``query_convs`` (:208-221); unlabeled: each scale starts from ``guided_q[i]`` (:229).
    ``token_dims`` = 2: the 2-D Swin-UNet variant (networks/vision_transformer.py:247-248) feeds decoder TOKENS [B, N, C]
    straight into the class decoders — proj_layers / norm_layers exist as parameters but are never called.
    Returns (feat_maps[3], updated_Qs[3])."""
    bs = feats[0].shape[0]
    maps, upd = [], []
    nxt = p[f"{pre}.{q_name}"].expand(bs, -1, -1) if modal == "labeled" else None
    for i, f in enumerate(feats):
        if token_dims:
            tok = f
        else:
            tok = _convnd(f, p[f"{pre}.proj_layers.{i}.weight"], p[f"{pre}.proj_layers.{i}.bias"])
            tok = _ln(p, f"{pre}.norm_layers.{i}", tok.flatten(2).transpose(1, 2))
        q_in = nxt if modal == "labeled" else guided_q[i].expand(bs, -1, -1)
        q_out, attn = class_decoder(p, f"{pre}.class_decoders.{i}", q_in, tok, heads[i])
        b, nc, h, n = attn.shape
        dims = token_dims or f.dim() - 2   # 3: int(np.cbrt(N)) (unet_3D_icl.py:215); 2: int(np.sqrt(N)) (unet_icl.py:311)
        r = int(round(n ** (1.0 / dims)))
        sp = (r,) * dims
        a = attn.contiguous().view(b * nc, h, *sp)
        a = separable_conv3d(p, f"{pre}.attn_convs0.{i}", a, training)
        a = _convnd(a, p[f"{pre}.attn_convs1.{i}.weight"], p[f"{pre}.attn_convs1.{i}.bias"])
        maps.append(a.squeeze(1).reshape(b, nc, *sp))
        nq = F.conv1d(q_out.permute(0, 2, 1), p[f"{pre}.query_convs.{i}.weight"],
                      p[f"{pre}.query_convs.{i}.bias"])
        nxt = nq.permute(0, 2, 1)
        upd.append(q_out.mean(dim=0, keepdim=True))
    return maps, upd


UNET3D_HEADS = (16, 8, 4)  # networks/unet_3D_icl.py:86


def unet_3d_icl_forward(p: P, x_lab, x_unlab=None, inference=False, training=True,
                        heads: Sequence[int] = UNET3D_HEADS):
    """unet_3D_icl.forward, networks/unet_3D_icl.py:99-148 (dropout/drop-path prob 0)."""
    final_lab, feats_lab = backbone(p, x_lab)
    if inference:
        return final_lab
    final_unlab, feats_unlab = backbone(p, x_unlab)
    maps_lab, qs_lab = inherent_consistent(p, "sspa", feats_lab, heads, None, "labeled", training)
    maps_con, _ = inherent_consistent(p, "sspa", feats_unlab, heads, None, "labeled", training)
    maps_unlab, _ = inherent_consistent(p, "uscl", feats_unlab, heads, qs_lab, "unlabeled", training)
    return final_lab, final_unlab, maps_lab, maps_unlab, maps_con


# ----------------------------------------------------------------------------
# Losses (L1-L5, L7)
# ----------------------------------------------------------------------------


def dice_loss(probs: torch.Tensor, target: torch.Tensor, n_classes: int, softmax: bool = False):
    """DiceLoss.forward, utils/losses.py:218-231: one-hot by equality, per class
    1-(2*sum(p*t)+1e-5)/(sum(p^2)+sum(t^2)+1e-5) over batch+space, mean over classes.
    ``target`` is [B,1,...]."""
    if softmax:
        probs = torch.softmax(probs, dim=1)
    loss = 0.0
    for i in range(n_classes):
        t = (target[:, 0] == i).float()
        s = probs[:, i]
        inter = torch.sum(s * t)
        loss = loss + (1 - (2 * inter + 1e-5) / (torch.sum(s * s) + torch.sum(t * t) + 1e-5))
    return loss / n_classes


def aux_loss_3d(maps: Sequence[torch.Tensor], labels: torch.Tensor, n_classes: int, size=(96, 96, 96)):
    """AuxLoss3D.forward, utils/losses.py:261-271 (size hard-coded to 96^3 there)."""
    ce, dc = 0.0, 0.0
    for m in maps:
        r = F.interpolate(m.float(), size=list(size), mode="trilinear")
        ce = ce + F.cross_entropy(r, labels.long())
        dc = dc + dice_loss(r, labels.unsqueeze(1), n_classes, softmax=True)
    return ce / len(maps) + dc / len(maps)


def softmax_dice_loss(a: torch.Tensor, b: torch.Tensor):
    """softmax_dice_loss + dice_loss1, utils/losses.py:42-59,22-30 (plain sums in the denominator)."""
    sa, sb = F.softmax(a, dim=1), F.softmax(b, dim=1)
    n = a.shape[1]
    d = 0.0
    for i in range(n):
        inter = torch.sum(sa[:, i] * sb[:, i])
        d = d + (1 - (2 * inter + 1e-5) / (torch.sum(sa[:, i]) + torch.sum(sb[:, i]) + 1e-5))
    return d / n


def pseudo_soft_loss_3d(maps: Sequence[torch.Tensor], predicts: torch.Tensor, size=(96, 96, 96)):
    """PseudoSoftLoss3D.forward, utils/losses.py:290-299: target detached."""
    tgt = predicts.detach()
    d = 0.0
    for m in maps:
        d = d + softmax_dice_loss(F.interpolate(m.float(), size=list(size), mode="trilinear"), tgt)
    return d / len(maps)


def softmax_mse_loss(inputs: Sequence[torch.Tensor], targets: Sequence[torch.Tensor]):
    """softmax_mse_loss, utils/losses.py:68-90: per scale mean((softmax(a)-softmax(detach b))^2), averaged."""
    loss = 0.0
    for a, b in zip(inputs, targets):
        loss = loss + torch.mean((F.softmax(a, dim=1) - F.softmax(b.detach(), dim=1)) ** 2)
    return loss / len(inputs)


def icl_losses(outputs, labels_lab: torch.Tensor, n_classes: int, size=(96, 96, 96),
               w_pse: float = 1.0, w_con: float = 10.0):
    """The 5-term objective of train_inherent_consistent_unet_3D_BraTS.py:105-112
    (AMOS: w_pse=0.1, ...AMOS22.py:230).  labels_lab is [Bl, D, H, W] int64."""
    final_lab, final_unlab, maps_lab, maps_unlab, maps_con = outputs
    soft = torch.softmax(final_lab, dim=1)
    l_ce = F.cross_entropy(final_lab, labels_lab)
    l_dice = dice_loss(soft, labels_lab.unsqueeze(1), n_classes)
    l_aux = aux_loss_3d(maps_lab, labels_lab, n_classes, size)
    l_pse = pseudo_soft_loss_3d(maps_unlab, final_unlab, size)
    l_con = softmax_mse_loss(maps_unlab, maps_con)
    total = l_dice + l_ce + l_aux + w_pse * l_pse + w_con * l_con
    return total, dict(dice=l_dice, ce=l_ce, aux=l_aux, pse=l_pse, con=l_con)


# ----------------------------------------------------------------------------
# Trainer step (T1)
# ----------------------------------------------------------------------------


def sgd_step(params: P, grads: Dict[str, torch.Tensor], bufs: Dict[str, torch.Tensor],
             lr: float, momentum: float = 0.9, wd: float = 1e-4) -> None:
    """torch.optim.SGD semantics used at train_..._BraTS.py:85-86,115: params whose grad is
    None are skipped entirely; first step initialises the momentum buffer with d_p."""
    with torch.no_grad():
        for k, g in grads.items():
            if g is None:
                continue
            d = g + wd * params[k]
            if k in bufs:
                bufs[k].mul_(momentum).add_(d)
            else:
                bufs[k] = d.clone()
            params[k].add_(bufs[k], alpha=-lr)


def poly_lr(base_lr: float, iter_num: int, max_iterations: int) -> float:
    """train_..._BraTS.py:117: lr computed from the PRE-increment iter_num, used from the next step."""
    return base_lr * (1.0 - iter_num / max_iterations) ** 0.9


def hard_dice(pred: torch.Tensor, gt: torch.Tensor) -> float:
    """2|A&B|/(|A|+|B|) with the reference's empty-mask conventions (val_3D.py:85-97)."""
    pred, gt = pred.bool(), gt.bool()
    ps, gs = int(pred.sum()), int(gt.sum())
    if ps > 0 and gs > 0:
        return 2.0 * int((pred & gt).sum()) / (ps + gs)
    if ps == 0 and gs == 0:
        return 1.0
    return 0.0


# ----------------------------------------------------------------------------
# Parameter specs (state_dict names/shapes in the reference's registration order)
# ----------------------------------------------------------------------------


def unet_conv3_shapes(pre: str, cin: int, cout: int):
    out = []
    for blk, ci in (("conv1", cin), ("conv2", cout)):
        out.append((f"{pre}{blk}.0.weight", (cout, ci, 3, 3, 3)))
        out.append((f"{pre}{blk}.0.bias", (cout,)))
    return out


def backbone_shapes(nc: int, in_ch: int, feature_scale: int = 4):
    """Parameter list of unet_3D (networks/unet_3D.py:22-61): 38 tensors."""
    f = [int(x / feature_scale) for x in (64, 128, 256, 512, 1024)]
    out = []
    out += unet_conv3_shapes("conv1.", in_ch, f[0])
    out += unet_conv3_shapes("conv2.", f[0], f[1])
    out += unet_conv3_shapes("conv3.", f[1], f[2])
    out += unet_conv3_shapes("conv4.", f[2], f[3])
    out += unet_conv3_shapes("center.", f[3], f[4])
    out += unet_conv3_shapes("up_concat4.conv.", f[4] + f[3], f[3])
    out += unet_conv3_shapes("up_concat3.conv.", f[3] + f[2], f[2])
    out += unet_conv3_shapes("up_concat2.conv.", f[2] + f[1], f[1])
    out += unet_conv3_shapes("up_concat1.conv.", f[1] + f[0], f[0])
    out += [("final.weight", (nc, f[0], 1, 1, 1)), ("final.bias", (nc,))]
    return out


def aligner_shapes(pre: str, in_chans: Sequence[int], res: Sequence[int], nc: int, heads: Sequence[int],
                   q_name: str = "guided_Q"):
    """Parameters of InherentConsistent in registration order (networks/unet_3D_icl.py:178-200):
    guided_Q first (nn.Parameter registered... last in __init__, but named_parameters lists direct
    parameters of a module before its children), then the six ModuleLists."""
    out = [(f"{pre}{q_name}", (1, nc, in_chans[0]))]
    L = range(len(in_chans))
    for i in L:
        c = in_chans[i]
        out += [(f"{pre}proj_layers.{i}.weight", (c, c, 1, 1, 1)), (f"{pre}proj_layers.{i}.bias", (c,))]
    for i in L:
        c = in_chans[i]
        out += [(f"{pre}norm_layers.{i}.weight", (c,)), (f"{pre}norm_layers.{i}.bias", (c,))]
    for i in L:
        c, n = in_chans[i], res[i] ** 3
        cd = f"{pre}class_decoders.{i}."
        out += [(cd + "norm1.weight", (c,)), (cd + "norm1.bias", (c,)),
                (cd + "norm1_query.weight", (c,)), (cd + "norm1_query.bias", (c,)),
                (cd + "attn.fc_q.weight", (c, c)), (cd + "attn.fc_q.bias", (c,)),
                (cd + "attn.fc_kv.weight", (2 * c, c)), (cd + "attn.fc_kv.bias", (2 * c,)),
                (cd + "attn.proj.weight", (c, c)), (cd + "attn.proj.bias", (c,)),
                (cd + "norm2.weight", (c,)), (cd + "norm2.bias", (c,)),
                (cd + "mlp.fc1.weight", (4 * c, c)), (cd + "mlp.fc1.bias", (4 * c,)),
                (cd + "mlp.fc2.weight", (c, 4 * c)), (cd + "mlp.fc2.bias", (c,)),
                (cd + "norm3.weight", (n,)), (cd + "norm3.bias", (n,)),
                (cd + "mlp2.fc1.weight", (n, n)), (cd + "mlp2.fc1.bias", (n,)),
                (cd + "mlp2.fc2.weight", (n, n)), (cd + "mlp2.fc2.bias", (n,))]
    for i in L:
        h = heads[i]
        b = f"{pre}attn_convs0.{i}.block."
        out += [(b + "depthwise.weight", (h, 1, 3, 3, 3)),
                (b + "bn_depth.weight", (h,)), (b + "bn_depth.bias", (h,)),
                (b + "pointwise.weight", (h, h, 1, 1, 1)),
                (b + "bn_point.weight", (h,)), (b + "bn_point.bias", (h,))]
    for i in L:
        out += [(f"{pre}attn_convs1.{i}.weight", (1, heads[i], 1, 1, 1)), (f"{pre}attn_convs1.{i}.bias", (1,))]
    for i in L:
        c = in_chans[i]
        out += [(f"{pre}query_convs.{i}.weight", (c // 2, c, 1)), (f"{pre}query_convs.{i}.bias", (c // 2,))]
    return out


def aligner_buffers(pre: str, heads: Sequence[int]):
    out = {}
    for i, h in enumerate(heads):
        for bn in ("bn_depth", "bn_point"):
            b = f"{pre}attn_convs0.{i}.block.{bn}."
            out[b + "running_mean"] = torch.zeros(h)
            out[b + "running_var"] = torch.ones(h)
    return out


def unet_3d_icl_shapes(nc: int, in_ch: int = 1):
    f = (256, 128, 64)
    return (backbone_shapes(nc, in_ch)
            + aligner_shapes("sspa.", f, (6, 12, 24), nc, UNET3D_HEADS)
            + aligner_shapes("uscl.", f, (6, 12, 24), nc, UNET3D_HEADS))


def make_params(shapes, base_seed: int = 1337, requires_grad: bool = False, strip: str = "") -> P:
    """Hash-filled parameter dict (icl_amd.utils.hashfill); `strip` = prefix to drop from the
    name before seeding (so a sub-module can be filled as if it were the root)."""
    from icl_amd.utils.hashfill import fill_like_reference_init
    p = {k: torch.empty(*s) for k, s in shapes}
    fill_like_reference_init([(k[len(strip):] if strip and k.startswith(strip) else k, t)
                              for k, t in p.items()], base_seed)
    if requires_grad:
        for t in p.values():
            t.requires_grad_()
    return p


# ----------------------------------------------------------------------------
# 2-D U-Net ICL (BASELINE config 1): networks/unet_icl.py, networks/unet.py
# ----------------------------------------------------------------------------

UNET2D_FT = (16, 32, 64, 128, 256)
UNET2D_HEADS = (8, 4, 2)          # params['num_heads'][::-1], unet_icl.py:214
UNET2D_RES = (32, 64, 128)        # input_resolution[1:4] for 256^2 inputs, unet_icl.py:213


def conv_block2d(p: P, pre: str, x: torch.Tensor, training: bool) -> torch.Tensor:
    """ConvBlock, networks/unet_icl.py:39-57: 2 x [Conv2d 3x3 -> BatchNorm2d -> LeakyReLU(0.01)], dropout prob 0."""
    for i, j in ((0, 1), (4, 5)):
        x = F.conv2d(x, p[f"{pre}.conv_conv.{i}.weight"], p[f"{pre}.conv_conv.{i}.bias"], padding=1)
        x = F.leaky_relu(_bn_train(p, f"{pre}.conv_conv.{j}", x, training), 0.01)
    return x


def up_block2d(p: P, pre: str, x1: torch.Tensor, x2: torch.Tensor, training: bool) -> torch.Tensor:
    """UpBlock with the default bilinear=True (Decoder never forwards the flag), unet_icl.py:75-96:
    conv1x1 -> bilinear x2 align_corners=True -> cat([skip, up]) -> ConvBlock."""
    x1 = F.conv2d(x1, p[f"{pre}.conv1x1.weight"], p[f"{pre}.conv1x1.bias"])
    x1 = F.interpolate(x1, scale_factor=2, mode="bilinear", align_corners=True)
    return conv_block2d(p, f"{pre}.conv", torch.cat([x2, x1], 1), training)


def backbone2d(p: P, x: torch.Tensor, training: bool):
    """Encoder + Decoder, unet_icl.py:128-193.  Returns (logits, [x_1, x_2, x_3])."""
    x0 = conv_block2d(p, "encoder.in_conv", x, training)
    xs = [x0]
    for i in range(1, 5):
        xs.append(conv_block2d(p, f"encoder.down{i}.maxpool_conv.1", F.max_pool2d(xs[-1], 2), training))
    x_1 = up_block2d(p, "decoder.up1", xs[4], xs[3], training)
    x_2 = up_block2d(p, "decoder.up2", x_1, xs[2], training)
    x_3 = up_block2d(p, "decoder.up3", x_2, xs[1], training)
    x = up_block2d(p, "decoder.up4", x_3, xs[0], training)
    out = F.conv2d(x, p["decoder.out_conv.weight"], p["decoder.out_conv.bias"], padding=1)
    return out, [x_1, x_2, x_3]


def unet_icl_2d_forward(p: P, x_lab, x_unlab=None, inference=False, training=True):
    """UNet_icl.forward, unet_icl.py:235-252: two separate backbone passes (BatchNorm sees each half on its own)."""
    out_lab, feats_lab = backbone2d(p, x_lab, training)
    if inference:
        return out_lab
    out_unlab, feats_unlab = backbone2d(p, x_unlab, training)
    maps_lab, qs_lab = inherent_consistent(p, "sspa", feats_lab, UNET2D_HEADS, None, "labeled", training)
    maps_con, _ = inherent_consistent(p, "sspa", feats_unlab, UNET2D_HEADS, None, "labeled", training)
    maps_unlab, _ = inherent_consistent(p, "uscl", feats_unlab, UNET2D_HEADS, qs_lab, "unlabeled", training)
    return out_lab, out_unlab, maps_lab, maps_unlab, maps_con


def _resize2d(m, size):
    return F.interpolate(m.float(), size=list(size), mode="bilinear")


def icl_losses_2d(outputs, labels_lab: torch.Tensor, n_classes: int, size=(256, 256), w_con: float = 50.0):
    """train_inherent_consistent_unet_2D.py:117-127: ce + dice(softmax=True) + AuxLoss + PseudoSoftLoss + 50*consistency
    (AuxLoss / PseudoSoftLoss: utils/losses.py:233-251,273-285 with resize=patch_size)."""
    out_lab, out_unlab, maps_lab, maps_unlab, maps_con = outputs
    l_ce = F.cross_entropy(out_lab, labels_lab.long())
    l_dice = dice_loss(out_lab, labels_lab.unsqueeze(1), n_classes, softmax=True)
    l_aux = 0.0
    for m in maps_lab:
        r = _resize2d(m, size)
        l_aux = l_aux + F.cross_entropy(r, labels_lab.long()) + dice_loss(r, labels_lab.unsqueeze(1), n_classes, softmax=True)
    l_aux = l_aux / len(maps_lab)
    tgt = out_unlab.detach()
    l_pse = sum(softmax_dice_loss(_resize2d(m, size), tgt) for m in maps_unlab) / len(maps_unlab)
    l_con = softmax_mse_loss(maps_unlab, maps_con)
    total = l_ce + l_dice + l_aux + l_pse + w_con * l_con
    return total, dict(ce=l_ce, dice=l_dice, aux=l_aux, pse=l_pse, con=l_con)


def conv_block2d_shapes(pre, cin, cout):
    out = []
    for i, j, ci in ((0, 1, cin), (4, 5, cout)):
        out += [(f"{pre}conv_conv.{i}.weight", (cout, ci, 3, 3)), (f"{pre}conv_conv.{i}.bias", (cout,)),
                (f"{pre}conv_conv.{j}.weight", (cout,)), (f"{pre}conv_conv.{j}.bias", (cout,))]
    return out


def backbone2d_shapes(nc: int, in_ch: int = 1):
    f = UNET2D_FT
    out = conv_block2d_shapes("encoder.in_conv.", in_ch, f[0])
    for i in range(1, 5):
        out += conv_block2d_shapes(f"encoder.down{i}.maxpool_conv.1.", f[i - 1], f[i])
    for i, (c1, c2) in enumerate(((f[4], f[3]), (f[3], f[2]), (f[2], f[1]), (f[1], f[0])), start=1):
        out += [(f"decoder.up{i}.conv1x1.weight", (c2, c1, 1, 1)), (f"decoder.up{i}.conv1x1.bias", (c2,))]
        out += conv_block2d_shapes(f"decoder.up{i}.conv.", c2 * 2, c2)
    out += [("decoder.out_conv.weight", (nc, f[0], 3, 3)), ("decoder.out_conv.bias", (nc,))]
    return out


def backbone2d_buffers(in_ch: int = 1):
    out = {}
    for k, s in backbone2d_shapes(2, in_ch):
        if k.endswith((".1.weight", ".5.weight")) and "conv_conv" in k:
            b = k[: -len("weight")]
            out[b + "running_mean"] = torch.zeros(s)
            out[b + "running_var"] = torch.ones(s)
    return out


def aligner_shapes_nd(pre, in_chans, res, nc, heads, dims):
    """aligner_shapes with conv weights of the right rank (dims = 2: unet_icl.py:253-300)."""
    out = []
    for k, s in aligner_shapes(pre, in_chans, [1] * len(res), nc, heads):
        out.append((k, s))
    fixed = []
    n_tok = [r ** dims for r in res]
    for k, s in out:
        if ".norm3." in k or ".mlp2." in k:
            i = int(k.split("class_decoders.")[1].split(".")[0])
            s = tuple(n_tok[i] if v == 1 else v for v in s)
        if dims == 2 and len(s) == 5:
            s = s[:2] + s[3:]
        fixed.append((k, s))
    return fixed


def unet_icl_2d_shapes(nc: int, in_ch: int = 1):
    f = (UNET2D_FT[3], UNET2D_FT[2], UNET2D_FT[1])
    return (backbone2d_shapes(nc, in_ch)
            + aligner_shapes_nd("sspa.", f, UNET2D_RES, nc, UNET2D_HEADS, 2)
            + aligner_shapes_nd("uscl.", f, UNET2D_RES, nc, UNET2D_HEADS, 2))
