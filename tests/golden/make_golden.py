#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REAL reference (run in the build container only).

    python tests/golden/make_golden.py [--only unit|model2|model16|wgrads|steps|steps10|model2d|swin|swin64|swinunet2d]

The reference (/root/reference/code) is imported as-is; the only stand-ins are the five
trivial MONAI symbols its U-Net ICL files import: MONAI is
not installed and there is no network.  Nothing from the reference is copied: the .npz
files hold seeds, shapes and OUTPUT numbers only.  Weights and inputs come from
icl_amd.utils.hashfill (bit-identical on every platform), keyed by state_dict name.

Parity mode: every nn.Dropout.p and DropPath.drop_prob forced to 0, model
kept in train() so BatchNorm uses batch statistics; eval() only for the inference branch.
"""
import argparse
import importlib
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from icl_amd.utils.hashfill import (fill_like_reference_init, synthetic_labels,  # noqa: E402
                                    synthetic_volume)

REF = "/root/reference/code"


def install_monai_stub():
    class DropPath(nn.Module):
        def __init__(self, drop_prob=0.0, scale_by_keep=True):
            super().__init__()
            self.drop_prob, self.scale_by_keep = drop_prob, scale_by_keep

        def forward(self, x):
            if self.drop_prob == 0.0 or not self.training:
                return x
            keep = 1 - self.drop_prob
            m = x.new_empty((x.shape[0],) + (1,) * (x.ndim - 1)).bernoulli_(keep)
            return x * (m / keep if self.scale_by_keep else m)

    class _Conv:
        CONV = "conv"

        def __getitem__(self, key):
            return {1: nn.Conv1d, 2: nn.Conv2d, 3: nn.Conv3d}[key[1]]

    def ensure_tuple_rep(x, d):
        return tuple(x) if isinstance(x, (tuple, list)) else (x,) * d

    def optional_import(mod, name=""):
        m = importlib.import_module(mod)
        return (getattr(m, name) if name else m), True

    monai = types.ModuleType("monai")
    nets = types.ModuleType("monai.networks")
    layers = types.ModuleType("monai.networks.layers")
    utils = types.ModuleType("monai.utils")
    layers.DropPath, layers.Conv, layers.trunc_normal_ = DropPath, _Conv(), nn.init.trunc_normal_
    utils.ensure_tuple_rep, utils.optional_import = ensure_tuple_rep, optional_import
    monai.networks, monai.utils, nets.layers = nets, utils, layers
    sys.modules.update({"monai": monai, "monai.networks": nets,
                        "monai.networks.layers": layers, "monai.utils": utils})


def install_monai_block_stubs():
    """SwinUNETR-ICL only: the five MONAI 1.0.1 blocks the reference imports at
    swinunetr_icl.py:22-23 are not available (MONAI is not installed, no network).  These stand-ins follow the published
    MONAI 1.0.1 definitions (networks/blocks/{mlp,patchembedding,dynunet_block,unetr_block}.py) and keep MONAI's
    state_dict key names.  The goldens made with them pin the reference's OWN vendored Swin code (rows S1-S4), its model
    wiring and the aligners (S6); the res-block / up-block arithmetic (S5) stays "parity unpinned"."""
    from collections import OrderedDict

    def _conv(cin, cout, k, stride=1, transposed=False, bias=False):
        if transposed:
            c = nn.ConvTranspose3d(cin, cout, kernel_size=k, stride=stride, padding=0, output_padding=0, bias=bias)
        else:
            c = nn.Conv3d(cin, cout, kernel_size=k, stride=stride, padding=(k - stride + 1) // 2 if k > 1 else 0, bias=bias)
        return nn.Sequential(OrderedDict(conv=c))   # monai Convolution(conv_only=True): one child named "conv"

    class MLPBlock(nn.Module):
        def __init__(self, hidden_size, mlp_dim, dropout_rate=0.0, act="GELU", dropout_mode="vit"):
            super().__init__()
            self.linear1 = nn.Linear(hidden_size, mlp_dim)
            self.linear2 = nn.Linear(mlp_dim, hidden_size)
            self.fn = nn.GELU()
            self.drop1 = nn.Dropout(dropout_rate)
            self.drop2 = nn.Dropout(dropout_rate) if dropout_mode == "swin" else self.drop1

        def forward(self, x):
            return self.drop2(self.linear2(self.drop1(self.fn(self.linear1(x)))))

    class PatchEmbed(nn.Module):
        def __init__(self, patch_size=2, in_chans=1, embed_dim=48, norm_layer=None, spatial_dims=3):
            super().__init__()
            self.patch_size = tuple(patch_size)
            self.proj = nn.Conv3d(in_chans, embed_dim, kernel_size=self.patch_size, stride=self.patch_size)
            self.norm = norm_layer(embed_dim) if norm_layer is not None else None

        def forward(self, x):
            assert all(s % p == 0 for s, p in zip(x.shape[2:], self.patch_size)) and self.norm is None
            return self.proj(x)

    class UnetResBlock(nn.Module):
        def __init__(self, spatial_dims, in_channels, out_channels, kernel_size, stride, norm_name, **kw):
            super().__init__()
            assert norm_name == "instance" and spatial_dims == 3
            self.conv1 = _conv(in_channels, out_channels, kernel_size, stride)
            self.conv2 = _conv(out_channels, out_channels, kernel_size, 1)
            self.lrelu = nn.LeakyReLU(negative_slope=0.01, inplace=True)
            self.norm1 = nn.InstanceNorm3d(out_channels)
            self.norm2 = nn.InstanceNorm3d(out_channels)
            self.downsample = in_channels != out_channels or stride != 1
            if self.downsample:
                self.conv3 = _conv(in_channels, out_channels, 1, stride)
                self.norm3 = nn.InstanceNorm3d(out_channels)

        def forward(self, inp):
            residual = inp
            out = self.lrelu(self.norm1(self.conv1(inp)))
            out = self.norm2(self.conv2(out))
            if self.downsample:
                residual = self.norm3(self.conv3(residual))
            out += residual
            return self.lrelu(out)

    class UnetrBasicBlock(nn.Module):
        def __init__(self, spatial_dims, in_channels, out_channels, kernel_size, stride, norm_name, res_block=False):
            super().__init__()
            assert res_block
            self.layer = UnetResBlock(spatial_dims, in_channels, out_channels, kernel_size, stride, norm_name)

        def forward(self, inp):
            return self.layer(inp)

    class UnetrUpBlock(nn.Module):
        def __init__(self, spatial_dims, in_channels, out_channels, kernel_size, upsample_kernel_size, norm_name, res_block=False):
            super().__init__()
            assert res_block
            self.transp_conv = _conv(in_channels, out_channels, upsample_kernel_size, upsample_kernel_size, transposed=True)
            self.conv_block = UnetResBlock(spatial_dims, out_channels + out_channels, out_channels, kernel_size, 1, norm_name)

        def forward(self, inp, skip):
            out = self.transp_conv(inp)
            return self.conv_block(torch.cat((out, skip), dim=1))

    class UnetOutBlock(nn.Module):
        def __init__(self, spatial_dims, in_channels, out_channels, dropout=None):
            super().__init__()
            self.conv = _conv(in_channels, out_channels, 1, 1, bias=True)

        def forward(self, inp):
            return self.conv(inp)

    blocks = types.ModuleType("monai.networks.blocks")
    blocks.MLPBlock, blocks.PatchEmbed, blocks.UnetOutBlock = MLPBlock, PatchEmbed, UnetOutBlock
    blocks.UnetrBasicBlock, blocks.UnetrUpBlock = UnetrBasicBlock, UnetrUpBlock
    sys.modules["monai.networks.blocks"] = blocks
    sys.modules["monai.networks"].blocks = blocks


def import_reference():
    install_monai_stub()
    sys.argv = ["x"]
    sys.path.insert(0, REF)
    from networks.unet_3D import unet_3D
    from networks import unet_3D_icl as m3
    from networks.utils import UnetConv3, UnetUp3_CT
    from utils import losses
    return dict(unet_3D=unet_3D, m3=m3, UnetConv3=UnetConv3, UnetUp3_CT=UnetUp3_CT, losses=losses)


def parity_mode(model):
    for m in model.modules():
        if isinstance(m, nn.Dropout):
            m.p = 0.0
        if m.__class__.__name__ == "DropPath":
            m.drop_prob = 0.0
    return model


def fill(model, seed=1337):
    fill_like_reference_init(list(model.named_parameters()), seed)


def npy(t):
    return t.detach().cpu().numpy()


def grads_of(model):
    return {k: (None if p.grad is None else npy(p.grad)) for k, p in model.named_parameters()}


# ---------------------------------------------------------------- unit vectors
def gen_unit(R, out):
    torch.manual_seed(0)
    d = {}
    # A1: UnetConv3(3->4) on [2,3,8,8,8]
    m = R["UnetConv3"](3, 4, True, kernel_size=(3, 3, 3), padding_size=(1, 1, 1))
    fill(m)
    x = synthetic_volume((2, 3, 8, 8, 8), 11).requires_grad_()
    y = m(x)
    gy = synthetic_volume(tuple(y.shape), 12)
    y.backward(gy)
    d["conv3_y"], d["conv3_gx"] = npy(y), npy(x.grad)
    for k, g in grads_of(m).items():
        d["conv3_g." + k] = g
    # A3: UnetUp3_CT(8, 4): skip [2,4,8,8,8], deep [2,8,4,4,4]
    m = R["UnetUp3_CT"](8, 4, True)
    fill(m)
    s = synthetic_volume((2, 4, 8, 8, 8), 13).requires_grad_()
    dd = synthetic_volume((2, 8, 4, 4, 4), 14).requires_grad_()
    y = m(s, dd)
    y.backward(synthetic_volume(tuple(y.shape), 15))
    d["up_y"], d["up_gskip"], d["up_gdeep"] = npy(y), npy(s.grad), npy(dd.grad)
    for k, g in grads_of(m).items():
        d["up_g." + k] = g
    # A2/A5/A7: plain unet_3D nc=3 in_channels=2 on [1,2,32,32,32] (eval: dropout identity)
    m = R["unet_3D"](n_classes=3, in_channels=2)
    fill(m)
    m.eval()
    x = synthetic_volume((1, 2, 32, 32, 32), 16).requires_grad_()
    y = m(x)
    y.backward(synthetic_volume(tuple(y.shape), 17))
    d["unet32_y"], d["unet32_gx"] = npy(y), npy(x.grad)
    d["unet32_keys"] = np.array(list(m.state_dict().keys()))
    for k, g in grads_of(m).items():
        d["unet32_gn." + k] = np.float64(np.linalg.norm(g.astype(np.float64)))
    d["unet32_g.final.weight"] = grads_of(m)["final.weight"]
    d["unet32_g.conv1.conv1.0.weight"] = grads_of(m)["conv1.conv1.0.weight"]
    d["unet32_g.center.conv2.0.weight_sub"] = grads_of(m)["center.conv2.0.weight"][::16, ::16]
    # B1-B6: InherentConsistent tiny, both modes
    m3 = R["m3"]
    for tag, train in (("tr", True), ("ev", False)):
        al = m3.InherentConsistent(in_chans=(32, 16, 8), depths=(2, 2, 2), patch_size=(2, 2, 2),
                                   input_resolution=[2, 4, 8], num_classes=3, num_heads=(4, 2, 1))
        parity_mode(al)
        fill(al)
        al.train(train)
        feats = [synthetic_volume((2, 32, 2, 2, 2), 21).requires_grad_(),
                 synthetic_volume((2, 16, 4, 4, 4), 22).requires_grad_(),
                 synthetic_volume((2, 8, 8, 8, 8), 23).requires_grad_()]
        maps, qs = al(feats, "labeled")
        maps_u, qs_u = al(feats, qs, "unlabeled")
        loss = sum((mm * synthetic_volume(tuple(mm.shape), 30 + i)).sum() for i, mm in enumerate(maps)) \
            + sum((mm * synthetic_volume(tuple(mm.shape), 40 + i)).sum() for i, mm in enumerate(maps_u))
        loss.backward()
        for i in range(3):
            d[f"al_{tag}_map{i}"], d[f"al_{tag}_q{i}"] = npy(maps[i]), npy(qs[i])
            d[f"al_{tag}_mapu{i}"], d[f"al_{tag}_qu{i}"] = npy(maps_u[i]), npy(qs_u[i])
            d[f"al_{tag}_gfeat{i}"] = npy(feats[i].grad)
        for k, g in grads_of(al).items():
            if g is None:
                continue
            d[f"al_{tag}_gn." + k] = np.float64(np.linalg.norm(g.astype(np.float64)))
            if g.size <= 4096:
                d[f"al_{tag}_g." + k] = g
            else:  # big token-axis MLPs: strided sample
                d[f"al_{tag}_gs." + k] = g.reshape(-1)[::37].copy()
        d[f"al_{tag}_none"] = np.array([k for k, g in grads_of(al).items() if g is None])
        if train:  # BN running stats after the two train-mode calls
            for k, v in al.state_dict().items():
                if "running" in k:
                    d["al_tr_buf." + k] = npy(v)
    # single modules: Query_Attention(8, heads 2), Class_Decoder(dim 8, res 2^3)
    qa = m3.Query_Attention(8, num_heads=2, qkv_bias=True)
    fill(qa)
    q = synthetic_volume((2, 3, 8), 51).requires_grad_()
    xx = synthetic_volume((2, 8, 8), 52).requires_grad_()
    o, a = qa(q, xx)
    (o.sum() + (a * synthetic_volume(tuple(a.shape), 53)).sum()).backward()
    d["qa_o"], d["qa_a"], d["qa_gq"], d["qa_gx"] = npy(o), npy(a), npy(q.grad), npy(xx.grad)
    # L1-L5 at the hard-coded 96^3
    L = R["losses"]
    nc = 3
    lab = synthetic_labels((1, 96, 96, 96), 61, nc)
    logit = synthetic_volume((1, nc, 96, 96, 96), 62).requires_grad_()
    dl = L.DiceLoss(nc)
    l1 = dl(torch.softmax(logit, 1), lab.unsqueeze(1))
    l2 = nn.CrossEntropyLoss()(logit, lab)
    (l1 + l2).backward()
    d["loss_dice"], d["loss_ce"] = npy(l1), npy(l2)
    d["loss_dice_ce_glogit_sub"] = npy(logit.grad)[:, :, ::8, ::8, ::8]
    maps = [synthetic_volume((1, nc, r, r, r), 63 + i).requires_grad_() for i, r in enumerate((6, 12, 24))]
    la = L.AuxLoss3D(nc)(maps, lab)
    la.backward()
    d["loss_aux"] = npy(la)
    for i in range(3):
        d[f"loss_aux_g{i}"] = npy(maps[i].grad)
        maps[i].grad = None
    pred = synthetic_volume((1, nc, 96, 96, 96), 70).requires_grad_()
    lp = L.PseudoSoftLoss3D(nc)(maps, pred)
    lp.backward()
    d["loss_pse"] = npy(lp)
    assert pred.grad is None
    for i in range(3):
        d[f"loss_pse_g{i}"] = npy(maps[i].grad)
        maps[i].grad = None
    maps_b = [synthetic_volume((1, nc, r, r, r), 73 + i).requires_grad_() for i, r in enumerate((6, 12, 24))]
    lc = L.softmax_mse_loss(maps, maps_b)
    lc.backward()
    d["loss_con"] = npy(lc)
    assert maps_b[0].grad is None
    for i in range(3):
        d[f"loss_con_g{i}"] = npy(maps[i].grad)
    np.savez_compressed(os.path.join(out, "unit.npz"), **d)
    print("unit.npz", sum(v.nbytes for v in d.values()) / 1e6, "MB (raw)")


# ---------------------------------------------------------------- full model vectors
def gen_model(R, out, nc):
    m3, L = R["m3"], R["losses"]
    model = m3.unet_3D_icl(n_classes=nc, in_channels=1)
    parity_mode(model)
    fill(model)
    d = {}
    d["keys"] = np.array(list(model.state_dict().keys()))
    d["param_keys"] = np.array([k for k, _ in model.named_parameters()])
    vol = synthetic_volume((2, 1, 96, 96, 96), 1337)
    lab = synthetic_labels((1, 96, 96, 96), 4242, nc)
    # inference branch, eval mode
    model.eval()
    with torch.no_grad():
        y = model(vol[:1], inference=True)
    d["inf_logits_sub"] = npy(y)[:, :, ::8, ::8, ::8]
    d["inf_logits_mean"] = npy(y.double().mean(dim=(0, 2, 3, 4)))
    d["inf_logits_l2"] = npy(y.double().pow(2).sum(dim=(0, 2, 3, 4)).sqrt())
    d["inf_logits_slab"] = npy(y)[:, :, 40:44, 40:56, 40:56]
    # training step, train mode (dropout / drop-path prob 0)
    model.train()
    outs = model(vol[:1], vol[1:])
    for name, t in (("final_lab", outs[0]), ("final_unlab", outs[1])):
        d[name + "_sub"] = npy(t)[:, :, ::8, ::8, ::8]
        d[name + "_l2"] = npy(t.double().pow(2).sum(dim=(0, 2, 3, 4)).sqrt())
    for name, lst in (("maps_lab", outs[2]), ("maps_unlab", outs[3]), ("maps_con", outs[4])):
        for i, t in enumerate(lst):
            d[f"{name}{i}"] = npy(t)
    soft = torch.softmax(outs[0], 1)
    l_ce = nn.CrossEntropyLoss()(outs[0], lab)
    l_dice = L.DiceLoss(nc)(soft, lab.unsqueeze(1))
    l_aux = L.AuxLoss3D(nc)(outs[2], lab)
    l_pse = L.PseudoSoftLoss3D(nc)(outs[3], outs[1])
    l_con = L.softmax_mse_loss(outs[3], outs[4])
    w_pse = 0.1 if nc == 16 else 1.0  # AMOS trainer weight (…AMOS22.py:230)
    loss = l_dice + l_ce + l_aux + w_pse * l_pse + 10 * l_con
    d["losses"] = np.array([float(l_dice), float(l_ce), float(l_aux), float(l_pse), float(l_con), float(loss)])
    # soft dice per class of the labeled prediction (metric parity, 1e-4)
    sd = []
    for c in range(nc):
        t = (lab == c).float()
        p = soft[:, c]
        sd.append(float((2 * (p * t).sum() + 1e-5) / ((p * p).sum() + (t * t).sum() + 1e-5)))
    d["soft_dice"] = np.array(sd)
    opt = torch.optim.SGD(model.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4)
    opt.zero_grad()
    loss.backward()
    gn, none = {}, []
    for k, p in model.named_parameters():
        if p.grad is None:
            none.append(k)
        else:
            gn[k] = float(p.grad.double().pow(2).sum().sqrt())
    d["grad_none"] = np.array(none)
    d["grad_norm_keys"] = np.array(list(gn.keys()))
    d["grad_norms"] = np.array(list(gn.values()))
    sd_ = dict(model.named_parameters())
    for k in ("final.weight", "final.bias", "conv1.conv1.0.weight", "sspa.guided_Q",
              "sspa.class_decoders.0.attn.fc_q.weight", "uscl.attn_convs1.2.weight",
              "sspa.attn_convs0.2.block.depthwise.weight", "sspa.query_convs.0.weight"):
        d["grad." + k] = npy(sd_[k].grad)
    d["grad.sspa.class_decoders.2.mlp2.fc1.weight_sub"] = npy(
        sd_["sspa.class_decoders.2.mlp2.fc1.weight"].grad)[::432, ::432]
    d["grad.center.conv2.0.weight_sub"] = npy(sd_["center.conv2.0.weight"].grad)[::16, ::16]
    opt.step()
    d["post_sgd_norms"] = np.array([float(p.detach().double().pow(2).sum().sqrt())
                                    for _, p in model.named_parameters()])
    d["post_sgd.final.weight"] = npy(sd_["final.weight"])
    d["post_sgd.conv1.conv1.0.weight"] = npy(sd_["conv1.conv1.0.weight"])
    np.savez_compressed(os.path.join(out, f"model_unet3d_icl_nc{nc}.npz"), **d)
    print(f"model nc={nc}: losses", d["losses"], "grad_none", len(none))


# ---------------------------------------------------------------- three consecutive trainer steps (row T1)
STEPS_MAX_ITER = 10      # a short schedule so that the poly learning rate moves visibly within three steps


def gen_model_dense_wgrads(R, out, nc=2):
    """Round 5: the weight gradients of the 96^3 / 48^3 3x3x3 convolutions of the reference step IN FULL — the
    tensors the split-product weight-gradient kernel produces at its largest shapes — so that the norm-only bands of the model golden are
    backed by dense elementwise comparisons.  Same model, inputs and loss as gen_model(nc = 2)."""
    m3, L = R["m3"], R["losses"]
    model = m3.unet_3D_icl(n_classes=nc, in_channels=1)
    parity_mode(model)
    fill(model)
    vol = synthetic_volume((2, 1, 96, 96, 96), 1337)
    lab = synthetic_labels((1, 96, 96, 96), 4242, nc)
    model.train()
    # round 6: two INPUT gradients of 96^3 convolutions as well (the tensors the input-gradient kernels produce at their largest shapes):
    # the inputs of `up_concat1.conv.conv1` (48 channels: the concat buffer) and of `conv1.conv2` (16 channels), kept by forward
    # pre-hooks on the Conv3d modules; each sample's gradient is stored as a dense 12^3 block plus a stride-8 lattice of the volume
    kept = {}

    def keep(name):
        def hook(module, args):
            args[0].retain_grad()
            kept.setdefault(name, []).append(args[0])       # (the backbone runs once per stream: labeled, then unlabeled)
        return hook
    hooks = [model.up_concat1.conv.conv1[0].register_forward_pre_hook(keep("up_concat1.conv.conv1")),
             model.conv1.conv2[0].register_forward_pre_hook(keep("conv1.conv2"))]
    outs = model(vol[:1], vol[1:])
    for h in hooks:
        h.remove()
    soft = torch.softmax(outs[0], 1)
    loss = (L.DiceLoss(nc)(soft, lab.unsqueeze(1)) + nn.CrossEntropyLoss()(outs[0], lab) + L.AuxLoss3D(nc)(outs[2], lab)
            + L.PseudoSoftLoss3D(nc)(outs[3], outs[1]) + 10 * L.softmax_mse_loss(outs[3], outs[4]))
    loss.backward()
    sd_ = dict(model.named_parameters())
    d = {"loss": np.array([float(loss)])}
    for k in ("up_concat1.conv.conv1.0.weight", "up_concat1.conv.conv2.0.weight", "conv1.conv2.0.weight",
              "up_concat2.conv.conv1.0.weight", "conv2.conv2.0.weight"):
        d["grad." + k] = npy(sd_[k].grad)
    for name, ts in kept.items():
        assert len(ts) == 2 and ts[0].grad is not None, name
        # (the unlabeled stream's upper decoder feeds only `final_unlab`, which the losses read as a detached pseudo-label: its input
        # gradient does not exist in the reference — the stream index of every stored gradient is recorded)
        have = [i for i, t in enumerate(ts) if t.grad is not None]
        g = torch.cat([ts[i].grad for i in have], 0)         # [streams with a gradient, C, 96, 96, 96]
        d[f"dgrad.{name}.streams"] = np.array(have)
        d[f"dgrad.{name}.block"] = npy(g[:, :, 40:52, 40:52, 40:52])
        d[f"dgrad.{name}.lattice"] = npy(g[:, :, 3::8, 3::8, 3::8])
        d[f"dgrad.{name}.maxabs"] = np.array([float(g.abs().max())])
    np.savez_compressed(os.path.join(out, f"model_unet3d_icl_nc{nc}_wgrads.npz"), **d)
    print("dense wgrads:", {k: v.shape for k, v in d.items()}, sum(v.nbytes for v in d.values()) / 1e6, "MB (raw)")


def gen_model_steps(R, out, nc=2, steps=3):
    """Three iterations of the reference loop body (train_inherent_consistent_unet_3D_BraTS.py:99-121) on the real 785 M-parameter
    model: SGD(lr 0.01, momentum 0.9, wd 1e-4), momentum carried over, a new batch per step, and the poly learning rate computed
    from the PRE-increment iter_num after each optimiser step — so steps 1 and 2 both run at base_lr and step 3 at
    base_lr * (1 - 1/max_iter)**0.9.  Parity mode (dropout / drop-path 0)."""
    m3, L = R["m3"], R["losses"]
    model = m3.unet_3D_icl(n_classes=nc, in_channels=1)
    parity_mode(model)
    fill(model)
    model.train()
    base_lr, max_iterations = 0.01, STEPS_MAX_ITER
    optimizer = torch.optim.SGD(model.parameters(), lr=base_lr, momentum=0.9, weight_decay=0.0001)
    ce_loss, dice_loss = nn.CrossEntropyLoss(), L.DiceLoss(nc)
    aux_loss, pse_loss = L.AuxLoss3D(nc), L.PseudoSoftLoss3D(nc)
    named = dict(model.named_parameters())
    big = "sspa.class_decoders.2.mlp2.fc1.weight"
    w0_big = named[big].detach()[::432, ::432].double().clone()
    d = {"param_keys": np.array(list(named.keys())), "max_iterations": np.array(max_iterations), "base_lr": np.array(base_lr)}
    losses, lrs = [], []
    iter_num = 0
    for s in range(steps):
        vol = synthetic_volume((2, 1, 96, 96, 96), 1337 + s)
        lab = synthetic_labels((1, 96, 96, 96), 4242 + s, nc)
        outputs = model(vol[:1], vol[1:])
        soft = torch.softmax(outputs[0], dim=1)
        l_ce = ce_loss(outputs[0], lab[:1])
        l_dice = dice_loss(soft, lab[:1].unsqueeze(1))
        l_aux = aux_loss(outputs[2], lab[:1])
        l_pse = pse_loss(outputs[3], outputs[1])
        l_con = L.softmax_mse_loss(outputs[3], outputs[4])
        loss = l_dice + l_ce + l_aux + l_pse + 10 * l_con
        lrs.append(optimizer.param_groups[0]["lr"])
        optimizer.zero_grad()
        loss.backward()
        optimizer.step()
        lr_ = base_lr * (1.0 - iter_num / max_iterations) ** 0.9
        for g in optimizer.param_groups:
            g["lr"] = lr_
        iter_num += 1
        losses.append([float(l_dice), float(l_ce), float(l_aux), float(l_pse), float(l_con), float(loss)])
        d[f"post_step{s + 1}_norms"] = np.array([float(p.detach().double().pow(2).sum().sqrt()) for p in named.values()])
        print(f"steps: step {s + 1} lr {lrs[-1]:.6f} losses {losses[-1]}")
    d["losses"] = np.array(losses)
    d["lr_used"] = np.array(lrs)
    for k in ("final.weight", "final.bias", "conv1.conv1.0.weight", "sspa.class_decoders.0.attn.fc_q.weight", "uscl.attn_convs1.2.weight"):
        d["post_step3." + k] = npy(named[k])
    d["post_step3." + big + "_sub"] = npy(named[big])[::432, ::432]
    d["delta_step3." + big + "_sub"] = (named[big].detach()[::432, ::432].double() - w0_big).numpy()
    mom_keys = [k for k, p in named.items() if "momentum_buffer" in optimizer.state.get(p, {})]
    d["momentum_keys"] = np.array(mom_keys)
    d["momentum_norms"] = np.array([float(optimizer.state[named[k]]["momentum_buffer"].double().pow(2).sum().sqrt()) for k in mom_keys])
    d["momentum." + big + "_sub"] = npy(optimizer.state[named[big]]["momentum_buffer"])[::432, ::432]
    d["momentum.final.weight"] = npy(optimizer.state[named["final.weight"]]["momentum_buffer"])
    np.savez_compressed(os.path.join(out, f"model_unet3d_icl_nc{nc}_steps.npz"), **d)


def gen_model_steps10(R, out, nc=2, steps=10, max_iterations=20):
    """Round 6: TEN iterations of the reference loop body (train_inherent_consistent_unet_3D_BraTS.py:99-121) on
    the 785 M-parameter model, every step recorded — six loss terms, the learning rate used, all parameter norms, and per step the small
    tensors and sampled 13,824^2 weights / momentum the three-step golden holds only after its last step.  A longer horizon than
    gen_model_steps (kept as it is: its file pins steps 1-3 at max_iterations = 10); here max_iterations = 20, so the ten steps run at
    base_lr * (1 - k / 20) ** 0.9 for k = 0, 0, 1, ..., 8 (the reference computes the rate from the PRE-increment iter_num)."""
    m3, L = R["m3"], R["losses"]
    model = m3.unet_3D_icl(n_classes=nc, in_channels=1)
    parity_mode(model)
    fill(model)
    model.train()
    base_lr = 0.01
    optimizer = torch.optim.SGD(model.parameters(), lr=base_lr, momentum=0.9, weight_decay=0.0001)
    ce_loss, dice_loss = nn.CrossEntropyLoss(), L.DiceLoss(nc)
    aux_loss, pse_loss = L.AuxLoss3D(nc), L.PseudoSoftLoss3D(nc)
    named = dict(model.named_parameters())
    big = "sspa.class_decoders.2.mlp2.fc1.weight"
    w0_big = named[big].detach()[::432, ::432].double().clone()
    small = ("final.weight", "final.bias", "sspa.class_decoders.0.attn.fc_q.weight", "uscl.attn_convs1.2.weight")
    d = {"param_keys": np.array(list(named.keys())), "max_iterations": np.array(max_iterations), "base_lr": np.array(base_lr),
         "small_keys": np.array(small)}
    losses, lrs = [], []
    iter_num = 0
    for s in range(steps):
        vol = synthetic_volume((2, 1, 96, 96, 96), 1337 + s)
        lab = synthetic_labels((1, 96, 96, 96), 4242 + s, nc)
        outputs = model(vol[:1], vol[1:])
        soft = torch.softmax(outputs[0], dim=1)
        l_ce = ce_loss(outputs[0], lab[:1])
        l_dice = dice_loss(soft, lab[:1].unsqueeze(1))
        l_aux = aux_loss(outputs[2], lab[:1])
        l_pse = pse_loss(outputs[3], outputs[1])
        l_con = L.softmax_mse_loss(outputs[3], outputs[4])
        loss = l_dice + l_ce + l_aux + l_pse + 10 * l_con
        lrs.append(optimizer.param_groups[0]["lr"])
        optimizer.zero_grad()
        loss.backward()
        optimizer.step()
        lr_ = base_lr * (1.0 - iter_num / max_iterations) ** 0.9
        for g in optimizer.param_groups:
            g["lr"] = lr_
        iter_num += 1
        losses.append([float(l_dice), float(l_ce), float(l_aux), float(l_pse), float(l_con), float(loss)])
        t = s + 1
        d[f"post_step{t}_norms"] = np.array([float(p.detach().double().pow(2).sum().sqrt()) for p in named.values()])
        for k in small:
            v = npy(named[k])
            d[f"post_step{t}.{k}"] = (v[::8, ::8] if v.ndim == 2 and v.size > 4096 else v).copy()      # (the 256 x 256 fc_q: every 8th row / column;
            #                                     COPIES: npy() is a view of the live parameter, which the later steps keep changing)
        d[f"momentum_step{t}.final.weight"] = npy(optimizer.state[named["final.weight"]]["momentum_buffer"]).copy()
        d[f"delta_step{t}.{big}_sub"] = (named[big].detach()[::432, ::432].double() - w0_big).numpy()
        d[f"momentum_step{t}.{big}_sub"] = npy(optimizer.state[named[big]]["momentum_buffer"])[::432, ::432].copy()
        print(f"steps10: step {t} lr {lrs[-1]:.6f} losses {losses[-1]}", flush=True)
    d["losses"] = np.array(losses)
    d["lr_used"] = np.array(lrs)
    np.savez_compressed(os.path.join(out, f"model_unet3d_icl_nc{nc}_steps10.npz"), **d)


# ---------------------------------------------------------------- 2-D U-Net ICL (BASELINE config 1)
def gen_model2d(R, out, nc=4):
    from networks.unet import UNet
    from networks.unet_icl import UNet_icl
    L = R["losses"]
    d = {}
    plain = UNet(1, nc)
    d["plain_keys"] = np.array(list(plain.state_dict().keys()))
    model = UNet_icl(1, nc)
    parity_mode(model)
    fill(model)
    d["keys"] = np.array(list(model.state_dict().keys()))
    d["param_keys"] = np.array([k for k, _ in model.named_parameters()])
    img = synthetic_volume((4, 1, 256, 256), 2024)
    lab = synthetic_labels((2, 256, 256), 2025, nc)
    model.train()
    outs = model(img[:2], img[2:])
    for name, t in (("out_lab", outs[0]), ("out_unlab", outs[1])):
        d[name + "_sub"] = npy(t)[:, :, ::8, ::8]
        d[name + "_l2"] = npy(t.double().pow(2).sum(dim=(0, 2, 3)).sqrt())
    for name, lst in (("maps_lab", outs[2]), ("maps_unlab", outs[3]), ("maps_con", outs[4])):
        for i, t in enumerate(lst):
            st = (1, 2, 4)[i]
            d[f"{name}{i}_sub"] = npy(t)[:, :, ::st, ::st]
            d[f"{name}{i}_l2"] = npy(t.double().pow(2).sum().sqrt())
    l_ce = nn.CrossEntropyLoss()(outs[0], lab.long())
    l_dice = L.DiceLoss(nc)(outs[0], lab.unsqueeze(1), softmax=True)
    l_aux = L.AuxLoss(nc, resize=[256, 256])(outs[2], lab)
    l_pse = L.PseudoSoftLoss(nc, resize=[256, 256])(outs[3], outs[1])
    l_con = L.softmax_mse_loss(outs[3], outs[4])
    loss = l_ce + l_dice + l_aux + l_pse + 50 * l_con
    d["losses"] = np.array([float(l_ce.detach()), float(l_dice.detach()), float(l_aux.detach()), float(l_pse.detach()),
                            float(l_con.detach()), float(loss.detach())])
    opt = torch.optim.SGD(model.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4)
    opt.zero_grad()
    loss.backward()
    gn, none = {}, []
    for k, p in model.named_parameters():
        if p.grad is None:
            none.append(k)
        else:
            gn[k] = float(p.grad.double().pow(2).sum().sqrt())
    d["grad_none"] = np.array(none)
    d["grad_norm_keys"] = np.array(list(gn.keys()))
    d["grad_norms"] = np.array(list(gn.values()))
    sd_ = dict(model.named_parameters())
    for k in ("decoder.out_conv.weight", "encoder.in_conv.conv_conv.0.weight", "encoder.in_conv.conv_conv.1.weight",
              "decoder.up1.conv1x1.weight", "sspa.guided_Q"):
        d["grad." + k] = npy(sd_[k].grad)
    opt.step()
    d["post_sgd_norms"] = np.array([float(p.detach().double().pow(2).sum().sqrt()) for _, p in model.named_parameters()])
    bufs = dict(model.named_buffers())
    for k in ("encoder.in_conv.conv_conv.1.running_mean", "encoder.in_conv.conv_conv.1.running_var",
              "decoder.up4.conv.conv_conv.5.running_var", "sspa.attn_convs0.2.block.bn_depth.running_var"):
        d["buf." + k] = npy(bufs[k])
    # inference branch in eval mode (uses the running statistics just updated by the train-mode pass)
    model.eval()
    with torch.no_grad():
        y = model(img[:2], inference=True)
    d["inf_logits_sub"] = npy(y)[:, :, ::8, ::8]
    np.savez_compressed(os.path.join(out, f"model_unet2d_icl_nc{nc}.npz"), **d)
    print("model2d: losses", d["losses"], "grad_none", len(none))


# ---------------------------------------------------------------- SwinUNETR-ICL (BASELINE config 4)
def gen_swin(R, out, nc=2, side=96):
    """side = 96: the BASELINE configs[3] shape (GPU parity tests); side = 64: the same model class on 64^3 volumes (aligner
    grids 4^3 / 8^3 / 16^3, 141 M parameters) — the vector the CPU oracle is pinned against in the default test run."""
    install_monai_block_stubs()
    from networks.swinunetr_icl import SwinUNETR_icl
    L = R["losses"]
    model = SwinUNETR_icl(img_size=(side, side, side), in_channels=1, out_channels=nc, feature_size=48, drop_rate=0.0,
                          attn_drop_rate=0.0, dropout_path_rate=0.0, use_checkpoint=False)   # net_factory_3d.py:54-63
    parity_mode(model)
    fill(model)
    d = {}
    d["keys"] = np.array(list(model.state_dict().keys()))
    d["param_keys"] = np.array([k for k, _ in model.named_parameters()])
    d["param_shapes"] = np.array([",".join(map(str, p.shape)) for _, p in model.named_parameters()])
    vol = synthetic_volume((2, 1, side, side, side), 1337)
    lab = synthetic_labels((1, side, side, side), 4242, nc)
    model.train()
    with torch.no_grad():
        hs = model.swinViT(vol[:1], True)
    for i, h in enumerate(hs):
        d[f"hidden{i}_l2"] = npy(h.double().pow(2).sum().sqrt())
        d[f"hidden{i}_sub"] = npy(h)[:, ::max(1, h.shape[1] // 8), ::max(1, h.shape[2] // 6), ::max(1, h.shape[3] // 6), ::max(1, h.shape[4] // 6)]
    outs = model(vol[:1], vol[1:])
    for name, t in (("final_lab", outs[0]), ("final_unlab", outs[1])):
        d[name + "_sub"] = npy(t)[:, :, ::8, ::8, ::8]
        d[name + "_l2"] = npy(t.double().pow(2).sum(dim=(0, 2, 3, 4)).sqrt())
    for name, lst in (("maps_lab", outs[2]), ("maps_unlab", outs[3]), ("maps_con", outs[4])):
        for i, t in enumerate(lst):
            d[f"{name}{i}"] = npy(t)
    soft = torch.softmax(outs[0], 1)
    l_ce = nn.CrossEntropyLoss()(outs[0], lab)
    l_dice = L.DiceLoss(nc)(soft, lab.unsqueeze(1))
    if side == 96:
        l_aux = L.AuxLoss3D(nc)(outs[2], lab)
        l_pse = L.PseudoSoftLoss3D(nc)(outs[3], outs[1])
    else:   # AuxLoss3D / PseudoSoftLoss3D resize to a hard-coded 96^3 (utils/losses.py:263,292): plain quadratics drive the
        #     same gradient paths through the aligners here; the losses themselves are pinned by the unit and 96^3 vectors
        l_aux = sum(t.pow(2).mean() for t in outs[2])
        l_pse = sum(t.pow(2).mean() for t in outs[3])
    l_con = L.softmax_mse_loss(outs[3], outs[4])
    loss = l_dice + l_ce + l_aux + l_pse + 10 * l_con     # train_inherent_consistent_swinunetr_3D_BraTS.py
    d["losses"] = np.array([float(v.detach()) for v in (l_dice, l_ce, l_aux, l_pse, l_con, loss)])
    opt = torch.optim.SGD(model.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4)
    opt.zero_grad()
    loss.backward()
    gn, none = {}, []
    for k, p in model.named_parameters():
        if p.grad is None:
            none.append(k)
        else:
            gn[k] = float(p.grad.double().pow(2).sum().sqrt())
    d["grad_none"] = np.array(none)
    d["grad_norm_keys"] = np.array(list(gn.keys()))
    d["grad_norms"] = np.array(list(gn.values()))
    sd_ = dict(model.named_parameters())
    for k in ("out.conv.conv.weight", "swinViT.patch_embed.proj.weight", "swinViT.layers1.0.blocks.1.attn.relative_position_bias_table",
              "swinViT.layers4.0.blocks.0.attn.qkv.bias", "decoder5.transp_conv.conv.weight"):
        g = npy(sd_[k].grad)
        d["grad." + k] = g if g.size <= 8192 else g.reshape(-1)[::97].copy()
    opt.step()
    d["post_sgd_norms"] = np.array([float(p.detach().double().pow(2).sum().sqrt()) for _, p in model.named_parameters()])
    model.eval()
    with torch.no_grad():
        y = model(vol[:1], inference=True)
    d["inf_logits_sub"] = npy(y)[:, :, ::8, ::8, ::8]
    np.savez_compressed(os.path.join(out, f"model_swinunetr_icl_nc{nc}.npz" if side == 96 else f"model_swinunetr_icl_{side}_nc{nc}.npz"), **d)
    print("swin: losses", d["losses"], "grad_none", len(none), "params", sum(p.numel() for p in model.parameters()))


# ---------------------------------------------------------------- 2D Swin-UNet ICL
def install_timm_turtle_stubs():
    """networks/vision_transformer.py imports `timm`, `timm.models.layers.{DropPath, to_2tuple, trunc_normal_}` and, by accident,
    `from turtle import back` (needs tkinter).  None of them carries arithmetic of the path: DropPath is the per-sample
    stochastic depth (forced to identity in parity mode), to_2tuple a tuple helper, trunc_normal_ an initialiser that the
    hash fill overwrites."""
    class DropPath(nn.Module):
        def __init__(self, drop_prob=0.0):
            super().__init__()
            self.drop_prob = drop_prob

        def forward(self, x):
            if self.drop_prob == 0.0 or not self.training:
                return x
            keep = 1.0 - self.drop_prob
            mask = x.new_empty((x.shape[0],) + (1,) * (x.ndim - 1)).bernoulli_(keep)
            return x.div(keep) * mask

    timm = types.ModuleType("timm")
    models = types.ModuleType("timm.models")
    layers = types.ModuleType("timm.models.layers")
    layers.DropPath = DropPath
    layers.to_2tuple = lambda v: tuple(v) if isinstance(v, (tuple, list)) else (v, v)
    layers.trunc_normal_ = lambda t, std=1.0, **kw: nn.init.trunc_normal_(t, std=std)
    timm.models, models.layers = models, layers
    sys.modules.update({"timm": timm, "timm.models": models, "timm.models.layers": layers})
    turtle = types.ModuleType("turtle")
    turtle.back = None
    sys.modules["turtle"] = turtle


def swinunet_config():
    """configs/swin_tiny_patch4_window7_224_lite.yaml over the defaults of networks/config.py (yacs is not installed)."""
    NS = types.SimpleNamespace
    return NS(DATA=NS(IMG_SIZE=224),
              MODEL=NS(DROP_RATE=0.0, DROP_PATH_RATE=0.2, PRETRAIN_CKPT=None,
                       SWIN=NS(PATCH_SIZE=4, IN_CHANS=3, EMBED_DIM=96, DEPTHS=[2, 2, 2, 2], NUM_HEADS=[3, 6, 12, 24], WINDOW_SIZE=7,
                               MLP_RATIO=4.0, QKV_BIAS=True, QK_SCALE=False, APE=False, PATCH_NORM=True)),
              TRAIN=NS(USE_CHECKPOINT=False))


def gen_swinunet2d(R, out, nc=4):
    install_timm_turtle_stubs()
    from networks.vision_transformer import SwinUnet
    L = R["losses"]
    model = SwinUnet(swinunet_config(), img_size=224, num_classes=nc)      # net_factory.py:85-86
    parity_mode(model)
    fill(model)
    d = {}
    d["keys"] = np.array(list(model.state_dict().keys()))
    d["param_keys"] = np.array([k for k, _ in model.named_parameters()])
    d["param_shapes"] = np.array([",".join(map(str, p.shape)) for _, p in model.named_parameters()])
    img = synthetic_volume((4, 1, 224, 224), 3024)
    lab = synthetic_labels((2, 224, 224), 3025, nc)
    model.train()
    with torch.no_grad():
        x3 = img[:2].repeat(1, 3, 1, 1)
        xe, skips = model.swin_unet.forward_features(x3)
        d["enc_out"] = npy(xe)[:, ::7, ::32]
        for i, t in enumerate(skips):
            d[f"skip{i}_l2"] = npy(t.double().pow(2).sum().sqrt())
        xl, feats = model.swin_unet.forward_up_features(xe, skips)
        for i, t in enumerate(feats):
            d[f"feat{i}_sub"] = npy(t)[:, ::13, ::16]
    outs = model(img[:2], img[2:])
    for name, t in (("out_lab", outs[0]), ("out_unlab", outs[1])):
        d[name + "_sub"] = npy(t)[:, :, ::8, ::8]
        d[name + "_l2"] = npy(t.double().pow(2).sum(dim=(0, 2, 3)).sqrt())
    for name, lst in (("maps_lab", outs[2]), ("maps_unlab", outs[3]), ("maps_con", outs[4])):
        for i, t in enumerate(lst):
            st = (1, 2, 4)[i]
            d[f"{name}{i}_sub"] = npy(t)[:, :, ::st, ::st]
            d[f"{name}{i}_l2"] = npy(t.double().pow(2).sum().sqrt())
    l_ce = nn.CrossEntropyLoss()(outs[0], lab.long())
    l_dice = L.DiceLoss(nc)(outs[0], lab.unsqueeze(1), softmax=True)
    l_aux = L.AuxLoss(nc)(outs[2], lab)                       # resize default [224, 224]
    l_pse = L.PseudoSoftLoss(nc)(outs[3], outs[1])
    l_con = L.softmax_mse_loss(outs[3], outs[4])
    loss = l_ce + l_dice + l_aux + l_pse + 50 * l_con         # train_inherent_consistent_swinunet_2D.py:148-155
    d["losses"] = np.array([float(v.detach()) for v in (l_ce, l_dice, l_aux, l_pse, l_con, loss)])
    opt = torch.optim.SGD(model.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4)
    opt.zero_grad()
    loss.backward()
    gn, none = {}, []
    for k, p in model.named_parameters():
        if p.grad is None:
            none.append(k)
        else:
            gn[k] = float(p.grad.double().pow(2).sum().sqrt())
    d["grad_none"] = np.array(none)
    d["grad_norm_keys"] = np.array(list(gn.keys()))
    d["grad_norms"] = np.array(list(gn.values()))
    sd_ = dict(model.named_parameters())
    for k in ("swin_unet.output.weight", "swin_unet.patch_embed.proj.weight", "swin_unet.layers.0.blocks.1.attn.relative_position_bias_table",
              "swin_unet.layers_up.0.expand.weight", "swin_unet.concat_back_dim.1.weight", "sspa.guided_Q"):
        g = npy(sd_[k].grad)
        d["grad." + k] = g if g.size <= 8192 else g.reshape(-1)[::97].copy()
    opt.step()
    d["post_sgd_norms"] = np.array([float(p.detach().double().pow(2).sum().sqrt()) for _, p in model.named_parameters()])
    model.eval()
    with torch.no_grad():
        y = model(img[:2], inference=True)
    d["inf_logits_sub"] = npy(y)[:, :, ::8, ::8]
    np.savez_compressed(os.path.join(out, f"model_swinunet2d_icl_nc{nc}.npz"), **d)
    print("swinunet2d: losses", d["losses"], "grad_none", len(none), "params", sum(p.numel() for p in model.parameters()))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="all")
    a = ap.parse_args()
    torch.set_num_threads(os.cpu_count())
    R = import_reference()
    if a.only in ("all", "unit"):
        gen_unit(R, HERE)
    if a.only in ("all", "model2"):
        gen_model(R, HERE, 2)
    if a.only in ("all", "model16"):
        gen_model(R, HERE, 16)
    if a.only in ("all", "wgrads"):
        gen_model_dense_wgrads(R, HERE)
    if a.only in ("all", "steps"):
        gen_model_steps(R, HERE)
    if a.only in ("all", "steps10"):
        gen_model_steps10(R, HERE)
    if a.only in ("all", "model2d"):
        gen_model2d(R, HERE)
    if a.only in ("all", "swin"):
        gen_swin(R, HERE)
    if a.only in ("all", "swin64"):
        gen_swin(R, HERE, side=64)
    if a.only in ("all", "swinunet2d"):
        gen_swinunet2d(R, HERE)
