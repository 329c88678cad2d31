"""Block-by-block comparison of the HIP SwinUNETR against the CPU oracle at 96^3 (debugging aid, run on the GPU box)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from icl_amd.networks import swinunetr as SW  # noqa: E402
from icl_amd.utils.hashfill import synthetic_volume  # noqa: E402
from oracle import swin_oracle as S  # noqa: E402


def rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


dev = torch.device("cuda")
p = S.make_swin_params(2, icl=False)
m = SW.SwinUNETR((96, 96, 96), 1, 2, feature_size=48, device=dev)
with torch.no_grad():
    for k, t in m.state_dict().items():
        t.copy_(p[k])
x = synthetic_volume((1, 1, 96, 96, 96), 1337)
g = lambda t: t.to(dev)
with torch.no_grad():
    t0 = time.time()
    hw = S.swin_vit(p, x)
    print("oracle swin_vit", time.time() - t0, flush=True)
    blocks = [("encoder1", x), ("encoder2", hw[0]), ("encoder3", hw[1]), ("encoder4", hw[2]), ("encoder10", hw[4])]
    outs = {}
    for name, inp in blocks:
        want = S.unet_res_block(p, name + ".layer", inp)
        got = getattr(m, name)(g(inp))
        outs[name] = want
        print(name, tuple(inp.shape), rel(got, want), flush=True)
        lay = getattr(m, name).layer
        c1 = lay.conv1(g(inp))
        c1w = torch.nn.functional.conv3d(inp, p[name + ".layer.conv1.conv.weight"], padding=1)
        print("   conv1", rel(c1, c1w), flush=True)
    cur = outs["encoder10"]
    skips = [hw[3], outs["encoder4"], outs["encoder3"], outs["encoder2"], outs["encoder1"]]
    for name, skip in zip(("decoder5", "decoder4", "decoder3", "decoder2", "decoder1"), skips):
        want = S.unetr_up_block(p, name, cur, skip)
        got = getattr(m, name)(g(cur), g(skip))
        up = getattr(m, name).transp_conv(g(cur))
        upw = torch.nn.functional.conv_transpose3d(cur, p[name + ".transp_conv.conv.weight"], stride=2)
        print(name, tuple(cur.shape), rel(got, want), "transp", rel(up, upw), flush=True)
        cur = want
    hs = m.swinViT(g(x), True)
    for i, (a, b) in enumerate(zip(hs, hw)):
        print("hidden", i, rel(a, b), flush=True)
    lw = torch.nn.functional.conv3d(cur, p["out.conv.conv.weight"], p["out.conv.conv.bias"])
    print("out block", rel(m.out(g(cur)), lw), flush=True)
    full, feats = m.run_backbone(g(x))
    print("full logits", rel(full, lw), flush=True)
    # stage-by-stage inside the transformer
    x0 = m.swinViT.patch_embed(g(x))
    x0w = torch.nn.functional.conv3d(x, p["swinViT.patch_embed.proj.weight"], p["swinViT.patch_embed.proj.bias"], stride=2)
    print("patch_embed", rel(x0.permute(0, 4, 1, 2, 3), x0w), flush=True)
    curw = x0w
    for i, heads in enumerate(S.SWIN_HEADS):
        nxt = S.basic_layer(p, f"swinViT.layers{i + 1}.0", curw.contiguous(), heads)
        got = getattr(m.swinViT, f"layers{i + 1}")[0](g(curw.permute(0, 2, 3, 4, 1).contiguous()))
        print("stage", i + 1, rel(got.permute(0, 4, 1, 2, 3), nxt), float(nxt.abs().max()), flush=True)
        curw = nxt
