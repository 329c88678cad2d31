"""CPU tests of the drop-in surface around the hot path (SURVEY.md §8 rows b, f1, f2, f3): the `compat/` import root the
unchanged reference trainers run against, checkpoint warm-starts (`load_from`, the trainers' key rewriting), the host transforms
and the MONAI-style sliding-window validation."""
import os
import subprocess
import sys
import types

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(HERE, "golden"))
from make_golden_datafeed import CASES, case_volume  # noqa: E402

GOLD = np.load(os.path.join(HERE, "golden", "datafeed.npz"))


def test_compat_root_resolves_every_import_of_the_reference_trainers():
    """`PYTHONPATH=compat:<repo>`: the flat imports of train_inherent_consistent_{unet_3D_BraTS,unet_3D_AMOS22,swinunetr_3D_BraTS,
    unet_2D}.py (:17-23) resolve to icl_amd; without a HIP device the factory itself still raises (no CPU model)."""
    code = (
        "import sys; sys.argv = ['train.py', '--exp', 'x', '--max_iterations', '5', '--labeled_num', '25']\n"
        "from networks.net_factory_3d import net_factory_3d\n"
        "from networks.net_factory import net_factory\n"
        "from utils import losses, ramps\n"
        "from val_3D import test_all_case_base, test_all_case_amos\n"
        "from val_2D import test_single_volume_ours\n"
        "from dataloaders.brats2019 import (BraTS2019, RandomCrop, RandomRotFlip, ToTensor, TwoStreamBatchSampler)\n"
        "import networks.net_factory_3d as f, icl_amd.networks.net_factory_3d as g\n"
        "assert f.net_factory_3d is g.net_factory_3d and losses.DiceLoss.__module__ == 'icl_amd.utils.losses'\n"
        "assert abs(ramps.sigmoid_rampup(10, 40) - 0.060054667895) < 1e-9 and ramps.sigmoid_rampup(3, 0) == 1.0\n"
        "assert [n for n in ('DiceLoss', 'AuxLoss3D', 'PseudoSoftLoss3D', 'AuxLoss', 'PseudoSoftLoss', 'softmax_mse_loss') if not hasattr(losses, n)] == []\n"
        "try:\n    net_factory_3d('unet_3D_icl', 1, 2); raise SystemExit('expected RuntimeError without a device')\n"
        "except RuntimeError: pass\n"
        "print('compat ok')\n")
    env = dict(os.environ, PYTHONPATH=os.path.join(ROOT, "compat") + os.pathsep + ROOT, CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES="")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, cwd="/tmp", timeout=300)
    assert out.returncode == 0 and "compat ok" in out.stdout, out.stdout + out.stderr


def test_host_transforms_equal_the_reference_transforms():
    from icl_amd.dataloaders.brats2019 import CenterCrop, RandomCrop, RandomRotFlip, ToTensor
    for n, (shape, patch, seed) in enumerate(CASES):
        image, label = case_volume(shape, seed)
        np.random.seed(seed)
        s = ToTensor()(RandomCrop(patch)(RandomRotFlip()({"image": image, "label": label})))
        assert s["image"].dtype == torch.float32 and s["label"].dtype == torch.int64
        assert np.array_equal(s["image"].numpy(), GOLD[f"aug.{n}.image"]) and np.array_equal(s["label"].numpy(), GOLD[f"aug.{n}.label"])
    image, label = case_volume((9, 30, 12), 13)
    c = CenterCrop((12, 12, 8))({"image": image, "label": label})
    assert c["image"].shape == (12, 12, 8) and c["label"].shape == (12, 12, 8)
    assert c["image"][0].sum() == 0 and np.array_equal(c["image"][6 - 4], image[0, 9:21, 2:10] * 0 + c["image"][2])   # 6 rows of padding in front


def _ssl_checkpoint(model):
    """A self-supervised SwinViT checkpoint as the reference expects it (swinunetr_icl.py:260-308,871-903): keys
    `module.<encoder key>`, the block MLPs named fc1 / fc2."""
    sd = {}
    g = torch.Generator().manual_seed(5)
    for k, t in model.swinViT.state_dict().items():
        name = "module." + k.replace("mlp.linear1", "mlp.fc1").replace("mlp.linear2", "mlp.fc2")
        sd[name] = torch.randn(t.shape, generator=g) if t.is_floating_point() else t.clone()
    return {"state_dict": sd}


@pytest.mark.parametrize("icl", [False, True])
def test_swinunetr_load_from_copies_the_ssl_encoder(icl):
    from icl_amd.networks.swinunetr import SwinUNETR
    from icl_amd.networks.swinunetr_icl import SwinUNETR_icl
    cls = SwinUNETR_icl if icl else SwinUNETR
    model = cls(img_size=(96, 96, 96) if icl else (32, 32, 32), in_channels=1, out_channels=2, feature_size=12, device=torch.device("cpu"))
    before = {k: v.clone() for k, v in model.state_dict().items()}
    w = _ssl_checkpoint(model)
    model.load_from(w)
    sd = w["state_dict"]
    copied = 0
    for k, t in model.swinViT.state_dict().items():
        src = "module." + k.replace("mlp.linear1", "mlp.fc1").replace("mlp.linear2", "mlp.fc2")
        # what the reference copies: patch_embed.proj, every tensor of the two blocks of each stage, the stage's downsample
        if k.startswith("patch_embed.proj") or ".blocks." in k or ".downsample." in k:
            assert torch.equal(t, sd[src]), k
            copied += 1
    assert copied >= 2 + 4 * (2 * 13 + 3)
    for k, t in model.state_dict().items():        # nothing outside the encoder moves
        if not k.startswith("swinViT."):
            assert torch.equal(t, before[k]), k


def test_swinunetr_trainer_side_key_rewriting_loads_the_same_tensors_as_the_reference_would():
    """train_inherent_consistent_swinunetr_3D_BraTS.py:77-99: `module.` -> `swinViT.` and load_state_dict(strict=False).  The
    SSL names fc1 / fc2 do not exist in the model (linear1 / linear2), so exactly those stay untouched — same as the reference,
    whose key set the golden test pins."""
    from icl_amd.networks.swinunetr_icl import SwinUNETR_icl
    model = SwinUNETR_icl(img_size=(96, 96, 96), in_channels=1, out_channels=2, feature_size=12, device=torch.device("cpu"))
    state_dict = _ssl_checkpoint(model)["state_dict"]
    before = {k: v.clone() for k, v in model.state_dict().items()}
    if "module." in list(state_dict.keys())[0]:
        for key in list(state_dict.keys()):
            state_dict[key.replace("module.", "swinViT.")] = state_dict.pop(key)
    msg = model.load_state_dict(state_dict, strict=False)
    assert msg.unexpected_keys and all(".mlp.fc" in k for k in msg.unexpected_keys)
    assert all(not k.startswith("swinViT.") or ".mlp.linear" in k for k in msg.missing_keys)
    for k, t in model.state_dict().items():
        if k in state_dict:
            assert torch.equal(t, state_dict[k]), k
        else:
            assert torch.equal(t, before[k]), k


def test_swinunet2d_load_from_mirrors_the_imagenet_encoder_into_the_decoder(tmp_path):
    """vision_transformer.py:110-146: `{'model': ...}` checkpoints fill the encoder and, key `layers.i` -> `layers_up.(3-i)`, the
    decoder stages whose shapes agree; checkpoints without 'model' are the 17-character-prefix form."""
    from icl_amd.networks.vision_transformer import SwinUnet, default_config
    cfg = default_config()
    model = SwinUnet(cfg, img_size=224, num_classes=4, device=torch.device("cpu"))
    enc = {k: v for k, v in model.swin_unet.state_dict().items()
           if k.startswith(("patch_embed.", "layers.", "norm.")) and v.is_floating_point()}
    g = torch.Generator().manual_seed(9)
    ckpt = {k: torch.randn(v.shape, generator=g) for k, v in enc.items()}
    ckpt["head.weight"] = torch.randn(1000, 768, generator=g)          # ImageNet classifier: no such key in the model
    path = os.path.join(tmp_path, "swin_tiny.pth")
    torch.save({"model": ckpt}, path)
    cfg.MODEL.PRETRAIN_CKPT = path
    before = {k: v.clone() for k, v in model.swin_unet.state_dict().items()}
    model.load_from(cfg)
    after = model.swin_unet.state_dict()
    mirrored = 0
    for k, v in ckpt.items():
        if k in after:
            assert torch.equal(after[k], v), k
        if k.startswith("layers."):
            up = "layers_up." + str(3 - int(k[7:8])) + k[8:]
            if up in after:
                if after[up].shape == v.shape:
                    assert torch.equal(after[up], v), up
                    mirrored += 1
                else:
                    assert torch.equal(after[up], before[up]), up
    assert mirrored > 20
    # second form: a full SwinUnet checkpoint saved from a wrapped model ("module.swin_unet." = 17 characters)
    wrapped = {"module.swin_unet." + k: torch.full_like(v, 0.5) if v.is_floating_point() else v for k, v in before.items()}
    torch.save(wrapped, path)
    model.load_from(cfg)
    for k, v in model.swin_unet.state_dict().items():
        if v.is_floating_point() and "output" not in k:
            assert bool((v == 0.5).all()), k


def test_monai_style_sliding_window_grid_and_averaging():
    from icl_amd.val_3D import _scan_starts, sliding_window_inference, test_all_case_amos
    # roi 96, overlap 0.25 -> stride 72; the last window is shifted back inside
    assert _scan_starts(96, 96, 0.25) == [0] and _scan_starts(100, 96, 0.25) == [0, 4]
    assert _scan_starts(240, 96, 0.25) == [0, 72, 144] and _scan_starts(250, 96, 0.25) == [0, 72, 144, 154]
    calls = []

    def predictor(win, inference=False):           # logits = (voxel, 2 * voxel): the window average must give the volume back
        calls.append((tuple(win.shape), inference))
        return torch.cat([win, 2 * win], 1)

    x = torch.arange(1 * 1 * 20 * 9 * 13, dtype=torch.float32).view(1, 1, 20, 9, 13) / 100
    y = sliding_window_inference(x, (8, 12, 8), 4, predictor, inference=True)
    assert y.shape == (1, 2, 20, 9, 13) and torch.allclose(y[:, :1], x, atol=1e-6) and torch.allclose(y[:, 1:], 2 * x, atol=1e-5)
    assert all(s[1:] == (1, 8, 12, 8) and s[0] <= 4 and inf for s, inf in calls)      # padded axis 1 (9 -> 12), <= 4 windows per call
    n_win = len(_scan_starts(20, 8, 0.25)) * 1 * len(_scan_starts(13, 8, 0.25))
    assert sum(s[0] for s, _ in calls) == n_win

    class Net(torch.nn.Module):                    # argmax over the averaged logits, cal_metric per foreground class
        def __init__(self):
            super().__init__()
            self.p = torch.nn.Parameter(torch.zeros(1))

        def forward(self, win, inference=False):
            return torch.cat([0.5 - win, win - 0.5, win - 10], 1)

    vol = torch.zeros(1, 1, 100, 98, 97)
    vol[..., 20:60, 30:70, 10:50] = 1.0
    lab = (vol > 0.5).long()                        # the loader yields [1, 1, D, H, W]
    m = test_all_case_amos(Net(), "unet_3D_icl", [{"image": vol, "label": lab}], num_classes=3)
    assert m[0][0][0] == 1.0 and m[1][0] == (1, 0)         # class 1 perfect, class 2 empty in both (the reference's (1, 0) convention)


def test_2d_slice_validation_batches_slices_without_changing_the_procedure():
    """val_2D.test_single_volume_ours: several slices per forward == the reference's slice-by-slice loop (zoom order 0 in, arg-max,
    zoom order 0 back), and the empty-mask metric conventions."""
    from scipy.ndimage import zoom
    from icl_amd.val_2D import calculate_metric_percase, test_single_volume_ours

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.p = torch.nn.Parameter(torch.zeros(1))
            self.calls = []

        def forward(self, x, inference=False):
            self.calls.append((x.shape[0], inference))
            return torch.cat([0.3 - x, x - 0.3, (x - 0.7) * 3], 1)      # 3 classes from the intensity

    g = torch.Generator().manual_seed(3)
    image = torch.rand(1, 7, 20, 26, generator=g)
    label = (image[0] > 0.3).long() + (image[0] > 0.9).long()
    net = Net()
    got = test_single_volume_ours(image, label.unsqueeze(0), net, None, 0, classes=3, patch_size=[32, 24], slices_per_batch=4)
    assert net.calls == [(4, True), (3, True)]
    pred = np.zeros((7, 20, 26), dtype=np.int64)
    for i in range(7):                                      # the reference loop (val_2D.py:38-50), one slice per forward
        sl = zoom(image[0, i].numpy(), (32 / 20, 24 / 26), order=0)
        out = torch.argmax(torch.softmax(net(torch.from_numpy(sl)[None, None].float(), inference=True), 1), 1)[0].numpy()
        pred[i] = zoom(out, (20 / 32, 26 / 24), order=0)
    want = [calculate_metric_percase(pred == c, label.numpy() == c) for c in (1, 2)]
    assert got == want and 0 < got[0][0] <= 1
    z = np.zeros((4, 4), bool)
    o = z.copy(); o[1, 1] = True
    assert calculate_metric_percase(z, z) == (1, 0) and calculate_metric_percase(o, z) == (0, 373.128664) == calculate_metric_percase(z, o)
