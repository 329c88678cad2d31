"""Diagnostic (GPU): relative error of the FULL weight gradients (and, round 6, of dense samples of two input gradients) of the 96^3 /
48^3 convolutions of one reference step against the
reference-generated golden (tests/golden/model_unet3d_icl_nc2_wgrads.npz), on both convolution paths.  Sets the bands of
tests/test_gpu_dropin.py::test_reference_loop_body_unet3d_icl_through_compat_root."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
sys.path.insert(0, os.path.join(HERE, "..", "..", "compat"))
from conftest import rel_err  # noqa: E402
from icl_amd.utils.hashfill import fill_like_reference_init, synthetic_labels, synthetic_volume  # noqa: E402
from test_gpu_dropin import _input_gradient_errors, _keep_input_gradients, _parity_mode, _reference_loop_body  # noqa: E402

g = np.load(os.path.join(HERE, "..", "golden", "model_unet3d_icl_nc2_wgrads.npz"))
for split in ("1", "0"):
    os.environ["ICL_CONV_SPLIT"] = split
    from networks.net_factory_3d import net_factory_3d
    model = net_factory_3d(net_type="unet_3D_icl", in_chns=1, class_num=2)
    fill_like_reference_init(list(model.named_parameters()))
    _parity_mode(model)
    dev = next(model.parameters()).device
    vol = synthetic_volume((2, 1, 96, 96, 96), 1337).to(dev)
    lab = synthetic_labels((1, 96, 96, 96), 4242, 2).to(dev)
    kept, unhook = _keep_input_gradients(model)
    got, grads = _reference_loop_body(model, vol, lab, labeled_bs=1, num_classes=2, base_lr=0.01)
    unhook()
    for name, part, err, frac in _input_gradient_errors(kept, g):      # round 6: dense samples of two input gradients at 96^3
        print(f"ICL_CONV_SPLIT={split} input gradient of {name:24s} {part:8s} max-norm error / max |reference| {err:.3e}   fraction of elements beyond 1e-3: {frac:.2e}")
    for k in g.files:
        if k.startswith("grad."):
            a, b = grads[k[5:]].cpu().numpy(), g[k]
            print(f"ICL_CONV_SPLIT={split} {k[5:]:36s} rel (max-norm) {rel_err(a, b):.3e}   rms {float(np.sqrt(((a - b) ** 2).mean()) / np.sqrt((b ** 2).mean())):.3e}")
    del model, grads
    torch.cuda.empty_cache()
