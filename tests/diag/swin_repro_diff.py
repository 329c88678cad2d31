"""Which tensors differ between identical SwinUNETR-ICL runs (tests/test_gpu_parity.py::test_swinunetr_icl_steps_are_bit_reproducible)?
Usage: swin_repro_diff.py [steps]   (0: gradients of one forward/backward; n: the state after n trainer steps).  Round 5: located the
gradient a deferred weight-gradient lane left uninitialised (profiles/r5_wgrad_defer_ab.txt)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from icl_amd import ops
from icl_amd.networks.swinunetr_icl import SwinUNETR_icl
from icl_amd.trainer import ICLConfig, ICLTrainer
from icl_amd.utils.hashfill import synthetic_labels, synthetic_volume
from test_gpu_parity import fill_like_reference_init
dev = torch.device("cuda", 0)
vol = synthetic_volume((2, 1, 96, 96, 96), 93).to(dev)
lab = synthetic_labels((1, 96, 96, 96), 94, 2).to(dev)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1
runs = []
for _ in range(4):
    ops.StepRNG.tensor = None
    torch.manual_seed(20241003)
    model = SwinUNETR_icl(img_size=(96, 96, 96), in_channels=1, out_channels=2, feature_size=48, device=dev)
    fill_like_reference_init(list(model.named_parameters()))
    model.train()
    tr = ICLTrainer(model, ICLConfig(num_classes=2, labeled_bs=1, max_iterations=10, update_in_backward=False))
    if steps == 0:
        tr._forward_backward(vol, lab)
        torch.cuda.synchronize()
        state = {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}
    else:
        for _ in range(steps):
            tr.step(vol, lab)
        state = {k: v.detach().clone() for k, v in model.state_dict().items()}
    runs.append(state)
    del tr, model
    torch.cuda.empty_cache()
for i in range(1, 4):
    diff = []
    for k in runs[0]:
        if not torch.equal(runs[0][k], runs[i][k]):
            d = (runs[0][k].double() - runs[i][k].double()).abs()
            idx = (d > 0).nonzero()
            diff.append(f"{k} {tuple(runs[0][k].shape)}: {int((d > 0).sum())} elements, max {float(d.max()):.3e} of {float(runs[0][k].abs().max()):.3e}, first at {idx[0].tolist()} last at {idx[-1].tolist()}")
    print(f"SWINDIFF run {i} vs 0: {len(diff)} tensors differ")
    for l in diff[:12]:
        print("   ", l)
