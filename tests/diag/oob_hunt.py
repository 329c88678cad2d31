"""Hunt for a kernel that writes in front of its own output buffer: guard bytes behind every LayerNorm-backward workspace of the level-1
aligner layers, checked (after a device synchronisation) behind every autograd backward node of icl_amd.ops."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from icl_amd import ops
from icl_amd.networks.unet_3D_icl import unet_3D_icl
from icl_amd.trainer import ICLConfig, ICLTrainer
from icl_amd.utils.hashfill import synthetic_labels, synthetic_volume
from test_gpu_parity import fill_like_reference_init, _parity_mode
dev = torch.device("cuda", 0)
vol = synthetic_volume((2, 1, 96, 96, 96), 1337).to(dev)
lab = synthetic_labels((1, 96, 96, 96), 4242, 2).to(dev)
GUARD = 1024          # floats
guards = []
orig_ws = ops._ws
in_ln = [False]

def guarded_ws(nbytes, like):
    if not in_ln[0]:
        return orig_ws(nbytes, like)
    n = (max(int(nbytes), 4) + 3) // 4
    t = torch.empty(GUARD + n + GUARD, dtype=torch.float32, device=like.device)
    t[:GUARD] = 12345.0
    t[GUARD + n:] = 12345.0
    guards.append((t, n))
    return t[GUARD:GUARD + n]

ops._ws = guarded_ws
orig_ln = ops._LayerNorm.backward

def ln_bwd(ctx, gy):
    in_ln[0] = gy.shape[-1] == 128
    try:
        return orig_ln(ctx, gy)
    finally:
        in_ln[0] = False

found = [False]

def check(name):
    if found[0]:
        return
    torch.cuda.synchronize()
    for t, n in guards:
        head, tail = t[:GUARD], t[GUARD + n:]
        bh, bt = (head != 12345.0).nonzero().flatten().tolist(), (tail != 12345.0).nonzero().flatten().tolist()
        if bh or bt:
            found[0] = True
            print(f"GUARD DAMAGED after backward of {name}: workspace of {n} floats; head guard elements {bh[:20]} ({len(bh)}), tail guard elements {bt[:20]} ({len(bt)})")
            if bt: print("   tail values", tail[bt[:8]].tolist())
            if bh: print("   head values", head[bh[:8]].tolist())
            return

def wrap(cls):
    ob = cls.backward
    def nb(ctx, *a):
        out = ob(ctx, *a)
        check(cls.__name__)
        return out
    cls.backward = staticmethod(nb)

ops._LayerNorm.backward = staticmethod(ln_bwd)
for k, v in list(vars(ops).items()):
    if isinstance(v, type) and issubclass(v, torch.autograd.Function) and v is not torch.autograd.Function:
        wrap(v)
ops.SideStream.enabled, ops.SideStream.lanes = (len(sys.argv) > 1 and sys.argv[1] == "lanes"), (3 if len(sys.argv) > 1 and sys.argv[1] == "lanes" else 0)
model = unet_3D_icl(n_classes=2, in_channels=1, device=dev)
fill_like_reference_init(list(model.named_parameters()))
_parity_mode(model)
model.train()
tr = ICLTrainer(model, ICLConfig(num_classes=2, labeled_bs=1, max_iterations=10, update_in_backward=False))
tr._forward_backward(vol, lab)
check("end of step")
print("guards:", len(guards), "damaged:", found[0])
