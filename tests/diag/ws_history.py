"""Who owned the memory of the LayerNorm-backward workspace (1,728 x 128 case) just before it: caching-allocator history of one step with lanes."""
import os, sys, torch, traceback
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
ops.SideStream.enabled, ops.SideStream.lanes = True, 3
model = unet_3D_icl(n_classes=2, in_channels=1, device=dev)
fill_like_reference_init(list(model.named_parameters()))
_parity_mode(model)
model.train()
tr = ICLTrainer(model, ICLConfig(num_classes=2, labeled_bs=1, max_iterations=10, update_in_backward=False))
tr._forward_backward(vol, lab)          # warm the allocator like the test's earlier runs do
torch.cuda.synchronize()
for p in model.parameters():
    p.grad = None
marks = []
orig = ops._LayerNorm.backward

def patched(ctx, gy):
    out = orig(ctx, gy)
    if gy.shape[-1] == 128 and gy.numel() // 128 == 1728:
        p = ops.DeferredBiasGrads.pending
        part_b = p[-1][1]
        marks.append((part_b.data_ptr(), part_b.numel() * 4, str(torch.cuda.current_stream(gy.device))))
    return out

ops._LayerNorm.backward = staticmethod(patched)
torch.cuda.memory._record_memory_history(enabled="all", context="all", stacks="python")
tr._forward_backward(vol, lab)
torch.cuda.synchronize()
snap = torch.cuda.memory._snapshot()
torch.cuda.memory._record_memory_history(enabled=None)
traces = snap["device_traces"][0]
print("events", len(traces), "marks", marks)
for (addr, nbytes, st) in marks:
    lo, hi = addr, addr + nbytes
    # the allocation that contains the partials, and every earlier event that overlaps it
    hist = [e for e in traces if e.get("addr") is not None and e["addr"] < hi and e["addr"] + e.get("size", 0) > lo]
    print(f"--- partials of dbeta at {addr:#x} (+{nbytes}) on {st[-18:]}: {len(hist)} overlapping events")
    for e in hist[-8:]:
        fr = [f"{os.path.basename(f['filename'])}:{f['line']}:{f['name']}" for f in e.get("frames", []) if "icl_amd" in f["filename"] or "tests" in f["filename"]][:6]
        print(f"   {e['action']:16s} addr {e['addr']:#x} size {e['size']} stream {e.get('stream')}  {fr}")
