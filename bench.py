#!/usr/bin/env python3
"""bench.py — ICL training-step throughput on MI355X (BASELINE.json metric: 3D volumes/sec/node).

    python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run, one rank per GPU)

A "step" is one full ICL iteration of the reference trainer (train_inherent_consistent_unet_3D_BraTS.py:99-121)
on a synthetic BraTS-shaped batch resident in HBM: per GPU 1 labeled + 1 unlabeled 96^3 volume (BASELINE.json
configs[1]: "3D U-Net ICL 96^3, num_classes=2, batch=2, 1xMI355X"); forward of both streams + 3 aligner calls,
5-term loss, backward, gradient all-reduce over RCCL when N>1, SGD(momentum, wd).  Dropout p=0.3 and
DropPath 0.02 are active as in training.  value = volumes processed by all ranks / max-over-ranks time.

Extra objects on the JSON line (tier contract ④):
  roofline      dominant kernel = conv3d_mfma_fwd_kernel (forward + input-gradient 3x3x3/1x1x1 convolutions):
                algorithmic FLOPs of its launches / their HIP-event durations, against the 157.3 TFLOP/s
                fp32 MFMA peak (MI355X_MICROARCH.md); `hbm_frac` = algorithmic conv bytes over the 8 TB/s HBM peak.
  cpu_baseline  the CPU oracle (oracle/icl_oracle.py, torch-CPU restatement of the reference) timed on rank 0
                at N=1 on the same workload for one step after one warm-up step.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md "Peak FP32 (matrix)"
PEAK_HBM_GBS = 8000.0          # MI355X_MICROARCH.md "HBM3E peak BW" (spec)


def cpu_baseline(num_classes: int, max_seconds: float = 120.0):
    """Time the oracle (CPU port of the reference) on the bench workload: 1 warm-up + 1 timed step."""
    from icl_amd.utils.hashfill import synthetic_labels, synthetic_volume
    from oracle import icl_oracle as O
    cores = os.cpu_count() or 1
    torch.set_num_threads(cores)
    p = O.make_params(O.unet_3d_icl_shapes(num_classes), requires_grad=True)
    p.update(O.aligner_buffers("sspa.", O.UNET3D_HEADS))
    p.update(O.aligner_buffers("uscl.", O.UNET3D_HEADS))
    names = [k for k, _ in O.unet_3d_icl_shapes(num_classes)]
    vol = synthetic_volume((2, 1, 96, 96, 96), 1337)
    lab = synthetic_labels((1, 96, 96, 96), 4242, num_classes)
    bufs = {}
    times = []
    t_all = time.time()
    for it in range(2):
        t0 = time.time()
        outs = O.unet_3d_icl_forward(p, vol[:1], vol[1:], training=True)
        total, _ = O.icl_losses(outs, lab, num_classes)
        for k in names:
            p[k].grad = None
        total.backward()
        O.sgd_step(p, {k: p[k].grad for k in names}, bufs, lr=0.01)
        times.append(time.time() - t0)
        if time.time() - t_all > max_seconds:
            break
    t = times[-1]
    return {"value": round(2.0 / t, 4), "unit": "volumes/s", "cores": cores, "kind": "port",
            "sample": f"{len(times)} full ICL step(s) of the same workload (2 volumes 96^3, nc={num_classes}); "
                      f"last step timed: {t:.2f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--num-classes", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timer", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device; the product path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from icl_amd import ops
    from icl_amd.networks.unet_3D_icl import unet_3D_icl
    from icl_amd.trainer import ICLConfig, ICLTrainer
    from icl_amd.utils.hashfill import synthetic_labels, synthetic_volume

    ddp = None
    if world > 1:
        import torch.distributed as dist
        from icl_amd.ddp import GradientReducer
        dist.init_process_group("nccl", device_id=dev)

    torch.manual_seed(1337 + rank)
    nc = args.num_classes
    model = unet_3D_icl(n_classes=nc, in_channels=1, device=dev)
    model.train()
    if world > 1:
        ddp = GradientReducer(model, world)
        ddp.broadcast_parameters()
    cfg = ICLConfig(num_classes=nc, labeled_bs=1, base_lr=0.02 if nc == 16 else 0.01,
                    w_pse=0.1 if nc == 16 else 1.0)
    trainer = ICLTrainer(model, cfg, ddp)
    # synthetic batch, resident in HBM before the timed region: [labeled, unlabeled]
    vol = synthetic_volume((2, 1, 96, 96, 96), 1337 + rank, device=dev)
    lab = synthetic_labels((1, 96, 96, 96), 4242 + rank, nc, device=dev)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        trainer.step(vol, lab)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        trainer.step(vol, lab)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    volumes = 2 * world * args.steps
    value = volumes / dt

    roof = None
    if rank == 0 and not args.no_kernel_timer:
        with ops.KernelTimer() as kt:
            for _ in range(2):
                trainer.step(vol, lab)
        summ = kt.summary()
        if "conv3d_mfma_fwd_kernel" in summ:
            n, ms, fl, by = summ["conv3d_mfma_fwd_kernel"]
            ach = fl / (ms * 1e-3) / 1e12
            roof = {"bound": "mfma", "kernel": "conv3d_mfma_fwd_kernel", "achieved": round(ach, 3),
                    "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / PEAK_F32_MFMA_TFLOPS, 4),
                    "traffic": None, "launches": n, "avg_launch_us": round(ms * 1e3 / n, 2),
                    "algorithmic_gbs": round(by / (ms * 1e-3) / 1e9, 1),
                    "hbm_frac": round(by / (ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4)}
            if "conv3d_mfma_wgrad_kernel" in summ:
                n2, ms2, fl2, by2 = summ["conv3d_mfma_wgrad_kernel"]
                roof["wgrad"] = {"kernel": "conv3d_mfma_wgrad_kernel(+zero,+unpack,+bias-grad)", "launches": n2,
                                 "achieved": round(fl2 / (ms2 * 1e-3) / 1e12, 3),
                                 "avg_launch_us": round(ms2 * 1e3 / n2, 2)}
                tot_ms = (ms + ms2) / 2.0  # per step (2 timed steps)
                roof["conv_ms_per_step"] = round(tot_ms, 3)
                # 2.021 GB algorithmic conv bytes per volume fwd+bwd, 2 volumes per step
                roof["conv_fwd_bwd_hbm_frac"] = round((by + by2) / 2.0 / (tot_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4)

    if rank == 0:
        out = {
            "metric": "3D volumes/sec/node (fwd+bwd, 96^3 patch)", "value": round(value, 3), "unit": "volumes/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"3D U-Net ICL BraTS-shape synthetic 96x96x96, num_classes={nc}, "
                                   f"batch=2 per GPU (1 labeled + 1 unlabeled), full ICL step incl. SGD",
                       "global_batch": 2 * world, "parallelism": f"dp{world}"},
            "roofline": roof,
        }
        if world == 1 and not args.no_cpu_baseline:
            del trainer, model
            torch.cuda.empty_cache()
            out["cpu_baseline"] = cpu_baseline(nc)
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
