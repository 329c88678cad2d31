#!/usr/bin/env python3
"""bench.py — ICL training-step throughput on MI355X (BASELINE.json metric: 3D volumes/sec/node).

    python bench.py --gpus N --steps K --warmup W

N > 1 without WORLD_SIZE in the environment: bench.py starts `python -m torch.distributed.run --nnodes=1 --nproc-per-node N
--master-addr 127.0.0.1 --master-port P bench.py ...` itself, as a CHILD process and before anything touches the GPU (no exec),
relays rank 0's JSON line and exits with the child's return code.  Launched by torch.distributed.run (the driver's N > 1 form),
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* come from the environment and nothing is spawned.

A "step" is one full ICL iteration of the reference trainer (train_inherent_consistent_unet_3D_BraTS.py:99-121)
on a synthetic BraTS-shaped batch resident in HBM: per GPU 1 labeled + 1 unlabeled 96^3 volume (BASELINE.json
configs[1]: "3D U-Net ICL 96^3, num_classes=2, batch=2, 1xMI355X"); forward of both streams + 3 aligner calls,
5-term loss, backward, gradient all-reduce over RCCL when N>1, SGD(momentum, wd).  Dropout p=0.3 and
DropPath 0.02 are active as in training.  value = volumes processed by all ranks / max-over-ranks time.

Extra objects on the JSON line (tier contract ④):
  roofline      dominant kernel = the forward/input-gradient convolution instantiation with the largest total time
                (named exactly as rocprofv3 --stats lists it; the C ABI reports which template its launcher picked):
                algorithmic FLOPs of its launches / their HIP-event durations (events on the launch stream), against the
                kernel's matrix peak (MI355X_MICROARCH.md): dense bf16 peak / 6 terms = 416.7 TFLOP/s for the split-product
                kernels (`peak_basis`), 157.3 TFLOP/s for the exact-fp32 MFMA kernels.  `all_conv` aggregates every conv launch of a
                step (fwd + dgrad + wgrad) and also gives the algorithmic-bytes fraction of the 8 TB/s HBM peak;
                `per_kernel` lists each instantiation for cross-checking against profiles/.
  cpu_baseline  the CPU oracle (oracle/icl_oracle.py, torch-CPU restatement of the reference) timed on rank 0
                at N=1: one full step of the same workload on <=16 host threads.
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
PEAK_BF16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md dense bf16 matrix peak
# csrc/kernels/conv_bf16x3.h multiplies fp32 operands as exact three-way bf16 splits: SIX bf16 MFMA terms per fp32 product.  Its
# speed of light in ALGORITHMIC (fp32) FLOP/s is therefore the bf16 peak / 6.
SPLIT_TERMS = 6
# matrix-pipe busy fraction of the kernel's cycles, PMC (profiles/r3_pmc_conv.md; <2, 8> / <3, 8>: profiles/r2_pmc_conv.md), static
# profiles/r5_pmc_conv.md: 0.2191 M kernel cycles for a 127 us isolated launch (16->16 @96^3); round 6: 0.2001 M cycles (profiles/r6_pk_add_ab.txt)
# for the 122-124 us best-of-rounds launch
KERNEL_CLOCK_GHZ = {"conv3d_bf16x3_fwd_ws_kernel<1>": 1.63}
PIPE_BUSY = {"conv3d_mfma_fwd_static_kernel": 0.81, "conv3d_bf16x3_fwd_kernel<1, 8, 8>": 0.50, "conv3d_bf16x3_fwd_kernel<2, 8, 8>": 0.52,
             "conv3d_bf16x3_fwd_kernel<3, 8, 8>": 0.62, "conv3d_wgrad_tr_kernel<1>": 0.44, "conv3d_wgrad_tr_kernel<2>": 0.49,
             "conv3d_bf16x3_fwd_kernel<1, 8, 24>": 0.52,
             # variant 60 (profiles/r3_pmc_conv.md, second table)
             "conv3d_bf16x3_fwd_kernel<1, 8, 60>": 0.56, "conv3d_bf16x3_fwd_kernel<2, 8, 60>": 0.55, "conv3d_bf16x3_fwd_kernel<3, 8, 60>": 0.66,
             # the loader-wave kernel for one cout block: 0.66 in rounds 4-5; round 6 (profiles/r6_pk_add_ab.txt): compiled with the SLP
             # vectoriser off (plain v_sub_f32 residuals beside the consumers' MFMAs) 0.725
             "conv3d_bf16x3_fwd_ws_kernel<1>": 0.725,
             # round 5 (profiles/r5_pmc_conv.md, second table): the weight gradient walking z-columns
             "conv3d_wgrad_zs_kernel<1, 10>": 0.52, "conv3d_wgrad_zs_kernel<2, 8>": 0.50,
             # (third table) three cout blocks — in the U-Net the exchanged-roles weight gradient of 48 -> 16
             "conv3d_wgrad_zs_kernel<3, 8>": 0.68}


def cpu_baseline(num_classes: int, model: str = "unet_3D_icl"):
    """Time the oracle (CPU port of the reference) on the bench workload: thread-pool warm-up on a 32^3 backbone,
    then full ICL steps (2 volumes each) until 10 s have passed.  Threads are capped at 16: torch-CPU oversubscribes badly on the 256-core
    GPU hosts (a 256-thread step took 222 s; 8 threads take ~10 s in the build container)."""
    from icl_amd.utils.hashfill import synthetic_labels, synthetic_volume
    from oracle import icl_oracle as O
    threads = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(threads)
    if model == "swinunetr_icl":
        from oracle import swin_oracle as S
        p = S.make_swin_params(num_classes, requires_grad=True)
        names = [k for k, _ in S.swinunetr_icl_shapes(num_classes)]
        forward = S.swinunetr_icl_forward
    else:
        p = O.make_params(O.unet_3d_icl_shapes(num_classes), requires_grad=True)
        p.update(O.aligner_buffers("sspa.", O.UNET3D_HEADS))
        p.update(O.aligner_buffers("uscl.", O.UNET3D_HEADS))
        names = [k for k, _ in O.unet_3d_icl_shapes(num_classes)]
        forward = O.unet_3d_icl_forward
    vol = synthetic_volume((2, 1, 96, 96, 96), 1337)
    lab = synthetic_labels((1, 96, 96, 96), 4242, num_classes)
    with torch.no_grad():   # thread-pool / allocator warm-up, not timed
        torch.nn.functional.conv3d(synthetic_volume((1, 16, 32, 32, 32), 5), synthetic_volume((16, 16, 3, 3, 3), 6), padding=1)
    t0 = time.time()
    steps, mom = 0, {}
    while steps == 0 or time.time() - t0 < 10.0:      # bounded sample: full steps until >= 10 s of CPU work
        for k in names:
            p[k].grad = None
        outs = forward(p, vol[:1], vol[1:], training=True)
        total, _ = O.icl_losses(outs, lab, num_classes)
        total.backward()
        O.sgd_step(p, {k: p[k].grad for k in names}, mom, lr=0.01)
        steps += 1
    t = time.time() - t0
    return {"value": round(2.0 * steps / t, 4), "unit": "volumes/s", "cores": threads, "kind": "port",
            "sample": f"{steps} full ICL step(s) of the same workload (2 volumes 96^3 each, nc={num_classes}) after a 32^3 conv warm-up: {t:.2f} s"}


HBM_TRAFFIC_FILE = "profiles/r6_hbm_traffic.json"


def reference_loop_bench(nc: int, steps: int, warmup: int, dev, fused: bool, graph: bool = False):
    """The loop body of the UNCHANGED reference trainer (/root/reference/code/train_inherent_consistent_unet_3D_BraTS.py:63-64,85-90,
    103-133) — its statements, its names — through the `compat/` import root (`networks.net_factory_3d`, `utils.losses`), eager, with the
    six `.item()` reads of its logging line; on the same HBM-resident synthetic batch as the headline.  ``fused``: the one-line optimiser
    swap of INTEGRATION.md §2 (`icl_amd.optim.FusedSGD` instead of `optim.SGD`), everything else untouched."""
    compat = os.path.join(os.path.dirname(os.path.abspath(__file__)), "compat")
    sys.path.insert(0, compat)
    try:
        import torch.optim as optim
        from torch.nn.modules.loss import CrossEntropyLoss
        from networks.net_factory_3d import net_factory_3d
        from utils import losses
        from icl_amd.utils.hashfill import synthetic_labels, synthetic_volume
        base_lr, max_iterations, labeled_bs, num_classes = 0.01, 30000, 1, nc
        torch.manual_seed(1337)
        model = net_factory_3d(net_type="unet_3D_icl", in_chns=1, class_num=num_classes)
        model.train()
        if fused:
            from icl_amd.optim import FusedSGD
            optimizer = FusedSGD(model.parameters(), lr=base_lr, momentum=0.9, weight_decay=0.0001, graph=graph)
        else:
            optimizer = optim.SGD(model.parameters(), lr=base_lr, momentum=0.9, weight_decay=0.0001)
        ce_loss = CrossEntropyLoss()
        dice_loss = losses.DiceLoss(num_classes)
        aux_loss = losses.AuxLoss3D(num_classes)
        pse_loss = losses.PseudoSoftLoss3D(num_classes)
        volume_batch = synthetic_volume((2, 1, 96, 96, 96), 1337, device=dev)
        label_batch = synthetic_labels((2, 96, 96, 96), 4242, num_classes, device=dev)
        iter_num = 0
        logged = None

        def iteration():
            nonlocal iter_num, logged
            outputs = model(volume_batch[:labeled_bs], volume_batch[labeled_bs:])
            outputs_soft = torch.softmax(outputs[0], dim=1)
            loss_ce = ce_loss(outputs[0], label_batch[:labeled_bs])
            loss_dice = dice_loss(outputs_soft, label_batch[:labeled_bs].unsqueeze(1))
            loss_aux = aux_loss(outputs[2], label_batch[:labeled_bs])
            loss_pse = pse_loss(outputs[3], outputs[1])
            loss_aux_consis = losses.softmax_mse_loss(outputs[3], outputs[4])
            loss = loss_dice + loss_ce + loss_aux + loss_pse + 10 * loss_aux_consis
            optimizer.zero_grad()
            loss.backward()
            optimizer.step()
            lr_ = base_lr * (1.0 - iter_num / max_iterations) ** 0.9
            for param_group in optimizer.param_groups:
                param_group['lr'] = lr_
            iter_num = iter_num + 1
            logged = (iter_num, loss.item(), loss_ce.item(), loss_dice.item(), loss_aux.item(), loss_pse.item(), 10 * loss_aux_consis.item())

        for _ in range(max(warmup, 2) + (4 if graph else 0)):      # graph=True: three eager iterations, the capture, one replay
            iteration()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(steps):
            iteration()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / steps
        dense = sum(p.grad.numel() for p in model.parameters() if p.grad is not None)
        out = {"ms_per_step": round(dt * 1e3, 3), "value": round(2.0 / dt, 3), "unit": "volumes/s",
               "optimizer": ("icl_amd.optim.FusedSGD(..., graph=True) (one-line swap; forward and backward of the model replayed as hipGraphs, "
                             "losses / zero_grad / step / .item() of the loop eager)" if fused and graph else
                             "icl_amd.optim.FusedSGD (one-line swap; step scope opened from the model's forward hook)" if fused else "torch.optim.SGD"),
               **({"graphed": bool(getattr(optimizer, "_graph_state", None) is not None), "capture_error": getattr(optimizer, "_graph_failed", None)}
                  if fused and graph else {}),
               "dense_gradient_elements": int(dense), "last_logged_loss": round(float(logged[1]), 6)}
        del model, optimizer
        torch.cuda.empty_cache()
        return out
    finally:
        sys.path.remove(compat)
        for name in [m for m in list(sys.modules) if m.split(".")[0] in ("networks", "utils", "val_3D", "dataloaders")]:
            del sys.modules[name]


def other_workload(model_name: str, nc: int, steps: int, warmup: int, dev):
    """One of BASELINE.json's other single-GPU configurations (configs[3]: SwinUNETR ICL, configs[4]: nc = 16), timed like the headline —
    capture, the faster of replay / eager (3-step probe), K steps between synchronisations — on a model of its own, AFTER the headline
    region.  Driver-observed numbers for these configurations; the headline fields stay configs[1]."""
    from icl_amd.networks.unet_3D_icl import unet_3D_icl
    from icl_amd.trainer import ICLConfig, ICLTrainer
    from icl_amd.utils.hashfill import synthetic_labels, synthetic_volume
    torch.manual_seed(1337)
    if model_name == "swinunetr_icl":
        from icl_amd.networks.swinunetr_icl import SwinUNETR_icl
        model = SwinUNETR_icl(img_size=(96, 96, 96), in_channels=1, out_channels=nc, feature_size=48, device=dev)
    else:
        model = unet_3D_icl(n_classes=nc, in_channels=1, device=dev)
    model.train()
    cfg = ICLConfig(num_classes=nc, labeled_bs=1, base_lr=0.02 if nc == 16 else 0.01, w_pse=0.1 if nc == 16 else 1.0)
    tr = ICLTrainer(model, cfg, None)
    vol = synthetic_volume((2, 1, 96, 96, 96), 1337, device=dev)
    lab = synthetic_labels((1, 96, 96, 96), 4242, nc, device=dev)
    tr.capture(vol, lab, warmup=max(warmup, 2))

    def timed(k):
        tr.step(vol, lab)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(k):
            last = tr.step(vol, lab)
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / k, last
    tr.use_graph = True
    t_graph, _ = timed(3)
    tr.use_graph = False
    t_eager, _ = timed(3)
    tr.use_graph = t_graph <= t_eager
    dt, last = timed(steps)
    loss = float(last["loss"])
    finite = loss == loss and abs(loss) != float("inf") and all(bool(torch.isfinite(p).all()) for p in model.parameters())
    out = {"workload": f"{'SwinUNETR' if model_name == 'swinunetr_icl' else '3D U-Net'} ICL BraTS-shape synthetic 96x96x96, num_classes={nc}, "
                       "batch=2 (1 labeled + 1 unlabeled), full ICL step incl. SGD",
           "ms_per_step": round(dt * 1e3, 3), "value": round(2.0 / dt, 3), "unit": "volumes/s", "steps": steps,
           "launch": "hipGraph replay" if tr.use_graph else "eager",
           "launch_probe": {"graph_ms": round(t_graph * 1e3, 3), "eager_ms": round(t_eager * 1e3, 3)}, "finite_after_timed_steps": finite}
    del tr, model
    torch.cuda.empty_cache()
    return out


def other_workload_child(model_name: str, nc: int, steps: int, warmup: int):
    """``other_workload`` in a CHILD process (started, never exec'ed: this process has the GPU open): whatever happens to it — an
    exception, a crash inside the HIP runtime, a hang (10-minute limit) — the headline line of this process is still printed."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--other-workload", f"{model_name}:{nc}", "--steps", str(steps), "--warmup", str(warmup)]
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if r.returncode == 0 and lines:
            return json.loads(lines[-1])
        return {"error": f"child exited with {r.returncode}", "stderr_tail": r.stderr[-400:]}
    except subprocess.TimeoutExpired:
        return {"error": "child timed out after 600 s"}
    except Exception as e:      # noqa: BLE001
        return {"error": repr(e)}


def hbm_traffic(kernel: str):
    """HBM bytes per launch of a kernel from the committed PMC run (FETCH_SIZE and WRITE_SIZE passes of rocprofv3 on one
    reference layer of that kernel, gfx950 correction applied) — counters cannot be collected from inside this process, so the
    number is a STATIC one, measured by that run for the layer named in `shape`; the JSON line says so (`source`)."""
    try:
        with open(os.path.join(ROOT, HBM_TRAFFIC_FILE)) as f:
            ks = json.load(f)["kernels"]
        # schedule variants 24 / 60 of the split-product forward kernel move the same bytes as variant 8 (same tiles, same staging)
        k = ks.get(kernel) or ks.get(kernel.replace(", 8, 24>", ", 8, 8>").replace(", 8, 60>", ", 8, 8>"))
        if not k:
            return None
        return {"source": f"static profile ({HBM_TRAFFIC_FILE}), not measured in this run",
                **{kk: k[kk] for kk in ("shape", "hbm_bytes", "algorithmic_bytes")}}
    except (OSError, ValueError, KeyError):
        return None


def self_launch(args, argv) -> int:
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start torch.distributed.run as a child process (one rank per
    GPU, rendezvous on 127.0.0.1) BEFORE anything initialises the GPU in this process — `torch.cuda.device_count()` does not — and
    never by exec.  The child ranks inherit stdout, so rank 0's JSON line is the last line this command prints; the return code is
    the launcher's."""
    import socket
    import subprocess
    if not args.launcher_selftest and os.environ.get("ICL_BENCH_SHARE_GPU") != "1":
        have = torch.cuda.device_count()
        if have < args.gpus:
            print(f"bench.py --gpus {args.gpus}: this node shows {have} HIP device(s); one rank per GPU is the only supported "
                  f"placement (RCCL refuses two ranks on one device).  Nothing was run.", file=sys.stderr, flush=True)
            return 2
    with socket.socket() as s:      # a free rendezvous port (the driver passes its own when it launches the ranks itself)
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL across processes needs it on this driver
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    print(f"[bench] self-launch: {' '.join(cmd)}", file=sys.stderr, flush=True)
    return subprocess.run(cmd, env=env).returncode


def launcher_selftest(args, world: int, rank: int) -> int:
    """The rank-side protocol of the timed region without a GPU: gloo group, W untimed and K timed 'steps' (a sleep that is longer on
    the last rank), barrier on both sides, MAX over ranks, ONE JSON line from rank 0."""
    import torch.distributed as dist
    dist.init_process_group("gloo")
    assert dist.get_world_size() == world and dist.get_rank() == rank
    step_s = 0.002 * (1 + (rank == world - 1))
    for _ in range(args.warmup):
        time.sleep(step_s)
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(step_s)
    dist.barrier()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        dt = float(t.item())
        print(json.dumps({"metric": "launcher selftest (no kernels)", "value": round(2 * world * args.steps / dt, 3), "unit": "sleeps/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
                          "config": {"workload": "sleep", "ranks": dist.get_world_size(), "backend": "gloo"}}), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--num-classes", type=int, default=2)
    ap.add_argument("--model", default="unet_3D_icl", choices=["unet_3D_icl", "swinunetr_icl"],
                    help="unet_3D_icl = BASELINE configs[1] (the headline line); swinunetr_icl = configs[3]")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timer", action="store_true")
    ap.add_argument("--other-workload", default=None, metavar="MODEL:NC",
                    help="internal (the child of config.other_workloads): time that workload alone and print its JSON object")
    ap.add_argument("--no-other-workloads", action="store_true",
                    help="skip config.other_workloads (the nc = 16 and SwinUNETR-ICL steps timed for 10 steps each after the headline region)")
    ap.add_argument("--no-exact-compare", action="store_true",
                    help="skip the reference timing of the same step with the convolutions on the exact-fp32 MFMA kernels")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel eagerly instead of replaying a hipGraph")
    ap.add_argument("--launch", default="auto", choices=["auto", "graph"],
                    help="one rank: auto = after capture, time 3 replayed and 3 eager steps and run the timed region in the faster mode "
                         "(with the aligner heads on their own stream an eager step beats the replay when the host keeps up); "
                         "graph = always replay.  N > 1 always replays.")
    ap.add_argument("--feed", action="store_true",
                    help="after the headline region, time the same K steps with every batch built on the device by the data feed "
                         "(TwoStreamBatchSampler indices -> OnDeviceAugment.batch: rot/flip/crop of 240x240x155 volumes resident in "
                         "HBM) and the feed alone; reported as `feed` (the headline `value` keeps its HBM-resident input)")
    ap.add_argument("--loop", default="trainer", choices=["trainer", "reference"],
                    help="reference: after the headline region also time the UNCHANGED reference loop body (compat/ import root, "
                         "net_factory_3d, utils.losses, six .item() per iteration, eager) with torch.optim.SGD and with the one-line "
                         "FusedSGD swap; reported as config.reference_loop beside the headline (3D U-Net ICL, one rank)")
    ap.add_argument("--force-ddp", action="store_true", help="with --gpus 1: run the data-parallel step on a one-rank RCCL group")
    ap.add_argument("--project-world", type=int, default=0,
                    help="with --gpus 1 --force-ddp: add config.ddp_plan.projection — this one-rank step projected to W ranks with "
                         "ddp.project_world at the assumed 300 / 250 GB/s all-gather / all-reduce rates per rank (a projection from this "
                         "run's own numbers, NOT a measurement; no multi-GPU node was available to any round)")
    ap.add_argument("--launcher-selftest", action="store_true",
                    help="exercise ONLY the multi-rank plumbing (self-launch, rendezvous, barrier-bracketed timing, MAX over ranks, "
                         "rank-0 JSON relay) on a gloo group with a sleep as the step: no GPU, no kernels; used by tests/")
    args = ap.parse_args()

    if args.other_workload:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a HIP device")
        mname, mnc = args.other_workload.split(":")
        torch.cuda.set_device(0)
        print(json.dumps(other_workload(mname, int(mnc), args.steps, args.warmup, torch.device("cuda", 0))), flush=True)
        return
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(self_launch(args, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"bench.py --gpus {args.gpus} does not match WORLD_SIZE={world} of the launcher")
    if args.launcher_selftest:
        raise SystemExit(launcher_selftest(args, world, rank))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device; the product path has no CPU fallback")
    # ICL_BENCH_SHARE_GPU=1 (with ICL_BENCH_BACKEND=gloo): every rank on device 0 — a REHEARSAL of the multi-rank code path (calibration,
    # three-graph capture, eager collectives between the replays, MAX-over-ranks timing, the plan on the JSON line) on a one-GPU box; RCCL
    # refuses two ranks on one device, gloo stages through the host: the numbers of such a run mean nothing and the line says so
    share = os.environ.get("ICL_BENCH_SHARE_GPU") == "1"
    backend = os.environ.get("ICL_BENCH_BACKEND", "nccl")
    if share:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from icl_amd import ops
    from icl_amd.networks.unet_3D_icl import unet_3D_icl
    from icl_amd.trainer import ICLConfig, ICLTrainer
    from icl_amd.utils.hashfill import synthetic_labels, synthetic_volume

    ddp = None
    use_ddp = world > 1 or args.force_ddp
    if use_ddp:
        import torch.distributed as dist
        from icl_amd.ddp import GradientReducer
        if world == 1:   # --force-ddp: a one-rank RCCL group, to exercise the data-parallel step on a single GPU
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        elif backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    torch.manual_seed(1337 + rank)
    nc = args.num_classes
    if args.model == "swinunetr_icl":
        from icl_amd.networks.swinunetr_icl import SwinUNETR_icl
        model = SwinUNETR_icl(img_size=(96, 96, 96), in_channels=1, out_channels=nc, feature_size=48, device=dev)
    else:
        model = unet_3D_icl(n_classes=nc, in_channels=1, device=dev)
    model.train()
    if use_ddp:
        ddp = GradientReducer(model, world, force=args.force_ddp)
        ddp.broadcast_parameters()
        # measure the node's all-gather / all-reduce rates on a parameter-sized buffer (MAX over ranks) and set the crossover of the
        # factored exchange from them instead of the assumed 300 GB/s; the chosen plan is printed on the JSON line (config.ddp_plan)
        ddp.calibrate()
    cfg = ICLConfig(num_classes=nc, labeled_bs=1, base_lr=0.02 if nc == 16 else 0.01,
                    w_pse=0.1 if nc == 16 else 1.0)
    trainer = ICLTrainer(model, cfg, ddp)
    # synthetic batch, resident in HBM before the timed region: [labeled, unlabeled]
    vol = synthetic_volume((2, 1, 96, 96, 96), 1337 + rank, device=dev)
    lab = synthetic_labels((1, 96, 96, 96), 4242 + rank, nc, device=dev)

    def barrier():
        if use_ddp:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    graphed = not args.no_graph
    launch_probe = None
    if graphed:
        # N > 1: forward/backward graph, eager RCCL collectives, optimiser graph (ICLTrainer.capture)
        ok = 1
        try:
            trainer.capture(vol, lab, warmup=max(args.warmup, 2))   # warm-up steps run eagerly inside capture()
        except Exception as e:   # noqa: BLE001 — a rank that cannot capture must not leave the others in a collective
            if ddp is None:
                raise
            print(f"[bench] rank {rank}: graph capture failed ({e!r}); falling back to eager launches", file=sys.stderr, flush=True)
            ok = 0
        if ddp is not None:
            flag = torch.tensor([ok], device=dev, dtype=torch.int32)
            torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MIN)
            if int(flag.item()) == 0:
                trainer.graph = trainer.graph_forward = trainer.graph_update = None
                ddp.static = False
                graphed = False
        trainer.step(vol, lab)
        if graphed and ddp is None and args.launch == "auto":
            def probe(use_graph, k=3):
                trainer.use_graph = use_graph
                trainer.step(vol, lab)
                torch.cuda.synchronize()
                t = time.perf_counter()
                for _ in range(k):
                    trainer.step(vol, lab)
                torch.cuda.synchronize()
                return (time.perf_counter() - t) / k
            t_graph, t_eager = probe(True), probe(False)
            launch_probe = {"graph_ms": round(t_graph * 1e3, 3), "eager_ms": round(t_eager * 1e3, 3)}
            trainer.use_graph = t_graph <= t_eager
            graphed = trainer.use_graph
    else:
        for _ in range(args.warmup):
            trainer.step(vol, lab)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        last = trainer.step(vol, lab)
    barrier()
    dt = time.perf_counter() - t0
    # a step timed on garbage is not a measurement (and runs FASTER: operands that do not toggle draw less power, the chip clocks higher —
    # profiles/r5_wgrad_defer_ab.txt): the last loss and every parameter must be finite after the timed region
    last_loss = float(last["loss"]) if isinstance(last, dict) and "loss" in last else float("nan")
    params_finite = all(bool(torch.isfinite(p).all()) for p in trainer.model.parameters())
    if not (last_loss == last_loss and abs(last_loss) != float("inf") and params_finite):
        raise SystemExit(f"bench.py: the timed steps ended in a non-finite state (loss {last_loss}, parameters finite: {params_finite}); no number reported")
    if use_ddp:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    volumes = 2 * world * args.steps
    value = volumes / dt
    ddp_phases = None
    if use_ddp and graphed and trainer.graph_update is not None:
        # where a data-parallel step spends its time (five more steps, after the timed region; MAX over ranks per phase): the first real
        # scaling record explains itself — backward is compute, exposed_collective is what xGMI adds, update grows with the gathered rows
        trainer.time_phases = True
        acc = torch.zeros(3, dtype=torch.float64, device=dev)
        for _ in range(5):
            trainer.step(vol, lab)
            ph = trainer.phase_times()
            acc += torch.tensor([ph["forward_backward_ms"], ph["exposed_collective_ms"], ph["update_ms"]], dtype=torch.float64, device=dev)
        trainer.time_phases = False
        acc /= 5
        torch.distributed.all_reduce(acc, op=torch.distributed.ReduceOp.MAX)
        ddp_phases = {"forward_backward_ms": round(float(acc[0]), 3), "exposed_collective_ms": round(float(acc[1]), 3),
                      "update_ms": round(float(acc[2]), 3), "note": "mean of 5 replayed steps after the timed region, MAX over ranks per phase; "
                      "the collectives are issued after backward (nothing of them is hidden under it)"}

    feed = None
    if args.feed and rank == 0:
        # the loader of the reference trainers (dataloaders/brats2019.py: h5 volume -> RandomRotFlip -> RandomCrop(96^3) -> ToTensor ->
        # collate, four worker processes) as the build runs it: volumes resident in HBM, one fused gather kernel per batch, the
        # random draws on the host in the reference's numpy order.  Synthetic volumes at the real BraTS2019 extent.
        import numpy as np
        from icl_amd.dataloaders.brats2019 import DeviceVolumeStore, OnDeviceAugment, TwoStreamBatchSampler
        nvol, shape = 8, (240, 240, 155)
        store = DeviceVolumeStore([(synthetic_volume(shape, 500 + i).numpy(), synthetic_labels(shape, 600 + i, nc).numpy().astype(np.uint8))
                                   for i in range(nvol)], dev)
        aug = OnDeviceAugment(store, (96, 96, 96))
        np.random.seed(1337)
        sampler = TwoStreamBatchSampler(list(range(2)), list(range(2, nvol)), 2, 1)

        def batches():
            while True:
                for idx in sampler:
                    yield list(idx)
        it = batches()
        for _ in range(3):
            b = aug.batch(next(it))
        torch.cuda.synchronize()
        n_feed = 200
        t1 = time.perf_counter()
        for _ in range(n_feed):
            b = aug.batch(next(it))
        torch.cuda.synchronize()
        feed_alone = (time.perf_counter() - t1) / n_feed
        for _ in range(2):
            b = aug.batch(next(it))
            trainer.step(b["image"], b["label"][:1])
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            b = aug.batch(next(it))
            trainer.step(b["image"], b["label"][:1])
        torch.cuda.synchronize()
        dt_feed = (time.perf_counter() - t1) / args.steps
        feed = {"batches_per_s_feed_alone": round(1.0 / feed_alone, 1), "ms_per_batch_feed_alone": round(feed_alone * 1e3, 4),
                "ms_per_step_with_feed": round(dt_feed * 1e3, 3), "ms_per_step_resident_input": round(dt / args.steps * 1e3, 3),
                "volumes_per_s_with_feed": round(2.0 / dt_feed, 3),
                "source": f"{nvol} synthetic volumes {shape[0]}x{shape[1]}x{shape[2]} fp32 + uint8 labels resident in HBM, "
                          "TwoStreamBatchSampler(2 labeled, 6 unlabeled, batch 2) -> OnDeviceAugment.batch (crop_rotflip_kernel)"}
        del store, aug

    ref_loop = None
    if args.loop == "reference" and rank == 0 and world == 1 and args.model == "unet_3D_icl" and nc == 2:
        # ICLTrainer's own eager step on the same batch, for the comparison the two loops are judged against
        trainer.use_graph = False
        for _ in range(2):
            trainer.step(vol, lab)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            trainer.step(vol, lab)
        torch.cuda.synchronize()
        eager_ms = (time.perf_counter() - t1) / args.steps * 1e3
        trainer.use_graph = graphed
        ref_loop = {"source": "train_inherent_consistent_unet_3D_BraTS.py:63-64,85-90,103-133 (loop body verbatim, compat/ import root, eager, "
                              "six .item() reads per iteration), same HBM-resident synthetic batch",
                    "torch_optim_sgd": reference_loop_bench(nc, args.steps, args.warmup, dev, fused=False),
                    "fused_sgd_one_line_swap": reference_loop_bench(nc, args.steps, args.warmup, dev, fused=True),
                    "fused_sgd_one_line_swap_graphed": reference_loop_bench(nc, args.steps, args.warmup, dev, fused=True, graph=True),
                    "icl_trainer_eager_ms_per_step": round(eager_ms, 3)}

    exact = None
    if rank == 0 and world == 1 and not args.no_exact_compare and os.environ.get("ICL_CONV_SPLIT", "1") != "0":
        # the same step with every convolution on the exact-fp32 MFMA kernels (ICL_CONV_SPLIT=0), for reference: a second model and
        # trainer (own capture), timed the same way, after the headline region
        os.environ["ICL_CONV_SPLIT"] = "0"
        try:
            torch.manual_seed(1337 + rank)
            if args.model == "swinunetr_icl":
                model2 = SwinUNETR_icl(img_size=(96, 96, 96), in_channels=1, out_channels=nc, feature_size=48, device=dev)
            else:
                model2 = unet_3D_icl(n_classes=nc, in_channels=1, device=dev)
            model2.train()
            tr2 = ICLTrainer(model2, cfg, None)
            if not args.no_graph:
                tr2.capture(vol, lab, warmup=max(args.warmup, 2))
            for _ in range(2):
                tr2.step(vol, lab)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                tr2.step(vol, lab)
            torch.cuda.synchronize()
            dt2 = time.perf_counter() - t1
            exact = {"ms_per_step": round(dt2 / args.steps * 1e3, 3), "value": round(2 * args.steps / dt2, 3),
                     "note": "same step with ICL_CONV_SPLIT=0 (3x3x3 forward / input gradient on v_mfma_f32_16x16x4_f32)"}
            del tr2, model2
            torch.cuda.empty_cache()
        finally:
            os.environ["ICL_CONV_SPLIT"] = "1"

    roof = None
    if rank == 0 and not args.no_kernel_timer:
        trainer.graph = trainer.graph_forward = trainer.graph_update = None   # per-kernel HIP-event timing needs eager launches ...
        trainer.ddp = None                              # ... of this rank's step alone: the other ranks are done
        side_was = ops.SideStream.enabled
        ops.SideStream.enabled = False                  # ... on ONE stream: a kernel's duration is its own, not a share of the GPU
        try:
            with ops.KernelTimer() as kt:
                for _ in range(3):
                    trainer.step(vol, lab)
        finally:
            ops.SideStream.enabled = side_was           # (config.other_workloads below is timed with the lanes on, like the headline)
        summ = kt.summary()
        conv = {k: v for k, v in summ.items() if k.startswith("conv3d_")}
        if conv:
            # dominant kernel = the conv instantiation with the largest total time (same name as in rocprofv3's stats)
            # (forward/dgrad launches only: a wgrad call also runs its slab-reduction kernel inside the timed bracket)
            fwd_only = {k: v for k, v in conv.items() if "_fwd_" in k}
            name, (n, ms, fl, by) = max(fwd_only.items(), key=lambda kv: kv[1][1])
            ach = fl / (ms * 1e-3) / 1e12
            split = "bf16x3" in name
            peak = PEAK_BF16_MFMA_TFLOPS / SPLIT_TERMS if split else PEAK_F32_MFMA_TFLOPS
            busy = PIPE_BUSY.get(name, PIPE_BUSY.get(name.split("<")[0]))
            tot_ms = sum(v[1] for v in conv.values())
            tot_fl = sum(v[2] for v in conv.values())
            tot_by = sum(v[3] for v in conv.values())
            traffic = hbm_traffic(name)
            roof = {"bound": "mfma", "kernel": name, "achieved": round(ach, 3), "peak": round(peak, 1),
                    "unit": "TFLOP/s", "frac": round(ach / peak, 4),
                    # achieved = algorithmic fp32 FLOPs (2 * 27 * Cin * Cout per voxel) / launch time.  The split-product kernel
                    # issues 6 bf16 MFMA terms per fp32 product (exact 3-way operand splits, fp32 accumulate): its roofline is the
                    # dense bf16 peak / 6; the exact-fp32 kernels are priced against the fp32 matrix peak.
                    "peak_basis": (f"dense bf16 MFMA peak {PEAK_BF16_MFMA_TFLOPS:.0f} / {SPLIT_TERMS} terms per fp32 product"
                                   if split else "dense fp32 MFMA peak"),
                    "vs_fp32_mfma_peak": round(ach / PEAK_F32_MFMA_TFLOPS, 4),
                    **({"launch_note": "a timed launch = the convolution kernel alone: the weight planes of every layer are split "
                                       "once per step (conv_bf16x3_split_weights_multi_kernel, one launch)"} if split else {}),
                    "traffic": traffic["hbm_bytes"] if traffic else None, "traffic_detail": traffic,
                    # PMC (profiles/r3_pmc_conv.md, static): fraction of the kernel's cycles with the matrix pipe busy; the rest of
                    # the gap to the 2.4 GHz peak is the clock the chip holds under matrix load (1.8-2.1 GHz)
                    "matrix_pipe_busy_static_profile": busy,
                    # kernel cycles (SQ_BUSY_CYCLES / 32) / duration of the isolated launch in the same PMC profile: the clock the chip holds
                    # under this kernel; `peak` is a 2.4 GHz figure, so frac ~= pipe busy x clock / 2.4 (x what the lanes take in the step)
                    **({"kernel_clock_ghz_static_profile": KERNEL_CLOCK_GHZ[name]} if name in KERNEL_CLOCK_GHZ else {}),
                    "launches_per_step": n // 3, "avg_launch_us": round(ms * 1e3 / n, 2),
                    # what the event pair itself adds to a bracketed launch on the idle eager stream (NOT subtracted above: the raw
                    # durations are the conservative ones; rocprofv3's per-kernel averages in profiles/ are shorter by about this much)
                    "event_bracket_overhead_us": round(ops.event_bracket_overhead_us(dev), 2),
                    "all_conv": {"ms_per_step": round(tot_ms / 3, 3), "achieved": round(tot_fl / (tot_ms * 1e-3) / 1e12, 3),
                                 "vs_fp32_mfma_peak": round(tot_fl / (tot_ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
                                 "algorithmic_gbs": round(tot_by / (tot_ms * 1e-3) / 1e9, 1),
                                 "hbm_frac": round(tot_by / (tot_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4)},
                    # every convolution kernel of the step; frac against that kernel's own matrix peak (bf16 / 6 for the split-product
                    # kernels, fp32 MFMA for the others; a weight-gradient bracket includes its slab-sum launch).  Since round 3 the
                    # 24^3 layers run on the split-product forward kernel too: launches that cannot fill the chip pull the average of
                    # their instantiation down (<2, 8, 60>: 13 launches per step, 4 of them at 48^3)
                    "per_kernel": {k: {"launches_per_step": v[0] // 3, "avg_launch_us": round(v[1] * 1e3 / v[0], 2),
                                       "tflops": round(v[2] / (v[1] * 1e-3) / 1e12, 2),
                                       "frac": round(v[2] / (v[1] * 1e-3) / 1e12 /
                                                     (PEAK_BF16_MFMA_TFLOPS / SPLIT_TERMS if ("bf16x3" in k or "wgrad_tr" in k or "wgrad_zs" in k) else PEAK_F32_MFMA_TFLOPS), 4)}
                                   for k, v in sorted(conv.items())}}
            st = {k: v for k, v in summ.items() if k.startswith("linear_stream")}
            if st:   # the 13,824^2 token-axis MLP products: the weight matrix streamed once per launch (csrc/kernels/gemm.h)
                roof["mlp2_weight_stream"] = {k: {"bound": "hbm", "launches_per_step": v[0] // 3, "avg_launch_us": round(v[1] * 1e3 / v[0], 2),
                                                  "achieved": round(v[3] / (v[1] * 1e-3) / 1e9, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                                  "frac": round(v[3] / (v[1] * 1e-3) / 1e9 / PEAK_HBM_GBS, 4),
                                                  "note": "launch = streaming kernel + slab sum; bytes = weight matrix"}
                                              for k, v in sorted(st.items())}
            fu = summ.get("linear_dgrad_sgd_kernel")
            if fu:   # single rank: input gradient + SGD step of the four 13,824^2 matrices, one pass over p and m (16 B per weight)
                gbs = fu[3] / (fu[1] * 1e-3) / 1e9
                roof["mlp2_dgrad_sgd_one_pass"] = {"bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                                   "frac": round(gbs / PEAK_HBM_GBS, 4), "launches_per_step": fu[0] // 3,
                                                   "avg_launch_us": round(fu[1] * 1e3 / fu[0], 2),
                                                   "algorithmic_bytes_per_launch": int(fu[3] / fu[0])}
            sg = summ.get("sgd_factored_kernel")
            if sg:   # the factored SGD update of the token-axis matrices that are not updated inside their backward (1,728^2; all, multi-rank)
                gbs = sg[3] / (sg[1] * 1e-3) / 1e9
                roof["sgd_factored_update"] = {"bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                               "frac": round(gbs / PEAK_HBM_GBS, 4), "launches_per_step": sg[0] // 3,
                                               "avg_launch_us": round(sg[1] * 1e3 / sg[0], 2),
                                               "algorithmic_bytes_per_launch": int(sg[3] / sg[0]),
                                               "note": "one rank: sspa's two 13,824^2 matrices as NARROW persistent launches (128 of the 256 CUs by design, "
                                                       "FusedSGD.update_placement 'deep': they run under the deep backward levels) + the four 1,728^2 ones"}
            wa = {k: v for k, v in summ.items() if k.startswith("window_attn")}
            if wa:   # SwinUNETR: the fused window-attention kernels (fp32 MFMA), forward and backward (dQKV + bias gradient)
                roof["window_attention"] = {k: {"launches_per_step": v[0] // 3, "ms_per_step": round(v[1] / 3, 3),
                                                "tflops": round(v[2] / (v[1] * 1e-3) / 1e12, 2),
                                                "frac": round(v[2] / (v[1] * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4)}
                                            for k, v in sorted(wa.items())}

    projection = None
    if rank == 0 and use_ddp and world == 1 and args.project_world > 1 and ddp.last_plan:
        from icl_amd.ddp import project_world
        mats = [dict(rows=m["rows_gathered"], elems=m["shape"][0] * m["shape"][1], out_rows=m["shape"][0]) for m in ddp.last_plan]
        # bytes one rank contributes to the factor-row all-gather: a g row (out floats) + an x row (in floats) per factor row of the
        # 13,824^2 matrices (the 1,728^2 ones add < 2 %)
        fac_bytes = 4.0 * sum(m["rows_gathered"] * (m["shape"][0] + m["shape"][1]) for m in ddp.last_plan)
        small = 4.0 * sum(p.numel() for p in model.parameters() if p.grad is not None)
        projection = project_world(mats, args.project_world, dt / args.steps * 1e3, small_grad_bytes=small, factor_bytes_per_rank=fac_bytes)
        projection["note"] = ("projected from this one-rank measurement with ddp.project_world (update times of tools/sgd_probe.py, assumed "
                              "link rates); the 8-GPU bench is the driver's to run")

    if rank == 0:
        out = {
            "metric": "3D volumes/sec/node (fwd+bwd, 96^3 patch)", "value": round(value, 3), "unit": "volumes/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{'SwinUNETR' if args.model == 'swinunetr_icl' else '3D U-Net'} ICL BraTS-shape synthetic 96x96x96, num_classes={nc}, "
                                   f"batch=2 per GPU (1 labeled + 1 unlabeled), full ICL step incl. SGD",
                       "global_batch": 2 * world, "parallelism": f"dp{world}",
                       **({"rehearsal": "every rank on ONE device over gloo (ICL_BENCH_SHARE_GPU=1): exercises the multi-rank code path, "
                                        "the timing is meaningless"} if share else {}),
                       **({"rccl_ranks": torch.distributed.get_world_size(), "collective_backend": torch.distributed.get_backend(),
                           **({"ddp_phases": ddp_phases} if ddp_phases else {}),
                           "ddp_plan": {"rates": ddp.rates, "matrices": ddp.last_plan, **({"projection": projection} if projection else {})}}
                          if use_ddp else {}),
                       "launch": ("eager" if not graphed else "hipGraph replay" if ddp is None else
                                  "hipGraph replay (forward/backward, optimiser) + eager RCCL collectives"),
                       **({"launch_probe": launch_probe} if launch_probe else {}),
                       "finite_after_timed_steps": {"last_loss": round(last_loss, 6), "parameters": params_finite},
                       # every tensor, accumulator and result is fp32.  With ICL_CONV_SPLIT on (default) the 3x3x3 forward / input-
                       # gradient products of the large layers are formed from exact three-way bf16 splits of the fp32 operands (six
                       # bf16 MFMA terms per product, fp32 accumulate): outputs as close to fp64 as the fp32 MFMA kernels'
                       # (tests/test_gpu_parity.py::test_split_bf16_convolution_is_as_accurate_as_the_fp32_mfma_path); the
                       # bf16 MFMA's internal 32-product sums are not rounded to nearest (-0.36 * 2^-24 coherent offset on same-sign
                       # sums); "accuracy_trade" states what was measured on the most cancellation-heavy gradient of the step
                       "conv_products": ("exact 3-way bf16 splits of fp32 operands, 6 MFMA terms, fp32 accumulate (ICL_CONV_SPLIT=1)"
                                         if os.environ.get("ICL_CONV_SPLIT", "1") != "0" else "v_mfma_f32_16x16x4_f32"),
                       **({"accuracy_trade": "split products: logits / maps / losses within 1e-3 of the reference, gradient norms within 1e-2, as with "
                                             "the fp32 MFMA kernels; PER STEP the deepest gradients sit about 2x further from the reference than on "
                                             "the exact path (dense 96^3 weight gradients, max-norm: conv1.conv2 5.6e-3 split vs 2.5e-3 exact, "
                                             "conv2.conv2 6.0e-3 vs 2.9e-3, up_concat2.conv.conv1 1.9e-3 vs 9.7e-4; profiles/r5_dense_wgrad_errors.txt); "
                                             "bf16 MFMA accumulation offset -0.36*2^-24 on same-sign sums; over 10 reference steps and over 200 "
                                             "steps split-vs-exact is indistinguishable from exact-vs-exact under a 1e-7 input perturbation "
                                             "(profiles/r6_ten_steps_noise.txt, profiles/r6_drift.txt: loss within 3.5e-2 of 2.5-4.0 at every step)"}
                          if os.environ.get("ICL_CONV_SPLIT", "1") != "0" else {}),
                       **({"exact_fp32_mfma_convolutions": exact} if exact else {}),
                       **({"reference_loop": ref_loop} if ref_loop else {}),
                       **({"feed": feed} if feed else {})},
            "roofline": roof,
        }
        if world == 1 and not use_ddp and not args.no_other_workloads and args.model == "unet_3D_icl" and nc == 2:
            # BASELINE.json configs[4] and configs[3] on this GPU, after the headline region and before the CPU baseline
            del trainer, model
            trainer = model = None
            torch.cuda.empty_cache()
            other = {}
            for key, (mname, mnc) in (("unet3d_icl_nc16", ("unet_3D_icl", 16)), ("swinunetr_icl_nc2", ("swinunetr_icl", 2))):
                other[key] = other_workload_child(mname, mnc, 10, args.warmup)
            out["config"]["other_workloads"] = other
        if world == 1 and not args.no_cpu_baseline:
            trainer = model = None
            torch.cuda.empty_cache()
            out["cpu_baseline"] = cpu_baseline(nc, args.model)
        try:    # RCCL / HIP banners sit in the C stdio buffer of a piped stdout: push them out first, the JSON line comes last
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        print(json.dumps(out), flush=True)
    if use_ddp:
        torch.distributed.barrier()    # rank 0 may still have been timing kernels
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
