"""World-size 2 / 3 / 4 / 8 checks of the data-parallel gradient reducer on the gloo backend (CPU).

SURVEY.md §8e: W ranks == W independent reference steps with averaged gradients; parameters whose grad is None
are skipped, never zero-filled (torch SGD skips them too, so weight decay must not touch them)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class Tiny(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.a = torch.nn.Linear(7, 5)
        self.unused = torch.nn.Linear(3, 3)       # never used in forward: grad stays None
        self.big = torch.nn.Parameter(torch.randn(300, 40))
        self.b = torch.nn.Linear(5, 2)

    def forward(self, x):
        return self.b(torch.tanh(self.a(x))) + (self.big.sum() * 1e-3)


def _worker(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from icl_amd.ddp import GradientReducer
    torch.manual_seed(100 + rank)          # different initial weights per rank: broadcast must fix that
    model = Tiny()
    # small buckets (multi-tensor and single-tensor paths) and a low overlap threshold so that `big` takes the
    # start-the-all-reduce-from-the-autograd-hook path that the 764 MB mlp2 gradients take on the GPUs
    red = GradientReducer(model, world, bucket_bytes=4096, overlap_min_elems=10000)
    red.broadcast_parameters()
    torch.manual_seed(7)
    data = torch.randn(world, 4, 7)        # every rank knows all shards so it can build the reference result
    opt = torch.optim.SGD(model.parameters(), lr=0.1, momentum=0.9, weight_decay=1e-2)
    model(data[rank]).pow(2).mean().backward()
    red.reduce_gradients()
    assert model.unused.weight.grad is None and model.unused.bias.grad is None
    got = {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
    unused_before = model.unused.weight.detach().clone()
    opt.step()
    assert torch.equal(model.unused.weight.detach(), unused_before)   # skipped: no weight decay applied
    # reference: same weights, mean over ranks of the per-shard gradients
    if rank == 0:
        torch.save({k: v for k, v in got.items()}, os.path.join(tmp, "g0.pt"))
    dist.barrier()
    g0 = torch.load(os.path.join(tmp, "g0.pt"))
    for k in got:
        assert torch.allclose(got[k], g0[k], rtol=0, atol=0), k   # all ranks hold identical averaged grads
    # independent recomputation on one process
    torch.manual_seed(100)                 # rank 0's initial weights are what was broadcast
    single = Tiny()
    acc = None
    for r in range(world):
        single.zero_grad(set_to_none=True)
        single(data[r]).pow(2).mean().backward()
        gs = {k: p.grad.clone() for k, p in single.named_parameters() if p.grad is not None}
        acc = gs if acc is None else {k: acc[k] + gs[k] for k in gs}
    for k in got:
        assert torch.allclose(got[k], acc[k] / world, rtol=1e-5, atol=1e-6), k
    dist.destroy_process_group()


def _worker_factored(rank, world, port, tmp):
    """Factored weight gradients (ops.FactoredGrads): ranks all-gather the (g, x) row blocks; the update applied by FusedSGD
    equals SGD on the rank-averaged dense gradient."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "hipemu"))
    from build_emu import build_emu
    from icl_amd import _lib, ops
    from icl_amd.ddp import GradientReducer
    from icl_amd.networks.aligner import Linear
    from icl_amd.optim import FusedSGD
    _lib._use_library_for_tests(build_emu(), host_pointers=True)
    ops.FactoredGrads.min_elems = 1000
    # two passes: every rank applies the whole gathered update / the crossover form (each rank updates its rows, then the ranks
    # all-gather the updated rows: GradientReducer.post_update)
    for shard_rows, nrows in ((0, 48), (1, 48), (1, 50)):
        # 48 rows divide by every tested world size; 50 only by 2: with 3 or 4 ranks the row-sharded form is not available
        # (p.shape[0] % world != 0) and the reducer must fall back to the whole gathered update without being told
        can_shard = bool(shard_rows) and nrows % world == 0
        torch.manual_seed(3)
        lin = Linear(64, nrows)
        red = GradientReducer(lin, world, shard_min_rows=shard_rows)
        red.broadcast_parameters()
        w0, b0 = lin.weight.detach().clone(), lin.bias.detach().clone()
        torch.manual_seed(11)
        data = torch.randn(world, 6, 64)
        opt = FusedSGD(lin.parameters(), lr=0.1, momentum=0.9, weight_decay=1e-2)
        want_w, want_b = w0, b0
        rw, rb = torch.nn.Parameter(w0.clone()), torch.nn.Parameter(b0.clone())
        ref_opt = torch.optim.SGD([rw, rb], lr=0.1, momentum=0.9, weight_decay=1e-2)
        for step in range(2):            # second step: momentum buffers of the (sharded) rows carry over
            opt.zero_grad()
            with ops.FactoredGrads(True):
                lin(data[rank] + step).pow(2).mean().backward()
            assert lin.weight.grad is None and lin.weight._icl_factors
            red.reduce_gradients()
            assert lin.weight._icl_factors[0][0].shape[0] == 6 * world
            assert (lin.weight._icl_shard is not None) == can_shard
            opt.step()
            red.post_update()
            # reference: dense gradients of every shard on one process, averaged, torch SGD
            gw, gb = torch.zeros_like(rw), torch.zeros_like(rb)
            for r in range(world):
                rw.grad = rb.grad = None
                torch.nn.functional.linear(data[r] + step, rw, rb).pow(2).mean().backward()
                gw += rw.grad / world
                gb += rb.grad / world
            rw.grad, rb.grad = gw, gb
            ref_opt.step()
            assert torch.allclose(lin.weight.detach(), rw.detach(), rtol=1e-5, atol=1e-6), (shard_rows, nrows, step)
            assert torch.allclose(lin.bias.detach(), rb.detach(), rtol=1e-5, atol=1e-6)
            assert lin.weight._icl_shard is None      # consumed by the step: the decision never outlives its factors
    if world == 8:
        # the only world size the scaling target names, with the factor-row counts of config 5 (num_classes = 16: 128 rows per rank in
        # `sspa`, 64 in `uscl` -> 1024 / 512 gathered rows) and the reducer's DEFAULT crossover (768 gathered rows): the 1024-row matrix
        # is updated row-sharded when its row count divides by 8 (13,824 does) and whole when it does not; the 512-row one whole
        for rows_rank, nrows, want_shard in ((128, 48, True), (128, 50, False), (64, 48, False)):
            torch.manual_seed(3)
            lin = Linear(64, nrows)
            red = GradientReducer(lin, world)
            assert red.shard_min_rows == 768
            red.broadcast_parameters()
            rw, rb = torch.nn.Parameter(lin.weight.detach().clone()), torch.nn.Parameter(lin.bias.detach().clone())
            ref_opt = torch.optim.SGD([rw, rb], lr=0.1, momentum=0.9, weight_decay=1e-2)
            opt = FusedSGD(lin.parameters(), lr=0.1, momentum=0.9, weight_decay=1e-2)
            torch.manual_seed(21)
            data = torch.randn(world, rows_rank, 64)
            for step in range(2):
                opt.zero_grad()
                with ops.FactoredGrads(True):
                    lin(data[rank] + step).pow(2).mean().backward()
                red.reduce_gradients()
                assert lin.weight._icl_factors[0][0].shape[0] == rows_rank * world
                assert (lin.weight._icl_shard is not None) == want_shard, (rows_rank, nrows)
                opt.step()
                red.post_update()
                gw, gb = torch.zeros_like(rw), torch.zeros_like(rb)
                for r in range(world):
                    rw.grad = rb.grad = None
                    torch.nn.functional.linear(data[r] + step, rw, rb).pow(2).mean().backward()
                    gw += rw.grad / world
                    gb += rb.grad / world
                rw.grad, rb.grad = gw, gb
                ref_opt.step()
                assert torch.allclose(lin.weight.detach(), rw.detach(), rtol=1e-5, atol=1e-6), (rows_rank, nrows, step)
                assert torch.allclose(lin.bias.detach(), rb.detach(), rtol=1e-5, atol=1e-6)
    # the shard decision flips between steps (sharded, sharded, whole, sharded, whole): the momentum rows the other ranks own are
    # exchanged before the buffer is used whole again, and a saved state holds the complete buffer on every rank
    torch.manual_seed(3)
    lin = Linear(64, 48)
    red = GradientReducer(lin, world, shard_min_rows=1)
    red.broadcast_parameters()
    rw, rb = torch.nn.Parameter(lin.weight.detach().clone()), torch.nn.Parameter(lin.bias.detach().clone())
    ref_opt = torch.optim.SGD([rw, rb], lr=0.1, momentum=0.9, weight_decay=1e-2)
    opt = FusedSGD(lin.parameters(), lr=0.1, momentum=0.9, weight_decay=1e-2)
    torch.manual_seed(12)
    data = torch.randn(world, 6, 64)
    for step, sharded in enumerate((1, 1, 0, 1, 0)):
        red.shard_min_rows = 1 if sharded else 0
        opt.zero_grad()
        with ops.FactoredGrads(True):
            lin(data[rank] + step).pow(2).mean().backward()
        red.reduce_gradients()
        assert (lin.weight._icl_shard is not None) == bool(sharded)
        opt.step()
        red.post_update()
        gw, gb = torch.zeros_like(rw), torch.zeros_like(rb)
        for r in range(world):
            rw.grad = rb.grad = None
            torch.nn.functional.linear(data[r] + step, rw, rb).pow(2).mean().backward()
            gw += rw.grad / world
            gb += rb.grad / world
        rw.grad, rb.grad = gw, gb
        ref_opt.step()
        assert torch.allclose(lin.weight.detach(), rw.detach(), rtol=1e-5, atol=1e-6), step
        if step == 3:     # saved right after a sharded step
            if sharded:
                # state_dict() is LOCAL (rank 0 alone may call it): with rows still sharded it refuses instead of communicating
                with pytest.raises(RuntimeError, match="consolidate_momentum"):
                    opt.state_dict()
            opt.consolidate_momentum()      # the collective, on every rank
            sd = opt.state_dict()
            m = sd["state"][0]["momentum_buffer"]
            assert torch.allclose(m, ref_opt.state[rw]["momentum_buffer"], rtol=1e-5, atol=1e-6)
            assert "momentum_shard" not in sd["state"][0]
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", [2, 3, 4, 8])
def test_factored_gradient_exchange(tmp_path, world):
    from hipemu.build_emu import build_emu  # noqa: F401  (build once, before the workers race for it)
    build_emu()
    mp.spawn(_worker_factored, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)


def _worker_calibrate(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from icl_amd import ops
    from icl_amd.ddp import GradientReducer, exchange_plan
    red = GradientReducer(Tiny(), world)
    before = ops.FactoredGrads.max_rows_gathered
    rates = red.calibrate(nbytes=1 << 20, iters=1)
    assert rates["measured"] and rates["allgather_gbps"] > 0 and rates["allreduce_gbps"] > 0
    everyone = [None] * world
    dist.all_gather_object(everyone, (rates, red.shard_min_rows, red.max_rows_gathered))
    assert all(e == everyone[0] for e in everyone)        # MAX-reduced times: every rank derives the same thresholds
    # the thresholds are the crossovers of the priced alternatives at the measured rates
    if red.shard_min_rows:
        assert exchange_plan(red.shard_min_rows, 13824 * 13824, world, rates["allgather_gbps"], rates["allreduce_gbps"])[0] == "shard"
    # the measurement stays on the reducer — nothing process-global changes, an explicit setting wins, and a saved plan
    # pins a resumed run to the same arithmetic path
    assert ops.FactoredGrads.max_rows_gathered == before
    explicit = GradientReducer(Tiny(), world, shard_min_rows=640)
    explicit.calibrate(nbytes=1 << 20, iters=1)
    assert explicit.shard_min_rows == 640 and explicit.rates["explicit"]["shard_min_rows"]
    resumed = GradientReducer(Tiny(), world)
    resumed.load_state_dict(red.state_dict())
    resumed.calibrate(nbytes=1 << 20, iters=1)
    assert (resumed.shard_min_rows, resumed.max_rows_gathered) == (red.shard_min_rows, red.max_rows_gathered)
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_reducer_calibration_sets_the_same_crossover_on_every_rank(tmp_path):
    mp.spawn(_worker_calibrate, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)


def test_exchange_plan_prices_the_alternatives():
    from icl_amd.ddp import exchange_plan, update_ms
    assert abs(update_ms(128) - 0.80) < 1e-9 and update_ms(192) > update_ms(128)
    # a slow all-gather keeps the whole update; a fast one shards from a few hundred rows on; indivisible rows never shard
    assert exchange_plan(512, 13824 * 13824, 8, 100.0)[0] == "whole"
    assert exchange_plan(512, 13824 * 13824, 8, 1000.0)[0] == "shard"
    assert "shard" not in exchange_plan(512, 13824 * 13824, 8, 1000.0, divisible=False)[1]
    # nc = 16 at 8 ranks (1024 / 512 gathered rows): projected per-matrix time at the assumed 300 GB/s
    mode, cost = exchange_plan(1024, 13824 * 13824, 8, 300.0, 250.0)
    assert mode == "shard" and cost["shard"] < cost["whole"] < cost["dense"]


def test_projection_of_the_eight_rank_step_from_a_one_rank_measurement():
    """ddp.project_world: config 5 (num_classes = 16: 128 / 64 factor rows per rank) on 8 ranks at the assumed 300 / 250 GB/s — the
    1024-row matrices shard, the 512-row ones stay whole, the row all-gathers hide under the whole updates except for their excess."""
    from icl_amd.ddp import project_world, update_ms
    n = 13824 * 13824
    mats = [dict(rows=128, elems=n, out_rows=13824)] * 2 + [dict(rows=64, elems=n, out_rows=13824)] * 2
    pr = project_world(mats, 8, base_ms=16.0, factor_bytes_per_rank=4.0 * 13824 * 2 * (128 + 128 + 64 + 64))
    assert [m["mode"] for m in pr["matrices"]] == ["shard", "shard", "whole", "whole"]
    assert abs(pr["update_extra_ms"] - (2 * (update_ms(1024) / 8 - update_ms(128)) + 2 * (update_ms(512) - update_ms(64)))) < 1e-2
    assert pr["row_allgather_exposed_ms"] == round(max(0.0, pr["row_allgather_ms"] - 2 * update_ms(512)), 3)
    assert 16.0 < pr["projected_ms_per_step"] < 21.0, pr
    one = project_world(mats, 1, base_ms=16.0)
    assert one["projected_ms_per_step"] == 16.0 and all(m["mode"] == "whole" for m in one["matrices"])


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", [2, 3, 4, 8])
def test_gradient_reducer(tmp_path, world):
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)


def _worker_icl_trainer(rank, world, port, tmp):
    """ICLTrainer + GradientReducer on a small instance of the real model class (unet_3D_icl: backbone, both aligners, factored
    token-axis MLP gradients, grad-None parameters) on the CPU emulation of the kernels: the gradients every rank ends up with are
    the mean of the per-rank gradients (gathered and averaged here in plain Python), parameters stay identical across ranks and
    the applied update is SGD(momentum, weight decay) on that mean."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), HIPEMU_THREADS="4")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "hipemu"))
    from build_emu import build_emu
    from icl_amd import _lib, ops
    from icl_amd.ddp import GradientReducer
    from icl_amd.networks.unet_3D_icl import unet_3D_icl
    from icl_amd.trainer import ICLConfig, ICLTrainer
    from icl_amd.utils.hashfill import synthetic_labels, synthetic_volume
    _lib._use_library_for_tests(build_emu(), host_pointers=True)
    torch.set_num_threads(2)
    ops.FactoredGrads.min_elems = 64 * 64          # the 64-token mlp2 matrices of the finest scale stay factored
    torch.manual_seed(50 + rank)                   # different initial weights per rank: broadcast must fix that
    model = unet_3D_icl(feature_scale=16, n_classes=2, in_channels=1, icl_in_resolutions=(1, 2, 4), icl_heads=(8, 4, 2))
    red = GradientReducer(model, world, bucket_bytes=1 << 16)
    red.broadcast_parameters()
    lr, mom, wd = 0.05, 0.9, 1e-2
    tr = ICLTrainer(model, ICLConfig(num_classes=2, labeled_bs=1, patch_size=(16, 16, 16), base_lr=lr, weight_decay=wd), ddp=red)
    vol = synthetic_volume((2, 1, 16, 16, 16), 900 + rank)
    lab = synthetic_labels((1, 16, 16, 16), 950 + rank, 2)
    p0 = {k: p.detach().clone() for k, p in model.named_parameters()}
    tr._forward_backward(vol, lab)

    def dense_grads():
        out = {}
        for k, p in model.named_parameters():
            fac = getattr(p, "_icl_factors", None)
            if fac:
                out[k] = sum(g.detach().t() @ x.detach() for g, x in fac)
            elif p.grad is not None:
                out[k] = p.grad.detach().clone()
        return out

    local = dense_grads()
    factored = [k for k, p in model.named_parameters() if getattr(p, "_icl_factors", None)]
    assert sum("mlp2" in k for k in factored) == 4 and len(factored) >= 4, factored   # (+ other Linear layers of >= 64 x 64)
    everyone = [None] * world
    dist.all_gather_object(everyone, {k: v.numpy() for k, v in local.items()})
    assert all(sorted(e) == sorted(local) for e in everyone)           # the same grad-None set on every rank
    none = [k for k, _ in model.named_parameters() if k not in local]
    assert len(none) == 33 and any("uscl.guided_Q" == k for k in none)   # skipped, never zero-filled
    red.reduce_gradients()
    reduced = dense_grads()
    assert sorted(reduced) == sorted(local)
    for k, v in reduced.items():
        mean = sum(torch.as_tensor(e[k]) for e in everyone) / world
        assert torch.allclose(v, mean, rtol=1e-4, atol=1e-7 + 1e-5 * float(mean.abs().max())), k
    tr._apply_update()
    sums = [None] * world
    dist.all_gather_object(sums, {k: float(p.detach().double().sum()) for k, p in model.named_parameters()})
    assert sums[0] == sums[rank]                                          # replicas stay bit-identical
    for k, p in model.named_parameters():
        if k in none:
            assert torch.equal(p.detach(), p0[k]), k                      # no weight decay on grad-None parameters
        else:
            want = p0[k] - lr * (reduced[k] + wd * p0[k])                 # first step: momentum buffer = d
            assert torch.allclose(p.detach(), want, rtol=1e-5, atol=1e-6), k
    # Round 5: the whole `step()` with ROW-SHARDED updates of the token-axis matrices — the updates of the sharded matrices first
    # (FusedSGD.step_subset), the all-gathers of their rows started asynchronously, the rest of the optimiser step under them, the
    # join at the end (ICLTrainer._step_body).  Same model, a second trainer whose reducer shards from one gathered row on: against a
    # twin that applies every gathered update whole, parameters agree to rounding after two steps and replicas stay bit-identical.
    def two_steps(shard_rows):
        torch.manual_seed(77)
        m = unet_3D_icl(feature_scale=16, n_classes=2, in_channels=1, icl_in_resolutions=(1, 2, 4), icl_heads=(8, 4, 2))
        r = GradientReducer(m, world, bucket_bytes=1 << 16, shard_min_rows=shard_rows)
        r.broadcast_parameters()
        t = ICLTrainer(m, ICLConfig(num_classes=2, labeled_bs=1, patch_size=(16, 16, 16), base_lr=lr, weight_decay=wd), ddp=r)
        for mod in m.modules():
            if mod.__class__.__name__ == "Dropout3":
                mod.p = 0.0
            if hasattr(mod, "drop_prob"):
                mod.drop_prob = 0.0
        nshard = 0
        for _ in range(2):
            t.step(vol, lab)
            nshard = max(nshard, len(r.sharded_params()))
        return {k: p.detach().clone() for k, p in m.named_parameters()}, nshard

    whole, n0 = two_steps(0)
    shard, n1 = two_steps(1)
    assert n0 == 0 and n1 >= 4, (n0, n1)                                   # the four mlp2 matrices (64 rows, divisible by the world size)
    for k in whole:
        assert torch.allclose(whole[k], shard[k], rtol=1e-5, atol=1e-6), k
    sums = [None] * world
    dist.all_gather_object(sums, {k: float(v.double().sum()) for k, v in shard.items()})
    assert sums[0] == sums[rank]
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_icl_trainer_with_gradient_reducer_world2(tmp_path):
    world = 2
    from hipemu.build_emu import build_emu  # noqa: F401
    build_emu()
    mp.spawn(_worker_icl_trainer, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
