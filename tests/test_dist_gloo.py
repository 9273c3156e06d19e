"""CPU, world_size 2 over gloo: the two collectives of the hot path -- the mean all-reduce of the flat
gradient (DDP train step) and the all-gather of per-candidate CEM costs -- behave as the N>1 GPU path
expects (same code, RCCL there)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, N, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from robot_aware_control_amd.trainer import allreduce_flat_grad
    from robot_aware_control_amd.trajectory_sampler import gather_costs, shard_bounds
    torch.manual_seed(0)
    full = torch.randn(world, 2_003)
    g = full[rank].clone()
    allreduce_flat_grad(g, bucket_mb=0)          # bucket size clamps to >= 1 element -> many buckets
    g2 = full[rank].clone()
    allreduce_flat_grad(g2, bucket_mb=64)        # one bucket
    ok_grad = torch.allclose(g, full.mean(0), atol=1e-6) and torch.allclose(g2, full.mean(0), atol=1e-6)
    # overlapped reducer: two parameters' slices go first (out of order, as the deferred wgrads finish), the rest last
    from robot_aware_control_amd.trainer import GradReducer
    g3 = full[rank].clone()
    pa, pb = torch.nn.Parameter(torch.zeros(300)), torch.nn.Parameter(torch.zeros(17, 20))
    pa.grad, pb.grad = g3[1200:1500], g3[100:440].view(17, 20)
    red = GradReducer(g3, bucket_mb=0)
    red.step = 128
    red.ready(pa)
    red.ready(pb)
    red.ready(torch.nn.Parameter(torch.zeros(3)))  # no grad / foreign storage: ignored
    red.finish()
    ok_grad = ok_grad and torch.allclose(g3, full.mean(0), atol=1e-6)
    costs = torch.arange(N, dtype=torch.float64) * -1.5
    lo, hi = shard_bounds(N, world, rank)
    got = gather_costs(costs[lo:hi].clone(), N, world, rank)
    ok_cost = np.array_equal(got, costs.numpy())
    # every rank then selects the same elites (cem.py:96-97)
    top = torch.from_numpy(got).topk(3).indices.tolist()
    q.put((rank, bool(ok_grad), bool(ok_cost), top))
    dist.destroy_process_group()


def _run(N):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, N, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return sorted(res)


def test_allreduce_and_cost_gather_world2():
    for N in (10, 11):  # even and ragged shards
        res = _run(N)
        assert all(r[1] and r[2] for r in res), res
        assert res[0][3] == res[1][3] == [0, 1, 2]


def _data_worker(rank, world, port, root, q):
    import argparse
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from robot_aware_control_amd import data as D
    c = argparse.Namespace(data_root=root, load_movement_info=False, video_length=8, n_past=1, n_future=2, action_dim=4,
                           robot_dim=5, robot_joint_dim=7, impute_autograsp_action=False, image_width=64, image_height=48,
                           seed=3, preload_ram=False, preprocess_action="raw", experiment="train_robonet",
                           model_use_heatmap=False, train_val_split=0.75, img_augmentation=False, data_threads=0,
                           batch_size=3, test_batch_size=2)
    train_loader, _ = D.create_loaders(c)
    gen = D.get_batch(train_loader, torch.device("cpu"), prefetch=False)
    epochs = []
    for _ in range(2):  # two epochs of this rank's shard (12 files / 2 ranks / batch 3 = 2 batches each)
        epochs.append([int(i) for _ in range(2) for i in next(gen)["idx"]])
    starts = train_loader.dataset._rng.randint(0, 1 << 30)  # the ranks' window-start streams differ
    q.put((rank, epochs, int(starts)))
    dist.destroy_process_group()


def test_data_parallel_ranks_train_on_different_batches(tmp_path):
    """create_loaders under torch.distributed: the train files are sharded over the ranks (DistributedSampler, reshuffled
    per epoch), and each rank draws its own window starts -- identical batches on every rank would make the gradient
    all-reduce average copies of one gradient (trainer.py has no distributed code to compare with: design of this repo)."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import make_synthetic_robonet as mk
    root = str(tmp_path)
    assert mk.write(root, per_view=4, length=10, seed=1) == 16
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_data_worker, args=(r, 2, port, root, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, e0, s0), (_, e1, s1) = res
    for a, b in zip(e0, e1):  # per epoch: disjoint shards that together cover the 12 training files
        assert not set(a) & set(b) and sorted(a + b) == list(range(12)), (a, b)
    assert e0[0] != e0[1]  # set_epoch: a new shuffle every epoch
    assert s0 != s1


def _torch_adam(p, g, m, v, lr, b1, b2, eps, step):
    """torch.optim.Adam's update on flat slices (the HIP kernel's stand-in on the CPU: this test is about WHICH
    elements each rank updates and what travels, not about the arithmetic, which tests/test_gpu_ops.py pins)."""
    m.mul_(b1).add_(g, alpha=1 - b1)
    v.mul_(b2).addcmul_(g, g, value=1 - b2)
    denom = (v / (1 - b2 ** step)).sqrt_().add_(eps)
    p.addcdiv_(m / (1 - b1 ** step), denom, value=-lr)


class _FlatModel(torch.nn.Module):
    """A parameter holder with SVGConvModel's flat-buffer contract (flat_parameters, parameters as views, _rac_off)."""

    def __init__(self, sizes, total):
        super().__init__()
        self._flat, self._grad = torch.zeros(total), torch.zeros(total)
        off = 0
        for i, n in enumerate(sizes):
            p = torch.nn.Parameter(torch.empty(0))
            p.data = self._flat[off:off + n]
            p.grad = self._grad[off:off + n]
            p._rac_off = off
            self.register_parameter(f"p{i}", p)
            off += (n + 3) // 4 * 4

    def flat_parameters(self):
        return self._flat, self._grad


def _shard_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from robot_aware_control_amd.optim import ShardedAdam, shard_plan
    from robot_aware_control_amd.trainer import ShardReducer
    total, sizes = 4096, [700, 1400, 37, 1000]
    torch.manual_seed(0)
    p0 = torch.randn(total)
    grads = torch.randn(3, world, total)          # three optimiser steps, one gradient per rank and step
    off, inside = 0, torch.zeros(total, dtype=torch.bool)
    for n in sizes:                               # (the alignment gaps between parameters never receive a gradient)
        inside[off:off + n] = True
        off += (n + 3) // 4 * 4
    grads = grads * inside
    # reference: all-reduce mean + the same Adam over the whole buffer on every rank
    ref_p, ref_m, ref_v = p0.clone(), torch.zeros(total), torch.zeros(total)
    for t in range(3):
        _torch_adam(ref_p, grads[t].mean(0), ref_m, ref_v, 1e-2, 0.9, 0.999, 1e-8, t + 1)
    model = _FlatModel(sizes, total)
    model._flat.copy_(p0)
    opt = ShardedAdam(model, lr=1e-2, betas=(0.9, 0.999))
    opt.bucket_elems = 1024                       # four buckets of two 512-element slices
    opt._adam = _torch_adam
    buckets, w, r = opt.plan()
    assert buckets == shard_plan(total, world, 1024) == [(0, 1024), (1024, 1024), (2048, 1024), (3072, 1024)] and (w, r) == (world, rank)
    for t in range(3):
        opt.wait_params()                         # the previous step's parameter all-gather
        model._grad.copy_(grads[t, rank])
        red = ShardReducer(model._grad, buckets, world, rank)
        red.ready(model.p1)                       # covers bucket 1 wholly ([700, 2100) contains [1024, 2048)): issued early
        early = list(red.issued)
        red.finish()
        opt.step()
    opt.wait_params()
    ok = early == [False, True, False, False] and torch.allclose(model._flat, ref_p, atol=1e-6, rtol=1e-6)
    # every rank ends with the same parameters; the optimiser state lives in slices and reassembles to the reference's
    sd = opt.state_dict()
    m_full = torch.zeros(total)
    for i, p in enumerate(model.parameters()):
        m_full[p._rac_off:p._rac_off + p.numel()] = sd["state"][i]["exp_avg"]
    mask = torch.zeros(total, dtype=torch.bool)
    for p in model.parameters():
        mask[p._rac_off:p._rac_off + p.numel()] = True
    ok = ok and torch.allclose(m_full[mask], ref_m[mask], atol=1e-6) and opt._ms.numel() == total // world
    # ... and loads back (a resumed run continues exactly)
    opt2 = ShardedAdam(model, lr=1e-2, betas=(0.9, 0.999))
    opt2.bucket_elems, opt2._adam = 1024, _torch_adam
    opt2.load_state_dict(sd)
    ok = ok and opt2._steps == 3 and torch.equal(opt2._ms, opt._ms) and torch.equal(opt2._vs, opt._vs)
    q.put((rank, bool(ok), model._flat.tolist()))
    dist.destroy_process_group()


def test_sharded_optimizer_equals_allreduce_adam_world2():
    """reduce-scatter -> Adam on 1/world slices -> parameter all-gather (optim.ShardedAdam + trainer.ShardReducer) leaves
    every rank with the parameters of all-reduce(mean) + full Adam, over three steps, with buckets issued out of order."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_shard_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1] for r in res), [r[:2] for r in res]
    assert res[0][2] == res[1][2]


def _ckpt_worker(rank, world, port, q, log_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import argparse

    from robot_aware_control_amd.optim import ShardedAdam
    from robot_aware_control_amd.trainer import PredictionTrainer, ShardReducer
    total, sizes = 4096, [700, 1400, 37, 1000]

    def trainer():  # PredictionTrainer's checkpoint methods over the flat-buffer contract (its constructor needs the GPU)
        tr = PredictionTrainer.__new__(PredictionTrainer)
        tr._config = argparse.Namespace(log_dir=log_dir, experiment="train_robonet")
        tr._device, tr._step = torch.device("cpu"), 0
        tr.model = _FlatModel(sizes, total)
        tr.optimizer = ShardedAdam(tr.model, lr=1e-2, betas=(0.9, 0.999))
        tr.optimizer.bucket_elems, tr.optimizer._adam = 1024, _torch_adam
        return tr

    tr = trainer()
    torch.manual_seed(0)
    tr.model._flat.copy_(torch.randn(total))
    grads = torch.randn(2, world, total)
    for t in range(2):
        tr.optimizer.wait_params()
        tr.model._grad.copy_(grads[t, rank])
        buckets, w, r = tr.optimizer.plan()
        red = ShardReducer(tr.model._grad, buckets, w, r)
        red.finish()
        tr.optimizer.step()
        tr._step += 1
    # the parameter all-gather of the last step is still pending here: the save must wait for it on every rank and
    # every rank must join the moment all-gathers, or rank 0 hangs in them (the path `train()` takes)
    path = tr._save_checkpoint()
    dist.barrier()
    ok = (path is not None) == (rank == 0) and os.path.exists(os.path.join(log_dir, "ckpt_2.pt"))
    tr2 = trainer()
    step = tr2._load_checkpoint(None)
    ok = ok and step == 2 and tr2.optimizer._steps == 2
    inside = torch.zeros(total, dtype=torch.bool)  # (alignment gaps between parameters are not part of a checkpoint)
    for p in tr.model.parameters():
        inside[p._rac_off:p._rac_off + p.numel()] = True
    buckets, w, r = tr.optimizer.plan()
    own = torch.cat([inside[s + r * (n // w):s + (r + 1) * (n // w)] for s, n in buckets])
    ok = ok and torch.equal(tr2.model._flat[inside], tr.model._flat[inside])
    ok = ok and torch.equal(tr2.optimizer._ms[own], tr.optimizer._ms[own])
    ok = ok and torch.equal(tr2.optimizer._vs[own], tr.optimizer._vs[own])
    q.put((rank, bool(ok), tr.model._flat.tolist()))
    dist.destroy_process_group()


def test_sharded_optimizer_checkpoint_through_the_trainer_world2(tmp_path):
    """PredictionTrainer._save_checkpoint / _load_checkpoint with the sharded optimiser: every rank joins the collectives
    that assemble the Adam moments (rank 0 alone would hang in them), rank 0 writes, every rank resumes with its slices."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ckpt_worker, args=(r, world, port, q, str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1] for r in res), [r[:2] for r in res]
    assert res[0][2] == res[1][2]


def test_shard_plan_rejects_world_sizes_that_do_not_divide_the_buffers():
    from robot_aware_control_amd.optim import shard_plan
    with pytest.raises(ValueError, match="world size must divide 256"):
        shard_plan(4096, 3, 1024)
    assert sum(n for _, n in shard_plan(4096, 8, 1000)) == 4096
