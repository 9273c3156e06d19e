"""CPU, world_size 2 over gloo: the two collectives of the hot path -- the mean all-reduce of the flat
gradient (DDP train step) and the all-gather of per-candidate CEM costs -- behave as the N>1 GPU path
expects (same code, RCCL there)."""
import os
import socket

import numpy as np
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
