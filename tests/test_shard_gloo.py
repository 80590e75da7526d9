"""CPU, world_size 2 over gloo: sharding the grid points and re-assembling the field with one
all-gather reproduces the single-process result exactly (the per-shard matcher here is the oracle,
because there is no GPU in this container; the code under test is mimc3_amd/shard.py)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_override, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    from mimc3_amd import shard, synth
    from oracle.oracle import Oracle
    dist.init_process_group("gloo", rank=rank, world_size=world)
    orc = Oracle("port")
    c = synth.make_small(seed=81, shift=(2, -3), angle_deg=-50.0, ocw=8, dimx=7, dimy=9, null_frac=0.05)
    H, W = c.i0.shape
    xy = c.xyuvav[:n_override]
    off, uv = orc.get_uv_pivot(xy, c.dt, c.mpp, c.ocw, H, W)
    n = xy.shape[0]
    lo, hi, per = shard.block_range(n, world, rank)
    sxy, soff, suv = shard.slice_problem(xy, off, uv, lo, hi)
    local = orc.match(c.i0, c.i1, sxy, c.offset, soff, suv, c.ocw, nthreads=1) if hi > lo else np.zeros((0, 3), np.float32)
    full = shard.all_gather_field(torch.from_numpy(local), n, per, world, rank).numpy()
    want = orc.match(c.i0, c.i1, xy, c.offset, off, uv, c.ocw, nthreads=1)
    same = np.array_equal(np.nan_to_num(full, nan=-9).view(np.uint32), np.nan_to_num(want, nan=-9).view(np.uint32))
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, bool(same), int(lo), int(hi)))


@pytest.mark.parametrize("n_points", [63, 62, 1])   # odd split, even split, fewer points than ranks
def test_shard_allgather_world2(n_points):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_points, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _, _ in res), res
    spans = sorted((lo, hi) for _, _, lo, hi in res)
    assert spans[0][0] == 0 and spans[-1][1] == n_points and spans[0][1] == spans[1][0]


def test_block_range_covers_everything():
    sys.path.insert(0, ROOT)
    from mimc3_amd import shard
    for n in (1, 7, 8, 9, 200000, 1000003):
        for world in (1, 2, 3, 8):
            spans = [shard.block_range(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert all(hi - lo <= per for lo, hi, per in spans)


def _worker_balanced(rank, world, port, fail_rank, q):
    """cost-balanced shares + padded all-gather + un-permute + the failure agreement in front of the collective"""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    from mimc3_amd import api, shard, synth
    from oracle.oracle import Oracle
    dist.init_process_group("gloo", rank=rank, world_size=world)
    orc = Oracle("port")
    c = synth.make_small(seed=82, shift=(2, -3), angle_deg=-50.0, ocw=8, dimx=31, dimy=29, h=420, w=440, null_frac=0.05)
    xy = c.xyuvav.copy()
    xy[: xy.shape[0] // 3, 4:6] *= 4.0                      # a fast third: several times the pivots (work) of the rest
    H, W = c.i0.shape
    off, uv = orc.get_uv_pivot(xy, c.dt, c.mpp, c.ocw, H, W)
    n = xy.shape[0]
    cost = api.point_cost(off, c.ocw)
    order, start, per, imb = shard.balanced_shares(cost, world, block=32)
    mine = order[start[rank]:start[rank + 1]]
    ok = shard.all_ok(rank != fail_rank)
    if not ok:
        dist.barrier(); dist.destroy_process_group()
        q.put((rank, "agreed-to-stop", float(imb)))
        return
    sxy, soff, suv = shard.gather_problem(xy, off, uv, mine)
    local = torch.full((per, 3), float("nan"))
    if len(mine):
        local[: len(mine)] = torch.from_numpy(orc.match(c.i0, c.i1, sxy, c.offset, soff, suv, c.ocw, nthreads=1))
    full = shard.unpermute(shard.all_gather_blocks(local, per, world), order, start, n).numpy()
    want = orc.match(c.i0, c.i1, xy, c.offset, off, uv, c.ocw, nthreads=2)
    same = np.array_equal(np.nan_to_num(full, nan=-9).view(np.uint32), np.nan_to_num(want, nan=-9).view(np.uint32))
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, bool(same), float(imb)))


@pytest.mark.parametrize("fail_rank", [-1, 1])
def test_balanced_shares_allgather_world2(fail_rank):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_balanced, args=(r, 2, port, fail_rank, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    if fail_rank < 0:
        assert all(ok is True for _, ok, _ in res), res
        assert all(imb <= 0.10 for _, _, imb in res), res
    else:
        assert all(ok == "agreed-to-stop" for _, ok, _ in res), res   # nobody was left waiting in the all-gather
