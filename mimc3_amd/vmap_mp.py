"""The whole data path (mimc3_vmap) sharded over the GPUs of one node: one process per GPU, torch.distributed.
(The native single-process form of the same thing is mimc3_mgpu_vmap, csrc/mgpu.cpp.)

Grid points are independent in the matcher (MIMC_module.c:820-838), so the 32 passes shard with no data-path collective:
cost-balanced shares of ~1k-point blocks (shard.balanced_shares), the image pair replicated.  The CP offset (a few
hundred points) is measured ONCE, on rank 0, and broadcast.  The ONE exchange is the all-gather of the per-rank candidate
blocks [32][per][3] (384 B per grid point; RCCL over xGMI with the "nccl" backend), after which every rank runs the cheap
post-processing (clustering, dpf0/dpf1, QM: 5 ms at 200k points) on the full tensor and holds the full result.
A rank-local failure is agreed on (all-reduce of a status flag) BEFORE the collective, so no rank is left waiting.
"""
import numpy as np

from . import api, shard


def vmap_sharded(ctx, xyuvav, dt, rank, world, device, **kw):
    """Returns the same dict as Context.vmap(); identical on every rank and identical to the single-GPU result."""
    import torch
    import torch.distributed as dist
    xy = np.ascontiguousarray(xyuvav, np.float64)
    n = xy.shape[0]
    H, W = ctx.H, ctx.W
    # ---- CP offset once (rank 0), broadcast: [status, off_u, off_v] + the flag plane
    r, flag = None, None
    err = None
    if rank == 0:
        try:
            r, flag = ctx.vmap_cp(xy, dt, **kw)
        except Exception as e:     # ANY failure must reach the agreement step below, or the peers wait in it for ever
            err = e
    if not shard.all_ok(err is None, device):
        raise err if err is not None else RuntimeError("vmap_sharded: the control-point stage failed on rank 0")
    if world > 1:
        nccl = dist.get_backend() == "nccl"
        head = torch.tensor([r.cp_status, r.offset_cp[0], r.offset_cp[1]] if rank == 0 else [0, 0, 0], dtype=torch.int32,
                            device=device if nccl else "cpu")
        fl = torch.from_numpy(flag if rank == 0 else np.zeros(n, np.uint8)).to(device if nccl else "cpu")
        dist.broadcast(head, 0); dist.broadcast(fl, 0)
        if rank:
            r = ctx.vmap_geometry(xy)
            r.cp_status, r.offset_cp[0], r.offset_cp[1] = (int(v) for v in head.tolist())
            flag = fl.cpu().numpy()
    if r.cp_status < 0:
        return ctx.vmap_finish(xy, dt, 0, r, flag, **kw)
    # ---- cost-balanced shares (every rank computes the same partition from the same pivot counts)
    cost = None
    vec_ocw = kw.get("vec_ocw", (7, 15, 30, 40))
    err = None
    try:
        for ocw in vec_ocw:
            off = api.get_uv_pivot_counts(xy, dt, r.mpp, ocw, H, W, aw_sf=kw.get("aw_sf", 1.8), aw_cre=kw.get("aw_cre", 10.0))
            cost = api.point_cost(off, ocw, cost)
        order, start, per, _ = shard.balanced_shares(cost, world)
        mine = order[start[rank]:start[rank + 1]]
        local = torch.full((32, per, 3), float("nan"), dtype=torch.float32, device=device)     # padded to the largest share
        # the fill above is enqueued on torch's stream, the passes below write the same buffer on the library's own
        # (non-blocking) stream: order them, or a late fill overwrites candidates with NaN
        torch.cuda.current_stream(device).synchronize()
        if len(mine):
            ctx.vmap_passes_points(np.ascontiguousarray(xy[mine]), dt, r, local.data_ptr(), per, **kw)
    except Exception as e:         # (torch OOM, ValueError, ...: not only Mimc3Error -- every rank must reach all_ok)
        err = e
    if not shard.all_ok(err is None, device):
        raise err if err is not None else RuntimeError("vmap_sharded: another rank failed in its matcher passes")
    gathered = shard.all_gather_blocks(local, per, world)                                    # [world][32][per][3]
    full = shard.unpermute(gathered.permute(0, 2, 1, 3), order, start, n).permute(1, 0, 2).contiguous()   # [32][n][3]
    torch.cuda.synchronize(device)
    return ctx.vmap_finish(xy, dt, full.data_ptr(), r, flag, **kw)
