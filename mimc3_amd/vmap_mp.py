"""The whole data path (mimc3_vmap) sharded over the GPUs of one node: one process per GPU, torch.distributed.

Grid points are independent in the matcher (MIMC_module.c:820-838), so the 32 passes shard by contiguous blocks of
grid points with no data-path collective; the image pair is replicated, the CP offset (a few hundred points) is
measured redundantly on every rank with the same seed.  The ONE exchange is the all-gather of the per-rank candidate
blocks [32][n_r][3] (384 B per grid point; RCCL over xGMI with the "nccl" backend), after which every rank runs the
cheap post-processing (clustering, dpf0/dpf1, QM: 5 ms at 200k points) on the full tensor and holds the full result.
"""
import numpy as np

from . import shard


def vmap_sharded(ctx, xyuvav, dt, rank, world, device, **kw):
    """Returns the same dict as Context.vmap(); identical on every rank and identical to the single-GPU result.
    `cp_seed` must be given (>= 0) when world > 1 so that all ranks shuffle the control-point candidates alike."""
    import torch
    import torch.distributed as dist
    if world > 1 and kw.get("cp_seed", -1) < 0:
        raise ValueError("vmap_sharded: pass cp_seed >= 0 (every rank must measure the same CP offset)")
    xy = np.ascontiguousarray(xyuvav, np.float64)
    n = xy.shape[0]
    lo, hi, per = shard.block_range(n, world, rank)
    block = torch.full((32, per, 3), float("nan"), dtype=torch.float32, device=device)   # padded to equal size
    local = torch.empty((32, max(hi - lo, 1), 3), dtype=torch.float32, device=device)
    r, flag = ctx.vmap_passes(xy, dt, lo, hi, local.data_ptr(), **kw)
    if r.cp_status < 0:
        return ctx.vmap_finish(xy, dt, 0, r, flag, **kw)
    if hi > lo:
        block[:, : hi - lo] = local[:, : hi - lo]
    if world == 1:
        full = block[:, :n].contiguous()
    else:
        gathered = torch.empty((world, 32, per, 3), dtype=torch.float32, device=device)
        if dist.get_backend() == "nccl":
            dist.all_gather_into_tensor(gathered, block)
        else:                                   # gloo: CPU tensors (tests, rehearsals)
            parts = [torch.empty_like(block, device="cpu") for _ in range(world)]
            dist.all_gather(parts, block.cpu())
            gathered = torch.stack(parts, 0).to(device)
        full = gathered.permute(1, 0, 2, 3).reshape(32, world * per, 3)[:, :n].contiguous()
    torch.cuda.synchronize(device)
    return ctx.vmap_finish(xy, dt, full.data_ptr(), r, flag, **kw)
