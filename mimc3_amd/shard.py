"""Grid-point sharding across the GPUs of one node.

Grid points are independent in the matcher (MIMC_module.c:820-838), so the path shards with no
data-path collective: rank r matches a contiguous block of points against the replicated image
pair.  The single exchange step is the re-assembly of the (du, dv, ncc) field on every rank -- the
QM pseudo-smoothing pass needs the whole field -- done with ONE all-gather of equal-sized blocks
(RCCL over xGMI when the backend is "nccl"; 12 bytes per point, latency-bound).
"""
import numpy as np


def block_range(n, world, rank):
    """Contiguous block [lo, hi) of rank `rank`; blocks are ceil(n/world) long, the last may be short/empty."""
    per = -(-n // world)
    lo = min(rank * per, n)
    return lo, min(lo + per, n), per


def slice_problem(xyuvav, piv_off, piv_uv, lo, hi):
    """Rows [lo,hi) of xyuvav with their CSR pivots re-based to start at 0."""
    off = np.ascontiguousarray(piv_off[lo:hi + 1] - piv_off[lo])
    uv = np.ascontiguousarray(piv_uv[piv_off[lo]:piv_off[hi]])
    return np.ascontiguousarray(xyuvav[lo:hi]), off, uv


def all_gather_field(local, n, per, world, rank):
    """All-gather the per-rank [<=per, 3] float32 results into the full [n, 3] field on every rank.

    `local` is a torch tensor on the rank's device.  Blocks are padded to `per` rows so that one
    fixed-size all-gather suffices; padding rows are dropped after the exchange.
    """
    import torch
    import torch.distributed as dist
    pad = torch.full((per, 3), float("nan"), dtype=torch.float32, device=local.device)
    pad[: local.shape[0]] = local
    if world == 1:
        return pad[:n]
    full = torch.empty((world * per, 3), dtype=torch.float32, device=local.device)
    if dist.get_backend() == "nccl":
        dist.all_gather_into_tensor(full, pad)
    else:  # gloo (CPU tests)
        parts = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(parts, pad)
        full = torch.cat(parts, 0)
    return full[:n]
