"""Grid-point sharding across the GPUs of one node, one process per GPU.

Grid points are independent in the matcher (MIMC_module.c:820-838), so the path shards with no
data-path collective: rank r matches its share of the points against the replicated image pair.
The single exchange step is the re-assembly of the (du, dv, ncc) field on every rank -- the QM
pseudo-smoothing pass needs the whole field -- done with ONE all-gather of equal-sized padded
blocks (RCCL over xGMI when the backend is "nccl"; 12 bytes per point, latency-bound).

Shares are COST-BALANCED: the work of a point grows with its pivot count (a-priori speed), which on
real velocity fields varies several-fold across the grid.  `balanced_shares` cuts the grid into
blocks of ~1k consecutive points and deals them heaviest-first to the least loaded rank
(mimc3_partition_points, the same host code the native multi-GPU driver uses); `block_range` is the
plain contiguous split, kept for callers that need index ranges.
"""
import numpy as np


def block_range(n, world, rank):
    """Contiguous block [lo, hi) of rank `rank`; blocks are ceil(n/world) long, the last may be short/empty."""
    per = -(-n // world)
    lo = min(rank * per, n)
    return lo, min(lo + per, n), per


def default_block(n, world):
    return int(max(256, min(4096, n // (world * 16) + 1)))


def balanced_shares(cost, world, block=None):
    """-> (order int32[N], start int32[world+1], per, imbalance): rank r owns grid points order[start[r]:start[r+1]],
    per = the largest share (the padded block size of the all-gather), imbalance = max load / mean load - 1."""
    from . import api
    n = len(cost)
    order, start, imb = api.partition_points(cost, world, block or default_block(n, world))
    return order, start, int(np.diff(start).max()), imb


def slice_problem(xyuvav, piv_off, piv_uv, lo, hi):
    """Rows [lo,hi) of xyuvav with their CSR pivots re-based to start at 0."""
    off = np.ascontiguousarray(piv_off[lo:hi + 1] - piv_off[lo])
    uv = np.ascontiguousarray(piv_uv[piv_off[lo]:piv_off[hi]])
    return np.ascontiguousarray(xyuvav[lo:hi]), off, uv


def gather_problem(xyuvav, piv_off, piv_uv, idx):
    """The grid points `idx` (any order) with their CSR pivots re-based to start at 0."""
    idx = np.asarray(idx, np.int64)
    cnt = (piv_off[idx + 1] - piv_off[idx]).astype(np.int64)
    off = np.zeros(len(idx) + 1, np.int64)
    np.cumsum(cnt, out=off[1:])
    sel = np.repeat(piv_off[idx] - off[:-1], cnt) + np.arange(off[-1])
    return np.ascontiguousarray(xyuvav[idx]), off, np.ascontiguousarray(piv_uv[sel])


def all_ok(ok, device=None):
    """Agree on success before a collective: a rank-local failure (bounds, zero pivots) must not leave the other ranks
    waiting in the all-gather.  Returns True iff every rank passed True."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return bool(ok)
    t = torch.tensor([0 if ok else 1], dtype=torch.int32, device=device if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return int(t.item()) == 0


def all_gather_blocks(local, per, world):
    """All-gather equal-sized per-rank blocks (torch tensor [per, ...] on the rank's device) -> [world, per, ...]."""
    import torch
    import torch.distributed as dist
    if world == 1:
        return local.unsqueeze(0)
    full = torch.empty((world,) + tuple(local.shape), dtype=local.dtype, device=local.device)
    if dist.get_backend() == "nccl":
        dist.all_gather_into_tensor(full, local.contiguous())
    else:  # gloo (CPU tests, one-GPU rehearsals)
        parts = [torch.empty_like(local, device="cpu") for _ in range(world)]
        dist.all_gather(parts, local.cpu())
        full = torch.stack(parts, 0).to(local.device)
    return full


class Unpermute:
    """Index tensors (built once per partition) that take all-gathered padded blocks back to grid order."""

    def __init__(self, order, start, per, device):
        import torch
        world = len(start) - 1
        src = np.concatenate([r * per + np.arange(int(start[r + 1] - start[r]), dtype=np.int64) for r in range(world)])
        self.src = torch.from_numpy(src).to(device)
        self.dst = torch.from_numpy(np.asarray(order, np.int64)).to(device)
        self.n = len(order)

    def __call__(self, gathered):
        """gathered [world, per, ...] -> [n, ...] in grid order (padding rows dropped)."""
        import torch
        flat = gathered.reshape((gathered.shape[0] * gathered.shape[1],) + tuple(gathered.shape[2:]))
        out = torch.empty((self.n,) + tuple(flat.shape[1:]), dtype=flat.dtype, device=flat.device)
        out[self.dst] = flat[self.src]
        return out


def unpermute(gathered, order, start, n):
    """gathered [world, per, ...] + the partition -> the grid-ordered tensor [n, ...] (padding rows dropped)."""
    assert n == len(order)
    return Unpermute(order, start, gathered.shape[1], gathered.device)(gathered)


def all_gather_field(local, n, per, world, rank):
    """Contiguous-block form: all-gather the per-rank [<=per, 3] float32 results into the full [n, 3] field on every rank."""
    import torch
    pad = torch.full((per, 3), float("nan"), dtype=torch.float32, device=local.device)
    pad[: local.shape[0]] = local
    if world == 1:
        return pad[:n]
    return all_gather_blocks(pad, per, world).reshape(world * per, 3)[:n]
