"""CPU: the cost-balanced block-cyclic partition of the grid points over the GPUs (mimc3_partition_points, the sharding of
the reference's OpenMP loop MIMC_module.c:816-838): on a skewed a-priori field the work imbalance stays below 10 % where
contiguous blocks are off by 20+ %; every point is owned exactly once; blocks stay whole and in grid order inside a rank."""
import numpy as np
import pytest

from mimc3_amd import api, shard, synth


def skewed_cost(dimx=500, dimy=400, H=4096, W=4096):
    """an outlet glacier in a slow ice sheet: speed (hence pivot count, hence work) varies ~10x across the grid"""
    xy = synth.make_grid(dimx, dimy, 52, 52, 8, 10, 1806.0)
    n = dimx * dimy
    ix = (np.arange(n) % dimx) / dimx
    iy = (np.arange(n) // dimx) / dimy
    scale = 0.2 + 3.0 * np.exp(-((ix - 0.3) ** 2 + (iy - 0.6) ** 2) / 0.02)
    xy[:, 4] *= scale; xy[:, 5] *= scale
    cost = None
    for ocw in (7, 15, 30, 40):
        cost = api.point_cost(api.get_uv_pivot_counts(xy, 16.0, 15.0, ocw, H, W), ocw, cost)
    return cost


@pytest.fixture(scope="module")
def cost():
    return skewed_cost()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_balanced_within_10_percent_on_a_skewed_field(cost, world):
    n = cost.shape[0]
    order, start, per, imb = shard.balanced_shares(cost, world)
    loads = np.array([cost[order[start[r]:start[r + 1]]].sum() for r in range(world)])
    assert abs(loads.max() / loads.mean() - 1 - imb) < 1e-9
    assert imb <= 0.10, imb
    assert np.array_equal(np.sort(order), np.arange(n))                       # a permutation: every point exactly once
    assert per == np.diff(start).max() and start[0] == 0 and start[-1] == n
    contiguous = np.array([cost[slice(*shard.block_range(n, world, r)[:2])].sum() for r in range(world)])
    assert contiguous.max() / contiguous.mean() - 1 > 2 * imb                  # what the plain split would have cost
    block = shard.default_block(n, world)
    for r in range(world):                                                     # whole blocks, ascending inside a rank
        mine = order[start[r]:start[r + 1]]
        assert np.all(np.diff(mine) > 0)
        assert np.all(np.diff(mine)[np.diff(mine) != 1] % 1 == 0)
        assert np.all((mine[np.r_[True, np.diff(mine) != 1]] % block) == 0)


def test_more_ranks_than_blocks_and_uniform_cost():
    order, start, imb = api.partition_points(np.ones(100), 8, 64)              # 2 blocks, 8 ranks: six ranks stay empty
    assert np.array_equal(np.sort(order), np.arange(100)) and (np.diff(start) > 0).sum() == 2
    order, start, imb = api.partition_points(np.ones(8192), 8, 1024)
    assert np.all(np.diff(start) == 1024) and imb == 0.0


def test_cost_model_follows_pivot_count():
    off = np.array([0, 10, 40], np.int64)
    c = api.point_cost(off, 16)
    assert c[1] > 2.5 * c[0] and np.allclose(c, (4 + 6 * np.array([10, 30])) * 33.0 ** 2)
