"""GPU: the multi-GPU driver of the whole data path (mimc3_amd/vmap_mp.py) rehearsed with 3 ranks sharing GPU 0 (gloo):
the sharded result equals the single-process mimc3_vmap bit for bit.  (CPU part: the block bookkeeping.)"""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, assert_bits_equal
from mimc3_amd import shard


def test_blocks_cover_the_grid():
    for n, world in [(360, 3), (200000, 8), (7, 4), (5, 8)]:
        seen = []
        for r in range(world):
            lo, hi, per = shard.block_range(n, world, r)
            assert hi - lo <= per and 0 <= lo <= hi <= n
            seen += list(range(lo, hi))
        assert seen == list(range(n))


@pytest.mark.gpu
def test_sharded_vmap_equals_single(tmp_path):
    from mimc3_amd import api
    from test_vmap_parity import vmap_case
    out = str(tmp_path / "sharded.npz")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "3", "--master-addr", "127.0.0.1",
                    "--master-port", "29571", os.path.join(ROOT, "tests", "mp_vmap_worker.py"), out], check=True, env=env, timeout=600)
    z = np.load(out)
    i0, i1, xy = vmap_case(seed=11, shift=(3, -2))
    with api.Context(0) as ctx:
        ctx.set_images(i0, i1)
        one = ctx.vmap(xy, 16.0, cp_seed=7, num_cp_min=20)
    assert tuple(z["offset"]) == one["offset_cp"]
    assert np.array_equal(z["flag_cp"], one["flag_cp"])
    for k in ("vx", "vy", "ex", "ey", "qual"):
        assert_bits_equal(z[k], one[k], k)
