"""Worker of tests/test_vmap_sharded.py: launched by torch.distributed.run, one process per rank, all ranks on GPU 0
(rehearsal of the multi-GPU driver on a one-GPU box; gloo backend).  Rank 0 saves the sharded result."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from mimc3_amd import api, vmap_mp  # noqa: E402
from test_vmap_parity import vmap_case  # noqa: E402


def main():
    out_path = sys.argv[1]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo")
    dev = torch.device("cuda", 0)
    i0, i1, xy = vmap_case(seed=11, shift=(3, -2))
    with api.Context(0) as ctx:
        ctx.set_images(i0, i1)
        got = vmap_mp.vmap_sharded(ctx, xy, 16.0, rank, world, dev, cp_seed=7, num_cp_min=20)
    if rank == 0:
        np.savez(out_path, **{k: got[k] for k in ("vx", "vy", "ex", "ey", "qual", "flag_cp")}, offset=np.array(got["offset_cp"]),
                 subint=np.array(got["cp_subint"], np.float32))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
