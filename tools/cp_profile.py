#!/usr/bin/env python3
"""The control-point stage alone at BASELINE C2 scale, for `rocprofv3 --kernel-trace --stats -- python3 tools/cp_profile.py [reps]`
(MIMC3_CP_TIMING=1 prints the wall time of each step)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mimc3_amd import api, synth  # noqa: E402

c = synth.make_case("C2")
xy = c.xyuvav.copy()
rng = np.random.default_rng(1)
slow = rng.random(xy.shape[0]) < 0.05
xy[slow, 4] = rng.uniform(-5, 5, slow.sum()); xy[slow, 5] = rng.uniform(-5, 5, slow.sum())
npeer = int(os.environ.get("CP_PEERS", "0"))          # further contexts on the same device sharing every segment (the multi-GPU form)
peers = [api.Context(0) for _ in range(npeer)]
with api.Context(0) as ctx:
    for q in [ctx] + peers:
        q.set_images(c.i0, c.i1)
    for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
        t = time.time(); st, off, flag, info, sduv = ctx.get_offset_image(xy, api.CLI_KERNELS, seed=7, peers=peers); dt = time.time() - t
        print(json.dumps({"cp_s": dt, "peers": npeer, "status": st, "offset": off.tolist(), "info": info.tolist(), "sduv": sduv.tolist()}), flush=True)
for q in peers:
    q.close()
