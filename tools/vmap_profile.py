#!/usr/bin/env python3
"""In-process mimc3_vmap at BASELINE scale, for `rocprofv3 --kernel-trace --stats -- python3 tools/vmap_profile.py [reps] [C2|C4]`."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mimc3_amd import api, synth  # noqa: E402

cfg = sys.argv[2] if len(sys.argv) > 2 else "C2"
c = synth.make_case("C4" if cfg == "C4s" else ("C2" if cfg == "C2x16" else cfg))
if cfg == "C2x16":     # 16-bit-like DN (Landsat 8): the f32 kernels
    c = synth.make_case("C2")
    c.i0[:] = c.i0 * 200; c.i1[:] = c.i1 * 200
if cfg == "C4s":      # C4's grid and image size with a 4 px shift: C4's own 12 px shift is beyond the CP stage's +-10 px pivots
    c.i0[:], c.i1[:] = synth.make_pair(c.i0.shape[0], c.i0.shape[1], (4, -4), seed=20260104, noise_dn=2, null_frac=0.02)
xy = c.xyuvav.copy()
rng = np.random.default_rng(1)
slow = rng.random(xy.shape[0]) < 0.05
xy[slow, 4] = rng.uniform(-5, 5, slow.sum()); xy[slow, 5] = rng.uniform(-5, 5, slow.sum())
with api.Context(0) as ctx:
    ctx.set_images(c.i0, c.i1)
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    for _ in range(reps):
        t = time.time(); out = ctx.vmap(xy, 16.0, cp_seed=7); dt = time.time() - t
        print(json.dumps({"vmap_s": dt, "offset": list(out["offset_cp"]), "N": int(xy.shape[0]), "finite": float(np.isfinite(out["vx"]).mean()) if out["vx"] is not None else None}), flush=True)
