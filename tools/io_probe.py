import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
from mimc3_amd import api, synth
c = synth.make_case("C2")
H, W = c.i0.shape
with api.Context(0) as ctx:
    ctx.set_images(c.i0, c.i1)
    cor = api.pivot_corridors(c.xyuvav, c.dt, c.mpp)
    pxy = api.pinned_empty(c.xyuvav.shape, np.float64); pxy[:] = c.xyuvav
    pcor = api.pinned_empty(cor.shape, np.uint8); pcor[:] = cor
    pout = api.pinned_empty((c.n, 3), np.float32)
    for i in range(4):
        t = time.perf_counter(); ctx.matching_ncc_dlc_cor(pxy, pcor, c.offset, c.ocw, out=pout); print("call", (time.perf_counter() - t) * 1e3, "ms", file=sys.stderr)
