import os, sys, time, numpy as np, torch
sys.path.insert(0, os.getcwd())
from mimc3_amd import api, synth
c = synth.make_case("C2")
n = c.xyuvav.shape[0]
with api.Context(0) as ctx:
    ctx.set_images(c.i0, c.i1)
    pxy = api.pinned_empty(c.xyuvav.shape, np.float64); pxy[:] = c.xyuvav
    pout = api.pinned_empty((n, 3), np.float32)
    cor = api.pivot_corridors(c.xyuvav, c.dt, c.mpp)
    pcor = api.pinned_empty(cor.shape, np.uint8); pcor[:] = cor
    for rep in range(3):
        for name, fn in (("cor", lambda: ctx.matching_ncc_dlc_cor(pxy, pcor, c.offset, c.ocw, out=pout)),
                         ("geo", lambda: ctx.matching_ncc_dlc_geo(pxy, c.offset, c.dt, c.mpp, c.ocw, out=pout))):
            fn()
            ts = []
            for _ in range(20):
                t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
            print(name, "median %.3f min %.3f mean %.3f max %.3f ms" % (np.median(ts) * 1e3, np.min(ts) * 1e3, np.mean(ts) * 1e3, np.max(ts) * 1e3), flush=True)
    t0 = time.perf_counter()
    for _ in range(10): api.pivot_corridors(c.xyuvav, c.dt, c.mpp)
    print("corridors alone %.3f ms" % ((time.perf_counter() - t0) * 100), flush=True)
