#!/usr/bin/env python3
"""N1 at BASELINE C5 scale (500x400 = 200k grid points, 32 passes): HIP vs oracle, and device times.
Test infrastructure (imports oracle/): run on the GPU box, e.g.  gpurun -- python tests/fullsize/n1_fullsize.py"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mimc3_amd import api, synth  # noqa: E402
from oracle.oracle import Oracle  # noqa: E402


def main():
    dimx, dimy, k = 500, 400, 32
    n = dimx * dimy
    xy = synth.make_grid(dimx, dimy, 60, 60, 20, 20, 1806.0, angle_deg=37.0)
    mps = float(np.float32(xy[1, 0] - xy[0, 0]))
    dp = synth.synth_candidates(dimx, dimy, seed=20260105, k=k, p_out=0.45)
    orc = Oracle("port")
    t = time.time(); rm, rn = orc.cluster_candidates(dp, kmax=k); t_clu = time.time() - t
    t = time.time(); r0 = orc.get_dpf0(rm, rn, dimx, dimy, 0.6); t_d0 = time.time() - t
    ruv = api.get_ruv_neighbor(xy, dimx, dimy, mps, 3.0)
    t = time.time(); rd, rx, ry = orc.get_dpf1(r0, ruv, rm, rn, xy, 16.0, 15.0); t_d1 = time.time() - t
    res = {"N": n, "ndp": k, "nn": int(ruv.shape[0]), "cpu_port_s": {"cluster": t_clu, "dpf0": t_d0, "dpf1": t_d1},
           "unassigned_dpf0": int((r0 < 0).sum())}
    dev = torch.device("cuda:0")
    with api.Context(0) as ctx:
        m, nc = ctx.calc_mean_var_num_dp_cluster(dp)
        d0 = ctx.get_dpf0(m, nc, dimx, dimy, 0.6)
        d1, x1, y1, sweeps = ctx.get_dpf1(d0, ruv, m, nc, xy, 16.0, 15.0)
        same = lambda a, b: bool(np.array_equal(np.nan_to_num(a, nan=-777.0), np.nan_to_num(b, nan=-777.0)))
        res["identical"] = {"nclus": same(nc, rn), "mvn": same(m, rm), "dpf0": same(d0, r0), "dpf1": same(d1, rd),
                            "dx": same(x1, rx), "dy": same(y1, ry)}
        res["sweeps"] = sweeps
        # device-resident timings
        d_dp = torch.from_numpy(dp).to(dev)
        d_m = torch.empty((n, k, 5), dtype=torch.float32, device=dev)
        d_nc = torch.empty(n, dtype=torch.int32, device=dev)
        d_seen = torch.zeros(1, dtype=torch.int32, device=dev)
        d_d0 = torch.empty(n, dtype=torch.int32, device=dev)
        d_dpf = torch.empty(n, dtype=torch.int32, device=dev)
        d_x = torch.empty(n, dtype=torch.float32, device=dev)
        d_y = torch.empty(n, dtype=torch.float32, device=dev)
        d_ruv = torch.from_numpy(ruv).to(dev)
        d_xy = torch.from_numpy(xy).to(dev)
        d_work = torch.empty(ctx.dpf1_workspace_bytes(n), dtype=torch.uint8, device=dev)
        s = torch.cuda.current_stream().cuda_stream
        def timed(fn, reps=5):
            fn(); torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record(); torch.cuda.synchronize()
            return e0.elapsed_time(e1) / reps
        res["gpu_ms"] = {
            "cluster": timed(lambda: ctx.calc_mean_var_num_dp_cluster_dev(d_dp.data_ptr(), k, n, k, d_m.data_ptr(), d_nc.data_ptr(), d_seen.data_ptr(), s)),
            "dpf0": timed(lambda: ctx.get_dpf0_dev(d_m.data_ptr(), d_nc.data_ptr(), n, k, 0.6, d_d0.data_ptr(), s)),
        }
        def run_d1():
            d_dpf.copy_(d_d0)
            return ctx.get_dpf1_dev(dimy, dimx, d_dpf.data_ptr(), d_x.data_ptr(), d_y.data_ptr(), d_ruv.data_ptr(), int(ruv.shape[0]),
                                    d_m.data_ptr(), k, d_nc.data_ptr(), d_xy.data_ptr(), 16.0, 15.0, d_work.data_ptr(), s)
        res["gpu_ms"]["dpf1"] = timed(run_d1, reps=3)
        res["identical"]["dev_dpf1"] = same(d_dpf.cpu().numpy().reshape(dimy, dimx), rd)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
