#!/usr/bin/env python3
"""Kernel-level bench for the matcher family on the GPU box:  python3 tools/kbench.py [quick|full|<names>] [--check N]

For each named workload on BASELINE C2's pair and grid: average kernel time over a few launches (HIP events through the
context's timing hooks), grid-points/s, and -- with --check N -- bit-identity of an evenly spaced N-point sample against
the compiled reference (oracle/_ref) or the C restatement.  One JSON line per workload.  Test / tuning infrastructure.

workloads:  u8_16 (the headline), u8_7 u8_15 u8_30 u8_40 (CLI chip sizes on the raw pair), ddx_30 ddx_40 (d/dx-filtered
pair: u8-through-offsets), lap_30 lap_40 (Laplacian: u16), f32_16 f32_40 (16-bit-like DN: tiled f32), gen_16 (general)"""
import json
import os
import sys
import time

import numpy as np
import torch   # BEFORE mimc3_amd.api: the process must hold ONE HIP runtime (torch bundles its own libamdhip64; loaded first,
               # it also satisfies libmimc3_hip.so's dependency -- the other order gives two runtimes and "no HIP device")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mimc3_amd import api, shard, synth  # noqa: E402

QUICK = ["u8_16", "u8_40"]
FULL = ["u8_16", "u8_7", "u8_15", "u8_30", "u8_40", "ddx_30", "ddx_40", "lap_30", "lap_40", "f32_16", "f32_40", "gen_16"]


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    check = 0
    if "--check" in sys.argv:
        check = int(sys.argv[sys.argv.index("--check") + 1])
        args = [a for a in args if a != str(check)]
    names = QUICK if (not args or args[0] == "quick") else (FULL if args[0] == "full" else args)
    reps = int(os.environ.get("KBENCH_REPS", "5"))
    c = synth.make_case("C2")
    H, W = c.i0.shape
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream()
    orc = None
    if check:
        from oracle import oracle as o
        orc = o.Oracle("reference" if o.available("reference") else "port")
    ctx = api.Context(0)
    state = {"variant": None}
    cpu_imgs = {}

    def set_variant(v):
        if state["variant"] == v:
            return
        if v in ("raw", "gen"):
            ctx.set_images(c.i0, c.i1)
        elif v == "f32":
            ctx.set_images(c.i0 * 200, c.i1 * 200)
        elif v in ("ddx", "lap"):
            if state["variant"] not in ("raw", "ddx", "lap"):
                ctx.set_images(c.i0, c.i1)
            ctx.filter_images(None)
            ctx.filter_images(api.CLI_KERNELS[0] if v == "ddx" else api.CLI_KERNELS[2])
        ctx.set_path("general" if v == "gen" else "auto")
        state["variant"] = v
        if check:
            cpu_imgs[v] = ctx.get_images(H, W)

    piv_cache = {}
    for name in names:
        kind, ocw = name.split("_")
        ocw = int(ocw)
        set_variant({"u8": "raw"}.get(kind, kind))
        if ocw not in piv_cache:
            off, uv = api.get_uv_pivot(c.xyuvav, c.dt, c.mpp, ocw, H, W)
            piv_cache[ocw] = (off, uv, api.pivot_extent(off, uv), torch.from_numpy(uv).to(dev), torch.from_numpy(off).to(dev))
        off, uv, ext, d_uv, d_off = piv_cache[ocw]
        d_xy = torch.from_numpy(c.xyuvav).to(dev)
        n_run = c.n
        if os.environ.get("KBENCH_N"):        # a small launch (latency view): the first KBENCH_N points only
            n_run = min(c.n, int(os.environ["KBENCH_N"]))
        d_out = torch.empty((c.n, 3), dtype=torch.float32, device=dev)
        times = []
        for r in range(reps + 1):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            ctx.matching_ncc_dlc_2_dev(d_xy.data_ptr(), n_run, c.offset, d_uv.data_ptr(), d_off.data_ptr(), ext, ocw, d_out.data_ptr(),
                                       stream=stream.cuda_stream)
            e1.record(stream)
            torch.cuda.synchronize()
            if r:
                times.append(e0.elapsed_time(e1))
        res = {"name": name, "path": ctx.last_path(), "ms": float(np.mean(times)), "ms_min": float(np.min(times)),
               "Mpts_per_s": n_run / np.mean(times) / 1e3, "points": n_run}
        if check:
            idx = np.unique(np.linspace(0, n_run - 1, check).astype(np.int64))
            sxy, soff, suv = shard.gather_problem(c.xyuvav, off, uv, idx)
            i0, i1 = cpu_imgs[state["variant"]]
            t = time.time()
            want = orc.match(i0, i1, sxy, c.offset, soff, suv, ocw)
            got = d_out.cpu().numpy()[idx]
            same = np.array_equal(np.isnan(got), np.isnan(want)) and np.array_equal(np.nan_to_num(got).view(np.uint32), np.nan_to_num(want).view(np.uint32))
            res["check"] = {"points": int(len(idx)), "bit_identical": bool(same), "cpu_s": time.time() - t,
                            "max_abs": float(np.nanmax(np.abs(got - want))) if not same else 0.0}
        print(json.dumps(res), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
