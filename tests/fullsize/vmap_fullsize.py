#!/usr/bin/env python3
"""The whole program at BASELINE C2 scale (4096^2 pair, 500x400 = 200k grid points, 32 matcher passes): the unmodified
reference program (oracle/_ref/MIMC3_ref, CPU, all host cores) vs the MIMC3_hip command line vs mimc3_vmap in-process.
Outputs compared byte for byte.  Test infrastructure; run on the GPU box:  gpurun -- python tests/fullsize/vmap_fullsize.py"""
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import fileio  # noqa: E402
from mimc3_amd import synth  # noqa: E402

PROG = os.path.join(ROOT, "oracle", "_ref", "MIMC3_ref")
CLI = os.path.join(ROOT, "mimc3_amd", "csrc", "MIMC3_hip")
OUTS = ("x", "y", "vx", "vy", "ex", "ey", "qual", "flagcp")


def main():
    run_ref = "--no-ref" not in sys.argv
    c = synth.make_case("C2")
    xy = c.xyuvav.copy()
    rng = np.random.default_rng(1)
    slow = rng.random(xy.shape[0]) < 0.05                     # 5 % slow points: control-point candidates
    xy[slow, 4] = rng.uniform(-5, 5, slow.sum()); xy[slow, 5] = rng.uniform(-5, 5, slow.sum())
    t0, t1 = "20200101000000", "20200117000000"
    res = {"workload": "C2 whole program", "H": int(c.i0.shape[0]), "W": int(c.i0.shape[1]), "N": int(xy.shape[0]), "passes": 32,
           "host_cores": os.cpu_count()}
    d = tempfile.mkdtemp(dir="/tmp")
    fileio.write_tiff(f"{d}/{t0}_i0.tif", c.i0.astype(np.uint8)); fileio.write_tiff(f"{d}/{t1}_i1.tif", c.i1.astype(np.uint8))
    fileio.write_gma(f"{d}/xyuvav.GMA", xy)
    args = [f"{d}/{t0}_i0.tif", f"{d}/{t1}_i1.tif", f"{d}/xyuvav.GMA"]
    os.makedirs(f"{d}/hip"); os.makedirs(f"{d}/ref")
    t = time.time()
    p = subprocess.run([CLI] + args + [f"{d}/hip"], env=dict(os.environ, MIMC3_CP_SEED="7"), capture_output=True, text=True)
    if os.environ.get("MIMC3_CLI_TIMING"):
        sys.stderr.write(p.stderr)
    res["cli_wall_s"] = time.time() - t
    res["cli_rc"] = p.returncode
    if p.returncode != 0:
        res["cli_tail"] = (p.stdout + p.stderr)[-1500:]
    print(json.dumps(res), flush=True)
    # in-process: mimc3_vmap only (images already uploaded), second call = warm
    from mimc3_amd import api
    with api.Context(0) as ctx:
        ctx.set_images(c.i0, c.i1)
        for rep in range(2):
            t = time.time(); out = ctx.vmap(xy, 16.0, cp_seed=7); dtv = time.time() - t
            res[f"vmap_call_s_{rep}"] = dtv
    res["offset_cp"] = list(out["offset_cp"]); res["finite_vx"] = float(np.isfinite(out["vx"]).mean())
    print(json.dumps(res), flush=True)
    if run_ref and os.path.exists(PROG):
        t = time.time()
        subprocess.run([PROG] + args + [f"{d}/ref"], check=True, env=dict(os.environ, MIMC3_REF_SEED="7"),
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        res["ref_program_wall_s"] = time.time() - t
        same = {}
        for k in OUTS:
            n = f"vmap_{t0}_{t1}_{k}.GMA"
            same[k] = open(f"{d}/ref/{n}", "rb").read() == open(f"{d}/hip/{n}", "rb").read()
        res["identical_files"] = same
        res["speedup_cli_vs_ref"] = res["ref_program_wall_s"] / res["cli_wall_s"]
    print(json.dumps(res))


if __name__ == "__main__":
    main()
