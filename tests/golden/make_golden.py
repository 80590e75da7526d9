#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the COMPILED REFERENCE (oracle/_ref/libmimc3_ref.so).

Run in the build container only (it needs /root/reference to have been compiled by
`make -C oracle ref`):   python tests/golden/make_golden.py

Each fixture is data only: seeded synthetic inputs (stored as uint8/uint16 DN to stay small) and
the reference's outputs for them (pivots, out[N,3]; QM planes before/after).  The reference's
matcher output is deterministic; the harness zero-fills malloc (T4).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mimc3_amd import synth  # noqa: E402
from oracle.oracle import Oracle  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))

# name -> make_small kwargs (+ optional post-edit hook)
MATCH_CASES = {
    # integer shift on the corridor, |cos|>|sin| branch, quadrant (+u, -v)
    "int_shift_q1": dict(seed=101, shift=(3, -2), angle_deg=30.0, ocw=7, speed=1200.0),
    # |sin|>|cos| branch, quadrant (-u, +v), sub-pixel bilinear shift, DN noise
    "subpix_q3": dict(seed=102, shift=(-2, 3), angle_deg=-120.0, ocw=10, speed=1500.0,
                      subpixel=(0.35, 0.6), noise_dn=2),
    # quadrant (+u, +v) (south-going: pivots (k, k-1) quirk T5), null blobs -> exclusion + some -3
    "nulls_q4": dict(seed=103, shift=(2, 2), angle_deg=-45.0, ocw=8, speed=1300.0, null_frac=0.12),
    # quadrant (-u, -v), shift 2 px OFF the corridor (T3: -2.0 cells enter the fit)
    "offcorridor_q2": dict(seed=104, shift=(-4, -1), angle_deg=135.0, ocw=7, speed=1400.0),
    # 16-bit DN (f32 products round, T1), non-zero CP offset applied to the window only (T6)
    "u16_offset": dict(seed=105, shift=(5, -3), angle_deg=20.0, ocw=9, speed=1600.0, bits=16, offset=(2, -1)),
    # T3, observable laziness: horizontal corridor (window only ocw+2 rows tall each side), true shift 2 px
    # off it: the climb leaves through the boundary break (:703-707) and the fit reads -2.0 cells
    "t3_break_horizontal": dict(seed=107, shift=(3, 2), angle_deg=0.0, ocw=7, speed=1200.0),
    "t3_break_vertical": dict(seed=109, shift=(-2, 3), angle_deg=90.0, ocw=7, speed=1300.0),
    # grid so close to the border that windows hang over the image edge (zero fill, :877-884)
    "edge_windows": dict(seed=106, shift=(1, -1), angle_deg=60.0, ocw=7, speed=900.0, margin=10,
                         h=120, w=128, dimx=10, dimy=9),
}


def edit_case(name, c):
    if name == "nulls_q4":
        # a fully null block: chips inside it are >80 % null -> (NaN, NaN, -3)
        c.i0[20:70, 20:80] = 0.0
        c.i1[90:140, 100:170] = 0.0
    return c


def main():
    ref = Oracle("reference")
    for name, kw in MATCH_CASES.items():
        c = edit_case(name, synth.make_small(**kw))
        H, W = c.i0.shape
        off, uv = ref.get_uv_pivot(c.xyuvav, c.dt, c.mpp, c.ocw, H, W)
        out = ref.match(c.i0, c.i1, c.xyuvav, c.offset, off, uv, c.ocw)
        # swapped pass exactly as the CLI does it (MIMC_main.c:272-293): images exchanged,
        # offset negated, pivots negated; (du,dv) negation is the caller's business
        out_sw = ref.match(c.i1, c.i0, c.xyuvav, -c.offset, off, -uv, c.ocw)
        bits = kw.get("bits", 8)
        dn = np.uint8 if bits == 8 else np.uint16
        assert np.array_equal(c.i0, c.i0.astype(dn).astype(np.float32))
        np.savez_compressed(
            os.path.join(OUT, f"match_{name}.npz"),
            i0=c.i0.astype(dn), i1=c.i1.astype(dn), xyuvav=c.xyuvav, offset=c.offset,
            ocw=np.int32(c.ocw), dt=np.float32(c.dt), mpp=np.float32(c.mpp),
            piv_off=off, piv_uv=uv, out=out, out_swapped=out_sw)
        inv = int((out[:, 2] == -3).sum())
        print(f"match_{name}: N={c.n} npiv={int(off[-1])} invalid={inv} "
              f"median=({np.nanmedian(out[:, 0]):.3f},{np.nanmedian(out[:, 1]):.3f}) shift={c.shift}")

    # QM: candidates -> (reference clustering, dpf0, dpf1) -> pseudo-smoothing before/after
    for name, (dimx, dimy, seed, ang) in {"qm_40x40": (40, 40, 5, 37.0), "qm_57x33": (57, 33, 8, -70.0)}.items():
        xy = synth.make_grid(dimx, dimy, 60, 60, 20, 20, 1806.0, angle_deg=ang)
        mps = float(np.float32(xy[1, 0] - xy[0, 0]))
        dp = synth.synth_candidates(dimx, dimy, seed=seed)
        mvn, nclus, dpf, dx, dy = ref.postprocess_prep(dp, xy, dimx, dimy, 16.0, 15.0, mps)
        ruv = ref.get_ruv_neighbor(xy, dimx, dimy, mps, 5.0)
        d2, x2, y2, _ = ref.qm(dpf, dx, dy, ruv, mvn, nclus, xy)
        kmax = int(nclus.max())
        np.savez_compressed(
            os.path.join(OUT, f"{name}.npz"),
            xyuvav=xy, meter_per_spacing=np.float32(mps), radius=np.float32(5.0), ruv=ruv,
            mvn=mvn[:, :kmax].copy(), nclus=nclus, dpf_in=dpf, dx_in=dx, dy_in=dy,
            dpf_out=d2, dx_out=x2, dy_out=y2)
        print(f"{name}: nn={ruv.shape[0]} kmax={kmax} changed={(d2 != dpf).sum()}")

    # N1: candidates -> clustering -> dpf0 -> dpf1, each stage's reference output kept
    for name, (dimx, dimy, seed, k, pout) in {"n1_40x40_k8": (40, 40, 5, 8, 0.45), "n1_61x47_k12": (61, 47, 9, 12, 0.5)}.items():
        xy = synth.make_grid(dimx, dimy, 60, 60, 20, 20, 1806.0, angle_deg=37.0)
        mps = float(np.float32(xy[1, 0] - xy[0, 0]))
        dp = synth.synth_candidates(dimx, dimy, seed=seed, k=k, p_out=pout)
        dp[:, ::7, 2] = 0.05          # below the ncc>0.1 gate (:1010)
        dp[:3, 5::11, 2] = -3.0       # invalid passes
        mvn, nclus = ref.cluster_candidates(dp)
        kmax = int(nclus.max())
        mvn = mvn[:, :kmax].copy()
        dpf0 = ref.get_dpf0(mvn, nclus, dimx, dimy, 0.6)
        ruv = ref.get_ruv_neighbor(xy, dimx, dimy, mps, 3.0)
        d1, x1, y1 = ref.get_dpf1(dpf0, ruv, mvn, nclus, xy, 16.0, 15.0)
        np.savez_compressed(
            os.path.join(OUT, f"{name}.npz"),
            dp=dp, xyuvav=xy,
            meter_per_spacing=np.float32(mps), radius=np.float32(3.0), dt=np.float32(16.0), mpp=np.float32(15.0),
            ruv=ruv, mvn=mvn, nclus=nclus, dpf0=dpf0, dpf1=d1, dx1=x1, dy1=y1)
        print(f"{name}: kmax={kmax} empty={(nclus == 0).sum()} dpf0<0={(dpf0 < 0).sum()} dpf1<0={(d1 < 0).sum()}")

    # N2: the CLI's three pre-filter kernels on an 8-bit image with nulls
    c = edit_case("nulls_q4", synth.make_small(**MATCH_CASES["nulls_q4"]))
    ks = {"ddx": np.array([[-1, 0, 1]], np.float32), "ddy": np.array([[-1], [0], [1]], np.float32),
          "laplacian": np.array([[-1 / 8] * 3, [-1 / 8, 1, -1 / 8], [-1 / 8] * 3], np.float32)}
    d = {"img": c.i0.astype(np.uint8)}
    for n, k in ks.items():
        d["k_" + n] = k
        d["out_" + n] = ref.float_conv2(c.i0, k)
    np.savez_compressed(os.path.join(OUT, "conv2_nulls.npz"), **d)
    print("conv2_nulls:", {n: (float(d["out_" + n].min()), float(d["out_" + n].max())) for n in ks})

    # N4: control-point offset on a small pair (dense grid so that there are enough slow candidates), shuffle seed pinned
    h, w, dimx, dimy = 260, 300, 14, 11
    i0, i1 = synth.make_pair(h, w, (2, -3), seed=77, null_frac=0.04, noise_dn=3)
    xy = synth.make_grid(dimx, dimy, 50, 50, 15, 15, 1806.0, angle_deg=30.0)
    rng = np.random.default_rng(77)
    slow = rng.random(dimx * dimy) < 0.8
    xy[slow, 4] = rng.uniform(-5, 5, slow.sum()); xy[slow, 5] = rng.uniform(-5, 5, slow.sum())
    par = dict(vec_ocw=(7, 15, 30, 40), aw_cre=10.0, num_cp_max=500, num_cp_min=10, ratio_cp=0.1, thres_spd_cp=10.0)
    d = dict(i0=i0.astype(np.uint8), i1=i1.astype(np.uint8), xyuvav=xy, seeds=np.array([5, 1234], np.int64),
             **{k: np.array(v) for k, v in par.items()})
    for sd in (5, 1234):
        rc, off, flag, _, _ = ref.get_offset_image(i0, i1, xy, list(ks.values()), sd, **par)
        d[f"rc_{sd}"] = np.int32(rc); d[f"offset_{sd}"] = off; d[f"flag_{sd}"] = flag
        print(f"cp_small seed {sd}: rc={rc} offset={off.tolist()} cps={int(flag.sum())}")
    np.savez_compressed(os.path.join(OUT, "cp_small.npz"), **d)

    # N3: the whole reference PROGRAM (oracle/_ref/MIMC3_ref = unmodified main() + zero-filling malloc + pinned shuffle
    # seed) on a small TIFF pair: its eight .GMA outputs and meta.txt
    import subprocess
    import tempfile
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import fileio
    prog = os.path.join(ROOT, "oracle", "_ref", "MIMC3_ref")
    h, w, dimx, dimy = 400, 420, 14, 12
    i0, i1 = synth.make_pair(h, w, (2, -1), seed=91, null_frac=0.02, noise_dn=2)
    xy = synth.make_grid(dimx, dimy, 70, 70, (w - 140) // dimx, (h - 140) // dimy, 900.0, angle_deg=30.0)
    rng = np.random.default_rng(91)
    slow = rng.random(dimx * dimy) < 0.6
    xy[slow, 4] = rng.uniform(-5, 5, slow.sum()); xy[slow, 5] = rng.uniform(-5, 5, slow.sum())
    t0, t1, seed = "20200101103000", "20200117103000", 7
    with tempfile.TemporaryDirectory() as d:
        fileio.write_tiff(f"{d}/{t0}_i0.tif", i0.astype(np.uint8)); fileio.write_tiff(f"{d}/{t1}_i1.tif", i1.astype(np.uint8))
        fileio.write_gma(f"{d}/xyuvav.GMA", xy)
        os.makedirs(f"{d}/out")
        subprocess.run([prog, f"{d}/{t0}_i0.tif", f"{d}/{t1}_i1.tif", f"{d}/xyuvav.GMA", f"{d}/out"], check=True,
                       env=dict(os.environ, MIMC3_REF_SEED=str(seed)), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        r = fileio.read_vmap(f"{d}/out", t0, t1)
    meta = {k: v for k, v in r.pop("meta").items() if not k.startswith("name_")}
    np.savez_compressed(os.path.join(OUT, "vmap_small.npz"), i0=i0.astype(np.uint8), i1=i1.astype(np.uint8), xyuvav=xy,
                        t0=t0, t1=t1, seed=np.int64(seed), meta_keys=np.array(list(meta)), meta_vals=np.array(list(meta.values())),
                        **{"out_" + k: v for k, v in r.items()})
    print("vmap_small:", meta, "finite vx", float(np.isfinite(r["vx"]).mean()))


if __name__ == "__main__":
    main()
