"""CPU: the chain of oracle functions in tests/pipeline_oracle.py (what the GPU box uses as the whole-program checker)
reproduces (a) the reference-program golden fixture and (b), where oracle/_ref/MIMC3_ref is present, a live run of
the unmodified reference program on fresh TIFF/.GMA files -- bit for bit, file formats included."""
import os
import subprocess

import numpy as np
import pytest

import fileio
from conftest import ROOT, assert_bits_equal, golden_files
from mimc3_amd import synth
from pipeline_oracle import oracle_vmap

PROG = os.path.join(ROOT, "oracle", "_ref", "MIMC3_ref")


def test_chain_vs_program_golden(oracle):
    z = np.load(golden_files("vmap_small")[0])
    got = oracle_vmap(oracle, z["i0"].astype(np.float32), z["i1"].astype(np.float32), z["xyuvav"], 16.0, cp_seed=int(z["seed"]))
    meta = dict(zip(z["meta_keys"].tolist(), z["meta_vals"].tolist()))
    assert got["cp_status"] == 1
    assert (str(got["offset_cp"][0]), str(got["offset_cp"][1])) == (meta["cp_offset_int_u"], meta["cp_offset_int_v"])
    assert ("%f" % got["cp_subint"][0], "%f" % got["cp_subint"][1]) == (meta["cp_offset_subint_u"], meta["cp_offset_subint_v"])
    for k in ("vx", "vy", "ex", "ey", "qual"):
        assert_bits_equal(got[k], z["out_" + k], k)
    assert np.array_equal(got["flag_cp"].reshape(z["out_flagcp"].shape), z["out_flagcp"])
    assert np.array_equal(z["out_x"][0], z["xyuvav"][:got["dimx"], 0])
    assert np.array_equal(z["out_y"][0], z["xyuvav"][::got["dimx"], 1])


@pytest.mark.skipif(not os.path.exists(PROG), reason="oracle/_ref/MIMC3_ref not built (needs /root/reference)")
def test_chain_vs_program_live(oracle, tmp_path):
    h, w, dimx, dimy = 380, 440, 15, 11
    i0, i1 = synth.make_pair(h, w, (-2, 3), seed=92, null_frac=0.05, noise_dn=4, bits=16)
    xy = synth.make_grid(dimx, dimy, 70, 70, (w - 140) // dimx, (h - 140) // dimy, 1200.0, angle_deg=-50.0)
    rng = np.random.default_rng(92)
    slow = rng.random(dimx * dimy) < 0.6
    xy[slow, 4] = rng.uniform(-5, 5, slow.sum()); xy[slow, 5] = rng.uniform(-5, 5, slow.sum())
    t0, t1 = "20210301000000", "20210309120000"                    # dt = 8.5 days
    d = str(tmp_path)
    fileio.write_tiff(f"{d}/{t0}_i0.tif", i0.astype(np.uint16)); fileio.write_tiff(f"{d}/{t1}_i1.tif", i1.astype(np.uint16))
    fileio.write_gma(f"{d}/xyuvav.GMA", xy)
    os.makedirs(f"{d}/out")
    subprocess.run([PROG, f"{d}/{t0}_i0.tif", f"{d}/{t1}_i1.tif", f"{d}/xyuvav.GMA", f"{d}/out"], check=True,
                   env=dict(os.environ, MIMC3_REF_SEED="5"), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    r = fileio.read_vmap(f"{d}/out", t0, t1)
    got = oracle_vmap(oracle, i0, i1, xy, 8.5, cp_seed=5)
    for k in ("vx", "vy", "ex", "ey", "qual"):
        assert_bits_equal(got[k], r[k], k)
    assert np.array_equal(got["flag_cp"].reshape(r["flagcp"].shape), r["flagcp"])
    assert r["meta"]["cp_offset_int_u"] == str(got["offset_cp"][0]) and r["meta"]["cp_offset_subint_v"] == "%f" % got["cp_subint"][1]
