"""GPU: the struct-level drop-in (include/mimc3_gma_shim.h: the reference's exact function names and
signatures over GMA structs) driven from a plain C program the way the reference's main() drives
the hot path, compared with the oracle."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, assert_bits_equal
from mimc3_amd import synth

CSRC = os.path.join(ROOT, "mimc3_amd", "csrc")


def build_driver(tmp):
    exe = os.path.join(tmp, "shim_driver")
    subprocess.check_call(["make", "-C", CSRC, "-j4"], stdout=subprocess.DEVNULL)
    subprocess.check_call(["gcc", "-O1", "-o", exe, os.path.join(ROOT, "tests", "shim_driver.c"),
                           "-I" + os.path.join(ROOT, "include"), os.path.join(CSRC, "libmimc3_gma_shim.a"),
                           "-L" + CSRC, "-lmimc3_hip", "-lstdc++", "-lm", "-Wl,-rpath," + CSRC])
    return exe


def test_shim_symbols_cpu():
    """CPU: the archive exports the reference names."""
    subprocess.check_call(["make", "-C", CSRC, "-j4"], stdout=subprocess.DEVNULL)
    syms = subprocess.check_output(["nm", "-g", os.path.join(CSRC, "libmimc3_gma_shim.a")]).decode()
    for name in ("get_uv_pivot", "matching_ncc_dlc_2", "get_ruv_neighbor", "get_dpf_pseudosmoothing", "get_offset_image",
                 "calc_mean_var_num_dp_cluster", "get_dpf0", "get_dpf1", "GMA_float_conv2", "mimc2_postprocess"):
        assert f" T {name}" in syms
    for name in ("dimx_vmap", "dimy_vmap", "param_mimc2", "num_dp", "dt"):       # resolved by the reference's main()
        assert f" U {name}" in syms


@pytest.mark.gpu
def test_shim_match(tmp_path, oracle):
    exe = build_driver(str(tmp_path))
    c = synth.make_small(seed=71, shift=(3, -2), angle_deg=25.0, ocw=15, speed=1500.0, h=210, w=230,
                         noise_dn=2, null_frac=0.05, offset=(1, -2))
    H, W = c.i0.shape
    fin, fout = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    with open(fin, "wb") as f:
        f.write(np.array([H, W, c.n, c.ocw, c.offset[0], c.offset[1]], np.int32).tobytes())
        f.write(np.array([c.dt, c.mpp, 1.8, 10.0], np.float32).tobytes())
        f.write(c.i0.tobytes()); f.write(c.i1.tobytes()); f.write(c.xyuvav.tobytes())
    subprocess.check_call([exe, "match", fin, fout], stdout=subprocess.DEVNULL)
    raw = open(fout, "rb").read()
    fwd = np.frombuffer(raw, np.float32, 3 * c.n).reshape(-1, 3)
    swp = np.frombuffer(raw, np.float32, 3 * c.n, offset=12 * c.n).reshape(-1, 3)
    tot = int(np.frombuffer(raw, np.int64, 1, offset=24 * c.n)[0])
    off, uv = oracle.get_uv_pivot(c.xyuvav, c.dt, c.mpp, c.ocw, H, W)
    assert tot == off[-1]
    assert_bits_equal(fwd, oracle.match(c.i0, c.i1, c.xyuvav, c.offset, off, uv, c.ocw), "forward")
    want_sw = oracle.match(c.i1, c.i0, c.xyuvav, -c.offset, off, -uv, c.ocw)
    want_sw[:, :2] = -want_sw[:, :2]
    assert_bits_equal(swp, want_sw, "swapped")


@pytest.mark.gpu
def test_shim_qm(tmp_path, oracle):
    from test_qm_parity import qm_inputs
    exe = build_driver(str(tmp_path))
    dimx, dimy = 44, 38
    xy, mvn, nclus, dpf, dx, dy = qm_inputs(dimx, dimy, 12, -30.0)
    mps = float(np.float32(xy[1, 0] - xy[0, 0]))
    fin, fout = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    with open(fin, "wb") as f:
        f.write(np.array([dimx, dimy, mvn.shape[1]], np.int32).tobytes())
        f.write(np.array([mps, 5.0], np.float32).tobytes())
        f.write(dpf.astype(np.int32).tobytes()); f.write(dx.tobytes()); f.write(dy.tobytes())
        f.write(mvn.tobytes()); f.write(nclus.tobytes()); f.write(xy.tobytes())
    subprocess.check_call([exe, "qm", fin, fout], stdout=subprocess.DEVNULL)
    raw = open(fout, "rb").read()
    n = dimx * dimy
    d = np.frombuffer(raw, np.int32, n).reshape(dimy, dimx)
    x = np.frombuffer(raw, np.float32, n, offset=4 * n).reshape(dimy, dimx)
    y = np.frombuffer(raw, np.float32, n, offset=8 * n).reshape(dimy, dimx)
    nn = int(np.frombuffer(raw, np.int32, 1, offset=12 * n)[0])
    ruv = oracle.get_ruv_neighbor(xy, dimx, dimy, mps, 5.0)
    assert nn == ruv.shape[0]
    wd, wx, wy, _ = oracle.qm(dpf, dx, dy, ruv, mvn, nclus, xy)
    assert np.array_equal(d, wd)
    assert_bits_equal(x, wx); assert_bits_equal(y, wy)


@pytest.mark.gpu
def test_shim_n1_and_conv2(tmp_path, oracle):
    """calc_mean_var_num_dp_cluster -> get_dpf0 -> get_dpf1 and GMA_float_conv2 through the reference's struct-level
    signatures (ragged mvn_dp out and in, in-place planes, in/out filter plane)"""
    exe = build_driver(str(tmp_path))
    dimx, dimy, ndp = 37, 29, 16
    xy = synth.make_grid(dimx, dimy, 60, 60, 20, 20, 1500.0, angle_deg=-20.0)
    mps = float(np.float32(xy[1, 0] - xy[0, 0]))
    dp = synth.synth_candidates(dimx, dimy, seed=21, k=ndp, p_out=0.5)
    dp[:, ::9, 2] = 0.0
    rng = np.random.default_rng(2)
    img = rng.integers(0, 256, size=(40, 53)).astype(np.float32); img[5:9, 7:20] = 0
    ker = rng.normal(size=(3, 3)).astype(np.float32)
    dirty = rng.uniform(-9, 9, img.shape).astype(np.float32)
    fin, fout = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    with open(fin, "wb") as f:
        f.write(np.array([dimx, dimy, ndp, img.shape[0], img.shape[1]], np.int32).tobytes())
        f.write(np.array([mps, 3.0, 16.0, 15.0], np.float32).tobytes())
        f.write(dp.tobytes()); f.write(xy.tobytes())
        f.write(img.tobytes()); f.write(ker.tobytes()); f.write(dirty.tobytes())
    subprocess.check_call([exe, "n1", fin, fout], stdout=subprocess.DEVNULL)
    raw = open(fout, "rb").read()
    n = dimx * dimy
    mvn, nclus = oracle.cluster_candidates(dp, kmax=ndp)
    d0 = oracle.get_dpf0(mvn, nclus, dimx, dimy, 0.6)
    ruv = oracle.get_ruv_neighbor(xy, dimx, dimy, mps, 3.0)
    d1, x1, y1 = oracle.get_dpf1(d0, ruv, mvn, nclus, xy, 16.0, 15.0)
    o = 0
    assert np.array_equal(np.frombuffer(raw, np.int32, n, o).reshape(dimy, dimx), d0); o += 4 * n
    assert np.array_equal(np.frombuffer(raw, np.int32, n, o), nclus); o += 4 * n
    tot = int(nclus.sum())
    ragged = np.frombuffer(raw, np.float32, 5 * tot, o).reshape(tot, 5); o += 20 * tot
    want = np.concatenate([mvn[g, :nclus[g]] for g in range(n)])
    assert_bits_equal(ragged, want, "mvn_dp")
    assert np.array_equal(np.frombuffer(raw, np.int32, n, o).reshape(dimy, dimx), d1); o += 4 * n
    assert_bits_equal(np.frombuffer(raw, np.float32, n, o).reshape(dimy, dimx), x1, "dx"); o += 4 * n
    assert_bits_equal(np.frombuffer(raw, np.float32, n, o).reshape(dimy, dimx), y1, "dy"); o += 4 * n
    assert_bits_equal(np.frombuffer(raw, np.float32, img.size, o).reshape(img.shape), oracle.float_conv2(img, ker, dirty), "conv2")
