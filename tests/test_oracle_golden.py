"""CPU: the restatement (oracle/mimc3_oracle.c) reproduces the reference-generated golden vectors
bit-for-bit -- this is what pins the oracle."""
import numpy as np
import pytest

from conftest import assert_bits_equal, golden_files, load_match_golden


@pytest.mark.parametrize("path", golden_files("match_"), ids=lambda p: p.split("match_")[-1][:-4])
def test_match_golden(oracle, path):
    g = load_match_golden(path)
    H, W = g["i0"].shape
    off, uv = oracle.get_uv_pivot(g["xyuvav"], g["dt"], g["mpp"], g["ocw"], H, W)
    assert np.array_equal(off, g["piv_off"]) and np.array_equal(uv, g["piv_uv"])
    out = oracle.match(g["i0"], g["i1"], g["xyuvav"], g["offset"], off, uv, g["ocw"])
    assert_bits_equal(out, g["out"], "forward")
    out_sw = oracle.match(g["i1"], g["i0"], g["xyuvav"], -g["offset"], off, -uv, g["ocw"])
    assert_bits_equal(out_sw, g["out_swapped"], "swapped")


@pytest.mark.parametrize("path", golden_files("qm_"), ids=lambda p: p.split("/")[-1][:-4])
def test_qm_golden(oracle, path):
    z = np.load(path)
    dimy, dimx = z["dpf_in"].shape
    ruv = oracle.get_ruv_neighbor(z["xyuvav"], dimx, dimy, float(z["meter_per_spacing"]), float(z["radius"]))
    assert np.array_equal(ruv, z["ruv"])
    d, x, y, stats = oracle.qm(z["dpf_in"], z["dx_in"], z["dy_in"], ruv, z["mvn"], z["nclus"], z["xyuvav"])
    assert stats[2] == 0, "fixture must not hit the T7 (NaN-fit) definition"
    assert np.array_equal(d, z["dpf_out"])
    assert_bits_equal(x, z["dx_out"], "dx")
    assert_bits_equal(y, z["dy_out"], "dy")
    assert (d != z["dpf_in"]).sum() > 0


@pytest.mark.parametrize("path", golden_files("n1_"), ids=lambda p: p.split("/")[-1][:-4])
def test_n1_golden(oracle, path):
    """N1 (clustering -> dpf0 -> dpf1) stage by stage against the reference's outputs."""
    z = np.load(path)
    dimy, dimx = z["dpf0"].shape
    kmax = z["mvn"].shape[1]
    mvn, nclus = oracle.cluster_candidates(z["dp"], kmax)
    assert np.array_equal(nclus, z["nclus"])
    assert_bits_equal(mvn, z["mvn"], "mvn")
    dpf0 = oracle.get_dpf0(z["mvn"], z["nclus"], dimx, dimy, 0.6)
    assert np.array_equal(dpf0, z["dpf0"])
    ruv = oracle.get_ruv_neighbor(z["xyuvav"], dimx, dimy, float(z["meter_per_spacing"]), float(z["radius"]))
    assert np.array_equal(ruv, z["ruv"])
    d, x, y = oracle.get_dpf1(z["dpf0"], ruv, z["mvn"], z["nclus"], z["xyuvav"], float(z["dt"]), float(z["mpp"]))
    assert np.array_equal(d, z["dpf1"])
    assert_bits_equal(x, z["dx1"], "dx"); assert_bits_equal(y, z["dy1"], "dy")
    assert (z["dpf0"] < 0).sum() > (d < 0).sum() > 0


@pytest.mark.parametrize("path", golden_files("conv2_"), ids=lambda p: p.split("/")[-1][:-4])
def test_conv2_golden(oracle, path):
    z = np.load(path)
    img = z["img"].astype(np.float32)
    for name in ("ddx", "ddy", "laplacian"):
        assert_bits_equal(oracle.float_conv2(img, z["k_" + name]), z["out_" + name], name)


def _cp_golden():
    z = np.load(golden_files("cp_small")[0])
    par = dict(vec_ocw=tuple(int(v) for v in z["vec_ocw"]), aw_cre=float(z["aw_cre"]), num_cp_max=int(z["num_cp_max"]),
               num_cp_min=int(z["num_cp_min"]), ratio_cp=float(z["ratio_cp"]), thres_spd_cp=float(z["thres_spd_cp"]))
    ks = [np.array([[-1, 0, 1]], np.float32), np.array([[-1], [0], [1]], np.float32),
          np.array([[-1 / 8] * 3, [-1 / 8, 1, -1 / 8], [-1 / 8] * 3], np.float32)]
    return z, par, ks


def test_cp_offset_golden(oracle):
    """N4: get_offset_image with the shuffle seed pinned -- status, offset and the control-point flags"""
    z, par, ks = _cp_golden()
    i0, i1 = z["i0"].astype(np.float32), z["i1"].astype(np.float32)
    for sd in z["seeds"]:
        rc, off, flag, info, _ = oracle.get_offset_image(i0, i1, z["xyuvav"], ks, int(sd), **par)
        assert rc == int(z[f"rc_{sd}"]) and np.array_equal(off, z[f"offset_{sd}"]) and np.array_equal(flag, z[f"flag_{sd}"])
        assert info[3] == flag.sum() > 0
