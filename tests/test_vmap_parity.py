"""GPU: mimc3_vmap (MIMC_main.c:203-402 on arrays: CP offset, 32 matcher passes on the raw and the three filtered
pairs, clustering, dpf0, dpf1, QM, unit conversion) against the same chain built from the oracle's functions.
Bar: bit-identical output planes, flags and scalars."""
import numpy as np
import pytest

from conftest import assert_bits_equal
from mimc3_amd import synth
from pipeline_oracle import oracle_vmap

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def api():
    from mimc3_amd import api as a
    return a


def vmap_case(seed, shift, h=560, w=600, dimx=20, dimy=18, null_frac=0.02, noise=2, speed=900.0):
    i0, i1 = synth.make_pair(h, w, shift, seed=seed, null_frac=null_frac, noise_dn=noise)
    xy = synth.make_grid(dimx, dimy, 70, 70, (w - 140) // dimx, (h - 140) // dimy, speed, angle_deg=30.0)
    rng = np.random.default_rng(seed)
    slow = rng.random(dimx * dimy) < 0.5
    xy[slow, 4] = rng.uniform(-5, 5, slow.sum()); xy[slow, 5] = rng.uniform(-5, 5, slow.sum())
    return i0, i1, xy


@pytest.mark.parametrize("case", [dict(seed=11, shift=(3, -2)), dict(seed=12, shift=(-2, 4), null_frac=0.08, noise=6)],
                         ids=lambda c: f"seed{c['seed']}")
def test_vmap_vs_oracle_chain(api, oracle, case):
    i0, i1, xy = vmap_case(**case)
    dt = 16.0
    ref = oracle_vmap(oracle, i0, i1, xy, dt, cp_seed=7, num_cp_min=20)
    with api.Context(0) as ctx:
        ctx.set_images(i0, i1)
        got = ctx.vmap(xy, dt, cp_seed=7, num_cp_min=20)
        again = ctx.vmap(xy, dt, cp_seed=7, num_cp_min=20)      # a second run on the same context = a second program run
    for k in ("vx", "vy", "ex", "ey", "qual"):
        assert_bits_equal(again[k], got[k], k + " (repeat)")
    assert ref["cp_status"] == got["cp_status"] == 1
    assert got["offset_cp"] == ref["offset_cp"] == case["shift"]
    assert (got["dimx"], got["dimy"]) == (ref["dimx"], ref["dimy"])
    assert np.array_equal(got["flag_cp"], ref["flag_cp"])
    assert np.float32(got["cp_subint"][0]) == np.float32(ref["cp_subint"][0]) and np.float32(got["cp_subint"][1]) == np.float32(ref["cp_subint"][1])
    for k in ("vx", "vy", "ex", "ey", "qual"):
        assert_bits_equal(got[k], ref[k], k)
    assert np.isfinite(got["vx"]).mean() > 0.9


def test_vmap_no_control_points(api, oracle):
    i0, i1, xy = vmap_case(seed=13, shift=(1, 1))
    xy[:, 4] = 500.0                                 # nothing is slow: no CP candidates
    with api.Context(0) as ctx:
        ctx.set_images(i0, i1)
        got = ctx.vmap(xy, 16.0, cp_seed=1)
    assert got["cp_status"] == -1 and got["vx"] is None and not got["flag_cp"].any()


def test_postprocess_vs_oracle(api, oracle):
    dimx, dimy = 60, 45
    xy = synth.make_grid(dimx, dimy, 60, 60, 20, 20, 1806.0, angle_deg=37.0)
    mps = float(np.float32(xy[1, 0] - xy[0, 0]))
    dp = synth.synth_candidates(dimx, dimy, seed=4, k=32, p_out=0.5)
    mvn, nclus = oracle.cluster_candidates(dp, kmax=32)
    d0 = oracle.get_dpf0(mvn, nclus, dimx, dimy, 0.6)
    d1, x1, y1 = oracle.get_dpf1(d0, oracle.get_ruv_neighbor(xy, dimx, dimy, mps, 3.0), mvn, nclus, xy, 16.0, 15.0)
    d2 = oracle.qm(d1, x1, y1, oracle.get_ruv_neighbor(xy, dimx, dimy, mps, 5.0), mvn, nclus, xy)[0].reshape(-1)
    want = np.full((5, dimx * dimy), np.nan, np.float32)
    ok = d2 >= 0
    want[:, ok] = mvn[np.arange(dimx * dimy)[ok], d2[ok], :].T
    with api.Context(0) as ctx:
        got = ctx.mimc2_postprocess(dp, xy, dimx, dimy, 16.0, 15.0, mps)
    assert_bits_equal(got.reshape(5, -1), want, "vxyexyqual")
