"""GPU: candidate clustering, dpf0 and dpf1 kernels (through the C ABI) against the reference-generated
golden vectors and the oracle.  Bar: cluster counts and ids identical, mvn / dpf_dx / dpf_dy bit-identical,
same sweep count.  (dpf1's interpolated intermediates pass through expf -- device vs glibc may differ in
the last ulp there -- but its outputs are cluster ids and copied cluster means.)"""
import numpy as np
import pytest

from conftest import assert_bits_equal, golden_files
from mimc3_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def api():
    from mimc3_amd import api as a
    return a


@pytest.fixture(scope="module")
def ctx(api):
    with api.Context(0) as c:
        yield c


@pytest.mark.parametrize("path", golden_files("n1_"), ids=lambda p: p.split("/")[-1][:-4])
def test_golden(api, ctx, path):
    z = np.load(path)
    dimy, dimx = z["dpf0"].shape
    kmax = z["mvn"].shape[1]
    mvn, nclus = ctx.calc_mean_var_num_dp_cluster(z["dp"], kmax)
    assert np.array_equal(nclus, z["nclus"])
    assert_bits_equal(mvn, z["mvn"], "mvn")
    dpf0 = ctx.get_dpf0(mvn, nclus, dimx, dimy, 0.6)
    assert np.array_equal(dpf0, z["dpf0"])
    ruv = api.get_ruv_neighbor(z["xyuvav"], dimx, dimy, float(z["meter_per_spacing"]), float(z["radius"]))
    assert np.array_equal(ruv, z["ruv"])
    d, x, y, _ = ctx.get_dpf1(dpf0, ruv, mvn, nclus, z["xyuvav"], float(z["dt"]), float(z["mpp"]))
    assert np.array_equal(d, z["dpf1"])
    assert_bits_equal(x, z["dx1"], "dx"); assert_bits_equal(y, z["dy1"], "dy")


def candidates(dimx, dimy, seed, k, pout):
    dp = synth.synth_candidates(dimx, dimy, seed=seed, k=k, p_out=pout)
    dp[:, ::7, 2] = 0.05          # whole points below the ncc gate -> empty cluster lists
    dp[:3, 5::11, 2] = -3.0       # invalid passes
    return dp


@pytest.mark.parametrize("dims", [(33, 21, 1, 1, 0.3), (64, 64, 2, 32, 0.5), (129, 67, 3, 32, 0.65), (50, 40, 4, 64, 0.5),
                                  (70, 3, 5, 16, 0.5)],
                         ids=lambda d: f"{d[0]}x{d[1]}_k{d[3]}")
def test_cluster_and_dpf0_vs_oracle(ctx, oracle, dims):
    dimx, dimy, seed, k, pout = dims
    dp = candidates(dimx, dimy, seed, k, pout)
    ref_mvn, ref_n = oracle.cluster_candidates(dp, kmax=k)
    mvn, nclus = ctx.calc_mean_var_num_dp_cluster(dp)            # kmax = ndp
    assert np.array_equal(nclus, ref_n)
    assert_bits_equal(mvn, ref_mvn, "mvn")
    for ratio in (0.6, 0.3, 0.95):
        assert np.array_equal(ctx.get_dpf0(mvn, nclus, dimx, dimy, ratio), oracle.get_dpf0(ref_mvn, ref_n, dimx, dimy, ratio))


def test_cluster_chain_and_ties(ctx, oracle):
    """single linkage: a chain of candidates 0.4 px apart is ONE cluster; exactly 0.5 px apart is two
    (strict <, MIMC_module.c:1152); duplicates; one candidate only; NaN candidates with a passing ncc"""
    k, n = 12, 6
    dp = np.zeros((k, n, 3), np.float32)
    dp[:, :, 2] = 0.9
    dp[:, 0, 0] = np.arange(k) * 0.4                       # chain
    dp[:, 1, 0] = (np.arange(k) % 2) * 0.5                 # two groups exactly 0.5 apart
    dp[:, 2, :2] = 1.25                                    # duplicates
    dp[1:, 3, 2] = 0.0                                     # one candidate
    dp[:, 4, 0] = np.array([0, 3, 0.1, 3.1, 6, 0.2, 6.3, 3.2, 9, 0.3, 6.1, 9.4], np.float32)   # interleaved
    dp[:, 5, 0] = np.arange(k) * 0.3; dp[:, 5, 1] = np.arange(k)[::-1] * 0.3
    ref_mvn, ref_n = oracle.cluster_candidates(dp, kmax=k)
    mvn, nclus = ctx.calc_mean_var_num_dp_cluster(dp)
    assert ref_n.tolist()[:4] == [1, 2, 1, 1]
    assert np.array_equal(nclus, ref_n)
    assert_bits_equal(mvn, ref_mvn, "mvn")
    # NaN candidate in the middle (fit of a flat surface): it takes an id but never carries it
    dp[4, 4, :2] = np.nan
    ref_mvn, ref_n = oracle.cluster_candidates(dp, kmax=k)
    mvn, nclus = ctx.calc_mean_var_num_dp_cluster(dp)
    assert np.array_equal(nclus, ref_n)
    assert_bits_equal(mvn, ref_mvn, "mvn with a NaN candidate")


def test_cluster_capacity(api, ctx):
    dp = candidates(20, 20, 7, 16, 0.9)
    with pytest.raises(api.Mimc3Error):
        ctx.calc_mean_var_num_dp_cluster(dp, kmax=2)


@pytest.mark.parametrize("dims", [(40, 40, 5, 8, 0.45, 37.0, 3.0), (61, 47, 9, 32, 0.5, -100.0, 3.0),
                                  (150, 110, 2, 16, 0.6, 80.0, 3.0), (90, 70, 11, 32, 0.75, 10.0, 5.0)],
                         ids=lambda d: f"{d[0]}x{d[1]}_p{d[4]}")
def test_dpf1_vs_oracle(api, ctx, oracle, dims):
    dimx, dimy, seed, k, pout, ang, radius = dims
    xy = synth.make_grid(dimx, dimy, 60, 60, 20, 20, 1806.0, angle_deg=ang)
    mps = float(np.float32(xy[1, 0] - xy[0, 0]))
    dp = candidates(dimx, dimy, seed, k, pout)
    mvn, nclus = oracle.cluster_candidates(dp, kmax=k)
    d0 = oracle.get_dpf0(mvn, nclus, dimx, dimy, 0.6)
    ruv = api.get_ruv_neighbor(xy, dimx, dimy, mps, radius)
    rd, rx, ry = oracle.get_dpf1(d0, ruv, mvn, nclus, xy, 16.0, 15.0)
    d, x, y, sweeps = ctx.get_dpf1(d0, ruv, mvn, nclus, xy, 16.0, 15.0)
    assert (d0 < 0).sum() > (rd < 0).sum()
    assert np.array_equal(d, rd)
    assert_bits_equal(x, rx, "dx"); assert_bits_equal(y, ry, "dy")
    assert sweeps >= 1


def test_dpf1_nothing_to_fill(api, ctx, oracle):
    """every point already assigned: one sweep per level is still run by the reference's loop, result = means"""
    dimx, dimy = 30, 20
    xy = synth.make_grid(dimx, dimy, 60, 60, 20, 20, 1806.0, angle_deg=20.0)
    mps = float(np.float32(xy[1, 0] - xy[0, 0]))
    dp = synth.synth_candidates(dimx, dimy, seed=3, k=8, p_out=0.0)
    mvn, nclus = oracle.cluster_candidates(dp, kmax=8)
    d0 = oracle.get_dpf0(mvn, nclus, dimx, dimy, 0.6)
    assert (d0 >= 0).all()
    ruv = api.get_ruv_neighbor(xy, dimx, dimy, mps, 3.0)
    rd, rx, ry = oracle.get_dpf1(d0, ruv, mvn, nclus, xy, 16.0, 15.0)
    d, x, y, sweeps = ctx.get_dpf1(d0, ruv, mvn, nclus, xy, 16.0, 15.0)
    assert np.array_equal(d, rd); assert_bits_equal(x, rx); assert_bits_equal(y, ry)
    assert sweeps == 1
