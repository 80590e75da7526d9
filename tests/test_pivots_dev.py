"""GPU: the device half of get_uv_pivot (csrc/pivot_kernel.hip: pivot lists expanded from host-made corridors) against the
oracle's get_uv_pivot (MIMC_module.c:543-602), and the fused entry mimc3_match_ncc_dlc_geo against the two-call form.
Bar: bit-equal CSR (offsets, lists, negated lists, extents); bit-identical matcher output."""
import numpy as np
import pytest

from conftest import assert_bits_equal
from mimc3_amd import api, synth

pytestmark = pytest.mark.gpu


def device_pivots(ctx, xy, dt, mpp, ocw, neg=False):
    from hipmem import DevArray
    n = xy.shape[0]
    d_xy = DevArray(src=np.ascontiguousarray(xy, np.float64))
    d_cor = DevArray(src=api.pivot_corridors(xy, dt, mpp))
    d_off = DevArray((n + 1,), np.int64)
    total, ext = ctx.get_uv_pivot_dev(d_xy.ptr, d_cor.ptr, n, ocw, d_off.ptr)          # call 1: sizes
    d_uv = DevArray((max(total, 1), 2), np.int32)
    d_uvn = DevArray((max(total, 1), 2), np.int32) if neg else None
    total2, ext2 = ctx.get_uv_pivot_dev(d_xy.ptr, d_cor.ptr, n, ocw, d_off.ptr, d_uv.ptr, d_uvn.ptr if neg else None, total)
    assert total2 == total and ext2 == ext
    return d_off.numpy(), d_uv.numpy()[:total], (d_uvn.numpy()[:total] if neg else None), ext


def directions(n=720):
    ang = np.deg2rad(np.arange(n) * 0.5 - 180.0)
    spd = 200.0 + 37.0 * (np.arange(n) % 101)
    xy = np.zeros((n, 6))
    xy[:, 2] = 400 + np.arange(n) % 7 + 0.25
    xy[:, 3] = 380 + np.arange(n) % 5 + 0.75
    xy[:, 4] = spd * np.cos(ang)
    xy[:, 5] = spd * np.sin(ang)
    return xy


def test_pivots_all_directions_vs_oracle(oracle):
    """every direction in half-degree steps (both branches of the step normalisation, all four truncation quadrants, T5)"""
    xy = directions()
    with api.Context(0) as ctx:
        ctx.set_images(np.ones((800, 830), np.float32), np.ones((800, 830), np.float32))   # the image size bounds the pivots
        for ocw in (7, 16, 40):
            off, uv, uvn, ext = device_pivots(ctx, xy, 16.0, 15.0, ocw, neg=True)
            want_off, want_uv = oracle.get_uv_pivot(xy, 16.0, 15.0, ocw, 800, 830)
            assert np.array_equal(off, want_off) and np.array_equal(uv, want_uv), ocw
            assert np.array_equal(uvn, -want_uv)
            assert ext == api.pivot_extent(want_off, want_uv)


def test_pivots_random_grid_with_clipped_corridors_vs_oracle(oracle):
    """60,000 points, every quadrant and speed (incl. ~0), some so close to the edge that the image bound ends the corridor"""
    n = 60000
    rng = np.random.default_rng(3)
    ang = rng.uniform(-np.pi, np.pi, n)
    spd = rng.uniform(0.0, 5000.0, n)
    spd[:50] = 0.0                                                # zero a-priori velocity: atan2(0, 0)
    xy = np.zeros((n, 6))
    xy[:, 2] = rng.uniform(45, 1955, n); xy[:, 3] = rng.uniform(45, 1755, n)
    xy[:, 4] = spd * np.cos(ang); xy[:, 5] = spd * np.sin(ang)
    with api.Context(0) as ctx:
        ctx.set_images(np.ones((1800, 2000), np.float32), np.ones((1800, 2000), np.float32))
        off, uv, _, ext = device_pivots(ctx, xy, 16.0, 15.0, 40)
    want_off, want_uv = oracle.get_uv_pivot(xy, 16.0, 15.0, 40, 1800, 2000)
    assert np.array_equal(off, want_off) and np.array_equal(uv, want_uv)
    assert ext == api.pivot_extent(want_off, want_uv)


def test_zero_pivot_point_is_refused():
    xy = directions(8)
    xy[3, 2] = 5.0                                                # chip cannot stay inside the image: no pivot (:589-591 overflows in the reference)
    with api.Context(0) as ctx:
        ctx.set_images(np.ones((800, 830), np.float32), np.ones((800, 830), np.float32))
        with pytest.raises(api.Mimc3Error) as e:
            device_pivots(ctx, xy, 16.0, 15.0, 16)
        assert e.value.code == -2


@pytest.mark.parametrize("ocw", [7, 16, 30])
def test_fused_entry_equals_pivots_then_match(oracle, ocw):
    """mimc3_match_ncc_dlc_geo == mimc3_get_uv_pivot + mimc3_match_ncc_dlc, forward and swapped, and == the oracle"""
    c = synth.make_small(seed=61, shift=(3, -2), angle_deg=205.0, ocw=ocw, dimx=24, dimy=20, h=420, w=460, null_frac=0.03, noise_dn=2, offset=(1, -1))
    H, W = c.i0.shape
    off, uv = api.get_uv_pivot(c.xyuvav, c.dt, c.mpp, ocw, H, W)
    with api.Context(0) as ctx:
        ctx.set_images(c.i0, c.i1)
        fw = ctx.matching_ncc_dlc_2(c.xyuvav, c.offset, off, uv, ocw)
        sw = ctx.matching_ncc_dlc_2(c.xyuvav, -c.offset, off, -uv, ocw, swap=True)
        gfw = ctx.matching_ncc_dlc_geo(c.xyuvav, c.offset, c.dt, c.mpp, ocw)
        gsw = ctx.matching_ncc_dlc_geo(c.xyuvav, -c.offset, c.dt, c.mpp, ocw, swap=True)
    assert_bits_equal(gfw, fw, "fused forward")
    assert_bits_equal(gsw, sw, "fused swapped")
    assert_bits_equal(gfw, oracle.match(c.i0, c.i1, c.xyuvav, c.offset, off, uv, ocw), "fused forward vs oracle")
    assert (fw[:, 2] > 0.5).mean() > 0.5
