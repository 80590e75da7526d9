"""GPU: the matrix-core matcher (match_mx_kernel.hip: dense NCC surfaces on v_mfma_i32_16x16x64_i8) against the oracle, bit for bit.
The parity suites reach it through "auto" on every 8-bit case; here are the cases that aim at ITS code paths: the clean and the general
form at every chip size, tiles that do not cover the whole cell grid (placement, climbs that leave the tile and are handed on), windows
and chips riddled with nulls, the closed-form and the correlated treatment of the never-written last window row / column, points at the
image edge, and the guarded fast finish against the register-tiled kernel on a full-size grid."""
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from conftest import ROOT, assert_bits_equal
from mimc3_amd import synth

pytestmark = pytest.mark.gpu

MX_OCW = (7, 15, 16, 30, 32, 40)


@pytest.fixture(scope="module")
def api():
    from mimc3_amd import api as a
    return a


def both_directions(api, ctx, c, off, uv, ocw, oracle, what):
    got = ctx.matching_ncc_dlc_2(c.xyuvav, c.offset, off, uv, ocw)
    assert ctx.last_path() == "u8_mfma"
    assert_bits_equal(got, oracle.match(c.i0, c.i1, c.xyuvav, c.offset, off, uv, ocw), what)
    sw = ctx.matching_ncc_dlc_2(c.xyuvav, -c.offset, off, -uv, ocw, swap=True)
    assert_bits_equal(sw, oracle.match(c.i1, c.i0, c.xyuvav, -c.offset, off, -uv, ocw), what + " swapped")
    return got


@pytest.mark.parametrize("null_frac", [0.0, 0.03, 0.15])
@pytest.mark.parametrize("ocw", MX_OCW)
def test_every_chip_size_clean_and_with_nulls(api, oracle, ocw, null_frac):
    """null_frac 0: every point takes the clean form (T4 in closed form); 0.03 / 0.15: window nulls, chip nulls and both."""
    c = synth.make_small(seed=5100 + ocw + int(100 * null_frac), shift=(3, -2), angle_deg=40.0, ocw=ocw, speed=1700.0,
                         h=2 * ocw + 230, w=2 * ocw + 240, dimx=6, dimy=5, noise_dn=2, null_frac=null_frac, offset=(1, -2))
    H, W = c.i0.shape
    off, uv = api.get_uv_pivot(c.xyuvav, c.dt, c.mpp, ocw, H, W)
    with api.Context(0) as ctx:
        ctx.set_images(c.i0, c.i1)
        both_directions(api, ctx, c, off, uv, ocw, oracle, f"ocw {ocw} nulls {null_frac}")


@pytest.mark.parametrize("forms", ["MIMC3_MX_GEN", "MIMC3_MX_WN", "both"])
@pytest.mark.parametrize("ocw", MX_OCW)
def test_forms_for_null_ridden_points(ocw, forms):
    """The window-null form and the general form (window and chip nulls on the matrix cores) are not the default -- the register-tiled
    kernel takes the null-ridden points: switched on, alone or together, they must give the oracle's bits at every chip size (a
    subprocess: the switches are read once per process)."""
    code = textwrap.dedent("""
        import sys, numpy as np
        sys.path.insert(0, %r); sys.path.insert(0, %r + "/tests")
        from conftest import assert_bits_equal
        from mimc3_amd import api, synth
        from oracle import oracle as orc
        o = orc.Oracle("port")
        ocw = %d
        c = synth.make_small(seed=5300 + ocw, shift=(2, 3), angle_deg=-35.0, ocw=ocw, speed=1500.0, h=2 * ocw + 220, w=2 * ocw + 230, dimx=5, dimy=5,
                             noise_dn=2, null_frac=0.12)
        H, W = c.i0.shape
        off, uv = api.get_uv_pivot(c.xyuvav, c.dt, c.mpp, ocw, H, W)
        with api.Context(0) as ctx:
            ctx.set_images(c.i0, c.i1)
            got = ctx.matching_ncc_dlc_2(c.xyuvav, c.offset, off, uv, ocw)
            assert ctx.last_path() == "u8_mfma"
            assert_bits_equal(got, o.match(c.i0, c.i1, c.xyuvav, c.offset, off, uv, ocw), "general form")
            sw = ctx.matching_ncc_dlc_2(c.xyuvav, -c.offset, off, -uv, ocw, swap=True)
            assert_bits_equal(sw, o.match(c.i1, c.i0, c.xyuvav, -c.offset, off, -uv, ocw), "general form swapped")
    """ % (ROOT, ROOT, ocw))
    env = dict(os.environ)
    for k in (("MIMC3_MX_GEN", "MIMC3_MX_WN") if forms == "both" else (forms,)):
        env[k] = "1"
    subprocess.check_call([sys.executable, "-c", code], env=env)


@pytest.mark.parametrize("angle", [45.0, -20.0, 100.0, 180.0])
@pytest.mark.parametrize("speed", [2600.0, 3200.0])
def test_tiles_that_do_not_cover_the_cell_grid(api, oracle, speed, angle):
    """17 to 29 pivots: the reachable cells outgrow the 32 x 32 tile, which is then centred on the pivots' starts; a climb that
    leaves it, or outlasts the 16 recorded scans, hands its point to the register-tiled kernel (the shift is far off the corridor
    for some of the points: long climbs)."""
    c = synth.make_small(seed=5400 + int(speed) + int(angle), shift=(9, 7), angle_deg=angle, ocw=16, speed=speed, h=330, w=340, dimx=6, dimy=6,
                         noise_dn=2, null_frac=0.03, margin=95)
    H, W = c.i0.shape
    off, uv = api.get_uv_pivot(c.xyuvav, c.dt, c.mpp, 16, H, W)
    assert 15 <= int(np.abs(uv[off[1:] - 1]).max()) <= 29
    with api.Context(0) as ctx:
        ctx.set_images(c.i0, c.i1)
        both_directions(api, ctx, c, off, uv, 16, oracle, f"speed {speed} angle {angle}")


@pytest.mark.parametrize("ocw", [7, 16])
def test_a_few_fast_points_among_slow_ones(api, oracle, ocw):
    """A velocity field with a few fast points (corridors of 35+ pivots: wider than the tile) among slow ones: the launch's longest
    corridor does not decide for the launch -- the kernel flags the fast points one by one for the register-tiled kernel (whose LDS
    carve is sized by them) and keeps the slow ones."""
    c = synth.make_small(seed=6200 + ocw, shift=(3, -4), angle_deg=-50.0, ocw=ocw, speed=1600.0, h=420, w=430, dimx=7, dimy=7,
                         noise_dn=2, null_frac=0.02, margin=130)
    xy = c.xyuvav.copy()
    xy[::9, 4:6] *= 3.4                                       # every ninth point 3.4 times as fast
    H, W = c.i0.shape
    off, uv = api.get_uv_pivot(xy, c.dt, c.mpp, ocw, H, W)
    last = np.abs(uv[off[1:] - 1]).max(axis=1)
    assert last.max() > 29 and np.median(last) <= 20, (last.max(), np.median(last))
    with api.Context(0) as ctx:
        ctx.set_images(c.i0, c.i1)
        got = ctx.matching_ncc_dlc_2(xy, c.offset, off, uv, ocw)
        assert ctx.last_path() == "u8_mfma"
        assert_bits_equal(got, oracle.match(c.i0, c.i1, xy, c.offset, off, uv, ocw), "mixed field")
        sw = ctx.matching_ncc_dlc_2(xy, -c.offset, off, -uv, ocw, swap=True)
        assert_bits_equal(sw, oracle.match(c.i1, c.i0, xy, -c.offset, off, -uv, ocw), "mixed field, swapped")
        ctx.set_path("u8px")
        assert_bits_equal(ctx.matching_ncc_dlc_2(xy, c.offset, off, uv, ocw), got, "register-tiled alone")


def test_points_at_the_image_edge_and_void_windows(api, oracle):
    """Windows that hang over the image edge (zeros outside, :877-884), points whose search area is more than 80 % void (-3),
    a chip with a single null pixel, a window whose only nulls are its never-written last row and column."""
    c = synth.make_small(seed=5500, shift=(2, -2), angle_deg=30.0, ocw=16, speed=1500.0, h=200, w=210, dimx=8, dimy=7, noise_dn=1, margin=17)
    c.i0[60, 70] = 0.0
    c.i1[:, :40] = 0.0                                  # a void band: -3 for the points whose windows sit in it
    H, W = c.i0.shape
    off, uv = api.get_uv_pivot(c.xyuvav, c.dt, c.mpp, 16, H, W)
    with api.Context(0) as ctx:
        ctx.set_images(c.i0, c.i1)
        got = both_directions(api, ctx, c, off, uv, 16, oracle, "edge")
    assert np.isnan(got[:, 0]).any() and (got[:, 2] > 0.9).any()


def test_flat_chips_and_flat_windows(api, oracle):
    """Zero variance: the reference's NCC is +-inf or NaN there (:734); the fast finish must send such cells to the exact path."""
    c = synth.make_small(seed=5600, shift=(1, 1), angle_deg=60.0, ocw=7, speed=1300.0, h=170, w=180, dimx=7, dimy=6)
    c.i0[40:80, 40:90] = 37.0
    c.i1[90:140, 60:130] = 91.0
    H, W = c.i0.shape
    off, uv = api.get_uv_pivot(c.xyuvav, c.dt, c.mpp, 7, H, W)
    with api.Context(0) as ctx:
        ctx.set_images(c.i0, c.i1)
        both_directions(api, ctx, c, off, uv, 7, oracle, "flat")


def test_window_of_16384_nulls_is_not_null_free(api, oracle):
    """ADVICE (round 3): one packed table query wraps its null count above 8,224 pixels -- a 130 x 130 written area holding a
    128 x 128 block of nulls read as null-free (16,384 + carry = 0 mod 2^14).  The count is now taken in sub-boxes."""
    c = synth.make_small(seed=5700, shift=(2, -2), angle_deg=45.0, ocw=32, speed=6600.0, h=440, w=440, dimx=3, dimy=3, margin=150, noise_dn=1)
    H, W = c.i0.shape
    off, uv = api.get_uv_pivot(c.xyuvav, c.dt, c.mpp, 32, H, W)
    g = 4                                                # the centre point: put the block inside its window
    k1 = int(off[g + 1])
    lu, lv = int(uv[k1 - 1, 0]), int(uv[k1 - 1, 1])
    dx2, dy2 = abs(lu) + 34, abs(lv) + 34
    assert 2 * dx2 >= 130 and 2 * dy2 >= 130
    u0, v0 = int(c.xyuvav[g, 2]), int(c.xyuvav[g, 3])
    c.i1[v0 - dy2 + 1:v0 - dy2 + 129, u0 - dx2 + 1:u0 - dx2 + 129] = 0.0
    want = oracle.match(c.i0, c.i1, c.xyuvav, c.offset, off, uv, 32)
    with api.Context(0) as ctx:
        ctx.set_images(c.i0, c.i1)
        for mode in ("auto", "u8px"):
            ctx.set_path(mode)
            assert_bits_equal(ctx.matching_ncc_dlc_2(c.xyuvav, c.offset, off, uv, 32), want, mode)


def test_c2_fast_finish_equals_register_tiled_kernel(api):
    """BASELINE C2, all 200,000 points, every chip size of the program: the matrix-core kernel (guarded rsq finish on ~2 x 10^8 cells)
    against the register-tiled kernel (the reference's own f64 operations), bit for bit."""
    c = synth.make_case("C2")
    H, W = c.i0.shape
    with api.Context(0) as ctx:
        ctx.set_images(c.i0, c.i1)
        for ocw in (7, 15, 16, 30, 40):
            off, uv = api.get_uv_pivot(c.xyuvav, c.dt, c.mpp, ocw, H, W)
            res = {}
            for mode in ("auto", "u8px"):
                ctx.set_path(mode)
                res[mode] = ctx.matching_ncc_dlc_2(c.xyuvav, c.offset, off, uv, ocw)
            assert_bits_equal(res["auto"], res["u8px"], f"ocw {ocw}")
