"""GPU: the raw-DN entry points (mimc3_ctx_set_images_u8/_u16: the widening of GMA_float_load_tiff, GMA.c:288-310, done
on the device) give exactly what widening on the host and mimc3_ctx_set_images gives; pinned and pageable sources."""
import numpy as np
import pytest

from conftest import assert_bits_equal
from mimc3_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def api():
    from mimc3_amd import api as a
    return a


@pytest.mark.parametrize("bits,w", [(8, 320), (8, 323), (16, 320), (16, 321)])
@pytest.mark.parametrize("pinned", [False, True])
def test_raw_dn_equals_host_widening(api, oracle, bits, w, pinned):
    c = synth.make_small(seed=900 + bits + w, shift=(2, -3), angle_deg=35.0, ocw=15, h=300, w=w, dimx=8, dimy=7, null_frac=0.04,
                         noise_dn=2 if bits == 8 else 200, bits=bits, margin=60)
    dt = np.uint8 if bits == 8 else np.uint16
    r0, r1 = c.i0.astype(dt), c.i1.astype(dt)
    assert np.array_equal(r0.astype(np.float32), c.i0)
    if pinned:
        p0, p1 = api.pinned_empty(r0.shape, dt), api.pinned_empty(r1.shape, dt)
        p0[:] = r0; p1[:] = r1
        r0, r1 = p0, p1
    H, W = c.i0.shape
    off, uv = api.get_uv_pivot(c.xyuvav, c.dt, c.mpp, c.ocw, H, W)
    want = oracle.match(c.i0, c.i1, c.xyuvav, c.offset, off, uv, c.ocw)
    with api.Context(0) as ctx:
        ctx.set_images_raw(r0, r1)
        g0, g1 = ctx.get_images(H, W)
        assert np.array_equal(g0, c.i0) and np.array_equal(g1, c.i1)
        got = ctx.matching_ncc_dlc_2(c.xyuvav, c.offset, off, uv, c.ocw)
        assert ctx.last_path() == ("u8_mfma" if bits == 8 else "f32_tiled")
        assert_bits_equal(got, want, "raw upload")
        # filtered passes start from the widened f32 pair
        ctx.filter_images(api.CLI_KERNELS[0])
        f = ctx.matching_ncc_dlc_2(c.xyuvav, c.offset, off, uv, c.ocw)
        ctx.filter_images(None)
        again = ctx.matching_ncc_dlc_2(c.xyuvav, c.offset, off, uv, c.ocw)
        assert_bits_equal(again, want, "back to the raw pair")
        ctx.set_images(c.i0, c.i1)
        ctx.filter_images(api.CLI_KERNELS[0])
        assert_bits_equal(f, ctx.matching_ncc_dlc_2(c.xyuvav, c.offset, off, uv, c.ocw), "filtered pass")


def test_large_pageable_upload_through_pinned_chunks(api):
    """a pageable f32 pair larger than the two 4 MiB staging chunks arrives intact, odd sizes included"""
    rng = np.random.default_rng(5)
    i0 = rng.integers(0, 256, (1531, 2049)).astype(np.float32)
    i1 = rng.integers(0, 60000, (1531, 2049)).astype(np.float32)
    with api.Context(0) as ctx:
        ctx.set_images(i0, i1)
        g0, g1 = ctx.get_images(*i0.shape)
    assert np.array_equal(g0, i0) and np.array_equal(g1, i1)
