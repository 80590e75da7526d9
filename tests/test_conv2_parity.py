"""GPU: the image pre-filter (GMA_float_conv2, MIMC_module.c:2517-2585) through the C ABI against the oracle and
a golden vector; then the CLI's filtered passes end to end: filter on the device, match on the filtered pair,
compare with oracle-filter + oracle-match.  Bar: bit-identical planes and matcher outputs."""
import numpy as np
import pytest

from conftest import assert_bits_equal, golden_files
from mimc3_amd import synth

pytestmark = pytest.mark.gpu

# the CLI's three kernels (MIMC_main.c:176-194)
K_DX = np.array([[-1, 0, 1]], np.float32)
K_DY = np.array([[-1], [0], [1]], np.float32)
K_LAP = np.array([[-1 / 8] * 3, [-1 / 8, 1, -1 / 8], [-1 / 8] * 3], np.float32)
KERNELS = {"ddx": K_DX, "ddy": K_DY, "laplacian": K_LAP}


@pytest.fixture(scope="module")
def api():
    from mimc3_amd import api as a
    return a


@pytest.fixture(scope="module")
def ctx(api):
    with api.Context(0) as c:
        yield c


@pytest.mark.parametrize("path", golden_files("conv2_"), ids=lambda p: p.split("/")[-1][:-4])
def test_golden(ctx, path):
    z = np.load(path)
    img = z["img"].astype(np.float32)
    for name in KERNELS:
        assert_bits_equal(ctx.GMA_float_conv2(img, z["k_" + name]), z["out_" + name], name)


@pytest.mark.parametrize("name", list(KERNELS))
@pytest.mark.parametrize("shape", [(64, 64), (37, 301), (200, 513), (3, 3), (1025, 260)])
def test_cli_kernels_vs_oracle(ctx, oracle, name, shape):
    h, w = shape
    rng = np.random.default_rng(h * 1000 + w)
    img = rng.integers(0, 256, size=(h, w)).astype(np.float32)
    img[rng.random((h, w)) < 0.03] = 0.0                     # null DN
    assert_bits_equal(ctx.GMA_float_conv2(img, KERNELS[name]), oracle.float_conv2(img, KERNELS[name]), name)


def test_general_kernel_dirty_border_and_16bit(ctx, oracle):
    rng = np.random.default_rng(7)
    img = rng.integers(0, 65536, size=(90, 131)).astype(np.float32)
    img[10:20, 30:50] = 0.0
    img[40, 60] = 0.4            # (int)(0.4+0.5) == 0: null;  0.6 is not
    img[41, 60] = 0.6
    img[42, 60] = -0.7           # (int)(-0.2) == 0: null;  -1.6 is not
    img[43, 60] = -1.6
    for kshape in [(5, 3), (2, 2), (1, 1), (9, 9), (4, 7)]:
        k = rng.normal(size=kshape).astype(np.float32)
        dirty = rng.uniform(-1e4, 1e4, img.shape).astype(np.float32)      # the border of `out` takes part in the minimum
        assert_bits_equal(ctx.GMA_float_conv2(img, k, dirty), oracle.float_conv2(img, k, dirty), f"k{kshape}")
        assert_bits_equal(ctx.GMA_float_conv2(img, k), oracle.float_conv2(img, k), f"k{kshape} zero border")


def test_all_null_and_errors(api, ctx, oracle):
    img = np.zeros((20, 30), np.float32)
    assert_bits_equal(ctx.GMA_float_conv2(img, K_LAP), oracle.float_conv2(img, K_LAP), "all null")
    with pytest.raises(api.Mimc3Error):
        ctx.GMA_float_conv2(img, np.ones((10, 10), np.float32))          # more than 81 taps
    with pytest.raises(api.Mimc3Error):
        ctx.GMA_float_conv2(np.ones((2, 2), np.float32), K_LAP)           # kernel larger than the image


@pytest.mark.parametrize("name", list(KERNELS))
def test_filtered_passes_end_to_end(api, oracle, name):
    """MIMC_main.c:304-350: filter both images, then the forward and the swapped matcher pass on the filtered pair"""
    c = synth.make_small(seed=31, shift=(3, -2), angle_deg=25.0, ocw=7, null_frac=0.04, noise_dn=2)
    H, W = c.i0.shape
    k = KERNELS[name]
    f0, f1 = oracle.float_conv2(c.i0, k), oracle.float_conv2(c.i1, k)
    off, uv = api.get_uv_pivot(c.xyuvav, c.dt, c.mpp, c.ocw, H, W)
    ref_fw = oracle.match(f0, f1, c.xyuvav, c.offset, off, uv, c.ocw)
    ref_sw = oracle.match(f1, f0, c.xyuvav, -c.offset, off, -uv, c.ocw)
    with api.Context(0) as ctx:
        ctx.set_images(c.i0, c.i1)
        raw = ctx.matching_ncc_dlc_2(c.xyuvav, c.offset, off, uv, c.ocw)
        assert ctx.last_path() == "u8_mfma"
        ctx.filter_images(k)
        g0, g1 = ctx.get_images(H, W)
        assert_bits_equal(g0, f0, "filtered i0"); assert_bits_equal(g1, f1, "filtered i1")
        fw = ctx.matching_ncc_dlc_2(c.xyuvav, c.offset, off, uv, c.ocw)
        # gradients of 8-bit data are integers (u8 kernel if they stay below 256, else the u8 kernel through per-point
        # offsets with the u16 kernel behind it), the Laplacian multiples of 1/8 (u16 kernel)
        assert ctx.last_path() == ("u16_scaled" if name == "laplacian" else "u8_mfma" if max(f0.max(), f1.max()) <= 255 else "u8_offset")
        sw = ctx.matching_ncc_dlc_2(c.xyuvav, -c.offset, off, -uv, c.ocw, swap=True)
        assert_bits_equal(fw, ref_fw, "forward"); assert_bits_equal(sw, ref_sw, "swapped")
        ctx.filter_images(None)                            # back to the raw pair
        assert_bits_equal(ctx.matching_ncc_dlc_2(c.xyuvav, c.offset, off, uv, c.ocw), raw, "raw again")
        assert ctx.last_path() == "u8_mfma"


def test_three_filters_reuse_the_output_planes(api, oracle):
    """MIMC_main.c:302-307: i0c/i1c are created once; each filter inherits the border the previous one left"""
    c = synth.make_small(seed=33, shift=(1, 2), angle_deg=-60.0, ocw=7, null_frac=0.05)
    H, W = c.i0.shape
    o0 = np.zeros_like(c.i0); o1 = np.zeros_like(c.i1)
    with api.Context(0) as ctx:
        ctx.set_images(c.i0, c.i1)
        for name in ("ddx", "ddy", "laplacian"):
            o0 = oracle.float_conv2(c.i0, KERNELS[name], o0)
            o1 = oracle.float_conv2(c.i1, KERNELS[name], o1)
            ctx.filter_images(KERNELS[name])
            g0, g1 = ctx.get_images(H, W)
            assert_bits_equal(g0, o0, name + " i0"); assert_bits_equal(g1, o1, name + " i1")
        assert not np.array_equal(o0, oracle.float_conv2(c.i0, KERNELS["laplacian"]))     # the inherited border matters
        ctx.set_images(c.i0, c.i1)                          # a new pair starts from zero planes again
        ctx.filter_images(KERNELS["laplacian"])
        assert_bits_equal(ctx.get_images(H, W)[0], oracle.float_conv2(c.i0, KERNELS["laplacian"]), "fresh")
        ctx.filter_images(KERNELS["ddx"])                   # inherits the Laplacian's border ...
        ctx.filter_images(None)                             # ... until the sequence is ended: back to raw, fresh planes next
        assert_bits_equal(ctx.get_images(H, W)[0], c.i0, "raw again")
        ctx.filter_images(KERNELS["ddx"])
        assert_bits_equal(ctx.get_images(H, W)[0], oracle.float_conv2(c.i0, KERNELS["ddx"]), "fresh after None")
