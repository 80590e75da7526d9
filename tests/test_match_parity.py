"""GPU: the HIP matcher (through the C ABI) against the committed golden vectors and the oracle.
Bar: bit-exact (out[N,3] as uint32 words; NaN == NaN) for integer-DN inputs."""
import numpy as np
import pytest

from conftest import assert_bits_equal, golden_files, load_match_golden
from mimc3_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def api():
    from mimc3_amd import api as a
    return a


U8_OCW = (7, 15, 16, 30, 32, 40)   # chip sizes the exact u8 kernel is instantiated for


F32T_OCW = (7, 15, 16, 30, 32, 40)     # chip sizes of the register-tiled f32 kernel


def expected_path(mode, i0, ocw, i1=None):
    imgs = [i0] if i1 is None else [i0, i1]
    is_u8 = all(float(i.max()) <= 255.0 and float(i.min()) >= 0 and np.array_equal(i, np.rint(i)) for i in imgs)
    is_si = all(float(i.min()) >= 0 and ((float(i.max()) <= 4095.0 and np.array_equal(i, np.rint(i))) or
                (float(i.max()) * 8 <= 4095.0 and np.array_equal(i * 8, np.rint(i * 8)))) for i in imgs)
    if mode in ("auto", "u8px") and ocw in U8_OCW and is_u8:
        return "u8_mfma" if mode == "auto" else "u8_exact"     # the matrix-core kernel first ("auto"), or the register-tiled kernel alone
    if ocw in U8_OCW and ((mode in ("auto", "u8px") and is_si and not is_u8) or (mode == "u16" and is_u8)):
        return "u16_scaled"
    return "f32_tiled" if (mode != "general" and ocw in F32T_OCW) else "general_f32"


@pytest.mark.parametrize("mode", ["auto", "u8px", "general", "f32", "u16"])
@pytest.mark.parametrize("path", golden_files("match_"), ids=lambda p: p.split("match_")[-1][:-4])
def test_golden(api, path, mode):
    g = load_match_golden(path)
    H, W = g["i0"].shape
    off, uv = api.get_uv_pivot(g["xyuvav"], g["dt"], g["mpp"], g["ocw"], H, W)
    with api.Context(0) as ctx:
        ctx.set_images(g["i0"], g["i1"])
        ctx.set_path(mode)
        out = ctx.matching_ncc_dlc_2(g["xyuvav"], g["offset"], off, uv, g["ocw"])
        assert ctx.last_path() == expected_path(mode, g["i0"], g["ocw"], g["i1"])
        assert_bits_equal(out, g["out"], "forward")
        out_sw = ctx.matching_ncc_dlc_2(g["xyuvav"], -g["offset"], off, -uv, g["ocw"], swap=True)
        assert_bits_equal(out_sw, g["out_swapped"], "swapped")


SMALL = [
    dict(seed=31, shift=(3, -2), angle_deg=10.0, ocw=7),
    dict(seed=32, shift=(-3, 1), angle_deg=170.0, ocw=12, noise_dn=3, null_frac=0.08),
    dict(seed=33, shift=(0, 4), angle_deg=-88.0, ocw=9, subpixel=(0.5, 0.25), bits=16),
    dict(seed=34, shift=(2, 2), angle_deg=-40.0, ocw=15, speed=2500.0, h=220, w=230, offset=(-1, 2)),
    dict(seed=35, shift=(6, -6), angle_deg=45.0, ocw=16, speed=1806.0, h=260, w=260, noise_dn=2, null_frac=0.03),
    dict(seed=36, shift=(1, 0), angle_deg=0.0, ocw=30, speed=700.0, h=300, w=320, dimx=5, dimy=4),   # 61x61 chip
    dict(seed=37, shift=(-2, -5), angle_deg=112.0, ocw=40, speed=900.0, h=360, w=340, dimx=4, dimy=4),  # 81x81 chip
    dict(seed=38, shift=(4, -3), angle_deg=38.0, ocw=7, speed=2000.0, noise_dn=2, null_frac=0.10, offset=(3, -2)),
    dict(seed=39, shift=(-3, -3), angle_deg=135.0, ocw=15, speed=1700.0, h=240, w=250, null_frac=0.06),
    dict(seed=40, shift=(5, 4), angle_deg=-38.0, ocw=32, speed=3000.0, h=400, w=420, dimx=5, dimy=5, noise_dn=1, null_frac=0.03),
    dict(seed=42, shift=(0, -2), angle_deg=91.0, ocw=30, speed=1500.0, h=330, w=300, dimx=4, dimy=5, null_frac=0.05),
    dict(seed=43, shift=(2, -2), angle_deg=45.0, ocw=16, speed=1806.0, h=200, w=210, margin=17, dimx=7, dimy=7),  # windows over the edge
]


@pytest.mark.parametrize("mode", ["auto", "u8px", "general", "f32", "u16"])
@pytest.mark.parametrize("kw", SMALL, ids=lambda k: f"seed{k['seed']}_ocw{k['ocw']}")
def test_vs_oracle(api, oracle, kw, mode):
    c = synth.make_small(**kw)
    H, W = c.i0.shape
    off, uv = api.get_uv_pivot(c.xyuvav, c.dt, c.mpp, c.ocw, H, W)
    want = oracle.match(c.i0, c.i1, c.xyuvav, c.offset, off, uv, c.ocw)
    with api.Context(0) as ctx:
        ctx.set_images(c.i0, c.i1)
        ctx.set_path(mode)
        got = ctx.matching_ncc_dlc_2(c.xyuvav, c.offset, off, uv, c.ocw)
        assert ctx.last_path() == expected_path(mode, c.i0, c.ocw, c.i1)
        sw = ctx.matching_ncc_dlc_2(c.xyuvav, -c.offset, off, -uv, c.ocw, swap=True)
    assert_bits_equal(got, want)
    assert_bits_equal(sw, oracle.match(c.i1, c.i0, c.xyuvav, -c.offset, off, -uv, c.ocw), "swapped")


@pytest.mark.parametrize("mode", ["auto", "u8px", "general", "f32", "u16"])
def test_c1_config_vs_oracle(api, oracle, mode):
    """BASELINE configs[0]: 512^2, 1,024 points, 33x33 chip / 65x65 window."""
    c = synth.make_case("C1")
    H, W = c.i0.shape
    off, uv = api.get_uv_pivot(c.xyuvav, c.dt, c.mpp, c.ocw, H, W)
    want = oracle.match(c.i0, c.i1, c.xyuvav, c.offset, off, uv, c.ocw)
    with api.Context(0) as ctx:
        ctx.set_images(c.i0, c.i1)
        ctx.set_path(mode)
        got = ctx.matching_ncc_dlc_2(c.xyuvav, c.offset, off, uv, c.ocw)
    assert_bits_equal(got, want)
    ok = got[:, 2] > 0.5
    assert ok.mean() > 0.9
    assert np.abs(np.median(got[ok, 0]) - c.shift[0]) < 0.2 and np.abs(np.median(got[ok, 1]) - c.shift[1]) < 0.2


def test_float_images_within_tolerance(api, oracle):
    """Non-integer DN: f64 sums are no longer exact, so the summation order shows in the last bits.
    north_star tolerance: |d(u,v)| <= 1e-4 px, identical invalid mask."""
    c = synth.make_small(seed=41, shift=(2, -3), angle_deg=-60.0, ocw=11, null_frac=0.04)
    rng = np.random.default_rng(1)
    i0 = (c.i0 * np.float32(0.731) + np.where(c.i0 > 0, rng.random(c.i0.shape, dtype=np.float32), 0)).astype(np.float32)
    i1 = (c.i1 * np.float32(0.731) + np.where(c.i1 > 0, rng.random(c.i1.shape, dtype=np.float32), 0)).astype(np.float32)
    H, W = i0.shape
    off, uv = api.get_uv_pivot(c.xyuvav, c.dt, c.mpp, c.ocw, H, W)
    want = oracle.match(i0, i1, c.xyuvav, c.offset, off, uv, c.ocw)
    with api.Context(0) as ctx:
        ctx.set_images(i0, i1)
        got = ctx.matching_ncc_dlc_2(c.xyuvav, c.offset, off, uv, c.ocw)
    assert np.array_equal(np.isnan(got), np.isnan(want))
    assert np.nanmax(np.abs(got - want)) <= 1e-4


@pytest.mark.parametrize("mode", ["auto", "u8px", "general", "f32", "u16"])
def test_long_climbs(api, oracle, mode):
    """Smooth texture + a shift far along the corridor: pivots climb 10+ scans to the peak, which
    exercises the u8 kernel's generic (sequential) replay behind the speculative one."""
    c = synth.make_small(seed=91, shift=(13, -13), angle_deg=45.0, ocw=16, speed=6000.0, h=300, w=300, dimx=6, dimy=6,
                         sigma=7.0, margin=70)
    H, W = c.i0.shape
    off, uv = api.get_uv_pivot(c.xyuvav, c.dt, c.mpp, c.ocw, H, W)
    want = oracle.match(c.i0, c.i1, c.xyuvav, c.offset, off, uv, c.ocw)
    with api.Context(0) as ctx:
        ctx.set_images(c.i0, c.i1)
        ctx.set_path(mode)
        got = ctx.matching_ncc_dlc_2(c.xyuvav, c.offset, off, uv, c.ocw)
    assert_bits_equal(got, want)
    assert np.abs(np.nanmedian(got[:, 0]) - 13) < 0.5


def test_u8_cache_overflow_hands_points_to_general_kernel(oracle):
    """Shrink the u8 kernel's per-point NCC cache (debug env knob) so that every point overflows: the
    points are appended to a device list and redone by the general kernel in list mode."""
    import subprocess, sys, os, textwrap
    from conftest import ROOT
    code = textwrap.dedent("""
        import sys, numpy as np
        sys.path.insert(0, %r)
        from mimc3_amd import api, synth
        c = synth.make_small(seed=95, shift=(3, -3), angle_deg=45.0, ocw=16, speed=1806.0, h=240, w=240, dimx=6, dimy=6,
                             noise_dn=2, null_frac=0.04)
        H, W = c.i0.shape
        off, uv = api.get_uv_pivot(c.xyuvav, c.dt, c.mpp, c.ocw, H, W)
        with api.Context(0) as ctx:
            ctx.set_images(c.i0, c.i1)
            ctx.set_path("u8px")           # the register-tiled kernel alone: its overflow hand-over is what is tested
            got = ctx.matching_ncc_dlc_2(c.xyuvav, c.offset, off, uv, c.ocw)
            assert ctx.last_path() == "u8_exact"
        np.save(sys.argv[1], got)
    """ % ROOT)
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        outs = []
        for slack in ("-100", "64"):
            f = os.path.join(d, f"o{slack}.npy")
            subprocess.check_call([sys.executable, "-c", code, f], env=dict(os.environ, MIMC3_U8_CACHE_SLACK=slack))
            outs.append(np.load(f))
    c = synth.make_small(seed=95, shift=(3, -3), angle_deg=45.0, ocw=16, speed=1806.0, h=240, w=240, dimx=6, dimy=6,
                         noise_dn=2, null_frac=0.04)
    H, W = c.i0.shape
    off, uv = oracle.get_uv_pivot(c.xyuvav, c.dt, c.mpp, c.ocw, H, W)
    want = oracle.match(c.i0, c.i1, c.xyuvav, c.offset, off, uv, c.ocw)
    assert_bits_equal(outs[0], want, "all points overflowed")
    assert_bits_equal(outs[1], want, "normal cache")


@pytest.mark.parametrize("kind", ["12bit", "eighths", "mixed"])
@pytest.mark.parametrize("ocw", [7, 16, 30])
def test_scaled_integer_planes(api, oracle, kind, ocw):
    """12-bit DN and multiples of 1/8 (what the CLI's Laplacian filter produces, MIMC_main.c:188-196) go
    through the exact u16 kernel: every f32 product of the reference is exact, sums rescale by powers of two."""
    c = synth.make_small(seed=110 + ocw, shift=(3, -2), angle_deg=33.0, ocw=ocw, speed=1400.0, h=300, w=310, dimx=6, dimy=5,
                         bits=12, null_frac=0.05, noise_dn=3)
    i0, i1 = c.i0, c.i1
    if kind == "eighths":
        i0, i1 = (i0 / 8).astype(np.float32), (i1 / 8).astype(np.float32)
    elif kind == "mixed":
        i1 = (i1 / 8).astype(np.float32)
    H, W = i0.shape
    off, uv = api.get_uv_pivot(c.xyuvav, c.dt, c.mpp, ocw, H, W)
    want = oracle.match(i0, i1, c.xyuvav, c.offset, off, uv, ocw)
    want_sw = oracle.match(i1, i0, c.xyuvav, -c.offset, off, -uv, ocw)
    with api.Context(0) as ctx:
        ctx.set_images(i0, i1)
        got = ctx.matching_ncc_dlc_2(c.xyuvav, c.offset, off, uv, ocw)
        assert ctx.last_path() == "u16_scaled"
        sw = ctx.matching_ncc_dlc_2(c.xyuvav, -c.offset, off, -uv, ocw, swap=True)
    assert_bits_equal(got, want)
    assert_bits_equal(sw, want_sw, "swapped")


def test_all_four_cli_chip_sizes_on_one_pair(api, oracle):
    """The CLI's schedule: ocw 7, 15, 30, 40 forward + swapped on one resident pair (MIMC_main.c:261-300)."""
    c = synth.make_small(seed=61, shift=(3, -3), angle_deg=45.0, ocw=40, speed=1200.0, h=330, w=340, dimx=5, dimy=5,
                         noise_dn=2, null_frac=0.02)
    H, W = c.i0.shape
    with api.Context(0) as ctx:
        ctx.set_images(c.i0, c.i1)
        for ocw in (7, 15, 30, 40):
            off, uv = api.get_uv_pivot(c.xyuvav, c.dt, c.mpp, ocw, H, W)
            for mode, path in (("auto", "u8_mfma"), ("u8px", "u8_exact")):
                ctx.set_path(mode)
                fwd = ctx.matching_ncc_dlc_2(c.xyuvav, c.offset, off, uv, ocw)
                assert ctx.last_path() == path
                assert_bits_equal(fwd, oracle.match(c.i0, c.i1, c.xyuvav, c.offset, off, uv, ocw), f"ocw{ocw} fwd {mode}")
                sw = ctx.matching_ncc_dlc_2(c.xyuvav, -c.offset, off, -uv, ocw, swap=True)
                assert_bits_equal(sw, oracle.match(c.i1, c.i0, c.xyuvav, -c.offset, off, -uv, ocw), f"ocw{ocw} swapped {mode}")


@pytest.mark.parametrize("null_frac", [0.10, 0.30])
@pytest.mark.parametrize("ocw", [30, 40])
def test_big_chips_with_many_nulls(api, oracle, ocw, null_frac):
    """Many null pixels on the 61x61 / 81x81 chips: hundreds of entries in the window- and chip-null lists of the sparse
    corrections (their caps overflow at 30 %: those points fall back to the six-sum body), nulls in the 81-row chip's tail
    rows (listed since its tail tasks run without derived masks).  u8, u16 and d/dx (u8 through offsets), both directions."""
    c = synth.make_small(seed=4000 + ocw + int(100 * null_frac), shift=(2, -3), angle_deg=60.0, ocw=ocw, speed=1000.0, h=340, w=350,
                         dimx=4, dimy=4, noise_dn=2, null_frac=null_frac)
    H, W = c.i0.shape
    off, uv = api.get_uv_pivot(c.xyuvav, c.dt, c.mpp, ocw, H, W)
    with api.Context(0) as ctx:
        ctx.set_images(c.i0, c.i1)
        for mode, path in (("auto", "u8_mfma"), ("u8px", "u8_exact"), ("u16", "u16_scaled")):
            ctx.set_path(mode)
            got = ctx.matching_ncc_dlc_2(c.xyuvav, c.offset, off, uv, ocw)
            assert ctx.last_path() == path
            assert_bits_equal(got, oracle.match(c.i0, c.i1, c.xyuvav, c.offset, off, uv, ocw), mode)
            sw = ctx.matching_ncc_dlc_2(c.xyuvav, -c.offset, off, -uv, ocw, swap=True)
            assert_bits_equal(sw, oracle.match(c.i1, c.i0, c.xyuvav, -c.offset, off, -uv, ocw), mode + " swapped")
        ctx.set_path("auto")
        ctx.filter_images(api.CLI_KERNELS[0])
        f0, f1 = ctx.get_images(H, W)
        gf = ctx.matching_ncc_dlc_2(c.xyuvav, c.offset, off, uv, ocw)
        assert ctx.last_path() in ("u8_mfma", "u8_exact", "u8_offset", "u16_scaled")
        assert_bits_equal(gf, oracle.match(f0, f1, c.xyuvav, c.offset, off, uv, ocw), "d/dx")


@pytest.mark.parametrize("ocw", [7, 15, 30, 40])
def test_offsets_from_tile_ranges_or_from_the_full_scan(api, oracle, ocw, monkeypatch):
    """u8 through per-point offsets (gradient of an 8-bit pair): the offsets come from the planes' 16x16-pixel tile ranges where
    that bound fits a byte, from the full scan of the point's pixels where it does not, and the point goes to the u16 kernel
    where even the exact range does not fit.  Any valid offsets give the same exact sums: with the tile bound switched off the
    result must not change, and both equal the reference.  Noise 12 DN spreads the local ranges around the 8-bit limit."""
    c = synth.make_small(seed=5100 + ocw, shift=(1, -2), angle_deg=25.0, ocw=ocw, speed=900.0, h=330, w=350, dimx=5, dimy=4,
                         noise_dn=12, null_frac=0.02)
    H, W = c.i0.shape
    off, uv = api.get_uv_pivot(c.xyuvav, c.dt, c.mpp, ocw, H, W)
    with api.Context(0) as ctx:
        ctx.set_images(c.i0, c.i1)
        ctx.filter_images(api.CLI_KERNELS[1])
        f0, f1 = ctx.get_images(H, W)
        want = oracle.match(f0, f1, c.xyuvav, c.offset, off, uv, ocw)
        a = ctx.matching_ncc_dlc_2(c.xyuvav, c.offset, off, uv, ocw)
        path = ctx.last_path()
        monkeypatch.setenv("MIMC3_NO_RANGE_TILES", "1")
        b = ctx.matching_ncc_dlc_2(c.xyuvav, c.offset, off, uv, ocw)
        monkeypatch.delenv("MIMC3_NO_RANGE_TILES")
    assert path in ("u8_offset", "u16_scaled", "u8_exact", "u8_mfma")
    assert_bits_equal(a, want, "tile bound")
    assert_bits_equal(b, want, "full scan")


def test_edge_cases(api):
    c = synth.make_small(seed=51)
    H, W = c.i0.shape
    off, uv = api.get_uv_pivot(c.xyuvav, c.dt, c.mpp, c.ocw, H, W)
    with api.Context(0) as ctx:
        with pytest.raises(api.Mimc3Error):        # images not set
            ctx.matching_ncc_dlc_2(c.xyuvav, c.offset, off, uv, c.ocw)
        ctx.set_images(c.i0, c.i1)
        bad = c.xyuvav.copy()
        bad[0, 2] = 2.0                            # chip would leave the image: refused, not UB
        with pytest.raises(api.Mimc3Error) as e:
            ctx.matching_ncc_dlc_2(bad, c.offset, off, uv, c.ocw)
        assert e.value.code == -2
        # all-null pair -> every point invalid (NaN, NaN, -3)
        ctx.set_images(np.zeros_like(c.i0), np.zeros_like(c.i1))
        out = ctx.matching_ncc_dlc_2(c.xyuvav, c.offset, off, uv, c.ocw)
        assert np.isnan(out[:, :2]).all() and (out[:, 2] == -3).all()
        # single grid point
        ctx.set_images(c.i0, c.i1)
        one = ctx.matching_ncc_dlc_2(c.xyuvav[:1], c.offset, off[:2], uv[:off[1]], c.ocw)
        full = ctx.matching_ncc_dlc_2(c.xyuvav, c.offset, off, uv, c.ocw)
        assert_bits_equal(one, full[:1])


# ---- 9-bit integer imagery (what the gradient filters make of 8-bit images): the u8 kernels through per-point offsets,
#      with the u16 kernel redoing the points whose chip or window does not fit a local 8-bit range -----------------------
def nine_bit_case(seed, ocw, wide_frac, null_frac=0.03, shift=(2, -3)):
    c = synth.make_small(seed=seed, shift=shift, angle_deg=35.0, ocw=ocw, h=300, w=320, dimx=12, dimy=11, null_frac=null_frac,
                         noise_dn=2, margin=ocw + 40)
    rng = np.random.default_rng(seed)
    # centre the 8-bit texture around 256 with a reduced amplitude: values 150..360, local range < 255 almost everywhere
    def remap(img):
        out = np.where(img > 0, np.round(150.0 + img * 0.8), 0.0).astype(np.float32)
        # a few bright spots push some windows / chips over the 8-bit range: those points must take the u16 fallback
        n = int(wide_frac * out.size / 400)
        ys = rng.integers(0, out.shape[0], n); xs = rng.integers(0, out.shape[1], n)
        out[ys, xs] = np.where(out[ys, xs] > 0, 500.0, 0.0)
        return out
    c.i0[:] = remap(c.i0); c.i1[:] = remap(c.i1)
    return c


@pytest.mark.parametrize("ocw", [7, 15, 16, 30, 40])
@pytest.mark.parametrize("wide_frac", [0.0, 0.02, 1.0], ids=["all_fit", "some_fallback", "mostly_fallback"])
def test_offset_u8_path_vs_oracle(api, oracle, ocw, wide_frac):
    c = nine_bit_case(seed=500 + ocw, ocw=ocw, wide_frac=wide_frac)
    H, W = c.i0.shape
    off, uv = api.get_uv_pivot(c.xyuvav, c.dt, c.mpp, c.ocw, H, W)
    want = oracle.match(c.i0, c.i1, c.xyuvav, c.offset, off, uv, c.ocw)
    want_sw = oracle.match(c.i1, c.i0, c.xyuvav, -c.offset, off, -uv, c.ocw)
    with api.Context(0) as ctx:
        ctx.set_images(c.i0, c.i1)
        got = ctx.matching_ncc_dlc_2(c.xyuvav, c.offset, off, uv, c.ocw)
        path = ctx.last_path()
        got_sw = ctx.matching_ncc_dlc_2(c.xyuvav, -c.offset, off, -uv, c.ocw, swap=True)
        ctx.set_path("u16")
        ref16 = ctx.matching_ncc_dlc_2(c.xyuvav, c.offset, off, uv, c.ocw)
        assert ctx.last_path() == "u16_scaled"
    assert path in ("u8_offset", "u16_scaled")          # the tile estimate decides whether the offset path is tried at all
    if wide_frac == 0.0:
        assert path == "u8_offset"
    assert_bits_equal(got, want, "forward"); assert_bits_equal(got_sw, want_sw, "swapped")
    assert_bits_equal(ref16, want, "u16 kernel alone")
    assert (got[:, 2] > -2.5).mean() > 0.5


@pytest.mark.parametrize("ocw", [7, 15, 30, 40])
@pytest.mark.parametrize("kind", ["16bit", "float"])
def test_16bit_and_float_imagery_on_the_tiled_f32_kernel(api, oracle, ocw, kind):
    """16-bit DN (Landsat-8-like: f32 products round, T1) and arbitrary float imagery take the register-tiled f32 kernel
    for all four CLI chip sizes.  f64 sums are not exact there, so the summation order shows in the last bits:
    north_star tolerance |d(u,v)| <= 1e-4 px, identical invalid mask; peak NCC within 1e-6."""
    c = synth.make_small(seed=700 + ocw, shift=(3, -2), angle_deg=40.0, ocw=ocw, h=300, w=320, dimx=7, dimy=6, null_frac=0.04,
                         noise_dn=2, margin=ocw + 40, bits=16 if kind == "16bit" else 8)
    i0, i1 = c.i0, c.i1
    if kind == "float":
        rng = np.random.default_rng(ocw)
        i0 = (i0 * np.float32(0.731) + np.where(i0 > 0, rng.random(i0.shape, dtype=np.float32), 0)).astype(np.float32)
        i1 = (i1 * np.float32(0.731) + np.where(i1 > 0, rng.random(i1.shape, dtype=np.float32), 0)).astype(np.float32)
    H, W = i0.shape
    off, uv = api.get_uv_pivot(c.xyuvav, c.dt, c.mpp, ocw, H, W)
    want = oracle.match(i0, i1, c.xyuvav, c.offset, off, uv, ocw)
    with api.Context(0) as ctx:
        ctx.set_images(i0, i1)
        got = ctx.matching_ncc_dlc_2(c.xyuvav, c.offset, off, uv, ocw)
        assert ctx.last_path() == "f32_tiled"
        ctx.set_path("general")
        gen = ctx.matching_ncc_dlc_2(c.xyuvav, c.offset, off, uv, ocw)
    for name, g in (("tiled", got), ("general", gen)):
        assert np.array_equal(np.isnan(g), np.isnan(want)), name
        assert np.nanmax(np.abs(g[:, :2] - want[:, :2])) <= 1e-4, name
        assert np.nanmax(np.abs(g[:, 2] - want[:, 2])) <= 1e-6, name
    assert (got[:, 2] > -2.5).mean() > 0.5
    if kind == "16bit":
        # integer DN: every f32 product is an integer (rounded to 24 significant bits when it exceeds them) and the f64
        # sums stay below 2^53, i.e. exact in ANY order -- 16-bit imagery is bit-identical as well
        assert_bits_equal(got, want, "tiled, 16-bit"); assert_bits_equal(gen, want, "general, 16-bit")


@pytest.mark.parametrize("mode", ["auto", "u8px", "u16", "f32", "general"])
@pytest.mark.parametrize("reach", [60, 100])
def test_long_corridor_big_chip_falls_back_when_lds_is_short(api, oracle, reach, mode):
    """A fast a-priori on the largest CLI chip (ocw 40): the window (2*(reach+42)+1)^2 outgrows the 160 KB LDS carve of
    the tiled f32 kernel at reach ~47, of the u16 kernel at ~55 and of the u8 kernel at ~75 (window + the f32 NCC cache of
    the 2*reach+6 wide compact grid).  Such launches take the next policy down to the
    general kernel (which reads the window through L2) -- the reference handles these inputs, so must the library."""
    speed = {60: 14100.0, 100: 23600.0}[reach]
    size = 2 * (reach + 42) + 140
    c = synth.make_small(seed=800 + reach, shift=(5, -5), angle_deg=45.0, ocw=40, speed=speed, h=size, w=size + 8, dimx=3, dimy=3,
                         margin=reach + 60, noise_dn=2, null_frac=0.02)
    H, W = c.i0.shape
    off, uv = api.get_uv_pivot(c.xyuvav, c.dt, c.mpp, c.ocw, H, W)
    mu = int(np.abs(uv[off[1:] - 1]).max())
    assert mu >= reach - 2, mu
    want = oracle.match(c.i0, c.i1, c.xyuvav, c.offset, off, uv, c.ocw)
    with api.Context(0) as ctx:
        ctx.set_images(c.i0, c.i1)
        ctx.set_path(mode)
        got = ctx.matching_ncc_dlc_2(c.xyuvav, c.offset, off, uv, c.ocw)
        path = ctx.last_path()
        sw = ctx.matching_ncc_dlc_2(c.xyuvav, -c.offset, off, -uv, c.ocw, swap=True)
    # ((60, "auto"): more than 64 pivots per point -- not a launch the matrix-core kernel takes)
    expect = {(60, "auto"): "u8_exact", (60, "u8px"): "u8_exact", (100, "u8px"): "general_f32", (60, "u16"): "general_f32", (60, "f32"): "general_f32", (60, "general"): "general_f32",
              (100, "auto"): "general_f32", (100, "u16"): "general_f32", (100, "f32"): "general_f32", (100, "general"): "general_f32"}
    assert path == expect[(reach, mode)]
    assert_bits_equal(got, want)
    assert_bits_equal(sw, oracle.match(c.i1, c.i0, c.xyuvav, -c.offset, off, -uv, c.ocw), "swapped")


@pytest.mark.parametrize("mode", ["auto", "u16"])
@pytest.mark.parametrize("ocw,angle", [(15, 10.0), (30, 10.0), (30, 80.0)])
def test_corridor_of_more_than_64_pivots(api, oracle, ocw, angle, mode):
    """~100 pivots per point at the two chip sizes that have a many-pivot kernel form (the control-point stage's): the pivots
    beyond the first 64 are climbed by lanes that already own a pivot and recorded in LDS; the exact replay reads them from
    there (visited bits in LDS: the compact grid is wider than 64 cells).  Raw pair (u8 / u16 kernels), its d/dx (u8 through
    per-point offsets), forward and swapped."""
    c = synth.make_small(seed=900 + ocw, shift=(5, -5), angle_deg=angle, ocw=ocw, speed=17000.0,
                         h=(2 * (40 + ocw) + 140) if angle < 45 else 2 * (100 + ocw + 2) + 140,
                         w=(2 * (100 + ocw + 2) + 140) if angle < 45 else 2 * (40 + ocw) + 140, dimx=3, dimy=3,
                         margin=100 + ocw + 20, noise_dn=2, null_frac=0.02)
    H, W = c.i0.shape
    off, uv = api.get_uv_pivot(c.xyuvav, c.dt, c.mpp, c.ocw, H, W)
    assert np.diff(off).min() > 64
    with api.Context(0) as ctx:
        ctx.set_images(c.i0, c.i1)
        ctx.set_path(mode)
        got = ctx.matching_ncc_dlc_2(c.xyuvav, c.offset, off, uv, c.ocw)
        assert ctx.last_path() == {"auto": "u8_exact", "u16": "u16_scaled"}[mode]      # (the many-pivot forms of the register-tiled kernels: no tables, no matrix-core pass)
        sw = ctx.matching_ncc_dlc_2(c.xyuvav, -c.offset, off, -uv, c.ocw, swap=True)
        assert_bits_equal(got, oracle.match(c.i0, c.i1, c.xyuvav, c.offset, off, uv, c.ocw))
        assert_bits_equal(sw, oracle.match(c.i1, c.i0, c.xyuvav, -c.offset, off, -uv, c.ocw), "swapped")
        if mode == "auto":
            ctx.filter_images(api.CLI_KERNELS[0])
            f0, f1 = ctx.get_images(H, W)
            gf = ctx.matching_ncc_dlc_2(c.xyuvav, c.offset, off, uv, c.ocw)
            assert ctx.last_path() in ("u8_mfma", "u8_exact", "u8_offset", "u16_scaled")      # (small gradients can still be 8-bit)
            assert_bits_equal(gf, oracle.match(f0, f1, c.xyuvav, c.offset, off, uv, c.ocw), "d/dx")


def test_long_corridor_16bit_and_float_pairs(api, oracle):
    """the same corridor on imagery that has no integer kernel (16-bit DN, floats): previously 'match kernel launch:
    invalid argument', now the general kernel"""
    c = synth.make_small(seed=861, shift=(5, -5), angle_deg=45.0, ocw=40, speed=14100.0, h=344, w=352, dimx=3, dimy=3,
                         margin=120, noise_dn=300, null_frac=0.02, bits=16)
    H, W = c.i0.shape
    off, uv = api.get_uv_pivot(c.xyuvav, c.dt, c.mpp, c.ocw, H, W)
    rng = np.random.default_rng(3)
    f0 = (c.i0 * np.float32(0.0031) + np.where(c.i0 > 0, rng.random(c.i0.shape, dtype=np.float32), 0)).astype(np.float32)
    f1 = (c.i1 * np.float32(0.0031) + np.where(c.i1 > 0, rng.random(c.i1.shape, dtype=np.float32), 0)).astype(np.float32)
    with api.Context(0) as ctx:
        ctx.set_images(c.i0, c.i1)
        got = ctx.matching_ncc_dlc_2(c.xyuvav, c.offset, off, uv, c.ocw)
        assert ctx.last_path() == "general_f32"
        assert_bits_equal(got, oracle.match(c.i0, c.i1, c.xyuvav, c.offset, off, uv, c.ocw), "16-bit")
        ctx.set_images(f0, f1)
        gf = ctx.matching_ncc_dlc_2(c.xyuvav, c.offset, off, uv, c.ocw)
        assert ctx.last_path() == "general_f32"
    wf = oracle.match(f0, f1, c.xyuvav, c.offset, off, uv, c.ocw)
    assert np.array_equal(np.isnan(gf), np.isnan(wf)) and np.nanmax(np.abs(gf[:, :2] - wf[:, :2])) <= 1e-4
