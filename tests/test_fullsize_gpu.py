"""GPU: BASELINE.json's configurations at their real shapes, where the driver's `pytest -m gpu` sees them.

  C2  4096^2 pair, 200,000 grid points, ocw 16: every kernel path against the compiled reference (oracle/_ref travels
      to the GPU box prebuilt; the C restatement stands in where it did not), bit-exact for the integer-DN pair;
      a float pair of the same size within north_star's 1e-4 px on the tiled f32 and the general kernel.
  C4  8192^2 pair, 1,000,000 grid points, ocw 32 (65^2 chip, 31 pivots, 129^2 window): all points on the GPU, an evenly
      spaced sample of them against the reference, plus a 2048^2 / 62,500-point case of the same shape in full.
  whole program  ~1024^2 pair, 10,700 grid points: live `MIMC3_ref` (the unmodified reference program) against the
      `MIMC3_hip` command line, every output file byte for byte.

The reference's loop being checked: MIMC_module.c:805-842 (matcher), MIMC_main.c:203-447 (program)."""
import os
import subprocess

import numpy as np
import pytest

import fileio
from conftest import ROOT, assert_bits_equal
from mimc3_amd import synth

pytestmark = pytest.mark.gpu

CLI = os.path.join(ROOT, "mimc3_amd", "csrc", "MIMC3_hip")
PROG = os.path.join(ROOT, "oracle", "_ref", "MIMC3_ref")


@pytest.fixture(scope="module")
def api():
    from mimc3_amd import api as a
    return a


@pytest.fixture(scope="module")
def checker():
    """the compiled reference where it travelled here, else the parity-pinned restatement"""
    from oracle import oracle as orc
    if orc.available("reference"):
        return orc.Oracle("reference")
    if not orc.available("port"):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "oracle"])
    return orc.Oracle("port")


def subset(piv_off, piv_uv, idx):
    cnt = (piv_off[idx + 1] - piv_off[idx]).astype(np.int64)
    off = np.zeros(len(idx) + 1, np.int64)
    np.cumsum(cnt, out=off[1:])
    sel = np.repeat(piv_off[idx] - off[:-1], cnt) + np.arange(off[-1])
    return off, np.ascontiguousarray(piv_uv[sel])


# ---------------------------------------------------------------------------------------------------------------------
# C2
# ---------------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def c2(api, checker):
    c = synth.make_case("C2")
    H, W = c.i0.shape
    off, uv = api.get_uv_pivot(c.xyuvav, c.dt, c.mpp, c.ocw, H, W)
    want = checker.match(c.i0, c.i1, c.xyuvav, c.offset, off, uv, c.ocw)
    return c, off, uv, want


@pytest.mark.parametrize("mode", ["auto", "u8px", "u16", "f32", "general"])
def test_c2_all_200k_points_vs_reference(api, c2, mode):
    c, off, uv, want = c2
    assert c.n == 200000
    with api.Context(0) as ctx:
        ctx.set_images(c.i0, c.i1)
        ctx.set_path(mode)
        got = ctx.matching_ncc_dlc_2(c.xyuvav, c.offset, off, uv, c.ocw)
        assert ctx.last_path() == {"auto": "u8_mfma", "u8px": "u8_exact", "u16": "u16_scaled", "f32": "f32_tiled", "general": "general_f32"}[mode]
    assert_bits_equal(got, want, f"C2 {mode}")
    ok = got[:, 2] > 0.5
    assert ok.mean() > 0.9 and abs(np.median(got[ok, 0]) - 4) < 0.1 and abs(np.median(got[ok, 1]) + 4) < 0.1


def test_c2_fused_pivot_entry_chunked(api, c2):
    """mimc3_match_ncc_dlc_geo / _cor at C2 size: corridors from the host, pivot lists made on the device, the grid pipelined through
    in four chunks (uploads and downloads under the matcher) -- the same bits as the reference on all 200,000 points"""
    c, off, uv, want = c2
    with api.Context(0) as ctx:
        ctx.set_images(c.i0, c.i1)
        got = ctx.matching_ncc_dlc_geo(c.xyuvav, c.offset, c.dt, c.mpp, c.ocw)
        cor = api.pivot_corridors(c.xyuvav, c.dt, c.mpp)
        got2 = ctx.matching_ncc_dlc_cor(c.xyuvav, cor, c.offset, c.ocw, out=api.pinned_empty((c.n, 3), np.float32))
    assert_bits_equal(got, want, "C2 fused entry")
    assert_bits_equal(np.array(got2), want, "C2 fused entry, corridors given, pinned result")


def test_c2_swapped_pass_vs_reference(api, checker, c2):
    """the CLI's second call per chip size: images exchanged, offset and pivots negated (MIMC_main.c:272-293)"""
    c, off, uv, _ = c2
    idx = np.arange(0, c.n, 4)
    soff, suv = subset(off, uv, idx)
    xy = np.ascontiguousarray(c.xyuvav[idx])
    want = checker.match(c.i1, c.i0, xy, -c.offset, soff, -suv, c.ocw)
    with api.Context(0) as ctx:
        ctx.set_images(c.i0, c.i1)
        got = ctx.matching_ncc_dlc_2(xy, -c.offset, soff, -suv, c.ocw, swap=True)
    assert_bits_equal(got, want, "C2 swapped")


@pytest.mark.parametrize("mode", ["f32", "general"])
def test_c2_float_pair_within_1e4_px(api, checker, c2, mode):
    """Non-integer imagery at C2 size: the f64 sums depend on the summation order in their last bits.
    north_star: |d(u,v)| <= 1e-4 px and the same invalid mask."""
    c, off, uv, _ = c2
    rng = np.random.default_rng(2)
    f0 = (c.i0 * np.float32(0.731) + np.where(c.i0 > 0, rng.random(c.i0.shape, dtype=np.float32), 0)).astype(np.float32)
    f1 = (c.i1 * np.float32(0.731) + np.where(c.i1 > 0, rng.random(c.i1.shape, dtype=np.float32), 0)).astype(np.float32)
    want = checker.match(f0, f1, c.xyuvav, c.offset, off, uv, c.ocw)
    with api.Context(0) as ctx:
        ctx.set_images(f0, f1)
        ctx.set_path(mode)
        got = ctx.matching_ncc_dlc_2(c.xyuvav, c.offset, off, uv, c.ocw)
        assert ctx.last_path() == ("f32_tiled" if mode == "f32" else "general_f32")
    assert np.array_equal(np.isnan(got), np.isnan(want))
    assert np.array_equal(got[:, 2] == -3.0, want[:, 2] == -3.0)
    assert np.nanmax(np.abs(got[:, :2] - want[:, :2])) <= 1e-4       # px
    assert np.nanmax(np.abs(got[:, 2] - want[:, 2])) <= 1e-6


def test_c2_16bit_pair_bit_identical(api, checker, c2):
    """16-bit DN (what Landsat-8 pairs are): the reference's f32 products round, its f64 sums of those integers are
    exact in any order -- bit-identical on the tiled f32 kernel."""
    c, off, uv, _ = c2
    i0, i1 = synth.make_pair(4096, 4096, c.shift, 20260102, noise_dn=40, null_frac=0.02, bits=16)
    idx = np.arange(0, c.n, 2)
    soff, suv = subset(off, uv, idx)
    xy = np.ascontiguousarray(c.xyuvav[idx])
    want = checker.match(i0, i1, xy, c.offset, soff, suv, c.ocw)
    with api.Context(0) as ctx:
        ctx.set_images(i0, i1)
        got = ctx.matching_ncc_dlc_2(xy, c.offset, soff, suv, c.ocw)
        assert ctx.last_path() == "f32_tiled"
    assert_bits_equal(got, want, "C2 16-bit")


# ---------------------------------------------------------------------------------------------------------------------
# C4
# ---------------------------------------------------------------------------------------------------------------------
def test_c4_shape_2048_all_points(api, checker):
    """C4's per-point shape (ocw 32, 45 degrees, 6,101 m/yr -> 31 pivots, 129^2 window) on a 2048^2 pair, 62,500 points."""
    i0, i1 = synth.make_pair(2048, 2048, (12, -12), 20260104, noise_dn=2, null_frac=0.02)
    xy = synth.make_grid(250, 250, 100, 100, 7, 7, 6101.0, perturb=0.1)
    off, uv = api.get_uv_pivot(xy, 16.0, 15.0, 32, 2048, 2048)
    npiv = off[1:] - off[:-1]
    assert npiv.max() >= 31 and xy.shape[0] == 62500
    zero = np.zeros(2, np.int32)
    want = checker.match(i0, i1, xy, zero, off, uv, 32)
    with api.Context(0) as ctx:
        ctx.set_images(i0, i1)
        for mode, path in (("auto", "u8_mfma"), ("u8px", "u8_exact"), ("u16", "u16_scaled"), ("general", "general_f32")):      # (31 pivots: wider than the matrix-core kernel's tile -- every point passes through it and is flagged)
            ctx.set_path(mode)
            got = ctx.matching_ncc_dlc_2(xy, zero, off, uv, 32)
            assert ctx.last_path() == path
            assert_bits_equal(got, want, f"C4 shape {mode}")
    ok = got[:, 2] > 0.5
    assert ok.mean() > 0.9 and abs(np.median(got[ok, 0]) - 12) < 0.1 and abs(np.median(got[ok, 1]) + 12) < 0.1


def test_c4_full_size_sampled_vs_reference(api, checker):
    """BASELINE configs[3] on ONE GPU: 8192^2 pair, 1,000,000 grid points, 65^2 chip, windows up to 133^2.  Every point
    runs on the GPU; an evenly spaced 60,000-point sample is compared with the reference (the CPU needs ~115 s for all)."""
    c = synth.make_case("C4")
    H, W = c.i0.shape
    assert (H, W) == (8192, 8192) and c.n == 1000000
    off, uv = api.get_uv_pivot(c.xyuvav, c.dt, c.mpp, c.ocw, H, W)
    with api.Context(0) as ctx:
        ctx.set_images(c.i0, c.i1)
        got = ctx.matching_ncc_dlc_2(c.xyuvav, c.offset, off, uv, c.ocw)
        assert ctx.last_path() == "u8_mfma"            # (the matrix-core kernel flags every point: corridors wider than its tile)
    idx = np.unique(np.linspace(0, c.n - 1, 60000).astype(np.int64))
    soff, suv = subset(off, uv, idx)
    want = checker.match(c.i0, c.i1, np.ascontiguousarray(c.xyuvav[idx]), c.offset, soff, suv, c.ocw)
    assert_bits_equal(got[idx], want, "C4 sample")
    ok = got[:, 2] > 0.5
    assert ok.mean() > 0.9 and abs(np.median(got[ok, 0]) - 12) < 0.1 and abs(np.median(got[ok, 1]) + 12) < 0.1


# ---------------------------------------------------------------------------------------------------------------------
# whole program, live
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.skipif(not os.path.exists(PROG), reason="oracle/_ref/MIMC3_ref did not travel to this box")
@pytest.mark.parametrize("bits", [8, 16])
def test_whole_program_live_10k_points(tmp_path, bits):
    """TIFF + xyuvav.GMA in, eight .GMA files + meta.txt out: the unmodified reference program against MIMC3_hip."""
    if not os.path.exists(CLI):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "mimc3_amd", "csrc"), "cli"])
    h, w, dimx, dimy = 1024, 1088, 107, 100
    i0, i1 = synth.make_pair(h, w, (3, -2), seed=20260110 + bits, null_frac=0.02, noise_dn=2 if bits == 8 else 300, bits=bits)
    xy = synth.make_grid(dimx, dimy, 70, 70, 8, 8, 1500.0, angle_deg=40.0)
    assert xy.shape[0] >= 10000
    rng = np.random.default_rng(bits)
    slow = rng.random(dimx * dimy) < 0.05
    xy[slow, 4] = rng.uniform(-5, 5, slow.sum()); xy[slow, 5] = rng.uniform(-5, 5, slow.sum())
    t0, t1 = "20240301000000", "20240317000000"
    dtype = np.uint8 if bits == 8 else np.uint16
    runs = {}
    for name in ("ref", "hip"):
        d = tmp_path / name
        os.makedirs(d / "out")
        fileio.write_tiff(f"{d}/{t0}_i0.tif", i0.astype(dtype)); fileio.write_tiff(f"{d}/{t1}_i1.tif", i1.astype(dtype))
        fileio.write_gma(f"{d}/xyuvav.GMA", xy)
        runs[name] = [f"{d}/{t0}_i0.tif", f"{d}/{t1}_i1.tif", f"{d}/xyuvav.GMA", f"{d}/out"]
    subprocess.run([PROG] + runs["ref"], check=True, env=dict(os.environ, MIMC3_REF_SEED="11"), stdout=subprocess.DEVNULL,
                   stderr=subprocess.DEVNULL)
    p = subprocess.run([CLI] + runs["hip"], env=dict(os.environ, MIMC3_CP_SEED="11"), capture_output=True, text=True)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    for k in ("x", "y", "vx", "vy", "ex", "ey", "qual", "flagcp"):
        name = f"vmap_{t0}_{t1}_{k}.GMA"
        fa, fb = open(f"{runs['ref'][3]}/{name}", "rb").read(), open(f"{runs['hip'][3]}/{name}", "rb").read()
        assert fa == fb, name
    ma = [l for l in open(f"{runs['ref'][3]}/vmap_{t0}_{t1}_meta.txt") if not l.startswith("name_")]
    mb = [l for l in open(f"{runs['hip'][3]}/vmap_{t0}_{t1}_meta.txt") if not l.startswith("name_")]
    assert ma == mb
    vx = fileio.read_gma(f"{runs['hip'][3]}/vmap_{t0}_{t1}_vx.GMA", np.float32)
    assert np.isfinite(vx).mean() > 0.9


# ---------------------------------------------------------------------------------------------------------------------
# C5: match -> cluster -> dpf0 / dpf1 -> QM capped at 10 sweeps, on the 500x400 grid
# ---------------------------------------------------------------------------------------------------------------------
def test_c5_qm_after_a_real_200k_point_match(api, checker, c2):
    """BASELINE configs[4]: the C2 pair, 200,000 grid points -> the CLI's eight raw-image passes (ocw 7/15/30/40, forward
    and swapped, MIMC_main.c:261-298) on the GPU -> mimc3_postprocess (clustering, dpf0, dpf1, QM with the sweep cap at 10,
    MIMC_module.c:893-991 / :1986-2312) on the device, against the same chain of the CPU checker fed the same candidates.
    The matcher half is checked on a sample of every pass (all 200,000 points of the ocw-16 pass are checked above); the
    post-processing half runs on the compiled reference where it travelled here, the QM with the cap on the pinned
    restatement (the reference has no cap parameter; when the capped run converges early both must agree)."""
    from oracle import oracle as orcmod
    c, _, _, _ = c2
    H, W = c.i0.shape
    n = c.n
    # ~30 % of the second image carries a decoy texture (independent of the first image), in 48-px blocks: the
    # eight candidates of the points there disagree, no cluster reaches the 0.6 quality of get_dpf0, and the QM pass has work
    rng = np.random.default_rng(5)
    decoy = synth.texture(H, W, 424242)
    i1 = c.i1.copy()
    for by, bx in zip(rng.integers(0, H - 48, 2200), rng.integers(0, W - 48, 2200)):
        i1[by:by + 48, bx:bx + 48] = decoy[by:by + 48, bx:bx + 48]
    port = orcmod.Oracle("port") if orcmod.available("port") else checker
    dp = np.empty((8, n, 3), np.float32)
    zero = np.zeros(2, np.int32)
    with api.Context(0) as ctx:
        ctx.set_images(c.i0, i1)
        for k, ocw in enumerate((7, 15, 30, 40)):
            off, uv = api.get_uv_pivot(c.xyuvav, c.dt, c.mpp, ocw, H, W)
            dp[2 * k] = ctx.matching_ncc_dlc_2(c.xyuvav, zero, off, uv, ocw)
            sw = ctx.matching_ncc_dlc_2(c.xyuvav, zero, off, -uv, ocw, swap=True)
            idx = np.sort(rng.choice(n, 1500, replace=False))
            soff, suv = subset(off, uv, idx)
            xs = np.ascontiguousarray(c.xyuvav[idx])
            assert_bits_equal(dp[2 * k][idx], checker.match(c.i0, i1, xs, zero, soff, suv, ocw), f"C5 pass ocw {ocw}")
            assert_bits_equal(sw[idx], checker.match(i1, c.i0, xs, zero, soff, -suv, ocw), f"C5 swapped pass ocw {ocw}")
            sw[:, :2] = -sw[:, :2]                                        # :288-293
            dp[2 * k + 1] = sw
        mps = float(np.float32(c.xyuvav[1, 0] - c.xyuvav[0, 0]))
        got = ctx.mimc2_postprocess(dp, c.xyuvav, c.dimx, c.dimy, c.dt, c.mpp, mps, qm_max_sweeps=10)
    # ---- the same chain on the CPU
    mvn, nclus = checker.cluster_candidates(dp, kmax=32)
    d0 = checker.get_dpf0(mvn, nclus, c.dimx, c.dimy, 0.6)
    d1, x1, y1 = checker.get_dpf1(d0, checker.get_ruv_neighbor(c.xyuvav, c.dimx, c.dimy, mps, 3.0), mvn, nclus, c.xyuvav, c.dt, c.mpp)
    ruv2 = checker.get_ruv_neighbor(c.xyuvav, c.dimx, c.dimy, mps, 5.0)
    d2, _, _, st = port.qm(d1, x1, y1, ruv2, mvn, nclus, c.xyuvav, max_sweeps=10)
    assert st is not None and 1 <= int(st[0]) <= 10 and int(st[2]) == 0          # sweeps run; no point hit the T7 definition
    initial = (mvn[np.arange(n), np.maximum(d1.reshape(-1), 0), 4] < 0.6) & (d1.reshape(-1) >= 0)
    assert initial.sum() > 1000, "the QM pass has points to investigate"
    if checker.kind == "reference" and int(st[0]) < 10:                              # converged under the cap: the uncapped reference agrees
        dr = checker.qm(d1, x1, y1, ruv2, mvn, nclus, c.xyuvav)[0]
        assert np.array_equal(dr, d2)
    d2 = d2.reshape(-1)
    want = np.full((5, n), np.nan, np.float32)
    ok = d2 >= 0
    want[:, ok] = mvn[np.arange(n)[ok], d2[ok], :].T
    assert_bits_equal(got.reshape(5, n), want, "C5 planes (du, dv, var_u, var_v, quality)")
    assert ok.mean() > 0.9 and (d2 != d1.reshape(-1)).sum() > 0, "the QM pass changed some picks"
