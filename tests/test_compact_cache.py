"""GPU: the compact LDS form of the big-chip kernels (two-level NCC cache: 16-bit entry per cell + value slots; 16-bit null
lists), which the launcher takes by itself only where it buys an occupancy step (BASELINE C4's 133^2 windows: covered by
tests/test_fullsize_gpu.py), forced here onto ordinary ocw 30 / 32 / 40 launches with MIMC3_COMPACT=1 in a worker process.
Bar: bit-identical to the oracle (find_ncc_peak's cmap semantics incl. -2.0 for unvisited cells, MIMC_module.c:677-681)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, assert_bits_equal

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("compact", ["1", "0"])
def test_forced_compact_form_is_bit_identical(oracle, tmp_path, compact):
    """compact = 1: the compact form on every launch; 0: never (the same cases on the regular form: its big cell grids keep the
    visited set of the exact replay in LDS)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import compact_worker
    out = str(tmp_path / "r.npz")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "compact_worker.py"), out], env=dict(os.environ, MIMC3_COMPACT=compact, MIMC3_LDS_DEBUG="1"),
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    r = np.load(out)
    from mimc3_amd import api
    for name, c in compact_worker.cases():
        H, W = c.i0.shape
        off, uv = api.get_uv_pivot(c.xyuvav, c.dt, c.mpp, c.ocw, H, W)
        want = oracle.match(c.i0, c.i1, c.xyuvav, c.offset, off, uv, c.ocw)
        assert_bits_equal(r[name + "_fw"], want, name + " forward")
        assert (want[:, 2] > 0.3).mean() > 0.5, name
        assert_bits_equal(r[name + "_sw"], oracle.match(c.i1, c.i0, c.xyuvav, -c.offset, off, -uv, c.ocw), name + " swapped")


@pytest.mark.parametrize("ocw", [30, 32])
def test_four_wave_forms_when_the_high_occupancy_forms_are_off(ocw):
    """Round 4: the u8 / u8-through-offsets kernels at ocw 30 / 32 have a second instantiation with a higher occupancy target, taken
    whenever the launch's LDS need admits it -- i.e. on every small test case.  MIMC3_HIGH_OCC=0 (read once per process) keeps the
    four-wave forms, which large windows still get: both the raw pair and its d/dx planes against the oracle."""
    import textwrap
    code = textwrap.dedent("""
        import sys, numpy as np
        sys.path.insert(0, %r); sys.path.insert(0, %r + "/tests")
        from conftest import assert_bits_equal
        from mimc3_amd import api, synth
        from oracle import oracle as orc
        o = orc.Oracle("port")
        ocw = %d
        c = synth.make_small(seed=7100 + ocw, shift=(-2, 3), angle_deg=25.0, ocw=ocw, speed=1700.0, h=2 * ocw + 270, w=2 * ocw + 290, dimx=5, dimy=5,
                             noise_dn=12, null_frac=0.04)            # (noise 12 DN: the gradient planes need more than 8 bits globally, not locally)
        H, W = c.i0.shape
        off, uv = api.get_uv_pivot(c.xyuvav, c.dt, c.mpp, ocw, H, W)
        with api.Context(0) as ctx:
            ctx.set_images(c.i0, c.i1)
            ctx.set_path("u8px")
            got = ctx.matching_ncc_dlc_2(c.xyuvav, c.offset, off, uv, ocw)
            assert ctx.last_path() == "u8_exact"
            assert_bits_equal(got, o.match(c.i0, c.i1, c.xyuvav, c.offset, off, uv, ocw), "u8, four-wave form")
            ctx.set_path("auto")
            paths = []
            for k in (0, 1):                                       # d/dx, d/dy: at least one of them needs the per-point offsets
                ctx.filter_images(None)
                ctx.filter_images(api.CLI_KERNELS[k])
                f0, f1 = ctx.get_images(H, W)
                got = ctx.matching_ncc_dlc_2(c.xyuvav, c.offset, off, uv, ocw)
                paths.append(ctx.last_path())
                assert_bits_equal(got, o.match(f0, f1, c.xyuvav, c.offset, off, uv, ocw), "gradient planes, four-wave form")
            assert "u8_offset" in paths, paths
    """ % (ROOT, ROOT, ocw))
    subprocess.check_call([sys.executable, "-c", code], env=dict(os.environ, MIMC3_HIGH_OCC="0"), timeout=600)
