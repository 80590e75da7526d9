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
