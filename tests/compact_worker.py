"""Worker of tests/test_compact_cache.py: runs in its own process so that MIMC3_COMPACT=1 (read once per process by the launcher)
forces the compact LDS form (two-level NCC cache, 16-bit null lists) on every big-chip u8 launch.  Saves the matcher outputs."""
import sys

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from mimc3_amd import api, synth  # noqa: E402


def cases():
    # (name, case): null-ridden pairs, a CP offset, windows over the image edge, long climbs (a shift far along the corridor)
    yield "ocw30_nulls", synth.make_small(seed=71, shift=(3, -2), angle_deg=35.0, ocw=30, dimx=14, dimy=12, h=520, w=560, null_frac=0.06, noise_dn=2)
    yield "ocw40_nulls_offset", synth.make_small(seed=72, shift=(-2, 3), angle_deg=200.0, ocw=40, dimx=12, dimy=10, h=560, w=600, null_frac=0.05,
                                                noise_dn=3, offset=(2, -1))
    yield "ocw32_long_corridor", synth.make_small(seed=73, shift=(9, -9), angle_deg=45.0, ocw=32, dimx=12, dimy=10, h=640, w=640, null_frac=0.03,
                                                 noise_dn=2, speed=5200.0, margin=110)
    # cell grids beyond 64 x 64 (BASELINE C4's are 70 x 70): the compact form keeps the replay's visited set in registers up to
    # 96 x 128 (three column words per row, rows r and r + 64 on lane r), LDS bit words beyond; the regular form always LDS
    yield "ocw32_grid72x72", synth.make_small(seed=75, shift=(10, -10), angle_deg=45.0, ocw=32, dimx=10, dimy=8, h=700, w=700, null_frac=0.03,
                                             noise_dn=2, speed=6400.0, margin=125)
    yield "ocw30_grid8x110", synth.make_small(seed=76, shift=(0, -12), angle_deg=92.0, ocw=30, dimx=10, dimy=8, h=760, w=520, null_frac=0.03,
                                             noise_dn=2, speed=7200.0, margin=130)
    yield "ocw30_grid132x10", synth.make_small(seed=77, shift=(14, 0), angle_deg=2.0, ocw=30, dimx=8, dimy=8, h=520, w=860, null_frac=0.02,
                                              noise_dn=2, speed=9000.0, margin=140)
    yield "ocw30_edge_windows", synth.make_small(seed=74, shift=(2, 2), angle_deg=-60.0, ocw=30, dimx=12, dimy=10, h=420, w=440, null_frac=0.02,
                                                noise_dn=2, speed=2500.0, margin=33)


def main():
    res = {}
    for name, c in cases():
        H, W = c.i0.shape
        off, uv = api.get_uv_pivot(c.xyuvav, c.dt, c.mpp, c.ocw, H, W)
        with api.Context(0) as ctx:
            ctx.set_images(c.i0, c.i1)
            ctx.set_path("u8px")           # the register-tiled kernel alone: its compact LDS form is what is tested
            res[name + "_fw"] = ctx.matching_ncc_dlc_2(c.xyuvav, c.offset, off, uv, c.ocw)
            assert ctx.last_path() == "u8_exact"
            res[name + "_sw"] = ctx.matching_ncc_dlc_2(c.xyuvav, -c.offset, off, -uv, c.ocw, swap=True)
    np.savez(sys.argv[1], **res)


if __name__ == "__main__":
    main()
