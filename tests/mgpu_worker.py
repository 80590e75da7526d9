"""Worker of tests/test_mgpu.py.  Runs in its own process WITHOUT torch (mimc3_amd.api is ctypes + numpy only), so that
the RCCL this library loads (dlopen of librccl.so.1) is the only RCCL in the process.  Saves the single-context results
and the multi-GPU driver's results (communicator over the devices given on the command line) side by side."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from mimc3_amd import api, synth  # noqa: E402


def vmap_case(seed, shift, h=560, w=600, dimx=20, dimy=18, null_frac=0.02, noise=2, speed=900.0):
    i0, i1 = synth.make_pair(h, w, shift, seed=seed, null_frac=null_frac, noise_dn=noise)
    xy = synth.make_grid(dimx, dimy, 70, 70, (w - 140) // dimx, (h - 140) // dimy, speed, angle_deg=30.0)
    rng = np.random.default_rng(seed)
    slow = rng.random(dimx * dimy) < 0.5
    xy[slow, 4] = rng.uniform(-5, 5, slow.sum()); xy[slow, 5] = rng.uniform(-5, 5, slow.sum())
    return i0, i1, xy


def c3_case(out_path, devices):
    """BASELINE configs[2] at its real size: C2's 200,000 points sharded over the ranks -- one matcher pass and the whole
    data path, against the single-context entry points on the same inputs."""
    c = synth.make_case("C2")
    H, W = c.i0.shape
    off, uv = api.get_uv_pivot(c.xyuvav, c.dt, c.mpp, c.ocw, H, W)
    xv = np.array(c.xyuvav, np.float64, copy=True)          # 5 % slow points: the control-point stage needs candidates
    rng = np.random.default_rng(1)
    slow = rng.random(xv.shape[0]) < 0.05
    xv[slow, 4] = rng.uniform(-5, 5, slow.sum()); xv[slow, 5] = rng.uniform(-5, 5, slow.sum())
    res = {}
    i0r, i1r = c.i0.astype(np.uint8), c.i1.astype(np.uint8)
    with api.Context(devices[0]) as ctx:
        ctx.set_images(c.i0, c.i1)
        res["match_one"] = ctx.matching_ncc_dlc_2(c.xyuvav, c.offset, off, uv, c.ocw)
        one = ctx.vmap(xv, 16.0, cp_seed=7)
    with api.MultiGpu(devices, comm_lib=os.environ.get("MIMC3_TEST_COMM_LIB") or None, repeat_devices=os.environ.get("MIMC3_TEST_REPEAT") == "1") as mg:
        mg.set_images(i0r, i1r)
        res["match_mg"] = mg.matching_ncc_dlc_2(c.xyuvav, c.offset, off, uv, c.ocw)
        res["imbalance_match"] = np.float64(mg.last_imbalance())
        many = mg.vmap(xv, 16.0, cp_seed=7)
        res["imbalance_vmap"] = np.float64(mg.last_imbalance())
    for k in ("vx", "vy", "ex", "ey", "qual", "flag_cp"):
        res["one_" + k] = one[k]; res["mg_" + k] = many[k]
    res["one_offset"] = np.array(one["offset_cp"]); res["mg_offset"] = np.array(many["offset_cp"])
    res["collectives"] = np.int64(fake_collectives())
    np.savez(out_path, **res)


def fake_collectives():
    """completed all-gathers of the stand-in communicator (tests/fake_rccl.c), -1 when the real RCCL is in use"""
    lib = os.environ.get("MIMC3_TEST_COMM_LIB", "")
    if "fake_rccl" not in lib:
        return -1
    import ctypes
    h = ctypes.CDLL(lib)                                   # same handle the library dlopen'ed (one copy per process)
    h.fake_rccl_collectives.restype = ctypes.c_ulonglong
    return int(h.fake_rccl_collectives())


def main():
    out_path = sys.argv[1]
    devices = [int(d) for d in sys.argv[2].split(",")]
    assert "torch" not in sys.modules
    if len(sys.argv) > 3 and sys.argv[3] == "c3":
        return c3_case(out_path, devices)
    res = {}
    # (1) one matcher pass: a skewed a-priori so that the shares differ
    c = synth.make_small(seed=83, shift=(2, -3), angle_deg=40.0, ocw=15, dimx=40, dimy=36, h=520, w=560, null_frac=0.04, noise_dn=2)
    xy = c.xyuvav.copy()
    xy[: xy.shape[0] // 3, 4:6] *= 3.0
    H, W = c.i0.shape
    off, uv = api.get_uv_pivot(xy, c.dt, c.mpp, c.ocw, H, W)
    with api.Context(devices[0]) as ctx:
        ctx.set_images(c.i0, c.i1)
        res["match_one"] = ctx.matching_ncc_dlc_2(xy, c.offset, off, uv, c.ocw)
        res["match_one_sw"] = ctx.matching_ncc_dlc_2(xy, -c.offset, off, -uv, c.ocw, swap=True)
    i0, i1, vxy = vmap_case(seed=11, shift=(3, -2))
    with api.Context(devices[0]) as ctx:
        ctx.set_images(i0, i1)
        one = ctx.vmap(vxy, 16.0, cp_seed=7, num_cp_min=20)
    with api.MultiGpu(devices, comm_lib=os.environ.get("MIMC3_TEST_COMM_LIB") or None, repeat_devices=os.environ.get("MIMC3_TEST_REPEAT") == "1") as mg:
        mg.set_images(c.i0.astype(np.uint8), c.i1.astype(np.uint8))          # raw 8-bit DN, replicated
        res["match_mg"] = mg.matching_ncc_dlc_2(xy, c.offset, off, uv, c.ocw)
        res["imbalance_match"] = np.float64(mg.last_imbalance())
        res["match_mg_sw"] = mg.matching_ncc_dlc_2(xy, -c.offset, off, -uv, c.ocw, swap=True)
        mg.set_images(i0, i1)
        many = mg.vmap(vxy, 16.0, cp_seed=7, num_cp_min=20)
        again = mg.vmap(vxy, 16.0, cp_seed=7, num_cp_min=20)
        res["imbalance_vmap"] = np.float64(mg.last_imbalance())
    for k in ("vx", "vy", "ex", "ey", "qual", "flag_cp"):
        res["one_" + k] = one[k]; res["mg_" + k] = many[k]; res["mg2_" + k] = again[k]
    res["one_offset"] = np.array(one["offset_cp"]); res["mg_offset"] = np.array(many["offset_cp"])
    res["one_subint"] = np.array(one["cp_subint"], np.float32); res["mg_subint"] = np.array(many["cp_subint"], np.float32)
    res["collectives"] = np.int64(fake_collectives())
    np.savez(out_path, **res)


if __name__ == "__main__":
    main()
