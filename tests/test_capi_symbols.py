"""CPU: libmimc3_hip.so loads and exports every function include/mimc3_hip.h declares; host-side
geometry (pivots, neighbour offsets) matches the oracle bit for bit; no compute call is made."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import ROOT, golden_files, load_match_golden

LIB = os.path.join(ROOT, "mimc3_amd", "csrc", "libmimc3_hip.so")


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(LIB):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "mimc3_amd", "csrc"), "-j4"])
    return ctypes.CDLL(LIB)


def test_every_declared_symbol_is_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "mimc3_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = sorted(set(re.findall(r"\b(mimc3_[a-z0-9_]+)\s*\(", hdr)))
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/mimc3_hip.h but not exported"


def test_import_api_loads_library(lib):
    from mimc3_amd import api
    assert "gfx950" in api.version()


def test_pivots_match_oracle_on_golden(lib, oracle):
    from mimc3_amd import api
    for path in golden_files("match_"):
        g = load_match_golden(path)
        H, W = g["i0"].shape
        off, uv = api.get_uv_pivot(g["xyuvav"], g["dt"], g["mpp"], g["ocw"], H, W)
        assert np.array_equal(off, g["piv_off"]) and np.array_equal(uv, g["piv_uv"]), path
        assert api.pivot_extent(off, uv) == (int((off[1:] - off[:-1]).max()),
                                             int(np.abs(uv[off[1:] - 1, 0]).max()), int(np.abs(uv[off[1:] - 1, 1]).max()))


def test_pivots_all_directions_vs_oracle(lib, oracle):
    from mimc3_amd import api
    n = 720
    ang = np.deg2rad(np.arange(n) * 0.5 - 180.0)
    spd = 200.0 + 37.0 * (np.arange(n) % 101)
    xy = np.zeros((n, 6))
    xy[:, 2] = 400 + np.arange(n) % 7 + 0.25
    xy[:, 3] = 380 + np.arange(n) % 5 + 0.75
    xy[:, 4] = spd * np.cos(ang)
    xy[:, 5] = spd * np.sin(ang)
    for ocw in (7, 16, 40):
        a = api.get_uv_pivot(xy, 16.0, 15.0, ocw, 800, 830)
        b = oracle.get_uv_pivot(xy, 16.0, 15.0, ocw, 800, 830)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def test_pivots_large_grid_threaded_vs_oracle(lib, oracle):
    """N >= 20000 takes the multi-threaded host path: same CSR as the serial oracle, every quadrant and speed"""
    from mimc3_amd import api
    n = 60000
    rng = np.random.default_rng(3)
    ang = rng.uniform(-np.pi, np.pi, n)
    spd = rng.uniform(0.0, 5000.0, n)
    xy = np.zeros((n, 6))
    xy[:, 2] = rng.uniform(45, 1955, n); xy[:, 3] = rng.uniform(45, 1755, n)       # some close to the edge: clipped corridors
    xy[:, 4] = spd * np.cos(ang); xy[:, 5] = spd * np.sin(ang)
    a = api.get_uv_pivot(xy, 16.0, 15.0, 40, 1800, 2000)
    b = oracle.get_uv_pivot(xy, 16.0, 15.0, 40, 1800, 2000)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def test_zero_pivot_point_is_refused(lib):
    from mimc3_amd import api
    xy = np.array([[0, 0, 3.0, 3.0, 100.0, 100.0]])       # closer than ocw to the edge -> no pivot
    with pytest.raises(api.Mimc3Error) as e:
        api.get_uv_pivot(xy, 16.0, 15.0, 7, 100, 100)
    assert e.value.code == -2


def test_ruv_neighbor_vs_oracle(lib, oracle):
    from mimc3_amd import api, synth
    for dimx, dimy, rad in [(40, 40, 5.0), (57, 33, 3.0), (12, 9, 5.0)]:
        xy = synth.make_grid(dimx, dimy, 60, 60, 20, 20, 1500.0)
        mps = float(np.float32(xy[1, 0] - xy[0, 0]))
        assert np.array_equal(api.get_ruv_neighbor(xy, dimx, dimy, mps, rad), oracle.get_ruv_neighbor(xy, dimx, dimy, mps, rad))
