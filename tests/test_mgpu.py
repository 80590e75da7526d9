"""GPU: the native multi-GPU driver (csrc/mgpu.cpp: one process, a host thread per device, RCCL communicator, ONE
ncclAllGather per call) against the single-context entry points.  On the one-GPU test box the communicator has one
rank: partition, padded blocks, the RCCL call, the un-permute kernel and the finish all run; with MIMC3_TEST_DEVICES=0,1,..
the same test covers a real multi-GPU node.  Bar: bit-identical (a point's result does not depend on the device)."""
import os
import subprocess
import sys

import numpy as np
import pytest

import fileio
from conftest import ROOT, assert_bits_equal, golden_files

pytestmark = pytest.mark.gpu

DEVICES = os.environ.get("MIMC3_TEST_DEVICES", "0")
CLI = os.path.join(ROOT, "mimc3_amd", "csrc", "MIMC3_hip")


@pytest.fixture(scope="module")
def results(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("mgpu") / "r.npz")
    env = dict(os.environ)
    env.pop("LD_PRELOAD", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "mgpu_worker.py"), out, DEVICES], env=env, capture_output=True,
                       text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    return np.load(out)


def test_sharded_matcher_pass_is_bit_identical(results):
    assert_bits_equal(results["match_mg"], results["match_one"], "forward")
    assert_bits_equal(results["match_mg_sw"], results["match_one_sw"], "swapped")
    assert (results["match_one"][:, 2] > 0.5).mean() > 0.5
    assert float(results["imbalance_match"]) <= 0.10


def test_sharded_program_is_bit_identical(results):
    assert tuple(results["mg_offset"]) == tuple(results["one_offset"]) == (3, -2)
    assert np.array_equal(results["mg_flag_cp"], results["one_flag_cp"])
    assert np.array_equal(results["mg_subint"].view(np.uint32), results["one_subint"].view(np.uint32))
    for k in ("vx", "vy", "ex", "ey", "qual"):
        assert_bits_equal(results["mg_" + k], results["one_" + k], k)
        assert_bits_equal(results["mg2_" + k], results["one_" + k], k + " (second call on the same communicator)")
    assert np.isfinite(results["one_vx"]).mean() > 0.9


def test_cli_with_device_list(tmp_path):
    """MIMC3_HIP_DEVICES=... takes the multi-GPU driver: same bytes as the reference-program golden"""
    if not os.path.exists(CLI):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "mimc3_amd", "csrc"), "cli"])
    z = np.load(golden_files("vmap_small")[0])
    t0, t1 = str(z["t0"]), str(z["t1"])
    d = str(tmp_path)
    fileio.write_tiff(f"{d}/{t0}_i0.tif", z["i0"].astype(np.uint8)); fileio.write_tiff(f"{d}/{t1}_i1.tif", z["i1"].astype(np.uint8))
    fileio.write_gma(f"{d}/xyuvav.GMA", z["xyuvav"])
    os.makedirs(f"{d}/out")
    p = subprocess.run([CLI, f"{d}/{t0}_i0.tif", f"{d}/{t1}_i1.tif", f"{d}/xyuvav.GMA", f"{d}/out"],
                       env=dict(os.environ, MIMC3_CP_SEED=str(int(z["seed"])), MIMC3_HIP_DEVICES=DEVICES), capture_output=True, text=True)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert "sharded over" in p.stdout
    r = fileio.read_vmap(f"{d}/out", t0, t1)
    for k in ("x", "y", "vx", "vy", "ex", "ey", "qual", "flagcp"):
        assert r[k].shape == z["out_" + k].shape and r[k].tobytes() == z["out_" + k].tobytes(), k
