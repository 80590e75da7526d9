import glob
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU restatement (oracle/mimc3_oracle.c), built on demand with gcc."""
    from oracle import oracle as orc
    if not orc.available("port"):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "oracle"])
    return orc.Oracle("port")


@pytest.fixture(scope="session")
def reference():
    """The compiled reference (oracle/_ref); skipped where it was never built."""
    from oracle import oracle as orc
    if not orc.available("reference"):
        if os.path.isdir("/root/reference"):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"])
        else:
            pytest.skip("oracle/_ref/libmimc3_ref.so not built (needs /root/reference)")
    return orc.Oracle("reference")


def golden_files(prefix):
    return sorted(glob.glob(os.path.join(GOLDEN, prefix + "*.npz")))


def load_match_golden(path):
    z = np.load(path)
    d = {k: z[k] for k in z.files}
    d["i0"] = d["i0"].astype(np.float32)
    d["i1"] = d["i1"].astype(np.float32)
    d["ocw"] = int(d["ocw"]); d["dt"] = float(d["dt"]); d["mpp"] = float(d["mpp"])
    return d


def assert_bits_equal(a, b, what=""):
    """Bit-exact float comparison, except that any NaN equals any NaN (payload/sign ignored)."""
    a = np.asarray(a, np.float32); b = np.asarray(b, np.float32)
    assert a.shape == b.shape, f"{what}: shape {a.shape} vs {b.shape}"
    na, nb = np.isnan(a), np.isnan(b)
    assert np.array_equal(na, nb), f"{what}: NaN masks differ at {np.argwhere(na != nb)[:5].tolist()}"
    ai = np.where(na, 0, a).view(np.uint32); bi = np.where(nb, 0, b).view(np.uint32)
    bad = np.argwhere(ai != bi)
    assert bad.size == 0, f"{what}: {len(bad)} elements differ, first {bad[:5].tolist()} a={a[tuple(bad[0])]!r} b={b[tuple(bad[0])]!r}"
