"""GPU: the MIMC3_hip command line (mimc3_amd/csrc/MIMC3_hip, the reference's main() over libmimc3_hip.so) on TIFF and
.GMA files against the reference-program golden fixture -- all eight .GMA outputs and meta.txt byte for byte -- and,
where the compiled reference program travelled to this box, against a live run of it; plus main()'s early exits."""
import os
import subprocess

import numpy as np
import pytest

import fileio
from conftest import ROOT, golden_files
from mimc3_amd import synth

pytestmark = pytest.mark.gpu

CLI = os.path.join(ROOT, "mimc3_amd", "csrc", "MIMC3_hip")
PROG = os.path.join(ROOT, "oracle", "_ref", "MIMC3_ref")
OUTS = ("x", "y", "vx", "vy", "ex", "ey", "qual", "flagcp")


@pytest.fixture(scope="module")
def cli():
    if not os.path.exists(CLI):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "mimc3_amd", "csrc"), "cli"])
    assert os.path.exists(CLI), "MIMC3_hip was not built (libtiff header missing?)"
    return CLI


def write_inputs(d, i0, i1, xy, t0, t1, dtype=np.uint8):
    fileio.write_tiff(f"{d}/{t0}_i0.tif", i0.astype(dtype)); fileio.write_tiff(f"{d}/{t1}_i1.tif", i1.astype(dtype))
    fileio.write_gma(f"{d}/xyuvav.GMA", xy)
    os.makedirs(f"{d}/out", exist_ok=True)
    return [f"{d}/{t0}_i0.tif", f"{d}/{t1}_i1.tif", f"{d}/xyuvav.GMA", f"{d}/out"]


def test_cli_vs_program_golden(cli, tmp_path):
    z = np.load(golden_files("vmap_small")[0])
    t0, t1 = str(z["t0"]), str(z["t1"])
    args = write_inputs(str(tmp_path), z["i0"], z["i1"], z["xyuvav"], t0, t1)
    p = subprocess.run([cli] + args, env=dict(os.environ, MIMC3_CP_SEED=str(int(z["seed"]))), capture_output=True, text=True)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    r = fileio.read_vmap(args[3], t0, t1)
    for k in OUTS:
        assert r[k].shape == z["out_" + k].shape and r[k].tobytes() == z["out_" + k].tobytes(), k
    meta = dict(zip(z["meta_keys"].tolist(), z["meta_vals"].tolist()))
    for k, v in meta.items():
        assert r["meta"][k] == v, k
    assert r["meta"]["name_i0"] == args[0] and r["meta"]["name_i1"] == args[1] and len(r["meta"]) == 7


@pytest.mark.skipif(not os.path.exists(PROG), reason="oracle/_ref/MIMC3_ref did not travel to this box")
def test_cli_vs_program_live_16bit(cli, tmp_path):
    h, w, dimx, dimy = 380, 440, 15, 11
    i0, i1 = synth.make_pair(h, w, (-2, 3), seed=92, null_frac=0.05, noise_dn=4, bits=16)
    xy = synth.make_grid(dimx, dimy, 70, 70, (w - 140) // dimx, (h - 140) // dimy, 1200.0, angle_deg=-50.0)
    rng = np.random.default_rng(92)
    slow = rng.random(dimx * dimy) < 0.6
    xy[slow, 4] = rng.uniform(-5, 5, slow.sum()); xy[slow, 5] = rng.uniform(-5, 5, slow.sum())
    t0, t1 = "20210301000000", "20210309120000"
    os.makedirs(tmp_path / "a"); os.makedirs(tmp_path / "b")
    a = write_inputs(str(tmp_path / "a"), i0, i1, xy, t0, t1, np.uint16)
    b = write_inputs(str(tmp_path / "b"), i0, i1, xy, t0, t1, np.uint16)
    subprocess.run([PROG] + a, check=True, env=dict(os.environ, MIMC3_REF_SEED="5"), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    p = subprocess.run([cli] + b, env=dict(os.environ, MIMC3_CP_SEED="5"), capture_output=True, text=True)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    for k in OUTS + ("meta.txt",):
        name = f"vmap_{t0}_{t1}_{k}" + ("" if k.endswith(".txt") else ".GMA")
        fa, fb = open(f"{a[3]}/{name}", "rb").read(), open(f"{b[3]}/{name}", "rb").read()
        if k == "meta.txt":       # the two name_ lines carry the (different) input paths
            fa = b"\n".join(l for l in fa.split(b"\n") if not l.startswith(b"name_"))
            fb = b"\n".join(l for l in fb.split(b"\n") if not l.startswith(b"name_"))
        assert fa == fb, name


HYBRID = os.path.join(ROOT, "oracle", "_ref", "MIMC3_main_on_hip")


@pytest.mark.skipif(not (os.path.exists(PROG) and os.path.exists(HYBRID)), reason="oracle/_ref programs did not travel to this box")
def test_reference_main_on_the_shim_vs_reference_program(tmp_path):
    """INTEGRATION.md option A taken to the end: the reference's OWN main() + GMA.c + MIMC_misc.c linked against
    libmimc3_gma_shim.a instead of MIMC_module.c (every module function main() calls is then the GPU implementation,
    through the reference's exact struct-level signatures) writes the same bytes as the reference program."""
    h, w, dimx, dimy = 360, 400, 13, 11
    i0, i1 = synth.make_pair(h, w, (1, -2), seed=94, null_frac=0.03, noise_dn=3)
    xy = synth.make_grid(dimx, dimy, 70, 70, (w - 140) // dimx, (h - 140) // dimy, 1000.0, angle_deg=70.0)
    rng = np.random.default_rng(94)
    slow = rng.random(dimx * dimy) < 0.6
    xy[slow, 4] = rng.uniform(-5, 5, slow.sum()); xy[slow, 5] = rng.uniform(-5, 5, slow.sum())
    t0, t1 = "20230501000000", "20230513000000"
    os.makedirs(tmp_path / "a"); os.makedirs(tmp_path / "b")
    a = write_inputs(str(tmp_path / "a"), i0, i1, xy, t0, t1)
    b = write_inputs(str(tmp_path / "b"), i0, i1, xy, t0, t1)
    subprocess.run([PROG] + a, check=True, env=dict(os.environ, MIMC3_REF_SEED="9"), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    p = subprocess.run([HYBRID] + b, env=dict(os.environ, MIMC3_CP_SEED="9"), capture_output=True, text=True)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    for k in OUTS:
        name = f"vmap_{t0}_{t1}_{k}.GMA"
        assert open(f"{a[3]}/{name}", "rb").read() == open(f"{b[3]}/{name}", "rb").read(), name


def test_cli_early_exits(cli, tmp_path):
    """vmap.tar already there -> skip (MIMC_main.c:122-130); no control points -> empty vmap.tar (:246-252); both -1"""
    i0, i1 = synth.make_pair(300, 320, (1, 1), seed=93)
    xy = synth.make_grid(8, 7, 70, 70, 22, 22, 900.0, angle_deg=10.0)        # nothing slow: no CP candidate
    t0, t1 = "20220101000000", "20220102000000"
    args = write_inputs(str(tmp_path), i0, i1, xy, t0, t1)
    tar = f"{args[3]}/vmap_{t0}_{t1}.tar"
    p = subprocess.run([cli] + args, capture_output=True, text=True)
    assert p.returncode == 255 and os.path.exists(tar) and os.path.getsize(tar) == 0, p.stdout[-800:] + p.stderr[-800:]
    assert sorted(os.listdir(args[3])) == [os.path.basename(tar)]
    p = subprocess.run([cli] + args, capture_output=True, text=True)
    assert p.returncode == 255 and "already exists" in p.stdout
    p = subprocess.run([cli, "nodir.tif", args[1], args[2], args[3]], capture_output=True, text=True)
    assert p.returncode == 2                                                    # no '/', no timestamp: refused
