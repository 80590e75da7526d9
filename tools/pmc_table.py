#!/usr/bin/env python3
"""tools/pmc_table.py <gpurun_out/tag> [points] -- one line per matcher kernel of a tools/profile.sh run: full-grid launches only,
average duration (trace pass), VALU / SALU / LDS / MFMA instructions per grid point, HBM bytes per launch (FETCH_SIZE x 2 +
WRITE_SIZE, gfx950 correction), WRITE_SIZE alone, VALU and MFMA busy fractions.  Text for profiles/<round>/."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
points = int(sys.argv[2]) if len(sys.argv) > 2 else 200000


def short(name):
    n = name.replace("void ", "").replace("mimc3::", "").replace("(anonymous namespace)::", "")
    n = n.replace("match_ncc_dlc_px<PxCfg<", "px<").replace("> >(MatchU8Args)", ">").replace("(MatchArgs)", "").replace("match_ncc_dlc_mx<mx::Cfg<", "mx<")
    return n.replace(", ", ",")


rows = []
for f in glob.glob(os.path.join(out, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        rows += [r for r in csv.DictReader(fh) if "match_ncc_dlc" in r["Kernel_Name"]]
full = defaultdict(int)
for r in rows:
    full[r["Kernel_Name"]] = max(full[r["Kernel_Name"]], int(r["Grid_Size"]))
acc = defaultdict(lambda: defaultdict(list))
for r in rows:
    if int(r["Grid_Size"]) == full[r["Kernel_Name"]]:
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = defaultdict(list)
res = {}
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_trace.csv"), recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            if r["Kernel_Name"] in full and int(r["Grid_Size_X"]) == full[r["Kernel_Name"]]:
                dur[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
                res[r["Kernel_Name"]] = (r["VGPR_Count"], r["Scratch_Size"], r["LDS_Block_Size"])
print("# matcher kernels of %s: full-grid launches, per grid point = / %d (launched points; flag- and list-mode launches work on fewer)" % (out, points))
print("%-44s %5s %9s %8s %8s %7s %7s %10s %9s %6s %6s %5s %7s" % ("kernel", "calls", "avg ms", "VALU/pt", "SALU/pt", "LDS/pt", "MFMA/pt", "HBM MB", "WRITE MB", "VALUb", "MFMAb", "VGPR", "scratch"))
for kn in sorted(acc, key=lambda k: short(k)):
    m = {c: sum(v) / len(v) for c, v in acc[kn].items()}
    if m.get("SQ_WAVES", 0) < 1000:
        continue
    d = dur.get(kn, [])
    cyc = m.get("GRBM_GUI_ACTIVE", 0) / 8
    vb = 4 * m.get("SQ_ACTIVE_INST_VALU", 0) / (1024 * cyc) if cyc else float("nan")
    mb = m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (1024 * cyc) if cyc else float("nan")
    hbm = (2 * m.get("FETCH_SIZE", 0) + m.get("WRITE_SIZE", 0)) * 1024 / 1e6
    vg, sc, _ = res.get(kn, ("?", "?", "?"))
    print("%-44.44s %5d %9.3f %8.0f %8.0f %7.0f %7.1f %10.1f %9.2f %6.2f %6.2f %5s %7s" % (
        short(kn)[:44], len(d), (sum(d) / len(d) / 1e6) if d else float("nan"), m.get("SQ_INSTS_VALU", 0) / points, m.get("SQ_INSTS_SALU", 0) / points,
        m.get("SQ_INSTS_LDS", 0) / points, m.get("SQ_INSTS_VALU_MFMA_I8", 0) / points, hbm, m.get("WRITE_SIZE", 0) * 1024 / 1e6, vb, mb, vg, sc))
