#!/usr/bin/env python3
"""tools/pmc_to_json.py <gpurun_out/tag> <config> [round-dir] -- after tools/profile.sh ran on the GPU box and its output was
merged back: condense the PMC passes of the matcher kernels into profiles/traffic_latest.json (what bench.py quotes in
`roofline`: HBM traffic, VALU / SALU / LDS instructions per grid point, VALU busy fraction, LDS-array active fraction) and
copy the text summaries into profiles/<round-dir>/.  Every entry carries the sha16 of the kernel source it was measured on,
so a stale entry is visible in the bench line (`roofline.pmc_stale`)."""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import KERNEL_NAMES, kernel_source_sha16  # noqa: E402

out, config = sys.argv[1], sys.argv[2]
rnd = sys.argv[3] if len(sys.argv) > 3 and "=" not in sys.argv[3] else None
# occupancy of the headline kernel as launched (rocprofv3's dispatch record does not show dynamic LDS): lds=<bytes> vgpr=<count>
opts = dict(a.split("=", 1) for a in sys.argv[3:] if "=" in a)

# counters of the FULL-SIZE launches only: bench.py's io leg launches the same kernel on chunks of the grid
rows = []
for f in glob.glob(os.path.join(out, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        rows += list(csv.DictReader(fh))
full_grid = defaultdict(int)
for r in rows:
    full_grid[r["Kernel_Name"]] = max(full_grid[r["Kernel_Name"]], int(r["Grid_Size"]))
acc = defaultdict(lambda: defaultdict(list))
for r in rows:
    if int(r["Grid_Size"]) == full_grid[r["Kernel_Name"]]:
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
points, ocw = None, None
try:
    cfgj = json.load(open(os.path.join(out, "trace_bench.json")))["config"]
    points, ocw = cfgj["grid_points_rank0"], cfgj.get("ocw")
except Exception:
    pass

# occupancy per kernel from the dispatch records of the trace pass: workgroups per CU by LDS (160 KiB, 256-byte granules) and
# waves per SIMD by registers (512 VGPRs per lane)
occupancy = {}
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_trace.csv"), recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            try:
                lds = (int(r["LDS_Block_Size"]) + 255) // 256 * 256
                wg = int(r["Workgroup_Size"]) // 64
                vg = int(r["VGPR_Count"]) + int(r.get("Accum_VGPR_Count", 0) or 0)
                by_lds = (160 * 1024 // max(lds, 1)) * wg / 4.0
                by_vgpr = 512 // max(vg, 1)
                if int(r["LDS_Block_Size"]) > 0:          # (dynamic LDS shows as 0 here: no occupancy from such a record)
                    occupancy[r["Kernel_Name"]] = min(by_lds, by_vgpr, 8)
            except (KeyError, ValueError):
                pass

path = os.path.join(ROOT, "profiles", "traffic_latest.json")
db = json.load(open(path)) if os.path.exists(path) else {}
db.setdefault(config, {})
for kn, cs in acc.items():
    key = None
    for short in list(KERNEL_NAMES.values()) + ["match_ncc_dlc_px<PxF32i>"]:
        tag = short.replace("match_ncc_dlc_px<", "").rstrip(">")
        if ("match_ncc_dlc_px" in kn and f"PxCfg<mimc3::{tag}," in kn.replace(" ", "")) or (short == "match_ncc_dlc_f32" and "match_ncc_dlc_f32" in kn):
            key = short
    if key is None or "SQ_WAVES" not in cs:
        continue
    # the bench line's kernels are the config's chip size in their regular (not many-pivot) form; the `program` leg launches others
    import re
    m = re.search(r"PxCfg<mimc3::(\w+),(\d+),(\d+),(\d+),(\d+),(true|false),(true|false)", kn.replace(" ", ""))
    if ocw is not None and m and (int(m.group(2)) != int(ocw) or m.group(7) == "true"):
        continue
    mean = {c: sum(v) / len(v) for c, v in cs.items()}
    waves = mean["SQ_WAVES"]
    if waves < 1000:                     # list-mode launches behind the main one
        continue
    nw = 4 if ",64,4," in kn.replace(" ", "") else (2 if "PxF32" in kn and ",32,2," in kn.replace(" ", "") else 1)
    pts = points or waves / nw
    e = {"source": os.path.relpath(out, ROOT) + " (tools/profile.sh)", "kernel_sha16": kernel_source_sha16(), "kernel": kn[kn.find("PxCfg"):][:60],
         "waves_per_launch": waves}
    if "FETCH_SIZE" in mean and "WRITE_SIZE" in mean:
        e.update(fetch_size_kb=mean["FETCH_SIZE"], write_size_kb=mean["WRITE_SIZE"], bytes=int((2 * mean["FETCH_SIZE"] + mean["WRITE_SIZE"]) * 1024))
    for c, k in (("SQ_INSTS_VALU", "valu_per_point"), ("SQ_INSTS_SALU", "salu_per_point"), ("SQ_INSTS_LDS", "lds_per_point")):
        if c in mean:
            e[k] = mean[c] / pts
    # SQ_ACTIVE_INST_VALU and SQ_WAVE_CYCLES both count quad-cycles summed over waves: their ratio is the share of a wave's
    # resident time in which it has a VALU instruction executing; times the waves resident per SIMD (the occupancy the LDS carve
    # and the register count allow, from the dispatch record) it is the fraction of time the SIMD's VALU is busy
    occ = occupancy.get(kn)
    if key == KERNEL_NAMES["u8_exact"] and "lds" in opts:
        lds = (int(opts["lds"]) + 255) // 256 * 256
        occ = min((160 * 1024 // lds) * nw / 4.0, 512 // int(opts.get("vgpr", 128)))
    if "SQ_ACTIVE_INST_VALU" in mean and "SQ_WAVE_CYCLES" in mean:
        e["valu_active_share_of_wave_time"] = mean["SQ_ACTIVE_INST_VALU"] / mean["SQ_WAVE_CYCLES"]
        if occ:
            e["waves_per_simd"] = occ
            e["valu_busy"] = min(1.0, e["valu_active_share_of_wave_time"] * occ)
    if "SQ_LDS_IDX_ACTIVE" in mean and "SQ_WAVE_CYCLES" in mean:
        e["lds_active_share_of_wave_time"] = mean["SQ_LDS_IDX_ACTIVE"] / mean["SQ_WAVE_CYCLES"]
    for c, k in (("SQ_WAIT_ANY", "wait_share_of_wave_time"), ("SQ_WAIT_INST_ANY", "issue_stall_share_of_wave_time"), ("SQ_ACTIVE_INST_ANY", "issuing_share_of_wave_time")):
        if c in mean and "SQ_WAVE_CYCLES" in mean:
            e[k] = mean[c] / mean["SQ_WAVE_CYCLES"]
    db[config][key] = {**db[config].get(key, {}), **e}
    print(key, json.dumps(e))
json.dump(db, open(path, "w"), indent=2)
if rnd:
    dst = os.path.join(ROOT, "profiles", rnd)
    os.makedirs(dst, exist_ok=True)
    for name in ("SUMMARY.txt", "trace_bench.json"):
        src = os.path.join(out, name)
        if os.path.exists(src):
            shutil.copy(src, os.path.join(dst, f"bench_{config}_rocprofv3_{name.lower()}" if name == "SUMMARY.txt" else f"bench_{config}_under_rocprof.json"))
    for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
        shutil.copy(f, os.path.join(dst, f"bench_{config}_rocprofv3_kernel_stats.csv"))
