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
    if key is None or "SQ_WAVES" not in cs or opts.get("head") == "1":      # head=1: only the two-launch headline entry below is (re)written
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

# ---- the headline pass since round 4: two launches per step -- the matrix-core kernel (clean points; it flags the rest) and the
#      register-tiled kernel in flag mode (the flagged points).  Run tools/profile.sh with --no-legs --no-program --no-f32-path so that
#      every full-grid px<PxU8> launch of the run IS a flag-mode one.  One entry, "match_ncc_dlc_mx", with both kernels' records.
def full_size_stats(kn):
    """mean of every counter over the full-grid launches of kernel `kn`, + average duration of those launches in the trace pass"""
    mean = {c: sum(v) / len(v) for c, v in acc[kn].items()}
    durs = []
    for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_trace.csv"), recursive=True):
        with open(f) as fh:
            rws = [r for r in csv.DictReader(fh) if r["Kernel_Name"] == kn]
        g = max((int(r["Grid_Size_X"]) for r in rws), default=0)
        durs += [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rws if int(r["Grid_Size_X"]) == g]
    return mean, (sum(durs) / len(durs) / 1e6 if durs else None), len(durs)


N_SIMD, N_XCD = 1024, 8
mx_kn = [k for k in acc if "match_ncc_dlc_mx" in k and (ocw is None or f"Cfg<{ocw}," in k.replace(" ", "")) and "false,false" in k.replace(" ", "")]
px_kn = [k for k in acc if "match_ncc_dlc_px" in k and "PxCfg<mimc3::PxU8," in k.replace(" ", "") and (ocw is None or f"PxU8,{ocw}," in k.replace(" ", ""))]
if mx_kn and px_kn and points:
    clean_points = int(opts.get("clean_points", 0)) or None
    recs, tot_bytes, tot_ms = [], 0, 0.0
    for kn, role in ((mx_kn[0], "matrix-core kernel, clean form: every point enters, the null-free ones are matched, the others flagged"),
                     (px_kn[0], "register-tiled kernel in flag mode: the flagged points (nulls in window or chip)")):
        mean, ms, ncalls = full_size_stats(kn)
        r = {"kernel": kn[kn.find("match_ncc_dlc"):][:90], "role": role, "avg_ms_trace": ms, "launches_in_trace": ncalls}
        if "FETCH_SIZE" in mean and "WRITE_SIZE" in mean:
            r["bytes"] = int((2 * mean["FETCH_SIZE"] + mean["WRITE_SIZE"]) * 1024); tot_bytes += r["bytes"]
            r["write_size_kb"] = mean["WRITE_SIZE"]
        if ms:
            tot_ms += ms
        for c, k in (("SQ_INSTS_VALU", "valu_per_launched_point"), ("SQ_INSTS_SALU", "salu_per_launched_point"), ("SQ_INSTS_LDS", "lds_per_launched_point"),
                     ("SQ_INSTS_VALU_MFMA_I8", "mfma_i8_per_launched_point")):
            if c in mean:
                r[k] = mean[c] / points
        # busy fractions against the launch's own duration: GRBM_GUI_ACTIVE sums the 8 XCDs' clocks; SQ_ACTIVE_INST_VALU counts quad-cycles
        # summed over waves, SQ_VALU_MFMA_BUSY_CYCLES plain cycles summed over SIMDs (16 per v_mfma_i32_16x16x64_i8: checked
        # against the instruction count)
        if "GRBM_GUI_ACTIVE" in mean:
            cyc = mean["GRBM_GUI_ACTIVE"] / N_XCD
            if "SQ_ACTIVE_INST_VALU" in mean:
                r["valu_busy"] = 4.0 * mean["SQ_ACTIVE_INST_VALU"] / (N_SIMD * cyc)
            if "SQ_VALU_MFMA_BUSY_CYCLES" in mean:
                r["mfma_busy"] = mean["SQ_VALU_MFMA_BUSY_CYCLES"] / (N_SIMD * cyc)
            if "SQ_WAVE_CYCLES" in mean:
                r["waves_per_simd_resident_avg"] = 4.0 * mean["SQ_WAVE_CYCLES"] / (N_SIMD * cyc)
            if "SQ_LDS_IDX_ACTIVE" in mean:
                r["lds_busy"] = 4.0 * mean["SQ_LDS_IDX_ACTIVE"] / (N_SIMD * cyc) / 4.0      # one LDS per CU, four SIMDs
        for c, k in (("SQ_WAIT_ANY", "wait_share_of_wave_time"), ("SQ_WAIT_INST_ANY", "issue_stall_share_of_wave_time"), ("SQ_ACTIVE_INST_ANY", "issuing_share_of_wave_time")):
            if c in mean and "SQ_WAVE_CYCLES" in mean:
                r[k] = mean[c] / mean["SQ_WAVE_CYCLES"]
        for c in ("TCC_HIT_sum", "TCC_MISS_sum", "SQ_LDS_BANK_CONFLICT"):
            if c in mean:
                r[c.lower()] = mean[c]
        recs.append(r)
    cw = 2 * int(ocw) + 1 if ocw else None
    e = {"source": os.path.relpath(out, ROOT) + " (tools/profile.sh --no-legs --no-program --no-f32-path)", "kernel_sha16": kernel_source_sha16(),
         "kernels": recs, "bytes": tot_bytes or None, "kernel_ms_profile_avg": tot_ms or None,
         "valu_per_point": sum(r.get("valu_per_launched_point", 0) for r in recs), "salu_per_point": sum(r.get("salu_per_launched_point", 0) for r in recs),
         "lds_per_point": sum(r.get("lds_per_launched_point", 0) for r in recs)}
    vb = [r.get("valu_busy") for r in recs if r.get("valu_busy") is not None]
    mb = recs[0].get("mfma_busy")
    # what the counters say bounds the step: VALU issue when the VALU pipes of both launches are busy most of the time and neither the
    # matrix pipe nor HBM is
    if vb and min(vb) > 0.6 and (mb or 0) < 0.5:
        e["bound"] = "valu-issue"
    if cw and clean_points and "mfma_i8_per_launched_point" in recs[0]:
        mf = recs[0]["mfma_i8_per_launched_point"] * points / clean_points
        surface = cw * cw * 31 * 31
        e["compute"] = {
            "clean_points": clean_points, "flagged_points": points - clean_points,
            "mfma_i8_per_clean_point": mf, "mac_issued_per_clean_point": mf * 16 * 16 * 64,
            "useful_mac_per_point": surface,
            "useful_mac_what": f"sum a b of all 31 x 31 cells of a point's surface: {cw}^2 x 31^2 (the operand band and the 32 x 32 tile carry zeros besides; "
                               "the box sums of b and b^2 are MFMAs too)",
            "mfma_useful_frac": surface / (mf * 16 * 16 * 64),
            "mfma_busy": mb, "valu_per_point": e["valu_per_point"],
            "valu_per_clean_point_upper": recs[0]["valu_per_launched_point"] * points / clean_points,
            "useful_valu_frac": None,
            "useful_valu_what": "the products moved to the matrix pipe: the VALU stream of the clean form is staging, operand alignment, the f64 finish and the climb -- "
                                "no multiply-accumulate of the sums is left in it",
            "i8_mac_per_s_achieved": surface * clean_points / (recs[0]["avg_ms_trace"] * 1e-3) if recs[0].get("avg_ms_trace") else None,
            "i8_mac_per_s_peak": 2.5e15,
        }
    db[config]["match_ncc_dlc_mx"] = e
    print("match_ncc_dlc_mx", json.dumps(e)[:600])
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
