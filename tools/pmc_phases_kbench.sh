#!/bin/bash
# tools/pmc_phases_kbench.sh <workload> -- like tools/pmc_phases.sh, for one tools/kbench.py workload (u8_40, ddx_40, lap_40 ...):
# VALU / SALU / LDS instructions per workgroup (= per grid point) with the kernel cut short after each phase.
WL=${1:-u8_40}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_phases_$WL
mkdir -p $OUT
export TMPDIR=/tmp KBENCH_REPS=1
cd /tmp
for k in 1 2 3 4 7 6 0; do
  MIMC3_U8_DEBUG_STOP=$k rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d $OUT/stop$k -o pmc -- python3 $ROOT/tools/kbench.py $WL > /dev/null 2> $OUT/stop$k.err || echo "stop $k failed"
  python3 - "$OUT/stop$k/pmc_counter_collection.csv" $k <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    if "match_ncc_dlc_px" in r["Kernel_Name"]:
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for kn, a in acc.items():
    w = max(a["SQ_WAVES"])
    if w < 1000: continue
    nw = 4 if ",64,4," in kn.replace(" ", "") or "64, 4" in kn else 1
    print("stop", sys.argv[2], kn[kn.find("PxCfg"):kn.find("PxCfg") + 40], {k: round(max(v) / w * nw) for k, v in a.items() if k != "SQ_WAVES"}, "per point")
PY
done
find $OUT -name "*.csv" -delete
