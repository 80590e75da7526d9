#!/bin/bash
# tools/pmc_phases.sh -- VALU/SALU/LDS instruction counts of the headline matcher kernel per PHASE: the kernel is cut
# short after phase k (MIMC3_U8_DEBUG_STOP=k) and rocprofv3 counts what ran.  Run on the GPU box from the repo root.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_phases
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for k in 1 2 3 4 7 6 0; do     # 1 window staged, 2 + chip, 3 + certain-set requests, 4 + their evaluation, 7 + speculative climb (and its evaluations), 6 + replay, 0 + fit
  MIMC3_U8_DEBUG_STOP=$k rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d $OUT/stop$k -o pmc -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-program --no-f32-path --qm-sweeps 0 > /dev/null 2> $OUT/stop$k.err || echo "stop $k failed"
  python3 - "$OUT/stop$k/pmc_counter_collection.csv" $k <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(list)
for r in rows:
    if "match_ncc_dlc_px" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
w = sum(acc["SQ_WAVES"]) / len(acc["SQ_WAVES"])
print("stop", sys.argv[2], {k: round(sum(v) / len(v) / w) for k, v in acc.items() if k != "SQ_WAVES"}, "per wave")
PY
done
