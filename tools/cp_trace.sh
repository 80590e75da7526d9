#!/bin/bash
# tools/cp_trace.sh -- on the GPU box: rocprofv3 kernel trace of the control-point stage alone; prints the kernel time line of
# the last call (start ms, duration ms, hardware queue)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r2/cpprof
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o cp -- python3 $ROOT/tools/cp_profile.py 3 > $OUT/run.log 2>&1
cd $ROOT
python3 - <<PY
import csv
rows=list(csv.DictReader(open("gpurun_out/r2/cpprof/cp_kernel_trace.csv")))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
t0=int(rows[0]["Start_Timestamp"])
# the last CP call = everything after the last cp_count_invalid
last=max(i for i,r in enumerate(rows) if "cp_count_invalid" in r["Kernel_Name"])
for r in rows[last:]:
    n=r["Kernel_Name"].replace("mimc3::","").replace("void match_ncc_dlc_","").replace("(anonymous namespace)::","")
    if "rocclr" in n: continue
    print("%-58.58s %9.3f %7.3f q%s" % (n, (int(r["Start_Timestamp"])-t0)/1e6, (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6, r.get("Queue_Id","")))
PY
rm -f gpurun_out/r2/cpprof/cp_kernel_trace.csv
