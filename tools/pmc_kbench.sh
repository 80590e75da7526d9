#!/bin/bash
# tools/pmc_kbench.sh <tag> <workloads...> -- on the GPU box: phase clocks (MIMC3_U8_STATS) and two rocprofv3 PMC passes of
# tools/kbench.py for the named workloads.  Output under gpurun_out/<tag>/.
set -o pipefail
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp KBENCH_REPS=2
cd /tmp
MIMC3_U8_STATS=1 python3 $ROOT/tools/kbench.py "$@" > $OUT/stats.log 2>&1 || { tail -5 $OUT/stats.log; exit 1; }
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  name=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc_$name -o pmc -- python3 $ROOT/tools/kbench.py "$@" > $OUT/pmc_$name.log 2>&1 || echo "pmc pass $name failed"
done
python3 $ROOT/tools/summarize_prof.py $OUT > $OUT/SUMMARY.txt 2>&1
grep -v "rocclr\|at::native\|prep_\|widen\|conv2\|range_tiles\|detect_" $OUT/SUMMARY.txt
grep "mimc3 u8 stats\|\"name\"" $OUT/stats.log
find $OUT -name "*counter_collection.csv" -delete
