#!/bin/bash
# tools/pmc_cmd.sh <tag> <python script + args...> -- on the GPU box: a kernel trace and two rocprofv3 PMC passes of one command.
# Output under gpurun_out/<tag>/ (SUMMARY.txt: per-kernel time and counters).
set -o pipefail
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 $ROOT/"$@" > $OUT/trace.log 2>&1 || echo "trace pass failed"
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_I8 SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"; do
  name=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc_$name -o pmc -- python3 $ROOT/"$@" > $OUT/pmc_$name.log 2>&1 || echo "pmc pass $name failed"
done
python3 $ROOT/tools/summarize_prof.py $OUT > $OUT/SUMMARY.txt 2>&1
grep -v "rocclr\|at::native\|prep_\|widen\|conv2\|range_tiles\|detect_" $OUT/SUMMARY.txt | head -60
find $OUT -name "*counter_collection.csv" -size +20M -delete
