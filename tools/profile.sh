#!/bin/bash
# tools/profile.sh <tag> [bench args...] -- run on the GPU box (via gpurun) from the repo root.
# Pass 1: rocprofv3 --kernel-trace --stats (per-kernel durations).
# Pass 2..: PMC counters, each in its own run (FETCH_SIZE and WRITE_SIZE cannot share a pass on
# gfx950; never combined with --sys-trace etc.).  Summaries land in gpurun_out/<tag>/.
set -o pipefail
TAG=${1:-prof}; shift
ARGS=${@:---steps 5 --warmup 2 --no-cpu-baseline}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 $ROOT/bench.py $ARGS > $OUT/trace_bench.json 2> $OUT/trace.err || { echo "trace pass failed"; tail -5 $OUT/trace.err; exit 1; }
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_I8 SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
  name=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc_$name -o pmc -- python3 $ROOT/bench.py $ARGS > /dev/null 2> $OUT/pmc_$name.err || echo "pmc pass $name failed"
done
python3 $ROOT/tools/summarize_prof.py $OUT > $OUT/SUMMARY.txt 2>&1
cat $OUT/SUMMARY.txt
