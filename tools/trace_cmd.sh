#!/bin/bash
# tools/trace_cmd.sh <tag> <python script + args...> -- on the GPU box: rocprofv3 kernel trace of one command; prints the matcher kernels' times.
set -o pipefail
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 $ROOT/"$@" > $OUT/trace.log 2>&1 || echo "trace pass failed"
python3 $ROOT/tools/summarize_prof.py $OUT > $OUT/SUMMARY.txt 2>&1
grep "grid=\|VGPR" $OUT/SUMMARY.txt | grep "px<\|mx::"
