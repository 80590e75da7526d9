#!/bin/bash
# tools/pmc_quick.sh <tag> [bench args] -- one PMC pass with the SQ instruction-mix counters only
TAG=${1:-q}; shift
ARGS=${@:---steps 3 --warmup 1 --no-cpu-baseline}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SMEM SQ_INSTS_VMEM_RD --output-format csv -d $OUT/pmc_mix -o pmc -- python3 $ROOT/bench.py $ARGS > $OUT/bench.json 2> $OUT/err.txt
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS --output-format csv -d $OUT/pmc_wait -o pmc -- python3 $ROOT/bench.py $ARGS > /dev/null 2>> $OUT/err.txt
python3 $ROOT/tools/summarize_prof.py $OUT | grep -v "copyBuffer\|fillBuffer\|prep_u8"
cut -c1-330 $OUT/bench.json
