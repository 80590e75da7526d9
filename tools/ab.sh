#!/bin/bash
# tools/ab.sh <out-file> <workloads...> -- on the GPU box: tools/kbench.py for the regular build and every A/B build under
# mimc3_amd/csrc/variants/ (tools/build_variant.sh), one line per build and workload; --check N after the workloads verifies
# each against the reference on an N-point sample.
OUT=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $(dirname $OUT)
: > $OUT
for L in $ROOT/mimc3_amd/csrc/libmimc3_hip.so $ROOT/mimc3_amd/csrc/variants/libmimc3_hip_*.so; do
  [ -f $L ] || continue
  v=$(basename $L .so | sed 's/libmimc3_hip_\?//'); [ -z "$v" ] && v=main
  MIMC3_HIP_LIB=$L python3 $ROOT/tools/kbench.py "$@" 2>/dev/null | python3 -c "
import sys, json
for line in sys.stdin:
    r = json.loads(line)
    c = r.get('check', {})
    print('%-10s %-8s %8.3f ms (min %8.3f) %s' % ('$v', r['name'], r['ms'], r['ms_min'], ('identical' if c.get('bit_identical') else 'DIFFERENT') if c else ''))
" >> $OUT
done
cat $OUT
