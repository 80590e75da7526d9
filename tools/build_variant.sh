#!/bin/bash
# tools/build_variant.sh <name> [-DMACRO=.. ...] -- A/B builds of the matcher family: compiles match_px_kernel.hip with the given
# defines and links it with the objects of the regular build into mimc3_amd/csrc/variants/libmimc3_hip_<name>.so.
# Select it at run time with MIMC3_HIP_LIB=<path> (mimc3_amd/api.py, tools only).  Tuning infrastructure, not product.
set -e
NAME=$1; shift
HERE=$(cd "$(dirname "$0")/../mimc3_amd/csrc" && pwd)
make -C $HERE -j8 > /dev/null
mkdir -p $HERE/variants
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-result --offload-arch=gfx950 "$@" -x hip -c $HERE/match_px_kernel.hip -o $HERE/variants/px_$NAME.o
OBJS=$(ls $HERE/build/*.o | grep -v "match_px_kernel.hip.o\|gma_shim.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,-soname,libmimc3_hip.so -o $HERE/variants/libmimc3_hip_$NAME.so $OBJS $HERE/variants/px_$NAME.o -lpthread -ldl
echo "built $HERE/variants/libmimc3_hip_$NAME.so"
