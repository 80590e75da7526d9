#!/bin/bash
# tools/san_tests.sh -- the CPU-only tests against the sanitizer build of the host side (make -C mimc3_amd/csrc san):
# AddressSanitizer + UndefinedBehaviorSanitizer over capi.cpp, host_geometry.cpp, pipeline.cpp, mgpu.cpp through the entry points
# that need no device (pivots, corridors, point costs, partition, the symbol table, the struct-level shim's symbols).
# The interpreter is not instrumented: the runtime is preloaded, and leak checking is off (CPython's own allocations would drown it).
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
[ -f "$ROOT/mimc3_amd/csrc/build_san/libmimc3_hip.so" ] || make -C "$ROOT/mimc3_amd/csrc" san
cd "$ROOT"
LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:abort_on_error=1:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
  MIMC3_HIP_LIB=$ROOT/mimc3_amd/csrc/build_san/libmimc3_hip.so \
  python3 -m pytest tests -q -m "not gpu" -k "capi or pivot or partition or shim or shard" "$@"
