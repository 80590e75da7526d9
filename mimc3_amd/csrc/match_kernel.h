// match_kernel.h -- launch interface of the matcher kernels (internal to libmimc3_hip.so).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mimc3 {

constexpr int kMatchThreads = 256;   // 4 wave64 per grid point

struct MatchArgs {
    const float *i0, *i1;        // device images [H][W]
    int32_t H, W;
    const double *xyuvav;        // device [N][6]
    int32_t N;
    int32_t off_u, off_v;        // CP offset, added to the search centre only (MIMC_module.c:827-828)
    const int32_t *piv_uv;       // device CSR payload [P][2]
    const int64_t *piv_off;      // device CSR offsets [N+1]
    int32_t ocw;
    int32_t swap;                // 0: chip from i0, window from i1; 1: exchanged
    float thr;                   // smallest f32 whose f64 value is >= MIN_DN (1e-10, MIMC_module.c:21)
    float *out;                  // device [N][3]
    // LDS carve (floats / pivots), filled by the launcher from the per-launch maxima
    int32_t lds_chip_f, lds_win_f, lds_cell_f, lds_npiv;
};

// max_abs_u/v: max over points of |last pivot| per axis; max_npiv: max pivots per point.
hipError_t launch_match_f32(MatchArgs a, int max_abs_u, int max_abs_v, int max_npiv, hipStream_t stream);

}  // namespace mimc3
