// pivot_kernel.h -- device half of get_uv_pivot (internal; see pivot_kernel.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mimc3 {

// one grid point's corridor as the host computes it (layout of mimc3::CorridorPOD, host_util.h): 24 bytes
struct CorridorDev { float incr_u, incr_v, norm_incr; double length; };
static_assert(sizeof(CorridorDev) == 24, "corridor record is 24 bytes");

// counts + CSR offsets.  d_ext6 (8-byte aligned): [0] max pivots per point, [1] [2] max |last pivot| per axis, [3] != 0 if a
// point has no pivot, [4..5] the total as int64 -- one 24-byte read-back sizes the pivot buffer and the matcher launch.
// (the points' (u, v) are read at d_xyuvav[xy_stride * g + xy_col + {0, 1}]: 6 / 2 for xyuvav rows, 2 / 0 for a packed [N][2] array)
hipError_t launch_pivot_count(const double *d_xyuvav, int xy_stride, int xy_col, const CorridorDev *d_cor, int N, int ocw, int H, int W, int32_t *d_cnt,
                              int64_t *d_piv_off, int32_t *d_ext6, hipStream_t s);
// the lists (and, when d_uvn is given, the negated copy of MIMC_main.c:272-279)
hipError_t launch_pivot_fill(const CorridorDev *d_cor, const int64_t *d_piv_off, int N, int32_t *d_uv, int32_t *d_uvn, hipStream_t s);

}  // namespace mimc3
