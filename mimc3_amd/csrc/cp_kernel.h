// cp_kernel.h -- device pieces of the control-point offset stage (internal).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mimc3 {

// number of pixels below 1e-5 in the (2*ocw+1)^2 chip of `img` around each centre (MIMC_module.c:83-112)
hipError_t launch_cp_count_invalid(const float *img, int32_t H, int32_t W, const int32_t *uv, int32_t n, int32_t ocw,
                                   int32_t *counts, hipStream_t stream);
// tile t of `atlas` ([n*cs][cs], cs = 2*half+1) = the chip of `img` around centre t (:238-251)
hipError_t launch_cp_extract(const float *img, int32_t H, int32_t W, const int32_t *uv, int32_t n, int32_t half, float *atlas,
                             hipStream_t stream);
// raw kh x kw stencil (null DN poisons, :2545) of the (cs+2)^2 chip around centre t into tmp[t] ([cs+2][cs+2], only the
// stencil's interior is written) and the minimum of its non-NaN values into imin[t] (1e37 if none)
hipError_t launch_cp_conv_min(const float *img, int32_t H, int32_t W, const int32_t *uv, int32_t n, int32_t half, const float *k,
                              int32_t kh, int32_t kw, float *tmp, float *imin, hipStream_t stream);
// tile t of `atlas` = tmp[t][1..cs][1..cs] shifted by mn[t] (NaN -> 0, else v-(mn-1); :2572-2580, :296-303)
hipError_t launch_cp_shift_copy(const float *tmp, const float *mn, int32_t n, int32_t half, float *atlas, hipStream_t stream);
// (du, dv) -> (-du, -dv) on an [n][3] matcher output (swapped pass, :376-377)
hipError_t launch_negate_uv(float *out, int32_t n, hipStream_t stream);

// out[i] = -in[i]: the pivots of a swapped pass (MIMC_main.c:272-279)
hipError_t launch_negate_i32(const int32_t *in, int32_t *out, int64_t n, hipStream_t stream);

// xy [n][6], piv [n][(2*awc+1)^2][2], poff [n+1] of the control-point stage's matcher problem (tile t of the chip atlas)
hipError_t launch_cp_fill_problem(double *xy, int32_t *piv, int64_t *poff, int32_t n, int32_t awc, int32_t half, int32_t cs, hipStream_t stream);

// multi-GPU re-assembly of all-gathered blocks: g [world][npass][per][3], perm [world*per] (grid index, -1 = padding)
// -> out [npass][N][3]
hipError_t launch_scatter_blocks(const float *g, const int32_t *perm, int32_t world, int32_t per, int32_t npass, int32_t N, float *out,
                                 hipStream_t stream);

}  // namespace mimc3
