// cp_kernel.hip -- device pieces of get_offset_image (MIMC_module.c:33-492) for gfx950: chip validity counts,
// the per-point image chips ("atlas": tile t = chip of control-point candidate t, stacked vertically so that the
// matcher kernel can treat the stack as one tall image), and the chip-local pre-filter.  The matching itself is the
// general matcher kernel run on the atlas with the full-square search area (MatchArgs::win_half).
//
// The reference filters every chip with GMA_float_conv2 into ONE output plane that it reuses from point to point
// (:259-262, :292-293); what survives in that plane's border takes part in the next chip's minimum.  The stencil and
// the per-chip interior minimum are computed here for all chips at once; the tiny sequential recurrence over the
// chips (border state -> minimum -> shift) runs on the host between the two launches (capi.cpp).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <stdint.h>
#include "cp_kernel.h"

namespace mimc3 {
namespace {

__global__ __launch_bounds__(256) void cp_count_invalid(const float *__restrict__ img, int32_t H, int32_t W,
                                                        const int32_t *__restrict__ uv, int32_t ocw, int32_t *__restrict__ counts)
{
    const int t = blockIdx.x;
    const int u = uv[2 * t], v = uv[2 * t + 1];
    const int cw = 2 * ocw + 1;
    int bad = 0;
    for (int q = threadIdx.x; q < cw * cw; q += blockDim.x) {
        const int r = q / cw, c = q - r * cw;
        bad += ((double)img[(size_t)(v - ocw + r) * W + u - ocw + c] < 0.00001) ? 1 : 0;     // :97 (f32 vs f64 constant)
    }
    __shared__ int total;
    if (threadIdx.x == 0) total = 0;
    __syncthreads();
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) bad += __shfl_xor(bad, o, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(&total, bad);
    __syncthreads();
    if (threadIdx.x == 0) counts[t] = total;
}

__global__ __launch_bounds__(256) void cp_extract(const float *__restrict__ img, int32_t H, int32_t W, const int32_t *__restrict__ uv,
                                                  int32_t half, float *__restrict__ atlas)
{
    const int t = blockIdx.x;
    const int u = uv[2 * t], v = uv[2 * t + 1];
    const int cs = 2 * half + 1;
    float *tile = atlas + (size_t)t * cs * cs;
    for (int q = threadIdx.x; q < cs * cs; q += blockDim.x) {
        const int r = q / cs, c = q - r * cs;
        tile[q] = img[(size_t)(v - half + r) * W + u - half + c];
    }
}

struct ConvK { float k[9]; int32_t kh, kw; };

__global__ __launch_bounds__(256) void cp_conv_min(const float *__restrict__ img, int32_t H, int32_t W, const int32_t *__restrict__ uv,
                                                   int32_t half, ConvK kk, float *__restrict__ tmp, float *__restrict__ imin)
{
    const int t = blockIdx.x;
    const int u = uv[2 * t], v = uv[2 * t + 1];
    const int ts = 2 * half + 3, h1 = half + 1;
    const int ox = kk.kw / 2, oy = kk.kh / 2;
    float *plane = tmp + (size_t)t * ts * ts;
    float mn = 1e+37f;
    for (int q = threadIdx.x; q < ts * ts; q += blockDim.x) {
        const int r = q / ts, c = q - r * ts;
        if (r < oy || r >= ts - oy || c < ox || c >= ts - ox) continue;
        float s = 0.0f;
        for (int i = 0; i < kk.kh; i++)
            for (int j = 0; j < kk.kw; j++) {
                const float p = img[(size_t)(v - h1 + r + i - oy) * W + u - h1 + c + j - ox];
                const float dn = (p > -1.5f && p < 0.5f) ? __builtin_nanf("") : p;      // :2545: (int32_t)(p + 0.5) == 0
                s += dn * kk.k[i * kk.kw + j];
            }
        plane[q] = s;
        if (s < mn) mn = s;                                                   // false for NaN (:2559)
    }
    __shared__ float part[4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const float x = __shfl_xor(mn, o, 64); mn = x < mn ? x : mn; }
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = mn;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; w++) mn = part[w] < mn ? part[w] : mn;
        imin[t] = mn;
    }
}

__global__ __launch_bounds__(256) void cp_shift_copy(const float *__restrict__ tmp, const float *__restrict__ mnv, int32_t half,
                                                     float *__restrict__ atlas)
{
    const int t = blockIdx.x;
    const int cs = 2 * half + 1, ts = cs + 2;
    const float mn = mnv[t];
    const float *plane = tmp + (size_t)t * ts * ts;
    float *tile = atlas + (size_t)t * cs * cs;
    for (int q = threadIdx.x; q < cs * cs; q += blockDim.x) {
        const int r = q / cs, c = q - r * cs;
        const float x = plane[(r + 1) * ts + c + 1];
        tile[q] = (x != x) ? 0.0f : x - (mn - 1.0f);
    }
}

__global__ __launch_bounds__(256) void negate_uv(float *out, int32_t n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[3 * (size_t)i] = -out[3 * (size_t)i];
    out[3 * (size_t)i + 1] = -out[3 * (size_t)i + 1];
}

__global__ __launch_bounds__(256) void negate_i32(const int32_t *__restrict__ in, int32_t *__restrict__ out, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = -in[i];
}

// multi-GPU re-assembly: the all-gathered per-rank blocks [world][npass][per][3] (rank r's point j = grid point perm[r*per+j],
// -1 = padding) -> the grid-ordered tensor out [npass][N][3]
__global__ __launch_bounds__(256) void scatter_blocks(const float *__restrict__ g, const int32_t *__restrict__ perm, int32_t world,
                                                      int32_t per, int32_t npass, int32_t N, float *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)world * per) return;
    const int32_t dst = perm[i];
    if (dst < 0) return;
    const int32_t r = (int32_t)(i / per), j = (int32_t)(i - (int64_t)r * per);
    for (int32_t ps = 0; ps < npass; ps++) {
        const float *src = g + (((size_t)r * npass + ps) * per + j) * 3;
        float *o = out + ((size_t)ps * N + dst) * 3;
        o[0] = src[0]; o[1] = src[1]; o[2] = src[2];
    }
}

}  // namespace

hipError_t launch_scatter_blocks(const float *g, const int32_t *perm, int32_t world, int32_t per, int32_t npass, int32_t N, float *out,
                                 hipStream_t stream)
{
    const int64_t n = (int64_t)world * per;
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(scatter_blocks, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, g, perm, world, per, npass, N, out);
    return hipGetLastError();
}

// the matcher problem of the control-point stage, identical for every candidate: tile t of the chip atlas is a grid point at
// (half, t*cs + half) with the (2*awc+1)^2 pivot set (MIMC_module.c:150-162) -- xy [n][6], piv [n][npiv][2], poff [n+1]
__global__ __launch_bounds__(256) void cp_fill_problem(double *__restrict__ xy, int32_t *__restrict__ piv, int64_t *__restrict__ poff,
                                                         int n, int awc, int half, int cs)
{
    const int side = 2 * awc + 1, npiv = side * side;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < (int64_t)n * npiv) {
        const int k = (int)(i % npiv);
        piv[2 * i] = k / side - awc; piv[2 * i + 1] = k % side - awc;
    }
    if (i <= n) poff[i] = (int64_t)npiv * i;
    if (i < n) {
        double *r = xy + 6 * i;
        r[0] = 0.0; r[1] = 0.0; r[2] = (double)half; r[3] = (double)i * cs + half; r[4] = 0.0; r[5] = 0.0;
    }
}

hipError_t launch_cp_fill_problem(double *xy, int32_t *piv, int64_t *poff, int32_t n, int32_t awc, int32_t half, int32_t cs, hipStream_t stream)
{
    const int64_t npiv = (int64_t)(2 * awc + 1) * (2 * awc + 1), work = std::max<int64_t>(n * npiv, (int64_t)n + 1);
    hipLaunchKernelGGL(cp_fill_problem, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, stream, xy, piv, poff, n, awc, half, cs);
    return hipGetLastError();
}

hipError_t launch_negate_i32(const int32_t *in, int32_t *out, int64_t n, hipStream_t stream)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(negate_i32, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, in, out, n);
    return hipGetLastError();
}

hipError_t launch_cp_count_invalid(const float *img, int32_t H, int32_t W, const int32_t *uv, int32_t n, int32_t ocw, int32_t *counts,
                                   hipStream_t stream)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(cp_count_invalid, dim3(n), dim3(256), 0, stream, img, H, W, uv, ocw, counts);
    return hipGetLastError();
}

hipError_t launch_cp_extract(const float *img, int32_t H, int32_t W, const int32_t *uv, int32_t n, int32_t half, float *atlas,
                             hipStream_t stream)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(cp_extract, dim3(n), dim3(256), 0, stream, img, H, W, uv, half, atlas);
    return hipGetLastError();
}

hipError_t launch_cp_conv_min(const float *img, int32_t H, int32_t W, const int32_t *uv, int32_t n, int32_t half, const float *k,
                              int32_t kh, int32_t kw, float *tmp, float *imin, hipStream_t stream)
{
    if (n <= 0) return hipSuccess;
    if (kh < 1 || kw < 1 || kh > 3 || kw > 3) return hipErrorInvalidValue;
    ConvK kk{};
    kk.kh = kh; kk.kw = kw;
    for (int i = 0; i < kh * kw; i++) kk.k[i] = k[i];
    hipLaunchKernelGGL(cp_conv_min, dim3(n), dim3(256), 0, stream, img, H, W, uv, half, kk, tmp, imin);
    return hipGetLastError();
}

hipError_t launch_cp_shift_copy(const float *tmp, const float *mn, int32_t n, int32_t half, float *atlas, hipStream_t stream)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(cp_shift_copy, dim3(n), dim3(256), 0, stream, tmp, mn, half, atlas);
    return hipGetLastError();
}

hipError_t launch_negate_uv(float *out, int32_t n, hipStream_t stream)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(negate_uv, dim3((n + 255) / 256), dim3(256), 0, stream, out, n);
    return hipGetLastError();
}

}  // namespace mimc3
