// pivot_kernel.hip -- the DLC pivot lists of get_uv_pivot (MIMC_module.c:543-602) expanded ON THE DEVICE from per-point
// corridors (gfx950).
//
// get_uv_pivot has two halves.  (1) The corridor of a point -- theta = atan2(vy, vx), the step (cos, sin) normalised so that
// its larger component is +-1, the corridor length (:559-573) -- needs libm's atan2 / cos / sin / sqrt and stays on the host
// (host_geometry.cpp, bit-equal to the reference's results).  (2) The pivot list itself is pure IEEE arithmetic on those
// numbers: a running f32 sum of the step, an f64 product for the length test, C truncation toward zero (:576-598).  That
// half runs here, with -ffp-contract=off, one thread per grid point: 24 bytes per point cross PCIe instead of the
// 8 x npiv bytes of the list (25.6 MB of the 37.6 MB a C2 pass moved), and the lists never exist on the host.
//
//   pivot_count   n(g) pivots that stay inside the image +- ocw and inside the corridor length (:576-585) + the extents a
//                 matcher launch is sized by (max n, max |last pivot| per axis) + "some point has no pivot" (:589-591)
//   pivot_scan    CSR offsets piv_off[0..N] (one workgroup; N <= a few million)
//   pivot_fill    pivot k = ((int)(u_k + 0.5), -(int)(v_k + 0.5)) with the same running sums (:592-598); optionally the
//                 negated copy main() makes in place for its swapped pass (MIMC_main.c:272-279)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "pivot_kernel.h"

namespace mimc3 {

__global__ __launch_bounds__(256) void pivot_count(const double *__restrict__ xyuvav, int xy_stride, int xy_col, const CorridorDev *__restrict__ cor, int N,
                                                   int ocw, int H, int W, int32_t *__restrict__ cnt, int32_t *__restrict__ ext)
{
    const int g = blockIdx.x * 256 + threadIdx.x;
    int n = 0, au = 0, av = 0;
    if (g < N) {
        const CorridorDev c = cor[g];
        const double *row = xyuvav + (size_t)xy_stride * (size_t)g + xy_col;
        const float fu = (float)row[0], fv = (float)row[1], fo = (float)ocw;
        const float wmax = (float)(W - 1), hmax = (float)(H - 1);
        float u = 0.0f, v = 0.0f, pu = 0.0f, pv = 0.0f;              // (pu, pv): the sums that make the LAST pivot
        for (;;) {
            const bool inside = (u + fu - fo > 0.0f) && (u + fu + fo < wmax) && (v + fv - fo > 0.0f) && (v + fv + fo < hmax);
            if (!inside) break;
            if (!(c.length > (double)c.norm_incr * (double)n)) break;
            ++n; pu = u; pv = v; u += c.incr_u; v += c.incr_v;
        }
        cnt[g] = n;
        if (n > 1) {                                                  // pivot 0 is (0, 0) by construction (:592-593)
            const int lu = (int32_t)((double)pu + 0.5), lv = -(int32_t)((double)pv + 0.5);
            au = lu < 0 ? -lu : lu; av = lv < 0 ? -lv : lv;
        }
    }
    int mn = n, zero = (g < N && n == 0) ? 1 : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        mn = max(mn, __shfl_xor(mn, o, 64)); au = max(au, __shfl_xor(au, o, 64)); av = max(av, __shfl_xor(av, o, 64));
        zero |= __shfl_xor(zero, o, 64);
    }
    // block maxima through LDS, then at most one atomic per block and word -- and none where the word already holds the value
    // (every wave hitting four global words directly is thousands of same-address atomics, which serialise)
    __shared__ int smx[4][4];
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { smx[0][wv] = mn; smx[1][wv] = au; smx[2][wv] = av; smx[3][wv] = zero; }
    __syncthreads();
    if (threadIdx.x < 4) {
        const int k = threadIdx.x;
        int v = smx[k][0];
        for (int w = 1; w < 4; w++) v = k == 3 ? (v | smx[k][w]) : max(v, smx[k][w]);
        if (v > __hip_atomic_load(&ext[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
            if (k == 3) atomicOr(&ext[3], 1); else atomicMax(&ext[k], v);
        }
    }
}

// piv_off[0] = 0, piv_off[g + 1] = cnt[0] + ... + cnt[g]; total also lands in total_out (8 bytes behind ext for one readback)
__global__ __launch_bounds__(1024) void pivot_scan(const int32_t *__restrict__ cnt, int N, int64_t *__restrict__ piv_off, int64_t *__restrict__ total_out)
{
    __shared__ long long part[1024];
    const int t = threadIdx.x;
    const int per = (N + 1023) / 1024, lo = t * per, hi = min(N, lo + per);
    long long s = 0;
    for (int i = lo; i < hi; i++) s += cnt[i];
    part[t] = s;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {                              // inclusive scan over the 1,024 partial sums
        const long long v = t >= o ? part[t - o] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    long long run = t ? part[t - 1] : 0;
    if (t == 0) piv_off[0] = 0;
    for (int i = lo; i < hi; i++) { run += cnt[i]; piv_off[i + 1] = run; }
    if (t == 1023) *total_out = part[1023];
}

__global__ __launch_bounds__(256) void pivot_fill(const CorridorDev *__restrict__ cor, const int64_t *__restrict__ piv_off, int N,
                                                  int32_t *__restrict__ uv, int32_t *__restrict__ uvn)
{
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (g >= N) return;
    const int64_t b = piv_off[g];
    const int n = (int)(piv_off[g + 1] - b);
    const float iu = cor[g].incr_u, iv = cor[g].incr_v;
    int2 *o = uv ? reinterpret_cast<int2 *>(uv) + b : nullptr;
    int2 *on = uvn ? reinterpret_cast<int2 *>(uvn) + b : nullptr;
    float u = 0.0f, v = 0.0f;
    for (int k = 0; k < n; k++) {
        int2 q;
        if (k == 0) q = make_int2(0, 0);
        else {
            u += iu; v += iv;
            q.x = (int32_t)((double)u + 0.5);                         // C truncation toward zero (:596)
            q.y = -(int32_t)((double)v + 0.5);                        // image v is down, a-priori vy is north (:597)
        }
        if (o) o[k] = q;
        if (on) on[k] = make_int2(-q.x, -q.y);
    }
}

hipError_t launch_pivot_count(const double *d_xyuvav, int xy_stride, int xy_col, const CorridorDev *d_cor, int N, int ocw, int H, int W, int32_t *d_cnt,
                              int64_t *d_piv_off, int32_t *d_ext6, hipStream_t s)
{
    hipError_t e = hipMemsetAsync(d_ext6, 0, 6 * sizeof(int32_t), s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(pivot_count, dim3((N + 255) / 256), dim3(256), 0, s, d_xyuvav, xy_stride, xy_col, d_cor, N, ocw, H, W, d_cnt, d_ext6);
    hipLaunchKernelGGL(pivot_scan, dim3(1), dim3(1024), 0, s, d_cnt, N, d_piv_off, reinterpret_cast<int64_t *>(d_ext6 + 4));
    return hipGetLastError();
}

hipError_t launch_pivot_fill(const CorridorDev *d_cor, const int64_t *d_piv_off, int N, int32_t *d_uv, int32_t *d_uvn, hipStream_t s)
{
    hipLaunchKernelGGL(pivot_fill, dim3((N + 255) / 256), dim3(256), 0, s, d_cor, d_piv_off, N, d_uv, d_uvn);
    return hipGetLastError();
}

}  // namespace mimc3
