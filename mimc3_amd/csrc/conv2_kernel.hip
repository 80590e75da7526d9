// conv2_kernel.hip -- the image pre-filter of the CLI's 24 filtered matcher passes, for gfx950 (MI355X).
//
// Replaces GMA_float_conv2 (MIMC_module.c:2517-2585): correlation with a small kernel over the interior where a
// null DN ((int)(value+0.5) == 0) poisons its whole stencil (:2545), the minimum over the WHOLE output plane --
// border included, which the stencil never writes (:2555-2565) -- and the shift `out -= min-1`, poisoned -> 0,
// over rows [oy, H-oy) x columns [ox, W) (:2568-2582: the right-hand border columns are shifted too).
//
// HBM-bound byte work: one thread per pixel, rows contiguous across lanes; the k_h x k_w taps of neighbouring
// lanes overlap in L1/L2.  Two launches because the shift needs the global minimum: pass 1 writes the raw
// stencil and folds a per-wave minimum into one ordered-int atomicMin, pass 2 shifts in place.  Products and
// sums are f32 in the reference's tap order (row-major over the kernel), -ffp-contract=off.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "conv2_kernel.h"

namespace mimc3 {
namespace {

__device__ __forceinline__ uint32_t key_of(float v)
{
    const uint32_t u = __float_as_uint(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float value_of(uint32_t k)
{
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k);
}

__global__ __launch_bounds__(256) void conv2_init(uint32_t *minkey) { *minkey = key_of(1e+37f); }   // :2554

// (int32_t)(p + 0.5) == 0 (:2545: f32 + f64 0.5, truncation) <=> -1.5 < p < 0.5; NaN and huge values are "not null"
__device__ __forceinline__ bool null_dn(float p) { return p > -1.5f && p < 0.5f; }

__global__ __launch_bounds__(256) void conv2_apply(Conv2Args a)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    const int ox = a.kw / 2, oy = a.kh / 2;
    uint32_t key = 0xFFFFFFFFu;
    // a block walks down its 256-column strip: one atomic per block at the end (same-address atomics serialise in L2)
    for (int r = blockIdx.y; r < a.H; r += gridDim.y) {
        float v = __builtin_nanf("");
        if (c < a.W) {
            const size_t idx = (size_t)r * a.W + c;
            if (r >= oy && r < a.H - oy && c >= ox && c < a.W - ox) {
                float s = 0.0f;
                for (int i = 0; i < a.kh; i++)
                    for (int j = 0; j < a.kw; j++) {
                        const float p = a.in[(size_t)(r + i - oy) * a.W + c + j - ox];
                        const float dn = null_dn(p) ? __builtin_nanf("") : p;
                        s += dn * a.k[i * a.kw + j];
                    }
                a.out[idx] = s;
                v = s;
            } else v = a.out[idx];                                            // border: whatever the buffer holds
        }
        // minimum of the non-NaN values (`out < dn_min` is false for NaN, :2559)
        if (v == v) key = min(key, key_of(v));
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) key = min(key, (uint32_t)__shfl_xor((int)key, off));
    __shared__ uint32_t part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = key;
    __syncthreads();
    if (threadIdx.x == 0) {
        key = min(min(part[0], part[1]), min(part[2], part[3]));
        if (key != 0xFFFFFFFFu) atomicMin(a.minkey, key);
    }
}

__global__ __launch_bounds__(256) void conv2_shift(Conv2Args a)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    const int r = blockIdx.y;
    const int ox = a.kw / 2, oy = a.kh / 2;
    if (c >= a.W || c < ox || r < oy || r >= a.H - oy) return;
    const float mn = value_of(*a.minkey);
    float *o = a.out + (size_t)r * a.W + c;
    const float v = *o;
    *o = (v != v) ? 0.0f : v - (mn - 1.0f);                                   // :2572-2580
}

}  // namespace

hipError_t launch_conv2(const Conv2Args &a, hipStream_t stream)
{
    const dim3 grid((a.W + 255) / 256, a.H), block(256);
    const dim3 grid_apply((a.W + 255) / 256, a.H < 256 ? a.H : 256);
    hipLaunchKernelGGL(conv2_init, dim3(1), dim3(1), 0, stream, a.minkey);
    hipLaunchKernelGGL(conv2_apply, grid_apply, block, 0, stream, a);
    hipLaunchKernelGGL(conv2_shift, grid, block, 0, stream, a);
    return hipGetLastError();
}

}  // namespace mimc3
