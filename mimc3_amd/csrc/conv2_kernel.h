// conv2_kernel.h -- launch interface of the image pre-filter (internal).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mimc3 {

constexpr int kConvMaxTaps = 81;      // up to 9x9

struct Conv2Args {
    const float *in;       // [H][W]
    float *out;            // [H][W] in/out: the border rows/columns are read, never written by the stencil
    int32_t H, W, kh, kw;
    float k[kConvMaxTaps]; // row-major kernel
    uint32_t *minkey;      // [1] ordered-int image of the running minimum
};
// stencil + minimum, then the shift; enqueued on `stream`, no sync
hipError_t launch_conv2(const Conv2Args &a, hipStream_t stream);

}  // namespace mimc3
