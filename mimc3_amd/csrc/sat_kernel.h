// sat_kernel.h -- packed summed-area tables of the zero-bordered integer planes (internal; see sat_kernel.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mimc3 {

// u8 planes: f(b) = b | b^2 << 21 | [b == 0] << 50   (a box of <= 81^2 pixels: sum b < 2^21, sum b^2 < 2^29, nulls < 2^13)
constexpr int kSatSqShift8 = 21, kSatNullShift8 = 50;
// u16 planes (q < 4096): f(q) = q | q^2 << 25        (sum q < 2^25, sum q^2 < 2^37); nulls in a second u32 table
constexpr int kSatSqShift16 = 25;

// f32 planes of INTEGRAL values < 2^20, possibly after a scale by 8 (16-bit imagery, its integer gradients, its Laplacian in 1/8 units): the reference's sums are sum (double)b and sum (double)(float)(b * b)
// (MIMC_module.c:726-730: the f32 product ROUNDS above 2^24, T1) -- both exact integers in any order.  One 16-byte entry:
//   a = sum b | nulls << 40   (a box of <= 81^2 pixels: sum b < 2^33, nulls < 2^13),   b = sum fl(b * b)  (< 2^53: the reference's own f64 sums are exact up to there)
constexpr int kSatNullShiftF = 40;
struct Sat2 {
    unsigned long long a, b;
    __host__ __device__ Sat2 operator+(const Sat2 &o) const { return Sat2{a + o.a, b + o.b}; }
    __host__ __device__ Sat2 operator-(const Sat2 &o) const { return Sat2{a - o.a, b - o.b}; }
    __host__ __device__ Sat2 &operator+=(const Sat2 &o) { a += o.a; b += o.b; return *this; }
};

// entries per table row: the plane's pitch + the zero column
static inline int sat_pitch(int Wp) { return Wp + 1; }
size_t sat_bytes(int Hp, int Wp);          // (Hp + 1) x sat_pitch(Wp) x 8
size_t sat_null_bytes(int Hp, int Wp);     // u16 planes only: (Hp + 1) x sat_pitch(Wp) x 4
size_t sat_scratch_bytes(int Hp, int Wp);  // column-pass partial sums
// The part of a plane a table covers: x0 <= x < x0 + w, y0 <= y < y0 + h (plane pixels).  Box queries are valid for boxes inside it
// (the prefix sums start at the region's origin: a box sum is a difference of four entries, any common origin serves).  The
// whole zero-bordered plane for ordinary pairs; the image area alone for the control-point stage's chip atlases, whose search
// areas never leave a tile (their planes are mostly border).
struct SatRegion { int x0, y0, w, h; };
// Enqueue the table build on `s`.  `Wp` = the plane's pitch in pixels; tables keep the full plane's geometry (sat_pitch(Wp)).
hipError_t launch_sat_u8(const unsigned char *plane, int Wp, SatRegion rg, unsigned long long *S, void *scratch, hipStream_t s);
hipError_t launch_sat_u16(const unsigned short *plane, int Wp, SatRegion rg, unsigned long long *S, unsigned int *Z, void *scratch, hipStream_t s);
size_t sat2_bytes(int Hp, int Wp);         // f32 planes: (Hp + 1) x sat_pitch(Wp) x 16; scratch: sat2_scratch_bytes
size_t sat2_scratch_bytes(int Hp, int Wp);
hipError_t launch_sat_f32i(const float *plane, int Wp, SatRegion rg, int shift /* pixel x 2^shift is the integer */, Sat2 *S, void *scratch, hipStream_t s);
// *d_flag |= 1 if some pixel is not an integer in [0, 2^20), |= 2 if some pixel x 8 is not
hipError_t launch_detect_int16(const float *img, size_t n, int *d_flag, hipStream_t s);

// box sum of the w x h pixels whose top-left plane pixel is (x, y): four loads, modular inclusion-exclusion
template <class T>
__device__ __forceinline__ T sat_box(const T *__restrict__ S, int Ws, int x, int y, int w, int h)
{
    const T *r0 = S + (size_t)y * Ws + x, *r1 = r0 + (size_t)h * Ws;
    return r1[w] - r0[w] - r1[0] + r0[0];
}

// Null count of a w x h box of a u8 plane from its packed table, exact for ANY box size.  The packed fields of one query are exact up to
// 8,224 pixels (sum b < 2^21; beyond that the field above takes a carry and the null count comes back off by it -- a window of
// 16,384 - carry nulls would read as null-free).  Larger boxes are cut into sub-boxes of at most 64 x 64 pixels, one per lane, and the
// counts are summed over the wave.  Wave-uniform arguments; every lane gets the result.
__device__ __forceinline__ int sat_nulls_u8(const unsigned long long *__restrict__ S, int Ws, int x, int y, int w, int h, int lane)
{
    if (w * h <= 8224) return (int)(sat_box(S, Ws, x, y, w, h) >> kSatNullShift8);
    const int nx = (w + 63) >> 6, ny = (h + 63) >> 6;
    int cnt = 0;
    for (int t = lane; t < nx * ny; t += 64) {
        const int j = t / nx, i = t - j * nx;
        const int w0 = w - 64 * i < 64 ? w - 64 * i : 64, h0 = h - 64 * j < 64 ? h - 64 * j : 64;
        cnt += (int)(sat_box(S, Ws, x + 64 * i, y + 64 * j, w0, h0) >> kSatNullShift8);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
    return cnt;
}

}  // namespace mimc3
