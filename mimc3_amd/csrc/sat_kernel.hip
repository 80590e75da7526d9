// sat_kernel.hip -- packed summed-area tables of the zero-bordered integer planes (gfx950).
//
// What it is for.  The NCC of one cell (MIMC_module.c:719-734) needs six sums over the chip-sized box of the search
// window under the chip: n, sx, sxx (chip side), sy, syy (window side) and sxy.  For a box without null pixels n, sx, sxx
// are constants of the grid point, and sy = sum b, syy = sum b^2 depend ONLY on the window image and the box position --
// not on the chip -- so they are box sums of the image.  A grid point evaluates ~92 overlapping boxes, neighbouring grid
// points share most of their windows, and the CLI runs 8 passes (4 chip sizes x 2 directions) on every image pair: instead
// of recomputing sum b and sum b^2 inside every cell evaluation (two of the three dot-product streams of the matcher's
// inner loop), they are read from a summed-area table of the plane that is built once per image pair, next to the plane:
//       S[y][x] = sum over y' < y, x' < x of f(P[y'][x'])            (mod 2^64; (Hp+1) x (Wp+1) entries)
//       box(x, y, w, h) = S[y+h][x+w] - S[y][x+w] - S[y+h][x] + S[y][x]
// with the three quantities PACKED into one 64-bit word, f(b) = b + b^2 * 2^kSqShift + [b == 0] * 2^kNullShift.
// A box of at most 81 x 81 8-bit pixels has sum b < 2^21, sum b^2 < 2^29, nulls < 2^13, so the packed box sum is < 2^63:
// the modular inclusion-exclusion above returns the three exact integers in their fields, whatever wrapped on the way.
// The null count of a box (null <=> DN == 0 for integral DN, :622/:723) gives the validity counts of a6 (:605-644) and
// tells whether a window or a chip holds any null at all before a single pixel is looked at.
// u16 planes (q < 4096): sum q < 2^25 and sum q^2 < 2^37 fill 62 bits, the null count lives in a second table (u32).
//
// Exactness: everything is integer arithmetic mod 2^64 on exact integers -- no rounding anywhere.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "sat_kernel.h"

namespace mimc3 {

namespace {

template <class PX> struct SatF;
template <> struct SatF<unsigned char> {
    __device__ static __forceinline__ unsigned long long f(unsigned v, float)
    {
        return (unsigned long long)v + ((unsigned long long)(v * v) << kSatSqShift8) + ((unsigned long long)(v == 0u) << kSatNullShift8);
    }
};
template <> struct SatF<unsigned short> {
    __device__ static __forceinline__ unsigned long long f(unsigned v, float)
    {
        return (unsigned long long)v + ((unsigned long long)(v * v) << kSatSqShift16);
    }
};

template <> struct SatF<float> {
    // `mul` = 2^s makes the pixel an integer (s = 0: DN; s = 3: the Laplacian's multiples of 1/8): the table holds the sums of the
    // scaled values, fl((v 2^s)^2) = fl(v^2) 2^2s exactly (a power of two commutes with the rounding), and the kernel scales back
    __device__ static __forceinline__ Sat2 f(float v, float mul)
    {
        const float w = v * mul;
        const float sq = w * w;                                      // the reference's f32 product (rounds above 2^24), scaled
        return Sat2{(unsigned long long)w + ((unsigned long long)(w == 0.0f) << kSatNullShiftF), (unsigned long long)sq};
    }
};
template <class PX> struct SatE { typedef unsigned long long type; };
template <> struct SatE<float> { typedef Sat2 type; };

// inclusive scan of a 64-bit value over the 64 lanes of a wave
__device__ __forceinline__ unsigned long long wave_scan(unsigned long long v, int lane)
{
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned long long t = __shfl_up(v, o, 64);
        if (lane >= o) v += t;
    }
    return v;
}
__device__ __forceinline__ Sat2 wave_scan(Sat2 v, int lane) { return Sat2{wave_scan(v.a, lane), wave_scan(v.b, lane)}; }

// ---- pass A: row prefix.  One workgroup per plane row y: S[y+1][x+1] = sum_{x' <= x} f(P[y][x']), S[y+1][0] = 0.
//      (the column pass then adds the rows up in place)
template <class PX>
__global__ __launch_bounds__(256) void sat_rows(const PX *__restrict__ plane, int pitch, int Wp, typename SatE<PX>::type *__restrict__ S, int Ws,
                                                unsigned int *__restrict__ Z, int have_z, float mul)
{
    typedef typename SatE<PX>::type E;
    __shared__ E wsum[4];
    __shared__ unsigned long long zsum[4];
    const int y = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const PX *row = plane + (size_t)y * pitch;
    E *out = S + (size_t)(y + 1) * Ws;
    unsigned int *zout = have_z ? Z + (size_t)(y + 1) * Ws : nullptr;
    E carry{};
    unsigned long long zcarry = 0;
    if (tid == 0) { out[0] = E{}; if (have_z) zout[0] = 0; }
    for (int x0 = 0; x0 < Wp; x0 += 256) {
        const int x = x0 + tid;
        const PX v = x < Wp ? row[x] : (PX)1;                       // (positions past the row end are never stored)
        E s = wave_scan(x < Wp ? SatF<PX>::f(v, mul) : E{}, lane);
        unsigned long long z = have_z ? wave_scan((x < Wp && v == (PX)0) ? 1ull : 0ull, lane) : 0ull;
        if (lane == 63) { wsum[wave] = s; zsum[wave] = z; }
        __syncthreads();
        E base = carry, tot{};
        unsigned long long zbase = zcarry, ztot = 0;
#pragma unroll
        for (int w = 0; w < 4; w++) {
            if (w < wave) { base += wsum[w]; zbase += zsum[w]; }
            tot += wsum[w]; ztot += zsum[w];
        }
        if (x < Wp) { out[x + 1] = base + s; if (have_z) zout[x + 1] = (unsigned int)(zbase + z); }
        carry += tot; zcarry += ztot;
        __syncthreads();
    }
}

// ---- pass B: per column x and row segment: sum of the segment's row-prefix values
template <class T>
__global__ __launch_bounds__(256) void sat_col_partial(const T *__restrict__ S, int Ws, int ncol, int rows, int seg, T *__restrict__ part)
{
    const int x = blockIdx.x * 256 + threadIdx.x, sg = blockIdx.y;
    if (x >= ncol) return;
    const int r0 = 1 + sg * seg, r1 = min(rows, r0 + seg);          // rows 1 .. rows-1 carry data (row 0 is the zero row)
    T acc{};
    for (int r = r0; r < r1; r++) acc += S[(size_t)r * Ws + x];
    part[(size_t)sg * Ws + x] = acc;
}
// ---- pass C: column prefix inside each segment, started from the sum of the segments above
template <class T>
__global__ __launch_bounds__(256) void sat_col_apply(T *__restrict__ S, int Ws, int ncol, int rows, int seg, const T *__restrict__ part)
{
    const int x = blockIdx.x * 256 + threadIdx.x, sg = blockIdx.y;
    if (x >= ncol) return;
    T run{};
    for (int s = 0; s < sg; s++) run += part[(size_t)s * Ws + x];
    const int r0 = 1 + sg * seg, r1 = min(rows, r0 + seg);
    for (int r = r0; r < r1; r += 4) {                              // four independent loads in flight
        T v[4];
#pragma unroll
        for (int k = 0; k < 4; k++) v[k] = (r + k < r1) ? S[(size_t)(r + k) * Ws + x] : T{};
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (r + k < r1) { run += v[k]; S[(size_t)(r + k) * Ws + x] = run; }
    }
}

constexpr int kSatSeg = 64;

// The table of the region [x0, x0 + w) x [y0, y0 + h) of the plane (pitch `Wp_full` pixels; the table keeps the full plane's
// geometry): entries S[y][x] for y0 <= y <= y0 + h, x0 <= x <= x0 + w are prefix sums from (x0, y0) -- a box query is a
// difference of four of them, so any origin serves as long as the box lies inside the region.
template <class PX>
hipError_t build(const PX *plane_full, int Wp_full, SatRegion rg, typename SatE<PX>::type *S_full, unsigned int *Z_full, void *scratch, hipStream_t s, float mul = 1.0f)
{
    typedef typename SatE<PX>::type E;
    const int Ws = sat_pitch(Wp_full), Hp = rg.h, Wp = rg.w;
    const PX *plane = plane_full + (size_t)rg.y0 * Wp_full + rg.x0;
    E *S = S_full + (size_t)rg.y0 * Ws + rg.x0;
    unsigned int *Z = Z_full ? Z_full + (size_t)rg.y0 * Ws + rg.x0 : nullptr;
    const int rows = Hp + 1, nseg = (Hp + kSatSeg - 1) / kSatSeg;
    hipError_t e = hipMemsetAsync(S, 0, sizeof(E) * (size_t)(Wp + 1), s);                  // row 0 of the region
    if (e != hipSuccess) return e;
    if (Z && (e = hipMemsetAsync(Z, 0, sizeof(unsigned int) * (size_t)(Wp + 1), s)) != hipSuccess) return e;
    hipLaunchKernelGGL(sat_rows<PX>, dim3(Hp), dim3(256), 0, s, plane, Wp_full, Wp, S, Ws, Z, Z ? 1 : 0, mul);
    E *part = static_cast<E *>(scratch);
    const dim3 grid((Wp + 1 + 255) / 256, nseg);
    hipLaunchKernelGGL(sat_col_partial<E>, grid, dim3(256), 0, s, S, Ws, Wp + 1, rows, kSatSeg, part);
    hipLaunchKernelGGL(sat_col_apply<E>, grid, dim3(256), 0, s, S, Ws, Wp + 1, rows, kSatSeg, part);
    if (Z) {
        unsigned int *zp = reinterpret_cast<unsigned int *>(part + (size_t)nseg * Ws);
        hipLaunchKernelGGL(sat_col_partial<unsigned int>, grid, dim3(256), 0, s, Z, Ws, Wp + 1, rows, kSatSeg, zp);
        hipLaunchKernelGGL(sat_col_apply<unsigned int>, grid, dim3(256), 0, s, Z, Ws, Wp + 1, rows, kSatSeg, zp);
    }
    return hipGetLastError();
}

}  // namespace

size_t sat_bytes(int Hp, int Wp) { return sizeof(unsigned long long) * (size_t)(Hp + 1) * sat_pitch(Wp); }
size_t sat_null_bytes(int Hp, int Wp) { return sizeof(unsigned int) * (size_t)(Hp + 1) * sat_pitch(Wp); }
size_t sat_scratch_bytes(int Hp, int Wp)
{
    const size_t nseg = (size_t)(Hp + kSatSeg - 1) / kSatSeg;
    return (sizeof(unsigned long long) + sizeof(unsigned int)) * nseg * sat_pitch(Wp);
}

size_t sat2_bytes(int Hp, int Wp) { return sizeof(Sat2) * (size_t)(Hp + 1) * sat_pitch(Wp); }
size_t sat2_scratch_bytes(int Hp, int Wp) { return sizeof(Sat2) * ((size_t)(Hp + kSatSeg - 1) / kSatSeg) * sat_pitch(Wp); }
hipError_t launch_sat_f32i(const float *plane, int Wp, SatRegion rg, int shift, Sat2 *S, void *scratch, hipStream_t s)
{
    return build<float>(plane, Wp, rg, S, nullptr, scratch, s, (float)(1 << shift));
}

// bit 0: some pixel is not an integer in [0, 2^20); bit 1: some pixel x 8 is not
__global__ void detect_int16(const float *img, size_t n, int *flag)
{
    int f = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float v = img[i], v8 = v * 8.0f;
        if (!(v >= 0.0f && v <= 1048575.0f && truncf(v) == v)) f |= 1;     // NaN fails every comparison
        if (!(v8 >= 0.0f && v8 <= 1048575.0f && truncf(v8) == v8)) f |= 2;
    }
    if (f) atomicOr(flag, f);
}
hipError_t launch_detect_int16(const float *img, size_t n, int *d_flag, hipStream_t s)
{
    hipLaunchKernelGGL(detect_int16, dim3(2048), dim3(256), 0, s, img, n, d_flag);
    return hipGetLastError();
}

hipError_t launch_sat_u8(const unsigned char *plane, int Wp, SatRegion rg, unsigned long long *S, void *scratch, hipStream_t s)
{
    return build<unsigned char>(plane, Wp, rg, S, nullptr, scratch, s);
}
hipError_t launch_sat_u16(const unsigned short *plane, int Wp, SatRegion rg, unsigned long long *S, unsigned int *Z, void *scratch, hipStream_t s)
{
    return build<unsigned short>(plane, Wp, rg, S, Z, scratch, s);
}

}  // namespace mimc3
