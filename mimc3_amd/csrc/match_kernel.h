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
    int32_t xy_stride, xy_col;   // the points' (u, v) sit at xyuvav[xy_stride * g + xy_col + {0, 1}]: 6, 2 (xyuvav rows) or 2, 0 (packed [N][2])
    int32_t N;
    int32_t off_u, off_v;        // CP offset, added to the search centre only (MIMC_module.c:827-828)
    const int32_t *piv_uv;       // device CSR payload [P][2]
    const int64_t *piv_off;      // device CSR offsets [N+1]
    int32_t ocw;
    int32_t swap;                // 0: chip from i0, window from i1; 1: exchanged
    int32_t win_half;            // 0: DLC window (|last pivot|+ocw+2, last row/column empty, :863-886); >0: the search area is
                                 // the full (2*win_half+1)^2 square (get_offset_image's image chips, :347)
    float thr;                   // smallest f32 whose f64 value is >= MIN_DN (1e-10, MIMC_module.c:21)
    float *out;                  // device [N][3]
    const int32_t *point_list;   // optional: process only these point indices (device list) ...
    const int32_t *point_count;  // ... whose length is read on the device (nullptr = all N points)
    // LDS carve (floats / pivots), filled by the launcher from the per-launch maxima
    int32_t lds_chip_f, lds_win_f, lds_cell_f, lds_npiv;
    // global workspace for the compact cell grid when it outgrows LDS (long diagonal corridors): kCellGlobalGrid slices
    unsigned char *cell_ws;
    size_t cell_ws_bytes, cell_ws_stride;
    int32_t cell_ws_cells;
};
constexpr int kCellGlobalGrid = 1024;   // persistent workgroups of the workspace mode (4 per CU)
size_t match_f32_workspace_bytes(int ocw, int max_abs_u, int max_abs_v, int max_npiv, int win_half);

// ---- exact-integer path for 8-bit imagery (match_u8_kernel.hip) -------------------------------
constexpr uint8_t kMxNulls = 2, kMxRest = 1, kMxWn = 3;
constexpr int kU8Pad = 256;      // zero border (pixels) around the u8 planes; multiple of 4

struct MatchU8Args {                // arguments of the register-tiled kernel family (match_px_kernel.hip)
    const unsigned char *p0, *p1;   // zero-bordered planes of i0, i1 (u8 or f32 pixels): pixel (u,v) at [(v+pad)*Wp + u+pad]
    int32_t Wp, pad;                // plane pitch (PIXELS; a whole number of dwords) and border
    float thr;                      // smallest f32 whose f64 value is >= MIN_DN (f32 policy)
    double scale0, scale1;          // u16 policy: plane k stores value * 2^s_k; scale_k = 2^-s_k (1.0 otherwise)
    int32_t H, W;
    const double *xyuvav;
    int32_t xy_stride, xy_col;      // see MatchArgs
    int32_t N;
    int32_t off_u, off_v;
    const int32_t *piv_uv;
    const int64_t *piv_off;
    int32_t ocw, swap;
    int32_t win_half;               // 0: DLC window (|last pivot|+ocw+2, last row/column empty); > 0: full (2*win_half+1)^2 search area (CP stage)
    float *out;
    int32_t *ovf_list, *ovf_count;  // points whose NCC cache overflowed: handed to the general kernel (list mode)
    const int32_t *point_list, *point_count;   // list mode: workgroup b handles point_list[b], b < *point_count (nullptr = all N points)
    int32_t *fail_list, *fail_count;           // PxU8o only: points whose chip or window does not fit a local 8-bit range
    // matrix-core kernel (match_mx_kernel.hip): one byte per grid point, zero before the first launch -- kMxNulls = handed from its clean
    // form to its window-null form, kMxRest = neither form takes the point (plain stores: a shared list counter serialises ~100,000
    // same-address atomics per launch, measured 1.0 ms)
    uint8_t *mx_flags;
    int32_t mx_preflag;             // the launch may hold corridors wider than the tile: a pre-pass has flagged those points (kMxRest), the kernel leaves them at once
    int32_t mx_gen_on, mx_wn_on;    // which of its forms for null-ridden points run behind the clean form (general / window nulls only); the others' points get kMxRest
    // flag mode of every kernel of the family: workgroup b handles point b only if point_flags[b] == flag_value
    const uint8_t *point_flags;
    int32_t flag_value;
    // LDS carve, filled by the launcher
    int32_t lds_pw, lds_off_val, lds_off_vis, lds_off_list, lds_off_sums, lds_off_piv, lds_list_cap;
    int32_t lds_off_chip, lds_off_lw, lds_off_lc;   // big-chip integer configs: LDS chip copy, window-null and chip-null lists
    int32_t lds_off_traj;                           // many-pivot configs: recorded climbs of the pivots beyond the first 64
    int32_t lds_off_vals, lds_nslot;                // compact configs: value slots of the two-level NCC cache
    // PxU8o: min | max << 16 of the non-null pixels of every 16x16-pixel tile of the two u16 planes (plane coordinates), or null
    const uint32_t *rt0, *rt1;
    int32_t rt_tw;                                  // tiles per plane row
    // packed summed-area tables of the two planes (sat_kernel.hip; policies with P::SAT), (Hp + 1) rows of sat_ws entries
    const void *sat0, *sat1;
    const void *satz0, *satz1;      // u16 planes: null counts (u32), same geometry
    int32_t sat_ws;
    int32_t lookahead;              // speculative climb: 3x3 blocks requested ahead along a straight move
    unsigned long long *stats;     // diagnostics only (env MIMC3_U8_STATS): per-phase s_memtime sums
    int32_t debug_stop;             // diagnostics only (env MIMC3_U8_DEBUG_STOP): leave the kernel after phase k; 0 = off
    int32_t dry_run;                // launcher only: compute the LDS carve and return hipSuccess / hipErrorInvalidValue (does not fit
                                    // 160 KB) without launching -- the C ABI asks this before it commits to a kernel policy
};

// f32 image -> zero-bordered u8 plane (plane must be pre-zeroed); *d_flag is set to 1 if any pixel
// is not an integer in [0,255] (then the u8 path must not be used for this image).
hipError_t launch_prep_u8(const float *img, int H, int W, unsigned char *plane, int Wp, int pad, int *d_flag, hipStream_t s);
// raw DN as the TIFF holds it -> f32 image (+ the u8 plane for 8-bit DN; plane pre-zeroed): the widening of
// GMA_float_load_tiff (GMA.c:288-310) done on the device so that only the raw bytes cross PCIe
hipError_t launch_widen_u8(const unsigned char *raw, int H, int W, float *img, unsigned char *plane, int Wp, int pad, hipStream_t s);
hipError_t launch_widen_u16(const unsigned short *raw, size_t n, float *img, hipStream_t s);
// instantiated chip sizes and border reach (|last pivot| + |CP offset| must fit in the border)
bool match_u8_supported(int ocw, int max_reach_u, int max_reach_v);
// same kernel family on zero-bordered u16 planes of scaled integers (q = value * 2^shift < 4096): same chip sizes as u8
hipError_t launch_detect_scaled_int(const float *img, size_t n, int *d_flags, hipStream_t s);
hipError_t launch_prep_u16(const float *img, int H, int W, unsigned short *plane, int Wp, int pad, int shift, hipStream_t s);
hipError_t launch_match_u16(MatchU8Args a, int max_abs_u, int max_abs_v, int max_npiv, hipStream_t stream);
// u16 planes of INTEGERS read through a per-point offset into the u8 kernels (points that do not fit go to fail_list)
hipError_t launch_match_u8o(MatchU8Args a, int max_abs_u, int max_abs_v, int max_npiv, hipStream_t stream);
// fraction-of-tiles estimate for the above: out[0] += tiles whose non-null range fits 8 bits, out[1] += tiles with data
hipError_t launch_range_tiles(const unsigned short *plane, int H, int W, int Wp, int pad, int *d_out2, hipStream_t s);
// tiles [ceil(Hp/16)][ceil(Wp/16)] of a zero-bordered u16 plane (Hp rows): min | max << 16 over the non-null pixels (0xffff | 0 if none)
hipError_t launch_range_tiles16(const unsigned short *plane, int Hp, int Wp, uint32_t *tiles, hipStream_t s);
// same kernel family on zero-bordered f32 planes (any f32 imagery; small chips only)
hipError_t launch_prep_f32(const float *img, int H, int W, float *plane, int Wp, int pad, hipStream_t s);
bool match_f32x_supported(int ocw, int max_reach_u, int max_reach_v);
hipError_t launch_match_f32x(MatchU8Args a, int max_abs_u, int max_abs_v, int max_npiv, hipStream_t stream);
hipError_t launch_match_u8(MatchU8Args a, int max_abs_u, int max_abs_v, int max_npiv, hipStream_t stream);
// dense correlation surfaces on the matrix cores (8-bit planes with tables; match_mx_kernel.hip): takes the points whose cell grid
// fits its tile and whose chip has no null, leaves the others in a.mx_rest_list for launch_match_u8 in list mode
bool match_mx_supported(int ocw, int max_npiv, int win_half, int max_abs_u, int max_abs_v);
hipError_t launch_match_mx(MatchU8Args a, hipStream_t stream);

// max_abs_u/v: max over points of |last pivot| per axis; max_npiv: max pivots per point.
hipError_t launch_match_f32(MatchArgs a, int max_abs_u, int max_abs_v, int max_npiv, hipStream_t stream);

}  // namespace mimc3
