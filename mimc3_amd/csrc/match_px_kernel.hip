// match_px_kernel.hip -- DLC/NCC matcher for gfx950, register-tiled kernel family.
//
// Same contract as match_kernel.hip (matching_ncc_dlc_2, MIMC_module.c:805-842).  One kernel template,
// match_ncc_dlc_px<PxCfg<Policy, OCW, LPC, NW, ...>>, and five pixel policies:
//   PxU8    image pairs whose pixels are all integers in [0,255] (what GMA_float_load_tiff yields for an 8-bit TIFF,
//           GMA.c:288-310).  Every running sum of the reference's NCC loop (MIMC_module.c:719-733: n, sx, sy, sxx, syy, sxy;
//           f32 products, f64 accumulation) is an exact integer < 2^31: v_dot4_u32_u8 on packed bytes, f64 only for the
//           final formula (:734).
//   PxU16   scaled-integer pairs q = value * 2^s < 4096 (12-bit DN; the Laplacian of an 8-bit pair in 1/8 units):
//           v_dot2_u32_u16, 64-bit cross-lane sums, exact power-of-two rescaling in the finish.
//   PxU8o   integer u16 planes whose LOCAL range fits a byte (the gradients of an 8-bit pair), staged through a per-point
//           offset onto the PxU8 machinery; the finish rebuilds the true integer sums.
//   PxF32   any f32 imagery: f32 products, f64 accumulation in registers -- the reference's arithmetic up to summation order.
//   PxF32i  PxF32 on planes whose pixels (x 1 or x 8) are integers below 2^20 (16-bit DN and its filtered forms): the sums
//           are exact integers in any order, so they can come from tables (below).
//
// Layout / decomposition (one workgroup of NW wave64 = one grid point):
//   * the images live in HBM as zero-bordered planes (border >= kU8Pad px, pitch a whole number of dwords), built once per
//     image pair, each with a packed SUMMED-AREA TABLE (sat_kernel.hip): the sums of a null-free box of the window (sum b,
//     sum b^2), the chip's sums and every null count are four-load box queries;
//   * the DLC window is staged into LDS as aligned dwords (keeps the global pixel phase `sh`);
//   * the chip lives in REGISTERS: each of the 64/LPC lane groups of a wave holds the whole chip, lane l of a group owns rows
//     l, l+LPC, ... as dwords (+ a few single-dword "tail" tasks);
//   * one evaluation round computes NW*64/LPC NCC cells: a lane slides over the aligned window dwords of its row
//     (v_alignbyte_b32 when a dword holds several pixels), accumulates, and the LPC lanes of a cell are reduced with DPP row
//     operations (groups of 32 / 64 lanes: LDS atomics into the cell's parking slot);
//   * per-cell modes: XY (null-free box and chip: only sxy is accumulated, the rest is constants + table), WN / GC / GENERAL
//     (window nulls: three or six masked sums), CHIPNULL, sparse corrections over null lists on the big chips;
//   * NCC cache: one f32 word per compact cell (or, COMPACT form for large windows, a 16-bit entry + value slots);
//   * speculative parallel climb + exact replay of the reference's sequential hill climb (:691-753).
// DESIGN.md section 4.1 describes each step and what was measured for it.
#include <hip/hip_runtime.h>
#include <type_traits>
#include <atomic>
#include <mutex>
#include <stdint.h>
#include <stdlib.h>
#include <stdio.h>
#include "match_kernel.h"
#include "sat_kernel.h"

#ifndef MIMC3_XY_ACC
#define MIMC3_XY_ACC 2          // independent dot4 chains of the sxy-only body (tuning switch)
#endif
#ifndef MIMC3_SAT_DEFER
#define MIMC3_SAT_DEFER -1      // table look-ups of a batch: keep the four corners in registers, combine them in the finish.  -1 = per config
#endif                          // (PxCfg::SAT_DEFER), 0 / 1 = never / always (A/B builds)
#ifndef MIMC3_WN
#define MIMC3_WN 1              // window-null cells of a clean chip: three sums + table instead of six
#endif
#ifndef MIMC3_NULL_FLAGS
#define MIMC3_NULL_FLAGS 1      // sparse-correction configs: skip the window-null walk of dirty-list boxes that hold no null (table look-up)
#endif
#ifndef MIMC3_OPQ_SMALL
#define MIMC3_OPQ_SMALL 1       // keep the chip-derived masks of the small chips out of registers too (only chips with nulls derive any: GC mode)
#endif
#ifndef MIMC3_FAST_REPLAY
#define MIMC3_FAST_REPLAY 1     // exact replay on scan centres the lanes decoded beforehand (register visited set, <= 64 pivots)
#endif
#ifndef MIMC3_WIDE_REPLAY
#define MIMC3_WIDE_REPLAY 1     // big chips: register visited set for cell grids up to 96 x 128
#endif
#ifndef MIMC3_GC
#define MIMC3_GC 1              // six-sum body with compile-time chip masks for null-free chips
#endif
#ifndef MIMC3_STAGE_KB
#define MIMC3_STAGE_KB 8        // window rows a thread fetches before it uses any
#endif
#ifndef MIMC3_EVEN_PITCH
#define MIMC3_EVEN_PITCH 0      // LDS window pitch not forced to an odd number of dwords
#endif
#ifndef MIMC3_FAST_BATCH
#define MIMC3_FAST_BATCH 64     // table cells of a one-wave config finished per pass (their slots hold one word)
#endif

namespace mimc3 {

static constexpr float kUnknown = 3.0f;
static constexpr float kWanted = 4.0f;

__device__ __forceinline__ uint32_t dot4(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_udot4(a, b, c, false); }
__device__ __forceinline__ uint32_t alignb(uint32_t hi, uint32_t lo, uint32_t s) { return __builtin_amdgcn_alignbyte(hi, lo, s); }
// 0x80 in every byte that is non-zero
__device__ __forceinline__ uint32_t nz80(uint32_t v) { return (((v & 0x7f7f7f7fu) + 0x7f7f7f7fu) | v) & 0x80808080u; }
__device__ __forceinline__ uint32_t ff_from80(uint32_t t) { return t | (t - (t >> 7)); }

template <int LPC>
__device__ __forceinline__ uint32_t group_sum(uint32_t v)
{
    int x = (int)v;
    x += __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xF, 0xF, true);    // quad_perm [1,0,3,2]
    x += __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xF, 0xF, true);    // quad_perm [2,3,0,1]
    if (LPC >= 8) x += __builtin_amdgcn_update_dpp(0, x, 0x141, 0xF, 0xF, true);    // row_half_mirror -> sum of 8 lanes
    if (LPC >= 16) x += __builtin_amdgcn_update_dpp(0, x, 0x140, 0xF, 0xF, true);   // row_mirror -> sum of the 16-lane row
    if (LPC >= 32) x += __shfl_xor(x, 16, 64);
    if (LPC >= 64) x += __shfl_xor(x, 32, 64);
    return (uint32_t)x;
}
__device__ __forceinline__ int wave_sum_i(int v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// first-wins arg-max over the 16 lanes of a DPP row (lexicographic max on (value, -index)); VALU only
__device__ __forceinline__ void argmax_row16(float &v, int &i)
{
#define MIMC3_ARGMAX_STEP(ctrl)                                                                           \
    {                                                                                                     \
        const float ov = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, 0xF, 0xF, true)); \
        const int oi = __builtin_amdgcn_update_dpp(0, i, ctrl, 0xF, 0xF, true);                           \
        const bool t = (ov > v) || (ov == v && oi < i);                                                   \
        v = t ? ov : v; i = t ? oi : i;                                                                   \
    }
    MIMC3_ARGMAX_STEP(0xB1) MIMC3_ARGMAX_STEP(0x4E) MIMC3_ARGMAX_STEP(0x141) MIMC3_ARGMAX_STEP(0x140)
#undef MIMC3_ARGMAX_STEP
}

// Evaluation modes, chosen PER CELL (the cells of a round share one mode):
//   FAST     chip has no null and the cell's box of the window has no null : n, sx, sxx constant
//   CHIPNULL chip has nulls, box has none                                    : n, sx, sxx constant
//   GENERAL  the box contains null window pixels                             : all six sums
//   XY       like FAST, for planes that come with a summed-area table (sat_kernel.hip): the window-side sums sy, syy of a
//            null-free box are box sums of the IMAGE and are read from the table; only sxy is accumulated here
//   WN       window nulls in the box, null-free chip, planes with a table: n = pixels - nulls of the box, sy, syy are the
//            table's (a null is a zero in both sums); sx, sxx (chip pixels over the non-null window pixels) and sxy are accumulated
//   GC       GENERAL for a null-free chip: the chip-side masks are the compile-time pad masks of the row tasks, nothing is
//            derived from the chip dwords -- so nothing chip-derived has to stay in registers across the evaluation loops
enum { M_FAST = 0, M_CHIPNULL = 1, M_GENERAL = 2, M_XY = 3, M_WN = 4, M_GC = 5 };

template <class S> struct AccT { uint32_t n; S sx, sy, sxx, syy, sxy; };

__device__ __forceinline__ double dpp_add_f64(double x, int ctrl_sel)
{
    const long long b = __double_as_longlong(x);
    int lo = (int)(b & 0xffffffffll), hi = (int)(b >> 32);
    int olo, ohi;
    switch (ctrl_sel) {
    case 0: olo = __builtin_amdgcn_update_dpp(0, lo, 0xB1, 0xF, 0xF, true); ohi = __builtin_amdgcn_update_dpp(0, hi, 0xB1, 0xF, 0xF, true); break;
    case 1: olo = __builtin_amdgcn_update_dpp(0, lo, 0x4E, 0xF, 0xF, true); ohi = __builtin_amdgcn_update_dpp(0, hi, 0x4E, 0xF, 0xF, true); break;
    case 2: olo = __builtin_amdgcn_update_dpp(0, lo, 0x141, 0xF, 0xF, true); ohi = __builtin_amdgcn_update_dpp(0, hi, 0x141, 0xF, 0xF, true); break;
    default: olo = __builtin_amdgcn_update_dpp(0, lo, 0x140, 0xF, 0xF, true); ohi = __builtin_amdgcn_update_dpp(0, hi, 0x140, 0xF, 0xF, true); break;
    }
    return x + __longlong_as_double(((long long)ohi << 32) | (unsigned int)olo);
}

// ---- pixel policy: 8-bit integral imagery, 4 pixels per dword, exact integer sums --------------------
struct PxU8 {
    static constexpr int BPP = 1, G = 4, LOG2G = 2;
    static constexpr bool SRC16 = false;
    static constexpr bool INTEGER = true;                     // exact integer sums: null corrections may be applied in any order
    static constexpr bool SAT = true;                         // the planes come with a packed summed-area table (sum b | sum b^2 | nulls)
    static constexpr bool SATZ = false;                       // ... whose null counts live in a second table (u16 planes)
    static constexpr bool SAT_CHIP = true;                    // the chip's sums and null count are table look-ups too
    static constexpr bool WN = true;                          // three-sum body for window-null boxes (per config: PxCfg::WN)
    typedef unsigned long long SatT;
    __device__ static __forceinline__ unsigned long long sat_s(SatT q) { return (uint32_t)q & ((1u << kSatSqShift8) - 1u); }
    __device__ static __forceinline__ unsigned long long sat_ss(SatT q) { return (uint32_t)(q >> kSatSqShift8) & ((1u << (kSatNullShift8 - kSatSqShift8)) - 1u); }
    __device__ static __forceinline__ int sat_nulls(SatT q) { return (int)(q >> kSatNullShift8); }
    // window-side sums of a box from its table entry, in the units the kernel accumulates in (z = nulls in the box, k = the
    // window offset of the per-point-offset policy, npx = pixels of the box)
    __device__ static __forceinline__ void sat_win_sums(uint32_t &sy, uint32_t &syy, SatT q, int, int, int, double) { sy = (uint32_t)sat_s(q); syy = (uint32_t)sat_ss(q); }
    __device__ static __forceinline__ uint32_t px_at(const unsigned char *p) { return *p; }
    typedef uint32_t Sum;
    static constexpr uint32_t lowmask_c(int npx) { return npx >= 4 ? 0xffffffffu : ((1u << (8 * npx)) - 1u); }
    __device__ static __forceinline__ uint32_t lowmask(int npx) { return npx >= 4 ? 0xffffffffu : ((1u << (8 * npx)) - 1u); }
    __device__ static __forceinline__ int npx(uint32_t m) { return __popc(m & 0x01010101u); }
    // "x < MIN_DN" (:622,:631) and "not (x >= MIN_DN)" (:723) coincide for integers: null <=> DN == 0
    __device__ static __forceinline__ int nbad(uint32_t v, uint32_t keep, float) { return npx(keep) - __popc(nz80(v) >> 7); }
    __device__ static __forceinline__ int nexcl(uint32_t v, uint32_t keep, float thr) { return nbad(v, keep, thr); }
    __device__ static __forceinline__ uint32_t sanitize(uint32_t a, float) { return a; }     // nulls are already 0
    // may the kept bytes of v hold a null?  (zero-byte detector; a borrow can raise a false alarm next to a real zero byte,
    // never a miss: the caller then counts exactly)
    __device__ static __forceinline__ bool maybe_excl(uint32_t v, uint32_t keep, float) { return ((v - 0x01010101u) & ~v & 0x80808080u & keep) != 0u; }
    __device__ static __forceinline__ void chip_acc(Sum &sx, Sum &sxx, uint32_t a) { sx = dot4(a, 0x01010101u, sx); sxx = dot4(a, a, sxx); }
    template <int MODE, bool OPQ>
    __device__ static __forceinline__ void task(AccT<Sum> &acc, uint32_t a, uint32_t pad01, uint32_t padff, bool static_pad, uint32_t bw, float)
    {
        // byte mask of the chip group: 0xFF where the chip pixel is valid (null pixels and the pad bytes of the
        // last group are 0 in `a`), derived on the fly -- keeping it in registers would cost a second chip image.
        // The chip is loop-invariant, so the compiler WOULD hoist every mask out of the evaluation loops (80+
        // VGPRs for the big chips, i.e. spills): the empty asm makes `a` opaque per task (OPQ, big chips only --
        // the small chips have the registers and are faster with the hoisted masks).
        if (MODE == M_XY) {     // pad bytes and nulls are 0 in `a`: no mask at all.  (`static_pad` doubles as the accumulator choice:
            // a single chain of dependent dot4 would stall on its own latency; the caller alternates two and adds them at the end)
            if (static_pad) acc.sxy = dot4(a, bw, acc.sxy); else acc.sy = dot4(a, bw, acc.sy);
            return;
        }
        if (MODE == M_WN) {     // the chip has no null (its pad bytes are 0): only the window's nulls mask
            const uint32_t t = nz80(bw);
            acc.sx = dot4(a, t >> 7, acc.sx);
            acc.sxx = dot4(a & ff_from80(t), a, acc.sxx);
            acc.sxy = dot4(a, bw, acc.sxy);
            return;
        }
        if (MODE == M_GC && static_pad) {     // null-free chip, row task: the chip mask is the pad mask
            const uint32_t t = nz80(bw);
            const uint32_t mb01 = t >> 7, mbff = ff_from80(t);
            acc.n = dot4(pad01, mb01, acc.n);
            acc.sx = dot4(a, mb01, acc.sx);
            acc.sy = dot4(pad01, bw, acc.sy);
            acc.sxy = dot4(a, bw, acc.sxy);
            acc.sxx = dot4(a & mbff, a, acc.sxx);
            acc.syy = dot4(padff == 0xffffffffu ? bw : (bw & padff), bw, acc.syy);
            return;
        }
        if (OPQ && !(MODE == M_FAST && static_pad)) asm volatile("" : "+v"(a));
        const uint32_t mf = (MODE == M_FAST && static_pad) ? padff : ff_from80(nz80(a));
        if (MODE == M_FAST) {
            const uint32_t m01 = static_pad ? pad01 : (mf & 0x01010101u);
            const uint32_t mff = static_pad ? padff : mf;
            acc.sy = dot4(m01, bw, acc.sy);
            acc.syy = dot4((static_pad && padff == 0xffffffffu) ? bw : (bw & mff), bw, acc.syy);
            acc.sxy = dot4(a, bw, acc.sxy);
        } else if (MODE == M_CHIPNULL) {
            acc.sy = dot4(mf & 0x01010101u, bw, acc.sy);
            acc.syy = dot4(bw & mf, bw, acc.syy);
            acc.sxy = dot4(a, bw, acc.sxy);
        } else {
            const uint32_t t = nz80(bw);
            const uint32_t mb01 = t >> 7, mbff = ff_from80(t);
            const uint32_t ma01 = mf & 0x01010101u;
            acc.n = dot4(ma01, mb01, acc.n);
            acc.sx = dot4(a, mb01, acc.sx);          // a == 0 where the chip pixel is null
            acc.sy = dot4(ma01, bw, acc.sy);         // bw == 0 where the window pixel is null
            acc.sxy = dot4(a, bw, acc.sxy);
            acc.sxx = dot4(a & mbff, a, acc.sxx);
            acc.syy = dot4(bw & mf, bw, acc.syy);
        }
    }
    template <int LPC> __device__ static __forceinline__ Sum gsum(Sum v) { return group_sum<LPC>(v); }
    typedef uint32_t Store;                                    // how a reduced sum is parked in LDS
    __device__ static __forceinline__ Store bits(Sum v) { return v; }
    // NCC from exact integer sums (MIMC_module.c:734), f64, no contraction
    __device__ static __forceinline__ float ncc(const Store *sp, double, double, int, int)
    {
        const double dn = (double)sp[0], dsx = (double)sp[1], dsy = (double)sp[2];
        const double num = dn * (double)sp[5] - dsx * dsy;
        const double den = sqrt((dn * (double)sp[3] - dsx * dsx) * (dn * (double)sp[4] - dsy * dsy));
        return (float)(num / den);
    }
};

// ---- pixel policy: scaled-integer imagery q = value * 2^s with q < 4096 (12 bits), 2 pixels per dword.
//      Covers 12-bit DN and what GMA_float_conv2 makes of 8-bit images (gradients: integers <= 511;
//      Laplacian: multiples of 1/8, MIMC_main.c:175-196).  The reference's f32 products q_a*q_b/2^(s_a+s_b)
//      are exact (< 2^24 significant bits), so exact integer sums rescaled by powers of two in f64 are the
//      reference's sums.  Per-lane partial sums fit 32 bits; the cross-lane reduction is 64-bit. -----------
typedef unsigned short us2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t dot2(uint32_t a, uint32_t b, uint32_t c)
{
    return __builtin_amdgcn_udot2(__builtin_bit_cast(us2_t, a), __builtin_bit_cast(us2_t, b), c, false);
}
__device__ __forceinline__ uint32_t nz8000(uint32_t v) { return (((v & 0x7fff7fffu) + 0x7fff7fffu) | v) & 0x80008000u; }
__device__ __forceinline__ uint32_t ffff_from8000(uint32_t t) { return t | (t - (t >> 15)); }
__device__ __forceinline__ unsigned long long dpp_add_u64(unsigned long long x, int ctrl_sel)
{
    int lo = (int)(uint32_t)x, hi = (int)(uint32_t)(x >> 32);
    int olo, ohi;
    switch (ctrl_sel) {
    case 0: olo = __builtin_amdgcn_update_dpp(0, lo, 0xB1, 0xF, 0xF, true); ohi = __builtin_amdgcn_update_dpp(0, hi, 0xB1, 0xF, 0xF, true); break;
    case 1: olo = __builtin_amdgcn_update_dpp(0, lo, 0x4E, 0xF, 0xF, true); ohi = __builtin_amdgcn_update_dpp(0, hi, 0x4E, 0xF, 0xF, true); break;
    case 2: olo = __builtin_amdgcn_update_dpp(0, lo, 0x141, 0xF, 0xF, true); ohi = __builtin_amdgcn_update_dpp(0, hi, 0x141, 0xF, 0xF, true); break;
    default: olo = __builtin_amdgcn_update_dpp(0, lo, 0x140, 0xF, 0xF, true); ohi = __builtin_amdgcn_update_dpp(0, hi, 0x140, 0xF, 0xF, true); break;
    }
    return x + (((unsigned long long)(uint32_t)ohi << 32) | (uint32_t)olo);
}
struct PxU16 {
    static constexpr int BPP = 2, G = 2, LOG2G = 1;
    static constexpr bool SRC16 = false;
    static constexpr bool INTEGER = true;
    static constexpr bool SAT = true;                         // sum q | sum q^2 << 25 in one table, null counts in a second one
    static constexpr bool SATZ = true;
    static constexpr bool SAT_CHIP = true;
    static constexpr bool WN = true;                          // (-2 % on the Laplacian's small chips)
    typedef unsigned long long SatT;
    __device__ static __forceinline__ unsigned long long sat_s(SatT q) { return q & ((1ull << kSatSqShift16) - 1ull); }
    __device__ static __forceinline__ unsigned long long sat_ss(SatT q) { return q >> kSatSqShift16; }
    __device__ static __forceinline__ int sat_nulls(SatT) { return 0; }
    __device__ static __forceinline__ void sat_win_sums(unsigned long long &sy, unsigned long long &syy, SatT q, int, int, int, double) { sy = sat_s(q); syy = sat_ss(q); }
    __device__ static __forceinline__ uint32_t px_at(const unsigned char *p) { return *reinterpret_cast<const unsigned short *>(p); }
    typedef unsigned long long Sum;                    // per-lane partials stay < 2^32; the reduction needs 64 bits
    static constexpr uint32_t lowmask_c(int npx) { return npx >= 2 ? 0xffffffffu : (npx == 1 ? 0x0000ffffu : 0u); }
    __device__ static __forceinline__ uint32_t lowmask(int npx) { return npx >= 2 ? 0xffffffffu : (npx == 1 ? 0x0000ffffu : 0u); }
    __device__ static __forceinline__ int npx(uint32_t m) { return __popc(m & 0x00010001u); }
    __device__ static __forceinline__ int nbad(uint32_t v, uint32_t keep, float) { return npx(keep) - __popc(nz8000(v) >> 15); }   // null <=> q == 0
    __device__ static __forceinline__ int nexcl(uint32_t v, uint32_t keep, float thr) { return nbad(v, keep, thr); }
    __device__ static __forceinline__ uint32_t sanitize(uint32_t a, float) { return a; }
    __device__ static __forceinline__ bool maybe_excl(uint32_t v, uint32_t keep, float) { return ((v - 0x00010001u) & ~v & 0x80008000u & keep) != 0u; }
    __device__ static __forceinline__ void chip_acc(Sum &sx, Sum &sxx, uint32_t a)
    {
        sx = dot2(a, 0x00010001u, (uint32_t)sx); sxx = dot2(a, a, (uint32_t)sxx);
    }
    template <int MODE, bool OPQ>
    __device__ static __forceinline__ void task(AccT<Sum> &acc, uint32_t a, uint32_t, uint32_t padff, bool static_pad, uint32_t bw, float)
    {
        if (MODE == M_XY) {                                         // (see PxU8::task)
            if (static_pad) acc.sxy = dot2(a, bw, (uint32_t)acc.sxy); else acc.sy = dot2(a, bw, (uint32_t)acc.sy);
            return;
        }
        if (MODE == M_WN) {
            const uint32_t t = nz8000(bw);
            acc.sx = dot2(a, t >> 15, (uint32_t)acc.sx);
            acc.sxx = dot2(a & ffff_from8000(t), a, (uint32_t)acc.sxx);
            acc.sxy = dot2(a, bw, (uint32_t)acc.sxy);
            return;
        }
        if (MODE == M_GC && static_pad) {
            const uint32_t t = nz8000(bw);
            const uint32_t mb01 = t >> 15, mbff = ffff_from8000(t);
            const uint32_t p01 = padff & 0x00010001u;
            acc.n = dot2(p01, mb01, acc.n);
            acc.sx = dot2(a, mb01, (uint32_t)acc.sx);
            acc.sy = dot2(p01, bw, (uint32_t)acc.sy);
            acc.sxy = dot2(a, bw, (uint32_t)acc.sxy);
            acc.sxx = dot2(a & mbff, a, (uint32_t)acc.sxx);
            acc.syy = dot2(padff == 0xffffffffu ? bw : (bw & padff), bw, (uint32_t)acc.syy);
            return;
        }
        if (OPQ && !(MODE == M_FAST && static_pad)) asm volatile("" : "+v"(a)); // see PxU8::task: keeps the masks out of registers
        const uint32_t mf = (MODE == M_FAST && static_pad) ? padff : ffff_from8000(nz8000(a));
        if (MODE == M_FAST || MODE == M_CHIPNULL) {
            acc.sy = dot2(mf & 0x00010001u, bw, (uint32_t)acc.sy);
            acc.syy = dot2((MODE == M_FAST && static_pad && padff == 0xffffffffu) ? bw : (bw & mf), bw, (uint32_t)acc.syy);
            acc.sxy = dot2(a, bw, (uint32_t)acc.sxy);
        } else {
            const uint32_t t = nz8000(bw);
            const uint32_t mb01 = t >> 15, mbff = ffff_from8000(t);
            const uint32_t ma01 = mf & 0x00010001u;
            acc.n = dot2(ma01, mb01, acc.n);
            acc.sx = dot2(a, mb01, (uint32_t)acc.sx);
            acc.sy = dot2(ma01, bw, (uint32_t)acc.sy);
            acc.sxy = dot2(a, bw, (uint32_t)acc.sxy);
            acc.sxx = dot2(a & mbff, a, (uint32_t)acc.sxx);
            acc.syy = dot2(bw & mf, bw, (uint32_t)acc.syy);
        }
    }
    template <int LPC> __device__ static __forceinline__ Sum gsum(Sum v)
    {
        v = dpp_add_u64(v, 0); v = dpp_add_u64(v, 1); v = dpp_add_u64(v, 2); v = dpp_add_u64(v, 3);
        if (LPC >= 32) v += __shfl_xor(v, 16, 64);
        if (LPC >= 64) v += __shfl_xor(v, 32, 64);
        return v;
    }
    typedef unsigned long long Store;
    __device__ static __forceinline__ Store bits(Sum v) { return v; }
    __device__ static __forceinline__ float ncc(const Store *sp, double sa, double sb, int, int)
    {
        const double dn = (double)(uint32_t)sp[0];
        const double sx = (double)sp[1] * sa, sy = (double)sp[2] * sb;            // exact: powers of two
        const double sxx = (double)sp[3] * (sa * sa), syy = (double)sp[4] * (sb * sb), sxy = (double)sp[5] * (sa * sb);
        const double num = dn * sxy - sx * sy;
        const double den = sqrt((dn * sxx - sx * sx) * (dn * syy - sy * sy));
        return (float)(num / den);
    }
};

// ---- pixel policy: scaled-integer imagery whose LOCAL dynamic range fits 8 bits, read through a per-point offset.
//      The gradient filters of 8-bit images give 9-bit integers (1..511) whose range inside one chip / one search
//      window almost never exceeds 254: such a point is staged as q' = q - k (k = local minimum - 1, separately for
//      chip and window, nulls stay 0) and runs on the u8 machinery (dot4, 4 px per dword).  The exact sums of the
//      true values follow from the primed ones:  sx = sx' + ka n,  sxx = sxx' + 2 ka sx' + ka^2 n,
//      sxy = sxy' + kb sx' + ka sy' + ka kb n  (n = pixels that take part), all in 64-bit integers, so the NCC is
//      bit-identical to the u16 path's.  Points that do not fit are handed to the u16 kernel through fail_list. -------
struct PxU8o : PxU8 {
    static constexpr bool SRC16 = true;
    // the u16 planes' tables serve the window side: the true box sums are converted to the kernel's offset units,
    //   sum (q - k) = sum q - k m,   sum (q - k)^2 = sum q^2 - 2 k sum q + k^2 m,   m = non-null pixels of the box.
    // The chip side stays as it is (its offset is only known after the range scan, and the sums are taken while staging).
    static constexpr bool SAT = true;
    static constexpr bool SATZ = true;
    static constexpr bool SAT_CHIP = false;
    static constexpr bool WN = false;                         // (+13 % at ocw 15: the null-count look-up of every dirty box is a second table)
    __device__ static __forceinline__ unsigned long long sat_s(SatT q) { return q & ((1ull << kSatSqShift16) - 1ull); }
    __device__ static __forceinline__ unsigned long long sat_ss(SatT q) { return q >> kSatSqShift16; }
    __device__ static __forceinline__ int sat_nulls(SatT) { return 0; }
    __device__ static __forceinline__ void sat_win_sums(uint32_t &sy, uint32_t &syy, SatT q, int z, int k, int npx, double)
    {
        const long long m = npx - z, K = k, s1 = (long long)sat_s(q), s2 = (long long)sat_ss(q);
        sy = (uint32_t)(s1 - K * m);                                // sums of bytes / squared bytes over <= 81^2 pixels: < 2^32
        syy = (uint32_t)(s2 - 2 * K * s1 + K * K * m);
    }
    __device__ static __forceinline__ float ncc(const Store *sp, double sa, double sb, int ka, int kb)
    {
        const long long n = sp[0], sx_ = sp[1], sy_ = sp[2], sxx_ = sp[3], syy_ = sp[4], sxy_ = sp[5];
        const long long A = ka, B = kb;
        const long long sxi = sx_ + A * n, syi = sy_ + B * n;
        const long long sxxi = sxx_ + 2 * A * sx_ + A * A * n, syyi = syy_ + 2 * B * sy_ + B * B * n;
        const long long sxyi = sxy_ + B * sx_ + A * sy_ + A * B * n;
        const double dn = (double)n;
        const double sx = (double)sxi * sa, sy = (double)syi * sb;                  // exact: powers of two (as PxU16::ncc)
        const double sxx = (double)sxxi * (sa * sa), syy = (double)syyi * (sb * sb), sxy = (double)sxyi * (sa * sb);
        const double num = dn * sxy - sx * sy;
        const double den = sqrt((dn * sxx - sx * sx) * (dn * syy - sy * sy));
        return (float)(num / den);
    }
    // four u16 pixels (two dwords) -> four u8 pixels q - k (0 stays 0); bytes are confined even when q - k is out of range.
    // Packed 16-bit arithmetic: (q - k) * min(q, 1) per half-word, then one byte permute gathers the four low bytes.
    __device__ static __forceinline__ us2_t as_us2(uint32_t v) { return __builtin_bit_cast(us2_t, v); }
    __device__ static __forceinline__ uint32_t as_u32(us2_t v) { return __builtin_bit_cast(uint32_t, v); }
    __device__ static __forceinline__ uint32_t pack(uint2 v, int k)
    {
        const us2_t kk = {(unsigned short)k, (unsigned short)k}, one = {1, 1};
        const us2_t x = as_us2(v.x), y = as_us2(v.y);
        const us2_t dx = (x - kk) * __builtin_elementwise_min(x, one), dy = (y - kk) * __builtin_elementwise_min(y, one);
        return __builtin_amdgcn_perm(as_u32(dy), as_u32(dx), 0x06040200u);
    }
    // running min / max (two packed 16-bit lanes each; empty = 0xffff / 0) over the non-null pixels of four u16 pixels
    // selected by the byte mask `keep` (0x00 / 0xff per pixel)
    struct Range2 { us2_t mn, mx; };
    __device__ static __forceinline__ Range2 range_empty() { return Range2{us2_t{0xffff, 0xffff}, us2_t{0, 0}}; }
    __device__ static __forceinline__ void range4(uint2 v, uint32_t keep, Range2 &r)
    {
        const us2_t one = {1, 1};
        const uint32_t kx = __builtin_amdgcn_perm(keep, keep, 0x01010000u), ky = __builtin_amdgcn_perm(keep, keep, 0x03030202u);
        const us2_t x = as_us2(v.x & kx), y = as_us2(v.y & ky);
        r.mx = __builtin_elementwise_max(r.mx, __builtin_elementwise_max(x, y));
        // pixels that do not count (null or not kept) are 0 here: raise them to 0xffff for the minimum
        const us2_t fx = as_us2(as_u32(x) | as_u32(__builtin_elementwise_min(x, one) - one)), fy = as_us2(as_u32(y) | as_u32(__builtin_elementwise_min(y, one) - one));
        r.mn = __builtin_elementwise_min(r.mn, __builtin_elementwise_min(fx, fy));
    }
    __device__ static __forceinline__ void range_out(const Range2 &r, int &mn, int &mx)     // -> the scalar convention: empty = (1 << 20, -1)
    {
        const int a = min((int)r.mn.x, (int)r.mn.y), b = max((int)r.mx.x, (int)r.mx.y);
        mn = (b == 0) ? (1 << 20) : a; mx = (b == 0) ? -1 : b;
    }
};

// ---- pixel policy: arbitrary f32 imagery, 1 pixel per dword, f32 products + f64 sums (:726-730) ------
struct PxF32 {
    static constexpr int BPP = 4, G = 1, LOG2G = 0;
    static constexpr bool SRC16 = false;
    static constexpr bool INTEGER = false;
    static constexpr bool SAT = false, SATZ = false, SAT_CHIP = false, WN = false;
    typedef unsigned long long SatT;
    __device__ static __forceinline__ unsigned long long sat_s(SatT) { return 0u; }
    __device__ static __forceinline__ unsigned long long sat_ss(SatT) { return 0u; }
    __device__ static __forceinline__ int sat_nulls(SatT) { return 0; }
    __device__ static __forceinline__ void sat_win_sums(double &, double &, SatT, int, int, int, double) {}
    __device__ static __forceinline__ uint32_t px_at(const unsigned char *p) { return *reinterpret_cast<const uint32_t *>(p); }
    typedef double Sum;
    static constexpr uint32_t lowmask_c(int npx) { return npx >= 1 ? 0xffffffffu : 0u; }
    __device__ static __forceinline__ uint32_t lowmask(int npx) { return npx >= 1 ? 0xffffffffu : 0u; }
    __device__ static __forceinline__ int npx(uint32_t m) { return m ? 1 : 0; }
    __device__ static __forceinline__ int nbad(uint32_t v, uint32_t keep, float thr) { return (keep && __uint_as_float(v) < thr) ? 1 : 0; }       // :622
    __device__ static __forceinline__ int nexcl(uint32_t v, uint32_t keep, float thr) { return (keep && !(__uint_as_float(v) >= thr)) ? 1 : 0; }  // :723 (NaN too)
    __device__ static __forceinline__ uint32_t sanitize(uint32_t a, float thr) { return (__uint_as_float(a) >= thr) ? a : 0u; }  // excluded chip pixels -> 0.0
    __device__ static __forceinline__ bool maybe_excl(uint32_t v, uint32_t keep, float thr) { return keep && !(__uint_as_float(v) >= thr); }   // covers "< thr" (:622) too
    __device__ static __forceinline__ void chip_acc(Sum &sx, Sum &sxx, uint32_t a)
    {
        const float f = __uint_as_float(a);
        sx += (double)f; sxx += (double)(f * f);
    }
    template <int MODE, bool OPQ>
    __device__ static __forceinline__ void task(AccT<Sum> &acc, uint32_t au, uint32_t, uint32_t, bool static_pad, uint32_t bu, float thr)
    {
        const float a = __uint_as_float(au), b = __uint_as_float(bu);   // a is 0.0 for excluded chip pixels and unused slots
        if (MODE == M_FAST && static_pad && !OPQ) {
            // a row task of a null-free chip: every slot is a pixel (one pixel per dword, no pad); the idle lanes of a short
            // chip hold a = 0 and read the window's zero row, so they add nothing either way: no selects.  (Small chips only:
            // on the 61/81-row chips the freer schedule costs registers -- ocw 40: 284 B of spills, 78.6 -> 91.5 ms.)
            acc.sy += (double)b; acc.syy += (double)(b * b); acc.sxy += (double)(a * b);
        } else if (MODE == M_FAST) {
            const bool on = au != 0u;                                    // unused tail slots only (the chip has no null here)
            const float bm = on ? b : 0.0f;
            acc.sy += (double)bm; acc.syy += (double)(bm * bm); acc.sxy += (double)(on ? a * b : 0.0f);
        } else if (MODE == M_CHIPNULL) {
            const bool on = au != 0u;
            const float bm = on ? b : 0.0f;
            acc.sy += (double)bm; acc.syy += (double)(bm * bm); acc.sxy += (double)(on ? a * b : 0.0f);
        } else {
            const bool ok = (au != 0u) && (b >= thr);                     // null exclusion (:723)
            const float a2 = ok ? a : 0.0f, b2 = ok ? b : 0.0f;
            acc.n += ok ? 1u : 0u;
            acc.sx += (double)a2; acc.sy += (double)b2;
            acc.sxx += (double)(a2 * a2); acc.syy += (double)(b2 * b2); acc.sxy += (double)(a2 * b2);
        }
    }
    template <int LPC> __device__ static __forceinline__ Sum gsum(Sum v)
    {
        v = dpp_add_f64(v, 0); v = dpp_add_f64(v, 1); v = dpp_add_f64(v, 2); v = dpp_add_f64(v, 3);
        if (LPC >= 32) v += __shfl_xor(v, 16, 64);
        if (LPC >= 64) v += __shfl_xor(v, 32, 64);
        return v;
    }
    typedef unsigned long long Store;
    __device__ static __forceinline__ Store bits(Sum v) { return (unsigned long long)__double_as_longlong(v); }
    __device__ static __forceinline__ float ncc(const Store *sp, double, double, int, int)
    {
        const double dn = (double)(uint32_t)sp[0];
        const double sx = __longlong_as_double((long long)sp[1]), sy = __longlong_as_double((long long)sp[2]);
        const double sxx = __longlong_as_double((long long)sp[3]), syy = __longlong_as_double((long long)sp[4]);
        const double sxy = __longlong_as_double((long long)sp[5]);
        const double num = dn * sxy - sx * sy;
        const double den = sqrt((dn * sxx - sx * sx) * (dn * syy - sy * sy));
        return (float)(num / den);
    }
};

// ---- pixel policy: f32 planes whose pixels are all INTEGERS in [0, 2^20), possibly in units of 1/8 (16-bit DN, the integer gradients and the Laplacian of such images).  The reference's arithmetic is the
//      f32 policy's -- f32 products that round above 2^24 (T1), f64 sums -- but every term is an integer below 2^40 and every
//      sum an integer below 2^53: exact in any order, so the window-side sums of a null-free box (sum b, sum fl(b b)) and the
//      chip's (sum a, sum fl(a a), nulls) are box queries of a 16-byte summed-area table (sat_kernel.hip) and the evaluation
//      of such a box keeps ONE product stream, sxy = sum fl(a b), instead of three. -------------------------------------------
struct PxF32i : PxF32 {
    static constexpr bool SAT = true, SATZ = false, SAT_CHIP = true;
    static constexpr bool WN = true;                          // (-10 % at ocw 16 and 40: three f64 streams instead of six)
    typedef Sat2 SatT;
    __device__ static __forceinline__ double sat_s(const SatT &q) { return (double)(q.a & ((1ull << kSatNullShiftF) - 1ull)); }
    __device__ static __forceinline__ double sat_ss(const SatT &q) { return (double)q.b; }
    __device__ static __forceinline__ int sat_nulls(const SatT &q) { return (int)(q.a >> kSatNullShiftF); }
    __device__ static __forceinline__ void sat_win_sums(double &sy, double &syy, const SatT &q, int, int, int, double sc) { sy = sat_s(q) * sc; syy = sat_ss(q) * (sc * sc); }   // exact: powers of two
    template <int MODE, bool OPQ>
    __device__ static __forceinline__ void task(AccT<Sum> &acc, uint32_t au, uint32_t p01, uint32_t pff, bool static_pad, uint32_t bu, float thr)
    {
        if (MODE == M_XY) {                                         // (see PxU8::task: `static_pad` picks one of two independent chains)
            const double pr = (double)(__uint_as_float(au) * __uint_as_float(bu));
            if (static_pad) acc.sxy += pr; else acc.sy += pr;
            return;
        }
        if (MODE == M_WN) {                                         // integral DN: a null window pixel is 0.0 (and makes the product 0)
            const float a = __uint_as_float(au), b = __uint_as_float(bu);
            const float a2 = (b != 0.0f) ? a : 0.0f;                  // (-0.0 is a null too)
            acc.sx += (double)a2; acc.sxx += (double)(a2 * a2); acc.sxy += (double)(a * b);
            return;
        }
        PxF32::template task<MODE, OPQ>(acc, au, p01, pff, static_pad, bu, thr);
    }
};

template <class P_, int OCW_, int LPC_, int NW_ = 1, int MINW_ = 2, bool CHL_ = false, bool MANY_ = false, bool COMPACT_ = false>
struct PxCfg {
    typedef P_ P;
    // Compact LDS form, chosen at launch time when the regular carve crosses an occupancy step that this one does not (large
    // windows: BASELINE C4's 133^2 window needs 50.6 KB per point = 3 points per CU; 40.6 KB = 4):
    //   * the NCC cache is two-level -- a 16-bit entry per compact cell (bit 15 requested, bit 14 value present, low bits: slot)
    //     and f32 value slots only for the cells a point really asks for -- instead of one f32 per cell of the whole grid;
    //     a look-up costs a second, dependent LDS access;
    //   * the null lists hold 16-bit entries (window coordinates below 256: checked at launch).
    static constexpr bool COMPACT = COMPACT_;
    // the table corners of a batch stay in registers until its finish pass (the look-up's latency hides behind the evaluation rounds,
    // eight registers more are live across them): pays on the one-wave u8 kernels (BASELINE C2: 2.86 -> 2.83 ms, ocw 7 2.05 -> 2.03),
    // costs where registers are short (u8 ocw 40 13.22 -> 13.38, integral f32 ocw 16 8.34 -> 8.59)
    static constexpr bool SAT_DEFER = std::is_same<P_, PxU8>::value && NW_ == 1;
    static constexpr int MINW = MINW_;                       // occupancy target, waves per SIMD
    static constexpr int OCW = OCW_, LPC = LPC_;
    static constexpr int NW = NW_, NT = 64 * NW_;            // waves / threads per grid point (one workgroup)
    static constexpr int CW = 2 * OCW + 1, NPX = CW * CW;
    static constexpr int GPR = (CW + P::G - 1) / P::G;       // dwords per chip row
    // A chip with fewer rows than the group has lanes (61 rows on 64 lanes, 15 on 16, 31 on 32) is ONE row round with a
    // few idle lanes (they hold zeros and read the window's all-zero T4 row) instead of being all "tail": row tasks have
    // compile-time pad masks and contiguous LDS reads, tail tasks derive their masks from the chip dword by dword.
    static constexpr bool SHORT = CW < LPC;
    static constexpr int RF = SHORT ? 1 : CW / LPC;          // full rounds: rows l + LPC*i
    static constexpr int REM = SHORT ? 0 : CW - RF * LPC;    // leftover rows, split into single-group tasks
    static constexpr int TT = (REM * GPR + LPC - 1) / LPC;   // tail tasks per lane
    static constexpr int LASTN = CW - P::G * (GPR - 1);      // valid pixels of the last dword of a row (1..G)
    static constexpr uint32_t LASTFF = P::lowmask_c(LASTN);
    static constexpr uint32_t LAST01 = LASTFF & 0x01010101u;
    static constexpr int CPR = 64 / LPC;                     // cells per evaluation round
    // big chips: keep chip-derived masks out of registers (see PxU8::task).  Small chips: where the hoisted masks made the kernel
    // spill -- measured, same box, with / without: u8 ocw 16 3.10 / 3.22 ms, d/dx ocw 15 4.17 / 4.27; where it did not spill the
    // hoisted masks stay (u8 ocw 7 2.33 / 2.28, ocw 15 3.25 / 3.14 -- eight mask-deriving tail tasks --, Laplacian ocw 15 6.21 / 6.04)
    // WN mode (window-null boxes of a null-free chip: three sums + table) per config, by measurement: every policy that allows it,
    // but on 8-bit planes only where a lane owns two or more full chip rows and there are no null lists (u8 ocw 16: 3.14 -> 3.08 ms;
    // ocw 7 / 15 / 40: +1.4 / +1.1 / +4 %, the byte masks being cheap next to the second table look-up)
    static constexpr bool WN = P_::WN && (!std::is_same<P_, PxU8>::value || ((SHORT ? 1 : CW / LPC_) >= 2 && !((LPC_ >= 64) && P_::INTEGER)));
    static constexpr bool kOpqSmall = MIMC3_OPQ_SMALL && ((std::is_same<P_, PxU8>::value && OCW_ == 16) || (std::is_same<P_, PxU8o>::value && (OCW_ == 15 || OCW_ == 16)));
    static constexpr bool OPQ = LPC_ >= 64 || kOpqSmall;
    // Big chips with exact integer sums: a cell whose box (or whose chip) holds null pixels is evaluated as the FAST body
    // (3 dot products per dword) plus CORRECTIONS summed over short lists of the null pixels -- window nulls take chip
    // values out of n, sx, sxx; chip nulls take window values out of sy, syy -- instead of the six-sum GENERAL body
    // (17 VALU per dword).  One wave evaluates one cell, so 64 lanes share the list walk.
    static constexpr bool SPARSE = (LPC_ >= 64) && P::INTEGER;
    static constexpr int CPITCH = 4 * GPR;                   // LDS chip copy: bytes per row
    // The evaluation reads the chip from its LDS copy instead of registers (SPARSE configs keep that copy anyway): the u16
    // policy's 81-row chip is 52 dwords per lane next to a 42-dword window row in flight, which does not fit 128 or 168
    // VGPRs; from LDS the kernel runs at twice the occupancy without spills (the kernel is VALU-bound, LDS has headroom).
    static constexpr bool CHIP_LDS = CHL_;                   // (the f32 policies' chips too: 35 .. 83 dwords per lane next to f64 accumulators -- see launch_match_f32x)
    static constexpr bool CHIP_COPY = SPARSE || CHIP_LDS;    // an LDS copy of the chip exists
    // Pivot sets of more than 64 (the 21x21 set of the control-point stage): the pivots beyond the first 64 have no lane of
    // their own; their speculative climbs are recorded in LDS so that the exact replay stays on the fast form.  A separate
    // instantiation: the plain configurations keep their register allocation.
    static constexpr bool MANYP = MANY_;
    // Tail tasks without masks (sparse-correction configs): a tail dword is all chip pixels except the LAST dword of a
    // chip row, whose pad bytes carry no pixel.  If no lane owns more than one such dword, the evaluation runs every tail
    // task with a full mask and takes the pad pixels of that one dword out again (one extra dword per lane and cell)
    // instead of deriving a byte mask from the chip dword in every tail task.
    static constexpr bool one_pad_task_per_lane()
    {
        for (int ln = 0; ln < LPC_; ln++) {
            int cnt = 0;
            for (int k = 0; k < TT; k++) {
                const int tt = ln + LPC_ * k;
                if (tt < REM * GPR && tt % GPR == GPR - 1) cnt++;
            }
            if (cnt > 1) return false;
        }
        return true;
    }
    static constexpr bool FULLTAIL = SPARSE && !CHIP_LDS && TT >= 4 && LASTN < P::G && one_pad_task_per_lane();   // (measured: no gain for the chip-from-LDS u16 ocw 40 form, a loss with one tail task: ocw 32)
};
static constexpr int kLwCap = 1024;    // window-null list entries (x | y << 16); more -> the point falls back to GENERAL
static constexpr int kLcCap = 512;     // chip-null list entries

struct U8Point {
    int dx2, dy2, Dx2, Dy2, csx, csy, ncell;
    int sh;            // byte phase of window column 0 inside its aligned dword
    int PW;            // LDS window pitch, bytes
    int zrow;          // an all-zero LDS row: the never-written last window row (T4), or one extra row behind a full-square search area
    float thr;         // smallest f32 whose f64 value is >= MIN_DN
};

// One evaluation round: lane group g (LPC lanes) evaluates the cell whose chip origin in window
// coordinates is (cx, cy) (== compact cell coordinates).  Returns group-reduced sums in every lane.
template <class C, int MODE, bool REDUCE = true, bool FULLTAIL = false>
__device__ __forceinline__ AccT<typename C::P::Sum> eval_round(const unsigned char *W, const U8Point &pt, int cx, int cy, int l,
                                          const uint32_t (&A)[C::RF > 0 ? C::RF : 1][C::GPR],
                                          const uint32_t (&AT)[C::TT > 0 ? C::TT : 1],
                                          const int (&toff)[C::TT > 0 ? C::TT : 1],
                                          AccT<typename C::P::Sum> acc = AccT<typename C::P::Sum>{0, 0, 0, 0, 0, 0},
                                          const unsigned char *CH = nullptr)
{
    typedef typename C::P P;
    [[maybe_unused]] const typename P::Sum sy_in = acc.sy;
    const int X = pt.sh + cx;                                       // pixel offset of the box inside the LDS row
    const uint32_t s = (uint32_t)((X & (P::G - 1)) * P::BPP);       // byte phase inside the first dword
    const unsigned char *base = W + cy * pt.PW + 4 * (X >> P::LOG2G);
    constexpr int NLD = C::GPR + (P::G > 1 ? 1 : 0);                // dwords a row task reads
    if constexpr (C::CHIP_LDS) {
        // chip AND window from LDS: the row is walked in chunks of eight dwords by a real loop (nothing to index in registers),
        // which bounds the loads in flight -- a fully unrolled 41-dword row keeps 83 values live
        constexpr int KC = 8;
#pragma unroll
        for (int i = 0; i < C::RF; i++) {
            const bool idle = C::SHORT && l >= C::CW;
            const uint32_t *rp = reinterpret_cast<const uint32_t *>(idle ? W + pt.zrow * pt.PW : base + (l + C::LPC * i) * pt.PW);
            const uint32_t *cp = reinterpret_cast<const uint32_t *>(CH + (idle ? 0 : (l + C::LPC * i)) * C::CPITCH);
#pragma unroll 1
            for (int j0 = 0; j0 < C::GPR - 1; j0 += KC) {             // the full dwords of the row
                uint32_t w[KC + 1], a[KC];
#pragma unroll
                for (int k = 0; k <= KC; k++) w[k] = rp[j0 + k];      // (reads a few dwords past a short last chunk: inside the pitch + next row)
#pragma unroll
                for (int k = 0; k < KC; k++) a[k] = (j0 + k < C::GPR - 1 && !idle) ? cp[j0 + k] : 0u;
#pragma unroll
                for (int k = 0; k < KC; k++) {
                    const uint32_t bw = (P::G > 1) ? alignb(w[k + 1], w[k], s) : w[k];
                    if (j0 + k < C::GPR - 1) P::template task<MODE, C::OPQ>(acc, a[k], 0x01010101u, 0xffffffffu, true, bw, pt.thr);
                }
            }
            {                                                         // the last dword of the row (pad mask)
                constexpr int j = C::GPR - 1;
                const uint32_t w0 = rp[j], w1 = (P::G > 1) ? rp[j + 1] : 0u;
                const uint32_t bw = (P::G > 1) ? alignb(w1, w0, s) : w0;
                const uint32_t av = idle ? 0u : cp[j];
                P::template task<MODE, C::OPQ>(acc, av, C::LAST01, C::LASTFF, true, bw, pt.thr);
            }
        }
    } else {
#pragma unroll
    for (int i = 0; i < C::RF; i++) {
        const uint32_t *rp = reinterpret_cast<const uint32_t *>(base + (l + C::LPC * i) * pt.PW);
        if (C::SHORT && l >= C::CW) rp = reinterpret_cast<const uint32_t *>(W + pt.zrow * pt.PW);   // idle lane: the zero row
        uint32_t w[NLD];
#pragma unroll
        for (int j = 0; j < NLD; j++) w[j] = rp[j];
#pragma unroll
        for (int j = 0; j < C::GPR; j++) {
            const uint32_t bw = (P::G > 1) ? alignb(w[j + (P::G > 1 ? 1 : 0)], w[j], s) : w[j];
            const uint32_t p01 = (j == C::GPR - 1) ? C::LAST01 : 0x01010101u;
            const uint32_t pff = (j == C::GPR - 1) ? C::LASTFF : 0xffffffffu;
            if constexpr (MODE == M_XY) P::template task<M_XY, C::OPQ>(acc, A[i][j], 0, 0, (MIMC3_XY_ACC < 2) || ((i * C::GPR + j) & 1) == 0, bw, pt.thr);
            else P::template task<MODE, C::OPQ>(acc, A[i][j], p01, pff, true, bw, pt.thr);
        }
    }
    }
#pragma unroll
    for (int k = 0; k < C::TT; k++) {
        const uint32_t *rp = reinterpret_cast<const uint32_t *>(base + toff[k]);
        const uint32_t bw = (P::G > 1) ? alignb(rp[P::G > 1 ? 1 : 0], rp[0], s) : rp[0];
        uint32_t av = AT[k];
        if constexpr (C::CHIP_LDS) {
            // chip dword of tail task k: row RF*LPC + tt / GPR, dword tt % GPR (0 for the lanes past the last task)
            const int tt = l + C::LPC * k;
            av = tt < C::REM * C::GPR ? *reinterpret_cast<const volatile uint32_t *>(CH + (C::RF * C::LPC + tt / C::GPR) * C::CPITCH + 4 * (tt % C::GPR)) : 0u;
        }
        if constexpr (MODE == M_XY) {
            // only sxy: the chip dword is 0 in its pad bytes, at its nulls and in the lanes past the last task -- no mask matters
            P::template task<M_XY, C::OPQ>(acc, av, 0, 0, (MIMC3_XY_ACC < 2) || (k & 1) == 0, bw, pt.thr);
        }
        else if constexpr (FULLTAIL && MODE == M_FAST) {
            // full mask: the caller corrects for pad pixels and lists the tail rows' nulls.  Only the last task has lanes
            // without a dword (they hold a zero chip dword but would still read a window dword): their window is zeroed.
            uint32_t bwt = bw;
            if (k == C::TT - 1 && (C::REM * C::GPR) % C::LPC != 0) bwt = (l + C::LPC * k < C::REM * C::GPR) ? bw : 0u;
            P::template task<M_FAST, C::OPQ>(acc, av, P::BPP == 1 ? 0x01010101u : 0x00010001u, 0xffffffffu, true, bwt, pt.thr);
        }
        else P::template task<MODE, C::OPQ>(acc, av, 0, 0, false, bw, pt.thr);   // tail tasks: pad/null masks come from the chip dword itself
    }
    if constexpr (MODE == M_XY && MIMC3_XY_ACC >= 2) { acc.sxy += acc.sy - sy_in; acc.sy = sy_in; }   // fold the second chain (sy carried the caller's corrections in)
    if (!REDUCE) return acc;                                        // lane-local partial sums (the caller reduces / parks them)
    acc.sxy = P::template gsum<C::LPC>(acc.sxy);
    if (MODE != M_XY && MODE != M_WN) { acc.sy = P::template gsum<C::LPC>(acc.sy); acc.syy = P::template gsum<C::LPC>(acc.syy); }
    if (MODE == M_WN) { acc.sx = P::template gsum<C::LPC>(acc.sx); acc.sxx = P::template gsum<C::LPC>(acc.sxx); }
    if (MODE == M_GENERAL || MODE == M_GC) {
        acc.n = group_sum<C::LPC>(acc.n); acc.sx = P::template gsum<C::LPC>(acc.sx); acc.sxx = P::template gsum<C::LPC>(acc.sxx);
    }
    return acc;
}

static constexpr int kStatW = 16;      // diagnostics: 8 phase clocks + cell counts (clean, dirty, evaluate calls) + null-list use
#ifndef MIMC3_SUM_BATCH
#define MIMC3_SUM_BATCH 32
#endif
static constexpr int kSumBatch = MIMC3_SUM_BATCH;   // cells whose reduced sums are parked in LDS before the f64 finish

#define MIMC3_STAMP(i)                                                                         \
    if (p.stats) {                                                                             \
        const unsigned long long t_now = __builtin_amdgcn_s_memtime();                         \
        if (threadIdx.x == 0) p.stats[kStatW * (size_t)blockIdx.x + i] += t_now - t_prev;                   \
        t_prev = t_now;                                                                        \
    }

template <class C>
__global__ __launch_bounds__(C::NT, C::MINW) void match_ncc_dlc_px(MatchU8Args p)
{
    typedef typename C::P P;
    typedef typename P::Sum Sum;
    // summed-area tables: every policy that has them, except the many-pivot forms -- those only serve the control-point stage, whose
    // chip atlases (a few hundred latency-bound points per match, planes that are mostly border) are not worth building tables for
    constexpr bool kSat = P::SAT && !C::MANYP, kSatChip = P::SAT_CHIP && kSat, kSatZ = P::SATZ && kSat;
    unsigned long long t_prev = p.stats ? __builtin_amdgcn_s_memtime() : 0ull;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l = lane & (C::LPC - 1), grp = lane / C::LPC;
    constexpr int OCW = C::OCW, CW = C::CW, GPR = C::GPR, NT = C::NT, NW = C::NW;

    int gidx = blockIdx.x;
    if (p.point_list) {                                      // list mode: the points another kernel handed over
        if (gidx >= *p.point_count) return;
        gidx = p.point_list[gidx];
    } else {
        const int nb = gridDim.x, per = nb >> 3;
        if (per > 0 && gidx < per * 8) gidx = (gidx & 7) * per + (gidx >> 3);   // XCD-contiguous point order
    }
    if (gidx >= p.N) return;
    if (p.point_flags && p.point_flags[gidx] != (uint8_t)p.flag_value) return;      // flag mode: the points the matrix-core kernel left

    const unsigned char *chip_pl = p.swap ? p.p1 : p.p0;
    const unsigned char *win_pl = p.swap ? p.p0 : p.p1;
    const int Wp = p.Wp, PAD = p.pad;

    // ---- point header -------------------------------------------------------------------------
    const double *row = p.xyuvav + (size_t)p.xy_stride * (size_t)gidx + p.xy_col;   // (u, v): columns 2, 3 of an xyuvav row, or a packed [N][2] array
    const int u0 = (int)row[0], v0 = (int)row[1];
    const int64_t pbeg = p.piv_off[gidx];
    const int npiv = (int)(p.piv_off[gidx + 1] - pbeg);
    const int32_t *pv_g = p.piv_uv + 2 * pbeg;
    U8Point pt;
    {
        const int lu = pv_g[2 * (npiv - 1)], lv = pv_g[2 * (npiv - 1) + 1];
        pt.dx2 = (lu < 0 ? -lu : lu) + OCW + 2;
        pt.dy2 = (lv < 0 ? -lv : lv) + OCW + 2;
    }
    // full-square search area (get_offset_image hands find_ncc_peak a whole image chip, MIMC_module.c:347): every row and
    // column is written, there is no empty last row / column (T4 applies to the DLC window of extract_sarea only)
    const bool full_win = p.win_half > 0;
    if (full_win) { pt.dx2 = p.win_half; pt.dy2 = p.win_half; }
    pt.Dx2 = 2 * pt.dx2 + 1; pt.Dy2 = 2 * pt.dy2 + 1;
    pt.csx = pt.Dx2 - 2 * OCW + 1; pt.csy = pt.Dy2 - 2 * OCW + 1;
    pt.ncell = pt.csx * pt.csy;
    pt.PW = p.lds_pw;
    pt.zrow = 2 * pt.dy2 + (full_win ? 1 : 0);
    const int wu0 = u0 + p.off_u - pt.dx2 + PAD;     // plane column of window column 0
    const int wv0 = v0 + p.off_v - pt.dy2 + PAD;     // plane row of window row 0
    pt.sh = wu0 & (P::G - 1);                        // pixel phase of window column 0 inside its aligned dword
    pt.thr = p.thr;
    // scaled-integer planes store q = value * 2^s: the sums are rescaled exactly (powers of two) in the finish
    const double sc_chip = p.swap ? p.scale1 : p.scale0, sc_win = p.swap ? p.scale0 : p.scale1;
    // Summed-area tables of the planes (sat_kernel.hip): the chip's sum a, sum a^2 and null count, the null count of the
    // window's written area -- two box queries, issued here so that they are in flight while LDS is cleared
    typedef typename P::SatT SatT;
    const SatT *sat_chip = nullptr, *sat_win = nullptr;
    const uint32_t *satz_win = nullptr;
    SatT chipQ{};
    int chip_nulls = 0, win_nulls = 0;
    if constexpr (kSat) {
        sat_chip = reinterpret_cast<const SatT *>(p.swap ? p.sat1 : p.sat0);
        sat_win = reinterpret_cast<const SatT *>(p.swap ? p.sat0 : p.sat1);
        satz_win = reinterpret_cast<const uint32_t *>(p.swap ? p.satz0 : p.satz1);
        const int wc = 2 * pt.dx2 + (full_win ? 1 : 0), wr = 2 * pt.dy2 + (full_win ? 1 : 0);          // the written area (:869-886)
        if constexpr (kSatZ) win_nulls = (int)sat_box(satz_win, p.sat_ws, wu0, wv0, wc, wr);
        else if constexpr (std::is_same<SatT, unsigned long long>::value && !P::SATZ && P::BPP == 1)
            win_nulls = sat_nulls_u8(reinterpret_cast<const unsigned long long *>(sat_win), p.sat_ws, wu0, wv0, wc, wr, lane);   // exact for any window size (one packed query is not, beyond 8,224 px)
        else win_nulls = P::sat_nulls(sat_box(sat_win, p.sat_ws, wu0, wv0, wc, wr));
        if constexpr (kSatChip) {
            chipQ = sat_box(sat_chip, p.sat_ws, u0 - OCW + PAD, v0 - OCW + PAD, CW, CW);
            if constexpr (kSatZ) chip_nulls = (int)sat_box(reinterpret_cast<const uint32_t *>(p.swap ? p.satz1 : p.satz0), p.sat_ws, u0 - OCW + PAD, v0 - OCW + PAD, CW, CW);
            else chip_nulls = P::sat_nulls(chipQ);
        }
    }
    (void)sat_chip; (void)sat_win; (void)satz_win; (void)chipQ; (void)chip_nulls; (void)win_nulls;

    // ---- LDS carve ------------------------------------------------------------------------------
    unsigned char *W = smem;                                              // [Dy2][PW]
    // NCC cache: one f32 per compact cell, read with ONE LDS access (the climb's lookups are latency chains):
    // kUnknown = never asked for, kWanted = queued for evaluation, anything else = the NCC (NaN is a value)
    float *val = reinterpret_cast<float *>(smem + p.lds_off_val);         // [csy][csx]
    uint32_t *vis = reinterpret_cast<uint32_t *>(smem + p.lds_off_vis);   // visited bits, rows padded to words
    uint16_t *list = reinterpret_cast<uint16_t *>(smem + p.lds_off_list); // cells queued for evaluation (cy<<8|cx): clean boxes from the front, dirty from the back
    typedef typename P::Store Store;
    Store *sums = reinterpret_cast<Store *>(smem + p.lds_off_sums);        // [kSumBatch][6] reduced sums
    int32_t *pivs = reinterpret_cast<int32_t *>(smem + p.lds_off_piv);    // [npiv][2]
    const int lcap = p.lds_list_cap;

    const int vpitch = ((pt.csx + 31) >> 5) << 5;          // visited bits: one row = whole 32-bit words
    for (int i = tid; i < ((pt.csy * vpitch) >> 5); i += NT) vis[i] = 0u;
    for (int i = tid; i < 2 * npiv; i += NT) pivs[i] = pv_g[i];
    // control words: [0] queue fill, clean | dirty << 16, [3] queue overflow,
    // [4] null pixels in the window, [5..8] their bounding box (x0,x1,y0,y1), [9] driver decision
    int32_t *qcnt = reinterpret_cast<int32_t *>(sums + 6 * kSumBatch);   // behind the sums
    // ([11..14]: PxU8o only -- min/max of the non-null window and chip pixels)
    for (int i = tid; i < 6 * kSumBatch; i += NT) sums[i] = 0;             // parking slots of the atomically parked sums
    // [16] window-null list length, [17] chip-null list length, [18] a list overflowed (SPARSE configs)
    // ([19..22]: PxU8o only -- the same from the planes' 16x16-pixel tile ranges: a cheap bound tried first)
    if (tid < 32) qcnt[tid] = (tid == 5 || tid == 7 || tid == 11 || tid == 13 || tid == 19 || tid == 21) ? (1 << 20) : ((tid == 6 || tid == 8 || tid == 12 || tid == 14 || tid == 20 || tid == 22) ? -1 : 0);
    unsigned char *CH = smem + p.lds_off_chip;                             // SPARSE: chip copy [CW][CPITCH]
    uint32_t *Lw = reinterpret_cast<uint32_t *>(smem + p.lds_off_lw);      // SPARSE: null pixels of the window (x | y << 16)
    uint32_t *Lc = reinterpret_cast<uint32_t *>(smem + p.lds_off_lc);      // SPARSE: null pixels of the chip's row-task rows
    (void)CH; (void)Lw; (void)Lc;
    __syncthreads();
    // NCC of a compact cell, kUnknown when it has not been evaluated
    // scan centres pass the boundary test (:703), i.e. lie in [2, cs-3]; their 3x3 and the fit's 3x3 reach [1, cs-2]: the
    // outermost ring of compact cells is never touched and has no cache word
    const int vpitchc = pt.csx - 2;
    uint16_t *map16 = reinterpret_cast<uint16_t *>(val);                   // COMPACT: [csy-2][csx-2] entries, value slots behind
    float *vals = reinterpret_cast<float *>(smem + (C::COMPACT ? p.lds_off_vals : 0));
    (void)map16; (void)vals;
    if constexpr (C::COMPACT) { for (int i = tid; i < (vpitchc * (pt.csy - 2) + 1) / 2; i += NT) reinterpret_cast<uint32_t *>(val)[i] = 0u; }
    else {
        // (16 bytes per store: the carve is 16-byte aligned and whatever follows the cache within the last quad is cleared or written
        //  later -- the visited bits start on their own 16-byte boundary)
        const int ncw = vpitchc * (pt.csy - 2);
        const float4 u4 = make_float4(kUnknown, kUnknown, kUnknown, kUnknown);
        for (int i = tid; i < (ncw >> 2); i += NT) reinterpret_cast<float4 *>(val)[i] = u4;
        if (tid < (ncw & 3)) val[(ncw & ~3) + tid] = kUnknown;
    }
    auto cell_index = [&](int cx, int cy) __attribute__((always_inline)) -> int { return (cy - 1) * vpitchc + (cx - 1); };
    auto lookup = [&](int cx, int cy) __attribute__((always_inline)) -> float {
        if constexpr (C::COMPACT) {
            const uint32_t e = map16[cell_index(cx, cy)];
            const float v = vals[e & 0x3fffu];                             // (slot 0 for an unknown cell: read and dropped)
            return (e & 0x8000u) ? ((e & 0x4000u) ? v : kWanted) : kUnknown;
        } else return val[cell_index(cx, cy)];
    };
    // claims the cell for evaluation: true for exactly one requester
    auto claim = [&](int cx, int cy) __attribute__((always_inline)) -> bool {
        if constexpr (C::COMPACT) {
            const int i = cell_index(cx, cy);
            const uint32_t bit = 0x8000u << (16 * (i & 1));
            return (atomicOr(reinterpret_cast<uint32_t *>(val) + (i >> 1), bit) & bit) == 0u;
        } else return atomicCAS(reinterpret_cast<uint32_t *>(&val[cell_index(cx, cy)]), __float_as_uint(kUnknown), __float_as_uint(kWanted)) == __float_as_uint(kUnknown);
    };
    // COMPACT: the winner of a cell gives it a value slot (qcnt[2] counts them; more than the carve holds -> the general kernel)
    auto give_slot = [&](int cx, int cy, int slot) __attribute__((always_inline)) { map16[cell_index(cx, cy)] = (uint16_t)(0x8000u | (uint32_t)slot); };
    auto store_ncc = [&](int cx, int cy, float v) __attribute__((always_inline)) {
        if constexpr (C::COMPACT) {
            const int i = cell_index(cx, cy);
            const uint32_t slot = map16[i] & 0x3fffu;
            vals[slot] = v;
            map16[i] = (uint16_t)(0xC000u | slot);
        } else val[cell_index(cx, cy)] = v;
    };
    (void)give_slot;
    // null lists: 32-bit entries x | y << 16, or (COMPACT) 16-bit entries x | y << 8; absent = all ones
    auto nl_get = [&](const uint32_t *L, int i) __attribute__((always_inline)) -> uint32_t {
        if constexpr (C::COMPACT) {
            const uint32_t e = reinterpret_cast<const uint16_t *>(L)[i];
            return e == 0xffffu ? 0xffffffffu : ((e & 0xffu) | ((e >> 8) << 16));
        } else return L[i];
    };
    auto nl_put = [&](uint32_t *L, int i, uint32_t x, uint32_t y) __attribute__((always_inline)) {
        if constexpr (C::COMPACT) reinterpret_cast<uint16_t *>(L)[i] = (uint16_t)(x | (y << 8));
        else L[i] = x | (y << 16);
    };

    // ---- stage the window as aligned dwords; count nulls and bound them (a5, a6) -----------------
    int ka = 0, kb = 0;                                      // PxU8o: per-point offsets of chip and window (0 otherwise)
    int bad_win = 0, exc_win = 0;                            // "x < MIN_DN" count (:631) / pixels the NCC loop skips (:723)
    int nbx0 = 1 << 20, nbx1 = -1, nby0 = 1 << 20, nby1 = -1;   // bounding box of the skipped pixels (window coords, dword-granular in x)
    {
        const int wcols = 2 * pt.dx2 + (full_win ? 1 : 0), wrows = 2 * pt.dy2 + (full_win ? 1 : 0);   // written area (:869-886)
        const int nd = (pt.sh + wcols + P::G - 1) >> P::LOG2G;            // aligned dwords per row
        // every thread keeps ONE dword column c and walks down the rows r0, r0 + rstep, ...: column masks, addresses and the
        // LDS offset are loop invariants / plain increments (one division per thread instead of one per dword)
        // (a row wider than the workgroup -- long corridors on one-wave configs -- takes several column sweeps: cstep)
        const int rstep = NT / nd > 0 ? NT / nd : 1;                     // rows covered per sweep
        const int c_first = tid % nd, r0 = tid / nd, cstep = nd <= NT ? nd : NT;
        const bool col_on = r0 < rstep;                                   // the last NT - rstep*nd threads have no column
        const int lastp = (pt.sh + wcols) & (P::G - 1);                   // valid pixels in the last dword (0 = all)
        const uint32_t first_ff = ~P::lowmask(pt.sh);
        const uint32_t last_ff = lastp ? P::lowmask(lastp) : 0xffffffffu;
        auto col_keep = [&](int c) __attribute__((always_inline)) -> uint32_t {
            uint32_t k = 0xffffffffu;
            if (c == 0) k &= first_ff;
            if (c == nd - 1) k &= last_ff;
            return k;
        };
        const uint32_t *gbase = reinterpret_cast<const uint32_t *>(win_pl + ((size_t)wv0 * Wp + (wu0 - pt.sh)) * P::BPP);
        const int gpitch = (Wp * P::BPP) >> 2;
        // PxU8o: the source planes are u16; four pixels = one aligned uint2
        const uint2 *gbase16 = reinterpret_cast<const uint2 *>(win_pl + ((size_t)wv0 * Wp + (wu0 - pt.sh)) * 2);
        const int gpitch16 = Wp >> 2;
        (void)gbase16; (void)gpitch16;
        bool have_k = false;                                  // PxU8o: offsets found from the tile ranges (the exact scan is skipped)
        if constexpr (P::SRC16) {
            if (p.rt0) {
                // A bound first: the ranges of the 16x16-pixel tiles the window / the chip touch (a superset of their pixels).
                // If both fit 8 bits, offsets from these bounds are as good as the exact ones -- the finish rebuilds the true
                // integer sums from ANY offsets that keep every pixel inside a byte -- and the full scan below is not needed.
                const uint32_t *wrt = p.swap ? p.rt0 : p.rt1, *crt = p.swap ? p.rt1 : p.rt0;
                int tmn = 0xffff, tmx = 0, cmn2 = 0xffff, cmx2 = 0;
                {
                    const int tx0 = wu0 >> 4, ty0 = wv0 >> 4, ntx = ((wu0 + wcols - 1) >> 4) - tx0 + 1, nty = ((wv0 + wrows - 1) >> 4) - ty0 + 1;
                    for (int i = tid; i < ntx * nty; i += NT) {
                        const int iy = i / ntx, ix = i - iy * ntx;
                        const uint32_t t = wrt[(ty0 + iy) * p.rt_tw + tx0 + ix];
                        tmn = min(tmn, (int)(t & 0xffffu)); tmx = max(tmx, (int)(t >> 16));
                    }
                    const int cu0 = u0 - OCW + PAD, cv0 = v0 - OCW + PAD;
                    const int cx0 = cu0 >> 4, cy0 = cv0 >> 4, ncx = ((cu0 + CW - 1) >> 4) - cx0 + 1, ncy = ((cv0 + CW - 1) >> 4) - cy0 + 1;
                    for (int i = tid; i < ncx * ncy; i += NT) {
                        const int iy = i / ncx, ix = i - iy * ncx;
                        const uint32_t t = crt[(cy0 + iy) * p.rt_tw + cx0 + ix];
                        cmn2 = min(cmn2, (int)(t & 0xffffu)); cmx2 = max(cmx2, (int)(t >> 16));
                    }
                }
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) {
                    tmn = min(tmn, __shfl_xor(tmn, o, 64)); tmx = max(tmx, __shfl_xor(tmx, o, 64));
                    cmn2 = min(cmn2, __shfl_xor(cmn2, o, 64)); cmx2 = max(cmx2, __shfl_xor(cmx2, o, 64));
                }
                if (NW > 1) {
                    if (lane == 0) { atomicMin(&qcnt[19], tmn); atomicMax(&qcnt[20], tmx); atomicMin(&qcnt[21], cmn2); atomicMax(&qcnt[22], cmx2); }
                    __syncthreads();
                    tmn = qcnt[19]; tmx = qcnt[20]; cmn2 = qcnt[21]; cmx2 = qcnt[22];
                }
                if ((tmx == 0 || tmx - tmn <= 254) && (cmx2 == 0 || cmx2 - cmn2 <= 254)) {
                    kb = tmx == 0 ? 0 : tmn - 1;
                    ka = cmx2 == 0 ? 0 : cmn2 - 1;
                    have_k = true;
                }
            }
        }
        if constexpr (P::SRC16) if (!have_k) {
            // pass 1: local range of the window and of the chip; a point that does not fit 8 bits goes to the u16 kernel
            // (both scans fetch eight values before they use any: the loops are chains of global loads otherwise)
            int mn = 1 << 20, mx = -1;
            PxU8o::Range2 rw = PxU8o::range_empty();
            if (col_on)
                for (int c = c_first; c < nd; c += cstep) {
                    const uint32_t keep = col_keep(c);
                    for (int rb = r0; rb < wrows; rb += 8 * rstep) {
                        uint2 t[8];
#pragma unroll
                        for (int k = 0; k < 8; k++) { const int r = rb + k * rstep; t[k] = gbase16[(size_t)(r < wrows ? r : rb) * gpitch16 + c]; }
#pragma unroll
                        for (int k = 0; k < 8; k++) if (rb + k * rstep < wrows) PxU8o::range4(t[k], keep, rw);
                    }
                }
            PxU8o::range_out(rw, mn, mx);
            int cmn = 1 << 20, cmx = -1;
            PxU8o::Range2 rc = PxU8o::range_empty();
            {
                // the chip as aligned 4-pixel groups: row rr, group cc of the (GPR + 1) groups that cover its CW pixels
                const int cu0 = u0 - OCW + PAD, cv0 = v0 - OCW + PAD;
                const int cph = cu0 & 3;
                const uint2 *cb = reinterpret_cast<const uint2 *>(chip_pl + ((size_t)cv0 * Wp + (cu0 - cph)) * 2);
                constexpr int GR = GPR + 1, NG = CW * GR;
                for (int q0 = tid; q0 < NG; q0 += 8 * NT) {
                    uint2 t[8];
                    int rr[8], cc[8];
#pragma unroll
                    for (int k = 0; k < 8; k++) {
                        const int q = q0 + k * NT < NG ? q0 + k * NT : q0;
                        rr[k] = q / GR; cc[k] = q - rr[k] * GR;
                        t[k] = cb[(size_t)rr[k] * (Wp >> 2) + cc[k]];
                    }
#pragma unroll
                    for (int k = 0; k < 8; k++) {
                        if (q0 + k * NT >= NG) continue;
                        // pixels of the group that belong to the chip: columns cph .. cph + CW - 1 of the row
                        uint32_t keep = 0u;
#pragma unroll
                        for (int b = 0; b < 4; b++) { const int col = 4 * cc[k] + b - cph; if (col >= 0 && col < CW) keep |= 0xffu << (8 * b); }
                        PxU8o::range4(t[k], keep, rc);
                    }
                }
            }
            PxU8o::range_out(rc, cmn, cmx);
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                mn = min(mn, __shfl_xor(mn, o, 64)); mx = max(mx, __shfl_xor(mx, o, 64));
                cmn = min(cmn, __shfl_xor(cmn, o, 64)); cmx = max(cmx, __shfl_xor(cmx, o, 64));
            }
            if (lane == 0) { atomicMin(&qcnt[11], mn); atomicMax(&qcnt[12], mx); atomicMin(&qcnt[13], cmn); atomicMax(&qcnt[14], cmx); }
            __syncthreads();
            mn = qcnt[11]; mx = qcnt[12]; cmn = qcnt[13]; cmx = qcnt[14];
            const bool fit = (mx < mn || mx - mn <= 254) && (cmx < cmn || cmx - cmn <= 254);
            if (!fit) {
                if (tid == 0) p.fail_list[atomicAdd(p.fail_count, 1)] = gidx;
                return;
            }
            kb = mx < mn ? 0 : mn - 1;
            ka = cmx < cmn ? 0 : cmn - 1;
        }
        // The row walk is a chain of global loads: eight rows are fetched before any of them is used (one memory latency per
        // eight rows instead of one per row), and the rare null handling runs after the batch on the values in registers.
        // (32-bit dword offsets from the point's uniform base: one add per load instead of a 64-bit multiply-add; whole
        //  batches of eight rows carry no bounds tests, only the last partial batch does)
        const uint32_t gstep = (uint32_t)(rstep * (P::SRC16 ? gpitch16 : gpitch));     // dwords (uint2s) between this thread's rows
        const int wstep = rstep * pt.PW;
        if (col_on)
        for (int c = c_first; c < nd; c += cstep) {
            const uint32_t keep = col_keep(c);
            const int x0 = P::G * c - pt.sh;                              // window column of the dword's first pixel
            bool hit = false;                                             // this column holds excluded pixels (x range of the null box)
            constexpr int KB = MIMC3_STAGE_KB;
            auto batch = [&](auto tail_c, auto nonull_c, int rb) __attribute__((always_inline)) {
                constexpr bool TAIL = decltype(tail_c)::value;
                constexpr bool NONULL = decltype(nonull_c)::value;       // the table says the written area holds no null: plain copy
                uint32_t v[KB];
                [[maybe_unused]] uint2 v16[KB];
                const uint32_t g0 = (uint32_t)rb * (uint32_t)(P::SRC16 ? gpitch16 : gpitch) + (uint32_t)c;
#pragma unroll
                for (int k = 0; k < KB; k++) {
                    const uint32_t gi = (TAIL && rb + k * rstep >= wrows) ? g0 : g0 + (uint32_t)k * gstep;   // rows past the end re-read the batch's first row (discarded)
                    if constexpr (P::SRC16) v16[k] = gbase16[gi];
                    else v[k] = gbase[gi];
                }
                unsigned char *wp = W + rb * pt.PW + 4 * c;
                bool susp = false;
#pragma unroll
                for (int k = 0; k < KB; k++) {
                    if constexpr (P::SRC16) v[k] = PxU8o::pack(v16[k], kb);
                    v[k] &= keep;                                         // pixels outside the written columns -> 0 (covers T4 column)
                    if (!TAIL || rb + k * rstep < wrows) {
                        *reinterpret_cast<uint32_t *>(wp + k * wstep) = v[k];
                        if constexpr (!NONULL) susp = susp || P::maybe_excl(v[k], keep, pt.thr);
                    }
                }
                if (!NONULL && susp) {                                    // rare: count exactly, bound, list
#pragma unroll
                    for (int k = 0; k < KB; k++) {
                        const int r = rb + k * rstep;
                        if ((TAIL && r >= wrows) || !P::maybe_excl(v[k], keep, pt.thr)) continue;
                        bad_win += P::nbad(v[k], keep, pt.thr);
                        const int nz = P::nexcl(v[k], keep, pt.thr);
                        exc_win += nz;
                        if (!nz) continue;
                        hit = true;
                        nby0 = min(nby0, r); nby1 = max(nby1, r);
                        if constexpr (C::SPARSE) {
                            int at = atomicAdd(&qcnt[16], nz);
                            if (at + nz > kLwCap) qcnt[18] = 1;
                            else {
#pragma unroll
                                for (int q = 0; q < P::G; q++) {
                                    const uint32_t pm = P::lowmask(1) << (8 * P::BPP * q);
                                    if ((keep & pm) && !(v[k] & pm)) nl_put(Lw, at++, (uint32_t)(x0 + q), (uint32_t)r);
                                }
                            }
                        }
                    }
                }
            };
            int rb = r0;
            if (kSat && win_nulls == 0) {
                for (; rb + (KB - 1) * rstep < wrows; rb += KB * rstep) batch(std::false_type{}, std::true_type{}, rb);
                if (rb < wrows) batch(std::true_type{}, std::true_type{}, rb);
            } else {
                for (; rb + (KB - 1) * rstep < wrows; rb += KB * rstep) batch(std::false_type{}, std::false_type{}, rb);
                if (rb < wrows) batch(std::true_type{}, std::false_type{}, rb);
            }
            if (hit) { nbx0 = min(nbx0, x0); nbx1 = max(nbx1, x0 + P::G - 1); }
        }
        // T4: the last window row is never written by the reference -> zeros; also clear the dwords
        // after each row's last written dword (read by the sliding loads of the right-most cells)
        const int ndz = pt.PW >> 2;
        for (int c = tid; c < ndz; c += NT) *reinterpret_cast<uint32_t *>(W + wrows * pt.PW + 4 * c) = 0u;
        for (int r = tid; r < wrows; r += NT)
            for (int c = nd; c < ndz; c++) *reinterpret_cast<uint32_t *>(W + r * pt.PW + 4 * c) = 0u;
        if (!(kSat && win_nulls == 0)) {             // (a window the table calls null-free was copied without looking: nothing to reduce)
        bad_win = wave_sum_i(bad_win); exc_win = wave_sum_i(exc_win);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            nbx0 = min(nbx0, __shfl_xor(nbx0, o, 64)); nbx1 = max(nbx1, __shfl_xor(nbx1, o, 64));
            nby0 = min(nby0, __shfl_xor(nby0, o, 64)); nby1 = max(nby1, __shfl_xor(nby1, o, 64));
        }
        }
        if (NW > 1) {                                        // combine the waves' partial results through LDS
            if (lane == 0) {
                atomicAdd(&qcnt[4], bad_win); atomicAdd(&qcnt[10], exc_win);
                atomicMin(&qcnt[5], nbx0); atomicMax(&qcnt[6], nbx1); atomicMin(&qcnt[7], nby0); atomicMax(&qcnt[8], nby1);
            }
            __syncthreads();
            bad_win = qcnt[4]; exc_win = qcnt[10]; nbx0 = qcnt[5]; nbx1 = qcnt[6]; nby0 = qcnt[7]; nby1 = qcnt[8];
        }
    }
    if (p.debug_stop == 1) return;
    MIMC3_STAMP(0)
    const bool win_clean = (exc_win == 0);                   // nothing the NCC loop would skip inside the written area
    if (!full_win) bad_win += pt.Dx2 + pt.Dy2 - 1;           // + the never-written last row and column

    // ---- chip -> registers (a4): every lane group holds the whole chip --------------------------
    constexpr int RFA = C::RF > 0 ? C::RF : 1, TTA = C::TT > 0 ? C::TT : 1;
    uint32_t A[RFA][GPR], AT[TTA];
    int toff[TTA];
    int padoff = -1;                                         // FULLTAIL configs: LDS offset of this lane's row-end tail dword (or none)
    int bad_chip = 0, exc_chip = 0;
    bool chip_susp = false;                                  // some chip dword of this lane may hold a null (integer policies: counted afterwards)
    Sum SX = 0, SXX = 0;
    {
        const int cu0 = u0 - OCW + PAD, cv0 = v0 - OCW + PAD;
        const int sap = cu0 & (P::G - 1);
        const uint32_t sa = (uint32_t)(sap * P::BPP);
        const uint32_t *gbase = reinterpret_cast<const uint32_t *>(chip_pl + ((size_t)cv0 * Wp + (cu0 - sap)) * P::BPP);
        const int gpitch = (Wp * P::BPP) >> 2;
        const uint2 *gbase16 = reinterpret_cast<const uint2 *>(chip_pl + ((size_t)cv0 * Wp + (cu0 - sap)) * 2);   // PxU8o source
        const int gpitch16 = Wp >> 2;
        (void)gbase16; (void)gpitch16;
        // aligned dword j of chip row `row` (PxU8o: converted from the u16 plane through the chip offset)
        auto chip_dword = [&](int row, int j) __attribute__((always_inline)) -> uint32_t {
            if constexpr (P::SRC16) return PxU8o::pack(gbase16[(size_t)row * gpitch16 + j], ka);
            else return gbase[(size_t)row * gpitch + j];
        };
        constexpr int NLD = GPR + (P::G > 1 ? 1 : 0);
#pragma unroll
        for (int i = 0; i < C::RF; i++) {
            uint32_t g[NLD];
            const bool rowok = !C::SHORT || l < C::CW;                  // SHORT: lanes past the last chip row hold zeros
#pragma unroll
            for (int j = 0; j < NLD; j++) g[j] = chip_dword(rowok ? l + C::LPC * i : 0, j);
#pragma unroll
            for (int j = 0; j < GPR; j++) {
                uint32_t a = (P::G > 1) ? alignb(g[j + (P::G > 1 ? 1 : 0)], g[j], sa) : g[j];
                const uint32_t pff = rowok ? ((j == GPR - 1) ? C::LASTFF : 0xffffffffu) : 0u;
                a &= pff;
                if constexpr (kSatChip) { }                                                       // counts and sums come from the table
                else if constexpr (P::INTEGER) chip_susp = chip_susp || P::maybe_excl(a, pff, pt.thr);    // the exact counts are taken afterwards, and only then
                else { bad_chip += P::nbad(a, pff, pt.thr); exc_chip += P::nexcl(a, pff, pt.thr); a = P::sanitize(a, pt.thr); }
                A[i][j] = a;
                if constexpr (!kSatChip) P::chip_acc(SX, SXX, a);
                if constexpr (C::CHIP_COPY) {
                    if ((j % NW) == wave && rowok) *reinterpret_cast<uint32_t *>(CH + (l + C::LPC * i) * C::CPITCH + 4 * j) = a;   // every wave holds the whole chip: each writes a share of the copy
                }
            }
        }
#pragma unroll
        for (int k = 0; k < C::TT; k++) {
            const int tt = l + C::LPC * k;
            const bool on = tt < C::REM * GPR;
            const int rr = C::RF * C::LPC + (on ? tt / GPR : 0), j = on ? tt % GPR : 0;
            const uint32_t g0 = chip_dword(rr, j);
            uint32_t a = (P::G > 1) ? alignb(chip_dword(rr, j + (P::G > 1 ? 1 : 0)), g0, sa) : g0;
            const uint32_t pff = on ? ((j == GPR - 1) ? C::LASTFF : 0xffffffffu) : 0u;
            a &= pff;
            if constexpr (kSatChip) { }
            else if constexpr (P::INTEGER) chip_susp = chip_susp || P::maybe_excl(a, pff, pt.thr);
            else { bad_chip += P::nbad(a, pff, pt.thr); exc_chip += P::nexcl(a, pff, pt.thr); a = P::sanitize(a, pt.thr); }
            AT[k] = a;
            toff[k] = rr * pt.PW + 4 * j;
            if (C::FULLTAIL && on && j == GPR - 1) padoff = toff[k];
            if constexpr (!kSatChip) P::chip_acc(SX, SXX, a);
            if constexpr (C::CHIP_COPY) {                  // tail rows: in the LDS copy (window nulls look chip values up there);
                if ((k % NW) == wave && on) *reinterpret_cast<uint32_t *>(CH + rr * C::CPITCH + 4 * j) = a;   // their own nulls are masked by the tail tasks
            }
        }
        // (with a table the chip's null count is known: only the sparse-correction configs look at the pixels again, to LIST them)
        if constexpr (kSatChip) chip_susp = C::SPARSE && chip_nulls != 0;
        if (P::INTEGER && chip_susp) {                        // rare: the lane's chip dwords again, counted exactly (and listed)
#pragma unroll
            for (int i = 0; i < C::RF; i++) {
                const bool rowok = !C::SHORT || l < C::CW;
#pragma unroll
                for (int j = 0; j < GPR; j++) {
                    const uint32_t pff = rowok ? ((j == GPR - 1) ? C::LASTFF : 0xffffffffu) : 0u;
                    const uint32_t a = A[i][j];
                    if (!P::maybe_excl(a, pff, pt.thr)) continue;
                    bad_chip += P::nbad(a, pff, pt.thr);
                    const int nzc = P::nexcl(a, pff, pt.thr);
                    exc_chip += nzc;
                    if constexpr (C::SPARSE) {
                        if (wave == 0 && rowok && nzc) {
                            int at = atomicAdd(&qcnt[17], nzc);
                            if (at + nzc > kLcCap) qcnt[18] = 1;
                            else {
#pragma unroll
                                for (int k = 0; k < P::G; k++) {
                                    const uint32_t pm = P::lowmask(1) << (8 * P::BPP * k);
                                    if ((pff & pm) && !(a & pm)) nl_put(Lc, at++, (uint32_t)(P::G * j + k), (uint32_t)(l + C::LPC * i));
                                }
                            }
                        }
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < C::TT; k++) {
                const int tt = l + C::LPC * k;
                const bool on = tt < C::REM * GPR;
                const int j = on ? tt % GPR : 0;
                const uint32_t pff = on ? ((j == GPR - 1) ? C::LASTFF : 0xffffffffu) : 0u;
                if (P::maybe_excl(AT[k], pff, pt.thr)) {
                    bad_chip += P::nbad(AT[k], pff, pt.thr);
                    const int nzc = P::nexcl(AT[k], pff, pt.thr);
                    exc_chip += nzc;
                    if constexpr (C::FULLTAIL || (kSat && C::SPARSE)) {   // maskless tail tasks (and every XY body): the tail rows' nulls are corrected from the list too
                        if (wave == 0 && on && nzc) {
                            int at = atomicAdd(&qcnt[17], nzc);
                            if (at + nzc > kLcCap) qcnt[18] = 1;
                            else {
#pragma unroll
                                for (int q = 0; q < P::G; q++) {
                                    const uint32_t pm = P::lowmask(1) << (8 * P::BPP * q);
                                    if ((pff & pm) && !(AT[k] & pm)) nl_put(Lc, at++, (uint32_t)(P::G * j + q), (uint32_t)(C::RF * C::LPC + tt / GPR));
                                }
                            }
                        }
                    }
                }
            }
        }
        if constexpr (kSatChip) {
            bad_chip = exc_chip = chip_nulls;                // null <=> DN == 0 for integral DN: both counts (:622, :723)
            SX = (Sum)P::sat_s(chipQ); SXX = (Sum)P::sat_ss(chipQ);
            if constexpr (!P::INTEGER) { SX = (Sum)((double)SX * sc_chip); SXX = (Sum)((double)SXX * (sc_chip * sc_chip)); }   // f32 planes of scaled integers: back to pixel units (exact)
        } else {
            bad_chip = (int)group_sum<C::LPC>((uint32_t)bad_chip); exc_chip = (int)group_sum<C::LPC>((uint32_t)exc_chip);
            SX = P::template gsum<C::LPC>(SX); SXX = P::template gsum<C::LPC>(SXX);
        }
    }
    const uint32_t NV = (uint32_t)(C::NPX - exc_chip);       // chip pixels that take part: n, sx, sxx are constants when the box is clean
    const int clean_mode = (exc_chip == 0) ? M_FAST : M_CHIPNULL;
    MIMC3_STAMP(1)
    if (p.debug_stop == 2) { if (tid == 0) p.out[3 * (size_t)gidx] = (float)((uint32_t)P::bits(SX) + (uint32_t)P::bits(SXX) + bad_chip + A[0][0] + AT[0]); return; }
    __syncthreads();   // single-wave workgroup: orders the LDS stores above before the reads below
    const bool sparse_on = C::SPARSE && qcnt[18] == 0;     // both null lists complete
    const int nLw = sparse_on ? qcnt[16] : 0, nLc = sparse_on ? qcnt[17] : 0;
    (void)nLw; (void)nLc;
    if (p.stats && tid == 0) {
        p.stats[kStatW * (size_t)blockIdx.x + 11] += sparse_on ? 1 : 0;
        p.stats[kStatW * (size_t)blockIdx.x + 12] += (unsigned)nLw; p.stats[kStatW * (size_t)blockIdx.x + 13] += (unsigned)nLc;
    }

    // ---- validity (a6, :635) --------------------------------------------------------------------
    {
        const float max_ratio = 0.8f;
        const float rc = (float)bad_chip / (float)(CW * CW);
        const float rw = (float)bad_win / (float)(pt.Dx2 * pt.Dy2);
        if (rc > max_ratio || rw > max_ratio) {
            if (tid == 0) {
                const float nanv = __builtin_nanf("");
                p.out[3 * (size_t)gidx + 0] = nanv;
                p.out[3 * (size_t)gidx + 1] = nanv;
                p.out[3 * (size_t)gidx + 2] = -3.0f;
            }
            return;
        }
    }
    static_assert(OCW >= 2, "T4 cmap cells are reachable by the fit only when ocw == 1; not instantiated");
    // a cell's 33x33 (CW x CW) box of the window is null-free iff it avoids the null bounding box and
    // the zero last row/column (T4)
    auto box_clean = [&](int cx, int cy) __attribute__((always_inline)) -> bool {
        if (!full_win && (cx == pt.csx - 2 || cy == pt.csy - 2)) return false;
        if (win_clean) return true;
        return (cx > nbx1) || (cx + CW - 1 < nbx0) || (cy > nby1) || (cy + CW - 1 < nby0);
    };

    // ---- request queue: a cell is requested at most once (compare-and-swap kUnknown -> kWanted on its cache word); clean
    //      boxes are queued from the front of `list`, dirty boxes from the back; ONE packed counter (clean | dirty << 16)
    //      so that a request costs two dependent LDS round trips.  A batch that outgrows the queue hands the point over
    //      to the general kernel (qcnt[3]).
    auto request = [&](int cx, int cy) __attribute__((always_inline)) {
        if (!claim(cx, cy)) return;                                            // asked for or known already
        if constexpr (C::COMPACT) {
            const int slot = atomicAdd(&qcnt[2], 1);
            if (slot >= p.lds_nslot) { qcnt[3] = 1; return; }
            give_slot(cx, cy, slot);
        }
        const bool cl = box_clean(cx, cy);
        const uint32_t q = (uint32_t)atomicAdd(&qcnt[0], cl ? 1 : (1 << 16));
        const int ia = (int)(q & 0xffffu), ib = (int)(q >> 16);
        if (ia + ib + 1 > lcap) { qcnt[3] = 1; return; }
        const uint16_t packed = (uint16_t)((cy << 8) | cx);
        if (cl) list[ia] = packed; else list[lcap - 1 - ib] = packed;
    };
    auto inside = [&](int pu, int pvv) __attribute__((always_inline)) -> bool {     // the reference's boundary test (:703), true = scan allowed
        return !(pu - OCW <= 1 || pu + OCW >= pt.Dx2 - 1 || pvv - OCW <= 1 || pvv + OCW >= pt.Dy2 - 1);
    };
    // One scan of the reference's 3x3 loop (:719-741) on the cached values around centre (cu, cv), from running maximum `sm`:
    // returns -2 if a cell is not evaluated yet, else the 3x3 index that took the maximum (-1: none did) and the new `sm`.
    // The sequential "if (v > max) { max = v; arg = j; }" ends on the FIRST cell that attains the overall maximum, if that
    // exceeds the old one; NaN never wins (max3 skips NaNs like the compare does).  Unknown / wanted cells hold 3.0 / 4.0,
    // above every NCC, so one maximum answers "all known?" as well.
    auto scan9 = [&](int cu, int cv, float &sm) __attribute__((always_inline)) -> int {
        float v[9];
#pragma unroll
        for (int j = 0; j < 9; j++) v[j] = lookup(cu + (j / 3 - 1) - OCW, cv + (j % 3 - 1) - OCW);
        const float m = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(v[0], v[1]), v[2]), __builtin_fmaxf(__builtin_fmaxf(v[3], v[4]), v[5])),
                                        __builtin_fmaxf(__builtin_fmaxf(v[6], v[7]), v[8]));
        if (m >= 2.5f) return -2;
        if (!(m > sm)) return -1;
        int mv = 8;
#pragma unroll
        for (int j = 7; j >= 0; j--) mv = (v[j] == m) ? j : mv;
        sm = m;
        return mv;
    };
    // request the whole 3x3 around compact cell (cx0, cy0): the nine compare-and-swaps are issued together, one bump of
    // the packed counter reserves the queue entries of the cells this lane won
    auto request9 = [&](int cx0, int cy0) __attribute__((always_inline)) {
        bool mine[9];
#pragma unroll
        for (int j = 0; j < 9; j++) mine[j] = claim(cx0 + (j / 3 - 1), cy0 + (j % 3 - 1));
        uint32_t won = 0;
#pragma unroll
        for (int j = 0; j < 9; j++)
            if (mine[j]) won |= 1u << j;
        if (!won) return;
        [[maybe_unused]] int slot0 = 0;
        if constexpr (C::COMPACT) {
            slot0 = atomicAdd(&qcnt[2], __popc(won));
            if (slot0 + __popc(won) > p.lds_nslot) { qcnt[3] = 1; return; }
        }
        // clean boxes of the 3x3 from three column tests and three row tests (bit j = 3 * column + row, as above): a box is
        // clean if its column range or its row range misses the null bounding box, and it does not touch the T4 row / column
        // (the T4 row / column is cell csy - 2 / csx - 2: only a 3x3 centred on csy - 3 / csx - 3 holds it, as its last row / column)
        const uint32_t row_t4 = (!full_win && cy0 == pt.csy - 3) ? 4u : 0u;
        const bool col_t4 = !full_win && cx0 == pt.csx - 3;
        uint32_t cl;
        if (win_clean) {                                     // (wave-uniform: 46 % of BASELINE C2's windows hold no null)
            cl = (7u & ~row_t4) * (col_t4 ? 0x09u : 0x49u);
        } else {
            uint32_t row_ok = 0u;
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const int cy = cy0 + k - 1;
                row_ok |= (cy > nby1 || cy + CW - 1 < nby0) ? (1u << k) : 0u;
            }
            cl = 0u;
#pragma unroll
            for (int i = 0; i < 3; i++) {
                const int cx = cx0 + i - 1;
                const bool col_ok = cx > nbx1 || cx + CW - 1 < nbx0;
                const uint32_t part = (i == 2 && col_t4) ? 0u : ((col_ok ? 7u : row_ok) & ~row_t4);
                cl |= part << (3 * i);
            }
        }
        const uint32_t wa = won & cl, wb = won & ~cl;
        const int na = __popc(wa), nb = __popc(wb);
        const uint32_t q = (uint32_t)atomicAdd(&qcnt[0], na | (nb << 16));
        int ia = (int)(q & 0xffffu), ib = (int)(q >> 16);
        if (ia + ib + na + nb > lcap) { qcnt[3] = 1; return; }
#pragma unroll
        for (int j = 0; j < 9; j++) {
            if (!((won >> j) & 1u)) continue;
            const int cx = cx0 + (j / 3 - 1), cy = cy0 + (j % 3 - 1);
            const uint16_t packed = (uint16_t)((cy << 8) | cx);
            if ((wa >> j) & 1u) list[ia++] = packed; else list[lcap - 1 - ib++] = packed;
            if constexpr (C::COMPACT) give_slot(cx, cy, slot0++);
        }
    };
    // round 0 = the certain set: every pivot whose start passes the boundary test scans its whole 3x3
    for (int k = tid; k < npiv; k += NT) {
        const int pu = pivs[2 * k] + pt.dx2, pvv = pivs[2 * k + 1] + pt.dy2;
        if (!inside(pu, pvv)) continue;
        request9(pu - OCW, pvv - OCW);
    }
    __syncthreads();
    if (p.debug_stop == 3) return;
    MIMC3_STAMP(2)

    // evaluates the `cnt` cells ids[0], ids[dir], ids[2*dir], ... in mode `mode`; NCC -> val[]
    //
    // How a cell's sums reach the f64 finish.  Groups of 16 lanes (small chips) reduce entirely with DPP row operations and
    // lane 0 stores.  Groups of 32 / 64 lanes would need ds_bpermute stages (an LDS round trip each, six sums in a row =
    // a long serial tail per cell): the integer policies instead reduce each 16-lane row with DPP and let the row leaders
    // ADD their partial sums into the cell's parking slot with LDS atomics (no return value: nothing waits on them; integer
    // adds commute, so the result is exact and deterministic).  The slots are zero between batches.
    constexpr bool kAPark = (C::LPC >= 32) && P::INTEGER;
    constexpr bool kSatDefer = MIMC3_SAT_DEFER < 0 ? C::SAT_DEFER : (MIMC3_SAT_DEFER != 0);
    auto evaluate = [&](const uint16_t *ids, int dir, int cnt, int mode) __attribute__((always_inline)) {
        const bool dirty_list = (mode == M_GENERAL);
        // dirty boxes of a null-free chip, planes with a table: the WN body (three sums) instead of the six-sum GENERAL body --
        // except where a cell of the wave's round touches the never-written last row / column (T4: the table does not know it)
        const bool wn_ok = kSat && C::WN && MIMC3_WN && dirty_list && exc_chip == 0;
        // SPARSE: the lane's slice of the null lists lives in registers for the whole call (lists are per point)
        uint32_t ew[4] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu}, ec[2] = {0xffffffffu, 0xffffffffu};
        if constexpr (C::SPARSE) {
            if (sparse_on && cnt > 0) {
                if (dirty_list) {
#pragma unroll
                    for (int k = 0; k < 4; k++) if (l + k * C::LPC < nLw) ew[k] = nl_get(Lw, l + k * C::LPC);
                }
#pragma unroll
                for (int k = 0; k < 2; k++) if (l + k * C::LPC < nLc) ec[k] = nl_get(Lc, l + k * C::LPC);
            }
        }
        // table cells of the one-wave configs park ONE word (sxy): 64 of them fit where 32 six-word slots do, and the f64 finish
        // then runs on all 64 lanes
        constexpr int kFastBatch = (6 * kSumBatch < MIMC3_FAST_BATCH) ? 6 * kSumBatch : MIMC3_FAST_BATCH;
        const bool one_word = kSat && !kAPark && NT == 64 && mode == M_FAST && kFastBatch > kSumBatch;
        const int batch_cells = one_word ? kFastBatch : kSumBatch;
        for (int b0 = 0; b0 < cnt; b0 += batch_cells) {
            const int nb = (cnt - b0) < batch_cells ? (cnt - b0) : batch_cells;
            // thread t finishes cell b0 + t of this batch: its window-side box sums (sum b, sum b^2) come from the table -- four
            // loads issued now, consumed after the evaluation rounds below
            // (the four corners stay in registers until the finish: combining them here would wait for the loads right away)
            [[maybe_unused]] SatT cellQ{}, q00{}, q01{}, q10{}, q11{};
            [[maybe_unused]] int cellZ = 0;
            if constexpr (kSat) {
                const bool need = mode == M_FAST || (C::SPARSE && sparse_on) || (dirty_list && wn_ok);
                if (need && tid < nb) {
                    const uint32_t pk = ids[dir * (b0 + tid)];
                    const SatT *r0 = sat_win + (size_t)(wv0 + (int)((pk >> 8) & 0xffu)) * p.sat_ws + (wu0 + (int)(pk & 0xffu)), *r1 = r0 + (size_t)CW * p.sat_ws;
                    q00 = r0[0]; q01 = r0[CW]; q10 = r1[0]; q11 = r1[CW];
                    if (!kSatDefer) cellQ = q11 - q01 - q10 + q00;
                    if constexpr (kSatZ) {        // the nulls of a dirty box: n of a WN cell; the offset policy's unit conversion (its unmasked sums count a null as 0, not as -k)
                        if (dirty_list) cellZ = (int)sat_box(satz_win, p.sat_ws, wu0 + (int)(pk & 0xffu), wv0 + (int)((pk >> 8) & 0xffu), CW, CW);
                    }
                }
            }
            // Sparse-correction configs: a box is on the dirty list because it meets the BOUNDING BOX of the window's nulls; the
            // table knows whether it holds a null at all.  Boxes without one skip the walk over the window-null list (their
            // corrections to n, sx, sxx are zero): the finishing lanes publish one flag per cell of the batch.
            [[maybe_unused]] unsigned char *nzf = reinterpret_cast<unsigned char *>(&qcnt[24]);       // [kSumBatch]
            constexpr bool kNullFlags = C::SPARSE && kSat && !kSatDefer && kSumBatch <= 32 && MIMC3_NULL_FLAGS;
            // (pays where a box is a small part of the window -- BASELINE C4: 65^2 of 133^2, many dirty-list boxes hold no null, -1.8 % --
            //  and costs a barrier per batch where it is not: C2's 81^2 of 113^2, +1 %)
            const bool flags_on = kNullFlags && 3 * CW * CW <= pt.Dx2 * pt.Dy2;
            if constexpr (kNullFlags) {
                if (dirty_list && sparse_on && flags_on) {
                    if (tid < nb) nzf[tid] = (unsigned char)((kSatZ ? cellZ : P::sat_nulls(cellQ)) != 0);
                    __syncthreads();
                }
            }
            for (int r0 = 0; r0 < nb; r0 += C::CPR * NW) {
                const int slot = r0 + wave * C::CPR + grp;
                const bool on = slot < nb;
                const uint32_t pk = on ? ids[dir * (b0 + slot)] : 0x00000101u;
                const int cx = (int)(pk & 0xffu), cy = (int)((pk >> 8) & 0xffu);
                AccT<Sum> acc;
                bool six = dirty_list;                                // all six sums are cell-specific (else n, sx, sxx are the point's constants)
                bool done = false;
                bool wn_round = false;                                // this wave's cells of the round run the WN body
                if constexpr (C::SPARSE) {
                    // one wave = one cell.  Unless the box touches the never-written last row/column (T4: a whole row of
                    // nulls) the cell is the FAST body plus corrections over the null lists.
                    if (sparse_on && !(dirty_list && !full_win && (cx == pt.csx - 2 || cy == pt.csy - 2))) {
                        AccT<Sum> a0{0, 0, 0, 0, 0, 0};
                        uint32_t cn = 0;
                        Sum csx = 0, csxx = 0, csy = 0, csyy = 0;
                        auto corr_w = [&](const uint32_t (&e)[4]) __attribute__((always_inline)) {
                            uint32_t av[4];
                            bool in[4];
#pragma unroll
                            for (int k = 0; k < 4; k++) {          // window nulls inside the box: their chip pixels leave n, sx, sxx
                                const int ddx = (int)(e[k] & 0xffffu) - cx, ddy = (int)(e[k] >> 16) - cy;
                                in[k] = (unsigned)ddx < (unsigned)CW && (unsigned)ddy < (unsigned)CW;
                                av[k] = P::px_at(CH + (in[k] ? (int)__umul24(ddy, C::CPITCH) + ddx * P::BPP : 0));
                            }
#pragma unroll
                            for (int k = 0; k < 4; k++) {
                                const uint32_t a1 = in[k] ? av[k] : 0u;
                                cn += a1 ? 1u : 0u; csx += a1; csxx += (Sum)__umul24(a1, a1);
                            }
                        };
                        const unsigned char *Wc = W + cy * pt.PW + (pt.sh + cx) * P::BPP;
                        auto corr_c = [&](const uint32_t (&e)[2]) __attribute__((always_inline)) {
                            uint32_t bv[2];
#pragma unroll
                            for (int k = 0; k < 2; k++) {          // chip nulls (row-task rows): the window pixels under them leave sy, syy
                                const bool ok = e[k] != 0xffffffffu;
                                bv[k] = P::px_at(Wc + (ok ? (int)__umul24(e[k] >> 16, pt.PW) + (int)(e[k] & 0xffffu) * P::BPP : 0));
                                bv[k] = ok ? bv[k] : 0u;
                            }
#pragma unroll
                            for (int k = 0; k < 2; k++) { csy += bv[k]; csyy += (Sum)__umul24(bv[k], bv[k]); }
                        };
                        bool wnull = dirty_list;
                        if constexpr (kNullFlags) { if (flags_on) wnull = dirty_list && on && nzf[slot] != 0; }
                        six = wnull;
                        if (wnull) {
                            corr_w(ew);
                            for (int i0 = l + 4 * C::LPC; i0 < nLw; i0 += 4 * C::LPC) {       // long lists: the rest from LDS
                                uint32_t e[4];
#pragma unroll
                                for (int k = 0; k < 4; k++) e[k] = (i0 + k * C::LPC < nLw) ? nl_get(Lw, i0 + k * C::LPC) : 0xffffffffu;
                                corr_w(e);
                            }
                        }
                        if (nLc > 0) {
                            corr_c(ec);
                            for (int i0 = l + 2 * C::LPC; i0 < nLc; i0 += 2 * C::LPC) {
                                uint32_t e[2];
#pragma unroll
                                for (int k = 0; k < 2; k++) e[k] = (i0 + k * C::LPC < nLc) ? nl_get(Lc, i0 + k * C::LPC) : 0xffffffffu;
                                corr_c(e);
                            }
                        }
                        // lane-local, modulo 2^32 / 2^64: the reduced totals are exact.  (The u16 policy's lanes accumulate
                        // 32 bits inside a 64-bit sum, so there the correction is subtracted after the body.)
                        if constexpr (C::FULLTAIL && !kSat) {         // (before the body: its LDS round trip overlaps the row loads)
                            // the window pixels under the pad bytes of this lane's row-end tail dword leave sy, syy again
                            const int X = pt.sh + cx;
                            const uint32_t sft = (uint32_t)((X & (P::G - 1)) * P::BPP);
                            const uint32_t *rp = reinterpret_cast<const uint32_t *>(W + cy * pt.PW + 4 * (X >> P::LOG2G) + (padoff >= 0 ? padoff : 0));
                            const uint32_t pb = padoff >= 0 ? (alignb(rp[1], rp[0], sft) & ~C::LASTFF) : 0u;
                            P::chip_acc(csy, csyy, pb);
                        }
                        constexpr bool kFold = sizeof(Sum) == 4;
                        if (kFold) { a0.sy = (Sum)0 - csy; a0.syy = (Sum)0 - csyy; }
                        // (with a table the body adds nothing to sy, syy: they leave as the corrections alone, the table's box sums
                        //  -- nulls are zeros in them -- are added in the finish)
                        acc = eval_round<C, kSat ? M_XY : M_FAST, false, C::FULLTAIL>(W, pt, cx, cy, l, A, AT, toff, a0, CH);
                        if (!kFold) { acc.sy -= csy; acc.syy -= csyy; }
                        acc.n = 0u - cn; acc.sx = (Sum)0 - csx; acc.sxx = (Sum)0 - csxx;      // + the point's constants, in the finish
                        done = true;
                    } else {
                        six = dirty_list;
                    }
                }
                if (!done) {
                    if (mode == M_FAST) acc = eval_round<C, kSat ? M_XY : M_FAST, !kAPark>(W, pt, cx, cy, l, A, AT, toff, AccT<Sum>{0, 0, 0, 0, 0, 0}, CH);
                    else if (mode == M_CHIPNULL) acc = eval_round<C, M_CHIPNULL, !kAPark>(W, pt, cx, cy, l, A, AT, toff, AccT<Sum>{0, 0, 0, 0, 0, 0}, CH);
                    else {
                        if constexpr (kSat) {
                            const bool t4 = on && !full_win && (cx == pt.csx - 2 || cy == pt.csy - 2);
                            wn_round = wn_ok && __ballot(t4) == 0ull;
                        }
                        if (kSat && wn_round) acc = eval_round<C, kSat ? M_WN : M_GENERAL, !kAPark>(W, pt, cx, cy, l, A, AT, toff, AccT<Sum>{0, 0, 0, 0, 0, 0}, CH);
                        else if (MIMC3_GC && exc_chip == 0) acc = eval_round<C, M_GC, !kAPark>(W, pt, cx, cy, l, A, AT, toff, AccT<Sum>{0, 0, 0, 0, 0, 0}, CH);   // null-free chip: nothing derived from the chip dwords
                        else acc = eval_round<C, M_GENERAL, !kAPark>(W, pt, cx, cy, l, A, AT, toff, AccT<Sum>{0, 0, 0, 0, 0, 0}, CH);
                    }
                }
                Store *sp = sums + 6 * slot;
                if constexpr (kAPark) {
                    // 16-lane row sums (DPP), then the row leaders add into the slot
                    const bool lead = on && (l & 15) == 0;
                    const Sum rsxy = P::template gsum<16>(acc.sxy);
                    if (lead) atomicAdd(&sp[5], P::bits(rsxy));
                    // sy, syy: the body's sums; with a table only the chip-null corrections of a sparse cell (else the table has it all)
                    if (!kSat || (done ? nLc > 0 : (mode != M_FAST && !wn_round))) {
                        const Sum rsy = P::template gsum<16>(acc.sy), rsyy = P::template gsum<16>(acc.syy);
                        if (lead) { atomicAdd(&sp[2], P::bits(rsy)); atomicAdd(&sp[4], P::bits(rsyy)); }
                    }
                    if (six) {
                        const Sum rsx = P::template gsum<16>(acc.sx), rsxx = P::template gsum<16>(acc.sxx);
                        if (lead) { atomicAdd(&sp[1], P::bits(rsx)); atomicAdd(&sp[3], P::bits(rsxx)); }
                        if (!wn_round) {
                            const uint32_t rn = group_sum<16>(acc.n);
                            if (lead) atomicAdd(&sp[0], (Store)rn);
                        }
                    }
                } else {
                    if (kSat && mode == M_FAST) {
                        if (on && l == 0) { if (one_word) sums[slot] = P::bits(acc.sxy); else sp[5] = P::bits(acc.sxy); }   // the other five sums are the point's constants and the table's
                    } else {
                        if (mode != M_GENERAL) { acc.n = NV; acc.sx = SX; acc.sxx = SXX; }
                        if (on && l == 0) {
                            if (kSat && wn_round) { sp[1] = P::bits(acc.sx); sp[3] = P::bits(acc.sxx); sp[5] = P::bits(acc.sxy); }
                            else { sp[0] = (Store)acc.n; sp[1] = P::bits(acc.sx); sp[2] = P::bits(acc.sy); sp[3] = P::bits(acc.sxx); sp[4] = P::bits(acc.syy); sp[5] = P::bits(acc.sxy); }
                        }
                    }
                }
            }
            __syncthreads();
            if (tid < nb) {
                if (kSatDefer) cellQ = q11 - q01 - q10 + q00;
                const uint32_t pk = ids[dir * (b0 + tid)];
                const int cx = (int)(pk & 0xffu), cy = (int)((pk >> 8) & 0xffu);
                Store *sp = sums + 6 * tid;
                Store v[6] = {0, 0, 0, 0, 0, 0};
                if (one_word) v[5] = sums[tid];
                else { v[0] = sp[0]; v[1] = sp[1]; v[2] = sp[2]; v[3] = sp[3]; v[4] = sp[4]; v[5] = sp[5]; }
                // did this cell's round run the WN body?  (the decision was taken per wave and round: no cell of its CPR consecutive
                // slots touches the T4 row / column; the finishing lanes all sit in wave 0)
                bool wn_cell = false;
                if constexpr (kSat) {
                    if (wn_ok) {
                        const bool t4 = !full_win && (cx == pt.csx - 2 || cy == pt.csy - 2);
                        const unsigned long long tm = __ballot(t4);
                        wn_cell = ((tm >> (tid & ~(C::CPR - 1))) & ((1ull << C::CPR) - 1ull)) == 0ull;
                        if (C::SPARSE && sparse_on) wn_cell = false;       // (those cells took the sparse-correction path, or are T4)
                    }
                    if (wn_cell) {
                        Sum ty = 0, tyy = 0;
                        const int z = kSatZ ? cellZ : P::sat_nulls(cellQ);
                        if (p.stats && z == 0) atomicAdd(&p.stats[kStatW * (size_t)blockIdx.x + 15], 1ull);
                        P::sat_win_sums(ty, tyy, cellQ, z, kb, C::NPX, sc_win);
                        v[0] = (Store)(uint32_t)(C::NPX - z); v[2] = P::bits(ty); v[4] = P::bits(tyy);
                    }
                }
                if constexpr (kAPark) {
                    // cells whose n, sx, sxx are the point's constants (minus the corrections that were added above)
                    const bool dense = dirty_list && !(C::SPARSE && sparse_on && !(!full_win && (cx == pt.csx - 2 || cy == pt.csy - 2)));
                    (void)wn_cell;
                    if (!dense) { v[0] += (Store)NV; v[1] += P::bits(SX); v[3] += P::bits(SXX); }
                    if constexpr (kSat) {
                        // window-side sums from the table: every cell but the dense ones and the masked CHIPNULL bodies of the
                        // configs without null lists (those accumulated sy, syy themselves)
                        const bool sparse_cell = C::SPARSE && sparse_on && !(dirty_list && !full_win && (cx == pt.csx - 2 || cy == pt.csy - 2));
                        if (sparse_cell || mode == M_FAST) {
                            Sum ty = 0, tyy = 0;
                            P::sat_win_sums(ty, tyy, cellQ, cellZ, kb, C::NPX, sc_win);
                            v[2] += P::bits(ty); v[4] += P::bits(tyy);
                        }
                    }
                    if (C::SPARSE) v[0] = (Store)(uint32_t)v[0];          // n travels as a 32-bit count (corrections wrap modulo 2^32)
#pragma unroll
                    for (int k = 0; k < 6; k++) sp[k] = 0;                 // the slot is empty for the next batch
                }
                if constexpr (kSat && !kAPark) {
                    if (mode == M_FAST) {
                        Sum ty = 0, tyy = 0;
                        P::sat_win_sums(ty, tyy, cellQ, 0, kb, C::NPX, sc_win);
                        v[0] = (Store)NV; v[1] = P::bits(SX); v[2] = P::bits(ty); v[3] = P::bits(SXX); v[4] = P::bits(tyy);
                    }
                }
                store_ncc(cx, cy, P::ncc(v, sc_chip, sc_win, ka, kb));
            }
            __syncthreads();
        }
    };
    // ---- driver loop: [evaluate everything queued] -> [speculative climb step | exact state machine] ----
    // Speculative parallel climb (stages 0..kSpecRounds-1): lane k follows pivot k's hill climb on the
    // cached values, ignoring the visited state (which can only END a real climb earlier), so that the
    // cells the sequential state machine will ask for are evaluated in a few bulk batches.  Purely a
    // prefetch: it never touches `vis`; the result is decided by the exact replay (stage kSpecRounds).
    constexpr int kSpecRounds = 16;
    const int kLookahead = p.lookahead;
    int stage = 0;
    int su = 0, sv = 0;
    bool alive = false;
    float smax = -2.0f;
    if (wave == 0 && lane < npiv) {
        su = pivs[2 * lane] + pt.dx2; sv = pivs[2 * lane + 1] + pt.dy2;
        alive = inside(su, sv);
    }
    const int start_u = su, start_v = sv;            // lane k keeps pivot k's start for the replay (k < 64)
    // lane k's speculative trajectory as 4-bit codes per scan: 0 = no scan, 1..9 = the scan updated the
    // running maximum at 3x3 index code-1 (5 = centre: no move), 10 = scanned without update
    unsigned long long traj = 0ull;
    int nsc = 0;                                     // scans recorded so far
    bool replay_generic = false;
    bool cut = false;                                // this lane's speculation was stopped before its climb ended
    // many-pivot configurations: pivots lane+64, lane+128, ... -- bit j of xdone: pivot lane + 64 (j + 1) is recorded
    uint32_t xdone = 0u;
    bool xpending = false;
    unsigned long long *trajL = reinterpret_cast<unsigned long long *>(smem + (C::MANYP ? p.lds_off_traj : 0));   // [npiv] recorded climbs
    unsigned char *tinfo = reinterpret_cast<unsigned char *>(trajL + npiv);      // [npiv] scans recorded | still climbing << 7
    unsigned char *Tl = tinfo + npiv;                                            // [npiv] scans really performed (exact replay)
    (void)xdone; (void)xpending; (void)trajL; (void)tinfo; (void)Tl;
    // wave-uniform state of the reference's loops (:691-753)
    int k = 0, pu = 0, pvv = 0, du = 0, dv = 0, newncc = 0;
    bool fresh = true;
    float nccmax = -2.0f, best = -2.0f;
    int peak_u = pt.dx2, peak_v = pt.dy2;

    for (int guard = 0; guard <= pt.ncell + 16; guard++) {
        if (qcnt[3]) {                                   // NCC cache overflow: let the general kernel redo this point
            if (tid == 0) p.ovf_list[atomicAdd(p.ovf_count, 1)] = gidx;
            return;
        }
        {   // evaluate everything queued, then empty the queue (the only call site of `evaluate`)
            const int nA = qcnt[0] & 0xffff, nB = (int)((uint32_t)qcnt[0] >> 16);
            if (p.stats && tid == 0) {
                p.stats[kStatW * (size_t)blockIdx.x + 8] += nA; p.stats[kStatW * (size_t)blockIdx.x + 9] += nB;
                p.stats[kStatW * (size_t)blockIdx.x + 10] += (nA + nB) ? 1 : 0;
            }
            __syncthreads();
            if (tid == 0) qcnt[0] = 0;
            evaluate(list, 1, nA, clean_mode);
            evaluate(list + lcap - 1, -1, nB, M_GENERAL);
            __syncthreads();
        }
        if (p.debug_stop == 4) return;
        MIMC3_STAMP(3)
        // the sequential part runs on wave 0 only (lane k <-> pivot k); the other waves of the workgroup
        // wait for its decision: 0 = cells were queued, evaluate and come back; 1 = the climb is finished
        auto step = [&]() __attribute__((always_inline)) -> int {
        if (stage < kSpecRounds) {
            // each alive lane keeps scanning while its whole 3x3 is cached (values evaluated for other
            // pivots count too); when it runs into unknown cells it queues them, plus the 3x3 one step
            // further in the direction it just moved (lookahead: straight climbs advance 2 scans per batch)
            int ldu = 0, ldv = 0;
            while (alive && !cut && nsc < kSpecRounds) {
                float sm = smax;
                const int mv = scan9(su, sv, sm);
                if (mv == -2) break;                                 // wait for the batch that holds the missing cells
                smax = sm;
                const bool moved = (mv >= 0 && mv != 4);
                ldu = moved ? mv / 3 - 1 : 0; ldv = moved ? mv % 3 - 1 : 0;
                su += ldu; sv += ldv;
                traj |= (unsigned long long)(mv >= 0 ? mv + 1 : 10) << (4 * nsc);
                nsc++;
                alive = moved && inside(su, sv);
            }
            if (alive && !cut && nsc < kSpecRounds) {
                request9(su - OCW, sv - OCW);
                for (int la = 1; la <= kLookahead; la++)              // straight-line lookahead along the last move
                    if ((ldu | ldv) != 0 && inside(su + la * ldu, sv + la * ldv))
                        request9(su + la * ldu - OCW, sv + la * ldv - OCW);
            }
            // Pivots beyond the first 64 (the 21x21 set of the control-point stage) have no lane of their own: each lane walks
            // pivots lane+64, lane+128, ... from their starts over the cached values.  A walk that runs into cells it cannot
            // read queues them and is repeated in the next stage; one that gets through is recorded in LDS (trajectory, scan
            // count, still-climbing flag) for the exact replay and never walked again.
            if constexpr (C::MANYP) {
                xpending = false;
                int j = 0;
                for (int k2 = lane + 64; k2 < npiv; k2 += 64, j++) {
                    if (j < 32 && ((xdone >> j) & 1u)) continue;
                    int eu = pivs[2 * k2] + pt.dx2, ev = pivs[2 * k2 + 1] + pt.dy2;
                    bool go = inside(eu, ev), stalled = false;
                    float em = -2.0f;
                    unsigned long long tr = 0ull;
                    int n = 0;
                    while (go && n < kSpecRounds) {
                        float sm = em;
                        const int mv = scan9(eu, ev, sm);
                        if (mv == -2) { request9(eu - OCW, ev - OCW); stalled = true; break; }
                        em = sm;
                        const bool moved = (mv >= 0 && mv != 4);
                        if (moved) { eu += mv / 3 - 1; ev += mv % 3 - 1; }
                        tr |= (unsigned long long)(mv >= 0 ? mv + 1 : 10) << (4 * n);
                        n++;
                        go = moved && inside(eu, ev);
                    }
                    if (stalled) { xpending = true; continue; }
                    trajL[k2] = tr;
                    tinfo[k2] = (unsigned char)(n | (go ? 0x80 : 0));
                    if (j < 32) xdone |= 1u << j;
                }
            }
            stage++;
            MIMC3_STAMP(4)
            const bool more = __any((alive && !cut && nsc < kSpecRounds) || xpending) && stage < kSpecRounds;
            if (more && qcnt[0] != 0) return 0;                      // evaluate the queued cells, then scan on
            // A lane that is still alive here had its speculation CUT (kSpecRounds scans, or the edge of the cache band).  Real
            // climbs are usually much shorter than speculative ones (the visited state ends them), so the exact replay runs
            // on the recorded prefixes first; only if a cut pivot really consumes its whole prefix the generic loops take over.
            replay_generic = C::MANYP ? (__any(xpending) != 0) : (npiv > 64);   // (plain configurations record the first 64 pivots only)
            stage = kSpecRounds;
            if (qcnt[0] != 0) return 0;                             // evaluate what was queued first
        }
        if (p.debug_stop == 7) return 1;                             // (profiling: everything but the replay and the fit)
        if (!replay_generic) {
            // ---- exact replay from the recorded trajectories.  Every scan's move and running maximum
            //      depend only on NCC values (identical to the reference's compare sequence, :736-741);
            //      the visited state only decides HOW MANY scans of a pivot really happen (newncc != 0,
            //      :699) -- that part is sequential over pivots and is all that is done here.
            MIMC3_STAMP(5)
            int T = 0;                                   // lane k: number of scans pivot k really performs
            const bool regmask = (pt.csx <= 64 && pt.csy <= 64);
            // visited set in REGISTERS when it fits: lane r holds the 64-bit column mask of compact row r;
            // a scan's 3x3 = 3 rows x 3 adjacent bits, tested/set with readlane + scalar ops (no LDS
            // round trip in this sequential chain).  Otherwise the LDS bit array is used.
            uint32_t vlo = 0, vhi = 0;
            const int c1 = lane / 3 - 1, c2 = lane % 3 - 1;
            unsigned long long ntr = 0ull;           // many-pivot configurations: the record of pivot kk + 1 (>= 64), read one pivot ahead
            int nqu = 0, nqv = 0, ninfo = 0;
            bool xcut = false;
            // Register visited set, one lane per pivot: every lane first turns ITS trajectory into move deltas (4 bits per scan) -- in
            // parallel, where the sequential loop used to decode the 4-bit result codes scan by scan with a table on the scalar unit;
            // what stays sequential is three lane reads per pivot and, per scan, the 3 x 3-bit test-and-set on the three row lanes,
            // the continue test and the step (the chain was 19 k of a point's 126 k cycles at BASELINE C2; 12 k now).
            constexpr bool kFastReplay = MIMC3_FAST_REPLAY && !C::MANYP;
            // big chips: cell grids of up to 96 x 128 (BASELINE C4: 70 x 70) keep the visited set in registers too -- three column
            // words per row, lane r holding rows r and r + 64 -- instead of LDS bit words read and OR-ed inside the chain
            // (only in the compact LDS form, which is what large windows run: in the regular big-chip kernels the six extra registers
            //  cost 2-3 % at BASELINE C2's geometry, where every grid fits 64 x 64 -- u8 ocw 30 10.92 vs 10.66 ms, d/dx ocw 40 19.33 vs
            //  18.76; BASELINE C4: 161.5 -> 158.6 ms)
            constexpr bool kWideReplay = kFastReplay && MIMC3_WIDE_REPLAY && C::COMPACT;
            const bool regmask2 = kWideReplay && !regmask && pt.csx <= 96 && pt.csy <= 128;
            uint32_t va2 = 0u, vb0 = 0u, vb1 = 0u, vb2 = 0u;
            (void)va2; (void)vb0; (void)vb1; (void)vb2;
            if (kFastReplay && (regmask || regmask2)) {
                if (p.stats && tid == 0) p.stats[kStatW * (size_t)blockIdx.x + 14] += 1;
                // lane k: pivot k's moves as 4 bits per scan, (du + 1) | (dv + 1) << 2 -- 5 = the scan did not move
                uint32_t dlo = 0u, dhi = 0u;
#pragma unroll
                for (int t = 0; t < kSpecRounds; t++) {
                    if (__ballot(nsc > t) == 0ull) break;
                    const uint32_t code = (uint32_t)(traj >> (4 * t)) & 15u;
                    const bool moved = (code - 1u) < 9u && code != 5u;
                    const uint32_t mv = code - 1u, q3 = (mv * 11u) >> 5;
                    const uint32_t d = moved ? (q3 | ((mv - 3u * q3) << 2)) : 5u;
                    if (t < 8) dlo |= d << (4 * (t & 7)); else dhi |= d << (4 * (t & 7));
                }
                const uint32_t head = (uint32_t)(start_u - OCW) | ((uint32_t)(start_v - OCW) << 8) | ((uint32_t)nsc << 16);
                for (int kk = 0; kk < npiv; kk++) {
                    const uint32_t hd = (uint32_t)__builtin_amdgcn_readlane((int)head, kk);
                    const int n_k = (int)(hd >> 16);
                    int ccx = (int)(hd & 0xffu), cy1 = (int)((hd >> 8) & 0xffu) - 1;      // centre column, centre row - 1
                    unsigned long long cur = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)dhi, kk) << 32) |
                                             (uint32_t)__builtin_amdgcn_readlane((int)dlo, kk);
                    int t = 0;
                    while (t < n_k) {
                        // rows ccy - 1 .. ccy + 1 (lanes ccy - 1 .. ccy + 1) test and set their three bits; fresh3 = the lanes that held an
                        // unvisited one.  Written out: the compiler keeps the wave-uniform values of this chain in vector registers
                        // and rebuilds every mask as 0 / 1 values (33 instructions per scan instead of these 10).
                        unsigned long long fresh3, sv;
                        uint32_t tmp;
                        if (kWideReplay && regmask2) {
                            const int sh = ccx - 1;
                            const unsigned long long w01 = sh < 64 ? (7ull << (sh & 63)) : 0ull, w12 = sh >= 32 ? (7ull << ((sh - 32) & 63)) : 0ull;
                            const uint32_t m0 = (uint32_t)w01, m1 = sh < 32 ? (uint32_t)(w01 >> 32) : (uint32_t)w12, m2 = (uint32_t)(w12 >> 32);
                            const int lane64 = lane + 64;
                            unsigned long long frA;
                            asm volatile("v_subrev_u32_e32 %[tmp], %[y], %[l1]\n\t"
                                         "v_cmp_gt_u32_e32 vcc, 3, %[tmp]\n\t"
                                         "s_and_saveexec_b64 %[sv], vcc\n\t"
                                         "v_bitop3_b32 %[tmp], %[m0], %[a0], %[m0] bitop3:0x30\n\t"
                                         "v_bitop3_b32 %[tmp], %[m1], %[tmp], %[a1] bitop3:0xdc\n\t"
                                         "v_bitop3_b32 %[tmp], %[m2], %[tmp], %[a2] bitop3:0xdc\n\t"
                                         "v_cmp_ne_u32_e32 vcc, 0, %[tmp]\n\t"
                                         "v_or_b32_e32 %[a0], %[m0], %[a0]\n\t"
                                         "v_or_b32_e32 %[a1], %[m1], %[a1]\n\t"
                                         "v_or_b32_e32 %[a2], %[m2], %[a2]\n\t"
                                         "s_mov_b64 exec, %[sv]\n\t"
                                         "s_mov_b64 %[fa], vcc\n\t"
                                         "v_subrev_u32_e32 %[tmp], %[y], %[l65]\n\t"
                                         "v_cmp_gt_u32_e32 vcc, 3, %[tmp]\n\t"
                                         "s_and_saveexec_b64 %[sv], vcc\n\t"
                                         "v_bitop3_b32 %[tmp], %[m0], %[b0], %[m0] bitop3:0x30\n\t"
                                         "v_bitop3_b32 %[tmp], %[m1], %[tmp], %[b1] bitop3:0xdc\n\t"
                                         "v_bitop3_b32 %[tmp], %[m2], %[tmp], %[b2] bitop3:0xdc\n\t"
                                         "v_cmp_ne_u32_e32 vcc, 0, %[tmp]\n\t"
                                         "v_or_b32_e32 %[b0], %[m0], %[b0]\n\t"
                                         "v_or_b32_e32 %[b1], %[m1], %[b1]\n\t"
                                         "v_or_b32_e32 %[b2], %[m2], %[b2]\n\t"
                                         "s_mov_b64 exec, %[sv]\n\t"
                                         "s_or_b64 %[fr], %[fa], vcc"
                                         : [tmp] "=&v"(tmp), [sv] "=&s"(sv), [fa] "=&s"(frA), [fr] "=s"(fresh3), [a0] "+v"(vlo), [a1] "+v"(vhi), [a2] "+v"(va2),
                                           [b0] "+v"(vb0), [b1] "+v"(vb1), [b2] "+v"(vb2)
                                         : [y] "s"(cy1), [l1] "v"(lane), [l65] "v"(lane64), [m0] "s"(m0), [m1] "s"(m1), [m2] "s"(m2)
                                         : "vcc", "scc");
                        } else {
                        const unsigned long long m3 = 7ull << (ccx - 1);
                        const uint32_t m3lo = (uint32_t)m3, m3hi = (uint32_t)(m3 >> 32);
                        asm volatile("v_subrev_u32_e32 %[tmp], %[y], %[l1]\n\t"
                                     "v_cmp_gt_u32_e32 vcc, 3, %[tmp]\n\t"
                                     "s_and_saveexec_b64 %[sv], vcc\n\t"
                                     "v_bitop3_b32 %[tmp], %[mlo], %[vlo], %[mlo] bitop3:0x30\n\t"      // m3lo & ~vlo
                                     "v_bitop3_b32 %[tmp], %[mhi], %[tmp], %[vhi] bitop3:0xdc\n\t"      // | (m3hi & ~vhi)
                                     "v_cmp_ne_u32_e32 vcc, 0, %[tmp]\n\t"
                                     "v_or_b32_e32 %[vlo], %[mlo], %[vlo]\n\t"
                                     "v_or_b32_e32 %[vhi], %[mhi], %[vhi]\n\t"
                                     "s_mov_b64 exec, %[sv]\n\t"
                                     "s_mov_b64 %[fr], vcc"
                                     : [tmp] "=&v"(tmp), [sv] "=&s"(sv), [fr] "=s"(fresh3), [vlo] "+v"(vlo), [vhi] "+v"(vhi)
                                     : [y] "s"(cy1), [l1] "v"(lane), [mlo] "s"(m3lo), [mhi] "s"(m3hi)
                                     : "vcc", "scc");
                        }
                        t++;
                        const uint32_t d = (uint32_t)cur & 15u;
                        cur >>= 4;
                        ccx += (int)(d & 3u) - 1; cy1 += (int)(d >> 2) - 1;          // (d = 5, no move: + 0)
                        if (fresh3 == 0ull) break;
                        if (d == 5u) break;
                    }
                    T = (lane == kk) ? t : T;
                }
            } else
            for (int kk = 0; kk < npiv; kk++) {
                unsigned long long tr;
                int qu, qv, info = 0;
                if (!C::MANYP || kk < 64) {
                    tr = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(traj >> 32), kk) << 32) |
                         (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)traj, kk);
                    qu = __builtin_amdgcn_readlane(start_u, kk); qv = __builtin_amdgcn_readlane(start_v, kk);
                } else {
                    tr = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(ntr >> 32)) << 32) |
                         (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)ntr);
                    qu = __builtin_amdgcn_readfirstlane(nqu); qv = __builtin_amdgcn_readfirstlane(nqv);
                    info = __builtin_amdgcn_readfirstlane(ninfo);
                }
                if (C::MANYP && kk + 1 >= 64 && kk + 1 < npiv) {
                    ntr = trajL[kk + 1]; ninfo = tinfo[kk + 1];
                    nqu = pivs[2 * (kk + 1)] + pt.dx2; nqv = pivs[2 * (kk + 1) + 1] + pt.dy2;
                }
                bool cont = true;
                int Tk = 0;
                for (int t = 0; t < kSpecRounds; t++) {
                    const int code = (int)((tr >> (4 * t)) & 15ull);
                    if (code == 0 || !cont) break;
                    bool unv;                        // (kept as a lane-mask test: no 0/1 value goes through a vector register)
                    if (regmask) {
                        // the three rows ccy-1..ccy+1 test and set their own 3 bits in parallel
                        const int ccx = qu - OCW, ccy = qv - OCW;
                        const unsigned long long m3 = 7ull << (ccx - 1);
                        const uint32_t m3lo = (uint32_t)m3, m3hi = (uint32_t)(m3 >> 32);
                        const bool mine = (unsigned)(lane - (ccy - 1)) < 3u;
                        unv = __ballot(mine && (((~vlo & m3lo) | (~vhi & m3hi)) != 0u)) != 0ull;
                        vlo = mine ? (vlo | m3lo) : vlo;
                        vhi = mine ? (vhi | m3hi) : vhi;
                    } else {
                        const int vb = (lane < 9) ? (qv + c2 - OCW) * vpitch + (qu + c1 - OCW) : 0;
                        const bool unvis = (lane < 9) && (((vis[vb >> 5] >> (vb & 31)) & 1u) == 0u);
                        unv = __ballot(unvis) != 0ull;
                        if (unvis) atomicOr(&vis[vb >> 5], 1u << (vb & 31));
                    }
                    Tk = t + 1;
                    const bool moved = (code <= 9) && (code != 5);
                    if (moved) {                     // 3x3 index code-1 -> (column, row) step; the quotient by 3 from a 2-bit table (scalar)
                        const int mv = code - 1;
                        const int q3 = (int)((0x2A540u >> (2 * mv)) & 3u);
                        qu += q3 - 1; qv += (mv - 3 * q3) - 1;
                    }
                    cont = moved && unv;
                }
                if (!C::MANYP || kk < 64) {
                    if (lane == kk) T = Tk;
                } else {
                    if (lane == 0) Tl[kk] = (unsigned char)Tk;
                    xcut = xcut || ((info & 0x80) != 0 && Tk == (info & 0x7f));
                }
            }
            if (__any(alive && T == nsc) || xcut) {          // a cut pivot really went through its whole recorded prefix: the generic loops decide
                replay_generic = true;
                if (!regmask)                        // the large-grid form marked its scans in LDS: start over from a clean visited set
                    for (int i = lane; i < ((pt.csy * vpitch) >> 5); i += 64) vis[i] = 0u;
            } else {
            if (regmask && lane < pt.csy) {   // publish the visited rows for the fit (csx <= 64: vpitch is 32 or 64)
                vis[(lane * vpitch) >> 5] = vlo;
                if (vpitch > 32) vis[((lane * vpitch) >> 5) + 1] = vhi;
            }
            if (kWideReplay && regmask2) {        // (up to three words per row; rows lane and lane + 64)
                if (lane < pt.csy) { uint32_t *r = &vis[(lane * vpitch) >> 5]; r[0] = vlo; if (vpitch > 32) r[1] = vhi; if (vpitch > 64) r[2] = va2; }
                if (lane + 64 < pt.csy) { uint32_t *r = &vis[((lane + 64) * vpitch) >> 5]; r[0] = vb0; if (vpitch > 32) r[1] = vb1; if (vpitch > 64) r[2] = vb2; }
            }
            MIMC3_STAMP(7)
            // lane k: where pivot k ended and with which maximum (:744-752): after the last updating scan
            // the pivot sits on the arg-max cell, so the running maximum is that cell's NCC
            int fu = start_u, fv = start_v;
            bool upd = false;
            for (int t = 0; t < T; t++) {
                const int code = (int)((traj >> (4 * t)) & 15ull);
                if (code <= 9) { const int mv = code - 1; const int q3 = (mv * 11) >> 5; fu += q3 - 1; fv += (mv - 3 * q3) - 1; upd = true; }
            }
            uint32_t fpos = ((uint32_t)fv << 16) | (uint32_t)fu;
            const float fmax = upd ? lookup(fu - OCW, fv - OCW) : -2.0f;
            float bv = (lane < npiv) ? fmax : -__builtin_inff();
            int bi = lane;
            if constexpr (C::MANYP) {            // the lane's other pivots, in pivot order: a later one must be strictly better (:747)
                for (int k2 = lane + 64; k2 < npiv; k2 += 64) {
                    const unsigned long long tr2 = trajL[k2];
                    const int T2 = Tl[k2];
                    int gu = pivs[2 * k2] + pt.dx2, gv = pivs[2 * k2 + 1] + pt.dy2;
                    bool upd2 = false;
                    for (int t = 0; t < T2; t++) {
                        const int code = (int)((tr2 >> (4 * t)) & 15ull);
                        if (code <= 9) { const int mv = code - 1; const int q3 = (mv * 11) >> 5; gu += q3 - 1; gv += (mv - 3 * q3) - 1; upd2 = true; }
                    }
                    const float g = upd2 ? lookup(gu - OCW, gv - OCW) : -2.0f;
                    if (g > bv) { bv = g; bi = k2; fpos = ((uint32_t)gv << 16) | (uint32_t)gu; }
                }
            }
            argmax_row16(bv, bi);
#pragma unroll
            for (int o = 16; o <= 32; o <<= 1) {
                const float ov = __shfl_xor(bv, o, 64);
                const int oi = __shfl_xor(bi, o, 64);
                if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
            }
            bv = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(bv)));
            bi = __builtin_amdgcn_readfirstlane(bi);
            if (bv > -2.0f) {                            // strict >, first pivot attaining the maximum wins
                const uint32_t pk = (uint32_t)__builtin_amdgcn_readlane((int)fpos, bi & 63);   // pivot bi lives on lane bi mod 64
                peak_u = (int)(pk & 0xffffu); peak_v = (int)(pk >> 16); best = bv;
            }
            return 1;
            }
        }
        // ---- exact hill climb (resumable), generic form: the reference's sequential loops ----------
        bool finished = false;
        for (;;) {
            if (fresh) {
                if (k >= npiv) { finished = true; break; }
                if (k < 64) { pu = __builtin_amdgcn_readlane(start_u, k); pvv = __builtin_amdgcn_readlane(start_v, k); }
                else { pu = pivs[2 * k] + pt.dx2; pvv = pivs[2 * k + 1] + pt.dy2; }
                nccmax = -2.0f; du = -1; dv = -1; newncc = 1; fresh = false;
            }
            bool end_pivot = !((du != 0 || dv != 0) && newncc != 0);
            if (!end_pivot && !inside(pu, pvv)) end_pivot = true;            // boundary break (:703-707)
            if (end_pivot) {
                if (nccmax > best) { peak_u = pu; peak_v = pvv; best = nccmax; }   // :747-752
                k++; fresh = true;
                continue;
            }
            const bool act = lane < 9;
            const int c1 = lane / 3 - 1, c2 = lane % 3 - 1;
            const int cx = pu + c1 - OCW, cy = pvv + c2 - OCW;
            const float v = act ? lookup(cx, cy) : kUnknown;
            const int vb = act ? cy * vpitch + cx : 0;
            const bool unvis = act && (((vis[vb >> 5] >> (vb & 31)) & 1u) == 0u);
            const bool missing = unvis && (v == kUnknown || v == kWanted);
            if (__ballot(missing)) {
                if (missing) request(cx, cy);
                break;
            }
            newncc = __popcll(__ballot(unvis));
            if (unvis) atomicOr(&vis[vb >> 5], 1u << (vb & 31));
            float bv = (act && v == v) ? v : -__builtin_inff();
            int bi = lane;
            argmax_row16(bv, bi);                       // lanes 0..15 now hold (max, first index attaining it)
            bv = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(bv)));
            bi = __builtin_amdgcn_readfirstlane(bi);
            du = 0; dv = 0;
            if (bv > nccmax) { nccmax = bv; du = bi / 3 - 1; dv = bi % 3 - 1; }
            pu += du; pvv += dv;
        }
        return finished ? 1 : 0;
            };
        if (wave == 0) {
            const int decision = step();
            if (lane == 0) qcnt[9] = decision;
        }
        __syncthreads();
        if (qcnt[9]) break;
    }
    if (qcnt[3]) {                                       // the last climb step left the cache band
        if (tid == 0) p.ovf_list[atomicAdd(p.ovf_count, 1)] = gidx;
        return;
    }
    MIMC3_STAMP(3)
    if (p.debug_stop == 6 || p.debug_stop == 7) { if (tid == 0) p.out[3 * (size_t)gidx] = best + (float)peak_u + (float)peak_v; return; }
    // ---- 3x3 quadratic fit (:757-788) ----------------------------------------------------------
    // The reference's arithmetic, value by value (f32 linear combinations of the nine NCC values, widened, divided by 36 in
    // f64; the two offsets as f32 expressions divided by the determinant in f64) -- but the five divisions by 36 run as ONE
    // division on lanes 0..4 and the two by the determinant as one on lanes 0..1: an f64 division is ~30 instructions, and
    // one lane doing seven of them in a row was 3 % of the kernel's vector instructions.
    if (tid < 5) {
        float n9[9];
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const int vb = (peak_v - 1 + r - OCW) * vpitch + (peak_u - 1 + c - OCW);
                n9[3 * r + c] = ((vis[vb >> 5] >> (vb & 31)) & 1u) ? lookup(peak_u - 1 + c - OCW, peak_v - 1 + r - OCW) : -2.0f;   // visited = scanned = inside the band
            }
        const float e0 = 6 * n9[0] - 12 * n9[1] + 6 * n9[2] + 6 * n9[3] - 12 * n9[4] + 6 * n9[5] + 6 * n9[6] - 12 * n9[7] + 6 * n9[8];
        const float e1 = 9 * n9[0] - 9 * n9[2] - 9 * n9[6] + 9 * n9[8];
        const float e2 = 6 * n9[0] + 6 * n9[1] + 6 * n9[2] - 12 * n9[3] - 12 * n9[4] - 12 * n9[5] + 6 * n9[6] + 6 * n9[7] + 6 * n9[8];
        const float e3 = -6 * n9[0] + 6 * n9[2] - 6 * n9[3] + 6 * n9[5] - 6 * n9[6] + 6 * n9[8];
        const float e4 = -6 * n9[0] - 6 * n9[1] - 6 * n9[2] + 6 * n9[6] + 6 * n9[7] + 6 * n9[8];
        double cp = (double)(tid == 0 ? e0 : (tid == 1 ? e1 : (tid == 2 ? e2 : (tid == 3 ? e3 : e4))));
        cp /= 36;
        auto from_lane = [&](int ln) __attribute__((always_inline)) -> double {
            const long long bits = __double_as_longlong(cp);
            const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)bits, ln), hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(bits >> 32), ln);
            return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
        };
        const double cp0 = from_lane(0), cp1 = from_lane(1), cp2 = from_lane(2), cp3 = from_lane(3), cp4 = from_lane(4);
        const float num = tid == 0 ? (float)(-2 * cp2 * cp3 + cp1 * cp4) : (float)(-2 * cp0 * cp4 + cp1 * cp3);
        const double det = 4 * cp0 * cp2 - cp1 * cp1;
        float o = (float)((double)num / det);
        o += (float)(tid == 0 ? peak_u - pt.dx2 : peak_v - pt.dy2);
        if (tid < 2) p.out[3 * (size_t)gidx + tid] = o;
        if (tid == 0) p.out[3 * (size_t)gidx + 2] = best;
    }
    MIMC3_STAMP(3)
}

// ---- f32 image -> zero-bordered u8 plane, proving the image is 8-bit integral ---------------------
__global__ void prep_u8_plane(const float *img, int H, int W, unsigned char *plane, int Wp, int pad, int *not_u8)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= W || y >= H) return;
    const float v = img[(size_t)y * W + x];
    const float r = truncf(v);
    const bool ok = (v >= 0.0f) && (v <= 255.0f) && (r == v);   // NaN fails every comparison
    if (!ok) *not_u8 = 1;
    plane[(size_t)(y + pad) * Wp + (x + pad)] = ok ? (unsigned char)r : (unsigned char)0;
}

// ---- raw 8-bit DN -> f32 image (what GMA_float_load_tiff makes on the host, GMA.c:288-298) + zero-bordered u8 plane,
//      four pixels per thread.  The pair is 8-bit by construction: nothing to prove.
__global__ void widen_u8_plane(const unsigned char *__restrict__ raw, int H, int W, float *__restrict__ img,
                               unsigned char *__restrict__ plane, int Wp, int pad)
{
    const int x = 4 * (blockIdx.x * blockDim.x + threadIdx.x), y = blockIdx.y;
    if (x >= W || y >= H) return;
    const size_t i = (size_t)y * W + x;
    unsigned char *pl = plane + (size_t)(y + pad) * Wp + (x + pad);
    if (x + 3 < W && (i & 3) == 0) {                     // pad and Wp are multiples of 4: the plane dword is aligned when x is
        const uint32_t v = *reinterpret_cast<const uint32_t *>(raw + i);
        *reinterpret_cast<float4 *>(img + i) = make_float4((float)(v & 0xffu), (float)((v >> 8) & 0xffu), (float)((v >> 16) & 0xffu), (float)(v >> 24));
        *reinterpret_cast<uint32_t *>(pl) = v;
    } else {
        for (int k = 0; k < 4 && x + k < W; k++) { img[i + k] = (float)raw[i + k]; pl[k] = raw[i + k]; }
    }
}
// ---- raw 16-bit DN -> f32 image (GMA.c:299-309)
__global__ void widen_u16_image(const unsigned short *__restrict__ raw, size_t n, float *__restrict__ img)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) img[i] = (float)raw[i];
}
hipError_t launch_widen_u8(const unsigned char *raw, int H, int W, float *img, unsigned char *plane, int Wp, int pad, hipStream_t s)
{
    dim3 blk(256), grd((W + 1023) / 1024, H);
    hipLaunchKernelGGL(widen_u8_plane, grd, blk, 0, s, raw, H, W, img, plane, Wp, pad);
    return hipGetLastError();
}
hipError_t launch_widen_u16(const unsigned short *raw, size_t n, float *img, hipStream_t s)
{
    hipLaunchKernelGGL(widen_u16_image, dim3(4096), dim3(256), 0, s, raw, n, img);
    return hipGetLastError();
}

// ---- f32 image -> zero-bordered u16 plane of q = value * 2^s (PxU16 policy) --------------------------
// flags bit0: some pixel is not an integer in [0,4095]; bit1: some pixel*8 is not an integer in [0,4095]
__global__ void detect_scaled_int(const float *img, size_t n, int *flags)
{
    int f = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float v = img[i], v8 = v * 8.0f;
        if (!(v >= 0.0f && v <= 4095.0f && truncf(v) == v)) f |= 1;
        if (!(v8 >= 0.0f && v8 <= 4095.0f && truncf(v8) == v8)) f |= 2;
    }
    if (f) atomicOr(flags, f);
}
__global__ void prep_u16_plane(const float *img, int H, int W, unsigned short *plane, int Wp, int pad, float mul)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= W || y >= H) return;
    plane[(size_t)(y + pad) * Wp + (x + pad)] = (unsigned short)(img[(size_t)y * W + x] * mul);
}
hipError_t launch_detect_scaled_int(const float *img, size_t n, int *d_flags, hipStream_t s)
{
    hipLaunchKernelGGL(detect_scaled_int, dim3(2048), dim3(256), 0, s, img, n, d_flags);
    return hipGetLastError();
}
hipError_t launch_prep_u16(const float *img, int H, int W, unsigned short *plane, int Wp, int pad, int shift, hipStream_t s)
{
    dim3 blk(256), grd((W + 255) / 256, H);
    hipLaunchKernelGGL(prep_u16_plane, grd, blk, 0, s, img, H, W, plane, Wp, pad, (float)(1 << shift));
    return hipGetLastError();
}

// ---- f32 image -> zero-bordered f32 plane (PxF32 policy) ---------------------------------------------
__global__ void prep_f32_plane(const float *img, int H, int W, float *plane, int Wp, int pad)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= W || y >= H) return;
    plane[(size_t)(y + pad) * Wp + (x + pad)] = img[(size_t)y * W + x];
}

hipError_t launch_prep_f32(const float *img, int H, int W, float *plane, int Wp, int pad, hipStream_t s)
{
    dim3 blk(256), grd((W + 255) / 256, H);
    hipLaunchKernelGGL(prep_f32_plane, grd, blk, 0, s, img, H, W, plane, Wp, pad);
    return hipGetLastError();
}

hipError_t launch_prep_u8(const float *img, int H, int W, unsigned char *plane, int Wp, int pad, int *d_flag, hipStream_t s)
{
    dim3 blk(256), grd((W + 255) / 256, H);
    hipLaunchKernelGGL(prep_u8_plane, grd, blk, 0, s, img, H, W, plane, Wp, pad, d_flag);
    return hipGetLastError();
}

// ---- launcher -------------------------------------------------------------------------------------
// LDS carve of one configuration for the launch's largest window; returns the bytes (fills `a` when given)
template <class C>
static size_t px_layout(MatchU8Args *a, int max_abs_u, int max_abs_v, int max_npiv)
{
    MatchU8Args tmp{};
    MatchU8Args &r = a ? *a : tmp;
    const bool full = r.win_half > 0;                       // full-square search area: no empty last row / column
    const int Dx2 = full ? 2 * r.win_half + 1 : 2 * (max_abs_u + C::OCW + 2) + 1, Dy2 = full ? 2 * r.win_half + 1 : 2 * (max_abs_v + C::OCW + 2) + 1;
    const int cells = (Dx2 - 2 * C::OCW + 1) * (Dy2 - 2 * C::OCW + 1);
    // pitch (dwords): written dwords + one zero dword, and the right-most cell's sliding read-ahead
    const int csx = Dx2 - 2 * C::OCW + 1;
    constexpr int G = C::P::G, LG = C::P::LOG2G;
    const int pw_a = ((G - 1 + (Dx2 - 1) + (full ? 1 : 0) + G - 1) >> LG) + 1, pw_b = ((G - 1 + csx - 2) >> LG) + C::GPR + 1;
    r.lds_pw = 4 * ((pw_a > pw_b ? pw_a : pw_b) | (MIMC3_EVEN_PITCH ? 0 : 1));   // odd dword pitch: lanes that own consecutive rows hit distinct banks
    // NCC cache slots per point: the certain set (<= 9 per pivot) + room for the climbs; long corridors
    // (many pivots) climb further.  Points that still overflow are redone by the general kernel.
    static const int slack_env = getenv("MIMC3_U8_CACHE_SLACK") ? atoi(getenv("MIMC3_U8_CACHE_SLACK")) : 0;   // tests shrink it to force the overflow path
    // the queue empties every batch; the certain set (<= 9 cells per pivot) is the largest batch
    int cap = 9 * max_npiv + (slack_env ? slack_env : (max_npiv <= 20 ? 16 : 4 * max_npiv));
    if (cap > cells) cap = cells;
    if (cap < 16) cap = 16;
    r.lds_list_cap = cap + 16;
    size_t off = (size_t)r.lds_pw * (Dy2 + (full ? 1 : 0));     // + the zero row behind a full-square search area
    if (C::COMPACT) {    // 16-bit entries for the (csx-2) x (csy-2) cells + value slots for the cells a point asks for
        const size_t ncell = (size_t)(csx - 2) * (Dy2 - 2 * C::OCW - 1);
        size_t nslot = 16 * (size_t)max_npiv + 64;
        if (nslot > ncell) nslot = ncell;
        if (nslot > 16383) nslot = 16383;
        r.lds_nslot = (int)nslot;
        off = (off + 15) & ~(size_t)15; r.lds_off_val = (int)off; off += (2 * ncell + 3) & ~(size_t)3;
        off = (off + 15) & ~(size_t)15; r.lds_off_vals = (int)off; off += 4 * nslot;
    } else {
    off = (off + 15) & ~(size_t)15; r.lds_off_val = (int)off; off += 4 * (size_t)(csx - 2) * (Dy2 - 2 * C::OCW - 1);   // (csx-2) x (csy-2) cache words
    }
    off = (off + 15) & ~(size_t)15; r.lds_off_vis = (int)off; off += 4 * (size_t)(((csx + 31) >> 5) * (Dy2 - 2 * C::OCW + 1));
    off = (off + 15) & ~(size_t)15; r.lds_off_list = (int)off; off += 2 * (size_t)r.lds_list_cap;
    off = (off + 15) & ~(size_t)15; r.lds_off_sums = (int)off; off += sizeof(typename C::P::Store) * 6 * kSumBatch + 128;
    off = (off + 15) & ~(size_t)15; r.lds_off_piv = (int)off; off += 8 * (size_t)max_npiv;
    if (C::MANYP) {     // per pivot: 8 B trajectory + 1 B (scans recorded | still climbing << 7) + 1 B scans really performed
        off = (off + 15) & ~(size_t)15; r.lds_off_traj = (int)off; off += 10 * (size_t)max_npiv;
    }
    if (C::CHIP_COPY) { off = (off + 15) & ~(size_t)15; r.lds_off_chip = (int)off; off += (size_t)C::CPITCH * C::CW; }
    if (C::SPARSE) {
        off = (off + 15) & ~(size_t)15; r.lds_off_lw = (int)off; off += (C::COMPACT ? 2 : 4) * (size_t)kLwCap;
        off = (off + 15) & ~(size_t)15; r.lds_off_lc = (int)off; off += (C::COMPACT ? 2 : 4) * (size_t)kLcCap;
    }
    off = (off + 15) & ~(size_t)15;
    return off;
}
static constexpr size_t kLdsCapBytes = 160 * 1024;

// hipFuncAttributeMaxDynamicSharedMemorySize is per device: remember which devices have it for this instantiation
template <class C>
static void px_set_lds_attr()
{
    static std::atomic<unsigned long long> done{0ull};   // bit d = device d configured (several devices may launch from several host threads)
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 64 && ((done.load() >> dev) & 1ull)) return;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&match_ncc_dlc_px<C>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsCapBytes);
    if (dev < 64) done.fetch_or(1ull << dev);
}

template <class C>
static hipError_t launch_cfg(MatchU8Args a, int max_abs_u, int max_abs_v, int max_npiv, hipStream_t stream)
{
    const size_t off = px_layout<C>(&a, max_abs_u, max_abs_v, max_npiv);
    static const bool lds_dbg = getenv("MIMC3_LDS_DEBUG") != nullptr;
    if (lds_dbg && !a.dry_run) fprintf(stderr, "[mimc3 lds] ocw %d: %zu bytes per workgroup\n", C::OCW, off);
    if (off > kLdsCapBytes) return hipErrorInvalidValue;   // nothing was launched: callers probe with dry_run and take the general kernel instead
    if (a.dry_run) return hipSuccess;
    px_set_lds_attr<C>();
    const unsigned nb = (unsigned)((a.N + 7) & ~7);
    static const int dbg = getenv("MIMC3_U8_DEBUG_STOP") ? atoi(getenv("MIMC3_U8_DEBUG_STOP")) : 0;
    // speculative 3x3 blocks requested ahead along a straight move: pays on the small chips (cheap evaluations, idle
    // lanes), costs on the big ones (every speculative cell is 60^2..81^2 pixels); measured 1 vs 0: +1 % / -4 %
    static const int look = getenv("MIMC3_U8_LOOKAHEAD") ? atoi(getenv("MIMC3_U8_LOOKAHEAD")) : -1;
    // (re-measured on the final build, C2's pair: on the RAW pair the big chips gain too -- smooth NCC surfaces, straight
    //  climbs: u8 ocw 30 / 40 -1.8 / -1.4 %, f32 ocw 40 -1.3 % -- while the filtered pairs, which are what the u8-through-
    //  offsets and u16 policies see in the program, lose: d/dx +1.3 / +2 %, Laplacian +4.4 %)
    //  (the small chips of those two policies as well: Laplacian ocw 7 / 15 -3.2 / -5 %, d/dx ocw 15 -1.3 % without look-ahead)
    constexpr bool kNoisy = C::P::SRC16 || std::is_same<typename C::P, PxU16>::value;
    // (long corridors on a big chip -- BASELINE C4: 33 pivots, ocw 32 -- lose with it too: 166.7 vs 165.6 ms)
    a.lookahead = look >= 0 ? look : ((kNoisy || (C::LPC >= 64 && max_npiv > 24)) ? 0 : 1);
    a.debug_stop = dbg;
    static unsigned long long *d_stats = nullptr;
    static const bool want_stats = getenv("MIMC3_U8_STATS") != nullptr;
    static size_t stats_n = 0;
    static std::mutex stats_mu;                      // diagnostics only: the buffer is shared by every launch of this instantiation
    std::unique_lock<std::mutex> stats_lock(stats_mu, std::defer_lock);
    if (want_stats) stats_lock.lock();
    if (want_stats && stats_n < (size_t)nb) {
        if (d_stats) (void)hipFree(d_stats);
        (void)hipMalloc(&d_stats, kStatW * sizeof(unsigned long long) * (size_t)nb);
        stats_n = nb;
    }
    a.stats = want_stats ? d_stats : nullptr;
    if (want_stats) (void)hipMemsetAsync(d_stats, 0, kStatW * sizeof(unsigned long long) * (size_t)nb, stream);
    static const size_t lds_pad = getenv("MIMC3_LDS_PAD") ? (size_t)atoi(getenv("MIMC3_LDS_PAD")) : 0;   // tuning: find the occupancy steps (LDS is granted in 256-byte units)
    hipLaunchKernelGGL(match_ncc_dlc_px<C>, dim3(nb), dim3(C::NT), off + lds_pad, stream, a);
    if (want_stats) {
        (void)hipStreamSynchronize(stream);
        unsigned long long *hh = (unsigned long long *)malloc(kStatW * sizeof(unsigned long long) * (size_t)nb);
        (void)hipMemcpy(hh, d_stats, kStatW * sizeof(unsigned long long) * (size_t)nb, hipMemcpyDeviceToHost);
        unsigned long long h[kStatW] = {0};
        for (size_t b = 0; b < (size_t)nb; b++) for (int i = 0; i < kStatW; i++) h[i] += hh[kStatW * b + i];
        free(hh);
        fprintf(stderr, "[mimc3 u8 stats] cycles/point: stage %.0f chip %.0f request %.0f eval+fit %.0f spec %.0f pre-replay %.0f replay-loop %.0f publish %.0f\n",
                (double)h[0] / a.N, (double)h[1] / a.N, (double)h[2] / a.N, (double)h[3] / a.N, (double)h[4] / a.N,
                (double)h[5] / a.N, (double)h[6] / a.N, (double)h[7] / a.N);
        {
            int32_t novf = 0;
            if (a.ovf_count) (void)hipMemcpy(&novf, a.ovf_count, sizeof(novf), hipMemcpyDeviceToHost);
            fprintf(stderr, "[mimc3 u8 stats] points handed to the general kernel: %d of %d\n", novf, a.N);
        }
        fprintf(stderr, "[mimc3 u8 stats] cells/point: clean-box %.1f dirty-box %.1f in %.1f evaluation batches; null lists in use %.3f of points, %.0f window / %.0f chip entries per point\n",
                (double)h[8] / a.N, (double)h[9] / a.N, (double)h[10] / a.N, (double)h[11] / a.N, (double)h[12] / a.N, (double)h[13] / a.N);
        fprintf(stderr, "[mimc3 u8 stats] replay on pre-decoded scan centres: %.3f of points; dirty-list boxes without a null: %.2f per point\n", (double)h[14] / a.N, (double)h[15] / a.N);
    }
    return hipGetLastError();
}

// Big-chip configurations come in two LDS forms (PxCfg::COMPACT): the compact one is taken when it holds more workgroups per
// CU than the regular one (at most four: 128 VGPRs x 4 waves per workgroup), or when only it fits at all.
template <class CD, class CC>
static hipError_t launch_pick(MatchU8Args a, int max_abs_u, int max_abs_v, int max_npiv, hipStream_t stream)
{
    MatchU8Args t1 = a, t2 = a;
    const size_t bd = px_layout<CD>(&t1, max_abs_u, max_abs_v, max_npiv), bc = px_layout<CC>(&t2, max_abs_u, max_abs_v, max_npiv);
    auto wgs = [](size_t b) { const size_t g = (b + 255) & ~(size_t)255; const int w = g ? (int)(kLdsCapBytes / g) : 0; return w > 4 ? 4 : w; };
    const bool full = a.win_half > 0;
    const int Dx2 = full ? 2 * a.win_half + 1 : 2 * (max_abs_u + CD::OCW + 2) + 1, Dy2 = full ? 2 * a.win_half + 1 : 2 * (max_abs_v + CD::OCW + 2) + 1;
    static const int force = getenv("MIMC3_COMPACT") ? atoi(getenv("MIMC3_COMPACT")) : -1;      // tests / tuning: 0 never, 1 whenever possible
    const bool possible = Dx2 <= 255 && Dy2 <= 255 && bc <= kLdsCapBytes;                       // 8-bit window coordinates in the null lists
    const bool want = force >= 0 ? force != 0 : wgs(bc) > wgs(bd);
    if (possible && want) return launch_cfg<CC>(a, max_abs_u, max_abs_v, max_npiv, stream);
    return launch_cfg<CD>(a, max_abs_u, max_abs_v, max_npiv, stream);
}

// Big chips whose windows leave LDS for more than four workgroups per CU (BASELINE C2's geometry at ocw 30 / 32: six / five) run faster
// at a higher occupancy target although the smaller register budget spills (measured on C2's grid, ms per pass at 4 / 5 / 6 waves per
// SIMD: u8 ocw 30 10.59 / 9.38 / 8.98, ocw 32 10.82 / 9.70 / 11.90, u8o ocw 30 14.26 / 13.07 / 13.97, ocw 32 14.58 / 13.58 / 16.43; ocw 40,
// the u16 kernels and the ocw-7 kernels lose: profiles/round4/kbench_occupancy_target_sweep_big_chips.txt).  CH = the same
// configuration at that target; it is taken when the launch's LDS need admits as many workgroups, else the four-wave form (or its
// compact variant) as before.
template <class CD, class CH>
static bool lds_admits_high(const MatchU8Args &a, int max_abs_u, int max_abs_v, int max_npiv)
{
    static const int off = getenv("MIMC3_HIGH_OCC") ? atoi(getenv("MIMC3_HIGH_OCC")) : 1;      // tuning / A-B: 0 = never
    static const bool forced_form = getenv("MIMC3_COMPACT") != nullptr;                          // tests force an LDS form: launch_pick's business
    if (!off || forced_form) return false;
    MatchU8Args t = a;
    const size_t b = px_layout<CD>(&t, max_abs_u, max_abs_v, max_npiv);
    const size_t g = (b + 255) & ~(size_t)255;
    return g != 0 && (int)(kLdsCapBytes / g) >= CH::MINW;
}

bool match_f32x_supported(int ocw, int max_reach_u, int max_reach_v)
{
    if (!(ocw == 7 || ocw == 15 || ocw == 16 || ocw == 30 || ocw == 32 || ocw == 40)) return false;     // chip rows must fit the register image
    return max_reach_u + 12 <= kU8Pad && max_reach_v + 12 <= kU8Pad && max_reach_u <= 120 && max_reach_v <= 120;
}

hipError_t launch_match_f32x(MatchU8Args a, int max_abs_u, int max_abs_v, int max_npiv, hipStream_t stream)
{
    // The chip of these kernels stays in registers although several forms spill (integral ocw 15 / 16 / 40: 124 / 172 / 420 B of
    // scratch per lane): read from an LDS copy instead (PxCfg::CHIP_LDS, no scratch at all, four waves per SIMD instead of three) the
    // integral kernels measured 4.50 / 11.29 / 10.21 / 47.9 / 107.9 ms at ocw 7 / 15 / 16 / 30 / 40 on BASELINE C2's grid against
    // 4.11 / 8.28 / 10.26 / 35.1 / 92.0 with the chip in registers -- the second LDS stream and the chunked row loop cost more than the
    // spills do.  Only ocw 16 takes the LDS form (same time, no scratch traffic: 512 MB of writes per launch less).
    if (a.N <= 0 && !a.dry_run) return hipSuccess;
    if (a.sat0 && a.sat1) {        // integral 16-bit DN: the planes come with summed-area tables
        switch (a.ocw) {
        case 7: return launch_cfg<PxCfg<PxF32i, 7, 16, 2, 4>>(a, max_abs_u, max_abs_v, max_npiv, stream);
        case 15: return launch_cfg<PxCfg<PxF32i, 15, 32, 2, 3>>(a, max_abs_u, max_abs_v, max_npiv, stream);
        case 16: return launch_cfg<PxCfg<PxF32i, 16, 32, 2, 4, true>>(a, max_abs_u, max_abs_v, max_npiv, stream);
        case 30: return launch_cfg<PxCfg<PxF32i, 30, 64, 4, 2>>(a, max_abs_u, max_abs_v, max_npiv, stream);
        case 32: return launch_cfg<PxCfg<PxF32i, 32, 64, 4, 2>>(a, max_abs_u, max_abs_v, max_npiv, stream);
        case 40: return launch_cfg<PxCfg<PxF32i, 40, 64, 4, 2>>(a, max_abs_u, max_abs_v, max_npiv, stream);
        default: return hipErrorInvalidValue;
        }
    }
    switch (a.ocw) {
    case 7: return launch_cfg<PxCfg<PxF32, 7, 16, 2, 4>>(a, max_abs_u, max_abs_v, max_npiv, stream);
    case 15: return launch_cfg<PxCfg<PxF32, 15, 32, 2, 3>>(a, max_abs_u, max_abs_v, max_npiv, stream);
    case 16: return launch_cfg<PxCfg<PxF32, 16, 32, 2, 3>>(a, max_abs_u, max_abs_v, max_npiv, stream);
    case 30: return launch_cfg<PxCfg<PxF32, 30, 64, 4, 2>>(a, max_abs_u, max_abs_v, max_npiv, stream);
    case 32: return launch_cfg<PxCfg<PxF32, 32, 64, 4, 2>>(a, max_abs_u, max_abs_v, max_npiv, stream);
    case 40: return launch_cfg<PxCfg<PxF32, 40, 64, 4, 2>>(a, max_abs_u, max_abs_v, max_npiv, stream);
    default: return hipErrorInvalidValue;
    }
}

hipError_t launch_match_u16(MatchU8Args a, int max_abs_u, int max_abs_v, int max_npiv, hipStream_t stream)
{
    if (a.N <= 0 && !a.dry_run) return hipSuccess;
    // Rectangular pivot sets (control-point stage): the many-pivot forms of its two chip sizes.  Those launches are a few
    // hundred points of several hundred cells each -- latency-bound, so the small chip also gets four waves per point and
    // 32 lanes per cell (8 cells per round instead of 4 on one wave).
    if (max_npiv > 64) {
        if (a.ocw == 15) return launch_cfg<PxCfg<PxU16, 15, 32, 4, 3, false, true>>(a, max_abs_u, max_abs_v, max_npiv, stream);
        if (a.ocw == 30) return launch_cfg<PxCfg<PxU16, 30, 64, 4, 4, false, true>>(a, max_abs_u, max_abs_v, max_npiv, stream);
    }
    switch (a.ocw) {
    case 7: return launch_cfg<PxCfg<PxU16, 7, 16, 1, 4>>(a, max_abs_u, max_abs_v, max_npiv, stream);
    case 15: return launch_cfg<PxCfg<PxU16, 15, 32, 1, 3>>(a, max_abs_u, max_abs_v, max_npiv, stream);
    case 16: return launch_cfg<PxCfg<PxU16, 16, 32, 1, 3>>(a, max_abs_u, max_abs_v, max_npiv, stream);
    case 30: return launch_cfg<PxCfg<PxU16, 30, 64, 4, 4>>(a, max_abs_u, max_abs_v, max_npiv, stream);   // (chip from LDS: 24.5 vs 21.6 ms)
    case 32: return launch_cfg<PxCfg<PxU16, 32, 64, 4, 3>>(a, max_abs_u, max_abs_v, max_npiv, stream);
    // 81-row chip at 2 px per dword = 52 chip dwords per lane: read from the LDS copy instead (3 waves/SIMD, 44 B of scratch
    // outside the loops): 37.9 ms against 46.1 with the chip in 216 VGPRs at 2 waves/SIMD, measured side by side
    case 40: return launch_cfg<PxCfg<PxU16, 40, 64, 4, 3, true>>(a, max_abs_u, max_abs_v, max_npiv, stream);
    default: return hipErrorInvalidValue;
    }
}

hipError_t launch_match_u8o(MatchU8Args a, int max_abs_u, int max_abs_v, int max_npiv, hipStream_t stream)
{
    if (a.N <= 0 && !a.dry_run) return hipSuccess;
    if (max_npiv > 64) {
        if (a.ocw == 15) return launch_cfg<PxCfg<PxU8o, 15, 32, 4, 4, false, true>>(a, max_abs_u, max_abs_v, max_npiv, stream);
        if (a.ocw == 30) return launch_cfg<PxCfg<PxU8o, 30, 64, 4, 4, false, true>>(a, max_abs_u, max_abs_v, max_npiv, stream);
    }
    switch (a.ocw) {       // the u8 configurations, fed from u16 planes through per-point offsets
    case 7: return launch_cfg<PxCfg<PxU8o, 7, 16, 1, 4>>(a, max_abs_u, max_abs_v, max_npiv, stream);
    case 15: return launch_cfg<PxCfg<PxU8o, 15, 16, 1, 4>>(a, max_abs_u, max_abs_v, max_npiv, stream);
    case 16: return launch_cfg<PxCfg<PxU8o, 16, 16, 1, 4>>(a, max_abs_u, max_abs_v, max_npiv, stream);
    case 30:
        if (lds_admits_high<PxCfg<PxU8o, 30, 64, 4, 4>, PxCfg<PxU8o, 30, 64, 4, 5>>(a, max_abs_u, max_abs_v, max_npiv))
            return launch_cfg<PxCfg<PxU8o, 30, 64, 4, 5>>(a, max_abs_u, max_abs_v, max_npiv, stream);
        return launch_cfg<PxCfg<PxU8o, 30, 64, 4, 4>>(a, max_abs_u, max_abs_v, max_npiv, stream);
    case 32:
        if (lds_admits_high<PxCfg<PxU8o, 32, 64, 4, 4>, PxCfg<PxU8o, 32, 64, 4, 5>>(a, max_abs_u, max_abs_v, max_npiv))
            return launch_cfg<PxCfg<PxU8o, 32, 64, 4, 5>>(a, max_abs_u, max_abs_v, max_npiv, stream);
        return launch_cfg<PxCfg<PxU8o, 32, 64, 4, 4>>(a, max_abs_u, max_abs_v, max_npiv, stream);
    case 40: return launch_cfg<PxCfg<PxU8o, 40, 64, 4, 4>>(a, max_abs_u, max_abs_v, max_npiv, stream);
    default: return hipErrorInvalidValue;
    }
}

// Is PxU8o worth trying on this plane?  One workgroup per 128x128 tile: does the tile's non-null range fit 8 bits?
__global__ __launch_bounds__(256) void range_tiles(const unsigned short *__restrict__ plane, int H, int W, int Wp, int pad, int *out2)
{
    const int x0 = blockIdx.x * 128, y0 = blockIdx.y * 128;
    int mn = 1 << 20, mx = -1;
    for (int q = threadIdx.x; q < 128 * 128; q += 256) {
        const int x = x0 + (q & 127), y = y0 + (q >> 7);
        if (x < W && y < H) {
            const int v = plane[(size_t)(y + pad) * Wp + x + pad];
            if (v) { mn = min(mn, v); mx = max(mx, v); }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { mn = min(mn, __shfl_xor(mn, o, 64)); mx = max(mx, __shfl_xor(mx, o, 64)); }
    __shared__ int smn[4], smx[4];
    if ((threadIdx.x & 63) == 0) { smn[threadIdx.x >> 6] = mn; smx[threadIdx.x >> 6] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; w++) { mn = min(mn, smn[w]); mx = max(mx, smx[w]); }
        if (mx >= mn) { atomicAdd(&out2[1], 1); if (mx - mn <= 254) atomicAdd(&out2[0], 1); }
    }
}

// ---- per-tile ranges of a u16 plane: 16 lanes = the 16 rows of one 16x16-pixel tile -------------------------------------
__global__ __launch_bounds__(256) void range_tiles16(const unsigned short *__restrict__ plane, int Hp, int Wp, int tw, int ntiles, uint32_t *__restrict__ tiles)
{
    const int gid = blockIdx.x * 256 + threadIdx.x;
    const int tile = gid >> 4, r = gid & 15;
    int mn = 0xffff, mx = 0;
    if (tile < ntiles) {
        const int ty = tile / tw, tx = tile - ty * tw;
        const int row = ty * 16 + r, col0 = tx * 16;
        if (row < Hp) {
            const unsigned short *q = plane + (size_t)row * Wp + col0;
#pragma unroll
            for (int c = 0; c < 16; c++) {
                const int v = (col0 + c < Wp) ? q[c] : 0;
                mx = max(mx, v);
                mn = v ? min(mn, v) : mn;
            }
        }
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) { mn = min(mn, __shfl_xor(mn, o, 16)); mx = max(mx, __shfl_xor(mx, o, 16)); }
    if (tile < ntiles && r == 0) tiles[tile] = (uint32_t)mn | ((uint32_t)mx << 16);
}

hipError_t launch_range_tiles16(const unsigned short *plane, int Hp, int Wp, uint32_t *tiles, hipStream_t s)
{
    const int tw = (Wp + 15) / 16, th = (Hp + 15) / 16, nt = tw * th;
    hipLaunchKernelGGL(range_tiles16, dim3((unsigned)((nt * 16 + 255) / 256)), dim3(256), 0, s, plane, Hp, Wp, tw, nt, tiles);
    return hipGetLastError();
}

hipError_t launch_range_tiles(const unsigned short *plane, int H, int W, int Wp, int pad, int *d_out2, hipStream_t s)
{
    hipLaunchKernelGGL(range_tiles, dim3((W + 127) / 128, (H + 127) / 128), dim3(256), 0, s, plane, H, W, Wp, pad, d_out2);
    return hipGetLastError();
}

bool match_u8_supported(int ocw, int max_reach_u, int max_reach_v)
{
    if (!(ocw == 7 || ocw == 15 || ocw == 16 || ocw == 30 || ocw == 32 || ocw == 40)) return false;
    // the window hangs over the image edge by at most |last pivot| + |CP offset| + 2 pixels (+ up to 7
    // bytes of aligned read-ahead): all of it must stay inside the zero border; compact cell coordinates
    // are packed in 8 bits (csx = 2|last|+6 <= 255)
    return max_reach_u + 12 <= kU8Pad && max_reach_v + 12 <= kU8Pad && max_reach_u <= 120 && max_reach_v <= 120;
}

hipError_t launch_match_u8(MatchU8Args a, int max_abs_u, int max_abs_v, int max_npiv, hipStream_t stream)
{
    if (a.N <= 0 && !a.dry_run) return hipSuccess;
    if (max_npiv > 64) {
        if (a.ocw == 15) return launch_cfg<PxCfg<PxU8, 15, 32, 4, 4, false, true>>(a, max_abs_u, max_abs_v, max_npiv, stream);
        if (a.ocw == 30) return launch_cfg<PxCfg<PxU8, 30, 64, 4, 4, false, true>>(a, max_abs_u, max_abs_v, max_npiv, stream);
    }
    switch (a.ocw) {
    case 7: return launch_cfg<PxCfg<PxU8, 7, 16, 1, 4>>(a, max_abs_u, max_abs_v, max_npiv, stream);
    case 15: return launch_cfg<PxCfg<PxU8, 15, 16, 1, 4>>(a, max_abs_u, max_abs_v, max_npiv, stream);
    case 16: return launch_cfg<PxCfg<PxU8, 16, 16, 1, 4>>(a, max_abs_u, max_abs_v, max_npiv, stream);
    // big chips: 4 waves share one point's LDS image (one cell per wave and round)
    case 30:
        if (lds_admits_high<PxCfg<PxU8, 30, 64, 4, 4>, PxCfg<PxU8, 30, 64, 4, 6>>(a, max_abs_u, max_abs_v, max_npiv))
            return launch_cfg<PxCfg<PxU8, 30, 64, 4, 6>>(a, max_abs_u, max_abs_v, max_npiv, stream);
        return launch_pick<PxCfg<PxU8, 30, 64, 4, 4>, PxCfg<PxU8, 30, 64, 4, 4, false, false, true>>(a, max_abs_u, max_abs_v, max_npiv, stream);
    case 32:
        if (lds_admits_high<PxCfg<PxU8, 32, 64, 4, 4>, PxCfg<PxU8, 32, 64, 4, 5>>(a, max_abs_u, max_abs_v, max_npiv))
            return launch_cfg<PxCfg<PxU8, 32, 64, 4, 5>>(a, max_abs_u, max_abs_v, max_npiv, stream);
        return launch_pick<PxCfg<PxU8, 32, 64, 4, 4>, PxCfg<PxU8, 32, 64, 4, 4, false, false, true>>(a, max_abs_u, max_abs_v, max_npiv, stream);
    case 40: return launch_pick<PxCfg<PxU8, 40, 64, 4, 4>, PxCfg<PxU8, 40, 64, 4, 4, false, false, true>>(a, max_abs_u, max_abs_v, max_npiv, stream);
    default: return hipErrorInvalidValue;
    }
}

}  // namespace mimc3
