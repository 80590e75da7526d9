// match_mx_kernel.hip -- DLC/NCC matcher for gfx950 on the matrix cores: 8-bit imagery, DENSE correlation surfaces.
//
// Same contract as match_px_kernel.hip (matching_ncc_dlc_2, MIMC_module.c:805-842), for the points whose reachable cell grid
// fits one 32 x 32 tile.  One workgroup of two wave64 = one grid point (wave w owns the cell columns [16 w, 16 w + 16)).  Instead of
// evaluating the cells a hill climb asks for (a request queue, evaluation batches, a speculative climb that waits for them), the
// kernel builds the COMPLETE surface of the tile:
//
//   sxy[dy][s] = sum_r sum_k a[r][k] * b[r + dy][k + s]                              (MIMC_module.c:719-733, the product stream)
//              = sum_r (W_r T_r)[dy][s]     W_r[dy][j] = b[r + dy][j]   A operand: 16 window rows, plain 16-byte LDS reads
//                                           T_r[j][s]  = a[r][j - s]    B operand: the Toeplitz band of chip row r, 0 outside
//   i.e. per chip row, 64 window columns and 16 x 16 cells one v_mfma_i32_16x16x64_i8, all into the cells' i32 accumulators (exact
//   integers).  u8 -> i8:  a' = a - 128 (a ^ 0x80), b' = b - 128, Toeplitz padding a' = 0:
//              sum ab = sum a'b' + 128 sum_box b + 128 sum a - 128^2 CW^2        (a null is a zero factor: no mask anywhere)
//   * window-side sums  sy = sum_box b, syy = sum_box b^2  of every cell: row-box sums by MFMA against a band of ones
//     (b^2 as two byte planes), then the vertical CW-row sums by a second MFMA against a band of ones (the row sums split
//     into byte planes; an accumulator tile is the next MFMA's B operand as it stands: its rows are the K index);
//   * null pixels (the two forms for null-ridden points, built and tested but not taken by default: launch_match_mx has the
//     measurements).  A window null q of the box takes a(q) out of n, sx, sxx: with zb = [b == 0] as the A operand the corrections
//     are three more correlations on the same pipe -- sum zb (box count), sum a zb (same B operand as sxy), sum a^2 zb (a^2 as two
//     byte planes of the chip) -- the window-null form.  A chip null p takes b(p) out of sy, syy: with za = [a == 0] as the B operand
//     -- formed from the Toeplitz rows of the chip rows that hold nulls, the only ones visited -- sum b za, sum b^2 za (b^2 as two
//     byte planes of the A operand), and sum zb za for n -- the general form.  The never-written last window row / column (T4) are
//     zeros of the staged tile, i.e. nulls like any other when the window's nulls are correlated anyway; in the clean form they are
//     applied in closed form: they take the chip's last column / row out of n, sx, sxx (table queries of the chip plane);
//   * the NCC of all 1,024 cells (:734, f64, no contraction; a guarded fast reciprocal square root with the exact sqrt / division
//     for the cells within 2^13 f64 ulps of an f32 rounding boundary) -> one f32 surface in LDS;
//   * the climb of every pivot on the complete surface (lane k = pivot k), the exact replay of the reference's sequential
//     visited-set semantics (:691-753) and the 3x3 fit (:757-788) as in match_px_kernel.hip -- on wave 0, wave 1 has left by then.
// What this kernel does not take it hands on through a flag byte per grid point (no host round trip, no shared counter: same-address
// atomics of 100,000 points serialise into a millisecond): a point with nulls in its window or chip, more than 64 pivots, a pivot
// set wider than the tile, a climb that leaves the tile or outlasts the 16 recorded scans is flagged for the register-tiled kernel
// (match_px_kernel.hip), which runs right behind in flag mode (or for this kernel's other forms, when they are switched on).
#include <hip/hip_runtime.h>
#include <atomic>
#include <mutex>
#include <stdint.h>
#include <stdlib.h>
#include <stdio.h>
#include "match_kernel.h"
#include "sat_kernel.h"

namespace mimc3 {

namespace mx {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

__device__ __forceinline__ uint32_t alignb(uint32_t hi, uint32_t lo, uint32_t s) { return __builtin_amdgcn_alignbyte(hi, lo, s); }
__device__ __forceinline__ uint32_t perm(uint32_t hi, uint32_t lo, uint32_t sel) { return __builtin_amdgcn_perm(hi, lo, sel); }
// a wave-uniform value the compiler loaded through the vector memory path (global stores in the kernel keep it from using scalar loads): into SGPRs
__device__ __forceinline__ unsigned long long uni64(unsigned long long v)
{
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ v4i mfma(const v4i &a, const v4i &b, const v4i &c) { return __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c, 0, 0, 0); }

// first-wins arg-max over the 16 lanes of a DPP row (lexicographic max on (value, -index)); VALU only
__device__ __forceinline__ void argmax_row16(float &v, int &i)
{
#define MIMC3_MX_ARGMAX_STEP(ctrl)                                                                        \
    {                                                                                                     \
        const float ov = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, 0xF, 0xF, true)); \
        const int oi = __builtin_amdgcn_update_dpp(0, i, ctrl, 0xF, 0xF, true);                           \
        const bool t = (ov > v) || (ov == v && oi < i);                                                   \
        v = t ? ov : v; i = t ? oi : i;                                                                   \
    }
    MIMC3_MX_ARGMAX_STEP(0xB1) MIMC3_MX_ARGMAX_STEP(0x4E) MIMC3_MX_ARGMAX_STEP(0x141) MIMC3_MX_ARGMAX_STEP(0x140)
#undef MIMC3_MX_ARGMAX_STEP
}

template <int OCW_, bool GEN_, bool CN_ = GEN_>
struct Cfg {
    static constexpr int OCW = OCW_, CW = 2 * OCW_ + 1, NPX = CW * CW;
    static constexpr bool GEN = GEN_;                   // null pixels in the window ...
    static constexpr bool CN = CN_;                     // ... and in the chip (false with GEN: the window-null form -- half the accumulators, a wave more per SIMD)
    // Two waves per grid point: wave w owns the cell columns [16 w, 16 w + 16) of the 32 x 32 tile (two 16 x 16 MFMA tiles, rows
    // 0..15 and 16..31), so each wave carries half the accumulators.  The front phases run on both, the climb / replay / fit on wave 0
    // after wave 1 has left: registers per wave, not LDS, bound how many points a CU works on.
    static constexpr int NW = 2, NT = 64 * NW;
    static constexpr int KW = CW + 31;                  // window columns (and rows) the 32 x 32 cells of a tile reach
    static constexpr int KCW = (CW + 15 + 63) / 64;     // MFMAs per chip row and 16 x 16 cell tile (K chunks of 64 window columns, from column 16 w)
    static constexpr int RDW = 16 * (NW - 1) + 64 * KCW;   // bytes of a tile row some wave reads
    static constexpr int SW = (KW + 15) & ~15;          // bytes of a tile row that are staged (what lies beyond is only ever weighted 0)
    static constexpr int PW0 = RDW > SW ? RDW : SW;
    static constexpr int PW = ((PW0 / 16) & 1) ? PW0 : PW0 + 16;   // pitch 16 x odd: few bank conflicts between the rows of a ds_read_b128 group
    static constexpr int NTR = (KW + 15) / 16;          // 16-row tiles of the row-box sums
    static constexpr int KCV = (NTR + 3) / 4;           // K chunks (64 tile rows) of the vertical sums
    static constexpr int CD = (CW + 3) / 4;             // dwords per chip row
    static constexpr int LASTN = CW - 4 * (CD - 1);     // valid bytes of a chip row's last dword
    static constexpr int CP0 = (CW + 15 + 3) & ~3;
    static constexpr int CP = CP0 > 64 * KCW ? CP0 : 64 * KCW;     // chip plane pitch: a row's right padding is the next row's left padding
    static constexpr int CH0 = 16;                      // leading zeros (row 0's left padding)
    static constexpr int CWE = CW + (CW & 1);           // chip rows the product loop walks: two per trip (an all-zero row behind an odd chip)
    static constexpr int CHB = (CH0 + CWE * CP + 16 + 15) & ~15;   // one chip plane (+ the read-ahead of the last row's last lane)
    static constexpr int VP = 33;                       // pitch (words) of the NCC surface
    static constexpr int LDS_W = (16 * NTR + 1) * PW;   // (+ the row the product loop's zero chip row reads)
    static constexpr int LDS_VAL = 4 * 32 * VP;
    static constexpr int NPL = GEN_ ? 3 : 1;            // chip planes: a', and in the general form (a^2 & 255)', (a^2 >> 8)' (the B operands of the window-null corrections)
    // general form, small chips: the window's null plane (0x80 where b == 0) staged next to the tile, so that the A operand of the window-null
    // correlations is a load; the big chips, short of LDS, form it from the tile's bytes in registers (3 instructions per dword)
    static constexpr bool ZPL = GEN_ && OCW_ <= 16;
    static constexpr int OFF_Z = ((LDS_W > LDS_VAL ? LDS_W : LDS_VAL) + 15) & ~15;       // the surface reuses the tile's bytes once the sums are in registers
    static constexpr int OFF_CH = OFF_Z + (ZPL ? LDS_W : 0);
    static constexpr int OFF_VIS = OFF_CH + NPL * CHB;
    static constexpr int OFF_PIV = OFF_VIS + 128;      // the pivots' starts, parked for wave 0's climbs
    static constexpr int OFF_RM = OFF_PIV + 512;       // general form: bit r = chip row r holds a null pixel
    static constexpr int LDS = OFF_RM + 16;
    static constexpr int WGS = 163840 / ((LDS + 255) & ~255);      // workgroups per CU that LDS admits
    static constexpr int MINW0 = GEN_ ? (CN_ ? 3 : 4) : 8;          // occupancy target, waves per SIMD (register budget) ...
    static constexpr int MINW1 = (WGS * NW) / 4 > 0 ? (WGS * NW) / 4 : 1;                   // ... never above what LDS admits anyway
    static constexpr int MINW = MINW0 < MINW1 ? MINW0 : MINW1;
};

// The constant band operands, one table per chip size (constant-initialised device data).  v_mfma_i32_16x16x64_i8: lane (n = lane & 15,
// h = lane >> 4) holds byte i of K slot (h, i) of row / column n.
//   bm[c][lane]     : 0xff in the bytes of a chip-row B operand that hold chip pixels (the others are Toeplitz padding) -- the same bytes as hb
//   hb[c][lane]     : B operand of the row-box sums -- byte i of chunk c <-> window column x = 16 w + 64 c + 16 h + i, cell column
//                     s = 16 w + n:  1 if s <= x < s + CW  (the wave index cancels)
//   vb[m][c][lane]  : A operand of the vertical sums -- cell row dy = 16 m + n; K slot (h, i) of chunk c <-> tile row
//                     y = 64 c + 16 (i >> 2) + 4 h + (i & 3), where the accumulator layout of the row sums puts it (register i & 3 of row
//                     tile 4 c + (i >> 2)):  1 if dy <= y < dy + CW
template <int CW, int KCW, int KCV>
struct alignas(16) Bands {
    uint32_t hb[KCW][64][4], vb[2][KCV][64][4], bm[KCW][64][4];
    constexpr Bands() : hb(), vb(), bm()
    {
        for (int l = 0; l < 64; l++) {
            const int n = l & 15, h = l >> 4;
            for (int i = 0; i < 16; i++) {
                for (int c = 0; c < KCW; c++) {
                    const int x = 64 * c + 16 * h + i;
                    if (n <= x && x < n + CW) { hb[c][l][i >> 2] |= 1u << (8 * (i & 3)); bm[c][l][i >> 2] |= 0xffu << (8 * (i & 3)); }
                }
                for (int m = 0; m < 2; m++)
                    for (int c = 0; c < KCV; c++) {
                        const int dy = 16 * m + n, y = 64 * c + 16 * (i >> 2) + 4 * h + (i & 3);
                        if (dy <= y && y < dy + CW) vb[m][c][l][i >> 2] |= 1u << (8 * (i & 3));
                    }
            }
        }
    }
};
template <int CW, int KCW, int KCV> __device__ const Bands<CW, KCW, KCV> kBands{};

constexpr int kStatW = 8;

// u8 pixels back from the signed bytes of an A operand, squared, split into the byte planes (x^2 & 255) and (x^2 >> 8), as signed bytes
typedef unsigned short us2 __attribute__((ext_vector_type(2)));
// the squares of the four pixels of a dword as the two byte planes (x^2 & 255, x^2 >> 8): the bytes spread to 16-bit halves (v_perm),
// two packed 16-bit multiplies (a byte's square fits 16 bits), the planes picked out by v_perm -- 6 instructions
__device__ __forceinline__ void squares_dword(uint32_t x, uint32_t &lo, uint32_t &hi)
{
    const uint32_t u01 = perm(0u, x, 0x0c010c00u), u23 = perm(0u, x, 0x0c030c02u);           // [b0, 0, b1, 0], [b2, 0, b3, 0]  (selector 0x0c: the byte 0)
    const us2 p01 = __builtin_bit_cast(us2, u01), p23 = __builtin_bit_cast(us2, u23);
    const uint32_t q01 = __builtin_bit_cast(uint32_t, (us2)(p01 * p01)), q23 = __builtin_bit_cast(uint32_t, (us2)(p23 * p23));
    lo = perm(q23, q01, 0x06040200u);
    hi = perm(q23, q01, 0x07050301u);
}
__device__ __forceinline__ void squares(const v4i &a, v4i &lo, v4i &hi)
{
#pragma unroll
    for (int k = 0; k < 4; k++) {
        uint32_t l, g;
        squares_dword((uint32_t)a[k] ^ 0x80808080u, l, g);
        lo[k] = (int)(l ^ 0x80808080u); hi[k] = (int)(g ^ 0x80808080u);
    }
}
// 0x80 (the signed byte -128) in every byte of an operand that is a null pixel (its signed byte is 0x80, i.e. the pixel 0; Toeplitz padding
// is 0x00): three instructions per dword.  A correlation with this plane is -128 times the correlation with the 0 / 1 null mask -- exactly
__device__ __forceinline__ uint32_t null80(uint32_t x) { return x & ~((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) & 0x80808080u; }
__device__ __forceinline__ v4i null80v(const v4i &a)
{
    v4i z;
#pragma unroll
    for (int k = 0; k < 4; k++) z[k] = (int)null80((uint32_t)a[k]);
    return z;
}
// 1 in every byte of the A operand that is a null pixel (b == 0, i.e. b' == 0x80)
__device__ __forceinline__ v4i nullbytes(const v4i &a)
{
    v4i z;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint32_t x = (uint32_t)a[k] ^ 0x80808080u;
        const uint32_t nz = (((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x) & 0x80808080u;           // 0x80 in every non-zero byte
        z[k] = (int)((nz ^ 0x80808080u) >> 7);
    }
    return z;
}
// The four registers of a 16 x 16 accumulator tile (rows 4 h + 0..3 of one 16-row tile) as ONE dword of each byte plane of the next
// MFMA's B operand (signed bytes: ^ 0x80, except the single plane of values < 128)
__device__ __forceinline__ void planes2(const v4i &R, uint32_t &p0, uint32_t &p1)
{
    const uint32_t t01 = perm((uint32_t)R[1], (uint32_t)R[0], 0x05010400u), t23 = perm((uint32_t)R[3], (uint32_t)R[2], 0x05010400u);   // [r0.b0, r1.b0, r0.b1, r1.b1]
    p0 = perm(t23, t01, 0x05040100u) ^ 0x80808080u;
    p1 = perm(t23, t01, 0x07060302u) ^ 0x80808080u;
}
__device__ __forceinline__ void planes3(const v4i &R, uint32_t &p0, uint32_t &p1, uint32_t &p2)
{
    const uint32_t r0 = (uint32_t)R[0], r1 = (uint32_t)R[1], r2 = (uint32_t)R[2], r3 = (uint32_t)R[3];
    const uint32_t t01 = perm(r1, r0, 0x05010400u), t23 = perm(r3, r2, 0x05010400u);
    const uint32_t u01 = perm(r1, r0, 0x07030602u), u23 = perm(r3, r2, 0x07030602u);     // [r0.b2, r1.b2, r0.b3, r1.b3]
    p0 = perm(t23, t01, 0x05040100u) ^ 0x80808080u;
    p1 = perm(t23, t01, 0x07060302u) ^ 0x80808080u;
    p2 = perm(u23, u01, 0x05040100u) ^ 0x80808080u;
}
__device__ __forceinline__ uint32_t planes1(const v4i &R)
{
    const uint32_t t01 = perm((uint32_t)R[1], (uint32_t)R[0], 0x05010400u), t23 = perm((uint32_t)R[3], (uint32_t)R[2], 0x05010400u);
    return perm(t23, t01, 0x05040100u);
}

// diagnostics (MIMC3_MX_STATS): phase clocks stay in scalar registers and are stored once, by MIMC3_MX_STATS_OUT, before the kernel's normal exit
#define MIMC3_MX_STAMP(i)                                                                      \
    if (p.stats) {                                                                             \
        const unsigned long long t_now = __builtin_amdgcn_s_memtime();                         \
        t_ph[i] = t_now - t_prev;                                                              \
        t_prev = t_now;                                                                        \
    }
#define MIMC3_MX_STATS_OUT                                                                     \
    if (p.stats && lane == 0) { _Pragma("unroll") for (int i_ = 0; i_ < 7; i_++) p.stats[kStatW * (size_t)blockIdx.x + i_] = t_ph[i_]; }

template <class C>
__global__ __launch_bounds__(C::NT, C::MINW) void match_ncc_dlc_mx(MatchU8Args p)
{
    constexpr int OCW = C::OCW, CW = C::CW, NPX = C::NPX, KCW = C::KCW, KCV = C::KCV, NTR = C::NTR, PW = C::PW, CP = C::CP, CH0 = C::CH0, VP = C::VP, NT = C::NT;
    constexpr bool GEN = C::GEN;
    __shared__ __attribute__((aligned(16))) unsigned char smem[C::LDS];
    unsigned char *WT = smem;
    float *val = reinterpret_cast<float *>(smem);                       // [32][VP] NCC surface (over the tile, once it is consumed)
    unsigned char *CH = smem + C::OFF_CH;                               // chip planes
    uint32_t *vis = reinterpret_cast<uint32_t *>(smem + C::OFF_VIS);    // visited bits of the 32 tile rows
    unsigned long long t_prev = p.stats ? __builtin_amdgcn_s_memtime() : 0ull;
    unsigned long long t_ph[7] = {0, 0, 0, 0, 0, 0, 0};
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    int gidx = blockIdx.x;
    if (p.point_list) {                                      // list mode: the points another kernel handed over
        if (gidx >= *p.point_count) return;
        gidx = p.point_list[gidx];
    } else {
        const int nb = gridDim.x, per = nb >> 3;
        if (per > 0 && gidx < per * 8) gidx = (gidx & 7) * per + (gidx >> 3);   // XCD-contiguous point order
    }
    if (gidx >= p.N) return;
    if (p.point_flags && p.point_flags[gidx] != (uint8_t)p.flag_value) return;      // flag mode: the points another kernel handed over
    if (p.mx_preflag && p.mx_flags[gidx] == kMxRest) return;                         // flagged by mx_preflag_kernel: corridor wider than the tile
    auto hand_on = [&](uint8_t to) __attribute__((always_inline)) {
        if (tid == 0) p.mx_flags[gidx] = to;
    };

    const unsigned char *chip_pl = p.swap ? p.p1 : p.p0;
    const unsigned char *win_pl = p.swap ? p.p0 : p.p1;
    const int Wp = p.Wp, PAD = p.pad;

    // ---- point header (as match_px_kernel.hip) ---------------------------------------------------------------------
    const double *row = p.xyuvav + (size_t)p.xy_stride * (size_t)gidx + p.xy_col;
    const int u0 = (int)row[0], v0 = (int)row[1];
    const int64_t pbeg = p.piv_off[gidx];
    const int npiv = (int)(p.piv_off[gidx + 1] - pbeg);
    const int32_t *pv_g = p.piv_uv + 2 * pbeg;
    // Everything the header needs from memory beyond (u, v) and the pivot range is issued together -- the last pivot, the twelve table
    // corners of the three chip-side queries (one per lane), the chip's corner pixel, this lane's pivot -- and only then waited for:
    // one memory round trip instead of one per query (the values are wave-uniform, but the kernel stores to global memory, so the
    // compiler loads them through the vector path and every readfirstlane is a wait).
    const int cu0 = u0 - OCW + PAD, cv0 = v0 - OCW + PAD;                       // plane position of chip pixel (0, 0)
    typedef unsigned long long SatT;
    const SatT *sat_chip = reinterpret_cast<const SatT *>(p.swap ? p.sat1 : p.sat0);
    const SatT *sat_win = reinterpret_cast<const SatT *>(p.swap ? p.sat0 : p.sat1);
    const int2 lastpv = *reinterpret_cast<const int2 *>(pv_g + 2 * (npiv - 1));
    SatT satv = 0;
    {   // lane 4 q + c: corner c of query q (0: the chip, 1: its last column, 2: its last row)
        const int q = (lane >> 2) & 3, c = lane & 3;
        const int bx = cu0 + (q == 1 ? CW - 1 : 0), by = cv0 + (q == 2 ? CW - 1 : 0), bw = q == 1 ? 1 : CW, bh = q == 2 ? 1 : CW;
        if (lane < 12) satv = sat_chip[(size_t)(by + ((c & 2) ? bh : 0)) * p.sat_ws + bx + ((c & 1) ? bw : 0)];
    }
    const uint32_t cornerv = chip_pl[(size_t)(cv0 + CW - 1) * Wp + cu0 + CW - 1];
    // the pivots (lane k of wave 0 = pivot k in the climbs): parked in LDS until the surface is there
    int2 pv_mine = make_int2(0, 0);
    if (wave == 0 && lane < npiv && npiv <= 64) pv_mine = *reinterpret_cast<const int2 *>(pv_g + 2 * lane);
    // ... and the first batch of the chip's pixels (aligned dwords of its rows; written to LDS once the tile is on its way)
    constexpr int CD = C::CD, CTASK = CW * CD, CNIT = (CTASK + NT - 1) / NT, CKB = CNIT < 5 ? CNIT : 5;
    const int csh = cu0 & 3;
    const uint32_t *cgb = reinterpret_cast<const uint32_t *>(chip_pl + (size_t)cv0 * Wp + (cu0 - csh));
    uint32_t clo[CKB], chi[CKB];
    auto chip_loads = [&](int it0) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < CKB; k++) {
            const int t = tid + NT * (it0 + k);
            const int r = t / CD, j = t - CD * r;
            const bool on = (it0 + k < CNIT) && t < CTASK;
            const uint32_t go = (uint32_t)(on ? r : 0) * (uint32_t)(Wp >> 2) + (uint32_t)(on ? j : 0);      // (32-bit offsets from a uniform base)
            clo[k] = cgb[go]; chi[k] = cgb[go + 1u];
        }
    };
    chip_loads(0);
    const int lu = __builtin_amdgcn_readfirstlane(lastpv.x), lv = __builtin_amdgcn_readfirstlane(lastpv.y);
    const int dx2 = (lu < 0 ? -lu : lu) + OCW + 2, dy2 = (lv < 0 ? -lv : lv) + OCW + 2;
    const int Dx2 = 2 * dx2 + 1, Dy2 = 2 * dy2 + 1;
    const int csx = Dx2 - 2 * OCW + 1, csy = Dy2 - 2 * OCW + 1;          // compact cells; a climb touches [1, cs - 2]
    const int wu0 = u0 + p.off_u - dx2 + PAD, wv0 = v0 + p.off_v - dy2 + PAD;   // plane position of window pixel (0, 0)
    // ---- what this kernel takes -----------------------------------------------------------------------------------
    // the tile: all reachable cells if they fit, else centred on the pivots' starts (a scan that leaves it hands the point on)
    int tx0 = 1, ty0 = 1;
    bool fits = true;
    {
        const int c0x = dx2 - OCW, c1x = c0x + lu, c0y = dy2 - OCW, c1y = c0y + lv;
        const int lox = min(c0x, c1x), hix = max(c0x, c1x), loy = min(c0y, c1y), hiy = max(c0y, c1y);
        if (csx - 2 > 32) { tx0 = min(max((lox + hix) / 2 - 15, 1), csx - 2 - 31); fits = fits && lox - 1 >= tx0 && hix + 1 <= tx0 + 31; }
        if (csy - 2 > 32) { ty0 = min(max((loy + hiy) / 2 - 15, 1), csy - 2 - 31); fits = fits && loy - 1 >= ty0 && hiy + 1 <= ty0 + 31; }
    }
    // the null count of the window's written area (:869-886) and the first batch of the tile's pixels: issued now, read below
    const int win_nulls_v = sat_nulls_u8(sat_win, p.sat_ws, wu0, wv0, 2 * dx2, 2 * dy2, lane);      // (exact for any window size)
    constexpr int NSEG = C::SW / 16, TTASK = C::KW * NSEG;           // (tile rows >= KW and columns >= SW are only ever weighted 0: left as they are)
    constexpr int TNIT = (TTASK + NT - 1) / NT, TKB = TNIT < 4 ? TNIT : 4;
    const int tsh = (wu0 + tx0) & 3;
    const uint32_t *tgb = reinterpret_cast<const uint32_t *>(win_pl + (size_t)(wv0 + ty0) * Wp + (wu0 + tx0 - tsh));
    uint32_t td[TKB][5];
    auto tile_loads = [&](int it0) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < TKB; k++) {
            const int t = tid + NT * (it0 + k);
            const int y = t / NSEG, q = t - NSEG * y;
            const bool on = (it0 + k < TNIT) && t < TTASK;
            const uint32_t go = (uint32_t)(on ? y : 0) * (uint32_t)(Wp >> 2) + 4u * (uint32_t)(on ? q : 0);
#pragma unroll
            for (int j = 0; j < 5; j++) td[k][j] = tgb[go + (uint32_t)j];
        }
    };
    tile_loads(0);
    // the chip's sums and null count, and -- for the closed-form T4 terms -- those of its last column and last row
    auto lane64 = [&](int l) __attribute__((always_inline)) -> SatT {
        const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)satv, l), hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(satv >> 32), l);
        return ((SatT)hi << 32) | lo;
    };
    const SatT chipQ = lane64(3) - lane64(1) - lane64(2) + lane64(0);
    const SatT colQ = lane64(7) - lane64(5) - lane64(6) + lane64(4);
    const SatT rowQ = lane64(11) - lane64(9) - lane64(10) + lane64(8);
    const uint32_t corner = (uint32_t)__builtin_amdgcn_readfirstlane((int)cornerv);

    const int chip_nulls = (int)(chipQ >> kSatNullShift8);
    const int win_nulls = __builtin_amdgcn_readfirstlane(win_nulls_v);
    if (npiv > 64 || !fits) { hand_on(kMxRest); return; }
    if (!GEN && (win_nulls != 0 || chip_nulls != 0)) {
        hand_on((chip_nulls == 0 && p.mx_wn_on) ? kMxWn : (p.mx_gen_on ? kMxNulls : kMxRest));
        return;
    }
    // general form: wn = the written area of the window holds nulls (then the never-written last row / column are nulls like any other,
    // else they are applied in closed form as in the clean form); cn = the chip holds nulls
    const bool wn = GEN && win_nulls != 0, cn = C::CN && chip_nulls != 0;

    // ---- validity (a6, :605-644): nulls of the chip / of the whole Dy2 x Dx2 search area (its last row and column are never written: zeros)
    {
        const float max_ratio = 0.8f;
        const float rc = (float)chip_nulls / (float)NPX;
        const float rw = (float)(win_nulls + Dx2 + Dy2 - 1) / (float)(Dx2 * Dy2);
        if (rc > max_ratio || rw > max_ratio) {
            if (tid == 0) {
                const float nanv = __builtin_nanf("");
                p.out[3 * (size_t)gidx + 0] = nanv; p.out[3 * (size_t)gidx + 1] = nanv; p.out[3 * (size_t)gidx + 2] = -3.0f;
            }
            return;
        }
    }
    const uint32_t SX = (uint32_t)chipQ & ((1u << kSatSqShift8) - 1u);
    const uint32_t SXX = (uint32_t)(chipQ >> kSatSqShift8) & ((1u << (kSatNullShift8 - kSatSqShift8)) - 1u);

    // ---- stage the tile: window pixels (tx0 + x, ty0 + y), x < SW, y < KW, as signed bytes b - 128 (both waves) ---------------
    {
#pragma unroll 1
        for (int it0 = 0; it0 < TNIT; it0 += TKB) {
            if (it0 > 0) tile_loads(it0);
#pragma unroll
            for (int k = 0; k < TKB; k++) {
                const int t = tid + NT * (it0 + k);
                const int y = t / NSEG, q = t - NSEG * y;
                if ((it0 + k < TNIT) && t < TTASK) {
                    uint4 w;
                    w.x = alignb(td[k][1], td[k][0], tsh) ^ 0x80808080u;
                    w.y = alignb(td[k][2], td[k][1], tsh) ^ 0x80808080u;
                    w.z = alignb(td[k][3], td[k][2], tsh) ^ 0x80808080u;
                    w.w = alignb(td[k][4], td[k][3], tsh) ^ 0x80808080u;
                    *reinterpret_cast<uint4 *>(WT + y * PW + 16 * q) = w;
                    if (C::ZPL && wn) *reinterpret_cast<uint4 *>(smem + C::OFF_Z + y * PW + 16 * q) = make_uint4(null80(w.x), null80(w.y), null80(w.z), null80(w.w));
                }
            }
        }
    }
    // ---- chip planes: zeros, then CW rows of (a ^ 0x80) [and of the two byte planes of a^2] -------------------------------
    {
        const uint4 z4 = make_uint4(0, 0, 0, 0);
        for (int i = tid; i < (C::NPL * C::CHB) / 16; i += NT) reinterpret_cast<uint4 *>(CH)[i] = z4;
        if (wave == 0) reinterpret_cast<int2 *>(smem + C::OFF_PIV)[lane] = pv_mine;
        if (GEN && tid < 4) reinterpret_cast<uint32_t *>(smem + C::OFF_RM)[tid] = 0u;
    }
    __syncthreads();
    {
        // T4: the search area's last column and last row are never written (:869-886): nulls.  (What lies beyond them inside the tile
        // only reaches cells no climb can touch.)
        const int zx = Dx2 - 1 - tx0, zy = Dy2 - 1 - ty0;
        if (zx < C::SW) for (int y = tid; y < C::KW; y += NT) { WT[y * PW + zx] = 0x80; if (C::ZPL && wn) smem[C::OFF_Z + y * PW + zx] = 0x80; }
        if (zy < C::KW) for (int x = tid; x < C::SW / 4; x += NT) {
            *reinterpret_cast<uint32_t *>(WT + zy * PW + 4 * x) = 0x80808080u;
            if (C::ZPL && wn) *reinterpret_cast<uint32_t *>(smem + C::OFF_Z + zy * PW + 4 * x) = 0x80808080u;
        }
        constexpr uint32_t LASTM = C::LASTN >= 4 ? 0xffffffffu : ((1u << (8 * C::LASTN)) - 1u);
#pragma unroll 1
        for (int it0 = 0; it0 < CNIT; it0 += CKB) {
            if (it0 > 0) chip_loads(it0);
#pragma unroll
            for (int k = 0; k < CKB; k++) {
                const int t = tid + NT * (it0 + k);
                const int r = t / CD, j = t - CD * r;
                if ((it0 + k < CNIT) && t < CTASK) {
                    const uint32_t a = alignb(chi[k], clo[k], csh);
                    const uint32_t m = (j == CD - 1) ? LASTM : 0xffffffffu;
                    *reinterpret_cast<uint32_t *>(CH + CH0 + CP * r + 4 * j) = (a ^ 0x80808080u) & m;
                    if (GEN && wn) {                          // the byte planes of a^2
                        uint32_t l, g;
                        squares_dword(a, l, g);
                        *reinterpret_cast<uint32_t *>(CH + C::CHB + CH0 + CP * r + 4 * j) = (l ^ 0x80808080u) & m;
                        *reinterpret_cast<uint32_t *>(CH + 2 * C::CHB + CH0 + CP * r + 4 * j) = (g ^ 0x80808080u) & m;
                    }
                    if (GEN && cn) {                          // a null pixel (DN 0) among the dword's chip pixels: flag the chip row
                        const uint32_t nz = (((a & 0x7f7f7f7fu) + 0x7f7f7f7fu) | a) & 0x80808080u & m;
                        if (nz != (0x80808080u & m)) atomicOr(reinterpret_cast<uint32_t *>(smem + C::OFF_RM) + (r >> 5), 1u << (r & 31));
                    }
                }
            }
        }
    }
    __syncthreads();
    MIMC3_MX_STAMP(0)

    // lane (n, h) of wave w: cell column s = 16 w + n; byte i of K chunk c <-> window column 16 w + 64 c + 16 h + i
    const int n = lane & 15, h = lane >> 4;
    const unsigned char *arow = WT + n * PW + 16 * wave + 16 * h;       // A operand: tile row r + 16 mt + n
    const int boff = CH0 + 16 * h - n;                                   // B operand: chip-row bytes 64 c + 16 h - n + i
    const uint32_t bsh = (uint32_t)boff & 3u;
    const unsigned char *brow = CH + (boff & ~3);

    // ---- the product surface (and the null corrections) ------------------------------------------------------------------
    //      Software-pipelined by hand: the LDS reads of chip row r + 1 are in flight while row r's MFMAs issue (left to the compiler,
    //      every row waited for its own reads: one LDS latency per row).  No branch inside the loop: with conditional reads the
    //      compiler's wait counts assume the shorter queue; an odd chip is followed by an all-zero row, whose products add nothing.
    const v4i zero4 = {0, 0, 0, 0};
    // accumulators (two 16 x 16 tiles each): sum a'b'; general form, window nulls zb = [b == 0]: sum zb a', sum zb (a^2 & 255)', sum zb (a^2 >> 8)';
    // chip nulls za = [a == 0]: sum b' za, sum (b^2 & 255)' za, sum (b^2 >> 8)' za, and sum zb za
    v4i acc[2] = {zero4, zero4}, accz[2] = {zero4, zero4}, accl[2] = {zero4, zero4}, acch[2] = {zero4, zero4};
    v4i accy[2] = {zero4, zero4}, accyl[2] = {zero4, zero4}, accyh[2] = {zero4, zero4}, acczz[2] = {zero4, zero4};
    (void)accz; (void)accl; (void)acch; (void)accy; (void)accyl; (void)accyh; (void)acczz;
    {
        // general form: which chip rows hold nulls (their Toeplitz rows are the only ones the chip-null correlations need)
        uint32_t rm0 = 0, rm1 = 0, rm2 = 0;
        if (GEN && cn) {
            const uint32_t *rmp = reinterpret_cast<const uint32_t *>(smem + C::OFF_RM);
            rm0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)rmp[0]); rm1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)rmp[1]);
            rm2 = (uint32_t)__builtin_amdgcn_readfirstlane((int)rmp[2]);
        }
        auto row_has_null = [&](int r) __attribute__((always_inline)) -> bool {
            const uint32_t w = r < 32 ? rm0 : (r < 64 ? rm1 : rm2);
            return ((w >> (r & 31)) & 1u) != 0u;
        };
        const unsigned char *zrow = smem + C::OFF_Z + (arow - WT);         // the null plane under the lane's A operand (ZPL)
        (void)zrow;
        // (double-buffered across rows: the tile rows and the a' plane.  The general form's other operands -- the a^2 planes, the null plane --
        //  are read at the start of a row's turn and used at its end, behind the MFMAs that do not need them: half the operand registers)
        struct RowOps { v4i a[2][KCW]; uint32_t d[KCW][5]; };
        auto issue = [&](int r, RowOps &o) __attribute__((always_inline)) {
#pragma unroll
            for (int m = 0; m < 2; m++)
#pragma unroll
                for (int c = 0; c < KCW; c++) o.a[m][c] = *reinterpret_cast<const v4i *>(arow + (r + 16 * m) * PW + 64 * c);
#pragma unroll
            for (int c = 0; c < KCW; c++) {
                const uint32_t *q = reinterpret_cast<const uint32_t *>(brow + CP * r + 64 * c);
#pragma unroll
                for (int k = 0; k < 5; k++) o.d[c][k] = q[k];
            }
        };
        auto shifted = [&](const uint32_t (&d)[KCW][5], v4i (&b)[KCW]) __attribute__((always_inline)) {
#pragma unroll
            for (int c = 0; c < KCW; c++)
#pragma unroll
                for (int k = 0; k < 4; k++) b[c][k] = (int)alignb(d[c][k + 1], d[c][k], bsh);
        };
        // (the null-plane operands are 0x80 = -128 per null: their correlations come out times -128, their product times 16384)
        auto consume = [&](const RowOps &o, int r) __attribute__((always_inline)) {
            [[maybe_unused]] uint32_t d1[KCW][5], d2[KCW][5];
            [[maybe_unused]] v4i az[2][KCW];
            if constexpr (GEN) {
#pragma unroll
                for (int c = 0; c < KCW; c++) {
                    const uint32_t *q1 = reinterpret_cast<const uint32_t *>(brow + C::CHB + CP * r + 64 * c), *q2 = reinterpret_cast<const uint32_t *>(brow + 2 * C::CHB + CP * r + 64 * c);
#pragma unroll
                    for (int k = 0; k < 5; k++) { d1[c][k] = q1[k]; d2[c][k] = q2[k]; }
                }
                if constexpr (C::ZPL) {
#pragma unroll
                    for (int m = 0; m < 2; m++)
#pragma unroll
                        for (int c = 0; c < KCW; c++) az[m][c] = *reinterpret_cast<const v4i *>(zrow + (r + 16 * m) * PW + 64 * c);
                }
            }
            v4i b[KCW];
            shifted(o.d, b);
#pragma unroll
            for (int m = 0; m < 2; m++)
#pragma unroll
                for (int c = 0; c < KCW; c++) acc[m] = mfma(o.a[m][c], b[c], acc[m]);
            if constexpr (GEN) {
                const bool cnrow = cn && row_has_null(r);
                v4i z[2][KCW];
                if (wn) {
#pragma unroll
                    for (int m = 0; m < 2; m++)
#pragma unroll
                        for (int c = 0; c < KCW; c++) {
                            if constexpr (C::ZPL) z[m][c] = az[m][c]; else z[m][c] = null80v(o.a[m][c]);
                            accz[m] = mfma(z[m][c], b[c], accz[m]);
                        }
                }
                if (cnrow) {
#pragma unroll
                    for (int c = 0; c < KCW; c++) {
                        const v4i za = null80v(b[c]);              // (a padding byte is 0x00, a null chip pixel 0x80)
#pragma unroll
                        for (int m = 0; m < 2; m++) {
                            v4i lo, hi;
                            squares(o.a[m][c], lo, hi);
                            accy[m] = mfma(o.a[m][c], za, accy[m]); accyl[m] = mfma(lo, za, accyl[m]); accyh[m] = mfma(hi, za, accyh[m]);
                            if (wn) acczz[m] = mfma(z[m][c], za, acczz[m]);
                        }
                    }
                }
                if (wn) {
                    v4i bq[KCW];
                    shifted(d1, bq);
#pragma unroll
                    for (int m = 0; m < 2; m++)
#pragma unroll
                        for (int c = 0; c < KCW; c++) accl[m] = mfma(z[m][c], bq[c], accl[m]);
                    shifted(d2, bq);
#pragma unroll
                    for (int m = 0; m < 2; m++)
#pragma unroll
                        for (int c = 0; c < KCW; c++) acch[m] = mfma(z[m][c], bq[c], acch[m]);
                }
            }
        };
        RowOps o0, o1;
        issue(0, o0);
#pragma unroll 1
        for (int r = 0; r < C::CWE; r += 2) {
            issue(r + 1, o1);
            __builtin_amdgcn_sched_barrier(0);
            consume(o0, r);
            __builtin_amdgcn_sched_barrier(0);
            issue(r + 2 < C::CWE ? r + 2 : r + 1, o0);          // (the last trip re-reads a row: nothing consumes it)
            __builtin_amdgcn_sched_barrier(0);
            consume(o1, r + 1);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if constexpr (GEN) {        // the two byte planes of each squared operand back into one sum (and out of the -128 scale): 16 registers fewer from here on
#pragma unroll
        for (int m = 0; m < 2; m++)
#pragma unroll
            for (int i = 0; i < 4; i++) {
                accl[m][i] = ((-accl[m][i]) >> 7) + 256 * ((-acch[m][i]) >> 7);
                accyl[m][i] = ((-accyl[m][i]) >> 7) + 256 * ((-accyh[m][i]) >> 7);
            }
    }
    MIMC3_MX_STAMP(1)

    // ---- window-side box sums of the wave's 32 x 16 cells: sum b, sum b^2 (and the null count) ------------------------------------
    //      row-box sums R[y][s] = sum_{x = s}^{s + CW - 1} q[y][x] by MFMA (A = 16 tile rows, B = the band of ones), then
    //      Box[dy][s] = sum_{y = dy}^{dy + CW - 1} R[y][s] by MFMA (A = the band of ones, B = the byte planes of R: the four registers
    //      of a 16-row accumulator tile are one dword of the B operand -- the accumulator layout has R's rows where B has its K index)
    v4i boxb[2], boxq[2], boxz[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    {
        const Bands<CW, KCW, KCV> &bd = kBands<CW, KCW, KCV>;
        v4i hb[KCW];
#pragma unroll
        for (int c = 0; c < KCW; c++) hb[c] = *reinterpret_cast<const v4i *>(&bd.hb[c][lane][0]);
        auto tile_a = [&](int t, v4i (&a)[KCW]) __attribute__((always_inline)) {
#pragma unroll
            for (int c = 0; c < KCW; c++) a[c] = *reinterpret_cast<const v4i *>(arow + 16 * t * PW + 64 * c);
        };
        {   // sum b (and the null count)
            v4i p0[KCV], p1[KCV], pz[KCV];
#pragma unroll
            for (int t = 0; t < 4 * KCV; t++) {
                uint32_t q0 = 0, q1 = 0, qz = 0;
                if (t < NTR) {
                    v4i a[KCW];
                    tile_a(t, a);
                    v4i R = zero4;
#pragma unroll
                    for (int c = 0; c < KCW; c++) R = mfma(a[c], hb[c], R);
#pragma unroll
                    for (int i = 0; i < 4; i++) R[i] += 128 * CW;
                    planes2(R, q0, q1);
                    if (GEN && wn) {                          // the row counts of the nulls (the null plane is -128 per null)
                        v4i Z = zero4;
#pragma unroll
                        for (int c = 0; c < KCW; c++) {
                            v4i zc;
                            if constexpr (C::ZPL) zc = *reinterpret_cast<const v4i *>(smem + C::OFF_Z + (arow - WT) + 16 * t * PW + 64 * c);
                            else zc = null80v(a[c]);
                            Z = mfma(zc, hb[c], Z);
                        }
#pragma unroll
                        for (int i = 0; i < 4; i++) Z[i] = (-Z[i]) >> 7;
                        qz = planes1(Z);
                    }
                }
                p0[t >> 2][t & 3] = (int)q0; p1[t >> 2][t & 3] = (int)q1; pz[t >> 2][t & 3] = (int)qz;
            }
#pragma unroll
            for (int m = 0; m < 2; m++) {
                v4i v0 = zero4, v1 = zero4;
#pragma unroll
                for (int c = 0; c < KCV; c++) {
                    const v4i vb = *reinterpret_cast<const v4i *>(&bd.vb[m][c][lane][0]);
                    v0 = mfma(vb, p0[c], v0); v1 = mfma(vb, p1[c], v1);
                    if (GEN && wn) boxz[m] = mfma(vb, pz[c], boxz[m]);
                }
#pragma unroll
                for (int i = 0; i < 4; i++) boxb[m][i] = (v0[i] + 128 * CW) + 256 * (v1[i] + 128 * CW);
            }
        }
        {   // sum b^2
            v4i p0[KCV], p1[KCV], p2[KCV];
#pragma unroll
            for (int t = 0; t < 4 * KCV; t++) {
                uint32_t q0 = 0, q1 = 0, q2 = 0;
                if (t < NTR) {
                    v4i a[KCW];
                    tile_a(t, a);
                    v4i Rl = zero4, Rh = zero4;
#pragma unroll
                    for (int c = 0; c < KCW; c++) {
                        v4i lo, hi;
                        squares(a[c], lo, hi);
                        Rl = mfma(lo, hb[c], Rl); Rh = mfma(hi, hb[c], Rh);
                    }
#pragma unroll
                    for (int i = 0; i < 4; i++) Rl[i] = (Rl[i] + 128 * CW) + 256 * (Rh[i] + 128 * CW);
                    planes3(Rl, q0, q1, q2);
                }
                p0[t >> 2][t & 3] = (int)q0; p1[t >> 2][t & 3] = (int)q1; p2[t >> 2][t & 3] = (int)q2;
            }
#pragma unroll
            for (int m = 0; m < 2; m++) {
                v4i v0 = zero4, v1 = zero4, v2 = zero4;
#pragma unroll
                for (int c = 0; c < KCV; c++) {
                    const v4i vb = *reinterpret_cast<const v4i *>(&bd.vb[m][c][lane][0]);
                    v0 = mfma(vb, p0[c], v0); v1 = mfma(vb, p1[c], v1); v2 = mfma(vb, p2[c], v2);
                }
#pragma unroll
                for (int i = 0; i < 4; i++) boxq[m][i] = (v0[i] + 128 * CW) + 256 * (v1[i] + 128 * CW) + 65536 * (v2[i] + 128 * CW);
            }
        }
    }

    MIMC3_MX_STAMP(2)
    __syncthreads();                                        // the tile's bytes become the NCC surface

    // ---- NCC of the 8 cells of this lane (:734): exact integer sums, the f64 formula rounded to f32 ------------------------------
    //   reference:  (float)( num / sqrt(P) ),  num = n sxy - sx sy  and  va = n sxx - sx^2,  vb = n syy - sy^2  exact integers in f64,
    //               P = va * vb rounded once, sqrt and the division correctly rounded: the f64 quotient Qd is within 2^-51 of num / sqrt(P).
    //   here:       r = v_rsq_f64(P) refined by one Newton step (relative error 2^-47.8 measured, tools/probes/mx_finish.hip; the
    //               instruction alone 2^-24.2), Q' = num * r'.  Q' and Qd round to the SAME f32 unless an f32 rounding boundary lies
    //               between them: a cell whose Q' is within 2^13 f64 ulps (2^-40 relative) of a boundary, or whose P is not positive (the
    //               reference's inf / NaN cases), is redone with the reference's own operations.  2^-15 of the cells; over 2^34 random
    //               cells the farthest one whose two results differed lay 7 ulps from its boundary.
    {
        uint32_t tix = threadIdx.x;
        asm volatile("" : "+v"(tix));                        // (formed again from the thread index: carried over from the header, it was the one register that spilled)
        const int sx_col = (int)(tix >> 6) * 16 + (int)(tix & 15u), cx = tx0 + sx_col;
        // clean form, T4 in closed form: a cell whose box reaches the never-written last column (row) loses the chip's last column (row)
        uint32_t cS = (uint32_t)colQ & ((1u << kSatSqShift8) - 1u), cSS = (uint32_t)(colQ >> kSatSqShift8) & ((1u << (kSatNullShift8 - kSatSqShift8)) - 1u);
        uint32_t rS = (uint32_t)rowQ & ((1u << kSatSqShift8) - 1u), rSS = (uint32_t)(rowQ >> kSatSqShift8) & ((1u << (kSatNullShift8 - kSatSqShift8)) - 1u);
        uint32_t SXo = SX, SXXo = SXX;
        // (opaque here: left alone, the compiler forms the f64 constants below in the header and carries twelve registers through every phase)
        asm volatile("" : "+s"(cS), "+s"(cSS), "+s"(rS), "+s"(rSS), "+s"(SXo), "+s"(SXXo));
        // T4 in closed form unless the window's own nulls are correlated anyway (wn): the never-written column / row take the chip's
        // last column / row out of n, sx, sxx -- its non-null pixels, with a chip that holds nulls
        const bool colT4 = !wn && cx == csx - 2;
        const int rT4 = wn ? -1 : csy - 2 - ty0;            // tile row of the cells that reach the never-written last row
        const int Na = chip_nulls;
        const int cN = (int)(colQ >> kSatNullShift8), rN = (int)(rowQ >> kSatNullShift8), kN = corner == 0u ? 1 : 0;   // nulls of the chip's last column / row / corner pixel
        // (n, sx, sxx) of this lane's cells before the window's own nulls: off / on the T4 row
        const int n0 = NPX - Na - (colT4 ? CW - cN : 0), sx0 = (int)SXo - (colT4 ? (int)cS : 0), sxx0 = (int)SXXo - (colT4 ? (int)cSS : 0);
        const int n1 = n0 - (CW - rN) + (colT4 ? 1 - kN : 0), sx1 = sx0 - (int)rS + (colT4 ? (int)corner : 0), sxx1 = sxx0 - (int)rSS + (colT4 ? (int)(corner * corner) : 0);
        double dn0 = 0, dsx0 = 0, va0 = 0, dn1 = 0, dsx1 = 0, va1 = 0;
        if (!wn) {
            dn0 = (double)n0; dsx0 = (double)sx0; va0 = dn0 * (double)sxx0 - dsx0 * dsx0;
            dn1 = (double)n1; dsx1 = (double)sx1; va1 = dn1 * (double)sxx1 - dsx1 * dsx1;
        }
        auto cell_in = [&](int m, int i, double &dn, double &dsx, double &va, int &sxy, int &sy, int &syy) __attribute__((always_inline)) {
            const int ry = 16 * m + 4 * h + i;
            sy = boxb[m][i]; syy = boxq[m][i];
            sxy = acc[m][i] + 128 * ((int)SXo + sy) - 16384 * NPX;          // (over all chip positions: a null is a zero factor)
            if (GEN && cn) {                                   // chip nulls take window pixels out of sy, syy
                sy -= ((-accy[m][i]) >> 7) + 128 * Na;
                syy -= accyl[m][i] + (128 + 256 * 128) * Na;
            }
            if (GEN && wn) {                                   // window nulls (T4 among them) take chip pixels out of n, sx, sxx
                const int nz = boxz[m][i];
                const int ca = ((-accz[m][i]) >> 7) + 128 * nz, caa = accl[m][i] + (128 + 256 * 128) * nz;
                dn = (double)(NPX - Na - nz + (cn ? (acczz[m][i] >> 14) : 0)); dsx = (double)((int)SXo - ca);
                va = dn * (double)((int)SXXo - caa) - dsx * dsx;
            } else {
                const bool rowT4 = ry == rT4;
                dn = rowT4 ? dn1 : dn0; dsx = rowT4 ? dsx1 : dsx0; va = rowT4 ? va1 : va0;
            }
        };
        uint32_t amb = 0u;
#pragma unroll
        for (int ci = 0; ci < 8; ci++) {
            const int m = ci >> 2, i = ci & 3, ry = 16 * m + 4 * h + i;
            double dn, dsx, va; int sxy, sy, syy;
            cell_in(m, i, dn, dsx, va, sxy, sy, syy);
            const double dsy = (double)sy;
            const double num = dn * (double)sxy - dsx * dsy;
            const double P = va * (dn * (double)syy - dsy * dsy);
            const double r = __builtin_amdgcn_rsq(P);
            const double g = P * r;
            const double e2 = __builtin_fma(-r, g, 1.0);
            const double r1 = __builtin_fma(0.5 * r, e2, r);
            const double q = num * r1;
            const uint32_t low = ((uint32_t)__double2loint(q) & 0x1fffffffu) - (0x10000000u - 0x2000u);     // distance to the f32 rounding boundary, + 2^13
            if (!(P > 0.0) || low <= 0x4000u) amb |= 1u << ci;
            val[ry * VP + sx_col] = (float)q;
            if constexpr (GEN) __builtin_amdgcn_sched_barrier(0);      // (general form: one cell at a time -- interleaved, the cells' f64 temporaries spill)
        }
        if (__any(amb != 0u)) {                            // rare (2^-15 of the cells): the reference's own operations
#pragma unroll
            for (int ci = 0; ci < 8; ci++) {
                if (!__any((amb >> ci) & 1u)) continue;
                const int m = ci >> 2, i = ci & 3, ry = 16 * m + 4 * h + i;
                double dn, dsx, va; int sxy, sy, syy;
                cell_in(m, i, dn, dsx, va, sxy, sy, syy);
                const double dsy = (double)sy;
                const double num = dn * (double)sxy - dsx * dsy;
                const double den = sqrt(va * (dn * (double)syy - dsy * dsy));
                if ((amb >> ci) & 1u) val[ry * VP + sx_col] = (float)(num / den);
            }
        }
    }
    __syncthreads();
    MIMC3_MX_STAMP(3)
    if (wave != 0) return;                                  // the sequential part needs one wave (lane k = pivot k)

    // ---- the climbs (lane k = pivot k) on the complete surface; trajectories as 4-bit codes per scan -------------------------
    auto inside = [&](int pu, int pvv) __attribute__((always_inline)) -> bool {     // the reference's boundary test (:703), true = scan allowed
        return !(pu - OCW <= 1 || pu + OCW >= Dx2 - 1 || pvv - OCW <= 1 || pvv + OCW >= Dy2 - 1);
    };
    // tile-relative cell of a scan centre
    auto relx = [&](int pu) __attribute__((always_inline)) -> int { return pu - OCW - tx0; };
    auto rely = [&](int pvv) __attribute__((always_inline)) -> int { return pvv - OCW - ty0; };
    constexpr int kSpecRounds = 16;
    int su, sv;
    {
        const int2 pv = reinterpret_cast<const int2 *>(smem + C::OFF_PIV)[lane];
        su = pv.x + dx2; sv = pv.y + dy2;
    }
    const int start_u = su, start_v = sv;
    // Lane k walks pivot k's climb on the complete surface, ignoring the visited state (which can only END a real climb earlier).  A scan's
    // move and running maximum depend on NCC values alone, with the reference's compare sequence (:736-741): recorded as move deltas, 4 bits
    // per scan ((du + 1) | (dv + 1) << 2; 5 = the scan did not move), plus one bit per scan "it raised the maximum".  Straight-line,
    // predicated: one wave-uniform loop, no divergent control flow (the compiler's exec-mask bookkeeping for a per-lane `while` was half
    // the loop's instructions).
    bool alive = lane < npiv && inside(su, sv);
    bool left = false;                                       // a scan would leave the tile
    uint32_t dlo = 0u, dhi = 0u, updm = 0u;
    int nsc = 0;
    {
        float smax = -2.0f;
#pragma unroll 1
        for (int t = 0; t < kSpecRounds; t++) {
            if (!__any(alive)) break;
            const int rx = relx(su), ry = rely(sv);
            const bool in = (unsigned)(rx - 1) <= 29u && (unsigned)(ry - 1) <= 29u;
            left = left || (alive && !in);
            const bool act = alive && in;
            const float *vp = val + (act ? ry : 1) * VP + (act ? rx : 1);          // (idle lanes read a harmless cell)
            float v[9];
#pragma unroll
            for (int j = 0; j < 9; j++) v[j] = vp[(j % 3 - 1) * VP + (j / 3 - 1)];
            const float m = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(v[0], v[1]), v[2]), __builtin_fmaxf(__builtin_fmaxf(v[3], v[4]), v[5])),
                                            __builtin_fmaxf(__builtin_fmaxf(v[6], v[7]), v[8]));
            uint32_t mv = 8u;
#pragma unroll
            for (int j = 7; j >= 0; j--) mv = (v[j] == m) ? (uint32_t)j : mv;
            const bool up = act && (m > smax);
            smax = up ? m : smax;
            const bool moved = up && mv != 4u;
            const uint32_t q3 = (mv * 11u) >> 5;                                        // mv / 3 for 0..8
            const uint32_t d = moved ? (q3 | ((mv - 3u * q3) << 2)) : 5u;
            su += (int)(d & 3u) - 1; sv += (int)(d >> 2) - 1;
            const uint32_t dsh = act ? d << (4 * (t & 7)) : 0u;
            if (t < 8) dlo |= dsh; else dhi |= dsh;
            updm |= (up ? 1u : 0u) << t;
            nsc += act ? 1 : 0;
            alive = moved && inside(su, sv);
        }
    }
    if (__any(left)) { hand_on(kMxRest); return; }
    MIMC3_MX_STAMP(4)

    // ---- exact replay: the visited state only decides HOW MANY scans of a pivot really happen (newncc != 0, :699) -- sequential over
    //      pivots; lane r holds the visited bits of tile row r (match_px_kernel.hip, "exact replay", with 32-bit rows)
    int T = 0;
    uint32_t vrow = 0u;
    {
        const uint32_t head = (uint32_t)(relx(start_u) & 0xff) | ((uint32_t)(rely(start_v) & 0xff) << 8) | ((uint32_t)nsc << 16);
        for (int kk = 0; kk < npiv; kk++) {
            const uint32_t hd = (uint32_t)__builtin_amdgcn_readlane((int)head, kk);
            const int n_k = (int)(hd >> 16);
            int ccx = (int)(hd & 0xffu), cy1 = (int)((hd >> 8) & 0xffu) - 1;      // centre column, centre row - 1 (tile-relative)
            unsigned long long cur = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)dhi, kk) << 32) |
                                     (uint32_t)__builtin_amdgcn_readlane((int)dlo, kk);
            int t = 0;
            while (t < n_k) {
                // rows cy1 .. cy1 + 2 (= lanes) test and set their three bits; fresh3 = the lanes that held an unvisited one
                unsigned long long fresh3, sv2;
                uint32_t tmp;
                const uint32_t m3 = 7u << (ccx - 1);
                asm volatile("v_subrev_u32_e32 %[tmp], %[y], %[l1]\n\t"
                             "v_cmp_gt_u32_e32 vcc, 3, %[tmp]\n\t"
                             "s_and_saveexec_b64 %[sv], vcc\n\t"
                             "v_bitop3_b32 %[tmp], %[m], %[v], %[m] bitop3:0x30\n\t"      // m3 & ~vrow
                             "v_cmp_ne_u32_e32 vcc, 0, %[tmp]\n\t"
                             "v_or_b32_e32 %[v], %[m], %[v]\n\t"
                             "s_mov_b64 exec, %[sv]\n\t"
                             "s_mov_b64 %[fr], vcc"
                             : [tmp] "=&v"(tmp), [sv] "=&s"(sv2), [fr] "=s"(fresh3), [v] "+v"(vrow)
                             : [y] "s"(cy1), [l1] "v"(lane), [m] "s"(m3)
                             : "vcc", "scc");
                t++;
                const uint32_t d = (uint32_t)cur & 15u;
                cur >>= 4;
                ccx += (int)(d & 3u) - 1; cy1 += (int)(d >> 2) - 1;
                if (fresh3 == 0ull) break;
                if (d == 5u) break;
            }
            T = (lane == kk) ? t : T;
        }
    }
    // a pivot whose speculation was cut at 16 scans and whose real climb consumed all of them: the register-tiled kernel decides
    if (__any(alive && T == nsc)) { hand_on(kMxRest); return; }
    if (lane < 32) vis[lane] = vrow;
    MIMC3_MX_STAMP(5)

    // ---- best of pivots (:744-752): after its last updating scan a pivot sits on the arg-max cell
    int peak_u = dx2, peak_v = dy2;
    float best = -2.0f;
    {
        int fu = start_u, fv = start_v;
        for (int t = 0; t < T; t++) {
            const uint32_t d = ((t < 8 ? dlo : dhi) >> (4 * (t & 7))) & 15u;
            fu += (int)(d & 3u) - 1; fv += (int)(d >> 2) - 1;
        }
        const bool upd = (updm & ((T >= 32 ? 0u : (1u << T)) - 1u)) != 0u;      // some scan among the T performed raised the maximum
        const uint32_t fpos = ((uint32_t)fv << 16) | (uint32_t)fu;
        const float fmax = upd ? val[rely(fv) * VP + relx(fu)] : -2.0f;
        float bv = (lane < npiv) ? fmax : -__builtin_inff();
        int bi = lane;
        argmax_row16(bv, bi);
#pragma unroll
        for (int o = 16; o <= 32; o <<= 1) {
            const float ov = __shfl_xor(bv, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        bv = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(bv)));
        bi = __builtin_amdgcn_readfirstlane(bi);
        if (bv > -2.0f) {                                    // strict >, first pivot attaining the maximum wins
            const uint32_t pk = (uint32_t)__builtin_amdgcn_readlane((int)fpos, bi & 63);
            peak_u = (int)(pk & 0xffffu); peak_v = (int)(pk >> 16); best = bv;
        }
    }
    __builtin_amdgcn_wave_barrier();                          // (one wave: its LDS operations are performed in order)
    // ---- 3x3 quadratic fit (:757-788), the reference's arithmetic value by value (match_px_kernel.hip) ------------------------------
    if (lane < 5) {
        float n9[9];
        const int px = relx(peak_u), py = rely(peak_v);
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const int x = px - 1 + c, y = py - 1 + r;
                const bool in = (unsigned)x < 32u && (unsigned)y < 32u;                          // (a cell outside the tile was never scanned)
                n9[3 * r + c] = (in && ((vis[in ? y : 0] >> (x & 31)) & 1u)) ? val[y * VP + x] : -2.0f;
            }
        const float e0 = 6 * n9[0] - 12 * n9[1] + 6 * n9[2] + 6 * n9[3] - 12 * n9[4] + 6 * n9[5] + 6 * n9[6] - 12 * n9[7] + 6 * n9[8];
        const float e1 = 9 * n9[0] - 9 * n9[2] - 9 * n9[6] + 9 * n9[8];
        const float e2 = 6 * n9[0] + 6 * n9[1] + 6 * n9[2] - 12 * n9[3] - 12 * n9[4] - 12 * n9[5] + 6 * n9[6] + 6 * n9[7] + 6 * n9[8];
        const float e3 = -6 * n9[0] + 6 * n9[2] - 6 * n9[3] + 6 * n9[5] - 6 * n9[6] + 6 * n9[8];
        const float e4 = -6 * n9[0] - 6 * n9[1] - 6 * n9[2] + 6 * n9[6] + 6 * n9[7] + 6 * n9[8];
        double cp = (double)(lane == 0 ? e0 : (lane == 1 ? e1 : (lane == 2 ? e2 : (lane == 3 ? e3 : e4))));
        cp /= 36;
        auto from_lane = [&](int ln) __attribute__((always_inline)) -> double {
            const long long bits = __double_as_longlong(cp);
            const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)bits, ln), hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(bits >> 32), ln);
            return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
        };
        const double cp0 = from_lane(0), cp1 = from_lane(1), cp2 = from_lane(2), cp3 = from_lane(3), cp4 = from_lane(4);
        const float num = lane == 0 ? (float)(-2 * cp2 * cp3 + cp1 * cp4) : (float)(-2 * cp0 * cp4 + cp1 * cp3);
        const double det = 4 * cp0 * cp2 - cp1 * cp1;
        float o = (float)((double)num / det);
        o += (float)(lane == 0 ? peak_u - dx2 : peak_v - dy2);
        if (lane < 2) p.out[3 * (size_t)gidx + lane] = o;
        if (lane == 0) p.out[3 * (size_t)gidx + 2] = best;
    }
    MIMC3_MX_STAMP(6)
    MIMC3_MX_STATS_OUT
}

// Pre-pass for launches whose longest corridor exceeds the tile (one thread per grid point): flags the points the kernel below would
// hand on for that reason -- the same placement test as in its header -- so that they cost it one byte load instead of a header's
// two memory round trips (BASELINE C4, where every point is such a point: 161.1 -> 159.9 ms per pass).
__global__ __launch_bounds__(256) void mx_preflag_kernel(MatchU8Args p)
{
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (g >= p.N) return;
    const int64_t pbeg = p.piv_off[g];
    const int npiv = (int)(p.piv_off[g + 1] - pbeg);
    if (npiv < 1) return;
    const int OCW = p.ocw;
    const int lu = p.piv_uv[2 * (pbeg + npiv - 1)], lv = p.piv_uv[2 * (pbeg + npiv - 1) + 1];
    const int dx2 = (lu < 0 ? -lu : lu) + OCW + 2, dy2 = (lv < 0 ? -lv : lv) + OCW + 2;
    const int csx = 2 * dx2 + 1 - 2 * OCW + 1, csy = 2 * dy2 + 1 - 2 * OCW + 1;
    bool fits = true;
    const int c0x = dx2 - OCW, c1x = c0x + lu, c0y = dy2 - OCW, c1y = c0y + lv;
    const int lox = min(c0x, c1x), hix = max(c0x, c1x), loy = min(c0y, c1y), hiy = max(c0y, c1y);
    if (csx - 2 > 32) { const int tx0 = min(max((lox + hix) / 2 - 15, 1), csx - 2 - 31); fits = fits && lox - 1 >= tx0 && hix + 1 <= tx0 + 31; }
    if (csy - 2 > 32) { const int ty0 = min(max((loy + hiy) / 2 - 15, 1), csy - 2 - 31); fits = fits && loy - 1 >= ty0 && hiy + 1 <= ty0 + 31; }
    if (npiv > 64 || !fits) p.mx_flags[g] = kMxRest;
}

template <class C>
static hipError_t launch_one(MatchU8Args a, hipStream_t stream)
{
    const unsigned nb = (unsigned)((a.N + 7) & ~7);
    static unsigned long long *d_stats = nullptr;
    static const bool want_stats = getenv("MIMC3_MX_STATS") != nullptr;
    static size_t stats_n = 0;
    static std::mutex stats_mu;                      // diagnostics only
    std::unique_lock<std::mutex> stats_lock(stats_mu, std::defer_lock);
    if (want_stats) {
        stats_lock.lock();
        if (stats_n < (size_t)nb) {
            if (d_stats) (void)hipFree(d_stats);
            (void)hipMalloc(&d_stats, kStatW * sizeof(unsigned long long) * (size_t)nb);
            stats_n = nb;
        }
        (void)hipMemsetAsync(d_stats, 0, kStatW * sizeof(unsigned long long) * (size_t)nb, stream);
    }
    a.stats = want_stats ? d_stats : nullptr;
    hipLaunchKernelGGL(match_ncc_dlc_mx<C>, dim3(nb), dim3(C::NT), 0, stream, a);
    if (want_stats) {
        (void)hipStreamSynchronize(stream);
        unsigned long long *hh = (unsigned long long *)malloc(kStatW * sizeof(unsigned long long) * (size_t)nb);
        (void)hipMemcpy(hh, d_stats, kStatW * sizeof(unsigned long long) * (size_t)nb, hipMemcpyDeviceToHost);
        unsigned long long hsum[kStatW] = {0};
        size_t live = 0;
        for (size_t b = 0; b < (size_t)nb; b++) { if (hh[kStatW * b]) live++; for (int i = 0; i < kStatW; i++) hsum[i] += hh[kStatW * b + i]; }
        free(hh);
        const double d = live ? (double)live : 1.0;
        int32_t nn = 0, nr = 0;
        {
            unsigned char *hf = (unsigned char *)malloc((size_t)a.N);
            (void)hipMemcpy(hf, a.mx_flags, (size_t)a.N, hipMemcpyDeviceToHost);
            for (int i = 0; i < a.N; i++) { nn += hf[i] == kMxNulls; nr += hf[i] == kMxRest; }
            free(hf);
        }
        fprintf(stderr, "[mimc3 mx stats] ocw %d %s: %zu points staged; cycles/point: stage %.0f products %.0f box sums %.0f ncc %.0f climb %.0f replay %.0f fit %.0f; lists so far: nulls %d rest %d\n",
                C::OCW, C::GEN ? (C::CN ? "general" : "window nulls") : "clean", live, hsum[0] / d, hsum[1] / d, hsum[2] / d, hsum[3] / d, hsum[4] / d, hsum[5] / d, hsum[6] / d, nn, nr);
    }
    return hipGetLastError();
}

}  // namespace mx

bool match_mx_supported(int ocw, int max_npiv, int win_half, int max_abs_u, int max_abs_v)
{
    static const int off = getenv("MIMC3_MX") ? atoi(getenv("MIMC3_MX")) : 1;      // tuning / A-B: 0 = never take this kernel
    if (!off) return false;
    if (win_half > 0) return false;                   // full-square search areas (control-point stage): many pivots, not this kernel
    // A point whose pivots (with the ring of cells their first scans touch) do not fit the 32 x 32 tile is flagged for the
    // register-tiled kernel by the kernel itself, point by point: a velocity field with a few fast points keeps its slow ones here.
    // (A launch whose every corridor is too long -- BASELINE C4: 31 pivots -- only passes through: mx_preflag_kernel marks such points
    // beforehand and the kernel leaves them after one byte load; 158.0 -> 159.9 ms per pass at C4, 161.1 without the pre-pass.)  The launch's
    // maxima still size the register-tiled kernel's LDS carve, as without this kernel.
    (void)max_abs_u; (void)max_abs_v;
    if (max_npiv > 64) return false;                  // (the many-pivot kernel forms: no tables)
    return ocw == 7 || ocw == 15 || ocw == 16 || ocw == 30 || ocw == 32 || ocw == 40;
}

template <bool GEN, bool CN>
static hipError_t launch_form(const MatchU8Args &a, hipStream_t stream)
{
    switch (a.ocw) {
    case 7: return mx::launch_one<mx::Cfg<7, GEN, CN>>(a, stream);
    case 15: return mx::launch_one<mx::Cfg<15, GEN, CN>>(a, stream);
    case 16: return mx::launch_one<mx::Cfg<16, GEN, CN>>(a, stream);
    case 30: return mx::launch_one<mx::Cfg<30, GEN, CN>>(a, stream);
    case 32: return mx::launch_one<mx::Cfg<32, GEN, CN>>(a, stream);
    case 40: return mx::launch_one<mx::Cfg<40, GEN, CN>>(a, stream);
    default: return hipErrorInvalidValue;
    }
}

// Up to three launches: the clean form over all points (or the caller's list), then the forms for the points it flagged.  Points none
// takes carry kMxRest in mx_flags afterwards.
hipError_t launch_match_mx(MatchU8Args a, hipStream_t stream)
{
    if (a.N <= 0) return hipSuccess;
    if (!a.mx_flags || !a.sat0 || !a.sat1) return hipErrorInvalidValue;
    // Which null-ridden points stay on the matrix cores: none by default.  Both forms for them are built, tested and bit-identical at
    // every chip size, but measured at BASELINE C2 (ns per point; the register-tiled kernel's sparse corrections: 18 at ocw 16, 66 at
    // ocw 40) the window-null form (null-free chip: four correlations) costs 16 and leaves that kernel the chip-null points alone, which
    // it then runs at 23.5 -- 2.84 ms per pass against 2.75; the general form (chip nulls too: eight correlations) costs 19 at ocw 16
    // and 86 at ocw 40, its operand planes leaving two to three workgroups per CU.  The clean form takes the points without nulls
    // (8.9 ns against 11.3, 40 against 66 at ocw 40).  MIMC3_MX_WN / MIMC3_MX_GEN = 1 turn the other two on.
    static const int wn_env = getenv("MIMC3_MX_WN") ? atoi(getenv("MIMC3_MX_WN")) : 0;
    static const int gen_env = getenv("MIMC3_MX_GEN") ? atoi(getenv("MIMC3_MX_GEN")) : 0;
    a.mx_wn_on = wn_env > 0 ? 1 : 0;
    a.mx_gen_on = gen_env != 0 ? 1 : 0;
    if (a.mx_preflag) hipLaunchKernelGGL(mx::mx_preflag_kernel, dim3((unsigned)((a.N + 255) / 256)), dim3(256), 0, stream, a);
    hipError_t e = launch_form<false, false>(a, stream);
    a.point_flags = a.mx_flags;
    if (e == hipSuccess && a.mx_wn_on) { a.flag_value = kMxWn; e = launch_form<true, false>(a, stream); }
    if (e == hipSuccess && a.mx_gen_on) { a.flag_value = kMxNulls; e = launch_form<true, true>(a, stream); }
    return e;
}

}  // namespace mimc3
