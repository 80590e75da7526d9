// match_u8_kernel.hip -- DLC/NCC matcher for gfx950, exact-integer path for 8-bit imagery.
//
// Same contract as match_kernel.hip (matching_ncc_dlc_2, MIMC_module.c:805-842) but for image
// pairs whose pixels are all integers in [0,255] (what GMA_float_load_tiff yields for an 8-bit
// TIFF, GMA.c:288-310).  For such data every running sum of the reference's NCC loop
// (MIMC_module.c:719-733: n, sx, sy, sxx, syy, sxy; f32 products, f64 accumulation) is an exact
// integer < 2^31, so the sums are computed with v_dot4_u32_u8 on packed bytes and converted to
// f64 only for the final formula (:734) -- bit-identical to the reference, ~4 MACs per VALU op.
//
// Layout / decomposition (one wave64 = one grid point, no workgroup barriers):
//   * the images live in HBM as zero-bordered u8 planes (border >= kU8Pad px, pitch % 4 == 0), built
//     once per image pair by prep_u8_plane (which also PROVES the pair is 8-bit integral);
//   * the DLC window is staged into LDS as aligned dwords (keeps the global byte phase `sh`);
//     null pixels are DN == 0, so masks are derived from the bytes themselves;
//   * the chip lives in REGISTERS: each of the 64/LPC lane groups holds the whole chip, lane l of a
//     group owns rows l, l+LPC, ... as packed dwords (+ a few single-group "tail" tasks);
//   * one evaluation round computes 64/LPC NCC cells at once: a lane slides over the aligned window
//     dwords of its row, v_alignbyte_b32 extracts the 4 window bytes under each chip group, dot4
//     accumulates; the LPC lanes of a cell are reduced with DPP row operations;
//   * FAST mode (no null pixel in chip or window): n, sx, sxx are per-point constants -> 3 dot4 per
//     4 pixels; otherwise GENERAL mode: 6 dot4 + byte-mask algebra per 4 pixels;
//   * the hill climb (MIMC_module.c:691-753) runs on the same wave as a resumable state machine over
//     the cached NCC values, exactly as in match_kernel.hip (first-wins arg-max, observable laziness).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "match_kernel.h"

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
    x += __builtin_amdgcn_update_dpp(0, x, 0x141, 0xF, 0xF, true);   // row_half_mirror
    x += __builtin_amdgcn_update_dpp(0, x, 0x140, 0xF, 0xF, true);   // row_mirror -> sum of the 16-lane row
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

template <int OCW_, int LPC_>
struct U8Cfg {
    static constexpr int OCW = OCW_, LPC = LPC_;
    static constexpr int CW = 2 * OCW + 1, NPX = CW * CW;
    static constexpr int GPR = (CW + 3) / 4;                 // packed dwords per chip row
    static constexpr int RF = CW / LPC;                      // full rounds: rows l + LPC*i
    static constexpr int REM = CW - RF * LPC;                // leftover rows, split into single-group tasks
    static constexpr int TT = (REM * GPR + LPC - 1) / LPC;   // tail tasks per lane
    static constexpr int LASTN = CW - 4 * (GPR - 1);         // valid bytes of the last group (1..4)
    static constexpr uint32_t LASTFF = LASTN == 4 ? 0xffffffffu : ((1u << (8 * LASTN)) - 1u);
    static constexpr uint32_t LAST01 = LASTFF & 0x01010101u;
    static constexpr int CPR = 64 / LPC;                     // cells per evaluation round
};

struct U8Point {
    int dx2, dy2, Dx2, Dy2, csx, csy, ncell;
    int sh;            // byte phase of window column 0 inside its aligned dword
    int PW;            // LDS window pitch, bytes
    uint32_t SX, SXX;  // chip constants (FAST mode)
    bool fast;
};

// accumulators of one evaluation round (per lane, before the group reduction)
struct Acc { uint32_t n, sx, sy, sxx, syy, sxy; };

template <class C, bool FAST>
__device__ __forceinline__ void task(Acc &acc, uint32_t a, uint32_t mf, uint32_t pad01, uint32_t padff, uint32_t bw)
{
    if (FAST) {   // chip and window free of nulls: mf == padff, n/sx/sxx are per-point constants
        acc.sy = dot4(pad01, bw, acc.sy);
        acc.syy = dot4(padff == 0xffffffffu ? bw : (bw & padff), bw, acc.syy);
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

// One evaluation round: lane group g (LPC lanes) evaluates the cell whose chip origin in window
// coordinates is (cx, cy) (== compact cell coordinates).  Returns group-reduced sums in every lane.
template <class C, bool FAST>
__device__ __forceinline__ Acc eval_round(const unsigned char *W, const U8Point &pt, int cx, int cy, int l,
                                          const uint32_t (&A)[C::RF > 0 ? C::RF : 1][C::GPR],
                                          const uint32_t (&MF)[C::RF > 0 ? C::RF : 1][C::GPR],
                                          const uint32_t (&AT)[C::TT > 0 ? C::TT : 1],
                                          const uint32_t (&MFT)[C::TT > 0 ? C::TT : 1],
                                          const int (&toff)[C::TT > 0 ? C::TT : 1])
{
    Acc acc{0, 0, 0, 0, 0, 0};
    const int X = pt.sh + cx;
    const uint32_t s = (uint32_t)(X & 3);
    const unsigned char *base = W + cy * pt.PW + (X & ~3);
#pragma unroll
    for (int i = 0; i < C::RF; i++) {
        const uint32_t *rp = reinterpret_cast<const uint32_t *>(base + (l + C::LPC * i) * pt.PW);
        uint32_t w[C::GPR + 1];
#pragma unroll
        for (int j = 0; j <= C::GPR; j++) w[j] = rp[j];
#pragma unroll
        for (int j = 0; j < C::GPR; j++) {
            const uint32_t bw = alignb(w[j + 1], w[j], s);
            const uint32_t p01 = (j == C::GPR - 1) ? C::LAST01 : 0x01010101u;
            const uint32_t pff = (j == C::GPR - 1) ? C::LASTFF : 0xffffffffu;
            task<C, FAST>(acc, A[i][j], MF[i][j], p01, pff, bw);
        }
    }
#pragma unroll
    for (int k = 0; k < C::TT; k++) {
        const uint32_t *rp = reinterpret_cast<const uint32_t *>(base + toff[k]);
        const uint32_t bw = alignb(rp[1], rp[0], s);
        // tail tasks carry their pad/null masks in MFT (zero for unused slots): use the general
        // byte-mask form of the FAST sums so one code path serves every lane
        if (FAST) {
            acc.sy = dot4(MFT[k] & 0x01010101u, bw, acc.sy);
            acc.syy = dot4(bw & MFT[k], bw, acc.syy);
            acc.sxy = dot4(AT[k], bw, acc.sxy);
        } else {
            task<C, false>(acc, AT[k], MFT[k], 0, 0, bw);
        }
    }
    acc.sy = group_sum<C::LPC>(acc.sy); acc.syy = group_sum<C::LPC>(acc.syy); acc.sxy = group_sum<C::LPC>(acc.sxy);
    if (!FAST) {
        acc.n = group_sum<C::LPC>(acc.n); acc.sx = group_sum<C::LPC>(acc.sx); acc.sxx = group_sum<C::LPC>(acc.sxx);
    }
    return acc;
}

// NCC from exact integer sums (MIMC_module.c:734), f64, no contraction
__device__ __forceinline__ float ncc_from_sums(uint32_t n, uint32_t sx, uint32_t sy, uint32_t sxx, uint32_t syy, uint32_t sxy)
{
    const double dn = (double)n, dsx = (double)sx, dsy = (double)sy;
    const double num = dn * (double)sxy - dsx * dsy;
    const double den = sqrt((dn * (double)sxx - dsx * dsx) * (dn * (double)syy - dsy * dsy));
    return (float)(num / den);
}

template <class C>
__global__ __launch_bounds__(64) void match_ncc_dlc_u8(MatchU8Args p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x;
    const int l = lane & (C::LPC - 1), grp = lane / C::LPC;
    constexpr int OCW = C::OCW, CW = C::CW, GPR = C::GPR;

    int gidx = blockIdx.x;
    {
        const int nb = gridDim.x, per = nb >> 3;
        if (per > 0 && gidx < per * 8) gidx = (gidx & 7) * per + (gidx >> 3);   // XCD-contiguous point order
    }
    if (gidx >= p.N) return;

    const unsigned char *chip_pl = p.swap ? p.p1 : p.p0;
    const unsigned char *win_pl = p.swap ? p.p0 : p.p1;
    const int Wp = p.Wp, PAD = p.pad;

    // ---- point header -------------------------------------------------------------------------
    const double *row = p.xyuvav + 6 * (size_t)gidx;
    const int u0 = (int)row[2], v0 = (int)row[3];
    const int64_t pbeg = p.piv_off[gidx];
    const int npiv = (int)(p.piv_off[gidx + 1] - pbeg);
    const int32_t *pv_g = p.piv_uv + 2 * pbeg;
    U8Point pt;
    {
        const int lu = pv_g[2 * (npiv - 1)], lv = pv_g[2 * (npiv - 1) + 1];
        pt.dx2 = (lu < 0 ? -lu : lu) + OCW + 2;
        pt.dy2 = (lv < 0 ? -lv : lv) + OCW + 2;
    }
    pt.Dx2 = 2 * pt.dx2 + 1; pt.Dy2 = 2 * pt.dy2 + 1;
    pt.csx = pt.Dx2 - 2 * OCW + 1; pt.csy = pt.Dy2 - 2 * OCW + 1;
    pt.ncell = pt.csx * pt.csy;
    pt.PW = p.lds_pw;
    const int wu0 = u0 + p.off_u - pt.dx2 + PAD;     // plane column of window column 0
    const int wv0 = v0 + p.off_v - pt.dy2 + PAD;     // plane row of window row 0
    pt.sh = wu0 & 3;

    // ---- LDS carve ------------------------------------------------------------------------------
    unsigned char *W = smem;                                              // [Dy2][PW]
    float *val = reinterpret_cast<float *>(smem + p.lds_off_val);         // [csy][csx]
    unsigned char *vis = smem + p.lds_off_vis;                            // [csy][csx]
    uint16_t *list = reinterpret_cast<uint16_t *>(smem + p.lds_off_list); // certain-set cell ids
    uint32_t *sums = reinterpret_cast<uint32_t *>(smem + p.lds_off_sums); // [64][6]
    int32_t *pivs = reinterpret_cast<int32_t *>(smem + p.lds_off_piv);    // [npiv][2]

    for (int i = lane; i < pt.ncell; i += 64) { val[i] = kUnknown; vis[i] = 0; }
    for (int i = lane; i < 2 * npiv; i += 64) pivs[i] = pv_g[i];

    // ---- stage the window as aligned dwords; count null bytes on the way (a5, a6) ---------------
    int bad_win = 0;
    {
        const int wcols = 2 * pt.dx2, wrows = 2 * pt.dy2;                 // written area (:869-886)
        const int nd = (pt.sh + wcols + 3) >> 2;                          // aligned dwords per row
        const uint32_t inv = (uint32_t)(0xffffffffu / (uint32_t)nd) + 1u; // exact idx/nd for idx*nd < 2^32
        const int tot = wrows * nd;
        const int lastb = (pt.sh + wcols) & 3;                            // valid bytes in the last dword (0 = all)
        const uint32_t first_ff = 0xffffffffu << (8 * pt.sh);
        const uint32_t last_ff = lastb ? ((1u << (8 * lastb)) - 1u) : 0xffffffffu;
        const uint32_t *gbase = reinterpret_cast<const uint32_t *>(win_pl + (size_t)wv0 * Wp + (wu0 & ~3));
        const int gpitch = Wp >> 2;
        for (int idx = lane; idx < tot; idx += 64) {
            const int r = (int)__umulhi((uint32_t)idx, inv);
            const int c = idx - r * nd;
            uint32_t v = gbase[(size_t)r * gpitch + c];
            uint32_t keep = 0xffffffffu;
            if (c == 0) keep &= first_ff;
            if (c == nd - 1) keep &= last_ff;
            v &= keep;                                                    // bytes outside the written columns -> 0 (covers T4 column)
            *reinterpret_cast<uint32_t *>(W + r * pt.PW + 4 * c) = v;
            bad_win += __popc(keep & 0x01010101u) - __popc((nz80(v) >> 7));
        }
        // T4: the last window row is never written by the reference -> zeros; also clear the dword
        // after each row's last written dword (read by the sliding loads of the right-most cells)
        const int ndz = pt.PW >> 2;
        for (int c = lane; c < ndz; c += 64) *reinterpret_cast<uint32_t *>(W + wrows * pt.PW + 4 * c) = 0u;
        for (int r = lane; r < wrows; r += 64)
            for (int c = nd; c < ndz; c++) *reinterpret_cast<uint32_t *>(W + r * pt.PW + 4 * c) = 0u;
        bad_win = wave_sum_i(bad_win) + pt.Dx2 + pt.Dy2 - 1;              // + the never-written last row and column
    }

    // ---- chip -> registers (a4): every lane group holds the whole chip --------------------------
    constexpr int RFA = C::RF > 0 ? C::RF : 1, TTA = C::TT > 0 ? C::TT : 1;
    uint32_t A[RFA][GPR], MF[RFA][GPR], AT[TTA], MFT[TTA];
    int toff[TTA];
    int bad_chip = 0;
    uint32_t SX = 0, SXX = 0;
    {
        const int cu0 = u0 - OCW + PAD, cv0 = v0 - OCW + PAD;
        const uint32_t sa = (uint32_t)(cu0 & 3);
        const uint32_t *gbase = reinterpret_cast<const uint32_t *>(chip_pl + (size_t)cv0 * Wp + (cu0 & ~3));
        const int gpitch = Wp >> 2;
#pragma unroll
        for (int i = 0; i < C::RF; i++) {
            const uint32_t *rp = gbase + (size_t)(l + C::LPC * i) * gpitch;
            uint32_t g[GPR + 1];
#pragma unroll
            for (int j = 0; j <= GPR; j++) g[j] = rp[j];
#pragma unroll
            for (int j = 0; j < GPR; j++) {
                uint32_t a = alignb(g[j + 1], g[j], sa);
                const uint32_t pff = (j == GPR - 1) ? C::LASTFF : 0xffffffffu;
                a &= pff;
                A[i][j] = a;
                const uint32_t t = nz80(a);
                MF[i][j] = ff_from80(t);
                bad_chip += __popc(pff & 0x01010101u) - __popc(t >> 7);
                SX = dot4(a, 0x01010101u, SX);
                SXX = dot4(a, a, SXX);
            }
        }
#pragma unroll
        for (int k = 0; k < C::TT; k++) {
            const int tt = l + C::LPC * k;
            const bool on = tt < C::REM * GPR;
            const int rr = C::RF * C::LPC + (on ? tt / GPR : 0), j = on ? tt % GPR : 0;
            const uint32_t *rp = gbase + (size_t)rr * gpitch + j;
            uint32_t a = alignb(rp[1], rp[0], sa);
            const uint32_t pff = on ? ((j == GPR - 1) ? C::LASTFF : 0xffffffffu) : 0u;
            a &= pff;
            AT[k] = a;
            const uint32_t t = nz80(a);
            MFT[k] = ff_from80(t);
            toff[k] = rr * pt.PW + 4 * j;
            bad_chip += __popc(pff & 0x01010101u) - __popc(t >> 7);
            SX = dot4(a, 0x01010101u, SX);
            SXX = dot4(a, a, SXX);
        }
        bad_chip = (int)group_sum<C::LPC>((uint32_t)bad_chip);
        SX = group_sum<C::LPC>(SX); SXX = group_sum<C::LPC>(SXX);
    }
    pt.SX = SX; pt.SXX = SXX;
    __syncthreads();   // single-wave workgroup: orders the LDS stores above before the reads below

    // ---- validity (a6, :635) --------------------------------------------------------------------
    {
        const float max_ratio = 0.8f;
        const float rc = (float)bad_chip / (float)(CW * CW);
        const float rw = (float)bad_win / (float)(pt.Dx2 * pt.Dy2);
        if (rc > max_ratio || rw > max_ratio) {
            if (lane == 0) {
                const float nanv = __builtin_nanf("");
                p.out[3 * (size_t)gidx + 0] = nanv;
                p.out[3 * (size_t)gidx + 1] = nanv;
                p.out[3 * (size_t)gidx + 2] = -3.0f;
            }
            return;
        }
    }
    pt.fast = (bad_chip == 0) && (bad_win == pt.Dx2 + pt.Dy2 - 1);
    if (OCW == 1)   // T4 cmap cells (only reachable by the fit when ocw == 1)
        for (int i = lane; i < pt.ncell; i += 64) {
            const int cy = i / pt.csx, cx = i - cy * pt.csx;
            if (cx + OCW == pt.Dx2 - 1 || cy + OCW == pt.Dy2 - 1) { val[i] = 0.0f; vis[i] = 1; }
        }

    // ---- certain set: 3x3 around every pivot start that passes the boundary test ----------------
    for (int k = lane; k < npiv; k += 64) {
        const int pu = pivs[2 * k] + pt.dx2, pvv = pivs[2 * k + 1] + pt.dy2;
        if (pu - OCW <= 1 || pu + OCW >= pt.Dx2 - 1 || pvv - OCW <= 1 || pvv + OCW >= pt.Dy2 - 1) continue;
#pragma unroll
        for (int j = 0; j < 9; j++) val[(pvv + (j % 3 - 1) - OCW) * pt.csx + (pu + (j / 3 - 1) - OCW)] = kWanted;
    }
    __syncthreads();
    int nlist = 0;
    for (int base = 0; base < pt.ncell; base += 64) {
        const int cell = base + lane;
        const bool want = (cell < pt.ncell) && (val[cell] == kWanted);
        const unsigned long long m = __ballot(want);
        if (want) list[nlist + __popcll(m & ((1ull << lane) - 1ull))] = (uint16_t)cell;
        nlist += __popcll(m);
    }
    __syncthreads();

    // evaluates `cnt` cells whose ids are ids[0..cnt) (LDS) and stores their NCC into val[]
    auto evaluate = [&](const uint16_t *ids, int cnt) {
        for (int b0 = 0; b0 < cnt; b0 += 64) {
            const int nb = (cnt - b0) < 64 ? (cnt - b0) : 64;
            for (int r0 = 0; r0 < nb; r0 += C::CPR) {
                const int slot = r0 + grp;
                const bool on = slot < nb;
                const int cell = on ? (int)ids[b0 + slot] : 0;
                const int cy = cell / pt.csx, cx = cell - cy * pt.csx;
                // FAST needs a null-free box: cells touching the zero last row/column (T4) do not qualify
                const bool edge = on && (cx == pt.csx - 2 || cy == pt.csy - 2);
                Acc acc;
                if (pt.fast && !__any(edge)) {
                    acc = eval_round<C, true>(W, pt, cx, cy, l, A, MF, AT, MFT, toff);
                    acc.n = C::NPX; acc.sx = pt.SX; acc.sxx = pt.SXX;
                } else {
                    acc = eval_round<C, false>(W, pt, cx, cy, l, A, MF, AT, MFT, toff);
                }
                if (on && l == 0) {
                    uint32_t *sp = sums + 6 * slot;
                    sp[0] = acc.n; sp[1] = acc.sx; sp[2] = acc.sy; sp[3] = acc.sxx; sp[4] = acc.syy; sp[5] = acc.sxy;
                }
            }
            __syncthreads();
            if (lane < nb) {
                const uint32_t *sp = sums + 6 * lane;
                val[ids[b0 + lane]] = ncc_from_sums(sp[0], sp[1], sp[2], sp[3], sp[4], sp[5]);
            }
            __syncthreads();
        }
    };
    evaluate(list, nlist);

    // ---- hill climb (resumable) + on-demand evaluation -----------------------------------------
    int k = 0, pu = 0, pvv = 0, du = 0, dv = 0, newncc = 0;
    bool fresh = true;
    float nccmax = -2.0f, best = -2.0f;
    int peak_u = pt.dx2, peak_v = pt.dy2;
    for (int guard = 0; guard <= pt.ncell + 8; guard++) {
        int npend = 0;
        bool finished = false;
        for (;;) {
            if (fresh) {
                if (k >= npiv) { finished = true; break; }
                pu = pivs[2 * k] + pt.dx2; pvv = pivs[2 * k + 1] + pt.dy2;
                nccmax = -2.0f; du = -1; dv = -1; newncc = 1; fresh = false;
            }
            bool end_pivot = !((du != 0 || dv != 0) && newncc != 0);
            if (!end_pivot && (pu - OCW <= 1 || pu + OCW >= pt.Dx2 - 1 || pvv - OCW <= 1 || pvv + OCW >= pt.Dy2 - 1))
                end_pivot = true;
            if (end_pivot) {
                if (nccmax > best) { peak_u = pu; peak_v = pvv; best = nccmax; }
                k++; fresh = true;
                continue;
            }
            const bool act = lane < 9;
            const int c1 = lane / 3 - 1, c2 = lane % 3 - 1;
            const int cidx = act ? (pvv + c2 - OCW) * pt.csx + (pu + c1 - OCW) : 0;
            const float v = val[cidx];
            const bool unvis = act && (vis[cidx] == 0);
            const bool missing = unvis && (v == kUnknown || v == kWanted);
            const unsigned long long mm = __ballot(missing);
            if (mm) {
                if (missing) list[__popcll(mm & ((1ull << lane) - 1ull))] = (uint16_t)cidx;
                npend = __popcll(mm);
                break;
            }
            newncc = __popcll(__ballot(unvis));
            if (unvis) vis[cidx] = 1;
            float bv = (act && v == v) ? v : -__builtin_inff();
            int bi = lane;
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) {
                const float ov = __shfl_xor(bv, o, 64);
                const int oi = __shfl_xor(bi, o, 64);
                if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
            }
            bv = __shfl(bv, 0, 64); bi = __shfl(bi, 0, 64);
            du = 0; dv = 0;
            if (bv > nccmax) { nccmax = bv; du = bi / 3 - 1; dv = bi % 3 - 1; }
            pu += du; pvv += dv;
        }
        if (finished) break;
        __syncthreads();
        evaluate(list, npend);
    }

    // ---- 3x3 quadratic fit (:757-788) ----------------------------------------------------------
    if (lane == 0) {
        float n9[9];
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const int cidx = (peak_v - 1 + r - OCW) * pt.csx + (peak_u - 1 + c - OCW);
                n9[3 * r + c] = vis[cidx] ? val[cidx] : -2.0f;
            }
        double cp0, cp1, cp2, cp3, cp4;
        cp0 = 6 * n9[0] - 12 * n9[1] + 6 * n9[2] + 6 * n9[3] - 12 * n9[4] + 6 * n9[5] + 6 * n9[6] - 12 * n9[7] + 6 * n9[8];
        cp1 = 9 * n9[0] - 9 * n9[2] - 9 * n9[6] + 9 * n9[8];
        cp2 = 6 * n9[0] + 6 * n9[1] + 6 * n9[2] - 12 * n9[3] - 12 * n9[4] - 12 * n9[5] + 6 * n9[6] + 6 * n9[7] + 6 * n9[8];
        cp3 = -6 * n9[0] + 6 * n9[2] - 6 * n9[3] + 6 * n9[5] - 6 * n9[6] + 6 * n9[8];
        cp4 = -6 * n9[0] - 6 * n9[1] - 6 * n9[2] + 6 * n9[6] + 6 * n9[7] + 6 * n9[8];
        cp0 /= 36; cp1 /= 36; cp2 /= 36; cp3 /= 36; cp4 /= 36;
        float o0 = (float)(-2 * cp2 * cp3 + cp1 * cp4);
        float o1 = (float)(-2 * cp0 * cp4 + cp1 * cp3);
        const double det = 4 * cp0 * cp2 - cp1 * cp1;
        o0 = (float)((double)o0 / det);
        o1 = (float)((double)o1 / det);
        o0 += (float)(peak_u - pt.dx2);
        o1 += (float)(peak_v - pt.dy2);
        p.out[3 * (size_t)gidx + 0] = o0;
        p.out[3 * (size_t)gidx + 1] = o1;
        p.out[3 * (size_t)gidx + 2] = best;
    }
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

hipError_t launch_prep_u8(const float *img, int H, int W, unsigned char *plane, int Wp, int pad, int *d_flag, hipStream_t s)
{
    dim3 blk(256), grd((W + 255) / 256, H);
    hipLaunchKernelGGL(prep_u8_plane, grd, blk, 0, s, img, H, W, plane, Wp, pad, d_flag);
    return hipGetLastError();
}

// ---- launcher -------------------------------------------------------------------------------------
template <class C>
static hipError_t launch_cfg(MatchU8Args a, int max_abs_u, int max_abs_v, int max_npiv, hipStream_t stream)
{
    const int Dx2 = 2 * (max_abs_u + C::OCW + 2) + 1, Dy2 = 2 * (max_abs_v + C::OCW + 2) + 1;
    const int cells = (Dx2 - 2 * C::OCW + 1) * (Dy2 - 2 * C::OCW + 1);
    // pitch: covering dwords of (phase 3 + Dx2 columns) + one zero dword + the sliding read-ahead
    a.lds_pw = 4 * (((3 + Dx2 + 3) >> 2) + 2);
    size_t off = (size_t)a.lds_pw * Dy2;
    off = (off + 15) & ~(size_t)15; a.lds_off_val = (int)off; off += 4 * (size_t)cells;
    off = (off + 15) & ~(size_t)15; a.lds_off_vis = (int)off; off += (size_t)cells;
    off = (off + 15) & ~(size_t)15; a.lds_off_list = (int)off; off += 2 * (size_t)(9 * max_npiv + 16);
    off = (off + 15) & ~(size_t)15; a.lds_off_sums = (int)off; off += 4 * 6 * 64;
    off = (off + 15) & ~(size_t)15; a.lds_off_piv = (int)off; off += 8 * (size_t)max_npiv;
    off = (off + 15) & ~(size_t)15;
    if (off > 160 * 1024) return hipErrorInvalidValue;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&match_ncc_dlc_u8<C>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    const unsigned nb = (unsigned)((a.N + 7) & ~7);
    hipLaunchKernelGGL(match_ncc_dlc_u8<C>, dim3(nb), dim3(64), off, stream, a);
    return hipGetLastError();
}

bool match_u8_supported(int ocw, int max_reach_u, int max_reach_v)
{
    if (!(ocw == 7 || ocw == 15 || ocw == 16 || ocw == 30 || ocw == 32 || ocw == 40)) return false;
    // the window hangs over the image edge by at most |last pivot| + |CP offset| + 2 pixels (+ up to 7
    // bytes of aligned read-ahead): all of it must stay inside the zero border
    return max_reach_u + 12 <= kU8Pad && max_reach_v + 12 <= kU8Pad;
}

hipError_t launch_match_u8(MatchU8Args a, int max_abs_u, int max_abs_v, int max_npiv, hipStream_t stream)
{
    if (a.N <= 0) return hipSuccess;
    switch (a.ocw) {
    case 7: return launch_cfg<U8Cfg<7, 16>>(a, max_abs_u, max_abs_v, max_npiv, stream);
    case 15: return launch_cfg<U8Cfg<15, 16>>(a, max_abs_u, max_abs_v, max_npiv, stream);
    case 16: return launch_cfg<U8Cfg<16, 16>>(a, max_abs_u, max_abs_v, max_npiv, stream);
    case 30: return launch_cfg<U8Cfg<30, 32>>(a, max_abs_u, max_abs_v, max_npiv, stream);
    case 32: return launch_cfg<U8Cfg<32, 32>>(a, max_abs_u, max_abs_v, max_npiv, stream);
    case 40: return launch_cfg<U8Cfg<40, 64>>(a, max_abs_u, max_abs_v, max_npiv, stream);
    default: return hipErrorInvalidValue;
    }
}

}  // namespace mimc3
