// match_kernel.hip -- DLC/NCC matcher for gfx950 (MI355X), general f32 path.
//
// Replaces the OpenMP loop of matching_ncc_dlc_2 (MIMC_module.c:805-842) and everything it calls
// (extract_refchip :845-855, extract_sarea :857-890, investigate_valid_grid :605-644,
// find_ncc_peak :647-801).
//
// Decomposition (one 256-thread workgroup = 4 wave64 per grid point):
//   1. coalesced HBM -> LDS staging of the (2ocw+1)^2 chip and the Dy2 x Dx2 DLC window
//      (zero outside the image, zero last row/column = T4), null counts reduced on the way;
//   2. the "certain set": every pivot whose start passes the boundary test visits its whole 3x3
//      in its first hill-climb iteration (:709-735), so those cells are evaluated up front, one
//      cell per wave at a time, each lane owning a strided slice of the chip pixels;
//   3. wave 0 runs the reference's sequential hill-climb as a resumable state machine on the
//      cached NCC values (9 lanes = the 3x3 scan, shuffle arg-max with first-wins ties); when a
//      climb step needs cells that are not cached yet it parks, all 4 waves evaluate them, resume;
//   4. lane 0 does the 3x3 quadratic fit in the reference's f32/f64 mix.
//
// Arithmetic (bit-parity with the reference on fixed-point-like data):
//   f32 pixel products, f64 accumulation, f64 NCC formula, -ffp-contract=off (no FMA fusion).
//   Per-lane partial sums + a shuffle tree change only the summation ORDER, which is exact
//   whenever the f64 sums are exact (8/16-bit DN and the reference's gradient/Laplacian filters).
//
// `visited` (LDS bytes) mirrors "cmap >= -1.0" of the reference: cells that were only evaluated
// speculatively but never scanned still read as -2.0 in the final fit (T3, observable laziness).
#include <hip/hip_runtime.h>
#include <atomic>
#include <stdint.h>
#include "match_kernel.h"

namespace mimc3 {

static constexpr float kUnknown = 3.0f;  // NCC not evaluated (a real NCC is in [-1,1] or NaN)
static constexpr float kWanted = 4.0f;   // member of the certain set, not evaluated yet

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ int wave_sum(int v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Per-point geometry, wave-uniform.
struct Geo {
    int ocw, cw, npx;        // chip half width, width, pixel count
    int dx2, dy2, Dx2, Dy2;  // window half sizes and sizes (MIMC_module.c:863-866)
    int csx, csy;            // compact NCC map extents: window coords [ocw, D-ocw]
};

// Window accessor: LDS copy, or straight from the image when the window does not fit in LDS.
template <bool WIN_LDS>
struct Win {
    const float *lds;   // [Dy2][Dx2]
    const float *img;   // image the window is cut from
    int H, W, u_org, v_org, Dx2, lim_x, lim_y;   // window (0,0) = image (u_org, v_org); written area < lim
    __device__ __forceinline__ float at(int r, int c) const
    {
        if (WIN_LDS) return lds[r * Dx2 + c];
        int iu = u_org + c, iv = v_org + r;
        bool in = (r < lim_y) && (c < lim_x) && (iu >= 0) && (iu < W) && (iv >= 0) && (iv < H);
        return in ? img[(size_t)iv * W + iu] : 0.0f;
    }
};

// One NCC evaluation by one whole wave (MIMC_module.c:717-734).  (pu,pv) = window coordinates
// of the chip centre.  Every lane returns the value.
template <bool WIN_LDS>
__device__ __forceinline__ float eval_cell(const float *chip, const Win<WIN_LDS> &win, const Geo &g,
                                           int pu, int pv, int lane_r0, int lane_c0, float thr)
{
    const int lane = threadIdx.x & 63;
    const int dr = 64 / g.cw, dc = 64 % g.cw;
    int r = lane_r0 + (pv - g.ocw), c = lane_c0 + (pu - g.ocw);
    int cc = lane_c0;
    int n = 0;
    double sx = 0.0, sy = 0.0, sxx = 0.0, syy = 0.0, sxy = 0.0;
    for (int q = lane; q < g.npx; q += 64) {
        float a = chip[q];
        float b = win.at(r, c);
        const bool ok = (a >= thr) && (b >= thr);   // null exclusion (:723)
        a = ok ? a : 0.0f;
        b = ok ? b : 0.0f;
        n += ok ? 1 : 0;
        const float aa = a * a, bb = b * b, ab = a * b;   // f32 products (:728-730)
        sx += (double)a; sy += (double)b;
        sxx += (double)aa; syy += (double)bb; sxy += (double)ab;
        cc += dc; c += dc; r += dr;
        if (cc >= g.cw) { cc -= g.cw; c -= g.cw; r += 1; }
    }
    n = wave_sum(n);
    sx = wave_sum(sx); sy = wave_sum(sy); sxx = wave_sum(sxx); syy = wave_sum(syy); sxy = wave_sum(sxy);
    const double dn = (double)n;
    const double num = dn * sxy - sx * sy;
    const double den = sqrt((dn * sxx - sx * sx) * (dn * syy - sy * sy));
    return (float)(num / den);
}

// CELL_G: the NCC cache and the visited flags of the compact cell grid live in a global workspace slice of this
// workgroup instead of LDS (a long DIAGONAL corridor: (2|last pivot|+6)^2 cells no longer fit 160 KB).  Workgroup
// barriers order the accesses exactly as for LDS (all waves of a workgroup share one CU's L1).
template <bool WIN_LDS, bool CELL_G>
__device__ __forceinline__ void match_point_f32(const MatchArgs &p, int gidx, unsigned char *smem)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int NW = kMatchThreads / 64;

    const float *chip_img = p.swap ? p.i1 : p.i0;
    const float *win_img = p.swap ? p.i0 : p.i1;

    // ---- point header ---------------------------------------------------------------------
    const double *row = p.xyuvav + (size_t)p.xy_stride * (size_t)gidx + p.xy_col;
    const int u0 = (int)row[0], v0 = (int)row[1];          // T6 truncation (:822-823)
    const int64_t pbeg = p.piv_off[gidx];
    const int npiv = (int)(p.piv_off[gidx + 1] - pbeg);
    const int32_t *pv_g = p.piv_uv + 2 * pbeg;
    Geo g;
    g.ocw = p.ocw; g.cw = 2 * p.ocw + 1; g.npx = g.cw * g.cw;
    {
        int lu = pv_g[2 * (npiv - 1)], lv = pv_g[2 * (npiv - 1) + 1];
        g.dx2 = (lu < 0 ? -lu : lu) + p.ocw + 2;
        g.dy2 = (lv < 0 ? -lv : lv) + p.ocw + 2;
    }
    if (p.win_half > 0) { g.dx2 = p.win_half; g.dy2 = p.win_half; }
    g.Dx2 = 2 * g.dx2 + 1; g.Dy2 = 2 * g.dy2 + 1;
    g.csx = g.Dx2 - 2 * g.ocw + 1; g.csy = g.Dy2 - 2 * g.ocw + 1;
    const int ncell = g.csx * g.csy;

    // ---- LDS carve (sizes from the host-side maxima) ------------------------------------------
    float *chip = reinterpret_cast<float *>(smem);
    float *wlds = chip + p.lds_chip_f;
    float *val = wlds + p.lds_win_f;                       // [csy][csx] NCC cache
    int32_t *pivs = reinterpret_cast<int32_t *>(val + p.lds_cell_f);   // [npiv][2]
    int32_t *ctl = pivs + 2 * p.lds_npiv;                  // control words, see below
    unsigned char *vis = reinterpret_cast<unsigned char *>(ctl + 16);   // [csy][csx] visited flags
    if (CELL_G) {
        unsigned char *ws = p.cell_ws + (size_t)blockIdx.x * p.cell_ws_stride;
        val = reinterpret_cast<float *>(ws);
        vis = ws + 4 * (size_t)p.cell_ws_cells;
    }
    // ctl[0]=bad chip count, ctl[1]=bad window count, ctl[2]=state-machine status, ctl[3]=#pending,
    // ctl[4..12]=pending compact cell ids

    if (tid < 4) ctl[tid] = 0;
    for (int i = tid; i < ncell; i += kMatchThreads) { val[i] = kUnknown; vis[i] = 0; }
    for (int i = tid; i < 2 * npiv; i += kMatchThreads) pivs[i] = pv_g[i];
    __syncthreads();

    // ---- stage chip (a4) and window (a5), count nulls (a6) ------------------------------------
    int bad_chip = 0, bad_win = 0;
    for (int q = tid; q < g.npx; q += kMatchThreads) {
        const int r = q / g.cw, c = q - r * g.cw;
        const float a = chip_img[(size_t)(v0 - g.ocw + r) * p.W + (u0 - g.ocw + c)];
        chip[q] = a;
        bad_chip += (a < p.thr) ? 1 : 0;
    }
    Win<WIN_LDS> win;
    win.lds = wlds; win.img = win_img; win.H = p.H; win.W = p.W; win.Dx2 = g.Dx2;
    win.u_org = u0 + p.off_u - g.dx2; win.v_org = v0 + p.off_v - g.dy2;
    win.lim_x = 2 * g.dx2 + (p.win_half > 0 ? 1 : 0); win.lim_y = 2 * g.dy2 + (p.win_half > 0 ? 1 : 0);
    {
        const int nwin = g.Dx2 * g.Dy2;
        for (int q = tid; q < nwin; q += kMatchThreads) {
            const int r = q / g.Dx2, c = q - r * g.Dx2;
            const int iu = win.u_org + c, iv = win.v_org + r;
            const bool in = (r < win.lim_y) && (c < win.lim_x) && (iu >= 0) && (iu < p.W) && (iv >= 0) && (iv < p.H);
            const float b = in ? win_img[(size_t)iv * p.W + iu] : 0.0f;
            if (WIN_LDS) wlds[q] = b;
            bad_win += (b < p.thr) ? 1 : 0;
        }
    }
    bad_chip = wave_sum(bad_chip); bad_win = wave_sum(bad_win);
    if (lane == 0) { atomicAdd(&ctl[0], bad_chip); atomicAdd(&ctl[1], bad_win); }

    // ---- certain set: 3x3 around every pivot start that passes the boundary test (:703) ---------
    for (int k = tid; k < npiv; k += kMatchThreads) {
        const int pu = pv_g[2 * k] + g.dx2, pvv = pv_g[2 * k + 1] + g.dy2;
        if (pu - g.ocw <= 1 || pu + g.ocw >= g.Dx2 - 1 || pvv - g.ocw <= 1 || pvv + g.ocw >= g.Dy2 - 1) continue;
        for (int j = 0; j < 9; j++) {
            const int cx = pu + (j / 3 - 1) - g.ocw, cy = pvv + (j % 3 - 1) - g.ocw;
            val[cy * g.csx + cx] = kWanted;
        }
    }
    __syncthreads();

    // ---- validity (:635) --------------------------------------------------------------------------
    {
        const float max_ratio = 0.8f;
        const float rc = (float)ctl[0] / (float)(g.cw * g.cw);
        const float rw = (float)ctl[1] / (float)(g.Dx2 * g.Dy2);
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
    // T4: the last row/column of the reference's cmap is zero-filled memory ("computed, NCC 0").
    // Only reachable by the fit when ocw == 1; kept for completeness.
    if (g.ocw == 1)
        for (int i = tid; i < ncell; i += kMatchThreads) {
            const int cy = i / g.csx, cx = i - cy * g.csx;
            if (cx + g.ocw == g.Dx2 - 1 || cy + g.ocw == g.Dy2 - 1) { val[i] = 0.0f; vis[i] = 1; }
        }

    const int lane_r0 = lane / g.cw, lane_c0 = lane - lane_r0 * g.cw;

    // ---- evaluate the certain set: each wave scans its 64-cell chunks for wanted cells ---------
    for (int base = wave * 64; base < ncell; base += NW * 64) {
        const int cell = base + lane;
        const bool want = (cell < ncell) && (val[cell] == kWanted);
        unsigned long long m = __ballot(want);
        while (m) {
            const int l = __ffsll((long long)m) - 1;
            m &= m - 1;
            const int cidx = base + l;
            const int cy = cidx / g.csx, cx = cidx - cy * g.csx;
            const float ncc = eval_cell<WIN_LDS>(chip, win, g, cx + g.ocw, cy + g.ocw, lane_r0, lane_c0, p.thr);
            if (lane == 0) val[cidx] = ncc;
        }
    }
    __syncthreads();

    // ---- hill climb (wave 0, resumable) + on-demand evaluation (all waves) ---------------------
    // wave-uniform state of the reference's loops (:691-753)
    int k = 0, pu = 0, pvv = 0, du = 0, dv = 0, newncc = 0;
    bool fresh = true;
    float nccmax = -2.0f, best = -2.0f;
    int peak_u = g.dx2, peak_v = g.dy2;

    for (;;) {
        if (wave == 0) {
            int status = -1;   // 1 finished, 0 need cells
            while (status < 0) {
                if (fresh) {
                    if (k >= npiv) { status = 1; break; }
                    pu = pivs[2 * k] + g.dx2; pvv = pivs[2 * k + 1] + g.dy2;
                    nccmax = -2.0f; du = -1; dv = -1; newncc = 1; fresh = false;
                }
                bool end_pivot = !((du != 0 || dv != 0) && newncc != 0);
                if (!end_pivot &&
                    (pu - g.ocw <= 1 || pu + g.ocw >= g.Dx2 - 1 || pvv - g.ocw <= 1 || pvv + g.ocw >= g.Dy2 - 1))
                    end_pivot = true;                                       // boundary break (:703-707)
                if (end_pivot) {
                    if (nccmax > best) { peak_u = pu; peak_v = pvv; best = nccmax; }   // :747-752
                    k++; fresh = true;
                    continue;
                }
                // 3x3 scan, lane j <-> (c1 = j/3-1 outer, c2 = j%3-1 inner)  (:709-711)
                const bool act = lane < 9;
                const int c1 = lane / 3 - 1, c2 = lane % 3 - 1;
                const int cidx = act ? (pvv + c2 - g.ocw) * g.csx + (pu + c1 - g.ocw) : 0;
                const float v = val[cidx];
                const bool unvis = act && (vis[cidx] == 0);
                const bool missing = unvis && (v == kUnknown || v == kWanted);
                const unsigned long long mm = __ballot(missing);
                if (mm) {
                    if (missing) {
                        const int slot = __popcll(mm & ((1ull << lane) - 1ull));
                        ctl[4 + slot] = cidx;
                    }
                    if (lane == 0) ctl[3] = __popcll(mm);
                    status = 0;
                    break;
                }
                newncc = __popcll(__ballot(unvis));
                if (unvis) vis[cidx] = 1;
                // first-wins arg-max over the scan order; NaN never wins (:736-741)
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
            if (lane == 0) ctl[2] = status;
        }
        __syncthreads();
        if (ctl[2] == 1) break;
        const int npend = ctl[3];
        for (int i = wave; i < npend; i += NW) {
            const int cidx = ctl[4 + i];
            const int cy = cidx / g.csx, cx = cidx - cy * g.csx;
            const float ncc = eval_cell<WIN_LDS>(chip, win, g, cx + g.ocw, cy + g.ocw, lane_r0, lane_c0, p.thr);
            if (lane == 0) val[cidx] = ncc;
        }
        __syncthreads();
    }

    // ---- 3x3 quadratic fit (:757-788), one lane -------------------------------------------------
    if (tid == 0) {
        float n9[9];
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const int cidx = (peak_v - 1 + r - g.ocw) * g.csx + (peak_u - 1 + c - g.ocw);
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
        o0 += (float)(peak_u - g.dx2);
        o1 += (float)(peak_v - g.dy2);
        p.out[3 * (size_t)gidx + 0] = o0;
        p.out[3 * (size_t)gidx + 1] = o1;
        p.out[3 * (size_t)gidx + 2] = best;
    }
}

template <bool WIN_LDS, bool CELL_G>
__global__ __launch_bounds__(kMatchThreads) void match_ncc_dlc_f32(MatchArgs p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if (p.point_list) {
        // list mode: re-run of the few points another kernel handed back (u8 path cache overflow);
        // a small persistent grid strides over the device-side list
        const int cnt = *p.point_count;
        for (int i = blockIdx.x; i < cnt; i += gridDim.x) {
            match_point_f32<WIN_LDS, CELL_G>(p, p.point_list[i], smem);
            __syncthreads();
        }
        return;
    }
    if (CELL_G) {                                   // persistent grid: one workspace slice per workgroup
        for (int i = blockIdx.x; i < p.N; i += gridDim.x) {
            match_point_f32<WIN_LDS, CELL_G>(p, i, smem);
            __syncthreads();
        }
        return;
    }
    // XCD-aware point order: workgroups are dealt round-robin over the 8 XCDs, so give each XCD a
    // contiguous run of grid points (neighbours share most of their window rows in that L2).
    int gidx = blockIdx.x;
    {
        const int nb = gridDim.x, per = nb >> 3;
        if (per > 0 && gidx < per * 8) gidx = (gidx & 7) * per + (gidx >> 3);
    }
    if (gidx >= p.N) return;
    match_point_f32<WIN_LDS, CELL_G>(p, gidx, smem);
}

// ---- host-side launcher --------------------------------------------------------------------------
static size_t lds_layout(MatchArgs &a, int ocw, int max_abs_u, int max_abs_v, int max_npiv, bool win_lds, bool cell_lds)
{
    const int cw = 2 * ocw + 1;
    int Dx2 = 2 * (max_abs_u + ocw + 2) + 1, Dy2 = 2 * (max_abs_v + ocw + 2) + 1;
    if (a.win_half > 0) Dx2 = Dy2 = 2 * a.win_half + 1;
    const int cells = (Dx2 - 2 * ocw + 1) * (Dy2 - 2 * ocw + 1);
    a.lds_chip_f = (cw * cw + 3) & ~3;
    a.lds_win_f = win_lds ? ((Dx2 * Dy2 + 3) & ~3) : 0;
    a.lds_cell_f = cell_lds ? ((cells + 3) & ~3) : 0;
    a.lds_npiv = (max_npiv + 1) & ~1;
    a.cell_ws_cells = (cells + 3) & ~3;
    size_t bytes = sizeof(float) * ((size_t)a.lds_chip_f + a.lds_win_f + a.lds_cell_f) +
                   sizeof(int32_t) * (2 * (size_t)a.lds_npiv + 16) + (cell_lds ? (size_t)((cells + 15) & ~15) : 16);
    return bytes;
}

// bytes of global workspace launch_match_f32 needs for this launch (0 unless the cell grid outgrows LDS)
size_t match_f32_workspace_bytes(int ocw, int max_abs_u, int max_abs_v, int max_npiv, int win_half)
{
    MatchArgs a{};
    a.win_half = win_half;
    if (lds_layout(a, ocw, max_abs_u, max_abs_v, max_npiv, false, true) <= 160 * 1024) return 0;
    return (size_t)kCellGlobalGrid * (5 * (size_t)a.cell_ws_cells + 256);
}

hipError_t launch_match_f32(MatchArgs a, int max_abs_u, int max_abs_v, int max_npiv, hipStream_t stream)
{
    if (a.N <= 0) return hipSuccess;
    static std::atomic<unsigned long long> attr_done{0ull};   // the attribute is per device: bit d = device d configured (contexts of several devices may launch from several host threads)
    const size_t kLdsCap = 160 * 1024;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev >= 64 || !((attr_done.load() >> dev) & 1ull)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&match_ncc_dlc_f32<true, false>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsCap);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&match_ncc_dlc_f32<false, false>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsCap);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&match_ncc_dlc_f32<false, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsCap);
        if (dev < 64) attr_done.fetch_or(1ull << dev);
    }
    // window and cell grid in LDS -> cell grid in LDS, window through L2 -> both outside LDS (persistent grid + workspace)
    int mode = 0;
    size_t bytes = lds_layout(a, a.ocw, max_abs_u, max_abs_v, max_npiv, true, true);
    if (bytes > kLdsCap) { mode = 1; bytes = lds_layout(a, a.ocw, max_abs_u, max_abs_v, max_npiv, false, true); }
    if (bytes > kLdsCap) {
        mode = 2; bytes = lds_layout(a, a.ocw, max_abs_u, max_abs_v, max_npiv, false, false);
        a.cell_ws_stride = 5 * (size_t)a.cell_ws_cells + 256;
        if (bytes > kLdsCap || !a.cell_ws || a.cell_ws_bytes < (size_t)kCellGlobalGrid * a.cell_ws_stride) return hipErrorInvalidValue;
    }
    // grid rounded up to a multiple of 8 so the XCD remap is a bijection on [0, 8*per)
    unsigned nb = (unsigned)((a.N + 7) & ~7);
    if (a.point_list) nb = nb < 256u ? nb : 256u;        // list mode: one persistent workgroup per CU (the list is usually empty)
    if (mode == 2) nb = nb < (unsigned)kCellGlobalGrid ? nb : (unsigned)kCellGlobalGrid;
    if (mode == 0)
        hipLaunchKernelGGL((match_ncc_dlc_f32<true, false>), dim3(nb), dim3(kMatchThreads), bytes, stream, a);
    else if (mode == 1)
        hipLaunchKernelGGL((match_ncc_dlc_f32<false, false>), dim3(nb), dim3(kMatchThreads), bytes, stream, a);
    else
        hipLaunchKernelGGL((match_ncc_dlc_f32<false, true>), dim3(nb), dim3(kMatchThreads), bytes, stream, a);
    return hipGetLastError();
}

}  // namespace mimc3
