/*
 * oracle/mimc3_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C CPU restatement of the MIMC3 hot path, written from
 * the reference's behaviour, flat arrays instead of GMA structs.  It is the checker for the HIP
 * path: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.  The
 * product library (mimc3_amd/csrc) never links, imports or falls back to anything in oracle/.
 *
 * PINNING: this file is verified bit-for-bit against the compiled reference itself
 * (oracle/_ref/libmimc3_ref.so, built by oracle/Makefile from the sources under /root/reference)
 * in tests/test_oracle_vs_ref.py, and against the committed golden vectors that the same
 * reference build produced (tests/golden/, tests/test_oracle_golden.py).
 *
 * Arithmetic notes (all deliberate, all mirrored by the HIP kernels):
 *  - products of two f32 pixels are rounded to f32 first, then accumulated in f64, in the
 *    reference's loop order (u outer, v inner)                          MIMC_module.c:719-733
 *  - the NCC formula is evaluated in f64 and stored as f32               MIMC_module.c:734
 *  - the 3x3 fit coefficients are f32 expressions assigned to f64        MIMC_module.c:769-780
 *  - the peak offsets are stored to f32 after every step                 MIMC_module.c:783-788
 *  - build with -ffp-contract=off: the reference binary (x86-64 baseline, no FMA) never fuses.
 *
 * Defined-where-the-reference-is-undefined (documented in DESIGN.md):
 *  T4  last row/column of the search window are 0.0 (reference: never written, :869-886)
 *  T7  QM: a NaN fit / empty candidate list leaves the point unchanged (reference: compares
 *      stale or uninitialised stack variables, :2165-2193)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_MIN_DN 0.0000000001 /* MIMC_module.c:21, a double constant compared against f32 pixels */

/* =========================================================================================
 * a2  DLC pivots                                                   MIMC_module.c:543-602
 * ======================================================================================= */
static int32_t pivots_for_point(const double *row, float dt, float mpp, float aw_sf, float aw_cre,
                                int32_t ocw, int32_t H, int32_t W, int32_t *uv /* may be NULL */)
{
    float u = 0.0f, v = 0.0f, incr_u, incr_v, norm_incr, theta;
    double length_pivot;
    theta = atan2(row[5], row[4]);                       /* :559  f64 -> f32 */
    incr_u = cos(theta);                                 /* :560  cos((double)theta) -> f32 */
    incr_v = sin(theta);
    if (fabs(incr_u) > fabs(incr_v)) {                   /* :562-571, incr_u is normalised FIRST */
        incr_u = incr_u / fabs(incr_u);
        incr_v = incr_v / fabs(incr_u);
    } else {
        incr_u = incr_u / fabs(incr_v);
        incr_v = incr_v / fabs(incr_v);
    }
    norm_incr = sqrt(incr_u * incr_u + incr_v * incr_v); /* f32 expression, f64 sqrt, -> f32 */
    length_pivot = sqrt(row[4] * row[4] + row[5] * row[5]) / mpp / 365 * dt * aw_sf + aw_cre + 1;
    int32_t n = 0;
    while (u + (float)row[2] - (float)ocw > 0 && u + (float)row[2] + (float)ocw < (float)(W - 1) &&
           v + (float)row[3] - (float)ocw > 0 && v + (float)row[3] + (float)ocw < (float)(H - 1) &&
           length_pivot > (double)(norm_incr * (double)n)) {
        n++;
        u += incr_u;
        v += incr_v;
    }
    if (uv && n > 0) {
        u = 0.0f; v = 0.0f;
        uv[0] = 0; uv[1] = 0;
        for (int32_t k = 1; k < n; k++) {
            u += incr_u;
            v += incr_v;
            uv[2 * k + 0] = (int32_t)(u + 0.5);          /* f64 add, truncation toward zero */
            uv[2 * k + 1] = -(int32_t)(v + 0.5);
        }
    }
    return n;
}

/* Returns total pivots; -1 if cap (pairs) too small; -2 if some point has zero pivots. */
int64_t orc_get_uv_pivot(const double *xyuvav, int32_t N, float dt, float mpp, float aw_sf,
                         float aw_cre, int32_t ocw, int32_t H, int32_t W,
                         int64_t *piv_off, int32_t *piv_uv, int64_t cap)
{
    int64_t tot = 0;
    int bad = 0;
    piv_off[0] = 0;
    for (int32_t g = 0; g < N; g++) {
        int32_t n = pivots_for_point(xyuvav + 6 * (size_t)g, dt, mpp, aw_sf, aw_cre, ocw, H, W, NULL);
        if (n <= 0) { bad = 1; n = 0; }
        tot += n;
        piv_off[g + 1] = tot;
    }
    if (bad) return -2;
    if (tot > cap) return -1;
    for (int32_t g = 0; g < N; g++)
        pivots_for_point(xyuvav + 6 * (size_t)g, dt, mpp, aw_sf, aw_cre, ocw, H, W, piv_uv + 2 * piv_off[g]);
    return tot;
}

/* =========================================================================================
 * a3-a7  matcher                                                   MIMC_module.c:605-890
 * ======================================================================================= */
typedef struct {
    int ocw, cw;          /* chip half width, chip width 2*ocw+1 */
    int dx2, dy2, Dx2, Dy2;
    const float *chip;    /* [cw][cw] row-major (v,u) */
    const float *win;     /* [Dy2][Dx2] */
    float *cmap;          /* [Dy2][Dx2] */
} match_ws;

static float ncc_at(const match_ws *w, int pu, int pv)
{
    int nsample = 0;
    double sx = 0, sy = 0, sxx = 0, syy = 0, sxy = 0;
    const int ocw = w->ocw, cw = w->cw;
    for (int c3 = -ocw; c3 <= ocw; c3++)                /* u outer  (:719) */
        for (int c4 = -ocw; c4 <= ocw; c4++) {          /* v inner  (:721) */
            float a = w->chip[(c4 + ocw) * cw + (c3 + ocw)];
            float b = w->win[(size_t)(pv + c4) * w->Dx2 + (pu + c3)];
            if (a >= ORC_MIN_DN && b >= ORC_MIN_DN) {
                float aa = a * a, bb = b * b, ab = a * b;   /* f32 products (:728-730) */
                nsample++;
                sy += b; sx += a; sxx += aa; syy += bb; sxy += ab;
            }
        }
    return (float)((nsample * sxy - sx * sy) /
                   sqrt((nsample * sxx - sx * sx) * (nsample * syy - sy * sy)));
}

/* find_ncc_peak (:647-801) on an explicit chip [cw][cw] and search area [Dy2][Dx2] (both row-major (v,u)) */
static void find_peak(const float *chip, int ocw, const float *win, int Dx2, int Dy2, const int32_t *piv, int32_t npiv,
                      float *out3)
{
    const int cw = 2 * ocw + 1;
    match_ws w;
    w.ocw = ocw; w.cw = cw; w.Dx2 = Dx2; w.Dy2 = Dy2; w.dx2 = Dx2 / 2; w.dy2 = Dy2 / 2;
    float *cmap = (float *)calloc((size_t)Dx2 * Dy2, sizeof(float));      /* T4: last row/col 0.0 */
    for (int r = 0; r < 2 * w.dy2; r++)
        for (int c = 0; c < 2 * w.dx2; c++) cmap[(size_t)r * Dx2 + c] = -2.0f;   /* :678-681 */
    w.chip = chip; w.win = win; w.cmap = cmap;

    float uvncc0 = 0.0f, uvncc1 = 0.0f, best = -2.0f;
    /* a6 validity (:605-644): counts run over the whole search area */
    int bad_chip = 0, bad_win = 0;
    for (int i = 0; i < cw * cw; i++) if (chip[i] < ORC_MIN_DN) bad_chip++;
    for (int i = 0; i < Dx2 * Dy2; i++) if (win[i] < ORC_MIN_DN) bad_win++;
    const float max_ratio = 0.8;
    if ((float)bad_chip / (float)(cw * cw) > max_ratio || (float)bad_win / (float)(Dx2 * Dy2) > max_ratio) {
        const float nanv = sqrt(-1.0);
        out3[0] = nanv; out3[1] = nanv; out3[2] = -3.0f;
        free(cmap);
        return;
    }
    int peak_u = w.dx2, peak_v = w.dy2;
    for (int k = 0; k < npiv; k++) {                    /* a7 hill climb per pivot (:691-753) */
        int pu = piv[2 * k] + w.dx2, pv = piv[2 * k + 1] + w.dy2;
        float nccmax = -2.0f;
        int du = -1, dv = -1, newncc = 1;
        while ((du != 0 || dv != 0) && newncc != 0) {
            du = 0; dv = 0;
            if (pu - ocw <= 1 || pu + ocw >= w.Dx2 - 1 || pv - ocw <= 1 || pv + ocw >= w.Dy2 - 1) break;
            newncc = 0;
            for (int c1 = -1; c1 <= 1; c1++)
                for (int c2 = -1; c2 <= 1; c2++) {
                    float *cell = &cmap[(size_t)(pv + c2) * w.Dx2 + (pu + c1)];
                    if (*cell < -1.0) { newncc++; *cell = ncc_at(&w, pu + c1, pv + c2); }
                    if (*cell > nccmax) { nccmax = *cell; du = c1; dv = c2; }
                }
            pu += du; pv += dv;
        }
        if (nccmax > best) { peak_u = pu; peak_v = pv; best = nccmax; }
    }
    /* 3x3 quadratic fit, f32 expressions widened on assignment (:757-788) */
    float n9[9];
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) n9[3 * r + c] = cmap[(size_t)(peak_v - 1 + r) * w.Dx2 + (peak_u - 1 + c)];
    double cp[6];
    cp[0] = 6 * n9[0] - 12 * n9[1] + 6 * n9[2] + 6 * n9[3] - 12 * n9[4] + 6 * n9[5] + 6 * n9[6] - 12 * n9[7] + 6 * n9[8];
    cp[1] = 9 * n9[0] - 9 * n9[2] - 9 * n9[6] + 9 * n9[8];
    cp[2] = 6 * n9[0] + 6 * n9[1] + 6 * n9[2] - 12 * n9[3] - 12 * n9[4] - 12 * n9[5] + 6 * n9[6] + 6 * n9[7] + 6 * n9[8];
    cp[3] = -6 * n9[0] + 6 * n9[2] - 6 * n9[3] + 6 * n9[5] - 6 * n9[6] + 6 * n9[8];
    cp[4] = -6 * n9[0] - 6 * n9[1] - 6 * n9[2] + 6 * n9[6] + 6 * n9[7] + 6 * n9[8];
    cp[5] = -4 * n9[0] + 8 * n9[1] - 4 * n9[2] + 8 * n9[3] + 20 * n9[4] + 8 * n9[5] - 4 * n9[6] + 8 * n9[7] - 4 * n9[8];
    for (int i = 0; i < 6; i++) cp[i] /= 36;
    /* The reference stores each step into a float (:782-788), i.e. the numerator is rounded to f32 BEFORE the f64
     * division.  gcc 11 -O3's SLP vectoriser has been seen to fuse the pair of statements below and drop that
     * rounding (1 ulp off on golden `edge_windows`), hence the volatile stores (and -fno-tree-slp-vectorize). */
    volatile float q0 = -2 * cp[2] * cp[3] + cp[1] * cp[4];
    volatile float q1 = -2 * cp[0] * cp[4] + cp[1] * cp[3];
    uvncc0 = q0; uvncc1 = q1;
    uvncc0 /= 4 * cp[0] * cp[2] - cp[1] * cp[1];
    uvncc1 /= 4 * cp[0] * cp[2] - cp[1] * cp[1];
    q0 = uvncc0; q1 = uvncc1;
    uvncc0 = q0; uvncc1 = q1;
    uvncc0 += (float)(peak_u - w.dx2);
    uvncc1 += (float)(peak_v - w.dy2);
    out3[0] = uvncc0; out3[1] = uvncc1; out3[2] = best;
    free(cmap);
}

static void match_point(const float *i0, const float *i1, int32_t H, int32_t W, const double *row,
                        const int32_t *offset, const int32_t *piv, int32_t npiv, int32_t ocw,
                        float *out3)
{
    const int cw = 2 * ocw + 1;
    int u0 = (int32_t)row[2], v0 = (int32_t)row[3];     /* T6 truncation (:822-823) */
    const int dx2 = abs(piv[2 * (npiv - 1) + 0]) + ocw + 2;     /* :863-866 */
    const int dy2 = abs(piv[2 * (npiv - 1) + 1]) + ocw + 2;
    const int Dx2 = 2 * dx2 + 1, Dy2 = 2 * dy2 + 1;
    float *chip = (float *)malloc(sizeof(float) * cw * cw);
    float *win = (float *)calloc((size_t)Dx2 * Dy2, sizeof(float));   /* T4: zero-filled */
    /* a4 chip: no bounds check in the reference; caller guarantees the margin (:845-855) */
    for (int r = 0; r < cw; r++)
        memcpy(chip + r * cw, i0 + (size_t)(v0 - ocw + r) * W + (u0 - ocw), sizeof(float) * cw);
    /* a5 window around uv0+offset, zero outside the image, last row/col untouched (:869-886) */
    int uc = u0 + offset[0], vc = v0 + offset[1];
    for (int r = 0; r < 2 * dy2; r++) {
        int cv = vc - dy2 + r;
        for (int c = 0; c < 2 * dx2; c++) {
            int cu = uc - dx2 + c;
            win[(size_t)r * Dx2 + c] = (cu >= 0 && cu < W && cv >= 0 && cv < H) ? i1[(size_t)cv * W + cu] : 0.0f;
        }
    }
    find_peak(chip, ocw, win, Dx2, Dy2, piv, npiv, out3);
    free(chip); free(win);
}

/* out[N][3] = (du, dv, ncc).  nthreads<=0: OpenMP default.  Returns 0, or -3 on a point whose
 * chip would leave the image (the reference reads out of bounds there) or has no pivots. */
int orc_match_ncc_dlc(const float *i0, const float *i1, int32_t H, int32_t W, const double *xyuvav,
                      int32_t N, const int32_t *offset, const int32_t *piv_uv,
                      const int64_t *piv_off, int32_t ocw, float *out, int32_t nthreads)
{
    for (int32_t g = 0; g < N; g++) {
        int u0 = (int32_t)xyuvav[6 * (size_t)g + 2], v0 = (int32_t)xyuvav[6 * (size_t)g + 3];
        if (u0 - ocw < 0 || u0 + ocw >= W || v0 - ocw < 0 || v0 + ocw >= H) return -3;
        if (piv_off[g + 1] - piv_off[g] < 1) return -3;
    }
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel for schedule(dynamic)
    for (int32_t g = 0; g < N; g++)
        match_point(i0, i1, H, W, xyuvav + 6 * (size_t)g, offset, piv_uv + 2 * piv_off[g],
                    (int32_t)(piv_off[g + 1] - piv_off[g]), ocw, out + 3 * (size_t)g);
    return 0;
}

/* =========================================================================================
 * a8  neighbour offsets                                            MIMC_module.c:1266-1327
 * ======================================================================================= */
int32_t orc_get_ruv_neighbor(const double *xyuvav, int32_t N, int32_t dimx, int32_t dimy,
                             float meter_per_spacing, float radius, int32_t *ruv, int32_t cap)
{
    (void)N;
    int32_t cu = dimx / 2, cv = dimy / 2, nn = 0;
    float cx = (float)xyuvav[6 * (size_t)cu + 0];
    float cy = (float)xyuvav[6 * (size_t)cv * dimx + 1];
    float lim = (radius * meter_per_spacing) * (radius * meter_per_spacing);
    for (int32_t v = 0; v < dimy; v++)
        for (int32_t u = 0; u < dimx; u++) {
            float fx = (float)xyuvav[6 * (size_t)u + 0];               /* x taken from grid row 0 */
            float fy = (float)xyuvav[6 * (size_t)v * dimx + 1];        /* y taken from grid column 0 */
            float ddx = fx - cx, ddy = fy - cy;
            float sq = ddx * ddx + ddy * ddy;
            if (sq <= lim) {
                if (nn < cap) { ruv[2 * nn] = u - cu; ruv[2 * nn + 1] = v - cv; }
                nn++;
            }
        }
    return nn <= cap ? nn : -1;
}

/* =========================================================================================
 * a9-a10  QM pseudo-smoothing                                      MIMC_module.c:1986-2496
 * ======================================================================================= */
/* 6x6 Gauss-Jordan without pivoting, operation order as :2430-2496 */
static void inv6(const double a[6][6], double I[6][6])
{
    double b[6][6];
    for (int i = 0; i < 6; i++)
        for (int j = 0; j < 6; j++) { b[i][j] = a[i][j]; I[i][j] = (i == j) ? 1 : 0; }
    for (int p = 0; p < 5; p++) {
        double pivot = b[p][p];
        for (int r = p + 1; r < 6; r++) {
            double coeff = b[r][p] / pivot;
            for (int c = 0; c < 6; c++) { b[r][c] -= b[p][c] * coeff; I[r][c] -= I[p][c] * coeff; }
        }
    }
    for (int p = 5; p >= 0; p--) {
        double pivot = b[p][p];
        for (int r = p - 1; r >= 0; r--) {
            double coeff = b[r][p] / pivot;
            for (int c = 5; c >= 0; c--) { b[r][c] -= b[p][c] * coeff; I[r][c] -= I[p][c] * coeff; }
        }
    }
    for (int i = 0; i < 6; i++)
        for (int j = 0; j < 6; j++) I[i][j] /= b[i][i];
}

/* weighted quadratic LSQ evaluated at (0,0), both outputs (:2314-2409) */
static void quadfit_origin(const int32_t *xy, const double *z, const double *w, int n, double out[2])
{
    double Nm[6][6], IN[6][6];
    for (int r = 0; r < 6; r++)
        for (int c = 0; c < 6; c++) {
            double acc = 0;
            for (int o = 0; o < n; o++) {
                double x = (double)xy[2 * o], y = (double)xy[2 * o + 1];
                double A[6] = { x * x, x * y, y * y, x, y, 1 };
                acc += A[r] * w[o] * A[c];
            }
            Nm[r][c] = acc;
        }
    inv6(Nm, IN);
    for (int oc = 0; oc < 2; oc++) {
        double atwb[6], coef[6];
        for (int r = 0; r < 6; r++) {
            double acc = 0;
            for (int o = 0; o < n; o++) {
                double x = (double)xy[2 * o], y = (double)xy[2 * o + 1];
                double A[6] = { x * x, x * y, y * y, x, y, 1 };
                acc += A[r] * w[o] * z[2 * o + oc];
            }
            atwb[r] = acc;
        }
        for (int r = 0; r < 6; r++) {
            double acc = 0;
            for (int c = 0; c < 6; c++) acc += IN[r][c] * atwb[c];
            coef[r] = acc;
        }
        const double t[6] = { 0.0, 0.0, 0.0, 0.0, 0.0, 1 };  /* terms at xyi=(0,0): NaN/inf in any coef propagates */
        double val = 0;
        for (int c = 0; c < 6; c++) val += t[c] * coef[c];
        out[oc] = val;
    }
}

/* In place on dpf/dpf_dx/dpf_dy ([dimy][dimx]).  mvn: padded [N][Kmax][5]; nclus[N].
 * max_sweeps: the reference's loop bound is NOI<=100, i.e. up to 101 sweeps (pass 101 to mirror).
 * stats[0]=sweeps run (the reference's NOI at exit), stats[1]=points processed in total,
 * stats[2]=points skipped by the T7 definition (NaN fit / no candidate). */
int orc_qm_pseudosmooth(int32_t dimy, int32_t dimx, int32_t *dpf, float *dpf_dx, float *dpf_dy,
                        const int32_t *ruv, int32_t nn, const float *mvn, int32_t Kmax,
                        const int32_t *nclus, const double *xyuvav, int32_t max_sweeps, int64_t *stats)
{
    const int32_t N = dimx * dimy;
    const double eig0 = 1500.0 / 300.0, eig1 = eig0 / 3.0;
    uint8_t *mask0 = (uint8_t *)malloc(N), *mask = (uint8_t *)malloc(N), *next = (uint8_t *)malloc(N);
    uint8_t **stack = (uint8_t **)calloc((size_t)max_sweeps + 2, sizeof(uint8_t *));
    float *bx = (float *)malloc(sizeof(float) * N), *by = (float *)malloc(sizeof(float) * N);
    int32_t *bid = (int32_t *)malloc(sizeof(int32_t) * N);
    int32_t *nxy = (int32_t *)malloc(sizeof(int32_t) * 2 * nn);
    double *nz = (double *)malloc(sizeof(double) * 2 * nn), *nw = (double *)malloc(sizeof(double) * nn);
    const float nanv = sqrt(-1.0);
    int64_t processed = 0, skipped = 0;
    for (int32_t i = 0; i < N; i++) {                                   /* :2029-2055 */
        int32_t id = dpf[i];
        mask0[i] = (id >= 0 && !(mvn[((size_t)i * Kmax + id) * 5 + 4] >= 0.6)) ? 1 : 0;
        bx[i] = nanv; by[i] = nanv; bid[i] = -1;
    }
    memcpy(mask, mask0, N);
    stack[0] = (uint8_t *)malloc(N); memcpy(stack[0], mask0, N);
    int32_t noi = 0, any = 1;
    while (noi < max_sweeps && any) {
        any = 0; noi++;
        memset(next, 0, N);
        for (int32_t v = 0; v < dimy; v++)
            for (int32_t u = 0; u < dimx; u++) {
                const int32_t idx = v * dimx + u;
                if (!mask[idx]) continue;
                int n = 0;
                for (int32_t k = 0; k < nn; k++) {                      /* :2108-2126 */
                    int32_t uu = u + ruv[2 * k], vv = v + ruv[2 * k + 1];
                    if (uu < 0 || uu >= dimx || vv < 0 || vv >= dimy) continue;
                    float fx = dpf_dx[vv * dimx + uu], fy = dpf_dy[vv * dimx + uu];
                    if (isnan(fx) || isnan(fy)) continue;
                    nxy[2 * n] = ruv[2 * k]; nxy[2 * n + 1] = ruv[2 * k + 1];
                    nz[2 * n] = fx; nz[2 * n + 1] = fy;
                    n++;
                }
                if (n < 10) continue;                                   /* :2131 */
                processed++;
                const double vx = xyuvav[6 * (size_t)idx + 4], vy = xyuvav[6 * (size_t)idx + 5];
                const double den = (eig0 * eig1) * (vx * vx + vy * vy);
                const double itm0 = (eig1 * vx * vx + eig0 * vy * vy) / den;      /* :2144-2146 */
                const double itm1 = ((eig0 - eig1) * vx * vy) / den;
                const double itm3 = (eig1 * vy * vy + eig0 * vx * vx) / den;
                for (int o = 0; o < n; o++)                             /* :2151-2153 */
                    nw[o] = exp(-(itm0 * nxy[2 * o] * nxy[2 * o] + 2 * itm1 * nxy[2 * o] * nxy[2 * o + 1] +
                                  itm3 * nxy[2 * o + 1] * nxy[2 * o + 1]));
                double fit[2];
                quadfit_origin(nxy, nz, nw, n, fit);
                const int32_t id = dpf[idx], nc = nclus[idx];
                const float *cl = mvn + (size_t)idx * Kmax * 5;
                double dmin = 1E+37; int32_t best = -1;
                for (int32_t c = 0; c < nc; c++) {                      /* :2167-2180 */
                    double cu_ = cl[5 * c], cv_ = cl[5 * c + 1];
                    double sq = (fit[0] - cu_) * (fit[0] - cu_) + (fit[1] - cv_) * (fit[1] - cv_);
                    if (sq < dmin) { dmin = sq; best = c; }
                }
                if (best < 0) { skipped++; continue; }                  /* T7 definition */
                double gu = cl[5 * id], gv = cl[5 * id + 1], qu = cl[5 * best], qv = cl[5 * best + 1];
                if ((gu - qu) * (gu - qu) + (gv - qv) * (gv - qv) < 0.0001) continue;   /* :2190 */
                bx[idx] = (float)qu; by[idx] = (float)qv; bid[idx] = best;
                any = 1;
                for (int o = 0; o < n; o++) {                           /* :2203-2210 */
                    int32_t j = (v + nxy[2 * o + 1]) * dimx + (u + nxy[2 * o]);
                    if (mask0[j]) next[j] = 1;
                }
            }
        for (int32_t i = 0; i < N; i++)                                 /* Jacobi commit :2218-2233 */
            if (bid[i] >= 0) {
                dpf_dx[i] = bx[i]; dpf_dy[i] = by[i]; dpf[i] = bid[i];
                bx[i] = nanv; by[i] = nanv; bid[i] = -1;
            }
        int fluct = 0;                                                  /* :2237-2268 */
        for (int32_t s = noi - 1; s >= 0 && !fluct; s--)
            if (memcmp(stack[s], next, N) == 0) fluct = 1;
        if (fluct) { noi--; break; }
        stack[noi] = (uint8_t *)malloc(N); memcpy(stack[noi], next, N);
        memcpy(mask, next, N);
    }
    if (stats) { stats[0] = noi; stats[1] = processed; stats[2] = skipped; }
    for (int32_t s = 0; s <= max_sweeps + 1; s++) free(stack[s]);
    free(stack); free(mask0); free(mask); free(next); free(bx); free(by); free(bid);
    free(nxy); free(nz); free(nw);
    return 0;
}

int orc_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* =========================================================================================
 * N1  candidate clustering, dpf0, dpf1                         MIMC_module.c:994-1263, :1330-1718
 * (the stages between the 32 matcher passes and the QM update)
 * ======================================================================================= */
/* dp: [ndp][N][3] matcher outputs.  mvn: padded [N][Kmax][5], nclus[N].  Returns the largest cluster
 * count, or -1 if it exceeds Kmax.  (calc_mean_var_num_dp_cluster :994-1130, cluster_euclidian :1133-1180;
 * single linkage < 0.5 px: clusters are the connected components, numbered by their first member) */
int32_t orc_cluster_candidates(const float *dp, int32_t ndp, int32_t N, int32_t Kmax, float *mvn, int32_t *nclus)
{
    const float min_ncc = 0.1, min_dist = 0.5;
    const float min_dist_sq = min_dist * min_dist;
    int32_t kmax_seen = 0;
    float *px = (float *)malloc(sizeof(float) * ndp), *py = (float *)malloc(sizeof(float) * ndp);
    int *lab = (int *)malloc(sizeof(int) * ndp), *todo = (int *)malloc(sizeof(int) * ndp);
    float *sx = (float *)malloc(sizeof(float) * ndp), *sy = (float *)malloc(sizeof(float) * ndp);
    float *sxx = (float *)malloc(sizeof(float) * ndp), *syy = (float *)malloc(sizeof(float) * ndp);
    int *cnt = (int *)malloc(sizeof(int) * ndp);
    memset(mvn, 0, sizeof(float) * 5 * (size_t)N * Kmax);
    for (int32_t g = 0; g < N; g++) {
        int nv = 0;
        for (int k = 0; k < ndp; k++) {
            const float *o = dp + ((size_t)k * N + g) * 3;
            if (o[2] > min_ncc) { px[nv] = o[0]; py[nv] = o[1]; nv++; }
        }
        int ncl = 0;
        for (int i = 0; i < nv; i++) lab[i] = 0;
        for (int i = 0; i < nv; i++) {
            if (lab[i]) continue;
            ncl++;
            /* grow the component of i.  NOTE the reference seeds with mark_row(i): i itself is labelled only
             * through its own diagonal entry d(i,i)=0 < 0.25 -- false when the candidate is NaN, in which case it
             * keeps label 0 and is revisited... the recursion then never labels it: reproduce via the same test */
            int nt = 0; todo[nt++] = i;
            int self = 0;
            { float dx = px[i] - px[i], dy = py[i] - py[i]; self = (dx * dx + dy * dy < min_dist_sq); }
            if (self) lab[i] = ncl;
            while (nt) {
                int r = todo[--nt];
                for (int c = 0; c < nv; c++) {
                    float dx = px[c] - px[r], dy = py[c] - py[r];
                    if (dx * dx + dy * dy < min_dist_sq && lab[c] == 0) { lab[c] = ncl; todo[nt++] = c; }
                }
            }
        }
        /* max id actually used (a NaN candidate consumes an id without ever carrying it) */
        int max_id = 0;
        for (int i = 0; i < nv; i++) if (lab[i] > max_id) max_id = lab[i];
        if (max_id > kmax_seen) kmax_seen = max_id;
        nclus[g] = max_id;
        if (max_id > Kmax) continue;
        for (int c = 0; c < max_id; c++) { sx[c] = sy[c] = sxx[c] = syy[c] = 0; cnt[c] = 0; }
        for (int i = 0; i < nv; i++) {
            if (lab[i] == 0) continue;          /* reference indexes sx[-1] here (UB); never happens for finite candidates */
            int c = lab[i] - 1;
            sx[c] += px[i]; sy[c] += py[i];
            sxx[c] += px[i] * px[i]; syy[c] += py[i] * py[i];
            cnt[c] += 1;
        }
        float *m = mvn + (size_t)g * Kmax * 5;
        for (int c = 0; c < max_id; c++) {
            m[5 * c + 0] = sx[c] / (float)cnt[c];
            m[5 * c + 1] = sy[c] / (float)cnt[c];
            m[5 * c + 2] = sxx[c] / (float)cnt[c] - m[5 * c + 0] * m[5 * c + 0];
            m[5 * c + 3] = syy[c] / (float)cnt[c] - m[5 * c + 1] * m[5 * c + 1];
            m[5 * c + 4] = (float)cnt[c] / (float)ndp;
        }
    }
    free(px); free(py); free(lab); free(todo); free(sx); free(sy); free(sxx); free(syy); free(cnt);
    return kmax_seen > Kmax ? -1 : kmax_seen;
}

/* get_dpf0 (:1224-1263): first cluster whose fraction exceeds the ratio, else -1 */
void orc_get_dpf0(const float *mvn, const int32_t *nclus, int32_t N, int32_t Kmax, float min_ratio, int32_t *dpf)
{
    for (int32_t g = 0; g < N; g++) {
        dpf[g] = -1;
        for (int32_t c = 0; c < nclus[g]; c++)
            if (mvn[((size_t)g * Kmax + c) * 5 + 4] > min_ratio) { dpf[g] = c; break; }
    }
}

/* get_dpf1 (:1330-1718): a-priori-scaled neighbour interpolation of the unassigned points (Jacobi sweeps),
 * 3x3 smoothing, snap to the nearest cluster.  dpf in/out, dx/dy out.  Returns the sweep count (NOI). */
int32_t orc_get_dpf1(int32_t dimy, int32_t dimx, int32_t *dpf, float *dx, float *dy, const int32_t *ruv, int32_t nn,
                     const float *mvn, int32_t Kmax, const int32_t *nclus, const double *xyuvav, float dt, float mpp)
{
    const int32_t N = dimx * dimy;
    const float nanv = sqrt(-1.0);
    float *bx = (float *)malloc(sizeof(float) * N), *by = (float *)malloc(sizeof(float) * N);
    float *noi = (float *)malloc(sizeof(float) * N);
    float (*v)[7] = (float (*)[7])malloc(sizeof(float) * 7 * (size_t)nn);
    for (int32_t i = 0; i < N; i++) {
        noi[i] = 1.0;
        if (dpf[i] >= 0) { dx[i] = mvn[((size_t)i * Kmax + dpf[i]) * 5]; dy[i] = mvn[((size_t)i * Kmax + dpf[i]) * 5 + 1]; }
        else { dx[i] = nanv; dy[i] = nanv; }
        bx[i] = nanv; by[i] = nanv;
    }
    int32_t NOI = 0, unprocessed = 1;
    for (int32_t thres_num = nn - 1; thres_num >= 3; thres_num--) {
        float thres_weight = 0.5;
        const float factor = 1.0 / 365.0 * dt / mpp;
        while (unprocessed != 0 && thres_weight >= 0.5) {
            thres_weight -= 0.02;
            int32_t processed = 1;
            while (processed != 0) {
                NOI++;
                processed = 0;
                for (int32_t cv = 0; cv < dimy; cv++)
                    for (int32_t cu = 0; cu < dimx; cu++) {
                        const int32_t g = cv * dimx + cu;
                        if (!(isnan(dx[g] + dy[g]) && nclus[g] != 0)) continue;
                        int32_t num = 0;
                        float dpe[2];
                        dpe[0] = xyuvav[6 * (size_t)g + 4] * factor;
                        dpe[1] = -xyuvav[6 * (size_t)g + 5] * factor;
                        const float mag_dpe = sqrt(dpe[0] * dpe[0] + dpe[1] * dpe[1]);
                        for (int32_t k = 0; k < nn; k++) {
                            const int32_t u = cu + ruv[2 * k], w = cv + ruv[2 * k + 1];
                            if (u < 0 || u >= dimx || w < 0 || w >= dimy) continue;
                            const int32_t h = w * dimx + u;
                            const float n0 = dx[h], n1 = dy[h];
                            if (isnan(n0 + n1)) continue;
                            const float a0 = (float)(xyuvav[6 * (size_t)h + 4]) * factor;
                            const float a1 = -(float)(xyuvav[6 * (size_t)h + 5]) * factor;
                            v[num][0] = (float)ruv[2 * k]; v[num][1] = (float)ruv[2 * k + 1];
                            v[num][4] = sqrt(n0 * n0 + n1 * n1);
                            v[num][5] = sqrt(a0 * a0 + a1 * a1);
                            v[num][6] = noi[h];
                            v[num][3] = v[num][4] / sqrt(a0 * a0 + a1 * a1);
                            num++;
                        }
                        if (num < thres_num) continue;
                        float w_min = 1E+37, w_max = -1E+37;
                        const float max_noi = 1.0;
                        int32_t id_max = 0, id_min = 0;
                        for (int32_t i = 0; i < num; i++) {
                            const float d0 = v[i][0], d1 = v[i][1];
                            const float mag = sqrt(d0 * d0 + d1 * d1);
                            float wc = (dpe[0] * d0 + dpe[1] * d1) / (mag_dpe * mag);
                            wc = wc > 0 ? wc : -wc;
                            if (wc >= thres_weight) {
                                v[i][2] = wc;
                                if (v[i][3] > w_max) { w_max = v[i][3]; id_max = i; }
                                if (v[i][3] < w_min) { w_min = v[i][3]; id_min = i; }
                            } else v[i][2] = 0.0;
                        }
                        v[id_max][2] = 0.0; v[id_min][2] = 0.0;
                        float s_mw = 0.0, s_w = 0.0, s_w2 = 0.0, w2, s_wdp = 0.0, s_wdpe = 0.0, s_noi = 0.0;
                        for (int32_t i = 0; i < num; i++) {
                            w2 = 1 / (1 + expf(-v[i][5] + 5)) / max_noi;
                            s_mw += v[i][2] * v[i][3] * w2;
                            s_w += v[i][2];
                            s_wdp += v[i][2] * w2 * v[i][4] / v[i][6];
                            s_wdpe += v[i][2] * w2 * v[i][5] / v[i][6];
                            s_noi += v[i][6];
                            s_w2 += v[i][2] * w2 / v[i][6];
                        }
                        (void)s_mw; (void)s_w2;
                        if (s_w >= 1.0) {
                            const float fm = s_wdp / s_wdpe;
                            bx[g] = dpe[0] * fm; by[g] = dpe[1] * fm;
                            noi[g] = s_noi / num + 1;
                            processed++;
                        }
                    }
                for (int32_t i = 0; i < N; i++)
                    if (!isnan(bx[i]) && !isnan(by[i])) { dx[i] = bx[i]; dy[i] = by[i]; bx[i] = nanv; by[i] = nanv; }
            }
            unprocessed = 0;
            for (int32_t i = 0; i < N; i++)
                if ((isnan(dx[i]) || isnan(dy[i])) && nclus[i] != 0) unprocessed++;
        }
    }
    /* 3x3 smoothing of the interpolated interior points (:1623-1666) */
    for (int32_t cv = 1; cv < dimy - 1; cv++)
        for (int32_t cu = 1; cu < dimx - 1; cu++) {
            const int32_t g = cv * dimx + cu;
            if (dpf[g] < 0 && !isnan(dx[g] + dy[g])) {
                float nd = 0.0, sxs = 0.0, sys = 0.0;
                for (int32_t b = -1; b <= 1; b++)
                    for (int32_t a = -1; a <= 1; a++) {
                        const int32_t h = (cv + b) * dimx + cu + a;
                        if (!isnan(dx[h] + dy[h])) { sxs += dx[h]; sys += dy[h]; nd = nd + 1; }
                    }
                bx[g] = sxs / nd; by[g] = sys / nd;
            } else { bx[g] = dx[g]; by[g] = dy[g]; }
        }
    for (int32_t cv = 1; cv < dimy - 1; cv++)
        for (int32_t cu = 1; cu < dimx - 1; cu++) { const int32_t g = cv * dimx + cu; dx[g] = bx[g]; dy[g] = by[g]; }
    /* snap to the nearest cluster (:1680-1706) */
    for (int32_t g = 0; g < N; g++) {
        if (!(dpf[g] < 0 && nclus[g] != 0)) continue;
        float best = 1E+37; int32_t id = 0;
        const float *m = mvn + (size_t)g * Kmax * 5;
        for (int32_t c = 0; c < nclus[g]; c++) {
            const float d0 = dx[g] - m[5 * c], d1 = dy[g] - m[5 * c + 1];
            const float sq = d0 * d0 + d1 * d1;
            if (sq < best) { best = sq; id = c; }
        }
        dpf[g] = id; dx[g] = m[5 * id]; dy[g] = m[5 * id + 1];
    }
    free(bx); free(by); free(noi); free(v);
    return NOI;
}

/* =======================================================================================
 * N2  image pre-filter                                             MIMC_module.c:2517-2585
 * GMA_float_conv2: 2-D correlation with a small kernel over the interior, null DN (value+0.5 truncating to 0)
 * poisoning its whole stencil, then the whole plane shifted so that its minimum becomes 1 and the poisoned
 * pixels 0.  `out` is IN/OUT: the reference never writes the border rows/columns of `out`, yet its minimum
 * search (:2555-2565) and the shift of the right-hand border columns (:2568-2582, the loop runs to dimx_in)
 * read them -- whatever the caller's buffer holds there takes part (fresh large mallocs: zeros, T4).
 * ======================================================================================= */
void orc_float_conv2(const float *in, int32_t H, int32_t W, const float *kernel, int32_t kh, int32_t kw, float *out)
{
    const int32_t ox = kw / 2, oy = kh / 2;
    const float nanv = sqrt(-1.0);
    for (int32_t r = oy; r < H - oy; r++)
        for (int32_t c = ox; c < W - ox; c++) {
            float s = 0;
            for (int32_t a = 0; a < kh; a++)
                for (int32_t b = 0; b < kw; b++) {
                    const float v = in[(size_t)(r + a - oy) * W + c + b - ox];
                    const float dn = (int32_t)(v + 0.5) ? v : nanv;          /* f32 + f64 0.5, truncation */
                    s += dn * kernel[a * kw + b];
                }
            out[(size_t)r * W + c] = s;
        }
    float mn = 1e+37;
    for (size_t i = 0; i < (size_t)H * W; i++)
        if (out[i] < mn) mn = out[i];
    for (int32_t r = oy; r < H - oy; r++)
        for (int32_t c = ox; c < W; c++) {
            float *o = out + (size_t)r * W + c;
            if (isnan(*o)) *o = 0;
            else *o -= mn - 1;
        }
}


/* =======================================================================================
 * N4  control-point offset                                          MIMC_module.c:33-492
 * get_offset_image: slow a-priori grid points whose ocw[2] chip is mostly valid are the candidates; they are
 * shuffled (GMA_double_randperm_row :494-541, srand(time(NULL)) -> `seed` here) and processed in segments:
 * per point 16 matches (raw + 3 chip-local filters) x (ocw[1], ocw[2]) x (forward, swapped) with the common
 * (2*AW_CRE+1)^2 rectangular pivot set on a (2*ocw_chip+1)^2 chip, clustered; clusters holding >= 60 % of the 16
 * vote with their mean.  offset = rounded mean of the votes.  Returns 1 (ok), -1 (not enough CPs, as the
 * reference), -3 (a candidate's chip would leave the image: the reference reads out of bounds there).
 * info[0]=#candidates, [1]=#CP threshold, [2]=#segments run, [3]=#CP found; sduv[2] = the vote sums.
 * ======================================================================================= */
int orc_get_offset_image(const float *i0, const float *i1, int32_t H, int32_t W, const double *xyuvav, int32_t N,
                         const int32_t *vec_ocw, float aw_cre, int32_t num_cp_max, int32_t num_cp_min, float ratio_cp,
                         float thres_spd_cp, const float *const *kern, const int32_t *kdim, uint32_t seed,
                         int32_t *offset, uint8_t *flag_cp, int32_t *info, float *sduv_out)
{
    const int ocw2 = vec_ocw[2];
    const int ocw_chip = vec_ocw[2] + aw_cre + 2;                 /* int + float -> float -> int (:51) */
    const int cs = 2 * ocw_chip + 1, ts = cs + 2;
    int32_t num_cp;
    if (N * ratio_cp > num_cp_max) num_cp = num_cp_max; else num_cp = (int32_t)(N * ratio_cp);
    uint8_t *cand = (uint8_t *)malloc(N);
    int32_t ncand = 0;
    for (int32_t g = 0; g < N; g++) {
        float spd = xyuvav[6 * (size_t)g + 4] * xyuvav[6 * (size_t)g + 4] + xyuvav[6 * (size_t)g + 5] * xyuvav[6 * (size_t)g + 5];
        cand[g] = spd < thres_spd_cp * thres_spd_cp;
        ncand += cand[g];
    }
    const int32_t thres_numpx = (ocw2 * 2 + 1) * (ocw2 * 2 + 1) / 2;
    for (int32_t g = 0; g < N; g++) {
        if (!cand[g]) continue;
        const int u = (int32_t)xyuvav[6 * (size_t)g + 2], v = (int32_t)xyuvav[6 * (size_t)g + 3];
        if (u - ocw_chip - 1 < 0 || u + ocw_chip + 1 >= W || v - ocw_chip - 1 < 0 || v + ocw_chip + 1 >= H) { free(cand); return -3; }
        int32_t bad0 = 0;
        for (int a = -ocw2; a <= ocw2; a++)
            for (int b = -ocw2; b <= ocw2; b++)
                if (i0[(size_t)(a + v) * W + b + u] < 0.00001) bad0++;
        if (bad0 > thres_numpx) { cand[g] = 0; ncand--; }        /* :106 tests i0's count twice, i1's never */
    }
    if (info) { info[0] = ncand; info[1] = num_cp; info[2] = 0; info[3] = 0; }
    if (ncand < num_cp_min) { free(cand); return -1; }
    if (num_cp > ncand) num_cp = (int32_t)((float)ncand * 0.75);
    if (info) info[1] = num_cp;

    int32_t *order = (int32_t *)malloc(sizeof(int32_t) * ncand), *tmp = (int32_t *)malloc(sizeof(int32_t) * ncand);
    { int32_t n = 0; for (int32_t g = 0; g < N; g++) if (cand[g]) tmp[n++] = g; }
    free(cand);
    srand(seed);                                                  /* :517, row shuffle of :519-538 on the row ids */
    for (int32_t lim = ncand - 1; lim >= 0; lim--) {
        const int32_t idx = lim != 0 ? (int32_t)(rand() % lim) : 0;
        order[lim] = tmp[idx]; tmp[idx] = tmp[0]; tmp[0] = tmp[lim];
    }
    free(tmp);

    const int npiv_side = (int)(aw_cre * 2 + 1);
    const int32_t npiv = (int32_t)((aw_cre * 2 + 1) * (aw_cre * 2 + 1));
    int32_t *piv = (int32_t *)malloc(sizeof(int32_t) * 2 * npiv);
    { int32_t k = 0; for (int a = -(int)aw_cre; a <= (int)aw_cre; a++) for (int b = -(int)aw_cre; b <= (int)aw_cre; b++) { piv[2 * k] = a; piv[2 * k + 1] = b; k++; } (void)npiv_side; }

    const int32_t nseg = ncand < num_cp_min ? 1 : ncand / num_cp;
    float sduv[2] = {0.0f, 0.0f};
    int32_t ncur = 0, ok = 0, segs = 0;
    for (int32_t sgm = 0; sgm < nseg; sgm++) {
        const int32_t beg = (int32_t)(ncand * ((float)sgm / (float)nseg)), end = (int32_t)(ncand * ((float)(sgm + 1) / (float)nseg));
        const int32_t n = end - beg;
        segs++;
        float *dp = (float *)calloc((size_t)16 * n * 3, sizeof(float));
        float *c0 = (float *)malloc(sizeof(float) * (size_t)n * cs * cs), *c1 = (float *)malloc(sizeof(float) * (size_t)n * cs * cs);
        for (int kk = -1; kk <= 2; kk++) {
            if (kk < 0) {
                for (int32_t t = 0; t < n; t++) {
                    const int32_t g = order[beg + t];
                    const int u = (int32_t)xyuvav[6 * (size_t)g + 2], v = (int32_t)xyuvav[6 * (size_t)g + 3];
                    for (int r = 0; r < cs; r++)
                        for (int c = 0; c < cs; c++) {
                            c0[((size_t)t * cs + r) * cs + c] = i0[(size_t)(v - ocw_chip + r) * W + u - ocw_chip + c];
                            c1[((size_t)t * cs + r) * cs + c] = i1[(size_t)(v - ocw_chip + r) * W + u - ocw_chip + c];
                        }
                }
            } else {
                float *t0 = (float *)malloc(sizeof(float) * ts * ts), *t1 = (float *)malloc(sizeof(float) * ts * ts);
                float *o0 = (float *)calloc((size_t)ts * ts, sizeof(float)), *o1 = (float *)calloc((size_t)ts * ts, sizeof(float));  /* T4 */
                for (int32_t t = 0; t < n; t++) {                /* o0/o1 persist from point to point (:259-262) */
                    const int32_t g = order[beg + t];
                    const int u = (int32_t)xyuvav[6 * (size_t)g + 2], v = (int32_t)xyuvav[6 * (size_t)g + 3];
                    for (int r = 0; r < ts; r++)
                        for (int c = 0; c < ts; c++) {
                            t0[r * ts + c] = i0[(size_t)(v - ocw_chip - 1 + r) * W + u - ocw_chip - 1 + c];
                            t1[r * ts + c] = i1[(size_t)(v - ocw_chip - 1 + r) * W + u - ocw_chip - 1 + c];
                        }
                    orc_float_conv2(t0, ts, ts, kern[kk], kdim[2 * kk], kdim[2 * kk + 1], o0);
                    orc_float_conv2(t1, ts, ts, kern[kk], kdim[2 * kk], kdim[2 * kk + 1], o1);
                    for (int r = 0; r < cs; r++)
                        for (int c = 0; c < cs; c++) {
                            c0[((size_t)t * cs + r) * cs + c] = o0[(r + 1) * ts + c + 1];
                            c1[((size_t)t * cs + r) * cs + c] = o1[(r + 1) * ts + c + 1];
                        }
                }
                free(t0); free(t1); free(o0); free(o1);
            }
            for (int c3 = 1; c3 < 3; c3++) {
                const int ocw = vec_ocw[c3], cw = 2 * ocw + 1;
                const int32_t slot = (c3 - 1) * 8 + (kk + 1) * 2;
#pragma omp parallel
                {
                    float *ref = (float *)malloc(sizeof(float) * cw * cw);
                    float r3[3];
#pragma omp for schedule(dynamic)
                    for (int32_t t = 0; t < n; t++) {
                        const float *a0 = c0 + (size_t)t * cs * cs, *a1 = c1 + (size_t)t * cs * cs;
                        for (int r = 0; r < cw; r++)
                            memcpy(ref + r * cw, a0 + (size_t)(ocw_chip - ocw + r) * cs + ocw_chip - ocw, sizeof(float) * cw);
                        find_peak(ref, ocw, a1, cs, cs, piv, npiv, r3);
                        float *d = dp + ((size_t)slot * n + t) * 3;
                        d[0] = r3[0]; d[1] = r3[1]; d[2] = r3[2];
                        for (int r = 0; r < cw; r++)
                            memcpy(ref + r * cw, a1 + (size_t)(ocw_chip - ocw + r) * cs + ocw_chip - ocw, sizeof(float) * cw);
                        find_peak(ref, ocw, a0, cs, cs, piv, npiv, r3);
                        d = dp + ((size_t)(slot + 1) * n + t) * 3;
                        d[0] = -r3[0]; d[1] = -r3[1]; d[2] = r3[2];
                    }
                    free(ref);
                }
            }
        }
        free(c0); free(c1);
        float *mvn = (float *)malloc(sizeof(float) * 5 * (size_t)n * 16);
        int32_t *ncl = (int32_t *)malloc(sizeof(int32_t) * n);
        orc_cluster_candidates(dp, 16, n, 16, mvn, ncl);
        for (int32_t t = 0; t < n; t++)
            for (int32_t c = 0; c < ncl[t]; c++)
                if (mvn[((size_t)t * 16 + c) * 5 + 4] >= 0.6) {
                    sduv[0] += mvn[((size_t)t * 16 + c) * 5];
                    sduv[1] += mvn[((size_t)t * 16 + c) * 5 + 1];
                    flag_cp[order[beg + t]] = 1;
                    ncur++;
                }
        free(mvn); free(ncl); free(dp);
        if (num_cp <= ncur) { ok = 1; break; }
    }
    free(order); free(piv);
    if (info) { info[2] = segs; info[3] = ncur; }
    if (sduv_out) { sduv_out[0] = sduv[0]; sduv_out[1] = sduv[1]; }
    if (ncur < num_cp && ncur >= num_cp_min) ok = 1;
    if (!ok) return -1;
    const float du = sduv[0] / (float)ncur, dv = sduv[1] / (float)ncur;
    offset[0] = du > 0 ? (int32_t)(du + 0.5) : (int32_t)(du - 0.5);
    offset[1] = dv > 0 ? (int32_t)(dv + 0.5) : (int32_t)(dv - 0.5);
    return 1;
}
