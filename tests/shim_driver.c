/* tests/shim_driver.c -- plain-C caller that uses the hot path the way the reference's main()
 * does (MIMC_main.c:261-300, MIMC_module.c:933-934), but through include/mimc3_gma_shim.h.
 * Built and run by tests/test_gma_shim.py on the GPU box.  Binary I/O only.
 *
 *   shim_driver match <in.bin> <out.bin>     forward + swapped pass as in main()
 *   shim_driver qm    <in.bin> <out.bin>     get_ruv_neighbor + get_dpf_pseudosmoothing
 *   shim_driver n1    <in.bin> <out.bin>     calc_mean_var_num_dp_cluster + get_dpf0 + get_dpf1 as mimc2_postprocess
 *                                            chains them (MIMC_module.c:904-926), and GMA_float_conv2 on a small plane
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "mimc3_gma_shim.h"

/* the process globals the reference's main() owns (MIMC_main.c:38-41) */
float dt;
int32_t num_dp;
int32_t num_grid, dimx_vmap, dimy_vmap;
param param_mimc2;

static void rd(FILE *f, void *p, size_t n) { if (fread(p, 1, n, f) != n) { fprintf(stderr, "short read\n"); exit(2); } }

static GMA_float *mk_float(int32_t nr, int32_t nc)
{
    GMA_float *g = malloc(sizeof *g);
    g->nrows = nr; g->ncols = nc;
    g->val = malloc(sizeof(float *) * (size_t)(nr > 0 ? nr : 1));
    g->data = malloc(sizeof(float) * (size_t)(nr > 0 ? nr : 1) * nc);
    for (int32_t r = 0; r < nr; r++) g->val[r] = g->data + (size_t)r * nc;
    return g;
}
static GMA_double *mk_double(int32_t nr, int32_t nc)
{
    GMA_double *g = malloc(sizeof *g);
    g->nrows = nr; g->ncols = nc;
    g->val = malloc(sizeof(double *) * (size_t)nr);
    g->data = malloc(sizeof(double) * (size_t)nr * nc);
    for (int32_t r = 0; r < nr; r++) g->val[r] = g->data + (size_t)r * nc;
    return g;
}
static GMA_int32 *mk_int32(int32_t nr, int32_t nc)
{
    GMA_int32 *g = malloc(sizeof *g);
    g->nrows = nr; g->ncols = nc;
    g->val = malloc(sizeof(int32_t *) * (size_t)nr);
    g->data = malloc(sizeof(int32_t) * (size_t)nr * nc);
    for (int32_t r = 0; r < nr; r++) g->val[r] = g->data + (size_t)r * nc;
    return g;
}

static int run_match(FILE *fi, FILE *fo)
{
    int32_t h[6]; float fl[4];
    rd(fi, h, sizeof h); rd(fi, fl, sizeof fl);
    const int32_t H = h[0], W = h[1], N = h[2], ocw = h[3];
    int32_t off[2] = { h[4], h[5] }, roff[2] = { -h[4], -h[5] };
    dt = fl[0]; param_mimc2.mpp = fl[1]; param_mimc2.AW_SF = fl[2]; param_mimc2.AW_CRE = fl[3];
    GMA_float *i0 = mk_float(H, W), *i1 = mk_float(H, W);
    GMA_double *xy = mk_double(N, 6);
    rd(fi, i0->data, sizeof(float) * (size_t)H * W);
    rd(fi, i1->data, sizeof(float) * (size_t)H * W);
    rd(fi, xy->data, sizeof(double) * 6 * (size_t)N);
    num_grid = N;
    GMA_int32 **piv = get_uv_pivot(xy, dt, param_mimc2, ocw, i1);
    GMA_float *fwd = matching_ncc_dlc_2(i0, i1, xy, off, piv, ocw, param_mimc2.AW_CRE, param_mimc2.AW_SF);
    for (int32_t g = 0; g < N; g++)                       /* reverse the pivots in place (:272-279) */
        for (int32_t k = 0; k < piv[g]->nrows; k++) { piv[g]->val[k][0] = -piv[g]->val[k][0]; piv[g]->val[k][1] = -piv[g]->val[k][1]; }
    GMA_float *swp = matching_ncc_dlc_2(i1, i0, xy, roff, piv, ocw, param_mimc2.AW_CRE, param_mimc2.AW_SF);
    for (int32_t g = 0; g < N; g++) { swp->val[g][0] = -swp->val[g][0]; swp->val[g][1] = -swp->val[g][1]; }   /* :289-293 */
    fwrite(fwd->data, sizeof(float), 3 * (size_t)N, fo);
    fwrite(swp->data, sizeof(float), 3 * (size_t)N, fo);
    int64_t tot = 0;
    for (int32_t g = 0; g < N; g++) tot += piv[g]->nrows;
    fwrite(&tot, sizeof tot, 1, fo);
    return 0;
}

static int run_qm(FILE *fi, FILE *fo)
{
    int32_t h[3]; float fl[2];
    rd(fi, h, sizeof h); rd(fi, fl, sizeof fl);
    const int32_t dimx = h[0], dimy = h[1], kmax = h[2], N = dimx * dimy;
    dimx_vmap = dimx; dimy_vmap = dimy; num_grid = N;
    param_mimc2.meter_per_spacing = fl[0];
    GMA_int32 *dpf = mk_int32(dimy, dimx);
    GMA_float *dx = mk_float(dimy, dimx), *dy = mk_float(dimy, dimx);
    GMA_double *xy = mk_double(N, 6);
    float *mvn = malloc(sizeof(float) * 5 * (size_t)N * kmax);
    int32_t *nclus = malloc(sizeof(int32_t) * (size_t)N);
    rd(fi, dpf->data, 4 * (size_t)N); rd(fi, dx->data, 4 * (size_t)N); rd(fi, dy->data, 4 * (size_t)N);
    rd(fi, mvn, sizeof(float) * 5 * (size_t)N * kmax); rd(fi, nclus, 4 * (size_t)N); rd(fi, xy->data, 48 * (size_t)N);
    GMA_float **mvn_dp = malloc(sizeof(GMA_float *) * (size_t)N);
    for (int32_t g = 0; g < N; g++) {
        mvn_dp[g] = mk_float(nclus[g], 5);
        memcpy(mvn_dp[g]->data, mvn + (size_t)g * kmax * 5, sizeof(float) * 5 * (size_t)nclus[g]);
    }
    GMA_int32 *ruv = get_ruv_neighbor(xy, fl[1]);
    get_dpf_pseudosmoothing(dpf, dx, dy, ruv, mvn_dp, xy);
    fwrite(dpf->data, 4, (size_t)N, fo); fwrite(dx->data, 4, (size_t)N, fo); fwrite(dy->data, 4, (size_t)N, fo);
    fwrite(&ruv->nrows, 4, 1, fo);
    return 0;
}

static int run_n1(FILE *fi, FILE *fo)
{
    int32_t h[5]; float fl[4];
    rd(fi, h, sizeof h); rd(fi, fl, sizeof fl);
    const int32_t dimx = h[0], dimy = h[1], ndp = h[2], ch = h[3], cw = h[4], N = dimx * dimy;
    dimx_vmap = dimx; dimy_vmap = dimy; num_grid = N; num_dp = ndp;
    param_mimc2.meter_per_spacing = fl[0]; dt = fl[2]; param_mimc2.mpp = fl[3];
    GMA_float **dp = malloc(sizeof(GMA_float *) * (size_t)ndp);
    for (int32_t k = 0; k < ndp; k++) { dp[k] = mk_float(N, 3); rd(fi, dp[k]->data, 12 * (size_t)N); }
    GMA_double *xy = mk_double(N, 6);
    rd(fi, xy->data, 48 * (size_t)N);
    GMA_float **mvn_dp = calc_mean_var_num_dp_cluster(dp, ndp);
    GMA_int32 *dpf = get_dpf0(mvn_dp, 0.6f);
    fwrite(dpf->data, 4, (size_t)N, fo);
    for (int32_t g = 0; g < N; g++) fwrite(&mvn_dp[g]->nrows, 4, 1, fo);
    for (int32_t g = 0; g < N; g++) fwrite(mvn_dp[g]->data, 4, 5 * (size_t)mvn_dp[g]->nrows, fo);
    GMA_int32 *ruv = get_ruv_neighbor(xy, fl[1]);
    GMA_float *dx = mk_float(dimy, dimx), *dy = mk_float(dimy, dimx);
    get_dpf1(dpf, dx, dy, ruv, mvn_dp, xy);
    fwrite(dpf->data, 4, (size_t)N, fo); fwrite(dx->data, 4, (size_t)N, fo); fwrite(dy->data, 4, (size_t)N, fo);
    /* pre-filter on a ch x cw plane with a 3x3 kernel; `out` comes in dirty (its border is read) */
    GMA_float *img = mk_float(ch, cw), *ker = mk_float(3, 3), *out = mk_float(ch, cw);
    rd(fi, img->data, 4 * (size_t)ch * cw); rd(fi, ker->data, 36); rd(fi, out->data, 4 * (size_t)ch * cw);
    GMA_float_conv2(img, ker, out);
    fwrite(out->data, 4, (size_t)ch * cw, fo);
    return 0;
}

int main(int argc, char **argv)
{
    if (argc != 4) { fprintf(stderr, "usage: %s match|qm|n1 in out\n", argv[0]); return 2; }
    FILE *fi = fopen(argv[2], "rb"), *fo = fopen(argv[3], "wb");
    if (!fi || !fo) { perror("open"); return 2; }
    int rc = strcmp(argv[1], "match") == 0 ? run_match(fi, fo) : (strcmp(argv[1], "n1") == 0 ? run_n1(fi, fo) : run_qm(fi, fo));
    fclose(fi); fclose(fo);
    mimc3_gma_shim_shutdown();
    return rc;
}
