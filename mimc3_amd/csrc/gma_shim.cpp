// gma_shim.cpp -- the reference's four hot-path entry points (exact names/signatures) over the
// flat C ABI of libmimc3_hip.so.  See include/mimc3_gma_shim.h.  Host-only translation unit.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../../include/mimc3_hip.h"
#include "../../include/mimc3_gma_shim.h"

// process globals owned by the reference's main() (MIMC_main.c:38-41)
extern "C" {
extern int32_t dimx_vmap, dimy_vmap, num_grid, num_dp;
extern float dt;
extern param param_mimc2;
}

namespace {

mimc3_ctx *g_ctx = nullptr;

[[noreturn]] void die(const char *where, int rc)
{
    std::fprintf(stderr, "mimc3 shim: %s failed (rc=%d): %s\n", where, rc, mimc3_last_error());
    std::abort();
}

mimc3_ctx *ctx()
{
    if (!g_ctx) {
        const char *dev = std::getenv("MIMC3_DEVICE");
        int rc = mimc3_ctx_create(dev ? std::atoi(dev) : 0, &g_ctx);
        if (rc) die("mimc3_ctx_create", rc);
    }
    return g_ctx;
}

template <class G, class T>
G *gma_create(int32_t nrows, int32_t ncols)   // same three malloc blocks as GMA_*_create (GMA.c:54-102)
{
    G *g = static_cast<G *>(std::malloc(sizeof(G)));
    g->nrows = nrows; g->ncols = ncols;
    g->val = static_cast<T **>(std::malloc(sizeof(T *) * (size_t)(nrows > 0 ? nrows : 1)));
    g->data = static_cast<T *>(std::malloc(sizeof(T) * (size_t)(nrows > 0 ? nrows : 1) * (size_t)(ncols > 0 ? ncols : 1)));
    for (int32_t r = 0; r < nrows; ++r) g->val[r] = g->data + (size_t)r * ncols;
    return g;
}

// the reference addresses arrays through val[r][c]; rows are contiguous in `data` for every array
// created by GMA_*_create, but be safe and gather through val[] when data is not the row base
template <class G, class T>
const T *flat(const G *g, std::vector<T> &tmp)
{
    bool contiguous = g->data != nullptr;
    for (int32_t r = 0; contiguous && r < g->nrows; ++r) contiguous = (g->val[r] == g->data + (size_t)r * g->ncols);
    if (contiguous) return g->data;
    tmp.resize((size_t)g->nrows * g->ncols);
    for (int32_t r = 0; r < g->nrows; ++r) std::memcpy(tmp.data() + (size_t)r * g->ncols, g->val[r], sizeof(T) * g->ncols);
    return tmp.data();
}

}  // namespace

extern "C" void mimc3_gma_shim_shutdown(void)
{
    if (g_ctx) mimc3_ctx_destroy(g_ctx);
    g_ctx = nullptr;
}

extern "C" GMA_int32 **get_uv_pivot(GMA_double *xyuvav, float dt, param p, int32_t ocw, GMA_float *i1)
{
    const int32_t N = xyuvav->nrows;
    std::vector<double> tmp;
    const double *xy = flat<GMA_double, double>(xyuvav, tmp);
    std::vector<int64_t> off((size_t)N + 1);
    int64_t total = 0;
    int rc = mimc3_get_uv_pivot(xy, N, dt, p.mpp, p.AW_SF, p.AW_CRE, ocw, i1->nrows, i1->ncols, off.data(), nullptr, 0, &total);
    if (rc) die("mimc3_get_uv_pivot", rc);
    std::vector<int32_t> uv(2 * (size_t)total);
    rc = mimc3_get_uv_pivot(xy, N, dt, p.mpp, p.AW_SF, p.AW_CRE, ocw, i1->nrows, i1->ncols, off.data(), uv.data(), total, &total);
    if (rc) die("mimc3_get_uv_pivot", rc);
    GMA_int32 **out = static_cast<GMA_int32 **>(std::malloc(sizeof(GMA_int32 *) * (size_t)N));
    for (int32_t g = 0; g < N; ++g) {
        const int32_t n = (int32_t)(off[g + 1] - off[g]);
        out[g] = gma_create<GMA_int32, int32_t>(n, 2);
        std::memcpy(out[g]->data, uv.data() + 2 * off[g], sizeof(int32_t) * 2 * (size_t)n);
    }
    std::printf("\n");   // the reference prints a newline here (MIMC_module.c:600)
    return out;
}

extern "C" GMA_float *matching_ncc_dlc_2(GMA_float *i0, GMA_float *i1, GMA_double *xyuvav, int32_t *offset,
                                         GMA_int32 **uv_pivot, int32_t ocw, float, float)
{
    const int32_t N = xyuvav->nrows;
    std::vector<float> t0, t1;
    std::vector<double> txy;
    // The CLI rewrites its filtered image buffers in place between passes (MIMC_main.c:309-310), so
    // pointer identity says nothing about content: upload on every call (2 x H x W x 4 bytes).
    int rc = mimc3_ctx_set_images(ctx(), flat<GMA_float, float>(i0, t0), flat<GMA_float, float>(i1, t1), i0->nrows, i0->ncols);
    if (rc) die("mimc3_ctx_set_images", rc);
    std::vector<int64_t> off((size_t)N + 1, 0);
    for (int32_t g = 0; g < N; ++g) off[g + 1] = off[g] + uv_pivot[g]->nrows;
    std::vector<int32_t> uv(2 * (size_t)off[N]);
    for (int32_t g = 0; g < N; ++g)
        for (int32_t k = 0; k < uv_pivot[g]->nrows; ++k) {
            uv[2 * (off[g] + k)] = uv_pivot[g]->val[k][0];
            uv[2 * (off[g] + k) + 1] = uv_pivot[g]->val[k][1];
        }
    GMA_float *out = gma_create<GMA_float, float>(N, 3);
    rc = mimc3_match_ncc_dlc(ctx(), flat<GMA_double, double>(xyuvav, txy), N, offset, uv.data(), off.data(), ocw, 0, out->data);
    if (rc) die("mimc3_match_ncc_dlc", rc);
    return out;
}

extern "C" GMA_int32 *get_ruv_neighbor(GMA_double *xyuvav, float radius_neighbor)
{
    std::vector<double> txy;
    const int32_t cap = 16384;
    std::vector<int32_t> ruv(2 * (size_t)cap);
    int32_t nn = 0;
    int rc = mimc3_get_ruv_neighbor(flat<GMA_double, double>(xyuvav, txy), xyuvav->nrows, dimx_vmap, dimy_vmap,
                                    param_mimc2.meter_per_spacing, radius_neighbor, ruv.data(), cap, &nn);
    if (rc) die("mimc3_get_ruv_neighbor", rc);
    GMA_int32 *out = gma_create<GMA_int32, int32_t>(nn, 2);
    std::memcpy(out->data, ruv.data(), sizeof(int32_t) * 2 * (size_t)nn);
    return out;
}

extern "C" void get_dpf_pseudosmoothing(GMA_int32 *dpf, GMA_float *dpf_dx, GMA_float *dpf_dy, GMA_int32 *ruv_neighbor,
                                        GMA_float **mvn_dp, GMA_double *xyuvav)
{
    const int32_t dimx = dimx_vmap, dimy = dimy_vmap, N = dimx * dimy;
    int32_t kmax = 1;
    for (int32_t g = 0; g < N; ++g) if (mvn_dp[g]->nrows > kmax) kmax = mvn_dp[g]->nrows;
    std::vector<float> mvn((size_t)N * kmax * 5, 0.0f);
    std::vector<int32_t> nclus((size_t)N);
    for (int32_t g = 0; g < N; ++g) {
        nclus[g] = mvn_dp[g]->nrows;
        for (int32_t c = 0; c < mvn_dp[g]->nrows; ++c)
            std::memcpy(mvn.data() + ((size_t)g * kmax + c) * 5, mvn_dp[g]->val[c], sizeof(float) * 5);
    }
    std::vector<int32_t> tr, td;
    std::vector<float> tx, ty;
    std::vector<double> txy;
    // in-place contract: work on contiguous copies when the caller's rows are not contiguous
    const int32_t *ruv = flat<GMA_int32, int32_t>(ruv_neighbor, tr);
    std::vector<int32_t> d((size_t)N);
    std::vector<float> x((size_t)N), y((size_t)N);
    for (int32_t r = 0; r < dimy; ++r) {
        std::memcpy(d.data() + (size_t)r * dimx, dpf->val[r], sizeof(int32_t) * dimx);
        std::memcpy(x.data() + (size_t)r * dimx, dpf_dx->val[r], sizeof(float) * dimx);
        std::memcpy(y.data() + (size_t)r * dimx, dpf_dy->val[r], sizeof(float) * dimx);
    }
    int32_t sweeps = 0;
    int rc = mimc3_qm_pseudosmooth(ctx(), dimy, dimx, d.data(), x.data(), y.data(), ruv, ruv_neighbor->nrows, mvn.data(), kmax,
                                   nclus.data(), flat<GMA_double, double>(xyuvav, txy), 101 /* NOI<=100, :2077 */, &sweeps);
    if (rc) die("mimc3_qm_pseudosmooth", rc);
    for (int32_t r = 0; r < dimy; ++r) {
        std::memcpy(dpf->val[r], d.data() + (size_t)r * dimx, sizeof(int32_t) * dimx);
        std::memcpy(dpf_dx->val[r], x.data() + (size_t)r * dimx, sizeof(float) * dimx);
        std::memcpy(dpf_dy->val[r], y.data() + (size_t)r * dimx, sizeof(float) * dimx);
    }
    std::printf("pseudosmoothing on device: NOI=%d\n", sweeps);
}

// ---------------------------------------------------------------------------------------------
// the stages either side of the hot path (N1, N2, N4) and the post-processing chain (N3)
// ---------------------------------------------------------------------------------------------
namespace {

// padded [N][K][5] + nclus[N] from the reference's ragged mvn_dp
void pad_mvn(GMA_float **mvn_dp, int32_t N, int32_t &K, std::vector<float> &mvn, std::vector<int32_t> &nclus)
{
    K = 1;
    for (int32_t g = 0; g < N; ++g) if (mvn_dp[g]->nrows > K) K = mvn_dp[g]->nrows;
    mvn.assign((size_t)N * K * 5, 0.0f);
    nclus.resize((size_t)N);
    for (int32_t g = 0; g < N; ++g) {
        nclus[g] = mvn_dp[g]->nrows;
        for (int32_t c = 0; c < mvn_dp[g]->nrows; ++c)
            std::memcpy(mvn.data() + ((size_t)g * K + c) * 5, mvn_dp[g]->val[c], sizeof(float) * 5);
    }
}

// pass-major [ndp][N][3] from the reference's array of per-pass outputs
void gather_dp(GMA_float **dp, int32_t ndp, int32_t N, std::vector<float> &out)
{
    out.resize((size_t)ndp * N * 3);
    for (int32_t k = 0; k < ndp; ++k)
        for (int32_t g = 0; g < N; ++g) std::memcpy(out.data() + ((size_t)k * N + g) * 3, dp[k]->val[g], sizeof(float) * 3);
}

}  // namespace

extern "C" int get_offset_image(GMA_float *i0, GMA_float *i1, GMA_float **kernel, GMA_double *xyuvav, int32_t *offset, GMA_uint8 *flag_cp)
{
    const int32_t N = xyuvav->nrows;
    std::vector<float> t0, t1, tk[3];
    std::vector<double> txy;
    int rc = mimc3_ctx_set_images(ctx(), flat<GMA_float, float>(i0, t0), flat<GMA_float, float>(i1, t1), i0->nrows, i0->ncols);
    if (rc) die("mimc3_ctx_set_images", rc);
    mimc3_cp_params p{};
    for (int k = 0; k < 4; ++k) p.vec_ocw[k] = param_mimc2.vec_ocw[k];
    p.aw_cre = param_mimc2.AW_CRE; p.num_cp_max = param_mimc2.num_cp_max; p.num_cp_min = param_mimc2.num_cp_min;
    p.ratio_cp = param_mimc2.ratio_cp; p.thres_spd_cp = param_mimc2.thres_spd_cp;
    for (int k = 0; k < 3; ++k) {
        p.kernel[k] = flat<GMA_float, float>(kernel[k], tk[k]);
        p.kdim[k][0] = kernel[k]->nrows; p.kdim[k][1] = kernel[k]->ncols;
    }
    const char *seed = std::getenv("MIMC3_CP_SEED");     // the reference seeds its shuffle with time(NULL) (:517)
    p.seed = seed ? std::atoll(seed) : -1;
    std::vector<uint8_t> flag((size_t)N, 0);
    int32_t status = -1, info[4] = {0, 0, 0, 0};
    rc = mimc3_get_offset_image(ctx(), flat<GMA_double, double>(xyuvav, txy), N, &p, offset, flag.data(), &status, info, nullptr);
    if (rc) die("mimc3_get_offset_image", rc);
    for (int32_t g = 0; g < N; ++g) if (flag[g]) flag_cp->val[g][0] = 1;
    std::printf("CP stage on device: %d candidates, %d control points, status %d\n", info[0], info[3], status);
    return status;
}

extern "C" GMA_float **calc_mean_var_num_dp_cluster(GMA_float **dp, int32_t num_dpoi)
{
    const int32_t N = dp[0]->nrows, K = num_dpoi;
    std::vector<float> flatdp, mvn((size_t)N * K * 5);
    std::vector<int32_t> nclus((size_t)N);
    gather_dp(dp, num_dpoi, N, flatdp);
    int32_t seen = 0;
    int rc = mimc3_cluster_candidates(ctx(), flatdp.data(), num_dpoi, N, K, mvn.data(), nclus.data(), &seen);
    if (rc) die("mimc3_cluster_candidates", rc);
    GMA_float **out = static_cast<GMA_float **>(std::malloc(sizeof(GMA_float *) * (size_t)N));
    for (int32_t g = 0; g < N; ++g) {
        out[g] = gma_create<GMA_float, float>(nclus[g], 5);
        std::memcpy(out[g]->data, mvn.data() + (size_t)g * K * 5, sizeof(float) * 5 * (size_t)nclus[g]);
    }
    return out;
}

extern "C" GMA_int32 *get_dpf0(GMA_float **mvn_dp, float min_matching_ratio)
{
    const int32_t N = dimx_vmap * dimy_vmap;
    int32_t K;
    std::vector<float> mvn;
    std::vector<int32_t> nclus;
    pad_mvn(mvn_dp, N, K, mvn, nclus);
    GMA_int32 *out = gma_create<GMA_int32, int32_t>(dimy_vmap, dimx_vmap);
    int rc = mimc3_get_dpf0(ctx(), mvn.data(), nclus.data(), N, K, min_matching_ratio, out->data);
    if (rc) die("mimc3_get_dpf0", rc);
    return out;
}

extern "C" void get_dpf1(GMA_int32 *dpf0, GMA_float *dpf_dx, GMA_float *dpf_dy, GMA_int32 *ruv_neighbor, GMA_float **mvn_dp,
                         GMA_double *xyuvav)
{
    const int32_t dimx = dimx_vmap, dimy = dimy_vmap, N = dimx * dimy;
    int32_t K;
    std::vector<float> mvn;
    std::vector<int32_t> nclus, tr;
    std::vector<double> txy;
    pad_mvn(mvn_dp, N, K, mvn, nclus);
    std::vector<int32_t> d((size_t)N);
    std::vector<float> x((size_t)N), y((size_t)N);
    for (int32_t r = 0; r < dimy; ++r) std::memcpy(d.data() + (size_t)r * dimx, dpf0->val[r], sizeof(int32_t) * dimx);
    int32_t sweeps = 0;
    int rc = mimc3_get_dpf1(ctx(), dimy, dimx, d.data(), x.data(), y.data(), flat<GMA_int32, int32_t>(ruv_neighbor, tr), ruv_neighbor->nrows,
                            mvn.data(), K, nclus.data(), flat<GMA_double, double>(xyuvav, txy), dt, param_mimc2.mpp, &sweeps);
    if (rc) die("mimc3_get_dpf1", rc);
    for (int32_t r = 0; r < dimy; ++r) {
        std::memcpy(dpf0->val[r], d.data() + (size_t)r * dimx, sizeof(int32_t) * dimx);
        std::memcpy(dpf_dx->val[r], x.data() + (size_t)r * dimx, sizeof(float) * dimx);
        std::memcpy(dpf_dy->val[r], y.data() + (size_t)r * dimx, sizeof(float) * dimx);
    }
}

extern "C" void GMA_float_conv2(GMA_float *in, GMA_float *kernel, GMA_float *out)
{
    std::vector<float> ti, tk, to;
    const float *src = flat<GMA_float, float>(in, ti);
    const float *ker = flat<GMA_float, float>(kernel, tk);
    // `out` is in/out (its border is read): work on a contiguous copy when the caller's rows are not contiguous
    bool contiguous = true;
    for (int32_t r = 0; contiguous && r < out->nrows; ++r) contiguous = (out->val[r] == out->data + (size_t)r * out->ncols);
    float *dst = out->data;
    if (!contiguous) {
        to.resize((size_t)out->nrows * out->ncols);
        for (int32_t r = 0; r < out->nrows; ++r) std::memcpy(to.data() + (size_t)r * out->ncols, out->val[r], sizeof(float) * out->ncols);
        dst = to.data();
    }
    int rc = mimc3_float_conv2(ctx(), src, in->nrows, in->ncols, ker, kernel->nrows, kernel->ncols, dst);
    if (rc) die("mimc3_float_conv2", rc);
    if (!contiguous)
        for (int32_t r = 0; r < out->nrows; ++r) std::memcpy(out->val[r], to.data() + (size_t)r * out->ncols, sizeof(float) * out->ncols);
}

extern "C" GMA_float **mimc2_postprocess(GMA_float **dp, GMA_double *xyuvav, float dt_)
{
    const int32_t dimx = dimx_vmap, dimy = dimy_vmap, N = dimx * dimy;
    std::vector<float> flatdp, out5((size_t)5 * N);
    std::vector<double> txy;
    gather_dp(dp, num_dp, N, flatdp);
    int rc = mimc3_postprocess(ctx(), flatdp.data(), num_dp, flat<GMA_double, double>(xyuvav, txy), dimx, dimy, dt_, param_mimc2.mpp,
                               param_mimc2.meter_per_spacing, param_mimc2.radius_neighbor_dpf1, param_mimc2.radius_neighbor_ps, 101,
                               out5.data());
    if (rc) die("mimc3_postprocess", rc);
    GMA_float **v = static_cast<GMA_float **>(std::malloc(sizeof(GMA_float *) * 5));
    for (int k = 0; k < 5; ++k) {
        v[k] = gma_create<GMA_float, float>(dimy, dimx);
        std::memcpy(v[k]->data, out5.data() + (size_t)k * N, sizeof(float) * (size_t)N);
    }
    return v;
}
