// pipeline.cpp -- the reference program's data path on resident arrays, built ONLY from the public C ABI
// (include/mimc3_hip.h) plus the HIP runtime for buffers:
//
//   mimc3_postprocess[_dev]  = mimc2_postprocess (MIMC_module.c:892-990): clustering -> dpf0 -> dpf1 -> QM -> planes
//   mimc3_vmap               = MIMC_main.c:203-402 between "xyuvav and images loaded" and "save the output":
//                              grid geometry, CP offset, the 32 matcher passes (4 chip sizes x {raw, d/dx, d/dy,
//                              Laplacian} x {forward, swapped}), postprocess, sub-integer CP mean removal, px -> m/yr
//
// Everything that touches pixels or candidates runs on the device; the host keeps what the reference does serially
// in f32 (the mean over the grid, :362-378) so that the sums round identically.  No CPU fallback.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>
#include <chrono>
#include <thread>
#include <cstdio>
#include <cstdlib>
#include "../../include/mimc3_hip.h"
#include "host_util.h"
#include "pipeline_internal.h"

namespace {

// context scratch slots (mimc3_ctx_workspace) used by the drivers in this file and in mgpu.cpp
enum { kSlotPost = 0, kSlotXy = 1, kSlotOut5 = 3, kSlotDp = 4, kSlotXyFull = 5, kSlotPiv0 = 11 /* ..14: one per chip size */, kSlotPivOff0 = 16 /* ..19 */ };

// a typed view of a piece of context scratch
struct View {
    void *p = nullptr;
    template <class T> T *as() const { return static_cast<T *>(p); }
};

struct Buf {
    void *p = nullptr;
    ~Buf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 1); }
    template <class T> T *as() const { return static_cast<T *>(p); }
};

// MIMC3_VMAP_TIMING=1: wall time of each stage of mimc3_vmap on stderr (the stream is drained at each mark)
struct StageClock {
    bool on = getenv("MIMC3_VMAP_TIMING") != nullptr;
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    void mark(const char *what, hipStream_t s)
    {
        if (!on) return;
        (void)hipStreamSynchronize(s);
        const auto n = std::chrono::steady_clock::now();
        fprintf(stderr, "[mimc3 vmap] %-28s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(n - t).count());
        t = n;
    }
};

int hip_fail(hipError_t e, const char *what)
{
    return mimc3::fail((int)e > 0 ? (int)e : MIMC3_ENODEV, std::string(what) + ": " + hipGetErrorString(e));
}

}  // namespace

#define HIP_TRY(expr)                                           \
    do {                                                        \
        hipError_t e_ = (expr);                                 \
        if (e_ != hipSuccess) return hip_fail(e_, #expr);       \
    } while (0)
#define RC_TRY(expr)                \
    do {                            \
        int rc_ = (expr);           \
        if (rc_) return rc_;        \
    } while (0)

extern "C" int mimc3_postprocess_dev(mimc3_ctx *ctx, const float *d_dp, int32_t ndp, const double *xyuvav, const double *d_xyuvav,
                                     int32_t dimx, int32_t dimy, float dt, float mpp, float meter_per_spacing,
                                     float radius_dpf1, float radius_ps, int32_t qm_max_sweeps, float *d_out5, void *stream)
{
    if (!ctx || !d_dp || !xyuvav || !d_xyuvav || !d_out5 || ndp < 1 || ndp > 64 || dimx < 1 || dimy < 1 || qm_max_sweeps < 1)
        return mimc3::fail(MIMC3_EINVAL, "mimc3_postprocess_dev: bad argument");
    const int32_t N = dimx * dimy, K = ndp;
    hipStream_t s = static_cast<hipStream_t>(stream);
    // neighbour offsets for the two radii (host geometry, get_ruv_neighbor :1266-1327)
    std::vector<int32_t> ruv1(2 * 4096), ruv2(2 * 4096);
    int32_t nn1 = 0, nn2 = 0;
    RC_TRY(mimc3_get_ruv_neighbor(xyuvav, N, dimx, dimy, meter_per_spacing, radius_dpf1, ruv1.data(), 4096, &nn1));
    RC_TRY(mimc3_get_ruv_neighbor(xyuvav, N, dimx, dimy, meter_per_spacing, radius_ps, ruv2.data(), 4096, &nn2));
    // working buffers live in ONE context-owned scratch slot, carved here: no allocation after the first call
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t b_mvn = al(sizeof(float) * 5 * (size_t)N * K), b_n = al(sizeof(int32_t) * (size_t)N), b_r1 = al(8 * (size_t)(nn1 > 0 ? nn1 : 1)),
                 b_r2 = al(8 * (size_t)(nn2 > 0 ? nn2 : 1)), b_w1 = al((size_t)mimc3_dpf1_workspace_bytes(N)),
                 b_w2 = al((size_t)mimc3_qm_workspace_bytes(N, qm_max_sweeps));
    void *base = nullptr;
    RC_TRY(mimc3_ctx_workspace(ctx, kSlotPost, b_mvn + 4 * b_n + 256 + b_r1 + b_r2 + b_w1 + b_w2, &base));
    char *cur = static_cast<char *>(base);
    auto take = [&](size_t b) { View v{cur}; cur += b; return v; };
    View mvn = take(b_mvn), ncl = take(b_n), kmax = take(256), dpf = take(b_n), dx = take(b_n), dy = take(b_n), r1 = take(b_r1),
         r2 = take(b_r2), w1 = take(b_w1), w2 = take(b_w2);
    HIP_TRY(hipMemcpyAsync(r1.p, ruv1.data(), 8 * (size_t)nn1, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(r2.p, ruv2.data(), 8 * (size_t)nn2, hipMemcpyHostToDevice, s));
    RC_TRY(mimc3_cluster_candidates_dev(ctx, d_dp, ndp, N, K, mvn.as<float>(), ncl.as<int32_t>(), kmax.as<int32_t>(), s));   // :904
    RC_TRY(mimc3_get_dpf0_dev(ctx, mvn.as<float>(), ncl.as<int32_t>(), N, K, 0.6f, dpf.as<int32_t>(), s));                    // :912
    int32_t sweeps = 0;
    RC_TRY(mimc3_get_dpf1_dev(ctx, dimy, dimx, dpf.as<int32_t>(), dx.as<float>(), dy.as<float>(), r1.as<int32_t>(), nn1, mvn.as<float>(), K,
                              ncl.as<int32_t>(), d_xyuvav, dt, mpp, w1.p, &sweeps, s));                                         // :926
    RC_TRY(mimc3_qm_pseudosmooth_dev(ctx, dimy, dimx, dpf.as<int32_t>(), dx.as<float>(), dy.as<float>(), r2.as<int32_t>(), nn2,
                                     mvn.as<float>(), K, ncl.as<int32_t>(), d_xyuvav, qm_max_sweeps, w2.p, nullptr, s));        // :933
    RC_TRY(mimc3_dpf_to_vxyexyqual_dev(ctx, dpf.as<int32_t>(), mvn.as<float>(), N, K, d_out5, s));                             // :937-970
    HIP_TRY(hipStreamSynchronize(s));
    return 0;
}

extern "C" int mimc3_postprocess(mimc3_ctx *ctx, const float *dp, int32_t ndp, const double *xyuvav, int32_t dimx, int32_t dimy,
                                 float dt, float mpp, float meter_per_spacing, float radius_dpf1, float radius_ps,
                                 int32_t qm_max_sweeps, float *out5)
{
    if (!ctx || !dp || !xyuvav || !out5 || ndp < 1 || dimx < 1 || dimy < 1) return mimc3::fail(MIMC3_EINVAL, "mimc3_postprocess: bad argument");
    const size_t N = (size_t)dimx * dimy;
    hipStream_t s = static_cast<hipStream_t>(mimc3_ctx_stream(ctx));
    Buf d_dp, d_xy, d_o;
    HIP_TRY(d_dp.alloc(12 * N * ndp));
    HIP_TRY(d_xy.alloc(48 * N));
    HIP_TRY(d_o.alloc(20 * N));
    HIP_TRY(hipMemcpyAsync(d_dp.p, dp, 12 * N * ndp, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(d_xy.p, xyuvav, 48 * N, hipMemcpyHostToDevice, s));
    RC_TRY(mimc3_postprocess_dev(ctx, d_dp.as<float>(), ndp, xyuvav, d_xy.as<double>(), dimx, dimy, dt, mpp, meter_per_spacing, radius_dpf1,
                                 radius_ps, qm_max_sweeps, d_o.as<float>(), s));
    HIP_TRY(hipMemcpyAsync(out5, d_o.p, 20 * N, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return 0;
}

namespace mimc3 {

// grid geometry (MIMC_main.c:209-223)
int vmap_geometry(const double *xyuvav, int32_t N, mimc3_vmap_result *res)
{
    int32_t g = 1;
    for (; g < N; g++)
        if ((int)xyuvav[6 * (size_t)g + 2] == (int)xyuvav[2]) break;
    res->dimx = g; res->dimy = N / g;
    res->mpp = (float)((xyuvav[6] - xyuvav[0]) / (xyuvav[8] - xyuvav[2]));
    res->spacing_grid = (float)(xyuvav[8] - xyuvav[2]);
    res->meter_per_spacing = (float)(xyuvav[6] - xyuvav[0]);
    if (res->dimx * res->dimy != N) return mimc3::fail(MIMC3_EINVAL, "mimc3_vmap: xyuvav is not a full dimy x dimx grid");
    return 0;
}

HostPivots::~HostPivots() {}     // everything lives in the context's scratch slots

// pivots of the four chip sizes (:264, :316) for the points xs[0..ns).  The corridor of a point (atan2 / cos / sin: libm,
// host, threaded, once for all four chip sizes) depends on nothing the CP stage produces; 24 bytes per point are uploaded and
// the lists -- forward and negated (:272-279) -- are expanded on the device (pivot_kernel.hip), on an auxiliary stream, so that
// all of it overlaps the control-point stage.  Safe to call from a worker thread; the error text is returned because
// mimc3_last_error() is thread-local.
int vmap_host_pivots(mimc3_ctx *ctx, const double *xs, int32_t ns, float dt, float mpp, const mimc3_vmap_params *p, int32_t H, int32_t W,
                     HostPivots hp[4], std::string &err)
{
    (void)H; (void)W;                                        // (the context's image size bounds the pivots)
    if (ns <= 0) return 0;
    static const bool tm = getenv("MIMC3_VMAP_TIMING") != nullptr;
    const auto t_begin = std::chrono::steady_clock::now();
    auto lap = [&](int c, const char *w) { if (tm) fprintf(stderr, "[mimc3 piv %d] %-10s at %.2f ms\n", c, w, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count()); };
    auto body = [&]() -> int {
        auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
        void *hcor = nullptr, *base = nullptr;
        RC_TRY(mimc3_ctx_host_workspace(ctx, 0, MIMC3_CORRIDOR_BYTES * (size_t)ns, &hcor));          // pinned, kept across calls
        RC_TRY(mimc3_pivot_corridors(xs, ns, dt, mpp, p->aw_sf, p->aw_cre, hcor));
        lap(-1, "corridors");
        RC_TRY(mimc3_ctx_workspace(ctx, kSlotXy, al(48 * (size_t)ns) + al(MIMC3_CORRIDOR_BYTES * (size_t)ns), &base));   // (also selects the context's device for this thread)
        double *d_xy = static_cast<double *>(base);
        void *d_cor = static_cast<char *>(base) + al(48 * (size_t)ns);
        hipStream_t st = static_cast<hipStream_t>(mimc3_ctx_aux_stream(ctx, 0));
        if (!st) return mimc3::fail(MIMC3_ESTATE, "mimc3_vmap: the context has no auxiliary stream");
        HIP_TRY(hipMemcpyAsync(d_xy, xs, 48 * (size_t)ns, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(d_cor, hcor, MIMC3_CORRIDOR_BYTES * (size_t)ns, hipMemcpyHostToDevice, st));
        for (int c = 0; c < 4; c++) {
            // offsets first (they size the lists): into a buffer of their own, then lists + negated lists behind them
            void *ob = nullptr;
            RC_TRY(mimc3_ctx_workspace(ctx, kSlotPivOff0 + c, 8 * ((size_t)ns + 1), &ob));
            hp[c].d_off = static_cast<int64_t *>(ob);
            int32_t ext[3] = {0, 0, 0};
            RC_TRY(mimc3_get_uv_pivot_dev(ctx, d_xy, d_cor, ns, p->vec_ocw[c], hp[c].d_off, nullptr, nullptr, 0, &hp[c].total, ext, st));
            const size_t tb = al(8 * (size_t)(hp[c].total > 0 ? hp[c].total : 1));
            void *ub = nullptr;
            RC_TRY(mimc3_ctx_workspace(ctx, kSlotPiv0 + c, 2 * tb, &ub));
            hp[c].d_uv = static_cast<int32_t *>(ub);
            hp[c].d_uvn = reinterpret_cast<int32_t *>(static_cast<char *>(ub) + tb);
            RC_TRY(mimc3_get_uv_pivot_dev(ctx, d_xy, d_cor, ns, p->vec_ocw[c], hp[c].d_off, hp[c].d_uv, hp[c].d_uvn, hp[c].total, &hp[c].total, ext, st));
            hp[c].mn = ext[0]; hp[c].mu = ext[1]; hp[c].mv = ext[2];
            lap(c, "lists");
        }
        hp[0].d_xy = d_xy;
        HIP_TRY(hipStreamSynchronize(st));
        return 0;
    };
    const int rc = body();
    if (rc) err = mimc3_last_error();                        // the message is thread-local: carry it over
    return rc;
}

// CP offset on the whole grid (:240-256): fills res->cp_status / offset_cp and flag_cp
int vmap_cp_offset(mimc3_ctx *const *ctxs, int32_t nctx, const double *xyuvav, int32_t N, const mimc3_vmap_params *p, uint8_t *flag_cp, mimc3_vmap_result *res)
{
    mimc3_ctx *ctx = ctxs[0];
    mimc3_cp_params cp{};
    for (int k = 0; k < 4; k++) cp.vec_ocw[k] = p->vec_ocw[k];
    cp.aw_cre = p->aw_cre; cp.num_cp_max = p->num_cp_max; cp.num_cp_min = p->num_cp_min;
    cp.ratio_cp = p->ratio_cp; cp.thres_spd_cp = p->thres_spd_cp; cp.seed = p->cp_seed;
    for (int k = 0; k < 3; k++) { cp.kernel[k] = p->kernel[k]; cp.kdim[k][0] = p->kdim[k][0]; cp.kdim[k][1] = p->kdim[k][1]; }
    std::memset(flag_cp, 0, (size_t)N);
    int32_t off[2] = {0, 0}, st = -1;
    RC_TRY(mimc3_ctx_filter_images(ctx, nullptr, 0, 0));
    RC_TRY(mimc3_get_offset_image_multi(ctxs, nctx, xyuvav, N, &cp, off, flag_cp, &st, nullptr, nullptr));
    res->cp_status = st;
    if (st >= 0) { res->offset_cp[0] = off[0]; res->offset_cp[1] = off[1]; }
    return 0;
}

// The 32 matcher passes (:261-350) for the points xs[0..ns) (any subset of the grid, any order: grid points are independent
// in the matcher) with the CP offset `off`, into d_dp [32][pass_stride][3] (device, pass-major).  Synchronises the context's stream.
int vmap_run_passes(mimc3_ctx *ctx, const double *xs, int32_t ns, const int32_t off[2], HostPivots hp[4], const mimc3_vmap_params *p,
                    float *d_dp, size_t pass_stride)
{
    // points per pass slot of d_dp (a multi-GPU driver pads its blocks); 0 = ns.  A stride below ns would make the passes write
    // past the caller's [32][pass_stride][3] buffer: refused, not silently raised
    if (pass_stride == 0) pass_stride = (size_t)ns;
    if (pass_stride < (size_t)ns) return mimc3::fail(MIMC3_EINVAL, "vmap_run_passes: pass_stride is smaller than the number of points");
    int32_t H = 0, W = 0;
    RC_TRY(mimc3_ctx_image_size(ctx, &H, &W));
    StageClock clk;
    hipStream_t s = static_cast<hipStream_t>(mimc3_ctx_stream(ctx));
    // ---- the reference refuses nothing, it reads out of bounds; this library refuses (see mimc3_match_ncc_dlc)
    int ocw_max = 0;
    for (int k = 0; k < 4; k++) ocw_max = p->vec_ocw[k] > ocw_max ? p->vec_ocw[k] : ocw_max;
    for (int32_t i = 0; i < ns; i++) {
        const int32_t u0 = (int32_t)xs[6 * (size_t)i + 2], v0 = (int32_t)xs[6 * (size_t)i + 3];
        if (u0 - ocw_max < 0 || u0 + ocw_max >= W || v0 - ocw_max < 0 || v0 + ocw_max >= H)
            return mimc3::fail(MIMC3_EBOUNDS, "mimc3_vmap: a grid point's chip leaves the image (u=" + std::to_string(u0) + ", v=" + std::to_string(v0) + ")");
    }
    (void)ns;
    void *d_xy = nullptr;
    d_xy = hp[0].d_xy;                                        // uploaded with the corridors (vmap_host_pivots)
    if (!d_xy) return mimc3::fail(MIMC3_ESTATE, "mimc3_vmap: pivots were not made");

    // ---- pivots: forward and negated copies are resident already (uploaded by vmap_host_pivots' threads)
    struct Piv { int32_t *uv, *uvn; int64_t *off; };
    Piv piv[4];
    for (int c = 0; c < 4; c++) {
        if (!hp[c].d_uv || !hp[c].d_off) return mimc3::fail(MIMC3_ESTATE, "mimc3_vmap: pivots were not uploaded");
        piv[c] = {hp[c].d_uv, hp[c].d_uvn, hp[c].d_off};
    }
    HIP_TRY(hipStreamSynchronize(s));
    clk.mark("grid upload", s);
    // ---- 32 matcher passes (:261-350): variant -1 = the pair as loaded, 0..2 = the three filters
    for (int kk = -1; kk <= 2; kk++) {
        if (kk >= 0) {
            RC_TRY(mimc3_ctx_filter_images(ctx, p->kernel[kk], p->kdim[kk][0], p->kdim[kk][1]));
            clk.mark("filter + planes", s);
        }
        for (int c = 0; c < 4; c++) {
            const int slot = (kk + 1) * 8 + c * 2;
            float *fw = d_dp + (size_t)slot * pass_stride * 3, *sw = fw + pass_stride * 3;
            RC_TRY(mimc3_match_ncc_dlc_dev(ctx, static_cast<const double *>(d_xy), ns, off[0], off[1], piv[c].uv, piv[c].off,
                                           hp[c].mn, hp[c].mu, hp[c].mv, p->vec_ocw[c], 0, fw, s));
            RC_TRY(mimc3_match_ncc_dlc_dev(ctx, static_cast<const double *>(d_xy), ns, -off[0], -off[1], piv[c].uvn, piv[c].off,
                                           hp[c].mn, hp[c].mu, hp[c].mv, p->vec_ocw[c], 1, sw, s));
            RC_TRY(mimc3_negate_uv_dev(ctx, sw, ns, s));     // :289-293
        }
        clk.mark("8 matcher passes", s);
    }
    RC_TRY(mimc3_ctx_filter_images(ctx, nullptr, 0, 0));
    HIP_TRY(hipStreamSynchronize(s));                       // d_dp is complete
    clk.mark("back to the raw pair", s);
    return 0;
}

}  // namespace mimc3

// CP offset on the whole grid, then the 32 matcher passes for grid points [lo, hi) only (grid points are independent in
// the matcher: this is the unit a multi-GPU driver shards).  d_dp: device, [32][hi-lo][3] pass-major.
extern "C" int mimc3_vmap_passes(mimc3_ctx *ctx, const double *xyuvav, int32_t N, float dt, const mimc3_vmap_params *p, int32_t lo,
                                 int32_t hi, float *d_dp, uint8_t *flag_cp, mimc3_vmap_result *res)
{
    if (!ctx || !xyuvav || !p || !flag_cp || !res || N < 2 || lo < 0 || hi > N || lo > hi || (hi > lo && !d_dp))
        return mimc3::fail(MIMC3_EINVAL, "mimc3_vmap_passes: bad argument");
    int32_t H = 0, W = 0;
    RC_TRY(mimc3_ctx_image_size(ctx, &H, &W));
    std::memset(res, 0, sizeof(*res));
    RC_TRY(mimc3::vmap_geometry(xyuvav, N, res));
    StageClock clk;
    hipStream_t s = static_cast<hipStream_t>(mimc3_ctx_stream(ctx));
    const int32_t ns = hi - lo;
    const double *xs = xyuvav + 6 * (size_t)lo;
    // the host pivots run on a host thread WHILE the device measures the CP offset
    mimc3::HostPivots hp[4];
    int piv_rc = 0;
    std::string piv_err;
    const float mpp = res->mpp;
    std::thread piv_worker([&]() { piv_rc = mimc3::vmap_host_pivots(ctx, xs, ns, dt, mpp, p, H, W, hp, piv_err); });
    struct Joiner { std::thread &t; ~Joiner() { if (t.joinable()) t.join(); } } joiner{piv_worker};
    RC_TRY(mimc3::vmap_cp_offset(&ctx, 1, xyuvav, N, p, flag_cp, res));
    if (res->cp_status < 0) return 0;                       // the CLI touches vmap.tar and gives up (:248-252)
    clk.mark("control-point offset", s);
    piv_worker.join();
    if (piv_rc) return mimc3::fail(piv_rc, piv_err);
    clk.mark("pivots: join", s);
    if (ns == 0) return 0;
    return mimc3::vmap_run_passes(ctx, xs, ns, res->offset_cp, hp, p, d_dp, 0);
}

// The same in pieces, for a multi-process driver that measures the CP offset once and shards by cost-balanced point sets:
extern "C" int mimc3_vmap_geometry(const double *xyuvav, int32_t N, mimc3_vmap_result *res)
{
    if (!xyuvav || !res || N < 2) return mimc3::fail(MIMC3_EINVAL, "mimc3_vmap_geometry: bad argument");
    std::memset(res, 0, sizeof(*res));
    return mimc3::vmap_geometry(xyuvav, N, res);
}

extern "C" int mimc3_vmap_cp(mimc3_ctx *ctx, const double *xyuvav, int32_t N, float dt, const mimc3_vmap_params *p, uint8_t *flag_cp,
                             mimc3_vmap_result *res)
{
    (void)dt;
    if (!ctx || !xyuvav || !p || !flag_cp || !res || N < 2) return mimc3::fail(MIMC3_EINVAL, "mimc3_vmap_cp: bad argument");
    std::memset(res, 0, sizeof(*res));
    RC_TRY(mimc3::vmap_geometry(xyuvav, N, res));
    return mimc3::vmap_cp_offset(&ctx, 1, xyuvav, N, p, flag_cp, res);
}

extern "C" int mimc3_vmap_passes_points(mimc3_ctx *ctx, const double *xs, int32_t n, float dt, const mimc3_vmap_params *p,
                                        const mimc3_vmap_result *res, float *d_dp, int64_t pass_stride)
{
    if (!ctx || !xs || !p || !res || !d_dp || n < 1 || res->cp_status < 0) return mimc3::fail(MIMC3_EINVAL, "mimc3_vmap_passes_points: bad argument");
    if (pass_stride > 0 && pass_stride < n) return mimc3::fail(MIMC3_EINVAL, "mimc3_vmap_passes_points: 0 < pass_stride < n (d_dp is [32][pass_stride][3]; 0 means n)");
    int32_t H = 0, W = 0;
    RC_TRY(mimc3_ctx_image_size(ctx, &H, &W));
    mimc3::HostPivots hp[4];
    std::string err;
    int rc = mimc3::vmap_host_pivots(ctx, xs, n, dt, res->mpp, p, H, W, hp, err);
    if (rc) return mimc3::fail(rc, err);
    return mimc3::vmap_run_passes(ctx, xs, n, res->offset_cp, hp, p, d_dp, pass_stride > 0 ? (size_t)pass_stride : 0);
}

// Post-processing of the complete candidate tensor d_dp [32][N][3] (device) and the unit conversion (:353-402).
// `res` is the one mimc3_vmap_passes filled (its geometry is used, cp_subint is added).
extern "C" int mimc3_vmap_finish(mimc3_ctx *ctx, const double *xyuvav, int32_t N, float dt, const mimc3_vmap_params *p, const float *d_dp,
                                 float *vx, float *vy, float *ex, float *ey, float *qual, mimc3_vmap_result *res)
{
    if (!ctx || !xyuvav || !p || !d_dp || !vx || !vy || !ex || !ey || !qual || !res || N < 2 || res->dimx * res->dimy != N)
        return mimc3::fail(MIMC3_EINVAL, "mimc3_vmap_finish: bad argument (res must come from mimc3_vmap_passes)");
    StageClock clk;
    hipStream_t s = static_cast<hipStream_t>(mimc3_ctx_stream(ctx));
    const size_t n = (size_t)N;
    View d_xy, d_out5;
    RC_TRY(mimc3_ctx_workspace(ctx, kSlotXyFull, 48 * n, &d_xy.p));
    RC_TRY(mimc3_ctx_workspace(ctx, kSlotOut5, 20 * n, &d_out5.p));
    HIP_TRY(hipMemcpyAsync(d_xy.p, xyuvav, 48 * n, hipMemcpyHostToDevice, s));
    // ---- postprocess (:353)
    RC_TRY(mimc3_postprocess_dev(ctx, d_dp, 32, xyuvav, d_xy.as<double>(), res->dimx, res->dimy, dt, res->mpp, res->meter_per_spacing,
                                 p->radius_neighbor_dpf1, p->radius_neighbor_ps, p->qm_max_sweeps > 0 ? p->qm_max_sweeps : 101,
                                 d_out5.as<float>(), s));
    clk.mark("postprocess", s);
    HIP_TRY(hipMemcpyAsync(vx, d_out5.as<float>(), 4 * n, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(vy, d_out5.as<float>() + n, 4 * n, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(ex, d_out5.as<float>() + 2 * n, 4 * n, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(ey, d_out5.as<float>() + 3 * n, 4 * n, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(qual, d_out5.as<float>() + 4 * n, 4 * n, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));

    // ---- sub-integer CP offset = grid mean, removed; px -> m/yr (:356-402), f32 in the reference's order
    float sdu = 0.0f, sdv = 0.0f;
    int32_t num = 0;
    for (size_t i = 0; i < n; i++)
        if (!std::isnan(vx[i]) && !std::isnan(vy[i])) { sdu += vx[i]; sdv += vy[i]; num++; }
    const float du_cp = sdu / (float)num, dv_cp = sdv / (float)num;
    res->cp_subint[0] = du_cp; res->cp_subint[1] = dv_cp;
    const float factor = res->mpp / dt * 365;
    for (size_t i = 0; i < n; i++) {
        const float a = vx[i] - du_cp, b = vy[i] - dv_cp;
        vx[i] = a * factor;
        vy[i] = -b * factor;
        ex[i] = (float)(std::sqrt((double)ex[i]) * (double)factor);
        ey[i] = (float)(std::sqrt((double)ey[i]) * (double)factor);
    }
    clk.mark("download + unit conversion", s);
    return 0;
}

extern "C" int mimc3_vmap(mimc3_ctx *ctx, const double *xyuvav, int32_t N, float dt, const mimc3_vmap_params *p, float *vx, float *vy,
                          float *ex, float *ey, float *qual, uint8_t *flag_cp, mimc3_vmap_result *res)
{
    if (!ctx || !xyuvav || !p || !vx || !vy || !ex || !ey || !qual || !flag_cp || !res || N < 2)
        return mimc3::fail(MIMC3_EINVAL, "mimc3_vmap: bad argument");
    void *d_dp = nullptr;
    RC_TRY(mimc3_ctx_workspace(ctx, kSlotDp, 12 * (size_t)N * 32, &d_dp));
    RC_TRY(mimc3_vmap_passes(ctx, xyuvav, N, dt, p, 0, N, static_cast<float *>(d_dp), flag_cp, res));
    if (res->cp_status < 0) return 0;
    return mimc3_vmap_finish(ctx, xyuvav, N, dt, p, static_cast<const float *>(d_dp), vx, vy, ex, ey, qual, res);
}
