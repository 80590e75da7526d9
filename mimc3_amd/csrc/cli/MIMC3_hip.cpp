// MIMC3_hip -- the reference program's command line over libmimc3_hip.so (MI355X).
//
//     MIMC3_hip <i0.tif> <i1.tif> <xyuvav.GMA> <outdir>
//
// Same arguments, same files in <outdir>, same exit behaviour as the reference's main() (MIMC_main.c:43-528):
//   * image paths must contain a '/' and their basename must start with YYYYMMDDhhmmss (getTimeStampStr,
//     MIMC_misc.c:133-153); dt = difference of the two timestamps in days (get_datenum / get_dt, :27-131)
//   * <outdir>/vmap_<t0>_<t1>.tar already there -> "vmap already exists. Skipping", exit -1 (:122-130)
//   * not enough control points -> an empty vmap_<t0>_<t1>.tar is created, exit -1 (:246-252)
//   * outputs: vmap_<t0>_<t1>_{x,y}.GMA (f64 1 x dim), _{vx,vy,ex,ey,qual}.GMA (f32 dimy x dimx), _flagcp.GMA (u8),
//     _meta.txt (7 key=value lines) (:404-447)
// Everything between "inputs loaded" and "save the output" is mimc3_vmap() on the GPU; there is no CPU path.
// Environment: MIMC3_HIP_DEVICE (default 0); MIMC3_HIP_DEVICES=0,1,... shards the grid points over several GPUs of the node
// (one host thread per device, RCCL all-gather of the candidate blocks: mimc3_mgpu_vmap); MIMC3_CP_SEED (pin the
// control-point shuffle; default time(NULL) like the reference).
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <chrono>
#include <thread>
#include <unistd.h>
#include <tiffio.h>
#include "../../../include/mimc3_hip.h"

namespace {

bool leap(int y) { return y % 4 == 0 && (y % 100 != 0 || y % 400 == 0); }          // MIMC_misc.c:9-25

double datenum(const char *s)                                                       // MIMC_misc.c:27-119
{
    char buf[8];
    auto field = [&](int at, int len) { std::memset(buf, 0, sizeof buf); std::memcpy(buf, s + at, len); return buf; };
    const int y = atoi(field(0, 4));
    const int m = atoi(field(4, 2));
    const int d = atoi(field(6, 2));
    const double H = atof(field(8, 2));
    const double M = atof(field(10, 2));
    const double S = atof(field(12, 2));
    double off = 0.0;
    for (int k = 0; k < y; k++) off += leap(k) ? 366.0 : 365.0;
    static const int acc_n[12] = {0, 31, 59, 90, 120, 151, 181, 212, 243, 273, 304, 334};
    static const int acc_l[12] = {0, 31, 60, 91, 121, 152, 182, 213, 244, 274, 305, 335};
    const int *acc = leap(y) ? acc_l : acc_n;
    off += (double)acc[m - 1] + (double)d + H / 24.0 + M / 1440.0 + S / 86400.0;
    return off;
}

bool timestamp_of(const char *path, char out[15])                                   // MIMC_misc.c:133-153
{
    const char *slash = std::strrchr(path, '/');
    if (!slash || std::strlen(slash + 1) < 14) return false;      // the reference reads garbage here; this refuses
    std::memcpy(out, slash + 1, 14);
    out[14] = '\0';
    for (int i = 0; i < 14; i++)
        if (out[i] < '0' || out[i] > '9') return false;
    return true;
}

// GMA_float_load_tiff (GMA.c:246-316): scanline reader; bytes per pixel = scanline size / width, 1 -> u8, 2 -> u16.
// The reference widens to float32 on the host (:288-310); here the RAW DN is kept (scanlines are read straight into
// the buffer that crosses PCIe) and the widening runs on the device (mimc3_ctx_set_images_u8/_u16).
// Anything the reference's reader would misread (1/4-bit, multi-sample, 32-bit) is refused instead.
struct RawImage {
    std::vector<unsigned char> px;      // H * W * bpp bytes, row-major
    int32_t H = 0, W = 0, bpp = 0;
};
bool load_tiff(const char *path, RawImage &img)
{
    TIFF *tif = TIFFOpen(path, "r");
    if (!tif) return false;
    uint32_t h = 0, w = 0;
    uint16_t bits = 0, spp = 1;
    TIFFGetField(tif, TIFFTAG_IMAGELENGTH, &h);
    TIFFGetField(tif, TIFFTAG_IMAGEWIDTH, &w);
    TIFFGetFieldDefaulted(tif, TIFFTAG_BITSPERSAMPLE, &bits);
    TIFFGetFieldDefaulted(tif, TIFFTAG_SAMPLESPERPIXEL, &spp);
    const tsize_t scan = TIFFScanlineSize(tif);
    const int bpp = bits / 8;
    if (h == 0 || w == 0 || scan <= 0 || (bits != 8 && bits != 16) || spp != 1 || (size_t)scan != (size_t)w * bpp) {
        fprintf(stderr, "%s: only single-sample 8- or 16-bit images are supported (bits=%d, samples=%d)\n", path, (int)bits, (int)spp);
        TIFFClose(tif);
        return false;
    }
    img.px.resize((size_t)h * w * bpp);
    for (uint32_t r = 0; r < h; r++)
        if (TIFFReadScanline(tif, img.px.data() + (size_t)r * scan, r, 0) < 0) { TIFFClose(tif); return false; }
    TIFFClose(tif);
    img.H = (int32_t)h; img.W = (int32_t)w; img.bpp = bpp;
    printf("Loading TIFF - row=%d, col=%d, bytes per pixel=%d\n", img.H, (int)scan, bpp);
    return true;
}

// .GMA container (GMA.c:168-244, :319-424): int32 rows, int32 cols, row-major payload
bool load_gma_double(const char *path, std::vector<double> &v, int32_t &rows, int32_t &cols)
{
    FILE *f = fopen(path, "rb");
    if (!f) return false;
    bool ok = fread(&rows, 4, 1, f) == 1 && fread(&cols, 4, 1, f) == 1 && rows > 0 && cols > 0;
    if (ok) { v.resize((size_t)rows * cols); ok = fread(v.data(), 8, v.size(), f) == v.size(); }
    fclose(f);
    return ok;
}
template <class T> bool save_gma(const std::string &path, const T *p, int32_t rows, int32_t cols)
{
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) return false;
    bool ok = fwrite(&rows, 4, 1, f) == 1 && fwrite(&cols, 4, 1, f) == 1 && fwrite(p, sizeof(T), (size_t)rows * cols, f) == (size_t)rows * cols;
    return fclose(f) == 0 && ok;
}

}  // namespace

int main(int argc, char *argv[])
{
    const char ver[] = "3.0.7";        // the reference version whose outputs this build reproduces (meta key MIMC_version)
    printf("\n\nMIMC version %s -- MI355X build (%s)\n\n", ver, mimc3_version());
    if (argc != 5) {
        fprintf(stderr, "usage: %s <i0.tif> <i1.tif> <xyuvav.GMA> <outdir>\n", argv[0]);
        return 2;
    }
    char t0[15], t1[15];
    if (!timestamp_of(argv[1], t0) || !timestamp_of(argv[2], t1)) {
        fprintf(stderr, "image paths must contain a '/' and the file names must start with YYYYMMDDhhmmss\n");
        return 2;
    }
    const std::string base = std::string(argv[4]) + "/vmap_" + t0 + "_" + t1;
    const std::string f_tar = base + ".tar";
    const float dt = (float)(datenum(t1) - datenum(t0));                                // get_dt, MIMC_misc.c:121-131
    printf("dt=%f days\n\nEarlier image: %s\nLatter image: %s\nxyuvav matrix: %s\n", dt, argv[1], argv[2], argv[3]);
    if (access(f_tar.c_str(), F_OK) == 0) {                                             // :122-130
        printf("vmap already exists. Skipping\n");
        return -1;
    }

    // ---- parameters and kernels (:134-194)
    static const float k_dx[3] = {-1, 0, 1}, k_dy[3] = {-1, 0, 1};
    static const float k_lap[9] = {-1.0 / 8, -1.0 / 8, -1.0 / 8, -1.0 / 8, 1.0, -1.0 / 8, -1.0 / 8, -1.0 / 8, -1.0 / 8};
    mimc3_vmap_params p{};
    p.vec_ocw[0] = 7; p.vec_ocw[1] = 15; p.vec_ocw[2] = 30; p.vec_ocw[3] = 40;
    p.aw_cre = 10.0f; p.aw_sf = 1.8f;
    p.radius_neighbor_dpf1 = 1000 / 300; p.radius_neighbor_ps = 5.0f;
    p.num_cp_max = 500; p.num_cp_min = 50; p.ratio_cp = 0.03f; p.thres_spd_cp = 10;
    p.kernel[0] = k_dx; p.kdim[0][0] = 1; p.kdim[0][1] = 3;
    p.kernel[1] = k_dy; p.kdim[1][0] = 3; p.kdim[1][1] = 1;
    p.kernel[2] = k_lap; p.kdim[2][0] = 3; p.kdim[2][1] = 3;
    const char *seed = getenv("MIMC3_CP_SEED");
    p.cp_seed = seed ? atoll(seed) : -1;
    p.qm_max_sweeps = 101;

    // MIMC3_CLI_TIMING=1: wall time of each step on stderr
    const bool timing = getenv("MIMC3_CLI_TIMING") != nullptr;
    auto t_prev = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!timing) return;
        const auto t = std::chrono::steady_clock::now();
        fprintf(stderr, "[MIMC3_hip] %-28s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(t - t_prev).count());
        t_prev = t;
    };
    // ---- inputs (:200-230)
    std::vector<double> xy;
    int32_t N = 0, ncol = 0;
    if (!load_gma_double(argv[3], xy, N, ncol) || ncol != 6) { fprintf(stderr, "cannot read %s as an [N][6] float64 .GMA\n", argv[3]); return 2; }
    // the HIP runtime and the device context(s) come up on a second thread while this one decodes the TIFFs
    const char *dev = getenv("MIMC3_HIP_DEVICE");
    std::vector<int32_t> devs;
    if (const char *dl = getenv("MIMC3_HIP_DEVICES")) {
        for (const char *q = dl; *q;) {
            char *end = nullptr;
            const long v = strtol(q, &end, 10);
            if (end == q) break;
            devs.push_back((int32_t)v);
            q = (*end == ',') ? end + 1 : end;
        }
        if (devs.empty()) { fprintf(stderr, "MIMC3_HIP_DEVICES: expected a comma-separated list of device ordinals\n"); return 2; }
    }
    mimc3_ctx *ctx = nullptr;
    mimc3_mgpu *mg = nullptr;
    int ctx_rc = 0;
    std::string ctx_err;
    std::thread ctx_thread([&]() {
#ifdef MIMC3_TEST_HOOKS      // test build only (tests/_build/MIMC3_hip_test): N ranks on one device over a stand-in communicator
        if (!devs.empty() && (getenv("MIMC3_TEST_COMM_LIB") || getenv("MIMC3_TEST_REPEAT"))) {
            const char *rp = getenv("MIMC3_TEST_REPEAT");
            ctx_rc = mimc3_mgpu_create_ex(devs.data(), (int32_t)devs.size(), getenv("MIMC3_TEST_COMM_LIB"), (rp && rp[0] == '1') ? MIMC3_MGPU_REPEAT_DEVICES : 0u, &mg);
        } else
#endif
        ctx_rc = devs.empty() ? mimc3_ctx_create(dev ? atoi(dev) : 0, &ctx) : mimc3_mgpu_create(devs.data(), (int32_t)devs.size(), &mg);
        if (ctx_rc) ctx_err = mimc3_last_error();
    });
    auto cleanup = [&]() { if (ctx) mimc3_ctx_destroy(ctx); if (mg) mimc3_mgpu_destroy(mg); ctx = nullptr; mg = nullptr; };
    // Every exit once the device context may exist goes through here: flush, then _exit.  Returning from main() would run
    // the static destructors of the HIP runtime (and of RCCL with MIMC3_HIP_DEVICES) with contexts, streams and communicators
    // still alive -- on some RCCL versions that hangs or aborts, turning a clean error code into a signal.
    auto leave = [&](int code) -> int {
        fflush(nullptr);
        if (getenv("MIMC3_CLI_TEARDOWN")) { cleanup(); return code; }     // (leak checkers)
        _exit(code);
    };
    RawImage i0, i1;
    lap("xyuvav read");
    const bool tiff_ok = load_tiff(argv[1], i0) && load_tiff(argv[2], i1);
    lap("TIFF decode");
    ctx_thread.join();
    lap("device context (rest of)");
    if (!tiff_ok) { fprintf(stderr, "cannot read the TIFF images\n"); return leave(2); }
    if (i0.H != i1.H || i0.W != i1.W) { fprintf(stderr, "the two images differ in size\n"); return leave(2); }
    if (ctx_rc) { fprintf(stderr, "%s\n", ctx_err.c_str()); return leave(3); }
    const int32_t H = i0.H, W = i0.W;
    int rc;
    if (i0.bpp == 1 && i1.bpp == 1)
        rc = mg ? mimc3_mgpu_set_images_u8(mg, i0.px.data(), i1.px.data(), H, W) : mimc3_ctx_set_images_u8(ctx, i0.px.data(), i1.px.data(), H, W);
    else if (i0.bpp == 2 && i1.bpp == 2) {
        const uint16_t *q0 = reinterpret_cast<const uint16_t *>(i0.px.data()), *q1 = reinterpret_cast<const uint16_t *>(i1.px.data());
        rc = mg ? mimc3_mgpu_set_images_u16(mg, q0, q1, H, W) : mimc3_ctx_set_images_u16(ctx, q0, q1, H, W);
    } else {                                  // one 8-bit and one 16-bit file: widen on the host like the reference
        std::vector<float> f0((size_t)H * W), f1((size_t)H * W);
        auto widen = [&](const RawImage &im, std::vector<float> &f) {
            if (im.bpp == 1) for (size_t k = 0; k < f.size(); k++) f[k] = (float)im.px[k];
            else { const uint16_t *q = reinterpret_cast<const uint16_t *>(im.px.data()); for (size_t k = 0; k < f.size(); k++) f[k] = (float)q[k]; }
        };
        widen(i0, f0); widen(i1, f1);
        rc = mg ? mimc3_mgpu_set_images(mg, f0.data(), f1.data(), H, W) : mimc3_ctx_set_images(ctx, f0.data(), f1.data(), H, W);
    }
    if (rc) { fprintf(stderr, "%s\n", mimc3_last_error()); return leave(3); }
    lap("pair upload");

    std::vector<float> vx(N), vy(N), ex(N), ey(N), qual(N);
    std::vector<uint8_t> flag(N);
    mimc3_vmap_result r{};
    rc = mg ? mimc3_mgpu_vmap(mg, xy.data(), N, dt, &p, vx.data(), vy.data(), ex.data(), ey.data(), qual.data(), flag.data(), &r)
            : mimc3_vmap(ctx, xy.data(), N, dt, &p, vx.data(), vy.data(), ex.data(), ey.data(), qual.data(), flag.data(), &r);
    if (rc) {
        fprintf(stderr, "%s\n", mimc3_last_error());
        return leave(3);
    }
    lap("vmap (data path)");
    if (mg) printf("grid points sharded over %d GPU(s), work imbalance %.1f %%\n", (int)mimc3_mgpu_ndev(mg), 100.0 * mimc3_mgpu_last_imbalance(mg));
    // (no teardown on the way out: freeing the device buffers and the HIP runtime's exit handlers cost 50+ ms and the
    //  process is about to end -- the outputs are written, then the process leaves through _exit)
    printf("MPP=%f, grid spacing=%f, meter per spacing=%fm\nDimension of the vmap: %d by %d (mapy / mapx)\n", r.mpp, r.spacing_grid,
           r.meter_per_spacing, r.dimy, r.dimx);
    if (r.cp_status < 0) {                                                              // :246-252
        printf("Generating dummy vmap file.\n");
        FILE *f = fopen(f_tar.c_str(), "ab");
        if (f) fclose(f);
        return leave(-1);
    }
    printf("Measured offset: [%d, %d] pixels (i1-i0)\n", r.offset_cp[0], r.offset_cp[1]);

    // ---- outputs (:404-447)
    std::vector<double> gx(r.dimx), gy(r.dimy);
    for (int32_t c = 0; c < r.dimx; c++) gx[c] = xy[6 * (size_t)c];
    for (int32_t c = 0; c < r.dimy; c++) gy[c] = xy[6 * (size_t)c * r.dimx + 1];
    printf("Saving the output\n");
    bool ok = save_gma(base + "_x.GMA", gx.data(), 1, r.dimx) && save_gma(base + "_y.GMA", gy.data(), 1, r.dimy) &&
              save_gma(base + "_vx.GMA", vx.data(), r.dimy, r.dimx) && save_gma(base + "_vy.GMA", vy.data(), r.dimy, r.dimx) &&
              save_gma(base + "_ex.GMA", ex.data(), r.dimy, r.dimx) && save_gma(base + "_ey.GMA", ey.data(), r.dimy, r.dimx) &&
              save_gma(base + "_qual.GMA", qual.data(), r.dimy, r.dimx) && save_gma(base + "_flagcp.GMA", flag.data(), r.dimy, r.dimx);
    FILE *fm = fopen((base + "_meta.txt").c_str(), "w");
    if (fm) {
        fprintf(fm, "MIMC_version=%s\nname_i0=%s\nname_i1=%s\ncp_offset_int_u=%d\ncp_offset_int_v=%d\ncp_offset_subint_u=%f\ncp_offset_subint_v=%f\n",
                ver, argv[1], argv[2], r.offset_cp[0], r.offset_cp[1], r.cp_subint[0], r.cp_subint[1]);
        ok = fclose(fm) == 0 && ok;
    } else ok = false;
    if (!ok) { fprintf(stderr, "could not write the outputs under %s\n", argv[4]); return leave(4); }
    lap("outputs written");
    printf("Processing completed\n");
    return leave(0);
}
