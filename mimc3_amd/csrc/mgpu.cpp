// mgpu.cpp -- native multi-GPU driver: ONE process, one host thread per device, an RCCL communicator over
// the devices.  The loop being sharded is the reference's OpenMP loop over grid points (MIMC_module.c:816-838): points are
// independent in the matcher, so every device matches its own cost-balanced set of point blocks against the replicated
// image pair with no data-path collective, and ONE ncclAllGather re-assembles the result on every device:
//     mimc3_mgpu_match_ncc_dlc   one matcher pass        -> all-gather of [per][3] blocks   (12 B per grid point)
//     mimc3_mgpu_vmap            the program's data path -> all-gather of [32][per][3] blocks (384 B per grid point)
// The CP offset is measured ONCE (device 0) while the host threads prepare the partition and the pivots; post-processing
// runs on device 0.  RCCL is loaded with dlopen on first use, so single-GPU users never need it and a host program that
// carries its own RCCL (PyTorch) does not clash with this library's.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <numeric>
#include <string>
#include <thread>
#include <vector>
#include "../../include/mimc3_hip.h"
#include "host_util.h"
#include "cp_kernel.h"
#include "pipeline_internal.h"

namespace {

// ---- the five RCCL entry points this driver uses, resolved at run time --------------------------------------------------
typedef struct ncclComm *ncclComm_t;
struct Rccl {
    void *lib = nullptr;
    int (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int /*ncclDataType_t*/, ncclComm_t, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    std::string err;
};
constexpr int kNcclFloat = 7;          // ncclFloat32 (rccl.h: ncclInt8 0, ncclUint8 1, ncclInt32 2, ncclUint32 3, ncclInt64 4, ncclUint64 5, ncclFloat16 6, ncclFloat32 7)

// The communicator library is loaded once per process, on first use; `lib` (mimc3_mgpu_create_ex: tests hand in a stand-in with the same
// six entry points) only matters on that first call.  Nothing here reads the environment.
Rccl &rccl(const char *lib = nullptr)
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [lib]() {
        const char *names[] = {lib, "librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"};
        for (const char *n : names) {
            if (!n || !*n) continue;
            r.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (r.lib) break;
        }
        if (!r.lib) { r.err = std::string("cannot load RCCL (librccl.so.1): ") + (dlerror() ? dlerror() : ""); return; }
        auto sym = [&](const char *n) { void *p = dlsym(r.lib, n); if (!p && r.err.empty()) r.err = std::string("RCCL symbol missing: ") + n; return p; };
        r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(sym("ncclCommInitAll"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
        r.AllGather = reinterpret_cast<decltype(r.AllGather)>(sym("ncclAllGather"));
        r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
        r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
    });
    return r;
}

int hip_fail(hipError_t e, const char *what)
{
    return mimc3::fail((int)e > 0 ? (int)e : MIMC3_ENODEV, std::string(what) + ": " + hipGetErrorString(e));
}
int nccl_fail(int e, const char *what)
{
    Rccl &r = rccl();
    return mimc3::fail(MIMC3_ENODEV, std::string(what) + ": RCCL error " + std::to_string(e) + " (" + (r.GetErrorString ? r.GetErrorString(e) : "?") + ")");
}

}  // namespace

#define HIP_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return hip_fail(e_, #expr); } while (0)
#define RC_TRY(expr) do { int rc_ = (expr); if (rc_) return rc_; } while (0)
#define NCCL_TRY(expr) do { int e_ = (expr); if (e_ != 0) return nccl_fail(e_, #expr); } while (0)

// context scratch slots used here (pipeline.cpp owns 0..5 and 11..14)
enum { kSlotLocal = 6, kSlotGather = 7, kSlotPerm = 8, kSlotFull = 9, kSlotMatchIo = 10 };

struct mimc3_mgpu {
    std::vector<int32_t> dev;
    std::vector<mimc3_ctx *> ctx;
    std::vector<ncclComm_t> comm;
    // the partition of the last call (diagnostics / tests)
    double last_imbalance = 0.0;
};

// ---- cost model + partition (host) --------------------------------------------------------------------------------------
// Work of a grid point in one matcher pass ~ NCC evaluations x chip area: every pivot scans a 3x3 of which 5 cells are new
// along a corridor (+4 for the first), plus about one extra climb step per pivot (MIMC_module.c:691-753).
extern "C" int mimc3_point_cost(const int64_t *piv_off, int32_t N, int32_t ocw, double *cost)
{
    if (!piv_off || !cost || N <= 0 || ocw < 1) return mimc3::fail(MIMC3_EINVAL, "mimc3_point_cost: bad argument");
    const double area = (double)(2 * ocw + 1) * (2 * ocw + 1);
    for (int32_t g = 0; g < N; g++) cost[g] += (4.0 + 6.0 * (double)(piv_off[g + 1] - piv_off[g])) * area;
    return 0;
}

// Cost-balanced block-cyclic partition: the grid is cut into blocks of `block` consecutive points (neighbours share most
// of their search windows: a block keeps that L2 locality), blocks are dealt heaviest-first to the least loaded rank
// (LPT), each rank then walks its blocks in grid order.  order[start[r] .. start[r+1]) = the points of rank r.
extern "C" int mimc3_partition_points(const double *cost, int32_t N, int32_t world, int32_t block, int32_t *order, int32_t *start,
                                      double *imbalance)
{
    if (!cost || !order || !start || N <= 0 || world <= 0 || block <= 0) return mimc3::fail(MIMC3_EINVAL, "mimc3_partition_points: bad argument");
    const int32_t nb = (N + block - 1) / block;
    std::vector<double> bc((size_t)nb, 0.0);
    for (int32_t g = 0; g < N; g++) bc[(size_t)(g / block)] += cost[g];
    std::vector<int32_t> ids((size_t)nb);
    std::iota(ids.begin(), ids.end(), 0);
    std::stable_sort(ids.begin(), ids.end(), [&](int32_t a, int32_t b) { return bc[(size_t)a] > bc[(size_t)b]; });
    std::vector<double> load((size_t)world, 0.0);
    std::vector<std::vector<int32_t>> mine((size_t)world);
    for (int32_t id : ids) {
        int32_t best = 0;
        for (int32_t r = 1; r < world; r++) if (load[(size_t)r] < load[(size_t)best]) best = r;
        load[(size_t)best] += bc[(size_t)id];
        mine[(size_t)best].push_back(id);
    }
    int32_t pos = 0;
    for (int32_t r = 0; r < world; r++) {
        start[r] = pos;
        std::sort(mine[(size_t)r].begin(), mine[(size_t)r].end());
        for (int32_t id : mine[(size_t)r])
            for (int32_t g = id * block; g < std::min(N, (id + 1) * block); g++) order[pos++] = g;
    }
    start[world] = pos;
    if (imbalance) {
        double mx = 0.0, sum = 0.0;
        for (double l : load) { mx = std::max(mx, l); sum += l; }
        *imbalance = sum > 0.0 ? mx / (sum / world) - 1.0 : 0.0;
    }
    return 0;
}

// ---- lifetime -----------------------------------------------------------------------------------------------------------
extern "C" int mimc3_mgpu_create(const int32_t *devices, int32_t ndev, mimc3_mgpu **out)
{
    return mimc3_mgpu_create_ex(devices, ndev, nullptr, 0u, out);
}

// The form the tests use: `comm_lib` names a library that exports the six RCCL entry points used here (null = RCCL itself);
// flags & MIMC3_MGPU_REPEAT_DEVICES lets a device be listed several times, so that a one-GPU box can run N ranks as N contexts of the
// same device over a stand-in communicator (tests/fake_rccl.c -- RCCL refuses such a list).  It measures nothing and is not a product
// mode; the product entry point above never takes either, and nothing in this library reads them from the environment.
extern "C" int mimc3_mgpu_create_ex(const int32_t *devices, int32_t ndev, const char *comm_lib, uint32_t flags, mimc3_mgpu **out)
{
    if (!out || !devices || ndev <= 0 || ndev > 64) return mimc3::fail(MIMC3_EINVAL, "mimc3_mgpu_create: bad argument");
    *out = nullptr;
    if (!(flags & MIMC3_MGPU_REPEAT_DEVICES))           // one rank per GPU
        for (int32_t a = 0; a < ndev; a++)
            for (int32_t b = a + 1; b < ndev; b++)
                if (devices[a] == devices[b]) return mimc3::fail(MIMC3_EINVAL, "mimc3_mgpu_create: a device is listed twice (one rank per GPU)");
    Rccl &r = rccl(comm_lib && *comm_lib ? comm_lib : nullptr);
    if (!r.err.empty()) return mimc3::fail(MIMC3_ENODEV, "mimc3_mgpu_create: " + r.err);
    mimc3_mgpu *mg = new mimc3_mgpu();
    mg->dev.assign(devices, devices + ndev);
    mg->ctx.assign((size_t)ndev, nullptr);
    mg->comm.assign((size_t)ndev, nullptr);
    for (int32_t k = 0; k < ndev; k++) {
        int rc = mimc3_ctx_create(devices[k], &mg->ctx[(size_t)k]);
        if (rc) { mimc3_mgpu_destroy(mg); return rc; }
    }
    std::vector<int> dl(mg->dev.begin(), mg->dev.end());
    int e = r.CommInitAll(mg->comm.data(), ndev, dl.data());
    if (e != 0) { for (auto &c : mg->comm) c = nullptr; mimc3_mgpu_destroy(mg); return nccl_fail(e, "ncclCommInitAll"); }
    *out = mg;
    return 0;
}

extern "C" void mimc3_mgpu_destroy(mimc3_mgpu *mg)
{
    if (!mg) return;
    for (size_t k = 0; k < mg->comm.size(); k++)
        if (mg->comm[k]) { (void)hipSetDevice(mg->dev[k]); (void)rccl().CommDestroy(mg->comm[k]); }
    for (auto *c : mg->ctx) if (c) mimc3_ctx_destroy(c);
    delete mg;
}

extern "C" int32_t mimc3_mgpu_ndev(mimc3_mgpu *mg) { return mg ? (int32_t)mg->dev.size() : 0; }
extern "C" mimc3_ctx *mimc3_mgpu_ctx(mimc3_mgpu *mg, int32_t rank) { return (mg && rank >= 0 && rank < (int32_t)mg->ctx.size()) ? mg->ctx[(size_t)rank] : nullptr; }
extern "C" double mimc3_mgpu_last_imbalance(mimc3_mgpu *mg) { return mg ? mg->last_imbalance : 0.0; }

namespace {

// run fn(rank) on one host thread per device; the first failure (rank order) is reported on the calling thread
template <class F>
int per_device(mimc3_mgpu *mg, F fn)
{
    const size_t n = mg->ctx.size();
    std::vector<int> rc(n, 0);
    std::vector<std::string> err(n);
    std::vector<std::thread> th;
    for (size_t k = 1; k < n; k++)
        th.emplace_back([&, k]() { rc[k] = fn((int32_t)k); if (rc[k]) err[k] = mimc3_last_error(); });
    rc[0] = fn(0);
    if (rc[0]) err[0] = mimc3_last_error();
    for (auto &t : th) t.join();
    for (size_t k = 0; k < n; k++)
        if (rc[k]) {
            // the other ranks may hold enqueued work on their contexts' cached buffers: drain every stream before the caller
            // sees the failure (and possibly destroys or reuses the contexts)
            for (size_t j = 0; j < n; j++) {
                (void)hipSetDevice(mg->dev[j]);
                (void)hipStreamSynchronize(static_cast<hipStream_t>(mimc3_ctx_stream(mg->ctx[j])));
            }
            return mimc3::fail(rc[k], "rank " + std::to_string(k) + " (device " + std::to_string(mg->dev[k]) + "): " + err[k]);
        }
    return 0;
}

// ONE collective: every rank contributes `count` floats from d_send[r] and receives world*count floats in d_recv[r]
int all_gather(mimc3_mgpu *mg, const std::vector<const float *> &d_send, const std::vector<float *> &d_recv, size_t count)
{
    Rccl &r = rccl();
    NCCL_TRY(r.GroupStart());
    for (size_t k = 0; k < mg->ctx.size(); k++) {
        HIP_TRY(hipSetDevice(mg->dev[k]));
        int e = r.AllGather(d_send[k], d_recv[k], count, kNcclFloat, mg->comm[k], static_cast<hipStream_t>(mimc3_ctx_stream(mg->ctx[k])));
        if (e != 0) {
            // close the group (the calls already posted are abandoned with it), then drain the streams: nothing of the
            // half-issued collective may still be running when the caller reuses the buffers
            (void)r.GroupEnd();
            for (size_t j = 0; j < mg->ctx.size(); j++) {
                (void)hipSetDevice(mg->dev[j]);
                (void)hipStreamSynchronize(static_cast<hipStream_t>(mimc3_ctx_stream(mg->ctx[j])));
            }
            return nccl_fail(e, "ncclAllGather");
        }
    }
    NCCL_TRY(r.GroupEnd());
    return 0;
}

}  // namespace

// ---- images: replicated on every device (parallel uploads) ----------------------------------------------------------------
extern "C" int mimc3_mgpu_set_images(mimc3_mgpu *mg, const float *i0, const float *i1, int32_t H, int32_t W)
{
    if (!mg) return mimc3::fail(MIMC3_EINVAL, "mimc3_mgpu_set_images: mg is NULL");
    return per_device(mg, [&](int32_t k) { return mimc3_ctx_set_images(mg->ctx[(size_t)k], i0, i1, H, W); });
}
extern "C" int mimc3_mgpu_set_images_u8(mimc3_mgpu *mg, const uint8_t *i0, const uint8_t *i1, int32_t H, int32_t W)
{
    if (!mg) return mimc3::fail(MIMC3_EINVAL, "mimc3_mgpu_set_images_u8: mg is NULL");
    return per_device(mg, [&](int32_t k) { return mimc3_ctx_set_images_u8(mg->ctx[(size_t)k], i0, i1, H, W); });
}
extern "C" int mimc3_mgpu_set_images_u16(mimc3_mgpu *mg, const uint16_t *i0, const uint16_t *i1, int32_t H, int32_t W)
{
    if (!mg) return mimc3::fail(MIMC3_EINVAL, "mimc3_mgpu_set_images_u16: mg is NULL");
    return per_device(mg, [&](int32_t k) { return mimc3_ctx_set_images_u16(mg->ctx[(size_t)k], i0, i1, H, W); });
}

// ---- one matcher pass, sharded ----------------------------------------------------------------------------------------------
extern "C" int mimc3_mgpu_match_ncc_dlc(mimc3_mgpu *mg, const double *xyuvav, int32_t N, const int32_t offset[2], const int32_t *piv_uv,
                                        const int64_t *piv_off, int32_t ocw, int32_t swap, float *out)
{
    if (!mg || !xyuvav || !offset || !piv_uv || !piv_off || !out || N <= 0) return mimc3::fail(MIMC3_EINVAL, "mimc3_mgpu_match_ncc_dlc: bad argument");
    const int32_t world = (int32_t)mg->ctx.size();
    std::vector<double> cost((size_t)N, 0.0);
    RC_TRY(mimc3_point_cost(piv_off, N, ocw, cost.data()));
    std::vector<int32_t> order((size_t)N), start((size_t)world + 1);
    const int32_t block = std::max(256, std::min(4096, N / (world * 16) + 1));
    RC_TRY(mimc3_partition_points(cost.data(), N, world, block, order.data(), start.data(), &mg->last_imbalance));
    int32_t per = 1;
    for (int32_t r = 0; r < world; r++) per = std::max(per, start[(size_t)r + 1] - start[(size_t)r]);
    std::vector<const float *> d_send((size_t)world);
    std::vector<float *> d_recv((size_t)world);
    // every rank: gather its points' rows and pivots on the host, upload, match (enqueue only)
    RC_TRY(per_device(mg, [&](int32_t r) -> int {
        mimc3_ctx *c = mg->ctx[(size_t)r];
        HIP_TRY(hipSetDevice(mg->dev[(size_t)r]));
        hipStream_t s = static_cast<hipStream_t>(mimc3_ctx_stream(c));
        const int32_t lo = start[(size_t)r], n = start[(size_t)r + 1] - lo;
        void *d_local = nullptr, *d_gather = nullptr;
        RC_TRY(mimc3_ctx_workspace(c, kSlotLocal, 12 * (size_t)per, &d_local));
        RC_TRY(mimc3_ctx_workspace(c, kSlotGather, 12 * (size_t)per * world, &d_gather));
        d_send[(size_t)r] = static_cast<const float *>(d_local); d_recv[(size_t)r] = static_cast<float *>(d_gather);
        if (n == 0) return 0;
        std::vector<double> xs(6 * (size_t)n);
        std::vector<int64_t> off((size_t)n + 1);
        off[0] = 0;
        for (int32_t j = 0; j < n; j++) {
            const int32_t g = order[(size_t)(lo + j)];
            std::memcpy(&xs[6 * (size_t)j], xyuvav + 6 * (size_t)g, 48);
            off[(size_t)j + 1] = off[(size_t)j] + (piv_off[g + 1] - piv_off[g]);
        }
        std::vector<int32_t> uv(2 * (size_t)off[(size_t)n]);
        for (int32_t j = 0; j < n; j++) {
            const int32_t g = order[(size_t)(lo + j)];
            std::memcpy(&uv[2 * (size_t)off[(size_t)j]], piv_uv + 2 * piv_off[g], 8 * (size_t)(piv_off[g + 1] - piv_off[g]));
        }
        int32_t mn = 0, mu = 0, mv = 0;
        RC_TRY(mimc3_pivot_extent(uv.data(), off.data(), n, &mn, &mu, &mv));
        int32_t H = 0, W = 0;
        RC_TRY(mimc3_ctx_image_size(c, &H, &W));
        for (int32_t j = 0; j < n; j++) {
            const int32_t u0 = (int32_t)xs[6 * (size_t)j + 2], v0 = (int32_t)xs[6 * (size_t)j + 3];
            if (u0 - ocw < 0 || u0 + ocw >= W || v0 - ocw < 0 || v0 + ocw >= H)
                return mimc3::fail(MIMC3_EBOUNDS, "mimc3_mgpu_match_ncc_dlc: a grid point's chip leaves the image");
        }
        const size_t b_xy = 48 * (size_t)n, b_uv = 8 * (size_t)off[(size_t)n], b_off = 8 * ((size_t)n + 1);
        auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
        void *io = nullptr;
        RC_TRY(mimc3_ctx_workspace(c, kSlotMatchIo, al(b_xy) + al(b_uv) + al(b_off), &io));
        char *b = static_cast<char *>(io);
        HIP_TRY(hipMemcpyAsync(b, xs.data(), b_xy, hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync(b + al(b_xy), uv.data(), b_uv, hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync(b + al(b_xy) + al(b_uv), off.data(), b_off, hipMemcpyHostToDevice, s));
        HIP_TRY(hipStreamSynchronize(s));                    // the host vectors above go out of scope
        return mimc3_match_ncc_dlc_dev(c, reinterpret_cast<const double *>(b), n, offset[0], offset[1], reinterpret_cast<const int32_t *>(b + al(b_xy)),
                                       reinterpret_cast<const int64_t *>(b + al(b_xy) + al(b_uv)), mn, mu, mv, ocw, swap, static_cast<float *>(d_local), s);
    }));
    RC_TRY(all_gather(mg, d_send, d_recv, 3 * (size_t)per));
    // device 0 un-permutes the gathered blocks into grid order and hands the field back
    mimc3_ctx *c0 = mg->ctx[0];
    HIP_TRY(hipSetDevice(mg->dev[0]));
    hipStream_t s0 = static_cast<hipStream_t>(mimc3_ctx_stream(c0));
    std::vector<int32_t> perm((size_t)world * per, -1);
    for (int32_t r = 0; r < world; r++)
        for (int32_t j = 0; j < start[(size_t)r + 1] - start[(size_t)r]; j++) perm[(size_t)r * per + j] = order[(size_t)(start[(size_t)r] + j)];
    void *d_perm = nullptr, *d_full = nullptr;
    RC_TRY(mimc3_ctx_workspace(c0, kSlotPerm, 4 * perm.size(), &d_perm));
    RC_TRY(mimc3_ctx_workspace(c0, kSlotFull, 12 * (size_t)N, &d_full));
    HIP_TRY(hipMemcpyAsync(d_perm, perm.data(), 4 * perm.size(), hipMemcpyHostToDevice, s0));
    HIP_TRY(mimc3::launch_scatter_blocks(d_recv[0], static_cast<const int32_t *>(d_perm), world, per, 1, N, static_cast<float *>(d_full), s0));
    HIP_TRY(hipMemcpyAsync(out, d_full, 12 * (size_t)N, hipMemcpyDeviceToHost, s0));
    HIP_TRY(hipStreamSynchronize(s0));
    for (int32_t r = 1; r < world; r++) {                    // the other ranks' collectives have completed too before the buffers are reused
        HIP_TRY(hipSetDevice(mg->dev[(size_t)r]));
        HIP_TRY(hipStreamSynchronize(static_cast<hipStream_t>(mimc3_ctx_stream(mg->ctx[(size_t)r]))));
    }
    return 0;
}

// ---- the program's data path, sharded (MIMC_main.c:203-402) -------------------------------------------------------------------
extern "C" int mimc3_mgpu_vmap(mimc3_mgpu *mg, const double *xyuvav, int32_t N, float dt, const mimc3_vmap_params *p, float *vx, float *vy,
                               float *ex, float *ey, float *qual, uint8_t *flag_cp, mimc3_vmap_result *res)
{
    if (!mg || !xyuvav || !p || !vx || !vy || !ex || !ey || !qual || !flag_cp || !res || N < 2)
        return mimc3::fail(MIMC3_EINVAL, "mimc3_mgpu_vmap: bad argument");
    const int32_t world = (int32_t)mg->ctx.size();
    int32_t H = 0, W = 0;
    RC_TRY(mimc3_ctx_image_size(mg->ctx[0], &H, &W));
    std::memset(res, 0, sizeof(*res));
    RC_TRY(mimc3::vmap_geometry(xyuvav, N, res));
    const float mpp = res->mpp;

    // while the devices measure the CP offset (once; every segment of candidates cut into one slice per rank), a host thread counts the pivots of the four chip sizes
    // for the whole grid (the cheap half of get_uv_pivot) and cuts the grid into cost-balanced shares
    std::vector<int32_t> order((size_t)N), start((size_t)world + 1);
    int part_rc = 0;
    std::string part_err;
    std::thread part_worker([&]() {
        std::vector<double> cost((size_t)N, 0.0);
        std::vector<int64_t> off((size_t)N + 1);
        for (int c = 0; c < 4 && !part_rc; c++) {
            int64_t total = 0;
            part_rc = mimc3_get_uv_pivot(xyuvav, N, dt, mpp, p->aw_sf, p->aw_cre, p->vec_ocw[c], H, W, off.data(), nullptr, 0, &total);
            if (!part_rc) part_rc = mimc3_point_cost(off.data(), N, p->vec_ocw[c], cost.data());
        }
        const int32_t block = std::max(256, std::min(4096, N / (world * 16) + 1));
        if (!part_rc) part_rc = mimc3_partition_points(cost.data(), N, world, block, order.data(), start.data(), &mg->last_imbalance);
        if (part_rc) part_err = mimc3_last_error();
    });
    struct Joiner { std::thread &t; ~Joiner() { if (t.joinable()) t.join(); } } joiner{part_worker};
    HIP_TRY(hipSetDevice(mg->dev[0]));
    RC_TRY(mimc3::vmap_cp_offset(mg->ctx.data(), world, xyuvav, N, p, flag_cp, res));
    part_worker.join();
    if (res->cp_status < 0) return 0;                        // the CLI touches vmap.tar and gives up (:248-252)
    if (part_rc) return mimc3::fail(part_rc, part_err);
    int32_t per = 1;
    for (int32_t r = 0; r < world; r++) per = std::max(per, start[(size_t)r + 1] - start[(size_t)r]);

    std::vector<const float *> d_send((size_t)world);
    std::vector<float *> d_recv((size_t)world);
    RC_TRY(per_device(mg, [&](int32_t r) -> int {
        mimc3_ctx *c = mg->ctx[(size_t)r];
        HIP_TRY(hipSetDevice(mg->dev[(size_t)r]));
        const int32_t lo = start[(size_t)r], n = start[(size_t)r + 1] - lo;
        void *d_local = nullptr, *d_gather = nullptr;
        RC_TRY(mimc3_ctx_workspace(c, kSlotLocal, 12 * 32 * (size_t)per, &d_local));
        RC_TRY(mimc3_ctx_workspace(c, kSlotGather, 12 * 32 * (size_t)per * world, &d_gather));
        d_send[(size_t)r] = static_cast<const float *>(d_local); d_recv[(size_t)r] = static_cast<float *>(d_gather);
        if (n == 0) return 0;
        std::vector<double> xs(6 * (size_t)n);
        for (int32_t j = 0; j < n; j++) std::memcpy(&xs[6 * (size_t)j], xyuvav + 6 * (size_t)order[(size_t)(lo + j)], 48);
        mimc3::HostPivots hp[4];
        std::string err;
        int rc = mimc3::vmap_host_pivots(c, xs.data(), n, dt, mpp, p, H, W, hp, err);
        if (rc) return mimc3::fail(rc, err);
        return mimc3::vmap_run_passes(c, xs.data(), n, res->offset_cp, hp, p, static_cast<float *>(d_local), (size_t)per);
    }));
    // the ONE exchange: candidate blocks [32][per][3] of every rank -> [world][32][per][3] on every rank
    RC_TRY(all_gather(mg, d_send, d_recv, 3 * 32 * (size_t)per));
    mimc3_ctx *c0 = mg->ctx[0];
    HIP_TRY(hipSetDevice(mg->dev[0]));
    hipStream_t s0 = static_cast<hipStream_t>(mimc3_ctx_stream(c0));
    std::vector<int32_t> perm((size_t)world * per, -1);
    for (int32_t r = 0; r < world; r++)
        for (int32_t j = 0; j < start[(size_t)r + 1] - start[(size_t)r]; j++) perm[(size_t)r * per + j] = order[(size_t)(start[(size_t)r] + j)];
    void *d_perm = nullptr, *d_full = nullptr;
    RC_TRY(mimc3_ctx_workspace(c0, kSlotPerm, 4 * perm.size(), &d_perm));
    RC_TRY(mimc3_ctx_workspace(c0, kSlotFull, 12 * 32 * (size_t)N, &d_full));
    HIP_TRY(hipMemcpyAsync(d_perm, perm.data(), 4 * perm.size(), hipMemcpyHostToDevice, s0));
    HIP_TRY(mimc3::launch_scatter_blocks(d_recv[0], static_cast<const int32_t *>(d_perm), world, per, 32, N, static_cast<float *>(d_full), s0));
    HIP_TRY(hipStreamSynchronize(s0));
    for (int32_t r = 1; r < world; r++) {
        HIP_TRY(hipSetDevice(mg->dev[(size_t)r]));
        HIP_TRY(hipStreamSynchronize(static_cast<hipStream_t>(mimc3_ctx_stream(mg->ctx[(size_t)r]))));
    }
    HIP_TRY(hipSetDevice(mg->dev[0]));
    return mimc3_vmap_finish(c0, xyuvav, N, dt, p, static_cast<const float *>(d_full), vx, vy, ex, ey, qual, res);
}
