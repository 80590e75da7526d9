// capi.cpp -- the C ABI of libmimc3_hip.so (include/mimc3_hip.h): context, resident images,
// host-buffer (drop-in) and device-buffer (resident) entry points of the matcher and QM paths.
// No CPU fallback: every compute entry point needs a HIP device and fails loudly without one.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <functional>
#include <string>
#include <thread>
#include <vector>
#include "../../include/mimc3_hip.h"
#include "host_util.h"
#include "match_kernel.h"
#include "sat_kernel.h"
#include "pivot_kernel.h"
#include "qm_kernel.h"
#include "n1_kernel.h"
#include "conv2_kernel.h"
#include "cp_kernel.h"
#include <ctime>
#include <cstdlib>

namespace mimc3 {
static thread_local std::string g_err;
int fail(int code, const char *msg) { g_err = msg ? msg : ""; return code; }
int fail(int code, const std::string &msg) { g_err = msg; return code; }
static int hip_fail(hipError_t e, const char *what)
{
    g_err = std::string(what) + ": " + hipGetErrorString(e);
    return (int)e > 0 ? (int)e : MIMC3_ENODEV;
}
}  // namespace mimc3

#define HIP_TRY(expr)                                                        \
    do {                                                                     \
        hipError_t e_ = (expr);                                              \
        if (e_ != hipSuccess) return mimc3::hip_fail(e_, #expr);             \
    } while (0)

// growable device buffer
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t bytes)
    {
        if (bytes <= cap) return hipSuccess;
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        size_t want = bytes + bytes / 8 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
    // owned: a context's buffers go with the context (mimc3_ctx_destroy selects the device first), so a member added later cannot be
    // forgotten there
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { release(); }
};

struct mimc3_ctx {
    int device = 0;
    hipStream_t stream = nullptr;       // owned; host-buffer entry points run here
    const float *d_i0 = nullptr, *d_i1 = nullptr;
    DevBuf own_i0, own_i1;              // used when images were uploaded from the host
    int32_t H = 0, W = 0;
    DevBuf pl0, pl1, flag;              // zero-bordered u8 planes (exact-integer path) + "not 8-bit" flag
    DevBuf sat0, sat1, sat_tmp;         // packed summed-area tables of pl0 / pl1 (sum b | sum b^2 | nulls; sat_kernel.hip), built with the planes
    DevBuf hsat0, hsat1, hsz0, hsz1;    // the same for the u16 planes hpl0 / hpl1: sum q | sum q^2, and the null counts
    bool sat_u8_ok = false, sat_u16_ok = false;   // tables hold the CURRENT planes (chip-atlas contexts build them only if a call needs them)
    DevBuf ovf;                         // [0] count, [1..] indices of points the u8 kernel handed back
    DevBuf fail;                        // [0] count, [1..] points the offset-u8 kernel handed to the u16 kernel
    bool u8o_ok = false;                // integer (shift 0) u16 planes whose local range mostly fits 8 bits: try PxU8o first
    DevBuf hpl0, hpl1;                  // zero-bordered u16 planes of scaled integers (q = value * 2^shift < 4096)
    DevBuf rt0, rt1;                    // PxU8o: min | max << 16 of every 16x16-pixel tile of hpl0 / hpl1 (valid while u8o_ok)
    bool u16_ok = false;                // the pair is scaled-integer (and not 8-bit): u16 planes are built
    bool hpl_valid = false;             // u16 planes hold the CURRENT pair
    int shift0 = 0, shift1 = 0;
    DevBuf fpl0, fpl1;                  // zero-bordered f32 planes (register-tiled f32 kernel), built on first use
    bool fplanes_ok = false;
    DevBuf fsat0, fsat1;                // their 16-byte summed-area tables when every pixel (x 1 or x 8) is an integer in [0, 2^20) (16-bit DN and its filtered forms)
    bool f32i_ok = false;
    int fshift0 = 0, fshift1 = 0;      // pixel x 2^shift is the integer the table sums
    int32_t Wp = 0;
    bool u8_ok = false;                 // both images proven to be integers in [0,255]
    int path_mode = 0;                  // 0 auto, 1 force the general f32 kernel, 2 no integer kernels, 3 no u8 kernel, 4 auto without the matrix-core kernel
    int last_path = -1;                 // 0 general f32/f64 kernel, 1 exact u8 kernel, ... (mimc3_hip.h), 5 matrix-core u8 kernel
    DevBuf xy, puv, poff, out;          // matcher staging for the host-buffer entry point
    DevBuf pcor, pcnt, pext;            // device pivots: corridors [N] x 24 B, counts [N], extents + total (24 B)
    int32_t xy_stride = 6, xy_col = 2;  // where the matcher finds a point's (u, v) in its `xyuvav` argument: xyuvav rows, or (internal) a packed [N][2] array
    hipEvent_t ev_chunk[2][8] = {};     // mimc3_match_ncc_dlc_cor: "chunk uploaded + counted" / "chunk matched"
    DevBuf qm_io, qm_work;              // QM staging / workspace
    DevBuf n1_io, n1_work;              // clustering / dpf0 / dpf1 staging and workspace
    const float *raw_i0 = nullptr, *raw_i1 = nullptr;   // the pair as handed over (before any pre-filter)
    DevBuf filt0, filt1, conv_io;       // pre-filtered pair (mimc3_ctx_filter_images), conv2 staging
    DevBuf cp_buf;                      // control-point stage: one arena carved per call
    DevBuf cp_pre;                      // control-point stage: the filtered planes of a whole segment (their minima are settled before the slices start)
    bool filt_live = false;             // filt0/filt1 hold the output planes of an earlier filter pass on this pair
    hipStream_t side[3] = {nullptr, nullptr, nullptr};   // CP stage: its 16 small matcher launches per segment overlap on 4 streams
    hipEvent_t ev_side[4] = {nullptr, nullptr, nullptr, nullptr};
    hipStream_t aux[4] = {nullptr, nullptr, nullptr, nullptr};   // mimc3_ctx_aux_stream: copy streams of the drivers' host threads
    bool child = false;                 // a control-point child context: no streams / children of its own beyond `stream`
    bool timing = false;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool timed = false;
    // host -> device staging: two pinned chunks that a pageable source is pipelined through (a pinned source is DMA'd directly)
    void *pin[2] = {nullptr, nullptr};
    hipEvent_t ev_pin[2] = {nullptr, nullptr};
    void *hslot[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // mimc3_ctx_host_workspace: pinned host scratch
    size_t hslot_cap[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    DevBuf slot[24];                    // mimc3_ctx_workspace: named scratch the drivers built on the ABI keep across calls
    bool no_u8o = false;                // internal (CP stage): never try the per-point-offset u8 form on this context's pairs
    int32_t lane = 0;                   // internal (CP stage): which scratch set (overflow lists) the next matcher call uses: calls on
    DevBuf ovf_alt[3], fail_alt[3];     // different streams of one context must not share them
    DevBuf mxl[4];                      // matrix-core kernel: one flag byte per grid point (one buffer per `lane`)
    int32_t win_half = 0;               // internal (CP stage): > 0 = the next matcher calls use a full (2*win_half+1)^2 search area
    mimc3_ctx *cp_child[4] = {nullptr, nullptr, nullptr, nullptr};   // CP stage: one context per image variant for its chip atlas (planes, kernel selection)
    DevBuf cellws;                      // general matcher: global cell-grid workspace for corridors whose cell grid outgrows LDS
    DevBuf raw_dn;                      // raw 8/16-bit DN as uploaded (mimc3_ctx_set_images_u8/_u16), widened on the device
};

static constexpr size_t kPinChunk = 4u << 20;

// Copy `bytes` from host memory to the device on the context's stream.  Pinned / registered sources go in one DMA;
// pageable ones are pipelined through two pinned 4 MiB chunks (host memcpy of chunk k+1 overlaps the DMA of chunk k),
// which is ~10x the rate hipMemcpy reaches from pageable memory on this platform.  Returns after enqueueing (pinned) or
// after the last chunk was handed to the DMA engine (pageable); the caller synchronises the stream.
static int h2d_copy(mimc3_ctx *c, void *dst, const void *src, size_t bytes)
{
    if (bytes == 0) return 0;
    hipPointerAttribute_t at{};
    if (hipPointerGetAttributes(&at, src) == hipSuccess && at.type == hipMemoryTypeHost) {
        HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
        return 0;
    }
    (void)hipGetLastError();                                   // "not a HIP pointer" is the expected answer for pageable memory
    if (bytes < (256u << 10)) {                               // small: the runtime's own staging is fine
        HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
        return 0;
    }
    for (int k = 0; k < 2; k++) {
        if (!c->pin[k]) HIP_TRY(hipHostMalloc(&c->pin[k], kPinChunk, hipHostMallocDefault));
        if (!c->ev_pin[k]) HIP_TRY(hipEventCreateWithFlags(&c->ev_pin[k], hipEventDisableTiming));
    }
    size_t off = 0;
    for (int k = 0; off < bytes; k ^= 1) {
        const size_t n = bytes - off < kPinChunk ? bytes - off : kPinChunk;
        HIP_TRY(hipEventSynchronize(c->ev_pin[k]));             // the DMA that last read this chunk has finished
        std::memcpy(c->pin[k], static_cast<const char *>(src) + off, n);
        HIP_TRY(hipMemcpyAsync(static_cast<char *>(dst) + off, c->pin[k], n, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipEventRecord(c->ev_pin[k], c->stream));
        off += n;
    }
    return 0;
}
// Device -> host, synchronous (returns with the bytes in `dst`): through the pinned chunks when `dst` is pageable.
static int d2h_copy(mimc3_ctx *c, void *dst, const void *src, size_t bytes)
{
    if (bytes == 0) return 0;
    hipPointerAttribute_t at{};
    const bool pinned = hipPointerGetAttributes(&at, dst) == hipSuccess && at.type == hipMemoryTypeHost;
    if (!pinned) (void)hipGetLastError();
    if (pinned || bytes < (256u << 10)) {
        HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        return 0;
    }
    for (int k = 0; k < 2; k++) {
        if (!c->pin[k]) HIP_TRY(hipHostMalloc(&c->pin[k], kPinChunk, hipHostMallocDefault));
        if (!c->ev_pin[k]) HIP_TRY(hipEventCreateWithFlags(&c->ev_pin[k], hipEventDisableTiming));
    }
    HIP_TRY(hipStreamSynchronize(c->stream));                    // nothing else may be using the chunks
    size_t off = 0, prev_off = 0, prev_n = 0;
    int k = 0;
    while (off < bytes || prev_n) {
        size_t n = 0;
        if (off < bytes) {
            n = bytes - off < kPinChunk ? bytes - off : kPinChunk;
            HIP_TRY(hipMemcpyAsync(c->pin[k], static_cast<const char *>(src) + off, n, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(hipEventRecord(c->ev_pin[k], c->stream));
        }
        if (prev_n) {                                            // drain the previous chunk while this one is in flight
            HIP_TRY(hipEventSynchronize(c->ev_pin[k ^ 1]));
            std::memcpy(static_cast<char *>(dst) + prev_off, c->pin[k ^ 1], prev_n);
        }
        prev_off = off; prev_n = n; off += n; k ^= 1;
    }
    return 0;
}
#define RC_TRY(expr) do { int rc_ = (expr); if (rc_) return rc_; } while (0)

extern "C" void *mimc3_host_alloc(size_t bytes)
{
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocPortable) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return p;
}
extern "C" void mimc3_host_free(void *p) { if (p) (void)hipHostFree(p); }

static float min_dn_threshold()
{
    // smallest f32 t with (double)t >= 1e-10: "x >= MIN_DN" (f32 promoted to f64, MIMC_module.c:21,:723)
    // is then exactly "x >= t" in f32.
    float t = (float)1e-10;
    if ((double)t < 1e-10) t = std::nextafterf(t, 1.0f);
    return t;
}

extern "C" const char *mimc3_last_error(void) { return mimc3::g_err.c_str(); }
extern "C" const char *mimc3_version(void) { return "mimc3_hip 0.1.0 (gfx950)"; }

static int ctx_create_impl(int device, mimc3_ctx **out, bool child);

extern "C" int mimc3_ctx_create(int device, mimc3_ctx **out) { return ctx_create_impl(device, out, false); }

static int ctx_create_impl(int device, mimc3_ctx **out, bool child)
{
    if (!out) return mimc3::fail(MIMC3_EINVAL, "mimc3_ctx_create: out is NULL");
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) return mimc3::fail(MIMC3_ENODEV, "mimc3_ctx_create: no HIP device (this library has no CPU fallback)");
    if (device < 0 || device >= ndev) return mimc3::fail(MIMC3_EINVAL, "mimc3_ctx_create: device index out of range");
    HIP_TRY(hipSetDevice(device));
    mimc3_ctx *c = new mimc3_ctx();
    c->device = device;
    e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete c; return mimc3::hip_fail(e, "hipStreamCreate"); }
    (void)hipEventCreate(&c->ev0);
    (void)hipEventCreate(&c->ev1);
    c->child = child;
    if (!child) {
        // everything the control-point stage and the drivers' host threads need later is created NOW, on an idle device:
        // hipStreamCreate / hipStreamDestroy take milliseconds each while kernels are running (measured 3-8 ms)
        for (auto &st : c->side) (void)hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
        for (auto &st : c->aux) (void)hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
        for (auto &ev : c->ev_side) (void)hipEventCreateWithFlags(&ev, hipEventDisableTiming);
        for (auto &ch : c->cp_child) (void)ctx_create_impl(device, &ch, true);
    }
    *out = c;
    return 0;
}

extern "C" void *mimc3_ctx_aux_stream(mimc3_ctx *c, int32_t k)
{
    return (c && k >= 0 && k < 4) ? static_cast<void *>(c->aux[k]) : nullptr;
}

extern "C" void mimc3_ctx_destroy(mimc3_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    c->own_i0.release(); c->own_i1.release();
    c->rt0.release(); c->rt1.release();
    c->pl0.release(); c->pl1.release(); c->flag.release(); c->ovf.release(); c->fail.release(); c->fpl0.release(); c->fpl1.release(); c->hpl0.release(); c->hpl1.release();
    c->xy.release(); c->puv.release(); c->poff.release(); c->out.release();
    c->qm_io.release(); c->qm_work.release();
    c->n1_io.release(); c->n1_work.release();
    c->filt0.release(); c->filt1.release(); c->conv_io.release(); c->cp_buf.release();
    for (auto &ch : c->cp_child) if (ch) { mimc3_ctx_destroy(ch); ch = nullptr; }
    for (auto &b : c->ovf_alt) b.release();
    for (auto &b : c->fail_alt) b.release();
    c->raw_dn.release(); c->cellws.release();
    for (auto &b : c->slot) b.release();
    for (auto &h : c->hslot) if (h) (void)hipHostFree(h);
    for (auto &pp : c->pin) if (pp) (void)hipHostFree(pp);
    for (auto &ev : c->ev_pin) if (ev) (void)hipEventDestroy(ev);
    for (auto &st : c->side) if (st) (void)hipStreamDestroy(st);
    for (auto &st : c->aux) if (st) (void)hipStreamDestroy(st);
    for (auto &ev : c->ev_side) if (ev) (void)hipEventDestroy(ev);
    for (auto &row : c->ev_chunk) for (auto &ev : row) if (ev) (void)hipEventDestroy(ev);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

// Build the zero-bordered u8 planes and prove (on the device) that both images are 8-bit integral.
// Runs once per image pair; the CLI then reuses the pair for 8 matcher passes (MIMC_main.c:261-300).
// what the tables of a context's planes cover: the whole zero-bordered plane (windows hang over the image edge), or -- chip
// atlases of the control-point stage, whose full-square search areas stay inside a tile -- the image area alone
static mimc3::SatRegion table_region(const mimc3_ctx *c)
{
    const int pad = mimc3::kU8Pad;
    if (c->child) return mimc3::SatRegion{pad, pad, c->W, c->H};
    return mimc3::SatRegion{0, 0, c->Wp, c->H + 2 * pad};
}

// The summed-area tables of the u8 planes: built once per image pair, right behind the planes (enqueued on the context's stream).
static int build_u8_tables(mimc3_ctx *c)
{
    const int Hp = c->H + 2 * mimc3::kU8Pad;
    HIP_TRY(c->sat0.reserve(mimc3::sat_bytes(Hp, c->Wp)));
    HIP_TRY(c->sat1.reserve(mimc3::sat_bytes(Hp, c->Wp)));
    HIP_TRY(c->sat_tmp.reserve(mimc3::sat_scratch_bytes(Hp, c->Wp)));
    HIP_TRY(mimc3::launch_sat_u8(static_cast<const unsigned char *>(c->pl0.p), c->Wp, table_region(c), static_cast<unsigned long long *>(c->sat0.p), c->sat_tmp.p, c->stream));
    HIP_TRY(mimc3::launch_sat_u8(static_cast<const unsigned char *>(c->pl1.p), c->Wp, table_region(c), static_cast<unsigned long long *>(c->sat1.p), c->sat_tmp.p, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));      // matcher calls may come in on any stream
    c->sat_u8_ok = true;
    return 0;
}

// ... and of the u16 planes (enqueued on `s`; callers order later use on other streams themselves)
static int build_u16_tables(mimc3_ctx *c, hipStream_t s)
{
    const int Hp = c->H + 2 * mimc3::kU8Pad;
    HIP_TRY(c->hsat0.reserve(mimc3::sat_bytes(Hp, c->Wp)));
    HIP_TRY(c->hsat1.reserve(mimc3::sat_bytes(Hp, c->Wp)));
    HIP_TRY(c->hsz0.reserve(mimc3::sat_null_bytes(Hp, c->Wp)));
    HIP_TRY(c->hsz1.reserve(mimc3::sat_null_bytes(Hp, c->Wp)));
    HIP_TRY(c->sat_tmp.reserve(mimc3::sat_scratch_bytes(Hp, c->Wp)));
    HIP_TRY(mimc3::launch_sat_u16(static_cast<const unsigned short *>(c->hpl0.p), c->Wp, table_region(c), static_cast<unsigned long long *>(c->hsat0.p),
                                  static_cast<unsigned int *>(c->hsz0.p), c->sat_tmp.p, s));
    HIP_TRY(mimc3::launch_sat_u16(static_cast<const unsigned short *>(c->hpl1.p), c->Wp, table_region(c), static_cast<unsigned long long *>(c->hsat1.p),
                                  static_cast<unsigned int *>(c->hsz1.p), c->sat_tmp.p, s));
    c->sat_u16_ok = true;
    return 0;
}

static int prepare_u8(mimc3_ctx *c, bool planes_built = false)
{
    c->u8_ok = false;
    c->fplanes_ok = false;
    c->sat_u8_ok = false; c->sat_u16_ok = false;
    const int pad = mimc3::kU8Pad;
    c->Wp = (c->W + 2 * pad + 3) & ~3;
    const size_t bytes = (size_t)(c->H + 2 * pad) * c->Wp;
    if (planes_built) {                 // raw 8-bit DN was widened straight into the planes (mimc3_ctx_set_images_u8)
        c->u8_ok = true; c->u16_ok = false; c->hpl_valid = false; c->u8o_ok = false;
        return c->child ? 0 : build_u8_tables(c);
    }
    HIP_TRY(c->pl0.reserve(bytes));
    HIP_TRY(c->pl1.reserve(bytes));
    HIP_TRY(c->flag.reserve(sizeof(int)));
    HIP_TRY(hipMemsetAsync(c->pl0.p, 0, bytes, c->stream));
    HIP_TRY(hipMemsetAsync(c->pl1.p, 0, bytes, c->stream));
    HIP_TRY(hipMemsetAsync(c->flag.p, 0, sizeof(int), c->stream));
    HIP_TRY(mimc3::launch_prep_u8(c->d_i0, c->H, c->W, static_cast<unsigned char *>(c->pl0.p), c->Wp, pad,
                                  static_cast<int *>(c->flag.p), c->stream));
    HIP_TRY(mimc3::launch_prep_u8(c->d_i1, c->H, c->W, static_cast<unsigned char *>(c->pl1.p), c->Wp, pad,
                                  static_cast<int *>(c->flag.p), c->stream));
    int not_u8 = 1;
    HIP_TRY(hipMemcpyAsync(&not_u8, c->flag.p, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->u8_ok = (not_u8 == 0);
    c->u16_ok = false;
    c->hpl_valid = false;
    c->u8o_ok = false;
    if (c->u8_ok && !c->child) RC_TRY(build_u8_tables(c));   // (a chip-atlas context: only if a call needs them, see mimc3_match_ncc_dlc_dev)
    if (!c->u8_ok) {
        // not 8-bit: is the pair "scaled integer" (12-bit DN, or what GMA_float_conv2 makes of 8-bit images:
        // integers / multiples of 1/8)?  Then the exact u16 kernel applies.
        int fl[2] = {3, 3};
        HIP_TRY(c->flag.reserve(2 * sizeof(int)));
        HIP_TRY(hipMemsetAsync(c->flag.p, 0, 2 * sizeof(int), c->stream));
        HIP_TRY(mimc3::launch_detect_scaled_int(c->d_i0, (size_t)c->H * c->W, static_cast<int *>(c->flag.p), c->stream));
        HIP_TRY(mimc3::launch_detect_scaled_int(c->d_i1, (size_t)c->H * c->W, static_cast<int *>(c->flag.p) + 1, c->stream));
        HIP_TRY(hipMemcpyAsync(fl, c->flag.p, 2 * sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        auto pick = [](int f) { return (f & 1) == 0 ? 0 : ((f & 2) == 0 ? 3 : -1); };
        const int s0 = pick(fl[0]), s1 = pick(fl[1]);
        if (s0 >= 0 && s1 >= 0) {
            const size_t hb = sizeof(unsigned short) * (size_t)(c->H + 2 * pad) * c->Wp;
            HIP_TRY(c->hpl0.reserve(hb));
            HIP_TRY(c->hpl1.reserve(hb));
            HIP_TRY(hipMemsetAsync(c->hpl0.p, 0, hb, c->stream));
            HIP_TRY(hipMemsetAsync(c->hpl1.p, 0, hb, c->stream));
            HIP_TRY(mimc3::launch_prep_u16(c->d_i0, c->H, c->W, static_cast<unsigned short *>(c->hpl0.p), c->Wp, pad, s0, c->stream));
            HIP_TRY(mimc3::launch_prep_u16(c->d_i1, c->H, c->W, static_cast<unsigned short *>(c->hpl1.p), c->Wp, pad, s1, c->stream));
            if (!c->child) RC_TRY(build_u16_tables(c, c->stream));
            HIP_TRY(hipStreamSynchronize(c->stream));
            c->shift0 = s0; c->shift1 = s1; c->u16_ok = true; c->hpl_valid = true;
            // 9-bit integers (gradients of 8-bit images): does the LOCAL range fit 8 bits almost everywhere?  Then the
            // u8 kernels can run them through per-point offsets (PxU8o); the few points that do not fit go to PxU16.
            c->u8o_ok = false;
            if (s0 == 0 && s1 == 0 && !c->no_u8o && !getenv("MIMC3_NO_U8O")) {
                int t[4] = {0, 0, 0, 0};
                HIP_TRY(c->flag.reserve(4 * sizeof(int)));
                HIP_TRY(hipMemsetAsync(c->flag.p, 0, 4 * sizeof(int), c->stream));
                HIP_TRY(mimc3::launch_range_tiles(static_cast<const unsigned short *>(c->hpl0.p), c->H, c->W, c->Wp, pad, static_cast<int *>(c->flag.p), c->stream));
                HIP_TRY(mimc3::launch_range_tiles(static_cast<const unsigned short *>(c->hpl1.p), c->H, c->W, c->Wp, pad, static_cast<int *>(c->flag.p) + 2, c->stream));
                HIP_TRY(hipMemcpyAsync(t, c->flag.p, sizeof(t), hipMemcpyDeviceToHost, c->stream));
                HIP_TRY(hipStreamSynchronize(c->stream));
                c->u8o_ok = 2 * t[0] >= t[1] && 2 * t[2] >= t[3];
                if (c->u8o_ok) {        // per-tile ranges: the kernel bounds a point's local range from them before it scans pixels
                    const int Hp = c->H + 2 * pad;
                    const size_t tb = sizeof(uint32_t) * (size_t)((c->Wp + 15) / 16) * ((Hp + 15) / 16);
                    HIP_TRY(c->rt0.reserve(tb));
                    HIP_TRY(c->rt1.reserve(tb));
                    HIP_TRY(mimc3::launch_range_tiles16(static_cast<const unsigned short *>(c->hpl0.p), Hp, c->Wp, static_cast<uint32_t *>(c->rt0.p), c->stream));
                    HIP_TRY(mimc3::launch_range_tiles16(static_cast<const unsigned short *>(c->hpl1.p), Hp, c->Wp, static_cast<uint32_t *>(c->rt1.p), c->stream));
                    HIP_TRY(hipStreamSynchronize(c->stream));
                }
            }
        }
    }
    return 0;
}

extern "C" int mimc3_ctx_set_path(mimc3_ctx *c, int32_t mode)
{
    if (!c || mode < 0 || mode > 4) return mimc3::fail(MIMC3_EINVAL, "mimc3_ctx_set_path: bad argument");
    c->path_mode = mode;
    return 0;
}

extern "C" int mimc3_ctx_last_path(mimc3_ctx *c) { return c ? c->last_path : MIMC3_EINVAL; }

extern "C" int mimc3_ctx_set_images(mimc3_ctx *c, const float *i0, const float *i1, int32_t H, int32_t W)
{
    if (!c || !i0 || !i1 || H <= 0 || W <= 0) return mimc3::fail(MIMC3_EINVAL, "mimc3_ctx_set_images: bad argument");
    HIP_TRY(hipSetDevice(c->device));
    const size_t bytes = sizeof(float) * (size_t)H * W;
    HIP_TRY(c->own_i0.reserve(bytes));
    HIP_TRY(c->own_i1.reserve(bytes));
    RC_TRY(h2d_copy(c, c->own_i0.p, i0, bytes));
    RC_TRY(h2d_copy(c, c->own_i1.p, i1, bytes));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->d_i0 = c->raw_i0 = static_cast<const float *>(c->own_i0.p);
    c->d_i1 = c->raw_i1 = static_cast<const float *>(c->own_i1.p);
    c->H = H; c->W = W; c->filt_live = false;
    return prepare_u8(c);
}

// Raw DN entry points: what the TIFF holds crosses PCIe (1 or 2 bytes per pixel instead of 4) and the widening to f32
// of GMA_float_load_tiff (GMA.c:288-310) runs on the device.  8-bit DN also lands directly in the u8 planes.
extern "C" int mimc3_ctx_set_images_u8(mimc3_ctx *c, const uint8_t *i0, const uint8_t *i1, int32_t H, int32_t W)
{
    if (!c || !i0 || !i1 || H <= 0 || W <= 0) return mimc3::fail(MIMC3_EINVAL, "mimc3_ctx_set_images_u8: bad argument");
    HIP_TRY(hipSetDevice(c->device));
    const size_t npx = (size_t)H * W, half = (npx + 255) & ~(size_t)255;
    HIP_TRY(c->own_i0.reserve(sizeof(float) * npx));
    HIP_TRY(c->own_i1.reserve(sizeof(float) * npx));
    HIP_TRY(c->raw_dn.reserve(2 * half));
    unsigned char *r0 = static_cast<unsigned char *>(c->raw_dn.p), *r1 = r0 + half;
    RC_TRY(h2d_copy(c, r0, i0, npx));
    RC_TRY(h2d_copy(c, r1, i1, npx));
    const int pad = mimc3::kU8Pad;
    c->H = H; c->W = W; c->filt_live = false;
    c->Wp = (W + 2 * pad + 3) & ~3;
    const size_t pbytes = (size_t)(H + 2 * pad) * c->Wp;
    HIP_TRY(c->pl0.reserve(pbytes));
    HIP_TRY(c->pl1.reserve(pbytes));
    HIP_TRY(hipMemsetAsync(c->pl0.p, 0, pbytes, c->stream));
    HIP_TRY(hipMemsetAsync(c->pl1.p, 0, pbytes, c->stream));
    HIP_TRY(mimc3::launch_widen_u8(r0, H, W, static_cast<float *>(c->own_i0.p), static_cast<unsigned char *>(c->pl0.p), c->Wp, pad, c->stream));
    HIP_TRY(mimc3::launch_widen_u8(r1, H, W, static_cast<float *>(c->own_i1.p), static_cast<unsigned char *>(c->pl1.p), c->Wp, pad, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->d_i0 = c->raw_i0 = static_cast<const float *>(c->own_i0.p);
    c->d_i1 = c->raw_i1 = static_cast<const float *>(c->own_i1.p);
    return prepare_u8(c, true);
}

extern "C" int mimc3_ctx_set_images_u16(mimc3_ctx *c, const uint16_t *i0, const uint16_t *i1, int32_t H, int32_t W)
{
    if (!c || !i0 || !i1 || H <= 0 || W <= 0) return mimc3::fail(MIMC3_EINVAL, "mimc3_ctx_set_images_u16: bad argument");
    HIP_TRY(hipSetDevice(c->device));
    const size_t npx = (size_t)H * W, half = (2 * npx + 255) & ~(size_t)255;
    HIP_TRY(c->own_i0.reserve(sizeof(float) * npx));
    HIP_TRY(c->own_i1.reserve(sizeof(float) * npx));
    HIP_TRY(c->raw_dn.reserve(2 * half));
    char *r0 = static_cast<char *>(c->raw_dn.p), *r1 = r0 + half;
    RC_TRY(h2d_copy(c, r0, i0, 2 * npx));
    RC_TRY(h2d_copy(c, r1, i1, 2 * npx));
    HIP_TRY(mimc3::launch_widen_u16(reinterpret_cast<const unsigned short *>(r0), npx, static_cast<float *>(c->own_i0.p), c->stream));
    HIP_TRY(mimc3::launch_widen_u16(reinterpret_cast<const unsigned short *>(r1), npx, static_cast<float *>(c->own_i1.p), c->stream));
    c->d_i0 = c->raw_i0 = static_cast<const float *>(c->own_i0.p);
    c->d_i1 = c->raw_i1 = static_cast<const float *>(c->own_i1.p);
    c->H = H; c->W = W; c->filt_live = false;
    return prepare_u8(c);              // 16-bit files may still hold 8- or 12-bit DN: classified on the device as usual
}

static int set_images_dev_impl(mimc3_ctx *c, const float *d_i0, const float *d_i1, int32_t H, int32_t W, bool producer_unknown)
{
    c->d_i0 = c->raw_i0 = d_i0; c->d_i1 = c->raw_i1 = d_i1; c->H = H; c->W = W; c->filt_live = false;
    HIP_TRY(hipSetDevice(c->device));
    if (producer_unknown) HIP_TRY(hipDeviceSynchronize());   // make the pixels visible whatever stream produced them
    return prepare_u8(c);
}

extern "C" int mimc3_ctx_set_images_dev(mimc3_ctx *c, const float *d_i0, const float *d_i1, int32_t H, int32_t W)
{
    if (!c || !d_i0 || !d_i1 || H <= 0 || W <= 0) return mimc3::fail(MIMC3_EINVAL, "mimc3_ctx_set_images_dev: bad argument");
    return set_images_dev_impl(c, d_i0, d_i1, H, W, true);
}

extern "C" int mimc3_ctx_enable_timing(mimc3_ctx *c, int32_t on)
{
    if (!c) return mimc3::fail(MIMC3_EINVAL, "mimc3_ctx_enable_timing: ctx is NULL");
    c->timing = on != 0; c->timed = false;
    return 0;
}

extern "C" int mimc3_ctx_last_kernel_ms(mimc3_ctx *c, float *ms)
{
    if (!c || !ms) return mimc3::fail(MIMC3_EINVAL, "mimc3_ctx_last_kernel_ms: bad argument");
    if (!c->timed) return mimc3::fail(MIMC3_ESTATE, "mimc3_ctx_last_kernel_ms: no timed launch recorded");
    HIP_TRY(hipEventSynchronize(c->ev1));
    HIP_TRY(hipEventElapsedTime(ms, c->ev0, c->ev1));
    return 0;
}

// ---------------------------------------------------------------------------------------------
// matcher
// ---------------------------------------------------------------------------------------------
// zero-bordered f32 copies of the pair for the register-tiled f32 kernel, once per image pair (enqueued on `s`)
static int build_f32_planes(mimc3_ctx *c, hipStream_t s)
{
    if (c->fplanes_ok) return 0;
    const size_t bytes = sizeof(float) * (size_t)(c->H + 2 * mimc3::kU8Pad) * c->Wp;
    HIP_TRY(c->fpl0.reserve(bytes));
    HIP_TRY(c->fpl1.reserve(bytes));
    HIP_TRY(hipMemsetAsync(c->fpl0.p, 0, bytes, s));
    HIP_TRY(hipMemsetAsync(c->fpl1.p, 0, bytes, s));
    HIP_TRY(mimc3::launch_prep_f32(c->d_i0, c->H, c->W, static_cast<float *>(c->fpl0.p), c->Wp, mimc3::kU8Pad, s));
    HIP_TRY(mimc3::launch_prep_f32(c->d_i1, c->H, c->W, static_cast<float *>(c->fpl1.p), c->Wp, mimc3::kU8Pad, s));
    c->fplanes_ok = true;
    // 16-bit DN and its filtered forms (every pixel, x 1 or x 8, an integer in [0, 2^20)): the f64 sums of the reference are exact integers in any order, and the
    // planes get summed-area tables like the integer planes (one read-back per image pair)
    c->f32i_ok = false;
    if (!getenv("MIMC3_NO_F32_TABLES") && !c->child) {     // (not for the control-point stage's chip atlases: a few hundred latency-bound points)
        int fl[2] = {3, 3};
        HIP_TRY(c->flag.reserve(2 * sizeof(int)));
        HIP_TRY(hipMemsetAsync(c->flag.p, 0, 2 * sizeof(int), s));
        HIP_TRY(mimc3::launch_detect_int16(c->d_i0, (size_t)c->H * c->W, static_cast<int *>(c->flag.p), s));
        HIP_TRY(mimc3::launch_detect_int16(c->d_i1, (size_t)c->H * c->W, static_cast<int *>(c->flag.p) + 1, s));
        HIP_TRY(hipMemcpyAsync(fl, c->flag.p, 2 * sizeof(int), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        auto pick = [](int f) { return (f & 1) == 0 ? 0 : ((f & 2) == 0 ? 3 : -1); };      // integers, or multiples of 1/8 (the Laplacian, MIMC_main.c:188-196)
        const int s0 = pick(fl[0]), s1 = pick(fl[1]);
        if (s0 >= 0 && s1 >= 0) {
            c->fshift0 = s0; c->fshift1 = s1;
            const int Hp = c->H + 2 * mimc3::kU8Pad;
            HIP_TRY(c->fsat0.reserve(mimc3::sat2_bytes(Hp, c->Wp)));
            HIP_TRY(c->fsat1.reserve(mimc3::sat2_bytes(Hp, c->Wp)));
            HIP_TRY(c->sat_tmp.reserve(mimc3::sat2_scratch_bytes(Hp, c->Wp)));
            HIP_TRY(mimc3::launch_sat_f32i(static_cast<const float *>(c->fpl0.p), c->Wp, table_region(c), s0, static_cast<mimc3::Sat2 *>(c->fsat0.p), c->sat_tmp.p, s));
            HIP_TRY(mimc3::launch_sat_f32i(static_cast<const float *>(c->fpl1.p), c->Wp, table_region(c), s1, static_cast<mimc3::Sat2 *>(c->fsat1.p), c->sat_tmp.p, s));
            c->f32i_ok = true;
        }
    }
    return 0;
}

extern "C" int mimc3_match_ncc_dlc_dev(mimc3_ctx *c, const double *d_xyuvav, int32_t N, int32_t off_u, int32_t off_v,
                                       const int32_t *d_piv_uv, const int64_t *d_piv_off, int32_t max_npiv,
                                       int32_t max_abs_piv_u, int32_t max_abs_piv_v, int32_t ocw, int32_t swap,
                                       float *d_out, void *stream)
{
    if (!c || !d_xyuvav || !d_piv_uv || !d_piv_off || !d_out || N <= 0 || max_npiv < 1 || max_abs_piv_u < 0 || max_abs_piv_v < 0)
        return mimc3::fail(MIMC3_EINVAL, "mimc3_match_ncc_dlc_dev: bad argument");
    if (ocw < 1) return mimc3::fail(MIMC3_EINVAL, "mimc3_match_ncc_dlc_dev: ocw must be >= 1");
    if (!c->d_i0 || !c->d_i1) return mimc3::fail(MIMC3_ESTATE, "mimc3_match_ncc_dlc_dev: images not set");
    if (c->child && c->win_half <= 0) return mimc3::fail(MIMC3_ESTATE, "mimc3_match_ncc_dlc_dev: a chip-atlas context only matches full-square search areas");   // (its tables cover the image area only)
    HIP_TRY(hipSetDevice(c->device));
    mimc3::MatchArgs a{};
    a.i0 = c->d_i0; a.i1 = c->d_i1; a.H = c->H; a.W = c->W;
    a.xyuvav = d_xyuvav; a.xy_stride = c->xy_stride; a.xy_col = c->xy_col; a.N = N; a.off_u = off_u; a.off_v = off_v;
    a.piv_uv = d_piv_uv; a.piv_off = d_piv_off; a.ocw = ocw; a.swap = swap ? 1 : 0;
    a.thr = min_dn_threshold();
    a.out = d_out;
    a.win_half = c->win_half;
    hipStream_t s = static_cast<hipStream_t>(stream);
    {   // the general kernel (the last resort of every policy) keeps the cell grid in a global workspace when it outgrows LDS
        const size_t ws = mimc3::match_f32_workspace_bytes(ocw, max_abs_piv_u, max_abs_piv_v, max_npiv, c->win_half);
        if (ws) {
            HIP_TRY(c->cellws.reserve(ws));
            a.cell_ws = static_cast<unsigned char *>(c->cellws.p); a.cell_ws_bytes = c->cellws.cap;
        }
    }
    if (c->timing) HIP_TRY(hipEventRecord(c->ev0, s));
    const int reach_u = max_abs_piv_u + (off_u < 0 ? -off_u : off_u), reach_v = max_abs_piv_v + (off_v < 0 ? -off_v : off_v);
    hipError_t e;
    // a tiled kernel is only chosen when the launch's largest window fits its LDS carve (a long corridor on a big chip
    // does not: 4 B/px at ocw 40 stops fitting at |last pivot| ~47 px); what does not fit takes the next policy down
    // to the general kernel, which can read the window from L2 -- the reference handles every such input
    typedef hipError_t (*px_launcher)(mimc3::MatchU8Args, int, int, int, hipStream_t);
    auto fits = [&](px_launcher fn) {
        mimc3::MatchU8Args probe{};
        probe.ocw = ocw; probe.dry_run = 1; probe.win_half = c->win_half;
        return fn(probe, max_abs_piv_u, max_abs_piv_v, max_npiv, nullptr) == hipSuccess;
    };
    const bool px_ok = mimc3::match_u8_supported(ocw, reach_u, reach_v);
    const bool auto_mode = c->path_mode == 0 || c->path_mode == 4;
    const bool want_u8 = auto_mode && c->u8_ok && px_ok && fits(mimc3::launch_match_u8);
    bool want_u16 = !want_u8 && (auto_mode || c->path_mode == 3) && px_ok && fits(mimc3::launch_match_u16);
    if (want_u16 && !c->u16_ok) {
        if (c->u8_ok && c->path_mode == 3 && !c->hpl_valid) {  // tests: 8-bit pairs are scaled integers too (shift 0)
            const size_t hb = sizeof(unsigned short) * (size_t)(c->H + 2 * mimc3::kU8Pad) * c->Wp;
            HIP_TRY(c->hpl0.reserve(hb));
            HIP_TRY(c->hpl1.reserve(hb));
            HIP_TRY(hipMemsetAsync(c->hpl0.p, 0, hb, s));
            HIP_TRY(hipMemsetAsync(c->hpl1.p, 0, hb, s));
            HIP_TRY(mimc3::launch_prep_u16(c->d_i0, c->H, c->W, static_cast<unsigned short *>(c->hpl0.p), c->Wp, mimc3::kU8Pad, 0, s));
            HIP_TRY(mimc3::launch_prep_u16(c->d_i1, c->H, c->W, static_cast<unsigned short *>(c->hpl1.p), c->Wp, mimc3::kU8Pad, 0, s));
            RC_TRY(build_u16_tables(c, s));
            c->shift0 = c->shift1 = 0;
            c->hpl_valid = true;
        }
        want_u16 = c->u8_ok && c->path_mode == 3;
    }
    const bool want_f32x = !want_u8 && !want_u16 && c->path_mode != 1 && mimc3::match_f32x_supported(ocw, reach_u, reach_v) &&
                           fits(mimc3::launch_match_f32x);
    if (want_u8 || want_u16 || want_f32x) {
        mimc3::MatchU8Args u{};
        u.Wp = c->Wp; u.pad = mimc3::kU8Pad; u.H = c->H; u.W = c->W; u.thr = a.thr;
        u.xyuvav = d_xyuvav; u.xy_stride = c->xy_stride; u.xy_col = c->xy_col; u.N = N; u.off_u = off_u; u.off_v = off_v;
        u.piv_uv = d_piv_uv; u.piv_off = d_piv_off; u.ocw = ocw; u.swap = swap ? 1 : 0; u.out = d_out;
        u.win_half = c->win_half;
        // points whose per-point NCC cache overflows (very long climbs) are appended to a device list and
        // redone by the general kernel right behind, in list mode: no host round trip
        DevBuf &ovf = c->lane ? c->ovf_alt[c->lane - 1] : c->ovf;
        DevBuf &failb = c->lane ? c->fail_alt[c->lane - 1] : c->fail;
        HIP_TRY(ovf.reserve(sizeof(int32_t) * ((size_t)N + 1)));
        HIP_TRY(hipMemsetAsync(ovf.p, 0, sizeof(int32_t), s));
        u.ovf_count = static_cast<int32_t *>(ovf.p);
        u.ovf_list = u.ovf_count + 1;
        // the many-pivot kernel forms (the control-point stage's 21x21 pivot set on its two chip sizes) read no tables
        const bool tables_needed = !(max_npiv > 64 && (ocw == 15 || ocw == 30));
        if (want_u8 && tables_needed && !c->sat_u8_ok) RC_TRY(build_u8_tables(c));
        if (want_u16 && tables_needed && !c->sat_u16_ok) { RC_TRY(build_u16_tables(c, c->stream)); HIP_TRY(hipStreamSynchronize(c->stream)); }
        if (want_u8) {
            u.p0 = static_cast<const unsigned char *>(c->pl0.p); u.p1 = static_cast<const unsigned char *>(c->pl1.p);
            u.sat0 = c->sat0.p; u.sat1 = c->sat1.p; u.sat_ws = mimc3::sat_pitch(c->Wp);
            if (tables_needed && c->path_mode == 0 && mimc3::match_mx_supported(ocw, max_npiv, c->win_half, max_abs_piv_u, max_abs_piv_v)) {
                // dense correlation surfaces on the matrix cores first; the points that kernel does not take (chips with nulls,
                // corridors wider than its tile, ...) are redone by the register-tiled kernel in list mode, no host round trip
                DevBuf &ml = c->mxl[c->lane];
                HIP_TRY(ml.reserve((size_t)N));
                HIP_TRY(hipMemsetAsync(ml.p, 0, (size_t)N, s));
                u.mx_flags = static_cast<uint8_t *>(ml.p);
                u.mx_preflag = (max_abs_piv_u > 29 || max_abs_piv_v > 29) ? 1 : 0;      // some corridors may be wider than the kernel's tile
                e = mimc3::launch_match_mx(u, s);
                if (e == hipSuccess) {
                    u.point_flags = u.mx_flags; u.flag_value = mimc3::kMxRest;
                    e = mimc3::launch_match_u8(u, max_abs_piv_u, max_abs_piv_v, max_npiv, s);
                }
                c->last_path = 5;
            } else {
            e = mimc3::launch_match_u8(u, max_abs_piv_u, max_abs_piv_v, max_npiv, s);
            c->last_path = 1;
            }
        } else if (want_u16) {
            u.p0 = static_cast<const unsigned char *>(c->hpl0.p); u.p1 = static_cast<const unsigned char *>(c->hpl1.p);
            u.scale0 = 1.0 / (double)(1 << c->shift0); u.scale1 = 1.0 / (double)(1 << c->shift1);
            u.sat0 = c->hsat0.p; u.sat1 = c->hsat1.p; u.satz0 = c->hsz0.p; u.satz1 = c->hsz1.p; u.sat_ws = mimc3::sat_pitch(c->Wp);
            if (c->u8o_ok && c->u16_ok && auto_mode) {
                // u8 machinery through per-point offsets first; what does not fit is redone by the u16 kernel in list mode
                HIP_TRY(failb.reserve(sizeof(int32_t) * ((size_t)N + 1)));
                HIP_TRY(hipMemsetAsync(failb.p, 0, sizeof(int32_t), s));
                u.fail_count = static_cast<int32_t *>(failb.p);
                u.fail_list = u.fail_count + 1;
                if (!getenv("MIMC3_NO_RANGE_TILES")) { u.rt0 = static_cast<const uint32_t *>(c->rt0.p); u.rt1 = static_cast<const uint32_t *>(c->rt1.p); u.rt_tw = (c->Wp + 15) / 16; }
                e = mimc3::launch_match_u8o(u, max_abs_piv_u, max_abs_piv_v, max_npiv, s);
                u.rt0 = u.rt1 = nullptr;
                if (e == hipSuccess) {
                    u.point_count = u.fail_count; u.point_list = u.fail_list;
                    u.fail_count = nullptr; u.fail_list = nullptr;
                    e = mimc3::launch_match_u16(u, max_abs_piv_u, max_abs_piv_v, max_npiv, s);
                }
                c->last_path = 4;
            } else {
                e = mimc3::launch_match_u16(u, max_abs_piv_u, max_abs_piv_v, max_npiv, s);
                c->last_path = 3;
            }
        } else {
            RC_TRY(build_f32_planes(c, s));
            u.p0 = static_cast<const unsigned char *>(c->fpl0.p); u.p1 = static_cast<const unsigned char *>(c->fpl1.p);
            if (c->f32i_ok) {
                u.sat0 = c->fsat0.p; u.sat1 = c->fsat1.p; u.sat_ws = mimc3::sat_pitch(c->Wp);
                u.scale0 = 1.0 / (double)(1 << c->fshift0); u.scale1 = 1.0 / (double)(1 << c->fshift1);
            }
            e = mimc3::launch_match_f32x(u, max_abs_piv_u, max_abs_piv_v, max_npiv, s);
            c->last_path = 2;
        }
        if (e == hipSuccess) {
            a.point_count = u.ovf_count;
            a.point_list = u.ovf_list;
            e = mimc3::launch_match_f32(a, max_abs_piv_u, max_abs_piv_v, max_npiv, s);
        }
    } else {
        e = mimc3::launch_match_f32(a, max_abs_piv_u, max_abs_piv_v, max_npiv, s);
        c->last_path = 0;
    }
    if (e != hipSuccess) return mimc3::hip_fail(e, "match kernel launch");
    if (c->timing) { HIP_TRY(hipEventRecord(c->ev1, s)); c->timed = true; }
    return 0;
}

extern "C" int mimc3_match_ncc_dlc(mimc3_ctx *c, const double *xyuvav, int32_t N, const int32_t offset[2],
                                   const int32_t *piv_uv, const int64_t *piv_off, int32_t ocw, int32_t swap, float *out)
{
    if (!c || !xyuvav || !offset || !piv_uv || !piv_off || !out || N <= 0)
        return mimc3::fail(MIMC3_EINVAL, "mimc3_match_ncc_dlc: bad argument");
    if (!c->d_i0 || !c->d_i1) return mimc3::fail(MIMC3_ESTATE, "mimc3_match_ncc_dlc: images not set");
    // The reference reads the chip without any bounds check (MIMC_module.c:852) and overflows on an
    // empty pivot list (:589-591).  Refuse those inputs instead of reproducing undefined behaviour.
    for (int32_t g = 0; g < N; ++g) {
        const int32_t u0 = (int32_t)xyuvav[6 * (size_t)g + 2], v0 = (int32_t)xyuvav[6 * (size_t)g + 3];
        if (u0 - ocw < 0 || u0 + ocw >= c->W || v0 - ocw < 0 || v0 + ocw >= c->H)
            return mimc3::fail(MIMC3_EBOUNDS, "mimc3_match_ncc_dlc: grid point " + std::to_string(g) + " chip leaves the image");
    }
    int32_t mn = 0, mu = 0, mv = 0;
    int rc = mimc3_pivot_extent(piv_uv, piv_off, N, &mn, &mu, &mv);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(c->device));
    const size_t P = (size_t)piv_off[N];
    HIP_TRY(c->xy.reserve(sizeof(double) * 6 * (size_t)N));
    HIP_TRY(c->puv.reserve(sizeof(int32_t) * 2 * P));
    HIP_TRY(c->poff.reserve(sizeof(int64_t) * ((size_t)N + 1)));
    HIP_TRY(c->out.reserve(sizeof(float) * 3 * (size_t)N));
    RC_TRY(h2d_copy(c, c->xy.p, xyuvav, sizeof(double) * 6 * (size_t)N));
    RC_TRY(h2d_copy(c, c->puv.p, piv_uv, sizeof(int32_t) * 2 * P));
    RC_TRY(h2d_copy(c, c->poff.p, piv_off, sizeof(int64_t) * ((size_t)N + 1)));
    rc = mimc3_match_ncc_dlc_dev(c, static_cast<const double *>(c->xy.p), N, offset[0], offset[1],
                                 static_cast<const int32_t *>(c->puv.p), static_cast<const int64_t *>(c->poff.p), mn, mu, mv,
                                 ocw, swap, static_cast<float *>(c->out.p), c->stream);
    if (rc) return rc;
    return d2h_copy(c, out, c->out.p, sizeof(float) * 3 * (size_t)N);
}

// ---------------------------------------------------------------------------------------------
// a2 on the device: pivot lists expanded from per-point corridors (pivot_kernel.hip)
// ---------------------------------------------------------------------------------------------
static_assert(sizeof(mimc3::CorridorPOD) == sizeof(mimc3::CorridorDev) && offsetof(mimc3::CorridorPOD, length) == offsetof(mimc3::CorridorDev, length),
              "host and device corridor records share one layout");

extern "C" int mimc3_pivot_corridors(const double *xyuvav, int32_t N, float dt, float mpp, float aw_sf, float aw_cre, void *cor)
{
    if (!xyuvav || !cor || N <= 0) return mimc3::fail(MIMC3_EINVAL, "mimc3_pivot_corridors: bad argument");
    mimc3::pivot_corridors(xyuvav, N, dt, mpp, aw_sf, aw_cre, static_cast<mimc3::CorridorPOD *>(cor));
    return 0;
}

// counts + offsets, then ONE 24-byte read-back (synchronises `s`): total pivots and the extents a matcher launch is sized by
static int pivots_count(mimc3_ctx *c, const double *d_xy, const void *d_cor, int32_t N, int32_t ocw, int64_t *d_off, hipStream_t s, int64_t *total,
                        int32_t ext[3])
{
    HIP_TRY(c->pcnt.reserve(sizeof(int32_t) * (size_t)N));
    HIP_TRY(c->pext.reserve(64));
    HIP_TRY(mimc3::launch_pivot_count(d_xy, 6, 2, static_cast<const mimc3::CorridorDev *>(d_cor), N, ocw, c->H, c->W, static_cast<int32_t *>(c->pcnt.p), d_off,
                                      static_cast<int32_t *>(c->pext.p), s));
    int32_t h[6] = {0, 0, 0, 0, 0, 0};
    HIP_TRY(hipMemcpyAsync(h, c->pext.p, sizeof(h), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    int64_t tot = 0;
    std::memcpy(&tot, &h[4], sizeof(tot));
    *total = tot;
    ext[0] = h[0]; ext[1] = h[1]; ext[2] = h[2];
    if (h[3]) return mimc3::fail(MIMC3_EBOUNDS, "mimc3_get_uv_pivot_dev: a grid point has zero pivots (too close to the image edge)");
    return 0;
}

extern "C" int mimc3_get_uv_pivot_dev(mimc3_ctx *c, const double *d_xyuvav, const void *d_cor, int32_t N, int32_t ocw, int64_t *d_piv_off,
                                      int32_t *d_piv_uv, int32_t *d_piv_uv_neg, int64_t cap, int64_t *total, int32_t extent[3], void *stream)
{
    if (!c || !d_xyuvav || !d_cor || !d_piv_off || !total || !extent || N <= 0 || ocw < 1)
        return mimc3::fail(MIMC3_EINVAL, "mimc3_get_uv_pivot_dev: bad argument");
    if (c->H <= 0 || c->W <= 0) return mimc3::fail(MIMC3_ESTATE, "mimc3_get_uv_pivot_dev: images not set (the image size bounds the pivots)");
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    RC_TRY(pivots_count(c, d_xyuvav, d_cor, N, ocw, d_piv_off, s, total, extent));
    if (!d_piv_uv && !d_piv_uv_neg) return 0;               // two-call protocol, as mimc3_get_uv_pivot
    if (cap < *total) return mimc3::fail(MIMC3_ECAP, "mimc3_get_uv_pivot_dev: pivot capacity too small");
    HIP_TRY(mimc3::launch_pivot_fill(static_cast<const mimc3::CorridorDev *>(d_cor), d_piv_off, N, d_piv_uv, d_piv_uv_neg, s));
    return 0;
}

// get_uv_pivot + matching_ncc_dlc_2 (MIMC_main.c:264-267 / :281-284) in one call, corridors given (host).  What crosses PCIe per
// grid point: its (u, v) (16 B: the matcher and the pivot kernel read nothing else of an xyuvav row), its corridor (24 B), its
// result (12 B).  The grid goes through in chunks: uploads + pivot counts on one copy stream, lists + matcher on the context's
// stream, downloads on a second copy stream -- the transfers of chunk k+1 / k-1 run under the matcher of chunk k.
// `produce`, when given, fills cor[g0, g1) on the host right before that chunk is packed and sent: the corridors of chunk k+1 are
// made (threaded libm work) while the device matches chunk k
static int match_cor_impl(mimc3_ctx *c, const double *xyuvav, const void *cor, int32_t N, const int32_t offset[2], int32_t ocw,
                          int32_t swap, float *out, const std::function<void(int32_t, int32_t)> *produce)
{
    if (!c || !xyuvav || !cor || !offset || !out || N <= 0) return mimc3::fail(MIMC3_EINVAL, "mimc3_match_ncc_dlc_cor: bad argument");
    if (!c->d_i0 || !c->d_i1) return mimc3::fail(MIMC3_ESTATE, "mimc3_match_ncc_dlc_cor: images not set");
    if (!c->aux[0] || !c->aux[1]) return mimc3::fail(MIMC3_ESTATE, "mimc3_match_ncc_dlc_cor: the context has no copy streams");
    HIP_TRY(hipSetDevice(c->device));
    const mimc3::CorridorPOD *hc = static_cast<const mimc3::CorridorPOD *>(cor);
    static const bool io_tm = getenv("MIMC3_IO_TIMING") != nullptr;
    auto t_prev = std::chrono::steady_clock::now();
    auto lap = [&](const char *w) {
        if (!io_tm) return;
        const auto t = std::chrono::steady_clock::now();
        fprintf(stderr, "[mimc3 io] %-18s %7.3f ms\n", w, std::chrono::duration<double, std::milli>(t - t_prev).count());
        t_prev = t;
    };
    // pack (u, v), check the chips against the image (see mimc3_match_ncc_dlc), bound the list sizes: n(g) <= length / |step| + 2
    void *huv_v = nullptr;
    RC_TRY(mimc3_ctx_host_workspace(c, 6, 16 * (size_t)N, &huv_v));                      // pinned, kept across calls
    double *huv = static_cast<double *>(huv_v);
    static const int k_env = getenv("MIMC3_IO_CHUNKS") ? atoi(getenv("MIMC3_IO_CHUNKS")) : 0;      // tuning: 1..8
    const int K = N >= 40000 ? (k_env >= 1 && k_env <= 8 ? k_env : 3) : 1;     // (measured at 200,000 points: 1 chunk 4.5 ms, 2: 4.1, 3: 3.75, 4: 3.9, 8: 4.5)
    // a small first chunk (1/8 of the grid) gets the device going early; the rest is cut evenly
    int32_t lo[9];
    lo[0] = 0;
    for (int k = 1; k <= K; k++) lo[k] = K == 1 ? N : (int32_t)((int64_t)N / 8 + ((int64_t)N - N / 8) * (k - 1) / (K - 1));
    lo[K] = N;
    if (K > 1 && lo[1] == 0) lo[1] = 1;
    // the list buffer keeps its size from call to call (first call: 24 pivots per point); a chunk that does not fit makes it grow
    size_t uv_cap = c->puv.cap / 8;
    if (uv_cap < 24 * (size_t)N) uv_cap = 24 * (size_t)N;
    HIP_TRY(c->puv.reserve(8 * uv_cap));
    HIP_TRY(c->xy.reserve(16 * (size_t)N));
    HIP_TRY(c->pcor.reserve(sizeof(mimc3::CorridorPOD) * (size_t)N));
    HIP_TRY(c->poff.reserve(sizeof(int64_t) * ((size_t)N + K)));
    HIP_TRY(c->pcnt.reserve(sizeof(int32_t) * (size_t)N));
    HIP_TRY(c->pext.reserve(64 * (size_t)K));
    HIP_TRY(c->out.reserve(sizeof(float) * 3 * (size_t)N));
    void *hext_v = nullptr;
    RC_TRY(mimc3_ctx_host_workspace(c, 5, 64 * (size_t)K, &hext_v));
    int32_t *hext = static_cast<int32_t *>(hext_v);
    for (int j = 0; j < 2; j++)
        for (int k = 0; k < K; k++)
            if (!c->ev_chunk[j][k]) HIP_TRY(hipEventCreateWithFlags(&c->ev_chunk[j][k], hipEventDisableTiming));
    hipStream_t up = c->aux[0], down = c->aux[1], s = c->stream;
    double *d_uv = static_cast<double *>(c->xy.p);
    char *d_cor = static_cast<char *>(c->pcor.p);
    // ---- copy stream, chunk by chunk: the host packs (u, v) and checks the chips against the image (see mimc3_match_ncc_dlc);
    //      (u, v) + corridors up, pivot counts + offsets, the 24 bytes that size lists and launch back
    auto upload = [&](int k) -> int {
        const size_t g0 = (size_t)lo[k], n = (size_t)(lo[k + 1] - lo[k]);
        if (produce) (*produce)(lo[k], lo[k + 1]);
        for (int32_t g = lo[k]; g < lo[k + 1]; ++g) {
            const double gu = xyuvav[6 * (size_t)g + 2], gv = xyuvav[6 * (size_t)g + 3];
            const int32_t u0 = (int32_t)gu, v0 = (int32_t)gv;
            if (u0 - ocw < 0 || u0 + ocw >= c->W || v0 - ocw < 0 || v0 + ocw >= c->H) {
                (void)hipStreamSynchronize(up);
                return mimc3::fail(MIMC3_EBOUNDS, "mimc3_match_ncc_dlc_cor: grid point " + std::to_string(g) + " chip leaves the image");
            }
            huv[2 * (size_t)g] = gu; huv[2 * (size_t)g + 1] = gv;
        }
        HIP_TRY(hipMemcpyAsync(d_uv + 2 * g0, huv + 2 * g0, 16 * n, hipMemcpyHostToDevice, up));
        HIP_TRY(hipMemcpyAsync(d_cor + sizeof(mimc3::CorridorPOD) * g0, hc + g0, sizeof(mimc3::CorridorPOD) * n, hipMemcpyHostToDevice, up));
        HIP_TRY(mimc3::launch_pivot_count(d_uv + 2 * g0, 2, 0, reinterpret_cast<const mimc3::CorridorDev *>(d_cor) + g0, (int)n, ocw, c->H, c->W,
                                          static_cast<int32_t *>(c->pcnt.p) + g0, static_cast<int64_t *>(c->poff.p) + g0 + k,
                                          reinterpret_cast<int32_t *>(static_cast<char *>(c->pext.p) + 64 * (size_t)k), up));
        HIP_TRY(hipMemcpyAsync(hext + 16 * k, static_cast<char *>(c->pext.p) + 64 * (size_t)k, 24, hipMemcpyDeviceToHost, up));
        HIP_TRY(hipEventRecord(c->ev_chunk[0][k], up));
        return 0;
    };
    // ---- the context's stream: lists + matcher of a chunk; second copy stream: its results down
    int rc = 0;
    int64_t uv_base = 0;
    const int32_t keep_stride = c->xy_stride, keep_col = c->xy_col;
    c->xy_stride = 2; c->xy_col = 0;
    // chunks alternate between two streams (and two sets of per-call scratch: c->lane), so that the first launches of chunk k+1 run
    // under the tail of chunk k's last one
    static const int two_env = getenv("MIMC3_IO_TWO_STREAMS") ? atoi(getenv("MIMC3_IO_TWO_STREAMS")) : 1;      // tuning / A-B
    const bool two = two_env != 0 && K > 1 && c->aux[2] != nullptr;
    auto process = [&](int k) -> int {
        const size_t g0 = (size_t)lo[k];
        const int32_t n = lo[k + 1] - lo[k];
        hipStream_t s = (two && (k & 1)) ? c->aux[2] : c->stream;
        c->lane = (two && (k & 1)) ? 1 : 0;
        hipError_t e = hipEventSynchronize(c->ev_chunk[0][k]);
        if (e != hipSuccess) { rc = mimc3::hip_fail(e, "chunk upload"); return rc; }
        const int32_t *h = hext + 16 * k;
        int64_t total = 0;
        std::memcpy(&total, &h[4], sizeof(total));
        if (h[3]) { rc = mimc3::fail(MIMC3_EBOUNDS, "mimc3_match_ncc_dlc_cor: a grid point has zero pivots (too close to the image edge)"); return rc; }
        if ((size_t)(uv_base + total) > uv_cap) {
            // the lists of this chunk do not fit behind the earlier ones: let those finish, then start over in a bigger buffer
            e = hipStreamSynchronize(c->stream);
            if (e == hipSuccess && two) e = hipStreamSynchronize(c->aux[2]);
            if (e != hipSuccess) { rc = mimc3::hip_fail(e, "pivot lists"); return rc; }
            uv_cap = 2 * (size_t)total > uv_cap ? 2 * (size_t)total + 2 * (size_t)(N - lo[k]) * 24 : 2 * uv_cap;
            e = c->puv.reserve(8 * uv_cap);
            if (e != hipSuccess) { rc = mimc3::hip_fail(e, "pivot lists"); return rc; }
            uv_base = 0;
        }
        int32_t *uv = static_cast<int32_t *>(c->puv.p) + 2 * uv_base;
        const int64_t *off = static_cast<const int64_t *>(c->poff.p) + g0 + k;
        e = hipStreamWaitEvent(s, c->ev_chunk[0][k], 0);
        // (the general kernel's global cell workspace -- windows that outgrow LDS altogether -- is one per context: such a chunk waits
        //  for its predecessor on the other stream)
        if (e == hipSuccess && two && k > 0 && mimc3::match_f32_workspace_bytes(ocw, h[1], h[2], h[0], c->win_half) != 0)
            e = hipStreamWaitEvent(s, c->ev_chunk[1][k - 1], 0);
        if (e == hipSuccess) e = mimc3::launch_pivot_fill(reinterpret_cast<const mimc3::CorridorDev *>(d_cor) + g0, off, n, swap ? nullptr : uv, swap ? uv : nullptr, s);
        if (e != hipSuccess) { rc = mimc3::hip_fail(e, "pivot lists"); return rc; }
        float *d_out = static_cast<float *>(c->out.p) + 3 * g0;
        rc = mimc3_match_ncc_dlc_dev(c, d_uv + 2 * g0, n, offset[0], offset[1], uv, off, h[0], h[1], h[2], ocw, swap, d_out, s);
        if (rc) return rc;
        e = hipEventRecord(c->ev_chunk[1][k], s);
        if (e == hipSuccess) e = hipStreamWaitEvent(down, c->ev_chunk[1][k], 0);
        if (e == hipSuccess) e = hipMemcpyAsync(out + 3 * g0, d_out, sizeof(float) * 3 * (size_t)n, hipMemcpyDeviceToHost, down);
        if (e != hipSuccess) { rc = mimc3::hip_fail(e, "result download"); return rc; }
        uv_base += total;
        return 0;
    };
    // chunk k+1 is packed and sent off before chunk k's matcher is enqueued: the device never waits for the host's loop
    // (the same order with `produce`: enqueuing chunk k's matcher BEFORE making chunk k+1's corridors measured 3.78 ms per pass against
    // 3.35 -- the matcher's enqueue waits for the chunk's list sizes to come back, and that wait is where the host has time to spare)
    rc = upload(0);
    for (int k = 0; k < K && !rc; k++) {
        if (k + 1 < K) rc = upload(k + 1);
        if (!rc) rc = process(k);
    }
    c->xy_stride = keep_stride; c->xy_col = keep_col;
    c->lane = 0;
    lap("chunks enqueued");
    // every stream drains before the buffers are reused (also on the error paths)
    hipError_t e1 = hipStreamSynchronize(up), e2 = hipStreamSynchronize(s), e3 = hipStreamSynchronize(down);
    if (two) { const hipError_t e4 = hipStreamSynchronize(c->aux[2]); if (e2 == hipSuccess) e2 = e4; }
    lap("drained");
    if (rc) return rc;
    if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess) return mimc3::hip_fail(e1 != hipSuccess ? e1 : (e2 != hipSuccess ? e2 : e3), "mimc3_match_ncc_dlc_cor");
    return 0;
}

extern "C" int mimc3_match_ncc_dlc_cor(mimc3_ctx *c, const double *xyuvav, const void *cor, int32_t N, const int32_t offset[2], int32_t ocw,
                                       int32_t swap, float *out)
{
    return match_cor_impl(c, xyuvav, cor, N, offset, ocw, swap, out, nullptr);
}

// the same with the corridors made here (the libm half of get_uv_pivot: threaded host code), chunk by chunk under the device's work
extern "C" int mimc3_match_ncc_dlc_geo(mimc3_ctx *c, const double *xyuvav, int32_t N, const int32_t offset[2], float dt, float mpp, float aw_sf,
                                       float aw_cre, int32_t ocw, int32_t swap, float *out)
{
    if (!c || !xyuvav || !offset || !out || N <= 0) return mimc3::fail(MIMC3_EINVAL, "mimc3_match_ncc_dlc_geo: bad argument");
    HIP_TRY(hipSetDevice(c->device));
    void *hcor = nullptr;
    RC_TRY(mimc3_ctx_host_workspace(c, 7, sizeof(mimc3::CorridorPOD) * (size_t)N, &hcor));      // pinned, kept across calls
    mimc3::CorridorPOD *hc = static_cast<mimc3::CorridorPOD *>(hcor);
    const std::function<void(int32_t, int32_t)> produce = [&](int32_t g0, int32_t g1) {
        mimc3::pivot_corridors(xyuvav + 6 * (size_t)g0, g1 - g0, dt, mpp, aw_sf, aw_cre, hc + g0);
    };
    return match_cor_impl(c, xyuvav, hcor, N, offset, ocw, swap, out, &produce);
}

// ---------------------------------------------------------------------------------------------
// QM pseudo-smoothing
// ---------------------------------------------------------------------------------------------
extern "C" int32_t mimc3_qm_launches_per_sweep(void) { return mimc3::kQmLaunchesPerSweep; }

extern "C" int64_t mimc3_qm_workspace_bytes(int32_t ngrid, int32_t max_sweeps)
{
    if (ngrid <= 0 || max_sweeps <= 0) return 0;
    return mimc3::qm_workspace_bytes(ngrid, max_sweeps);
}

extern "C" int mimc3_qm_pseudosmooth_dev(mimc3_ctx *c, int32_t dimy, int32_t dimx, int32_t *d_dpf, float *d_dpf_dx,
                                         float *d_dpf_dy, const int32_t *d_ruv, int32_t nn, const float *d_mvn, int32_t Kmax,
                                         const int32_t *d_nclus, const double *d_xyuvav, int32_t max_sweeps,
                                         void *d_work, int32_t *d_sweeps_done, void *stream)
{
    if (!c || !d_dpf || !d_dpf_dx || !d_dpf_dy || !d_ruv || !d_mvn || !d_nclus || !d_xyuvav || !d_work ||
        dimx <= 0 || dimy <= 0 || nn <= 0 || Kmax <= 0 || max_sweeps <= 0)
        return mimc3::fail(MIMC3_EINVAL, "mimc3_qm_pseudosmooth_dev: bad argument");
    HIP_TRY(hipSetDevice(c->device));
    mimc3::QmArgs a{};
    a.dimy = dimy; a.dimx = dimx; a.N = dimx * dimy;
    a.dpf = d_dpf; a.dx = d_dpf_dx; a.dy = d_dpf_dy; a.ruv = d_ruv; a.nn = nn; a.mvn = d_mvn; a.Kmax = Kmax;
    a.nclus = d_nclus; a.xyuvav = d_xyuvav; a.max_sweeps = max_sweeps;
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipError_t e = mimc3::launch_qm(a, d_work, s);
    if (e != hipSuccess) return mimc3::hip_fail(e, "qm kernel launch");
    if (d_sweeps_done) {
        const char *flags = static_cast<const char *>(d_work) + mimc3::qm_flags_offset_bytes(a.N);
        HIP_TRY(hipMemcpyAsync(d_sweeps_done, flags + 4 * mimc3::kQmSweeps, sizeof(int32_t), hipMemcpyDeviceToDevice, s));
    }
    return 0;
}

extern "C" int mimc3_qm_pseudosmooth(mimc3_ctx *c, int32_t dimy, int32_t dimx, int32_t *dpf, float *dpf_dx, float *dpf_dy,
                                     const int32_t *ruv, int32_t nn, const float *mvn, int32_t Kmax, const int32_t *nclus,
                                     const double *xyuvav, int32_t max_sweeps, int32_t *sweeps_done)
{
    if (!c || !dpf || !dpf_dx || !dpf_dy || !ruv || !mvn || !nclus || !xyuvav || dimx <= 0 || dimy <= 0 || nn <= 0 ||
        Kmax <= 0 || max_sweeps <= 0)
        return mimc3::fail(MIMC3_EINVAL, "mimc3_qm_pseudosmooth: bad argument");
    const size_t N = (size_t)dimx * dimy;
    for (size_t i = 0; i < N; ++i)
        if (nclus[i] < 0 || nclus[i] > Kmax || dpf[i] >= Kmax)
            return mimc3::fail(MIMC3_EINVAL, "mimc3_qm_pseudosmooth: cluster count/id exceeds Kmax");
    HIP_TRY(hipSetDevice(c->device));
    // one staging buffer, 256-byte aligned sections
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t o_dpf = 0, o_dx = o_dpf + al(4 * N), o_dy = o_dx + al(4 * N), o_ruv = o_dy + al(4 * N),
                 o_mvn = o_ruv + al(8 * (size_t)nn), o_ncl = o_mvn + al(20 * N * Kmax), o_xy = o_ncl + al(4 * N),
                 o_swp = o_xy + al(48 * N), total = o_swp + 256;
    HIP_TRY(c->qm_io.reserve(total));
    HIP_TRY(c->qm_work.reserve((size_t)mimc3::qm_workspace_bytes((int32_t)N, max_sweeps)));
    char *b = static_cast<char *>(c->qm_io.p);
    hipStream_t s = c->stream;
    HIP_TRY(hipMemcpyAsync(b + o_dpf, dpf, 4 * N, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(b + o_dx, dpf_dx, 4 * N, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(b + o_dy, dpf_dy, 4 * N, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(b + o_ruv, ruv, 8 * (size_t)nn, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(b + o_mvn, mvn, 20 * N * Kmax, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(b + o_ncl, nclus, 4 * N, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(b + o_xy, xyuvav, 48 * N, hipMemcpyHostToDevice, s));
    int rc = mimc3_qm_pseudosmooth_dev(c, dimy, dimx, reinterpret_cast<int32_t *>(b + o_dpf), reinterpret_cast<float *>(b + o_dx),
                                       reinterpret_cast<float *>(b + o_dy), reinterpret_cast<const int32_t *>(b + o_ruv), nn,
                                       reinterpret_cast<const float *>(b + o_mvn), Kmax, reinterpret_cast<const int32_t *>(b + o_ncl),
                                       reinterpret_cast<const double *>(b + o_xy), max_sweeps, c->qm_work.p,
                                       reinterpret_cast<int32_t *>(b + o_swp), s);
    if (rc) return rc;
    int32_t sw = 0;
    HIP_TRY(hipMemcpyAsync(dpf, b + o_dpf, 4 * N, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(dpf_dx, b + o_dx, 4 * N, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(dpf_dy, b + o_dy, 4 * N, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(&sw, b + o_swp, 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    if (sweeps_done) *sweeps_done = sw;
    return 0;
}

// ---------------------------------------------------------------------------------------------
// N1: candidate clustering, dpf0, dpf1
// ---------------------------------------------------------------------------------------------
static inline size_t al256(size_t b) { return (b + 255) & ~(size_t)255; }

extern "C" int mimc3_cluster_candidates_dev(mimc3_ctx *c, const float *d_dp, int32_t ndp, int32_t N, int32_t Kmax,
                                            float *d_mvn, int32_t *d_nclus, int32_t *d_kmax_seen, void *stream)
{
    if (!c || !d_dp || !d_mvn || !d_nclus || !d_kmax_seen || N <= 0 || Kmax <= 0 || ndp <= 0 || ndp > mimc3::kCluMaxPasses)
        return mimc3::fail(MIMC3_EINVAL, "mimc3_cluster_candidates_dev: bad argument (1 <= ndp <= 64)");
    HIP_TRY(hipSetDevice(c->device));
    mimc3::ClusterArgs a{};
    a.dp = d_dp; a.ndp = ndp; a.N = N; a.Kmax = Kmax; a.mvn = d_mvn; a.nclus = d_nclus; a.kmax_seen = d_kmax_seen;
    hipError_t e = mimc3::launch_cluster(a, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return mimc3::hip_fail(e, "cluster kernel launch");
    return 0;
}

extern "C" int mimc3_cluster_candidates(mimc3_ctx *c, const float *dp, int32_t ndp, int32_t N, int32_t Kmax, float *mvn,
                                        int32_t *nclus, int32_t *kmax_seen)
{
    if (!c || !dp || !mvn || !nclus || N <= 0 || Kmax <= 0 || ndp <= 0 || ndp > mimc3::kCluMaxPasses)
        return mimc3::fail(MIMC3_EINVAL, "mimc3_cluster_candidates: bad argument (1 <= ndp <= 64)");
    HIP_TRY(hipSetDevice(c->device));
    const size_t b_dp = 12 * (size_t)ndp * N, b_mvn = 20 * (size_t)N * Kmax, b_ncl = 4 * (size_t)N;
    const size_t o_dp = 0, o_mvn = o_dp + al256(b_dp), o_ncl = o_mvn + al256(b_mvn), o_k = o_ncl + al256(b_ncl), total = o_k + 256;
    HIP_TRY(c->n1_io.reserve(total));
    char *b = static_cast<char *>(c->n1_io.p);
    hipStream_t s = c->stream;
    HIP_TRY(hipMemcpyAsync(b + o_dp, dp, b_dp, hipMemcpyHostToDevice, s));
    int rc = mimc3_cluster_candidates_dev(c, reinterpret_cast<const float *>(b + o_dp), ndp, N, Kmax, reinterpret_cast<float *>(b + o_mvn),
                                          reinterpret_cast<int32_t *>(b + o_ncl), reinterpret_cast<int32_t *>(b + o_k), s);
    if (rc) return rc;
    int32_t k = 0;
    HIP_TRY(hipMemcpyAsync(mvn, b + o_mvn, b_mvn, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(nclus, b + o_ncl, b_ncl, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(&k, b + o_k, 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    if (kmax_seen) *kmax_seen = k;
    if (k > Kmax) return mimc3::fail(MIMC3_ECAP, "mimc3_cluster_candidates: a grid point has more clusters than Kmax");
    return 0;
}

extern "C" int mimc3_get_dpf0_dev(mimc3_ctx *c, const float *d_mvn, const int32_t *d_nclus, int32_t N, int32_t Kmax,
                                  float min_ratio, int32_t *d_dpf, void *stream)
{
    if (!c || !d_mvn || !d_nclus || !d_dpf || N <= 0 || Kmax <= 0)
        return mimc3::fail(MIMC3_EINVAL, "mimc3_get_dpf0_dev: bad argument");
    HIP_TRY(hipSetDevice(c->device));
    hipError_t e = mimc3::launch_dpf0(d_mvn, d_nclus, N, Kmax, min_ratio, d_dpf, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return mimc3::hip_fail(e, "dpf0 kernel launch");
    return 0;
}

extern "C" int mimc3_get_dpf0(mimc3_ctx *c, const float *mvn, const int32_t *nclus, int32_t N, int32_t Kmax, float min_ratio,
                              int32_t *dpf)
{
    if (!c || !mvn || !nclus || !dpf || N <= 0 || Kmax <= 0) return mimc3::fail(MIMC3_EINVAL, "mimc3_get_dpf0: bad argument");
    for (int32_t i = 0; i < N; ++i)
        if (nclus[i] < 0 || nclus[i] > Kmax) return mimc3::fail(MIMC3_EINVAL, "mimc3_get_dpf0: cluster count exceeds Kmax");
    HIP_TRY(hipSetDevice(c->device));
    const size_t b_mvn = 20 * (size_t)N * Kmax, b_n = 4 * (size_t)N;
    const size_t o_mvn = 0, o_ncl = al256(b_mvn), o_dpf = o_ncl + al256(b_n), total = o_dpf + al256(b_n);
    HIP_TRY(c->n1_io.reserve(total));
    char *b = static_cast<char *>(c->n1_io.p);
    hipStream_t s = c->stream;
    HIP_TRY(hipMemcpyAsync(b + o_mvn, mvn, b_mvn, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(b + o_ncl, nclus, b_n, hipMemcpyHostToDevice, s));
    int rc = mimc3_get_dpf0_dev(c, reinterpret_cast<const float *>(b + o_mvn), reinterpret_cast<const int32_t *>(b + o_ncl), N, Kmax,
                                min_ratio, reinterpret_cast<int32_t *>(b + o_dpf), s);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(dpf, b + o_dpf, b_n, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return 0;
}

extern "C" int64_t mimc3_dpf1_workspace_bytes(int32_t ngrid) { return ngrid > 0 ? mimc3::dpf1_workspace_bytes(ngrid) : 0; }

extern "C" int mimc3_get_dpf1_dev(mimc3_ctx *c, int32_t dimy, int32_t dimx, int32_t *d_dpf, float *d_dpf_dx, float *d_dpf_dy,
                                  const int32_t *d_ruv, int32_t nn, const float *d_mvn, int32_t Kmax, const int32_t *d_nclus,
                                  const double *d_xyuvav, float dt, float mpp, void *d_work, int32_t *sweeps_done, void *stream)
{
    if (!c || !d_dpf || !d_dpf_dx || !d_dpf_dy || !d_ruv || !d_mvn || !d_nclus || !d_xyuvav || !d_work || dimx <= 0 ||
        dimy <= 0 || nn <= 0 || Kmax <= 0)
        return mimc3::fail(MIMC3_EINVAL, "mimc3_get_dpf1_dev: bad argument");
    HIP_TRY(hipSetDevice(c->device));
    mimc3::Dpf1Args a{};
    a.dimy = dimy; a.dimx = dimx; a.N = dimx * dimy;
    a.dpf = d_dpf; a.dx = d_dpf_dx; a.dy = d_dpf_dy; a.ruv = d_ruv; a.nn = nn; a.mvn = d_mvn; a.Kmax = Kmax;
    a.nclus = d_nclus; a.xyuvav = d_xyuvav;
    a.factor = (float)(1.0 / 365.0 * dt / mpp);                     // MIMC_module.c:1391
    float tw = 0.5; tw -= 0.02;                                     // :1386, :1395 (f32 variable, f64 constant)
    a.thres_weight = tw;
    mimc3::dpf1_carve(a, d_work);
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipError_t e = mimc3::launch_dpf1_init(a, s);
    if (e != hipSuccess) return mimc3::hip_fail(e, "dpf1 init launch");
    // the reference's sweep count is data dependent and unbounded: enqueue batches, poll the device's done flag
    int32_t st[mimc3::kD1Words] = {0};
    // The reference's loop has no bound: a point whose fitted value is NaN counts as "processed" in every sweep and
    // keeps its inner while alive for ever (:1526, :1533-1545).  This library gives up instead of hanging the device.
    const int kMaxBatches = 1 << 15;                      // 2^20 sweeps
    int batches = 0;
    for (;;) {
        e = mimc3::launch_dpf1_sweeps(a, 32, s);
        if (e != hipSuccess) return mimc3::hip_fail(e, "dpf1 sweep launch");
        HIP_TRY(hipMemcpyAsync(st, a.state, sizeof(st), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        if (st[mimc3::kD1Done]) break;
        if (++batches >= kMaxBatches)
            return mimc3::fail(MIMC3_ESTATE, "mimc3_get_dpf1: no termination after 2^20 sweeps (the reference would loop for ever on this input)");
    }
    e = mimc3::launch_dpf1_finish(a, s);
    if (e != hipSuccess) return mimc3::hip_fail(e, "dpf1 finish launch");
    if (sweeps_done) *sweeps_done = st[mimc3::kD1Sweeps];
    return 0;
}

extern "C" int mimc3_get_dpf1(mimc3_ctx *c, int32_t dimy, int32_t dimx, int32_t *dpf, float *dpf_dx, float *dpf_dy,
                              const int32_t *ruv, int32_t nn, const float *mvn, int32_t Kmax, const int32_t *nclus,
                              const double *xyuvav, float dt, float mpp, int32_t *sweeps_done)
{
    if (!c || !dpf || !dpf_dx || !dpf_dy || !ruv || !mvn || !nclus || !xyuvav || dimx <= 0 || dimy <= 0 || nn <= 0 || Kmax <= 0)
        return mimc3::fail(MIMC3_EINVAL, "mimc3_get_dpf1: bad argument");
    const size_t N = (size_t)dimx * dimy;
    for (size_t i = 0; i < N; ++i)
        if (nclus[i] < 0 || nclus[i] > Kmax || dpf[i] >= nclus[i])
            return mimc3::fail(MIMC3_EINVAL, "mimc3_get_dpf1: cluster count/id out of range");
    HIP_TRY(hipSetDevice(c->device));
    const size_t o_dpf = 0, o_dx = o_dpf + al256(4 * N), o_dy = o_dx + al256(4 * N), o_ruv = o_dy + al256(4 * N),
                 o_mvn = o_ruv + al256(8 * (size_t)nn), o_ncl = o_mvn + al256(20 * N * Kmax), o_xy = o_ncl + al256(4 * N),
                 total = o_xy + al256(48 * N);
    HIP_TRY(c->n1_io.reserve(total));
    HIP_TRY(c->n1_work.reserve((size_t)mimc3::dpf1_workspace_bytes((int32_t)N)));
    char *b = static_cast<char *>(c->n1_io.p);
    hipStream_t s = c->stream;
    HIP_TRY(hipMemcpyAsync(b + o_dpf, dpf, 4 * N, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(b + o_ruv, ruv, 8 * (size_t)nn, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(b + o_mvn, mvn, 20 * N * Kmax, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(b + o_ncl, nclus, 4 * N, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(b + o_xy, xyuvav, 48 * N, hipMemcpyHostToDevice, s));
    int rc = mimc3_get_dpf1_dev(c, dimy, dimx, reinterpret_cast<int32_t *>(b + o_dpf), reinterpret_cast<float *>(b + o_dx),
                                reinterpret_cast<float *>(b + o_dy), reinterpret_cast<const int32_t *>(b + o_ruv), nn,
                                reinterpret_cast<const float *>(b + o_mvn), Kmax, reinterpret_cast<const int32_t *>(b + o_ncl),
                                reinterpret_cast<const double *>(b + o_xy), dt, mpp, c->n1_work.p, sweeps_done, s);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(dpf, b + o_dpf, 4 * N, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(dpf_dx, b + o_dx, 4 * N, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(dpf_dy, b + o_dy, 4 * N, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return 0;
}

// ---------------------------------------------------------------------------------------------
// N2: image pre-filter
// ---------------------------------------------------------------------------------------------
static int conv2_args(mimc3::Conv2Args &a, const float *d_in, int32_t H, int32_t W, const float *kernel, int32_t kh, int32_t kw,
                      float *d_out, uint32_t *d_min)
{
    if (!d_in || !d_out || !kernel || !d_min || H <= 0 || W <= 0 || kh <= 0 || kw <= 0 || kh * kw > mimc3::kConvMaxTaps ||
        kh > H || kw > W)
        return mimc3::fail(MIMC3_EINVAL, "conv2: bad argument (kernel at most 81 taps, no larger than the image)");
    a.in = d_in; a.out = d_out; a.H = H; a.W = W; a.kh = kh; a.kw = kw; a.minkey = d_min;
    std::memcpy(a.k, kernel, sizeof(float) * (size_t)kh * kw);
    return 0;
}

extern "C" int mimc3_float_conv2_dev(mimc3_ctx *c, const float *d_in, int32_t H, int32_t W, const float *kernel, int32_t kh,
                                     int32_t kw, float *d_out, void *d_scratch, void *stream)
{
    if (!c) return mimc3::fail(MIMC3_EINVAL, "mimc3_float_conv2_dev: ctx is NULL");
    mimc3::Conv2Args a{};
    int rc = conv2_args(a, d_in, H, W, kernel, kh, kw, d_out, static_cast<uint32_t *>(d_scratch));
    if (rc) return rc;
    HIP_TRY(hipSetDevice(c->device));
    hipError_t e = mimc3::launch_conv2(a, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return mimc3::hip_fail(e, "conv2 kernel launch");
    return 0;
}

extern "C" int mimc3_float_conv2(mimc3_ctx *c, const float *in, int32_t H, int32_t W, const float *kernel, int32_t kh, int32_t kw,
                                 float *out)
{
    if (!c || !in || !out || !kernel || H <= 0 || W <= 0) return mimc3::fail(MIMC3_EINVAL, "mimc3_float_conv2: bad argument");
    HIP_TRY(hipSetDevice(c->device));
    const size_t bytes = sizeof(float) * (size_t)H * W;
    HIP_TRY(c->conv_io.reserve(2 * al256(bytes) + 256));
    char *b = static_cast<char *>(c->conv_io.p);
    hipStream_t s = c->stream;
    HIP_TRY(hipMemcpyAsync(b, in, bytes, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(b + al256(bytes), out, bytes, hipMemcpyHostToDevice, s));     // `out` is in/out: its border is read
    int rc = mimc3_float_conv2_dev(c, reinterpret_cast<const float *>(b), H, W, kernel, kh, kw, reinterpret_cast<float *>(b + al256(bytes)),
                                   b + 2 * al256(bytes), s);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(out, b + al256(bytes), bytes, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return 0;
}

extern "C" int mimc3_ctx_filter_images(mimc3_ctx *c, const float *kernel, int32_t kh, int32_t kw)
{
    if (!c) return mimc3::fail(MIMC3_EINVAL, "mimc3_ctx_filter_images: ctx is NULL");
    if (!c->raw_i0 || !c->raw_i1) return mimc3::fail(MIMC3_ESTATE, "mimc3_ctx_filter_images: images not set");
    HIP_TRY(hipSetDevice(c->device));
    if (!kernel) {                                   // back to the pair as handed over; the next filter starts from fresh planes
        c->d_i0 = c->raw_i0; c->d_i1 = c->raw_i1;
        c->filt_live = false;
        return prepare_u8(c);
    }
    const size_t bytes = sizeof(float) * (size_t)c->H * c->W;
    HIP_TRY(c->filt0.reserve(bytes));
    HIP_TRY(c->filt1.reserve(bytes));
    HIP_TRY(c->conv_io.reserve(256));
    hipStream_t s = c->stream;
    // The reference allocates its two output planes ONCE (MIMC_main.c:302-303: fresh memory, zeros -- T4) and runs all
    // three filters into them (:306-307).  GMA_float_conv2 never writes the border of `out` but reads it (minimum,
    // right-hand columns of the shift), so what one filter leaves in the border rows/columns is input to the next:
    // the planes are cleared only for the first filter after the pair was set.
    if (!c->filt_live) {
        HIP_TRY(hipMemsetAsync(c->filt0.p, 0, bytes, s));
        HIP_TRY(hipMemsetAsync(c->filt1.p, 0, bytes, s));
        c->filt_live = true;
    }
    int rc = mimc3_float_conv2_dev(c, c->raw_i0, c->H, c->W, kernel, kh, kw, static_cast<float *>(c->filt0.p), c->conv_io.p, s);
    if (rc) return rc;
    rc = mimc3_float_conv2_dev(c, c->raw_i1, c->H, c->W, kernel, kh, kw, static_cast<float *>(c->filt1.p), c->conv_io.p, s);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(s));
    c->d_i0 = static_cast<const float *>(c->filt0.p);
    c->d_i1 = static_cast<const float *>(c->filt1.p);
    return prepare_u8(c);
}

extern "C" int mimc3_ctx_get_images(mimc3_ctx *c, float *i0, float *i1)
{
    if (!c || (!i0 && !i1)) return mimc3::fail(MIMC3_EINVAL, "mimc3_ctx_get_images: bad argument");
    if (!c->d_i0 || !c->d_i1) return mimc3::fail(MIMC3_ESTATE, "mimc3_ctx_get_images: images not set");
    HIP_TRY(hipSetDevice(c->device));
    const size_t bytes = sizeof(float) * (size_t)c->H * c->W;
    if (i0) HIP_TRY(hipMemcpyAsync(i0, c->d_i0, bytes, hipMemcpyDeviceToHost, c->stream));
    if (i1) HIP_TRY(hipMemcpyAsync(i1, c->d_i1, bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}

// ---------------------------------------------------------------------------------------------
// N4: control-point offset (get_offset_image, MIMC_module.c:33-492)
// Host side: candidate bookkeeping, the reference's row shuffle, segment loop, the sequential border recurrence
// of the reused filter plane, vote accumulation (f32, candidate order).  Device side: everything that touches
// pixels (validity counts, chips, chip-local filters, 16 matches per candidate, clustering).
// ---------------------------------------------------------------------------------------------
namespace {
struct Arena {
    char *base; size_t used = 0, cap;
    Arena(void *p, size_t c) : base(static_cast<char *>(p)), cap(c) {}
    template <class T> T *take(size_t n) { T *r = reinterpret_cast<T *>(base + used); used += al256(sizeof(T) * n); return r; }
};
}  // namespace

namespace {
// MIMC3_CP_TIMING=1: wall time of the steps of the control-point stage on stderr
struct CpClock {
    bool on = getenv("MIMC3_CP_TIMING") != nullptr;
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    void mark(const char *what, int a = -1, int b = -1)
    {
        if (!on) return;
        const auto n = std::chrono::steady_clock::now();
        fprintf(stderr, "[mimc3 cp] %-22s %3d %3d %8.3f ms\n", what, a, b, std::chrono::duration<double, std::milli>(n - t).count());
        t = n;
    }
};
}  // namespace

namespace {
struct CpGeom { int ocw2, ocw_chip, cs, ts, awc, npiv; };

// Device work of ONE SLICE of a segment of control-point candidates on context `c`: the chip atlases of the four image variants
// for the slice's n candidates (tile t = candidate t; the filtered ones shifted by the minima `mn` the caller settled for the
// WHOLE segment, T8), the 16 matches, the clusters.  mvn [n][16][5] / ncl [n] come back in host memory.  Candidates are as
// independent as grid points (MIMC_module.c:325-378): a multi-GPU driver gives every rank a slice (mimc3_get_offset_image_multi).
static int cp_slice(mimc3_ctx *c, const mimc3_cp_params *p, const CpGeom &g, const int32_t *uv, const float *const mn0[3], const float *const mn1[3],
                    int32_t n, float *const pre_t0[3], float *const pre_t1[3], float *mvn, int32_t *ncl, CpClock *clk, int sg)
{
    if (n <= 0) return 0;
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t s = c->stream;
    const int H = c->H, W = c->W, cs = g.cs, ts = g.ts, ocw_chip = g.ocw_chip, npiv = g.npiv;
    for (auto &st : c->side) if (!st) HIP_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    for (auto &ev : c->ev_side) if (!ev) HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    for (auto &ch : c->cp_child) if (!ch) RC_TRY(ctx_create_impl(c->device, &ch, true));
    const size_t nm = (size_t)n;
    // four image variants (raw, three pre-filters) x an atlas pair each, three pairs of stencil scratch
    const size_t need = al256(8 * nm) + al256(48 * nm) + al256(8 * npiv * nm) + al256(8 * (nm + 1)) + 8 * al256(4 * nm * cs * cs) +
                        (pre_t0 ? 0 : 6) * al256(4 * nm * ts * ts) + 12 * al256(4 * nm) + al256(4 * 48 * nm) + al256(4 * 80 * nm) + al256(4 * nm) + 256;
    HIP_TRY(c->cp_buf.reserve(need));
    Arena ar(c->cp_buf.p, c->cp_buf.cap);
    int32_t *d_uv = ar.take<int32_t>(2 * nm);
    double *d_xy = ar.take<double>(6 * nm);
    int32_t *d_piv = ar.take<int32_t>((size_t)2 * npiv * nm);
    int64_t *d_poff = ar.take<int64_t>(nm + 1);
    float *d_a0[4], *d_a1[4], *d_t0[3], *d_t1[3], *d_imin0[3], *d_imin1[3], *d_mn0[3], *d_mn1[3];
    for (int v = 0; v < 4; v++) { d_a0[v] = ar.take<float>(nm * cs * cs); d_a1[v] = ar.take<float>(nm * cs * cs); }
    for (int k = 0; k < 3; k++) {
        // (the slice that starts at the segment's first candidate finds its filtered planes made: the minima pass left them)
        d_t0[k] = pre_t0 ? pre_t0[k] : ar.take<float>(nm * ts * ts); d_t1[k] = pre_t1 ? pre_t1[k] : ar.take<float>(nm * ts * ts);
        d_imin0[k] = ar.take<float>(nm); d_imin1[k] = ar.take<float>(nm); d_mn0[k] = ar.take<float>(nm); d_mn1[k] = ar.take<float>(nm);
    }
    float *d_dp = ar.take<float>(48 * nm);
    float *d_mvn = ar.take<float>(80 * nm);
    int32_t *d_ncl = ar.take<int32_t>(nm);
    int32_t *d_kmax = ar.take<int32_t>(1);
    // common rectangular pivot set (:150-162), replicated per point for the CSR interface (filled on the device)
    HIP_TRY(mimc3::launch_cp_fill_problem(d_xy, d_piv, d_poff, n, g.awc, ocw_chip, cs, s));
    HIP_TRY(hipMemcpyAsync(d_uv, uv, 8 * nm, hipMemcpyHostToDevice, s));
    // ---- the chip atlases of the four image variants: raw chips, then the three pre-filters shifted by the settled minima
    HIP_TRY(mimc3::launch_cp_extract(c->raw_i0, H, W, d_uv, n, ocw_chip, d_a0[0], s));
    HIP_TRY(mimc3::launch_cp_extract(c->raw_i1, H, W, d_uv, n, ocw_chip, d_a1[0], s));
    for (int kk = 0; kk < 3; kk++) {
        const int kh = p->kdim[kk][0], kw = p->kdim[kk][1];
        if (!pre_t0) {
            HIP_TRY(mimc3::launch_cp_conv_min(c->raw_i0, H, W, d_uv, n, ocw_chip, p->kernel[kk], kh, kw, d_t0[kk], d_imin0[kk], s));
            HIP_TRY(mimc3::launch_cp_conv_min(c->raw_i1, H, W, d_uv, n, ocw_chip, p->kernel[kk], kh, kw, d_t1[kk], d_imin1[kk], s));
        }
        HIP_TRY(hipMemcpyAsync(d_mn0[kk], mn0[kk], 4 * nm, hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync(d_mn1[kk], mn1[kk], 4 * nm, hipMemcpyHostToDevice, s));
        HIP_TRY(mimc3::launch_cp_shift_copy(d_t0[kk], d_mn0[kk], n, ocw_chip, d_a0[kk + 1], s));
        HIP_TRY(mimc3::launch_cp_shift_copy(d_t1[kk], d_mn1[kk], n, ocw_chip, d_a1[kk + 1], s));
    }
    HIP_TRY(hipStreamSynchronize(s));                                          // the atlases are complete
    if (clk) clk->mark("atlases", sg);
    // ---- every atlas is an image pair of its own, handed to a child context that classifies it (8-bit / scaled integers /
    //      floats) and runs the same tiled kernels as the DLC passes: full-square search area (win_half), the 21x21 pivot
    //      set as a replicated CSR
    //      The 16 matches (:330-384): one match is a few hundred workgroups, latency-bound on its own -- the four of a
    //      variant (2 chip sizes x forward/swapped) go to four streams, each with its own overflow lists, and the variants
    //      follow each other on those streams without a host round trip; variant v+1 is classified while v's matches run
    auto matches = [&]() -> int {
        for (int v = 0; v < 4; v++) {
            mimc3_ctx *ch = c->cp_child[v];
            ch->path_mode = c->path_mode;
            // gradients go straight to the u16 kernel: over a whole 85x85 search area their range rarely fits the 8 bits
            // of the per-point-offset form, and a second launch for the points that do not costs a full kernel latency
            ch->no_u8o = true;
            RC_TRY(set_images_dev_impl(ch, d_a0[v], d_a1[v], n * cs, cs, false));
            ch->win_half = ocw_chip;
            if (!ch->u8_ok && !ch->u16_ok) {       // float / 16-bit atlas: the f32 planes are built lazily by the first match
                RC_TRY(build_f32_planes(ch, ch->stream));   // -- here the four matches start on four streams, so build them first
                HIP_TRY(hipStreamSynchronize(ch->stream));
            }
            for (int c3 = 1; c3 < 3; c3++) {
                const int ocw = p->vec_ocw[c3];
                const int32_t slot = (c3 - 1) * 8 + v * 2;
                const int reach = ocw_chip - ocw - 2;
                for (int sw = 0; sw < 2; sw++) {
                    const int lane_id = (c3 - 1) * 2 + sw;                     // 0..3
                    hipStream_t ms = lane_id == 0 ? s : c->side[lane_id - 1];
                    float *o = d_dp + (size_t)(slot + sw) * n * 3;
                    ch->lane = lane_id;
                    int rc = mimc3_match_ncc_dlc_dev(ch, d_xy, n, 0, 0, d_piv, d_poff, npiv, reach, reach, ocw, sw, o, ms);
                    ch->lane = 0;
                    if (rc) return rc;
                    if (sw) HIP_TRY(mimc3::launch_negate_uv(o, n, ms));         // :376-377
                }
            }
        }
        return 0;
    };
    const int mrc = matches();
    for (auto &ch : c->cp_child) ch->win_half = 0;
    for (auto &st : c->side) HIP_TRY(hipStreamSynchronize(st));
    HIP_TRY(hipStreamSynchronize(s));
    if (mrc) return mrc;
    if (clk) clk->mark("classify + matches", sg);
    // ---- clusters of the 16 matches (:392-413)
    mimc3::ClusterArgs ca{};
    ca.dp = d_dp; ca.ndp = 16; ca.N = n; ca.Kmax = 16; ca.mvn = d_mvn; ca.nclus = d_ncl; ca.kmax_seen = d_kmax;
    HIP_TRY(mimc3::launch_cluster(ca, s));
    HIP_TRY(hipMemcpyAsync(mvn, d_mvn, 4 * 80 * nm, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(ncl, d_ncl, 4 * nm, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return 0;
}
}  // namespace

// The control-point offset on one context, or -- nctx > 1 -- with the candidates of every segment cut into slices that the
// contexts (one per device, all holding the same image pair) match side by side, each on a host thread of its own.  Everything
// that is sequential in the reference stays on the calling thread: the rand() shuffle, the segment loop with its early exit, the
// chip-to-chip recurrence of the filtered planes' minima (T8), the f32 vote sums in candidate order -- the result does not depend
// on how many contexts share the work.
extern "C" int mimc3_get_offset_image_multi(mimc3_ctx *const *ctxs, int32_t nctx, const double *xyuvav, int32_t N, const mimc3_cp_params *p,
                                            int32_t offset[2], uint8_t *flag_cp, int32_t *status, int32_t *info, float *sduv_out)
{
    CpClock clk;
    if (!ctxs || nctx < 1 || nctx > 64 || !xyuvav || !p || !offset || !flag_cp || !status || N <= 0)
        return mimc3::fail(MIMC3_EINVAL, "mimc3_get_offset_image: bad argument");
    mimc3_ctx *c = ctxs[0];
    for (int32_t r = 0; r < nctx; r++) {
        if (!ctxs[r]) return mimc3::fail(MIMC3_EINVAL, "mimc3_get_offset_image: a context is NULL");
        if (!ctxs[r]->raw_i0 || !ctxs[r]->raw_i1) return mimc3::fail(MIMC3_ESTATE, "mimc3_get_offset_image: images not set");
        if (ctxs[r]->H != c->H || ctxs[r]->W != c->W) return mimc3::fail(MIMC3_ESTATE, "mimc3_get_offset_image: the contexts hold different image pairs");
        // (a context listed twice would have two host threads slicing on the same scratch, children and streams)
        for (int32_t q = 0; q < r; q++)
            if (ctxs[q] == ctxs[r]) return mimc3::fail(MIMC3_EINVAL, "mimc3_get_offset_image: a context is listed twice");
    }
    for (int k = 0; k < 3; k++)
        if (!p->kernel[k] || p->kdim[k][0] < 1 || p->kdim[k][0] > 3 || p->kdim[k][1] < 1 || p->kdim[k][1] > 3)
            return mimc3::fail(MIMC3_EINVAL, "mimc3_get_offset_image: three pre-filter kernels of at most 3x3 are required");
    CpGeom g{};
    g.ocw2 = p->vec_ocw[2];
    g.ocw_chip = (int32_t)(p->vec_ocw[2] + p->aw_cre + 2);                        // :51
    g.cs = 2 * g.ocw_chip + 1; g.ts = g.cs + 2;
    g.awc = (int)p->aw_cre;
    const int ocw2 = g.ocw2, ocw_chip = g.ocw_chip, awc = g.awc;
    if (ocw2 < 1 || p->vec_ocw[1] < 1 || p->vec_ocw[1] > ocw2 || awc < 0 || ocw_chip - ocw2 - 2 < 0)
        return mimc3::fail(MIMC3_EINVAL, "mimc3_get_offset_image: need 1 <= vec_ocw[1] <= vec_ocw[2] and AW_CRE >= 0");
    *status = -1;
    if (info) info[0] = info[1] = info[2] = info[3] = 0;
    if (sduv_out) sduv_out[0] = sduv_out[1] = 0.0f;
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t s = c->stream;
    const int H = c->H, W = c->W;

    // ---- candidates: slow a-priori points (:64-78) whose ocw[2] chip of i0 is mostly valid (:81-112)
    int32_t num_cp;
    if (N * p->ratio_cp > p->num_cp_max) num_cp = p->num_cp_max; else num_cp = (int32_t)(N * p->ratio_cp);
    std::vector<int32_t> rows;
    for (int32_t gi = 0; gi < N; gi++) {
        const float spd = xyuvav[6 * (size_t)gi + 4] * xyuvav[6 * (size_t)gi + 4] + xyuvav[6 * (size_t)gi + 5] * xyuvav[6 * (size_t)gi + 5];
        if (spd < p->thres_spd_cp * p->thres_spd_cp) rows.push_back(gi);
    }
    std::vector<int32_t> uv(2 * rows.size());
    for (size_t i = 0; i < rows.size(); i++) {
        const int u = (int32_t)xyuvav[6 * (size_t)rows[i] + 2], v = (int32_t)xyuvav[6 * (size_t)rows[i] + 3];
        if (u - ocw_chip - 1 < 0 || u + ocw_chip + 1 >= W || v - ocw_chip - 1 < 0 || v + ocw_chip + 1 >= H)
            return mimc3::fail(MIMC3_EBOUNDS, "mimc3_get_offset_image: grid point " + std::to_string(rows[i]) +
                                                  " control-point chip leaves the image (the reference reads out of bounds there)");
        uv[2 * i] = u; uv[2 * i + 1] = v;
    }
    if (!rows.empty()) {
        const size_t n0 = rows.size();
        HIP_TRY(c->cp_buf.reserve(al256(8 * n0) + al256(4 * n0)));
        Arena a0(c->cp_buf.p, c->cp_buf.cap);
        int32_t *d_uv = a0.take<int32_t>(2 * n0), *d_cnt = a0.take<int32_t>(n0);
        HIP_TRY(hipMemcpyAsync(d_uv, uv.data(), 8 * n0, hipMemcpyHostToDevice, s));
        HIP_TRY(mimc3::launch_cp_count_invalid(c->raw_i0, H, W, d_uv, (int32_t)n0, ocw2, d_cnt, s));
        std::vector<int32_t> cnt(n0);
        HIP_TRY(hipMemcpyAsync(cnt.data(), d_cnt, 4 * n0, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        const int32_t thres_numpx = (ocw2 * 2 + 1) * (ocw2 * 2 + 1) / 2;
        size_t w = 0;
        for (size_t i = 0; i < n0; i++)
            if (!(cnt[i] > thres_numpx)) { rows[w] = rows[i]; uv[2 * w] = uv[2 * i]; uv[2 * w + 1] = uv[2 * i + 1]; w++; }   // :106 (i0's count only)
        rows.resize(w); uv.resize(2 * w);
    }
    const int32_t ncand = (int32_t)rows.size();
    clk.mark("candidates");
    if (info) { info[0] = ncand; info[1] = num_cp; }
    if (ncand < p->num_cp_min) return 0;                                           // :119-123 (status -1)
    if (num_cp > ncand) num_cp = (int32_t)((float)ncand * 0.75);                   // :125-129
    if (info) info[1] = num_cp;

    // ---- GMA_double_randperm_row (:494-541) on the candidate ids
    std::vector<int32_t> order(ncand), tmp(ncand);
    for (int32_t i = 0; i < ncand; i++) tmp[i] = i;
    srand(p->seed < 0 ? (unsigned)time(nullptr) : (unsigned)p->seed);
    for (int32_t lim = ncand - 1; lim >= 0; lim--) {
        const int32_t idx = lim != 0 ? (int32_t)(rand() % lim) : 0;
        order[lim] = tmp[idx]; tmp[idx] = tmp[0]; tmp[0] = tmp[lim];
    }

    const int32_t nseg = ncand < p->num_cp_min ? 1 : ncand / num_cp;                // :171
    std::vector<int32_t> seg(nseg + 1);
    seg[0] = 0;
    int32_t nmax = 0;
    for (int32_t k = 1; k <= nseg; k++) {
        seg[k] = (int32_t)(ncand * ((float)k / (float)nseg));                      // :179
        nmax = std::max(nmax, seg[k] - seg[k - 1]);
    }
    g.npiv = (int32_t)((p->aw_cre * 2 + 1) * (p->aw_cre * 2 + 1));
    if (g.npiv != (2 * awc + 1) * (2 * awc + 1)) return mimc3::fail(MIMC3_EINVAL, "mimc3_get_offset_image: AW_CRE must be integral");

    // per-chip minima of the filtered chips of a whole segment (context 0): scratch for the stencil planes + the minima
    const size_t nm = (size_t)nmax, ts = (size_t)g.ts;
    clk.mark("setup + uploads");
    float sduv[2] = {0.0f, 0.0f};
    int32_t ncur = 0, segs = 0;
    bool ok = false;
    std::vector<int32_t> uv_seg;
    std::vector<float> imin0[3], imin1[3], mn0[3], mn1[3], mvn((size_t)80 * nmax);
    for (int k = 0; k < 3; k++) { imin0[k].resize(nmax); imin1[k].resize(nmax); mn0[k].resize(nmax); mn1[k].resize(nmax); }
    std::vector<int32_t> ncl(nmax);
    for (int32_t sg = 0; sg < nseg; sg++) {
        const int32_t beg = seg[sg], n = seg[sg + 1] - seg[sg];
        segs++;
        if (n <= 0) continue;
        uv_seg.resize(2 * (size_t)n);
        for (int32_t t = 0; t < n; t++) { uv_seg[2 * t] = uv[2 * order[beg + t]]; uv_seg[2 * t + 1] = uv[2 * order[beg + t] + 1]; }
        float *pre_t0[3], *pre_t1[3];
        {   // the filters' per-chip minima of the WHOLE segment, then the reference's plane-reuse rule: a serial scan on the host
            HIP_TRY(hipSetDevice(c->device));
            HIP_TRY(c->cp_pre.reserve(al256(8 * nm) + 6 * al256(4 * nm * ts * ts) + 6 * al256(4 * nm) + 256));
            Arena ar(c->cp_pre.p, c->cp_pre.cap);
            int32_t *d_uv = ar.take<int32_t>(2 * nm);
            HIP_TRY(hipMemcpyAsync(d_uv, uv_seg.data(), 8 * (size_t)n, hipMemcpyHostToDevice, s));
            for (int kk = 0; kk < 3; kk++) {
                const int kh = p->kdim[kk][0], kw = p->kdim[kk][1];
                pre_t0[kk] = ar.take<float>(nm * ts * ts); pre_t1[kk] = ar.take<float>(nm * ts * ts);
                float *d_m0 = ar.take<float>(nm), *d_m1 = ar.take<float>(nm);
                HIP_TRY(mimc3::launch_cp_conv_min(c->raw_i0, H, W, d_uv, n, ocw_chip, p->kernel[kk], kh, kw, pre_t0[kk], d_m0, s));
                HIP_TRY(mimc3::launch_cp_conv_min(c->raw_i1, H, W, d_uv, n, ocw_chip, p->kernel[kk], kh, kw, pre_t1[kk], d_m1, s));
                HIP_TRY(hipMemcpyAsync(imin0[kk].data(), d_m0, 4 * (size_t)n, hipMemcpyDeviceToHost, s));
                HIP_TRY(hipMemcpyAsync(imin1[kk].data(), d_m1, 4 * (size_t)n, hipMemcpyDeviceToHost, s));
            }
            HIP_TRY(hipStreamSynchronize(s));
        }
        for (int kk = 0; kk < 3; kk++) {
            // the reference's output plane is reused from point to point (:259-262): its never-written border cells
            // stay 0 (T4), its right-hand border columns accumulate the shifts (:2568-2582); both enter the minimum
            const int kh = p->kdim[kk][0], kw = p->kdim[kk][1];
            const int ox = kw / 2, oy = kh / 2;
            const bool has_zero = ox > 0 || oy > 0;
            for (int im = 0; im < 2; im++) {
                const float *imn = im ? imin1[kk].data() : imin0[kk].data();
                float *mn = im ? mn1[kk].data() : mn0[kk].data();
                float b = 0.0f;
                for (int32_t t = 0; t < n; t++) {
                    float m = 1e+37f;
                    if (imn[t] < m) m = imn[t];
                    if (has_zero && 0.0f < m) m = 0.0f;
                    if (ox > 0 && b < m) m = b;
                    mn[t] = m;
                    if (ox > 0) b = (b != b) ? 0.0f : b - (m - 1.0f);
                }
            }
        }
        clk.mark("minima", sg);
        // ---- the slices: contexts 1.. on host threads of their own, context 0 here
        {
            const int32_t R = nctx < n ? nctx : n;
            std::vector<int> rcs((size_t)R, 0);
            std::vector<std::string> errs((size_t)R);
            auto run = [&](int32_t r) {
                const int32_t t0 = (int32_t)((int64_t)n * r / R), t1 = (int32_t)((int64_t)n * (r + 1) / R);
                const float *a0[3] = {mn0[0].data() + t0, mn0[1].data() + t0, mn0[2].data() + t0};
                const float *a1[3] = {mn1[0].data() + t0, mn1[1].data() + t0, mn1[2].data() + t0};
                rcs[(size_t)r] = cp_slice(ctxs[r], p, g, uv_seg.data() + 2 * (size_t)t0, a0, a1, t1 - t0, r == 0 ? pre_t0 : nullptr, r == 0 ? pre_t1 : nullptr, mvn.data() + 80 * (size_t)t0, ncl.data() + t0,
                                          r == 0 ? &clk : nullptr, sg);
                if (rcs[(size_t)r]) errs[(size_t)r] = mimc3_last_error();        // (the message is thread-local)
            };
            std::vector<std::thread> th;
            for (int32_t r = 1; r < R; r++) th.emplace_back(run, r);
            run(0);
            for (auto &t : th) t.join();
            for (int32_t r = 0; r < R; r++)
                if (rcs[(size_t)r]) return mimc3::fail(rcs[(size_t)r], errs[(size_t)r]);
            HIP_TRY(hipSetDevice(c->device));
        }
        // ---- clusters holding >= 60 % vote with their mean, in candidate order (:392-413)
        for (int32_t t = 0; t < n; t++)
            for (int32_t k = 0; k < ncl[t]; k++)
                if (mvn[((size_t)t * 16 + k) * 5 + 4] >= 0.6) {
                    sduv[0] += mvn[((size_t)t * 16 + k) * 5];
                    sduv[1] += mvn[((size_t)t * 16 + k) * 5 + 1];
                    flag_cp[rows[order[beg + t]]] = 1;
                    ncur++;
                }
        clk.mark("clusters + votes", sg, n);
        if (num_cp <= ncur) { ok = true; break; }                                  // :424-430
    }
    if (info) { info[2] = segs; info[3] = ncur; }
    if (sduv_out) { sduv_out[0] = sduv[0]; sduv_out[1] = sduv[1]; }
    if (ncur < num_cp && ncur >= p->num_cp_min) ok = true;                          // :451-455
    if (!ok) return 0;                                                              // status -1 (:478-484)
    const float du = sduv[0] / (float)ncur, dv = sduv[1] / (float)ncur;             // :460-476
    offset[0] = du > 0 ? (int32_t)(du + 0.5) : (int32_t)(du - 0.5);
    offset[1] = dv > 0 ? (int32_t)(dv + 0.5) : (int32_t)(dv - 0.5);
    *status = 1;
    return 0;
}

extern "C" int mimc3_get_offset_image(mimc3_ctx *c, const double *xyuvav, int32_t N, const mimc3_cp_params *p, int32_t offset[2],
                                      uint8_t *flag_cp, int32_t *status, int32_t *info, float *sduv_out)
{
    if (!c) return mimc3::fail(MIMC3_EINVAL, "mimc3_get_offset_image: bad argument");
    mimc3_ctx *one[1] = {c};
    return mimc3_get_offset_image_multi(one, 1, xyuvav, N, p, offset, flag_cp, status, info, sduv_out);
}

// ---------------------------------------------------------------------------------------------
// small device helpers used by the whole-program driver (pipeline.cpp)
// ---------------------------------------------------------------------------------------------
extern "C" void *mimc3_ctx_stream(mimc3_ctx *c) { return c ? static_cast<void *>(c->stream) : nullptr; }

extern "C" int mimc3_negate_uv_dev(mimc3_ctx *c, float *d_out, int32_t N, void *stream)
{
    if (!c || !d_out || N <= 0) return mimc3::fail(MIMC3_EINVAL, "mimc3_negate_uv_dev: bad argument");
    HIP_TRY(hipSetDevice(c->device));
    hipError_t e = mimc3::launch_negate_uv(d_out, N, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return mimc3::hip_fail(e, "negate kernel launch");
    return 0;
}

extern "C" int mimc3_negate_pivots_dev(mimc3_ctx *c, const int32_t *d_piv_uv, int32_t *d_out, int64_t count, void *stream)
{
    if (!c || !d_piv_uv || !d_out || count <= 0) return mimc3::fail(MIMC3_EINVAL, "mimc3_negate_pivots_dev: bad argument");
    HIP_TRY(hipSetDevice(c->device));
    hipError_t e = mimc3::launch_negate_i32(d_piv_uv, d_out, 2 * count, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return mimc3::hip_fail(e, "negate kernel launch");
    return 0;
}

extern "C" int mimc3_dpf_to_vxyexyqual_dev(mimc3_ctx *c, const int32_t *d_dpf, const float *d_mvn, int32_t N, int32_t Kmax,
                                           float *d_out5, void *stream)
{
    if (!c || !d_dpf || !d_mvn || !d_out5 || N <= 0 || Kmax <= 0) return mimc3::fail(MIMC3_EINVAL, "mimc3_dpf_to_vxyexyqual_dev: bad argument");
    HIP_TRY(hipSetDevice(c->device));
    hipError_t e = mimc3::launch_gather(d_dpf, d_mvn, N, Kmax, d_out5, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return mimc3::hip_fail(e, "gather kernel launch");
    return 0;
}

extern "C" int mimc3_ctx_workspace(mimc3_ctx *c, int32_t slot, size_t bytes, void **d_ptr)
{
    if (!c || !d_ptr || slot < 0 || slot >= 24) return mimc3::fail(MIMC3_EINVAL, "mimc3_ctx_workspace: bad argument");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(c->slot[slot].reserve(bytes ? bytes : 1));
    *d_ptr = c->slot[slot].p;
    return 0;
}

extern "C" int mimc3_ctx_host_workspace(mimc3_ctx *c, int32_t slot, size_t bytes, void **h_ptr)
{
    if (!c || !h_ptr || slot < 0 || slot >= 8) return mimc3::fail(MIMC3_EINVAL, "mimc3_ctx_host_workspace: bad argument");
    if (bytes > c->hslot_cap[slot]) {
        HIP_TRY(hipSetDevice(c->device));
        if (c->hslot[slot]) { (void)hipHostFree(c->hslot[slot]); c->hslot[slot] = nullptr; c->hslot_cap[slot] = 0; }
        const size_t want = bytes + bytes / 8 + 4096;
        HIP_TRY(hipHostMalloc(&c->hslot[slot], want, hipHostMallocPortable));
        c->hslot_cap[slot] = want;
    }
    *h_ptr = c->hslot[slot];
    return 0;
}

extern "C" int mimc3_ctx_device(mimc3_ctx *c) { return c ? c->device : MIMC3_EINVAL; }

extern "C" int mimc3_ctx_image_size(mimc3_ctx *c, int32_t *H, int32_t *W)
{
    if (!c || !H || !W) return mimc3::fail(MIMC3_EINVAL, "mimc3_ctx_image_size: bad argument");
    if (!c->raw_i0) return mimc3::fail(MIMC3_ESTATE, "mimc3_ctx_image_size: images not set");
    *H = c->H; *W = c->W;
    return 0;
}
