/* tests/fake_rccl.c -- TEST-ONLY stand-in for the six RCCL entry points csrc/mgpu.cpp resolves with dlsym
 * (ncclCommInitAll, ncclAllGather, ncclGroupStart, ncclGroupEnd, ncclCommDestroy, ncclGetErrorString).
 *
 * Why it exists: the GPU test box has ONE device, and RCCL refuses a communicator that lists a device twice, so the
 * N > 1 branches of the native multi-GPU driver (a host thread per rank, padded blocks with world > 1, the grouped
 * all-gather issued from one thread, the un-permute over several ranks' blocks) could never run there.  With
 * mimc3_mgpu_create_ex(devices, n, <this library>, MIMC3_MGPU_REPEAT_DEVICES, ...) the driver runs N ranks as N contexts (N streams) of
 * the same device, and the "collective" below moves the blocks with device-to-device copies ordered by events:
 *     recv[r][k * count .. (k+1) * count) = send[k][0 .. count)      for every rank r and k
 * which is exactly what ncclAllGather produces.  What it does NOT test is xGMI transport: that needs a real node
 * (run tests/test_mgpu.py there with MIMC3_TEST_DEVICES=0,1,...: the plain mimc3_mgpu_create).
 *
 * Build: gcc -shared -fPIC -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include tests/fake_rccl.c -L/opt/rocm/lib -lamdhip64
 * Not part of the product; nothing under mimc3_amd/ refers to it. */
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define FAKE_MAX_RANKS 64

typedef struct fake_group {
    int world;
    int refs;
    int posted;                              /* all-gather calls seen since the last completed collective */
    const void *send[FAKE_MAX_RANKS];
    void *recv[FAKE_MAX_RANKS];
    size_t bytes[FAKE_MAX_RANKS];
    hipStream_t stream[FAKE_MAX_RANKS];
    int dev[FAKE_MAX_RANKS];
    unsigned long long collectives;          /* completed all-gathers (the test reads it through fake_rccl_collectives) */
} fake_group;

struct ncclComm {
    fake_group *g;
    int rank;
};
typedef struct ncclComm *ncclComm_t;

static int g_depth = 0;                      /* ncclGroupStart nesting (calls come from ONE host thread in mgpu.cpp) */
static fake_group *g_pending[16];
static int g_npending = 0;
static unsigned long long g_total = 0;

static size_t dtype_bytes(int t)
{
    switch (t) {                             /* rccl.h: int8 0, uint8 1, int32 2, uint32 3, int64 4, uint64 5, half 6, float 7, double 8 */
    case 0: case 1: return 1;
    case 2: case 3: case 7: return 4;
    case 4: case 5: case 8: return 8;
    case 6: return 2;
    default: return 0;
    }
}

static int run_collective(fake_group *g)
{
    hipEvent_t ev[FAKE_MAX_RANKS];
    int old = 0;
    (void)hipGetDevice(&old);
    for (int k = 0; k < g->world; k++) {     /* "rank k's block is ready" */
        if (hipSetDevice(g->dev[k]) != hipSuccess) return 1;
        if (hipEventCreateWithFlags(&ev[k], hipEventDisableTiming) != hipSuccess) return 1;
        if (hipEventRecord(ev[k], g->stream[k]) != hipSuccess) return 1;
    }
    for (int r = 0; r < g->world; r++) {
        if (hipSetDevice(g->dev[r]) != hipSuccess) return 1;
        for (int k = 0; k < g->world; k++) {
            if (g->bytes[k] != g->bytes[0]) return 2;                     /* all-gather: equal contributions */
            if (hipStreamWaitEvent(g->stream[r], ev[k], 0) != hipSuccess) return 1;
            if (hipMemcpyAsync((char *)g->recv[r] + (size_t)k * g->bytes[k], g->send[k], g->bytes[k], hipMemcpyDeviceToDevice,
                               g->stream[r]) != hipSuccess) return 1;
        }
    }
    /* a sender's block must not be overwritten before every receiver has copied it: each sender's stream waits for all copies */
    for (int r = 0; r < g->world; r++) {
        hipEvent_t done;
        if (hipSetDevice(g->dev[r]) != hipSuccess) return 1;
        if (hipEventCreateWithFlags(&done, hipEventDisableTiming) != hipSuccess) return 1;
        if (hipEventRecord(done, g->stream[r]) != hipSuccess) return 1;
        for (int k = 0; k < g->world; k++)
            if (k != r && hipStreamWaitEvent(g->stream[k], done, 0) != hipSuccess) return 1;
        (void)hipEventDestroy(done);         /* destruction is deferred until the event has completed */
    }
    for (int k = 0; k < g->world; k++) (void)hipEventDestroy(ev[k]);
    (void)hipSetDevice(old);
    g->posted = 0;
    g->collectives++;
    g_total++;
    return 0;
}

int ncclCommInitAll(ncclComm_t *comms, int ndev, const int *devlist)
{
    if (!comms || ndev <= 0 || ndev > FAKE_MAX_RANKS) return 4;          /* ncclInvalidArgument */
    fake_group *g = (fake_group *)calloc(1, sizeof(fake_group));
    if (!g) return 1;
    g->world = ndev;
    g->refs = ndev;
    for (int k = 0; k < ndev; k++) {
        g->dev[k] = devlist ? devlist[k] : k;
        comms[k] = (ncclComm_t)calloc(1, sizeof(struct ncclComm));
        comms[k]->g = g;
        comms[k]->rank = k;
    }
    return 0;
}

int ncclCommDestroy(ncclComm_t c)
{
    if (!c) return 0;
    if (--c->g->refs == 0) free(c->g);
    free(c);
    return 0;
}

int ncclGroupStart(void) { g_depth++; return 0; }

int ncclAllGather(const void *send, void *recv, size_t count, int dtype, ncclComm_t c, hipStream_t stream)
{
    const size_t es = dtype_bytes(dtype);
    if (!c || !send || !recv || es == 0) return 4;
    fake_group *g = c->g;
    g->send[c->rank] = send; g->recv[c->rank] = recv; g->bytes[c->rank] = count * es; g->stream[c->rank] = stream;
    g->posted++;
    if (g_depth == 0) {                      /* ungrouped call: legal only for a one-rank communicator here (one host thread) */
        if (g->world != 1) return 5;         /* ncclInvalidUsage */
        return run_collective(g);
    }
    int seen = 0;
    for (int i = 0; i < g_npending; i++) seen |= (g_pending[i] == g);
    if (!seen && g_npending < 16) g_pending[g_npending++] = g;
    return 0;
}

int ncclGroupEnd(void)
{
    if (g_depth <= 0) return 5;
    if (--g_depth > 0) return 0;
    int rc = 0;
    for (int i = 0; i < g_npending; i++) {
        fake_group *g = g_pending[i];
        if (g->posted != g->world) { rc = 5; g->posted = 0; continue; }   /* a rank is missing: a real RCCL would hang here */
        const int e = run_collective(g);
        if (e) rc = e;
    }
    g_npending = 0;
    return rc;
}

const char *ncclGetErrorString(int e)
{
    switch (e) {
    case 0: return "no error";
    case 1: return "fake_rccl: HIP call failed";
    case 2: return "fake_rccl: unequal contributions";
    case 4: return "fake_rccl: invalid argument";
    case 5: return "fake_rccl: invalid usage (a rank did not post its all-gather)";
    default: return "fake_rccl: unknown";
    }
}

/* test hook: completed all-gathers since the library was loaded */
unsigned long long fake_rccl_collectives(void) { return g_total; }
