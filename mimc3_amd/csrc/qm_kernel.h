// qm_kernel.h -- launch interface of the QM pseudo-smoothing kernels (internal).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mimc3 {

constexpr int kQmThreads = 128;
constexpr int kQmFinishThreads = 1024;
constexpr int kQmMaskWords = 8;        // mismatch bits of up to 256 stacked masks
constexpr int kQmFlagWords = 16;
enum { kQmAny = 0, kQmDone = 1, kQmSweeps = 2, kQmSkipped = 3, kQmTicket = 4 };
constexpr int kQmLaunchesPerSweep = 2;

struct QmArgs {
    int32_t dimy, dimx, N;
    int32_t *dpf;            // [N] cluster id, -1 none        (in/out)
    float *dx, *dy;          // [N] dpf_dx, dpf_dy             (in/out)
    const int32_t *ruv;      // [nn][2]
    int32_t nn;
    const float *mvn;        // [N][Kmax][5]
    int32_t Kmax;
    const int32_t *nclus;    // [N]
    const double *xyuvav;    // [N][6]
    int32_t max_sweeps;
    // workspace (carved by launch_qm)
    float *bx, *by;
    int32_t *bid;
    int32_t *flags, *diff;
    unsigned char *mask[2];
    unsigned char *stack;    // [max_sweeps+1][N]
};

int64_t qm_workspace_bytes(int32_t n, int32_t max_sweeps);
// Enqueues init + max_sweeps x (sweep, finish_sweep); flags[kQmSweeps] holds the
// reference's NOI at exit once the stream has drained.
hipError_t launch_qm(QmArgs a, void *work, hipStream_t stream);
// word offset of the flags block inside the workspace (for reading sweeps_done back)
static inline int64_t qm_flags_offset_bytes(int32_t n) { return 12 * (((int64_t)n + 255) & ~255LL); }

}  // namespace mimc3
