// n1_kernel.h -- launch interface of the candidate-clustering / dpf0 / dpf1 kernels (internal).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mimc3 {

constexpr int kCluPointsPerBlock = 64;   // grid points staged per block (coalesced pass-major reads)
constexpr int kCluThreads = 256;
constexpr int kCluMaxPasses = 64;        // one lane per matcher pass

struct ClusterArgs {
    const float *dp;         // [ndp][N][3] matcher outputs, pass-major (the reference's GMA_float **dp)
    int32_t ndp, N, Kmax;
    float *mvn;              // [N][Kmax][5] out (padding rows zeroed)
    int32_t *nclus;          // [N] out
    int32_t *kmax_seen;      // [1] out: atomicMax of the cluster counts
};
hipError_t launch_cluster(ClusterArgs a, hipStream_t stream);

hipError_t launch_dpf0(const float *mvn, const int32_t *nclus, int32_t N, int32_t Kmax, float min_ratio,
                       int32_t *dpf, hipStream_t stream);

// out5 [5][N] = (mean_u, mean_v, var_u, var_v, fraction) of the chosen cluster, NaN where dpf < 0
hipError_t launch_gather(const int32_t *dpf, const float *mvn, int32_t N, int32_t Kmax, float *out5, hipStream_t stream);

// dpf1 state words (device)
enum { kD1ThresNum = 0, kD1Done = 1, kD1Sweeps = 2, kD1Tally = 4 /* u64: ticket | processed | unprocessed */, kD1Words = 8 };
constexpr int kD1SweepThreads = 256;
constexpr int kD1MaxNeighbours = 4096;   // get_ruv_neighbor's capacity in this library

struct Dpf1Args {
    int32_t dimy, dimx, N;
    int32_t *dpf;            // [N] in: dpf0, out: dpf1
    float *dx, *dy;          // [N] out
    const int32_t *ruv;      // [nn][2]
    int32_t nn;
    const float *mvn;        // [N][Kmax][5]
    int32_t Kmax;
    const int32_t *nclus;    // [N]
    const double *xyuvav;    // [N][6]
    float factor;            // (float)(1.0/365.0*dt/mpp)          (:1391)
    float thres_weight;      // (float)((double)0.5f - 0.02)        (:1386-1395)
    // workspace (carved by the launchers)
    float4 *plane[2];        // [N] (dx, dy, noi, v3), ping-ponged by the sweeps
    float4 *rec;             // [N] (v4, v5, w2, -)
    float *bx, *by;          // [N] smoothing buffers
    int4 *ktab;              // [nn] (du, dv, |d| bits, -)
    int32_t *state;          // [kD1Words]
};
int64_t dpf1_workspace_bytes(int32_t n);
void dpf1_carve(Dpf1Args &a, void *work);
hipError_t launch_dpf1_init(const Dpf1Args &a, hipStream_t stream);
// `count` sweeps; each returns at once when state[kD1Done] is set
hipError_t launch_dpf1_sweeps(const Dpf1Args &a, int count, hipStream_t stream);
// 3x3 smoothing + snap to the nearest cluster (:1623-1706)
hipError_t launch_dpf1_finish(const Dpf1Args &a, hipStream_t stream);

}  // namespace mimc3
