// pipeline_internal.h -- pieces of the whole-program driver (pipeline.cpp) shared with the multi-GPU driver (mgpu.cpp).
// Internal to libmimc3_hip.so; both drivers are built only from the public C ABI plus these helpers.
#pragma once
#include <cstdint>
#include <string>
#include <vector>
#include "../../include/mimc3_hip.h"

namespace mimc3 {

// CSR pivots of one chip size for a set of points, resident on the device (context scratch): made from host corridors by the
// pivot kernels on an auxiliary stream, so that all of it overlaps the control-point stage
struct HostPivots {
    int64_t total = 0;
    int32_t *d_uv = nullptr, *d_uvn = nullptr;       // pivots, negated pivots (:272-279)
    int64_t *d_off = nullptr;
    double *d_xy = nullptr;                          // [0] only: the points' xyuvav rows, uploaded with the corridors
    int32_t mn = 0, mu = 0, mv = 0;
    HostPivots() = default;
    HostPivots(const HostPivots &) = delete;
    HostPivots &operator=(const HostPivots &) = delete;
    ~HostPivots();
};

int vmap_geometry(const double *xyuvav, int32_t N, mimc3_vmap_result *res);
int vmap_host_pivots(mimc3_ctx *ctx, const double *xs, int32_t ns, float dt, float mpp, const mimc3_vmap_params *p, int32_t H, int32_t W,
                     HostPivots hp[4], std::string &err);
// (nctx > 1: the candidates of a segment are matched in slices, one per context -- mimc3_get_offset_image_multi)
int vmap_cp_offset(mimc3_ctx *const *ctxs, int32_t nctx, const double *xyuvav, int32_t N, const mimc3_vmap_params *p, uint8_t *flag_cp, mimc3_vmap_result *res);
int vmap_run_passes(mimc3_ctx *ctx, const double *xs, int32_t ns, const int32_t off[2], HostPivots hp[4], const mimc3_vmap_params *p,
                    float *d_dp, size_t pass_stride);

}  // namespace mimc3
