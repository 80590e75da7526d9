// host_util.h -- error reporting shared by the host-side translation units (internal).
#pragma once
#include <cstdint>
#include <string>

namespace mimc3 {
// records the message for mimc3_last_error() (thread-local) and returns `code`
int fail(int code, const char *msg);
int fail(int code, const std::string &msg);

// Direction / length of one grid point's search corridor (MIMC_module.c:559-573): it does not depend on the chip size, so a
// driver that needs the pivots of several chip sizes computes it once per point (the atan2 / sin / cos are most of the cost).
struct CorridorPOD { float incr_u, incr_v, norm_incr; double length; };
// cor[N] for the N rows of xyuvav (threaded for large N)
void pivot_corridors(const double *xyuvav, int32_t N, float dt, float mpp, float aw_sf, float aw_cre, CorridorPOD *cor);
// mimc3_get_uv_pivot (same two-call protocol, same results) on precomputed corridors
// (`ext`, optional: what mimc3_pivot_extent would return -- max pivots per point, max |last pivot| per axis -- taken while filling)
int get_uv_pivot_cor(const CorridorPOD *cor, const double *xyuvav, int32_t N, int32_t ocw, int32_t H, int32_t W, int64_t *piv_off,
                     int32_t *piv_uv, int64_t cap, int64_t *total, int32_t *ext = nullptr);
}  // namespace mimc3
