/*
 * mimc3_gma_shim.h -- struct-level drop-in for the reference's functions on the hot path and the stages either
 * side of it, with the reference's EXACT names and signatures, implemented on top of libmimc3_hip.so.
 *
 *   replaces (declared in the reference's MIMC_module.h)      reference definition
 *   ----------------------------------------------------      -------------------------
 *   get_uv_pivot              MIMC_module.h:41                MIMC_module.c:543-602
 *   matching_ncc_dlc_2        MIMC_module.h:46                MIMC_module.c:805-842
 *   get_ruv_neighbor          MIMC_module.h:56                MIMC_module.c:1266-1327
 *   get_dpf_pseudosmoothing   MIMC_module.h:58                MIMC_module.c:1986-2312
 *   get_offset_image          MIMC_module.h:34                MIMC_module.c:33-492     (N4)
 *   calc_mean_var_num_dp_cluster  MIMC_module.h:49            MIMC_module.c:994-1130   (N1)
 *   get_dpf0                  MIMC_module.h:53                MIMC_module.c:1224-1263  (N1)
 *   get_dpf1                  MIMC_module.h:56                MIMC_module.c:1330-1718  (N1)
 *   GMA_float_conv2           MIMC_module.h:67                MIMC_module.c:2517-2585  (N2)
 *   mimc2_postprocess         MIMC_module.h:48                MIMC_module.c:892-990    (N3)
 * These are all the symbols the reference's main() takes from MIMC_module.c: MIMC_main.c + GMA.c + MIMC_misc.c
 * link against this archive WITHOUT MIMC_module.c (`make -C oracle hybrid` does exactly that as a test).
 *
 * The structs below are layout-compatible re-declarations of the reference's array types
 * (GMA.h:68-91: {int32 ncols; int32 nrows; T **val; T *data;}) and of `param` (MIMC_module.h:9-25,
 * passed BY VALUE to get_uv_pivot, so its layout is part of the ABI).  A maintainer who builds the
 * reference against this shim keeps including the reference's own GMA.h / MIMC_module.h; this
 * header exists so the shim can be compiled and tested without any reference file.
 *
 * Ownership follows the reference: returned arrays are malloc'd as {struct, val, data} and are
 * released by the caller with GMA_*_destroy (GMA.c:128-164).  The QM shim reads the reference's
 * process globals dimx_vmap/dimy_vmap (MIMC_main.c:40), exactly like the function it replaces.
 * Errors: the reference has no error channel; the shim prints mimc3_last_error() to stderr and
 * abort()s (a silent wrong answer would be worse than the reference's own UB on such inputs).
 */
#ifndef MIMC3_GMA_SHIM_H
#define MIMC3_GMA_SHIM_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#ifndef _GMA_   /* the reference's GMA.h defines this guard; when it is included first, use its types */
typedef struct { int32_t ncols; int32_t nrows; uint8_t **val; uint8_t *data; } GMA_uint8;
typedef struct { int32_t ncols; int32_t nrows; int32_t **val; int32_t *data; } GMA_int32;
typedef struct { int32_t ncols; int32_t nrows; float **val; float *data; } GMA_float;
typedef struct { int32_t ncols; int32_t nrows; double **val; double *data; } GMA_double;
#endif
#ifndef _MIMC2_MODULE_
typedef struct param {
    int32_t vec_ocw[4];
    float AW_CRE, AW_SF, spacing_grid, radius_neighbor, radius_neighbor_dpf1, radius_neighbor_ps;
    float meter_per_spacing, mpp;
    int32_t num_cp_max, num_cp_min;
    float ratio_cp, thres_spd_cp;
} param;
#endif

GMA_int32 **get_uv_pivot(GMA_double *xyuvav, float dt, param param_mimc2, int32_t ocw, GMA_float *i1);
GMA_float *matching_ncc_dlc_2(GMA_float *i0, GMA_float *i1, GMA_double *xyuvav, int32_t *offset,
                              GMA_int32 **uv_pivot, int32_t ocw, float AW_CRE, float AW_SF);
GMA_int32 *get_ruv_neighbor(GMA_double *xyuvav, float radius_neighbor);
void get_dpf_pseudosmoothing(GMA_int32 *dpf, GMA_float *dpf_dx, GMA_float *dpf_dy, GMA_int32 *ruv_neighbor,
                             GMA_float **mvn_dp, GMA_double *xyuvav);

int get_offset_image(GMA_float *i0, GMA_float *i1, GMA_float **kernel, GMA_double *xyuvav, int32_t *offset, GMA_uint8 *flag_cp);
GMA_float **calc_mean_var_num_dp_cluster(GMA_float **dp, int32_t num_dpoi);
GMA_int32 *get_dpf0(GMA_float **mvn_dp, float min_matching_ratio);
void get_dpf1(GMA_int32 *dpf0, GMA_float *dpf_dx, GMA_float *dpf_dy, GMA_int32 *ruv_neighbor, GMA_float **mvn_dp, GMA_double *xyuvav);
void GMA_float_conv2(GMA_float *in, GMA_float *kernel, GMA_float *out);
GMA_float **mimc2_postprocess(GMA_float **dp, GMA_double *xyuvav, float dt);

/* optional: release the shim's device context (images, workspaces) before exit */
void mimc3_gma_shim_shutdown(void);

#ifdef __cplusplus
}
#endif
#endif
