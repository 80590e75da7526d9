/*
 * mimc3_hip.h -- C ABI of libmimc3_hip.so: the MI355X (gfx950) implementation of MIMC3's
 * per-grid-point DLC/NCC matching loop and QM pseudo-smoothing update.
 *
 * The reference has no plugin/FFI layer; its seam for this path is three plain C functions
 * (MIMC_module.h:41,46,58).  Each entry point below names the reference interface it replaces.
 * Plain pointers and sizes only; no GMA structs here (see mimc3_gma_shim.h for the struct-level
 * drop-in with the reference's exact signatures), no torch types.
 *
 * Conventions
 *   - every function returns 0 on success, a negative MIMC3_E* code, or a positive hipError_t;
 *     mimc3_last_error() returns a thread-local message for the last failure.
 *   - images are row-major float32 [H][W] as produced by GMA_float_load_tiff (GMA.c:246-316).
 *   - xyuvav is row-major float64 [N][6] = map x, map y, image u, image v, a-priori vx, vy.
 *   - DLC pivots are CSR: piv_off int64 [N+1], piv_uv int32 [P][2] (u, v offsets), which is the
 *     flattening of the reference's ragged `GMA_int32 **uv_pivot`.
 *   - "_dev" functions take DEVICE pointers and enqueue on the given hipStream_t (passed as
 *     void*; NULL = the default stream) without synchronising; the others take HOST pointers,
 *     copy, run, and synchronise (drop-in semantics).
 *   - a context owns scratch that its calls share (overflow lists, planes, staging buffers): keep ONE stream in flight
 *     per context -- issue the "_dev" calls of a context on one stream, and drain it before a call that rebuilds the
 *     planes (set_images*, filter_images).  Concurrency comes from several contexts (one per device, or several per
 *     device: the control-point stage does exactly that internally).
 *   - there is NO CPU fallback anywhere in this library.
 */
#ifndef MIMC3_HIP_H
#define MIMC3_HIP_H
#include <stdint.h>
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

#define MIMC3_EINVAL   (-1)  /* bad argument (null pointer, non-positive size, ocw < 2 ...)      */
#define MIMC3_EBOUNDS  (-2)  /* a chip would leave the image, or a point has zero pivots: the
                                reference reads/writes out of bounds there (MIMC_module.c:852,
                                :589-591); this library refuses instead                        */
#define MIMC3_ECAP     (-3)  /* caller-provided output capacity too small                        */
#define MIMC3_ENODEV   (-4)  /* no usable HIP device                                             */
#define MIMC3_ESTATE   (-5)  /* context in the wrong state (e.g. images not set)                 */

typedef struct mimc3_ctx mimc3_ctx;   /* opaque: device id, stream, resident images, workspaces */

/* ---- context: keeps both images resident in HBM across the CLI's 32 matcher passes
 *      (MIMC_main.c:261-350 calls the matcher 32x on 4 image pairs) ------------------------- */
int  mimc3_ctx_create(int device, mimc3_ctx **out);
void mimc3_ctx_destroy(mimc3_ctx *ctx);
const char *mimc3_last_error(void);

/* Upload a host image pair (replaces the reference holding GMA_float *i0,*i1 in host RAM). */
int mimc3_ctx_set_images(mimc3_ctx *ctx, const float *i0, const float *i1, int32_t H, int32_t W);
/* The same from the RAW DN the TIFF holds (the widening to float32 of GMA_float_load_tiff, GMA.c:288-310, then runs on
 * the device): 1 or 2 bytes per pixel cross PCIe instead of 4, and 8-bit DN lands directly in the exact-integer
 * kernel's planes.  Results are identical to widening on the host and calling mimc3_ctx_set_images.
 * Any host pointer works; memory from mimc3_host_alloc (pinned) is DMA'd without the staging copy. */
int mimc3_ctx_set_images_u8(mimc3_ctx *ctx, const uint8_t *i0, const uint8_t *i1, int32_t H, int32_t W);
int mimc3_ctx_set_images_u16(mimc3_ctx *ctx, const uint16_t *i0, const uint16_t *i1, int32_t H, int32_t W);
void *mimc3_host_alloc(size_t bytes);     /* pinned host memory (NULL on failure); read TIFF scanlines straight into it */
void  mimc3_host_free(void *p);
/* Adopt device-resident images (no copy; caller keeps ownership, must outlive the context use). */
int mimc3_ctx_set_images_dev(mimc3_ctx *ctx, const float *d_i0, const float *d_i1, int32_t H, int32_t W);

/* Kernel selection.  By default (mode 0) the library picks, per matcher call:
 *   1 = exact-integer u8 kernel when BOTH resident images were proven (on the device, at set_images
 *       time) to hold only integers in [0,255] -- the 8-bit TIFF case of GMA_float_load_tiff
 *       (GMA.c:288-298) -- and ocw is one of 7, 15, 16, 30, 32, 40;
 *   3 = exact scaled-integer u16 kernel when both images hold only values q/2^s with q < 4096 (12-bit DN,
 *       or what GMA_float_conv2 makes of 8-bit images: integers <= 511 / multiples of 1/8), same ocw set;
 *   4 = the u8 kernel read through per-point offsets, for INTEGER images of kind 3 whose values stay within an
 *       8-bit range locally (the gradient filters of 8-bit images); the few points whose chip or window does
 *       not fit are redone by kernel 3 right behind.  Exact like 3 (same integer sums, rebuilt from q - k);
 *   2 = register-tiled f32 kernel (any f32 imagery, e.g. 16-bit DN) when ocw is one of 7, 15, 16, 30, 32, 40;
 *   5 = the matrix-core form of kernel 1 (dense correlation surfaces on v_mfma_i32_16x16x64_i8), taken first for the
 *       chip sizes it is built for; the points it does not take (null pixels in the window or chip, corridors wider than
 *       its 32 x 32 cell tile, ...) are flagged and done by kernel 1 right behind;
 *   0 = general f32 kernel (any ocw, any window size) otherwise.
 * All three give results bit-identical to the reference on integral-DN data.  mode 1 forces kernel 0,
 * mode 2 skips the integer kernels, mode 3 skips only the u8 kernel, mode 4 is mode 0 without kernel 5 (tests use
 * them to cover every kernel on 8-bit inputs too).
 * mimc3_ctx_last_path returns the kernel of the last call (<0 = none yet). */
int mimc3_ctx_set_path(mimc3_ctx *ctx, int32_t mode);
int mimc3_ctx_last_path(mimc3_ctx *ctx);

/* ---- a2: DLC pivot generator.  Replaces get_uv_pivot (MIMC_module.h:41, MIMC_module.c:543-602).
 *      Host code (libm-exact float/double mix of the reference).  Two-call protocol: pass
 *      piv_uv=NULL to get the total pivot count in *total and piv_off filled; then call again
 *      with capacity `cap` (pairs).  MIMC3_EBOUNDS if any point gets zero pivots. ------------- */
int mimc3_get_uv_pivot(const double *xyuvav, int32_t N, float dt, float mpp, float aw_sf, float aw_cre,
                       int32_t ocw, int32_t H, int32_t W,
                       int64_t *piv_off /*[N+1]*/, int32_t *piv_uv /*[cap][2] or NULL*/, int64_t cap,
                       int64_t *total);

/* ---- a3-a7: matcher.  Replaces matching_ncc_dlc_2 (MIMC_module.h:46, MIMC_module.c:805-842)
 *      including extract_refchip/extract_sarea/investigate_valid_grid/find_ncc_peak.
 *      out [N][3] = (du, dv, ncc_peak); invalid point = (NaN, NaN, -3)   (MIMC_module.c:685-687).
 *      `swap` != 0 matches i1 -> i0 (the CLI's "swapped forward" pass, MIMC_main.c:284): the
 *      caller still negates offset, pivots and the resulting (du,dv) exactly as main() does. --- */
int mimc3_match_ncc_dlc(mimc3_ctx *ctx, const double *xyuvav, int32_t N, const int32_t offset[2],
                        const int32_t *piv_uv, const int64_t *piv_off, int32_t ocw, int32_t swap,
                        float *out /*[N][3] host*/);

/* Device-resident variant used by bench.py and the multi-GPU shard path: all pointers are device
 * pointers; `max_abs_piv_u/v` = max over points of |last pivot| per axis (sizes the LDS window;
 * mimc3_pivot_extent() computes it from a host CSR).  Enqueues on `stream`, no sync. */
int mimc3_match_ncc_dlc_dev(mimc3_ctx *ctx, const double *d_xyuvav, int32_t N, int32_t off_u, int32_t off_v,
                            const int32_t *d_piv_uv, const int64_t *d_piv_off, int32_t max_npiv,
                            int32_t max_abs_piv_u, int32_t max_abs_piv_v, int32_t ocw, int32_t swap,
                            float *d_out, void *stream);
int mimc3_pivot_extent(const int32_t *piv_uv, const int64_t *piv_off, int32_t N,
                       int32_t *max_npiv, int32_t *max_abs_u, int32_t *max_abs_v);

/* ---- a2 on the device.  get_uv_pivot has two halves: the CORRIDOR of a point (theta = atan2(vy, vx), the normalised step,
 *      the corridor length, MIMC_module.c:559-573) needs libm and is computed on the host, bit-equal to the reference's;
 *      the pivot LIST (:576-598) is plain IEEE arithmetic on those numbers and is expanded by a kernel.  24 bytes per grid
 *      point cross PCIe instead of 8 bytes per pivot, and the lists never exist on the host.
 *      mimc3_pivot_corridors: cor = [N] records of MIMC3_CORRIDOR_BYTES bytes (host; opaque to the caller).
 *      mimc3_get_uv_pivot_dev: all pointers are device pointers; the context's image size bounds the pivots (as `i1` does in
 *      the reference).  Same two-call protocol as mimc3_get_uv_pivot (both list pointers NULL = offsets, total and extents
 *      only); d_piv_uv_neg (optional) receives the negated list main() makes in place for its swapped pass
 *      (MIMC_main.c:272-279).  extent[3] = what mimc3_pivot_extent returns.  Synchronises `stream` once (a 24-byte read-back).
 *      Results are bit-equal to mimc3_get_uv_pivot's. ----------------------------------------------------------------- */
#define MIMC3_CORRIDOR_BYTES 24
int mimc3_pivot_corridors(const double *xyuvav, int32_t N, float dt, float mpp, float aw_sf, float aw_cre, void *cor /*[N][24 B]*/);
int mimc3_get_uv_pivot_dev(mimc3_ctx *ctx, const double *d_xyuvav, const void *d_cor, int32_t N, int32_t ocw,
                           int64_t *d_piv_off /*[N+1]*/, int32_t *d_piv_uv /*[cap][2] or NULL*/, int32_t *d_piv_uv_neg /*[cap][2] or NULL*/,
                           int64_t cap, int64_t *total, int32_t extent[3], void *stream);
/* get_uv_pivot + matching_ncc_dlc_2 in one call, as main() pairs them (MIMC_main.c:264-267, :281-284): host buffers in and
 * out like mimc3_match_ncc_dlc, but the pivots are made on the device from the uploaded corridors.  `swap` != 0 is the
 * swapped pass: images exchanged AND the pivots negated (the caller still negates `offset` and the resulting (du, dv)). */
int mimc3_match_ncc_dlc_geo(mimc3_ctx *ctx, const double *xyuvav, int32_t N, const int32_t offset[2], float dt, float mpp,
                            float aw_sf, float aw_cre, int32_t ocw, int32_t swap, float *out /*[N][3] host*/);
/* The same with the corridors made beforehand by mimc3_pivot_corridors (they do not depend on the chip size: main() would make
 * them once for its eight raw-image calls, as it makes the pivot lists once per chip size).  Per grid point 16 B of (u, v),
 * 24 B of corridor go up and 12 B of result come down; the grid is pipelined through in chunks, the transfers running under
 * the matcher.  Pinned `xyuvav`-independent: any host pointers work, pinned `cor` / `out` (mimc3_host_alloc) overlap best. */
int mimc3_match_ncc_dlc_cor(mimc3_ctx *ctx, const double *xyuvav, const void *cor /*[N][24 B] host*/, int32_t N, const int32_t offset[2],
                            int32_t ocw, int32_t swap, float *out /*[N][3] host*/);

/* ---- a8: neighbour offsets.  Replaces get_ruv_neighbor (MIMC_module.h:56, :1266-1327).
 *      Host code.  Returns the count in *nn; MIMC3_ECAP if it exceeds cap (pairs). ------------- */
int mimc3_get_ruv_neighbor(const double *xyuvav, int32_t N, int32_t dimx, int32_t dimy,
                           float meter_per_spacing, float radius, int32_t *ruv /*[cap][2]*/, int32_t cap,
                           int32_t *nn);

/* ---- a9-a10: QM pseudo-smoothing.  Replaces get_dpf_pseudosmoothing (MIMC_module.h:58,
 *      MIMC_module.c:1986-2312) incl. quadfit2 / GMA_double_inv.  In place on dpf, dpf_dx, dpf_dy
 *      ([dimy][dimx]).  mvn = candidates padded to [N][Kmax][5] (mean_u, mean_v, var_u, var_v,
 *      fraction) with nclus[N] valid rows (flattening of `GMA_float **mvn_dp`).
 *      max_sweeps: the reference loops while NOI<=100, i.e. at most 101 sweeps -> pass 101 for
 *      drop-in behaviour (BASELINE config C5 passes 10).  sweeps_done may be NULL. ------------- */
int mimc3_qm_pseudosmooth(mimc3_ctx *ctx, int32_t dimy, int32_t dimx, int32_t *dpf, float *dpf_dx, float *dpf_dy,
                          const int32_t *ruv, int32_t nn, const float *mvn, int32_t Kmax, const int32_t *nclus,
                          const double *xyuvav, int32_t max_sweeps, int32_t *sweeps_done);
/* Device-resident variant: d_work must hold mimc3_qm_workspace_bytes(dimy*dimx, max_sweeps) bytes.
 * Runs up to max_sweeps sweeps back-to-back with device-side early-out (no host sync). */
int64_t mimc3_qm_workspace_bytes(int32_t ngrid, int32_t max_sweeps);
int32_t mimc3_qm_launches_per_sweep(void);   /* kernel launches enqueued per sweep (2: fit + commit/compare/decide) */
int mimc3_qm_pseudosmooth_dev(mimc3_ctx *ctx, int32_t dimy, int32_t dimx, int32_t *d_dpf, float *d_dpf_dx,
                              float *d_dpf_dy, const int32_t *d_ruv, int32_t nn, const float *d_mvn, int32_t Kmax,
                              const int32_t *d_nclus, const double *d_xyuvav, int32_t max_sweeps,
                              void *d_work, int32_t *d_sweeps_done, void *stream);

/* ---- N1 (the stages between the 32 matcher passes and the QM update) ------------------------------
 *      Candidate clustering.  Replaces calc_mean_var_num_dp_cluster (MIMC_module.h:49, MIMC_module.c:994-1130)
 *      with cluster_euclidian / mark_row (:1133-1222).  dp = the ndp matcher outputs, pass-major
 *      [ndp][N][3] (flattening of `GMA_float **dp`), 1 <= ndp <= 64.  mvn out = [N][Kmax][5] (mean_u, mean_v,
 *      var_u, var_v, fraction; padding rows zero), nclus out = [N].  *kmax_seen = the largest cluster count;
 *      MIMC3_ECAP if it exceeds Kmax (Kmax = ndp is always enough). ------------------------------------ */
int mimc3_cluster_candidates(mimc3_ctx *ctx, const float *dp, int32_t ndp, int32_t N, int32_t Kmax, float *mvn,
                             int32_t *nclus, int32_t *kmax_seen);
int mimc3_cluster_candidates_dev(mimc3_ctx *ctx, const float *d_dp, int32_t ndp, int32_t N, int32_t Kmax,
                                 float *d_mvn, int32_t *d_nclus, int32_t *d_kmax_seen, void *stream);

/*      Prominent-cluster pick.  Replaces get_dpf0 (MIMC_module.h:52, MIMC_module.c:1224-1263):
 *      dpf[g] = first cluster whose fraction > min_ratio, else -1. ---------------------------------- */
int mimc3_get_dpf0(mimc3_ctx *ctx, const float *mvn, const int32_t *nclus, int32_t N, int32_t Kmax, float min_ratio,
                   int32_t *dpf);
int mimc3_get_dpf0_dev(mimc3_ctx *ctx, const float *d_mvn, const int32_t *d_nclus, int32_t N, int32_t Kmax,
                       float min_ratio, int32_t *d_dpf, void *stream);

/*      A-priori-guided fill of the unassigned points.  Replaces get_dpf1 (MIMC_module.h:54,
 *      MIMC_module.c:1330-1718).  dpf in = dpf0, out = dpf1 ([dimy][dimx]); dpf_dx, dpf_dy out.  dt, mpp = the
 *      reference's globals `dt` and `param_mimc2.mpp`.  The sweep count is data dependent and unbounded in
 *      the reference, so even the _dev variant synchronises `stream` once per 32 sweeps to poll the device's
 *      done flag; d_work must hold mimc3_dpf1_workspace_bytes(dimy*dimx).  sweeps_done (host, may be NULL)
 *      receives the reference's NOI. -------------------------------------------------------------------- */
int mimc3_get_dpf1(mimc3_ctx *ctx, int32_t dimy, int32_t dimx, int32_t *dpf, float *dpf_dx, float *dpf_dy,
                   const int32_t *ruv, int32_t nn, const float *mvn, int32_t Kmax, const int32_t *nclus,
                   const double *xyuvav, float dt, float mpp, int32_t *sweeps_done);
int64_t mimc3_dpf1_workspace_bytes(int32_t ngrid);
int mimc3_get_dpf1_dev(mimc3_ctx *ctx, int32_t dimy, int32_t dimx, int32_t *d_dpf, float *d_dpf_dx, float *d_dpf_dy,
                       const int32_t *d_ruv, int32_t nn, const float *d_mvn, int32_t Kmax, const int32_t *d_nclus,
                       const double *d_xyuvav, float dt, float mpp, void *d_work, int32_t *sweeps_done, void *stream);

/* ---- N2: image pre-filter.  Replaces GMA_float_conv2 (MIMC_module.h:67, MIMC_module.c:2517-2585): correlation
 *      with a kh x kw kernel (row-major, at most 81 taps) over the interior, a null DN poisoning its stencil,
 *      then `out -= min-1` / poisoned -> 0.  `out` [H][W] is IN/OUT exactly as in the reference: its border
 *      rows/columns are never written by the stencil but take part in the minimum and (right-hand columns) in
 *      the shift, so pass the buffer the reference would have (a fresh GMA_float_create plane: zeros).
 *      _dev: d_scratch = 4 bytes of device memory; enqueues on `stream`, no sync. ---------------------- */
int mimc3_float_conv2(mimc3_ctx *ctx, const float *in, int32_t H, int32_t W, const float *kernel, int32_t kh,
                      int32_t kw, float *out);
int mimc3_float_conv2_dev(mimc3_ctx *ctx, const float *d_in, int32_t H, int32_t W, const float *kernel /*host*/,
                          int32_t kh, int32_t kw, float *d_out, void *d_scratch, void *stream);
/*      Filter the context's resident pair on the device and make the filtered pair the one the matcher uses
 *      (what MIMC_main.c:302-350 does with host copies for its 24 filtered passes).  Like the reference, the two
 *      output planes are created once per pair (zeros) and REUSED by consecutive calls, so the border a filter
 *      leaves behind is input to the next one; mimc3_ctx_set_images* starts over, and so does kernel = NULL,
 *      which goes back to the pair as handed over (one run of the reference program = one such sequence).
 *      Nothing crosses PCIe.
 *      mimc3_ctx_get_images downloads the pair currently in use (either pointer may be NULL). --------- */
int mimc3_ctx_filter_images(mimc3_ctx *ctx, const float *kernel, int32_t kh, int32_t kw);
int mimc3_ctx_get_images(mimc3_ctx *ctx, float *i0, float *i1);

/* ---- N4: control-point offset.  Replaces get_offset_image (MIMC_module.h:34, MIMC_module.c:33-492) incl.
 *      GMA_double_randperm_row (:494-541).  Works on the context's resident pair AS HANDED OVER (any
 *      mimc3_ctx_filter_images state is ignored).  The reference's globals travel in mimc3_cp_params; its
 *      srand(time(NULL)) becomes `seed` (< 0 = time(NULL)), so a run can be repeated.  *status receives the
 *      reference's return value: 1 = offset valid, -1 = not enough control points (offset untouched; the CLI
 *      then gives up on the pair, MIMC_main.c:246-252).  flag_cp [N] bytes: set to 1 for every grid point that
 *      voted (never cleared: pass zeros, as GMA_uint8_create gives).  info (may be NULL) = #candidates, CP
 *      threshold, segments run, CPs found; sduv (may be NULL) = the two vote sums.
 *      MIMC3_EBOUNDS if a candidate's chip (+-(vec_ocw[2]+AW_CRE+3) px) leaves the image. ---------------- */
typedef struct mimc3_cp_params {
    int32_t vec_ocw[4];       /* param.vec_ocw: [1] and [2] are matched, [2] sizes the chips and the validity test */
    float aw_cre;             /* param.AW_CRE: rectangular pivot set -AW_CRE..AW_CRE in u and v                  */
    int32_t num_cp_max, num_cp_min;
    float ratio_cp, thres_spd_cp;
    const float *kernel[3];   /* the CLI's three pre-filter kernels (MIMC_main.c:176-194), row-major              */
    int32_t kdim[3][2];       /* rows, cols of each (at most 3 x 3)                                               */
    int64_t seed;
} mimc3_cp_params;
int mimc3_get_offset_image(mimc3_ctx *ctx, const double *xyuvav, int32_t N, const mimc3_cp_params *params,
                           int32_t offset[2], uint8_t *flag_cp, int32_t *status, int32_t *info /*[4]*/,
                           float *sduv /*[2]*/);
/*      The same stage with its device work shared by `nctx` contexts (one per GPU, each holding the SAME pair):
 *      the candidates of every segment are cut into nctx contiguous slices -- candidates are as independent as
 *      grid points (MIMC_module.c:325-378) -- which the contexts match side by side on host threads of their
 *      own.  What the reference does in sequence stays on the calling thread (the rand() shuffle, the segment
 *      loop and its early exit, the chip-to-chip recurrence of the filtered planes' minima, the f32 vote sums in
 *      candidate order): the result is that of the one-context call, whatever nctx is.  ctxs[0] does the
 *      candidate selection and the segment-wide minima.  mimc3_mgpu_vmap calls this with all of its ranks.    */
int mimc3_get_offset_image_multi(mimc3_ctx *const *ctxs, int32_t nctx, const double *xyuvav, int32_t N,
                                 const mimc3_cp_params *params, int32_t offset[2], uint8_t *flag_cp,
                                 int32_t *status, int32_t *info /*[4]*/, float *sduv /*[2]*/);

/* ---- N3: the program's data path on arrays ---------------------------------------------------------------
 *      mimc3_postprocess replaces mimc2_postprocess (MIMC_module.h:48, MIMC_module.c:892-990): clustering ->
 *      dpf0 (ratio 0.6) -> dpf1 (radius_dpf1) -> QM pseudo-smoothing (radius_ps, qm_max_sweeps: 101 = reference)
 *      -> out5 [5][dimy*dimx] = mean_u, mean_v, var_u, var_v, fraction of the chosen cluster, NaN where none
 *      (the reference's vxyexyqual[0..4] BEFORE main()'s unit conversion).  dp = [ndp][N][3] pass-major.
 *      _dev: d_dp, d_xyuvav, d_out5 on the device; xyuvav also on the host (neighbour geometry is host code);
 *      synchronises `stream` (data-dependent sweep counts, temporary buffers). ------------------------------ */
int mimc3_postprocess(mimc3_ctx *ctx, const float *dp, int32_t ndp, const double *xyuvav, int32_t dimx, int32_t dimy,
                      float dt, float mpp, float meter_per_spacing, float radius_dpf1, float radius_ps,
                      int32_t qm_max_sweeps, float *out5);
int mimc3_postprocess_dev(mimc3_ctx *ctx, const float *d_dp, int32_t ndp, const double *xyuvav, const double *d_xyuvav,
                          int32_t dimx, int32_t dimy, float dt, float mpp, float meter_per_spacing, float radius_dpf1,
                          float radius_ps, int32_t qm_max_sweeps, float *d_out5, void *stream);

/*      mimc3_vmap = MIMC_main.c:203-402, from "xyuvav and both images loaded" to "save the output", on the
 *      context's resident pair: grid geometry (:209-223), CP offset (:240-256), the 32 matcher passes
 *      (:261-350; pivots, images, candidates never leave the device), mimc2_postprocess (:353), removal of the
 *      sub-integer CP offset and px -> m/yr (:356-402).  Outputs [dimy*dimx] f32 as the reference saves them:
 *      vx, vy (m/yr, vy north-positive), ex, ey (m/yr), qual; flag_cp [N] bytes.  res->cp_status = -1 means
 *      "not enough control points": nothing else is computed (the CLI then touches vmap_*.tar, :248-252). --- */
typedef struct mimc3_vmap_params {
    int32_t vec_ocw[4];                 /* MIMC_main.c:134-137: 7, 15, 30, 40                        */
    float aw_cre, aw_sf;                /* :154-155: 10.0, 1.8                                        */
    float radius_neighbor_dpf1;         /* :164: 1000/300 = 3 (grid spacings)                         */
    float radius_neighbor_ps;           /* :165: 5.0                                                  */
    int32_t num_cp_max, num_cp_min;     /* :168-169: 500, 50                                          */
    float ratio_cp, thres_spd_cp;       /* :170-171: 0.03, 10                                         */
    const float *kernel[3];             /* :176-194: d/dx 1x3, d/dy 3x1, Laplacian 3x3, row-major     */
    int32_t kdim[3][2];
    int64_t cp_seed;                    /* shuffle seed of the CP stage; < 0 = time(NULL)             */
    int32_t qm_max_sweeps;              /* 0 or 101 = reference                                        */
} mimc3_vmap_params;
typedef struct mimc3_vmap_result {
    int32_t dimx, dimy;
    float mpp, spacing_grid, meter_per_spacing;
    int32_t cp_status;                  /* 1 ok, -1 not enough control points                         */
    int32_t offset_cp[2];               /* integer CP offset (meta: cp_offset_int_u/v)                */
    float cp_subint[2];                 /* grid mean removed afterwards (meta: cp_offset_subint_u/v)  */
} mimc3_vmap_result;
int mimc3_vmap(mimc3_ctx *ctx, const double *xyuvav, int32_t N, float dt, const mimc3_vmap_params *params,
               float *vx, float *vy, float *ex, float *ey, float *qual, uint8_t *flag_cp, mimc3_vmap_result *res);
/*      The same in two steps, for a multi-GPU driver (grid points are independent in the matcher):
 *      mimc3_vmap_passes = geometry + CP offset on the WHOLE grid + the 32 passes for grid points [lo, hi) only, into
 *      d_dp [32][hi-lo][3] (device, pass-major); the caller all-gathers the blocks into [32][N][3] and every rank
 *      calls mimc3_vmap_finish (post-processing, unit conversion) with the `res` its own passes call filled.
 *      mimc3_vmap is passes(0, N) + finish. ---------------------------------------------------------------------- */
int mimc3_vmap_passes(mimc3_ctx *ctx, const double *xyuvav, int32_t N, float dt, const mimc3_vmap_params *params,
                      int32_t lo, int32_t hi, float *d_dp, uint8_t *flag_cp, mimc3_vmap_result *res);
int mimc3_vmap_finish(mimc3_ctx *ctx, const double *xyuvav, int32_t N, float dt, const mimc3_vmap_params *params,
                      const float *d_dp, float *vx, float *vy, float *ex, float *ey, float *qual, mimc3_vmap_result *res);
/*      ... and in finer pieces, for a driver that measures the CP offset ONCE and shards by cost-balanced point sets:
 *      mimc3_vmap_geometry = grid geometry only (:209-223); mimc3_vmap_cp = geometry + CP offset (:240-256) -> res, flag_cp;
 *      mimc3_vmap_passes_points = host pivots + the 32 passes for ANY set of grid points xs [n][6] with the CP offset in
 *      `res`, into d_dp [32][pass_stride][3] (pass_stride >= n points per pass slot; 0 = n). -------------------------------- */
int mimc3_vmap_geometry(const double *xyuvav, int32_t N, mimc3_vmap_result *res);
int mimc3_vmap_cp(mimc3_ctx *ctx, const double *xyuvav, int32_t N, float dt, const mimc3_vmap_params *params, uint8_t *flag_cp,
                  mimc3_vmap_result *res);
int mimc3_vmap_passes_points(mimc3_ctx *ctx, const double *xs, int32_t n, float dt, const mimc3_vmap_params *params,
                             const mimc3_vmap_result *res, float *d_dp, int64_t pass_stride);

/* ---- e: multi-GPU.  The loop that shards is the reference's OpenMP loop over grid points (MIMC_module.c:816-838): grid
 *      points are independent in the matcher, so each GPU matches its own share against the replicated pair and ONE
 *      all-gather re-assembles the result.  Two forms:
 *      (1) one process per GPU (torch.distributed / MPI launchers): mimc3_vmap_passes + the caller's own all-gather +
 *          mimc3_vmap_finish, above;
 *      (2) ONE process driving several GPUs, here: a host thread per device and an RCCL communicator over them
 *          (ncclCommInitAll; RCCL -- librccl.so.1 -- is loaded with dlopen on first use).
 *          The MIMC3_hip command line takes this form with MIMC3_HIP_DEVICES=0,1,...
 *      Shares are cost-balanced: mimc3_point_cost adds (4 + 6 npiv)(2 ocw + 1)^2 per point (NCC evaluations x chip area),
 *      mimc3_partition_points cuts the grid into blocks of `block` consecutive points, deals them heaviest-first to the
 *      least loaded rank and returns order[start[r] .. start[r+1]) = the points of rank r (each rank's blocks in grid
 *      order); *imbalance = max load / mean load - 1.  Host code. ------------------------------------------------------ */
int mimc3_point_cost(const int64_t *piv_off, int32_t N, int32_t ocw, double *cost /*[N], accumulated*/);
int mimc3_partition_points(const double *cost, int32_t N, int32_t world, int32_t block, int32_t *order /*[N]*/,
                           int32_t *start /*[world+1]*/, double *imbalance /*may be NULL*/);
typedef struct mimc3_mgpu mimc3_mgpu;
int  mimc3_mgpu_create(const int32_t *devices, int32_t ndev, mimc3_mgpu **out);   /* a context per device + the communicator */
/* Test / bring-up form of the above (the product entry point never takes these, and the library reads neither from the environment):
 * comm_lib = a library exporting the six RCCL entry points the driver uses (NULL: RCCL); MIMC3_MGPU_REPEAT_DEVICES in flags lets a
 * device be listed more than once (N ranks as N contexts of one GPU over a stand-in communicator). */
#define MIMC3_MGPU_REPEAT_DEVICES 1u
int  mimc3_mgpu_create_ex(const int32_t *devices, int32_t ndev, const char *comm_lib, uint32_t flags, mimc3_mgpu **out);
void mimc3_mgpu_destroy(mimc3_mgpu *mg);
int32_t mimc3_mgpu_ndev(mimc3_mgpu *mg);
mimc3_ctx *mimc3_mgpu_ctx(mimc3_mgpu *mg, int32_t rank);
double mimc3_mgpu_last_imbalance(mimc3_mgpu *mg);                                 /* of the last call's partition */
/*      the pair, replicated on every device (uploads run in parallel) */
int mimc3_mgpu_set_images(mimc3_mgpu *mg, const float *i0, const float *i1, int32_t H, int32_t W);
int mimc3_mgpu_set_images_u8(mimc3_mgpu *mg, const uint8_t *i0, const uint8_t *i1, int32_t H, int32_t W);
int mimc3_mgpu_set_images_u16(mimc3_mgpu *mg, const uint16_t *i0, const uint16_t *i1, int32_t H, int32_t W);
/*      mimc3_match_ncc_dlc / mimc3_vmap with the grid points sharded over the devices; same arguments, same results
 *      (bit-identical: a point's result does not depend on which device computes it).  The CP offset of mimc3_mgpu_vmap
 *      is measured once, on device 0; post-processing runs on device 0. */
int mimc3_mgpu_match_ncc_dlc(mimc3_mgpu *mg, const double *xyuvav, int32_t N, const int32_t offset[2], const int32_t *piv_uv,
                             const int64_t *piv_off, int32_t ocw, int32_t swap, float *out /*[N][3] host*/);
int mimc3_mgpu_vmap(mimc3_mgpu *mg, const double *xyuvav, int32_t N, float dt, const mimc3_vmap_params *params, float *vx, float *vy,
                    float *ex, float *ey, float *qual, uint8_t *flag_cp, mimc3_vmap_result *res);

/*      small device helpers the driver is built from: the context's own stream; (du,dv) -> (-du,-dv) of a swapped
 *      pass (MIMC_main.c:289-293); cluster map -> five planes (mimc2_postprocess :937-970 /
 *      convert_dpf_to_vxy_exy_qual, MIMC_module.h:64); size of the resident pair. --------------------------- */
void *mimc3_ctx_stream(mimc3_ctx *ctx);
int mimc3_negate_uv_dev(mimc3_ctx *ctx, float *d_out, int32_t N, void *stream);
int mimc3_negate_pivots_dev(mimc3_ctx *ctx, const int32_t *d_piv_uv, int32_t *d_out /*[count][2]*/, int64_t count, void *stream); /* :272-279 */
int mimc3_dpf_to_vxyexyqual_dev(mimc3_ctx *ctx, const int32_t *d_dpf, const float *d_mvn, int32_t N, int32_t Kmax,
                                float *d_out5, void *stream);
int mimc3_ctx_image_size(mimc3_ctx *ctx, int32_t *H, int32_t *W);
int mimc3_ctx_device(mimc3_ctx *ctx);
/*      device scratch owned by the context: slot 0..15, grows on demand, contents kept until the slot is asked for more
 *      bytes; freed with the context.  The drivers above the ABI (mimc3_postprocess, mimc3_vmap) keep their working
 *      buffers here instead of allocating per call.  One stream at a time per context. */
int mimc3_ctx_workspace(mimc3_ctx *ctx, int32_t slot, size_t bytes, void **d_ptr);
/*      one of the context's four auxiliary hipStream_t (k = 0..3; created with the context -- creating a stream while
 *      kernels run costs milliseconds): the drivers' host threads upload on them next to the context's own stream. */
void *mimc3_ctx_aux_stream(mimc3_ctx *ctx, int32_t k);
/*      the same for PINNED host memory (slot 0..7): pinning pages costs milliseconds per tens of MB, so the drivers keep
 *      their staging buffers (the pivot lists of the 32 passes) across calls.  Distinct slots may be asked for from
 *      distinct host threads at the same time. */
int mimc3_ctx_host_workspace(mimc3_ctx *ctx, int32_t slot, size_t bytes, void **h_ptr);

/* ---- measurement helper: average device time (ms) of the last matcher launch sequence,
 *      taken with hipEvents on the launch stream (bench.py's roofline leg). -------------------- */
int mimc3_ctx_enable_timing(mimc3_ctx *ctx, int32_t on);
int mimc3_ctx_last_kernel_ms(mimc3_ctx *ctx, float *ms);

const char *mimc3_version(void);

#ifdef __cplusplus
}
#endif
#endif /* MIMC3_HIP_H */
