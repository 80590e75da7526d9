// n1_kernel.hip -- the stages between the 32 matcher passes and the QM update, for gfx950 (MI355X):
//
//   cluster_candidates : calc_mean_var_num_dp_cluster (MIMC_module.c:994-1130) with cluster_euclidian /
//                        mark_row (:1133-1222) -- single-linkage clustering (< 0.5 px) of the <= 64 matcher
//                        candidates of a grid point, mean / variance / fraction per cluster
//   dpf0               : get_dpf0 (:1224-1263) -- first cluster whose fraction exceeds the ratio
//   dpf1_*             : get_dpf1 (:1330-1718) -- Jacobi interpolation of the unassigned points along the
//                        a-priori flow, 3x3 smoothing, snap to the nearest cluster
//
// Mapping.  Clustering is one wave per grid point, one lane per matcher pass: the adjacency row of a
// candidate is a 64-bit lane mask, connected components grow by ballot (a lane joins when its row meets
// the frontier), and each cluster's sums are accumulated in candidate order (f32, like the reference).
// A block stages 64 consecutive grid points of every pass through LDS so that the pass-major candidate
// planes are read coalesced.  dpf1 is one thread per grid point per sweep; the sweep/level control of the
// reference (:1383-1621) runs on the device in the sweep kernel's last block, the host only polls "done".
//
// Arithmetic follows the reference's C promotions literally (f32 products, sqrt() in f64 rounded to f32,
// one f32/f64 quotient), compiled with -ffp-contract=off.  expf() is taken as the f32 rounding of the f64
// exp(), which equals glibc's expf result except when the exact value lies within ~2^-29 ulp of a rounding
// boundary; the outputs are cluster ids and copied cluster means, so they are bit-identical unless such a
// case also flips a nearest-cluster decision ("N1").
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "n1_kernel.h"

namespace mimc3 {

namespace {

constexpr int kPitch = kCluPointsPerBlock * 3 + 1;   // odd LDS pitch: lane k reads row k without bank conflicts

__device__ __forceinline__ float lane_f(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
__device__ __forceinline__ uint64_t lane_u64(uint64_t v, int l)
{
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, l);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), l);
    return ((uint64_t)hi << 32) | lo;
}

__global__ __launch_bounds__(kCluThreads) void cluster_kernel(ClusterArgs a)
{
    extern __shared__ float sdp[];                       // [ndp][kPitch]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g0 = blockIdx.x * kCluPointsPerBlock;
    const int npts = min(kCluPointsPerBlock, a.N - g0);
    const int row = kCluPointsPerBlock * 3;
    for (int idx = tid; idx < a.ndp * row; idx += kCluThreads) {
        const int k = idx / row, r = idx - k * row;
        sdp[k * kPitch + r] = (r < npts * 3) ? a.dp[((size_t)k * a.N + g0) * 3 + r] : 0.0f;
    }
    __syncthreads();

    for (int p = wave; p < npts; p += kCluThreads / 64) {
        const int g = g0 + p;
        float x = 0.0f, y = 0.0f, c = -1.0f;
        if (lane < a.ndp) { x = sdp[lane * kPitch + 3 * p]; y = sdp[lane * kPitch + 3 * p + 1]; c = sdp[lane * kPitch + 3 * p + 2]; }
        const bool valid = (lane < a.ndp) && (c > 0.1f);                       // :1010
        const uint64_t vmask = __ballot(valid);
        // adjacency row of this lane's candidate (:1146-1160)
        uint64_t adj = 0;
        for (uint64_t m = vmask; m;) {
            const int j = __builtin_ctzll(m); m &= m - 1;
            const float ddx = lane_f(x, j) - x, ddy = lane_f(y, j) - y;
            if (ddx * ddx + ddy * ddy < 0.25f) adj |= 1ull << j;
        }
        if (!valid) adj = 0;

        uint64_t unl = vmask;
        int ncl = 0, max_id = 0;
        float r0 = 0.0f, r1 = 0.0f, r2 = 0.0f, r3 = 0.0f, r4 = 0.0f;         // row (lane) of mvn
        while (unl) {                                                         // :1162-1173, seeds in candidate order
            const int i = __builtin_ctzll(unl);
            ncl++;
            const uint64_t row_i = lane_u64(adj, i);
            uint64_t comp = ((row_i >> i) & 1) ? (1ull << i) : 0;             // a NaN candidate never labels itself
            unl &= ~(1ull << i);
            uint64_t frontier = 1ull << i;
            while (frontier) {
                const bool joins = ((adj & frontier) != 0) && ((unl >> lane) & 1);
                const uint64_t nf = __ballot(joins);
                comp |= nf; unl &= ~nf; frontier = nf;
            }
            if (comp) max_id = ncl;
            // sums in candidate order (:1075-1086)
            float sx = 0.0f, sy = 0.0f, sxx = 0.0f, syy = 0.0f; int cnt = 0;
            for (uint64_t m = comp; m;) {
                const int j = __builtin_ctzll(m); m &= m - 1;
                const float xj = lane_f(x, j), yj = lane_f(y, j);
                sx += xj; sy += yj; sxx += xj * xj; syy += yj * yj; cnt++;
            }
            if (lane == ncl - 1) {                                            // :1095-1104
                const float fc = (float)cnt;
                r0 = sx / fc; r1 = sy / fc;
                r2 = sxx / fc - r0 * r0; r3 = syy / fc - r1 * r1;
                r4 = fc / (float)a.ndp;
            }
        }
        if (lane == 0) { a.nclus[g] = max_id; if (max_id > 0) atomicMax(a.kmax_seen, max_id); }
        if (max_id > a.Kmax) max_id = 0;                                     // caller gets MIMC3_ECAP; rows left zero
        for (int r = lane; r < a.Kmax; r += 64) {
            float *o = a.mvn + ((size_t)g * a.Kmax + r) * 5;
            const bool live = (r == lane) && (r < max_id);
            o[0] = live ? r0 : 0.0f; o[1] = live ? r1 : 0.0f; o[2] = live ? r2 : 0.0f;
            o[3] = live ? r3 : 0.0f; o[4] = live ? r4 : 0.0f;
        }
    }
}

__global__ __launch_bounds__(256) void dpf0_kernel(const float *__restrict__ mvn, const int32_t *__restrict__ nclus, int32_t N,
                                                   int32_t Kmax, float min_ratio, int32_t *__restrict__ dpf)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= N) return;
    int id = -1;
    const int n = nclus[g];
    for (int c = 0; c < n; c++)
        if (mvn[((size_t)g * Kmax + c) * 5 + 4] > min_ratio) { id = c; break; }      // :1243-1252
    dpf[g] = id;
}

// cluster map -> the five output planes (mimc2_postprocess :937-970 / convert_dpf_to_vxy_exy_qual :2498-2515)
__global__ __launch_bounds__(256) void gather_kernel(const int32_t *__restrict__ dpf, const float *__restrict__ mvn, int32_t N,
                                                     int32_t Kmax, float *__restrict__ out5)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= N) return;
    const int id = dpf[g];
    const float nanv = __builtin_nanf("");
#pragma unroll
    for (int k = 0; k < 5; k++) out5[(size_t)k * N + g] = id >= 0 ? mvn[((size_t)g * Kmax + id) * 5 + k] : nanv;
}

__device__ __forceinline__ float sqrt_f(float s) { return (float)sqrt((double)s); }  // C: sqrt(float) is the f64 sqrt

// Per-point terms of get_dpf1's neighbour table (:1421-1448) that depend only on the neighbour h itself:
//   v5 = |a-priori(h)|, w2 = 1/(1+expf(5-v5)) are fixed; v4 = |dp(h)|, v3 = v4/|a-priori(h)| and v6 = noi(h) change
//   only when h receives a value.  They are kept per point instead of being recomputed per (g,k) pair:
//   field plane P[h] = (dx, dy, noi, v3)  -- ping-ponged, one 16-byte gather per neighbour
//   rec[h]          = (v4, v5, w2, -)     -- in place
// A sweep is latency bound (the whole grid is resident at once, so a launch lasts as long as one thread's chain
// of dependent loads): the neighbour offsets sit in LDS and the gathers are unconditional so that the compiler
// keeps several in flight.
__device__ __forceinline__ double apriori_mag(const Dpf1Args &a, int h, float &v5)
{
    const float a0 = (float)(a.xyuvav[6 * (size_t)h + 4]) * a.factor;
    const float a1 = -(float)(a.xyuvav[6 * (size_t)h + 5]) * a.factor;
    const float aa = a0 * a0 + a1 * a1;
    v5 = sqrt_f(aa);
    return sqrt((double)aa);
}

__global__ __launch_bounds__(256) void dpf1_init(Dpf1Args a)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) {
        a.state[kD1ThresNum] = a.nn - 1;
        a.state[kD1Done] = (a.nn - 1 < 3) ? 1 : 0;
        a.state[kD1Sweeps] = 0; a.state[kD1Tally] = 0; a.state[kD1Tally + 1] = 0;
    }
    if (i < a.nn) {
        const int ou = a.ruv[2 * i], ov = a.ruv[2 * i + 1];
        const float d0 = (float)ou, d1 = (float)ov;
        a.ktab[i] = make_int4(ou, ov, __float_as_int(sqrt_f(d0 * d0 + d1 * d1)), 0);   // :1466
    }
    if (i >= a.N) return;
    const float nanv = __builtin_nanf("");
    const int id = a.dpf[i];
    float x = nanv, y = nanv;
    if (id >= 0) { x = a.mvn[((size_t)i * a.Kmax + id) * 5]; y = a.mvn[((size_t)i * a.Kmax + id) * 5 + 1]; }   // :1352-1365
    float v5;
    const double sq = apriori_mag(a, i, v5);
    const float v4 = sqrt_f(x * x + y * y);
    const float e = (float)exp((double)(-v5 + 5.0f));                        // expf, see the header note
    a.plane[0][i] = make_float4(x, y, 1.0f, (float)((double)v4 / sq));
    a.rec[i] = make_float4(v4, v5, 1.0f / (1.0f + e) / 1.0f, 0.0f);
}

// One Jacobi sweep (:1399-1547) in ONE kernel: the field is ping-ponged (sweep s reads plane s&1 and writes the
// other one for every point), so the reference's "compute into buffers, then commit" (:1533-1545) needs no second
// pass; the block that finishes last runs the level control of :1383-1621 for the next launch.
__device__ __forceinline__ bool dpf1_point(const Dpf1Args &a, const float4 *__restrict__ src, const int4 *kt, int g, int thres_num,
                                           float4 &out)
{
    const float tw = a.thres_weight;
    const int cv = g / a.dimx, cu = g - cv * a.dimx;
    const float dpe0 = (float)(a.xyuvav[6 * (size_t)g + 4] * (double)a.factor);     // :1411-1413
    const float dpe1 = (float)(-a.xyuvav[6 * (size_t)g + 5] * (double)a.factor);
    const float mag_dpe = sqrt_f(dpe0 * dpe0 + dpe1 * dpe1);

    // pass 1: count (:1452) and the extremes of |dp|/|a-priori| among the direction-weighted neighbours (:1458-1489).
    // Gathers are issued eight at a time ahead of their use: a sweep lasts as long as one thread's chain of loads.
    constexpr int B = 8;
    int num = 0, id_max = 0, id_min = 0;
    float w_min = 1E+37f, w_max = -1E+37f;
    for (int k0 = 0; k0 < a.nn; k0 += B) {
        int4 t[B]; float4 f[B];
#pragma unroll
        for (int j = 0; j < B; j++) {
            t[j] = kt[min(k0 + j, a.nn - 1)];
            const int u = cu + t[j].x, w = cv + t[j].y;
            const bool in = (k0 + j < a.nn) & (u >= 0) & (u < a.dimx) & (w >= 0) & (w < a.dimy);
            f[j] = src[in ? w * a.dimx + u : g];                             // g itself is NaN: counts as absent
        }
#pragma unroll
        for (int j = 0; j < B; j++) {
            if (isnan(f[j].x + f[j].y)) continue;
            float wc = (dpe0 * (float)t[j].x + dpe1 * (float)t[j].y) / (mag_dpe * __int_as_float(t[j].z));
            wc = wc > 0 ? wc : -wc;
            if (wc >= tw) {
                if (f[j].w > w_max) { w_max = f[j].w; id_max = num; }
                if (f[j].w < w_min) { w_min = f[j].w; id_min = num; }
            }
            num++;
        }
    }
    if (num < thres_num) return false;

    // pass 2: the weighted sums, neighbours in the same order (:1497-1510)
    float s_w = 0.0f, s_wdp = 0.0f, s_wdpe = 0.0f, s_noi = 0.0f;
    int i = 0;
    for (int k0 = 0; k0 < a.nn; k0 += B) {
        int4 t[B]; float4 f[B], r[B];
#pragma unroll
        for (int j = 0; j < B; j++) {
            t[j] = kt[min(k0 + j, a.nn - 1)];
            const int u = cu + t[j].x, w = cv + t[j].y;
            const bool in = (k0 + j < a.nn) & (u >= 0) & (u < a.dimx) & (w >= 0) & (w < a.dimy);
            const int h = in ? w * a.dimx + u : g;
            f[j] = src[h];
            r[j] = a.rec[h];                                                 // v4, v5, w2
        }
#pragma unroll
        for (int j = 0; j < B; j++) {
            if (isnan(f[j].x + f[j].y)) continue;
            float wc = (dpe0 * (float)t[j].x + dpe1 * (float)t[j].y) / (mag_dpe * __int_as_float(t[j].z));
            wc = wc > 0 ? wc : -wc;
            const float v2 = (wc >= tw && i != id_max && i != id_min) ? wc : 0.0f;
            s_w += v2;
            s_wdp += v2 * r[j].z * r[j].x / f[j].z;
            s_wdpe += v2 * r[j].z * r[j].y / f[j].z;
            s_noi += f[j].z;
            i++;
        }
    }
    if (!(s_w >= 1.0f)) return false;                                        // :1512
    const float fm = s_wdp / s_wdpe;
    const float ox = dpe0 * fm, oy = dpe1 * fm;
    float v5;
    const double sq = apriori_mag(a, g, v5);
    const float v4 = sqrt_f(ox * ox + oy * oy);
    out = make_float4(ox, oy, s_noi / (float)num + 1.0f, (float)((double)v4 / sq));
    return true;
}

__global__ __launch_bounds__(kD1SweepThreads) void dpf1_sweep(Dpf1Args a)
{
    extern __shared__ int4 kt[];                                             // [nn] (du, dv, |d|)
    int32_t *st = a.state;
    if (st[kD1Done]) return;
    for (int k = threadIdx.x; k < a.nn; k += blockDim.x) kt[k] = a.ktab[k];
    __syncthreads();
    const int par = st[kD1Sweeps] & 1;
    const float4 *src = a.plane[par];
    float4 *dst = a.plane[par ^ 1];
    const int thres_num = st[kD1ThresNum];
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    bool proc = false, un = false;
    if (g < a.N) {
        float4 f = src[g];
        const bool open = a.nclus[g] != 0;
        if (open && isnan(f.x + f.y)) {                                      // :1406
            float4 nf;
            if (dpf1_point(a, src, kt, g, thres_num, nf)) {
                proc = true;                                                 // :1526 counts it even if the value is NaN
                if (!isnan(nf.x) && !isnan(nf.y)) {                          // :1533-1545
                    f = nf;
                    a.rec[g].x = sqrt_f(nf.x * nf.x + nf.y * nf.y);          // own v4; nobody reads it in this sweep
                } else f.z = nf.z;                                           // noi is written unconditionally (:1524)
            }
        }
        dst[g] = f;
        un = open && (isnan(f.x) || isnan(f.y));                             // :1551-1560
    }
    // one packed atomic per block: ticket (16 bits) | processed (24) | unprocessed (24).  No fence: the last
    // block consumes only the tally itself, the field is consumed by the next launch.
    const unsigned long long np = (unsigned)__syncthreads_count(proc), nu = (unsigned)__syncthreads_count(un);
    if (threadIdx.x != 0) return;
    unsigned long long *tally = reinterpret_cast<unsigned long long *>(st + kD1Tally);
    const unsigned long long mine = 1ull | (np << 16) | (nu << 40);
    const unsigned long long seen = atomicAdd(tally, mine) + mine;
    if ((seen & 0xFFFF) != gridDim.x) return;
    // last block out: control flow of :1383-1621 after this sweep
    const unsigned processed = (unsigned)(seen >> 16) & 0xFFFFFF, unproc = (unsigned)(seen >> 40);
    st[kD1Sweeps] += 1;
    if (processed == 0) {
        // the inner while ends; thres_weight is 0.48 < 0.5 by now, so the middle while ends too: next level
        st[kD1ThresNum] -= 1;
        if (unproc == 0 || st[kD1ThresNum] < 3) st[kD1Done] = 1;
    }
    *tally = 0;
}

// the field after the last sweep -> dx, dy
__global__ __launch_bounds__(256) void dpf1_settle(Dpf1Args a)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= a.N) return;
    const float4 f = a.plane[a.state[kD1Sweeps] & 1][g];
    a.dx[g] = f.x; a.dy[g] = f.y;
}

__global__ __launch_bounds__(256) void dpf1_smooth(Dpf1Args a)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= a.N) return;
    const int cv = g / a.dimx, cu = g - cv * a.dimx;
    if (cv < 1 || cv >= a.dimy - 1 || cu < 1 || cu >= a.dimx - 1) return;
    const float x = a.dx[g], y = a.dy[g];
    if (a.dpf[g] < 0 && !isnan(x + y)) {                                     // :1630-1655
        float nd = 0.0f, sx = 0.0f, sy = 0.0f;
        for (int b = -1; b <= 1; b++)
            for (int c = -1; c <= 1; c++) {
                const int h = (cv + b) * a.dimx + cu + c;
                const float hx = a.dx[h], hy = a.dy[h];
                if (!isnan(hx + hy)) { sx += hx; sy += hy; nd = nd + 1.0f; }
            }
        a.bx[g] = sx / nd; a.by[g] = sy / nd;
    } else { a.bx[g] = x; a.by[g] = y; }
}

__global__ __launch_bounds__(256) void dpf1_snap(Dpf1Args a)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= a.N) return;
    const int cv = g / a.dimx, cu = g - cv * a.dimx;
    float x = a.dx[g], y = a.dy[g];
    if (cv >= 1 && cv < a.dimy - 1 && cu >= 1 && cu < a.dimx - 1) { x = a.bx[g]; y = a.by[g]; }   // :1668-1676
    const int n = a.nclus[g];
    if (a.dpf[g] < 0 && n != 0) {                                            // :1680-1706
        float best = 1E+37f; int id = 0;
        const float *m = a.mvn + (size_t)g * a.Kmax * 5;
        for (int c = 0; c < n; c++) {
            const float d0 = x - m[5 * c], d1 = y - m[5 * c + 1];
            const float sq = d0 * d0 + d1 * d1;
            if (sq < best) { best = sq; id = c; }
        }
        a.dpf[g] = id; x = m[5 * id]; y = m[5 * id + 1];
    }
    a.dx[g] = x; a.dy[g] = y;
}

inline int64_t al256(int64_t b) { return (b + 255) & ~255LL; }

}  // namespace

hipError_t launch_cluster(ClusterArgs a, hipStream_t stream)
{
    hipError_t e = hipMemsetAsync(a.kmax_seen, 0, sizeof(int32_t), stream);
    if (e != hipSuccess) return e;
    const int blocks = (a.N + kCluPointsPerBlock - 1) / kCluPointsPerBlock;
    const size_t lds = sizeof(float) * (size_t)a.ndp * kPitch;
    hipLaunchKernelGGL(cluster_kernel, dim3(blocks), dim3(kCluThreads), lds, stream, a);
    return hipGetLastError();
}

hipError_t launch_dpf0(const float *mvn, const int32_t *nclus, int32_t N, int32_t Kmax, float min_ratio, int32_t *dpf,
                       hipStream_t stream)
{
    hipLaunchKernelGGL(dpf0_kernel, dim3((N + 255) / 256), dim3(256), 0, stream, mvn, nclus, N, Kmax, min_ratio, dpf);
    return hipGetLastError();
}

hipError_t launch_gather(const int32_t *dpf, const float *mvn, int32_t N, int32_t Kmax, float *out5, hipStream_t stream)
{
    hipLaunchKernelGGL(gather_kernel, dim3((N + 255) / 256), dim3(256), 0, stream, dpf, mvn, N, Kmax, out5);
    return hipGetLastError();
}

int64_t dpf1_workspace_bytes(int32_t n) { return 2 * al256(4 * (int64_t)n) + 3 * al256(16 * (int64_t)n) + 16 * kD1MaxNeighbours + 256; }

void dpf1_carve(Dpf1Args &a, void *work)
{
    char *b = static_cast<char *>(work);
    const int64_t s4 = al256(4 * (int64_t)a.N), s16 = al256(16 * (int64_t)a.N);
    a.plane[0] = reinterpret_cast<float4 *>(b);
    a.plane[1] = reinterpret_cast<float4 *>(b + s16);
    a.rec = reinterpret_cast<float4 *>(b + 2 * s16);
    b += 3 * s16;
    a.bx = reinterpret_cast<float *>(b);
    a.by = reinterpret_cast<float *>(b + s4);
    b += 2 * s4;
    a.ktab = reinterpret_cast<int4 *>(b);
    a.state = reinterpret_cast<int32_t *>(b + 16 * kD1MaxNeighbours);
}

hipError_t launch_dpf1_init(const Dpf1Args &a, hipStream_t stream)
{
    if (a.nn > kD1MaxNeighbours) return hipErrorInvalidValue;
    hipLaunchKernelGGL(dpf1_init, dim3((a.N + 255) / 256), dim3(256), 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_dpf1_sweeps(const Dpf1Args &a, int count, hipStream_t stream)
{
    const dim3 grid((a.N + kD1SweepThreads - 1) / kD1SweepThreads), block(kD1SweepThreads);
    if (grid.x > 0xFFFF || a.N >= (1 << 24)) return hipErrorInvalidValue;     // tally field widths
    for (int s = 0; s < count; s++) hipLaunchKernelGGL(dpf1_sweep, grid, block, sizeof(int4) * (size_t)a.nn, stream, a);
    return hipGetLastError();
}

hipError_t launch_dpf1_finish(const Dpf1Args &a, hipStream_t stream)
{
    const dim3 grid((a.N + 255) / 256), block(256);
    hipLaunchKernelGGL(dpf1_settle, grid, block, 0, stream, a);
    hipLaunchKernelGGL(dpf1_smooth, grid, block, 0, stream, a);
    hipLaunchKernelGGL(dpf1_snap, grid, block, 0, stream, a);
    return hipGetLastError();
}

}  // namespace mimc3
