// qm_kernel.hip -- QM pseudo-smoothing update for gfx950 (MI355X).
//
// Replaces get_dpf_pseudosmoothing (MIMC_module.c:1986-2312) with quadfit2 (:2314-2409) and
// GMA_double_inv (:2430-2496).  The reference is a serial Jacobi iteration: every sweep reads the
// previous dpf_dx/dpf_dy, writes candidates to buffers and commits them afterwards (:2218-2233),
// so all masked grid points of one sweep are independent -> one thread per grid point.
//
// Bit-parity strategy: each thread runs the reference's f64 arithmetic for its point in the
// reference's order (neighbours in ruv order, normal-matrix terms as (A_r*w)*A_c, Gauss-Jordan
// without pivoting in the same elimination order), compiled with -ffp-contract=off.  The only
// non-identical primitive is exp(): device libm vs glibc may differ in the last ulp, which moves
// the fitted value by ~1e-16 relative and can only change the result at an exact nearest-cluster
// tie.  Outputs are cluster ids and copied f32 cluster means, so they are
// otherwise bit-identical.
//
// Per sweep, TWO launches (all device side, early-out on a device flag, no host sync):
//   qm_sweep        : fit + nearest candidate per masked point, mark neighbours for the next sweep
//   qm_finish_sweep : Jacobi commit of the buffers; next mask vs every earlier mask of the stack ("fluctuation",
//                     :2237-2261); the block that retires last runs the termination logic of the while loop (:2077, :2263-2286)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "qm_kernel.h"

namespace mimc3 {

__device__ __forceinline__ void inv6(double (&b)[6][6], double (&I)[6][6])
{
#pragma unroll
    for (int i = 0; i < 6; i++)
#pragma unroll
        for (int j = 0; j < 6; j++) I[i][j] = (i == j) ? 1.0 : 0.0;
#pragma unroll
    for (int p = 0; p < 5; p++) {
        const double pivot = b[p][p];
#pragma unroll
        for (int r = p + 1; r < 6; r++) {
            const double coeff = b[r][p] / pivot;
#pragma unroll
            for (int c = 0; c < 6; c++) { b[r][c] -= b[p][c] * coeff; I[r][c] -= I[p][c] * coeff; }
        }
    }
#pragma unroll
    for (int p = 5; p >= 0; p--) {
        const double pivot = b[p][p];
#pragma unroll
        for (int r = p - 1; r >= 0; r--) {
            const double coeff = b[r][p] / pivot;
#pragma unroll
            for (int c = 5; c >= 0; c--) { b[r][c] -= b[p][c] * coeff; I[r][c] -= I[p][c] * coeff; }
        }
    }
#pragma unroll
    for (int i = 0; i < 6; i++)
#pragma unroll
        for (int j = 0; j < 6; j++) I[i][j] /= b[i][i];
}

__global__ __launch_bounds__(kQmThreads) void qm_init(QmArgs a)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) {
        a.flags[kQmAny] = 0; a.flags[kQmDone] = 0; a.flags[kQmSweeps] = 0;
        a.flags[kQmSkipped] = 0; a.flags[kQmTicket] = 0;
    }
    if (i >= a.N) return;
    const int id = a.dpf[i];
    unsigned char m = 0;
    if (id >= 0) m = (a.mvn[((size_t)i * a.Kmax + id) * 5 + 4] >= 0.6) ? 0 : 1;   // :2035-2049 (f32 vs f64 0.6)
    a.stack[i] = m;            // stack[0] = initial mask
    a.mask[0][i] = m;          // current mask of sweep 1
    a.mask[1][i] = 0;
    const float nanv = __builtin_nanf("");
    a.bx[i] = nanv; a.by[i] = nanv; a.bid[i] = -1;
}

__global__ __launch_bounds__(kQmThreads) void qm_sweep(QmArgs a, int sweep)
{
    if (a.flags[kQmDone]) return;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= a.N) return;
    const unsigned char *cur = a.mask[(sweep - 1) & 1];
    unsigned char *next = a.mask[sweep & 1];
    if (!cur[idx]) return;
    const int v = idx / a.dimx, u = idx - v * a.dimx;

    const double eig0 = 1500.0 / 300.0, eig1 = eig0 / 3.0;
    const double vx = a.xyuvav[6 * (size_t)idx + 4], vy = a.xyuvav[6 * (size_t)idx + 5];
    const double den = (eig0 * eig1) * (vx * vx + vy * vy);
    const double itm0 = (eig1 * vx * vx + eig0 * vy * vy) / den;        // :2144-2146
    const double itm1 = ((eig0 - eig1) * vx * vy) / den;
    const double itm3 = (eig1 * vy * vy + eig0 * vx * vx) / den;

    double Nm[6][6], tb[2][6];
#pragma unroll
    for (int r = 0; r < 6; r++) {
        tb[0][r] = 0.0; tb[1][r] = 0.0;
#pragma unroll
        for (int c = 0; c < 6; c++) Nm[r][c] = 0.0;
    }
    int n = 0;
    for (int k = 0; k < a.nn; k++) {                                     // :2108-2126, ruv order
        const int ox = a.ruv[2 * k], oy = a.ruv[2 * k + 1];
        const int uu = u + ox, vv = v + oy;
        if (uu < 0 || uu >= a.dimx || vv < 0 || vv >= a.dimy) continue;
        const float fx = a.dx[vv * a.dimx + uu], fy = a.dy[vv * a.dimx + uu];
        if (fx != fx || fy != fy) continue;
        n++;
        const double x = (double)ox, y = (double)oy;
        const double w = exp(-(itm0 * ox * ox + 2 * itm1 * ox * oy + itm3 * oy * oy));   // :2151-2153
        const double A[6] = { x * x, x * y, y * y, x, y, 1.0 };
        const double z0 = (double)fx, z1 = (double)fy;
#pragma unroll
        for (int r = 0; r < 6; r++) {
            const double aw = A[r] * w;
#pragma unroll
            for (int c = 0; c < 6; c++) Nm[r][c] += aw * A[c];           // :2351
            tb[0][r] += aw * z0;                                         // :2369
            tb[1][r] += aw * z1;
        }
    }
    if (n < 10) return;                                                  // :2131
    double IN[6][6];
    inv6(Nm, IN);
    double fit[2];
#pragma unroll
    for (int oc = 0; oc < 2; oc++) {
        double coef[6];
#pragma unroll
        for (int r = 0; r < 6; r++) {
            double acc = 0.0;
#pragma unroll
            for (int c = 0; c < 6; c++) acc += IN[r][c] * tb[oc][c];     // :2382
            coef[r] = acc;
        }
        double val = 0.0;                                                // :2397-2401 at xyi = (0,0)
#pragma unroll
        for (int c = 0; c < 5; c++) val += 0.0 * coef[c];
        val += 1.0 * coef[5];
        fit[oc] = val;
    }
    const int id = a.dpf[idx], nc = a.nclus[idx];
    const float *cl = a.mvn + (size_t)idx * a.Kmax * 5;
    double dmin = 1E+37;
    int best = -1;
    for (int c = 0; c < nc; c++) {                                       // :2167-2180
        const double cu = (double)cl[5 * c], cv = (double)cl[5 * c + 1];
        const double sq = (fit[0] - cu) * (fit[0] - cu) + (fit[1] - cv) * (fit[1] - cv);
        if (sq < dmin) { dmin = sq; best = c; }
    }
    if (best < 0) { atomicAdd(&a.flags[kQmSkipped], 1); return; }        // T7 definition
    const double gu = (double)cl[5 * id], gv = (double)cl[5 * id + 1];
    const double qu = (double)cl[5 * best], qv = (double)cl[5 * best + 1];
    if ((gu - qu) * (gu - qu) + (gv - qv) * (gv - qv) < 0.0001) return;  // :2190
    a.bx[idx] = cl[5 * best]; a.by[idx] = cl[5 * best + 1]; a.bid[idx] = best;
    a.flags[kQmAny] = 1;
    for (int k = 0; k < a.nn; k++) {                                     // :2203-2210
        const int uu = u + a.ruv[2 * k], vv = v + a.ruv[2 * k + 1];
        if (uu < 0 || uu >= a.dimx || vv < 0 || vv >= a.dimy) continue;
        const int j = vv * a.dimx + uu;
        const float fx = a.dx[j], fy = a.dy[j];
        if (fx != fx || fy != fy) continue;
        if (a.stack[j]) next[j] = 1;                                     // stack[0] = initial mask
    }
}

// Second (and last) launch of a sweep: the Jacobi commit (:2218-2233), the comparison of the next mask with every earlier mask
// of the stack ("fluctuation", :2237-2261) and -- by the block that retires last -- the termination logic of the while loop
// (:2077, :2263-2286).  The commit and the compare are per grid point; the decision needs every block's compare result (bit s =
// "the next mask differs from mask s somewhere"): waves OR their bits into LDS, the block publishes its words with device-scope
// stores into its own slots, a ticket counts retired blocks, and the last one ORs the slots.  Nothing contends on one address
// except the ticket (one atomic per 1,024-thread block) -- the first version let every WAVE OR into one global word and spent
// 40-60 us per sweep in those 3,000 same-address atomics, more than the three kernels it replaced.
__global__ __launch_bounds__(kQmFinishThreads) void qm_finish_sweep(QmArgs a, int sweep)
{
    __shared__ unsigned int sh[kQmMaskWords + 1];
    if (a.flags[kQmDone]) return;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool in = i < a.N;
    const int nw = (sweep + 31) >> 5;
    if (threadIdx.x <= kQmMaskWords) sh[threadIdx.x] = 0u;
    __syncthreads();
    if (in) {
        const int b = a.bid[i];
        if (b >= 0) {                                                    // :2222-2231
            a.dx[i] = a.bx[i]; a.dy[i] = a.by[i]; a.dpf[i] = b;
            const float nanv = __builtin_nanf("");
            a.bx[i] = nanv; a.by[i] = nanv; a.bid[i] = -1;
        }
        a.mask[(sweep - 1) & 1][i] = 0;   // was "current"; becomes "next" of sweep+1
    }
    const unsigned char m = in ? a.mask[sweep & 1][i] : (unsigned char)0;
    uint32_t word = 0u;                                                  // wave-uniform: mismatch bits of stack entries 32w .. 32w+31
    for (int s = 0; s < sweep; s++) {
        const bool ne = in && a.stack[(size_t)s * a.N + i] != m;
        if (__ballot(ne) != 0ull) word |= 1u << (s & 31);
        if ((s & 31) == 31 || s == sweep - 1) {
            if ((threadIdx.x & 63) == 0 && word) atomicOr(&sh[s >> 5], word);
            word = 0u;
        }
    }
    if (in) a.stack[(size_t)sweep * a.N + i] = m;                        // :2272-2284 (harmless if we stop)
    __syncthreads();
    // the block's words -> its own slots (device scope: the last block may sit on another XCD, behind another L2)
    if ((int)threadIdx.x < nw)
        __hip_atomic_store(reinterpret_cast<unsigned int *>(&a.diff[threadIdx.x * gridDim.x + blockIdx.x]), sh[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();                                                     // (workgroup scope: the slot stores above happen-before thread 0's release)
    // the ticket: release publishes this block's slots (and, transitively through the barrier, every thread's), acquire in the
    // block that draws the last ticket makes every other block's slots visible to it -- no hand-written wait counts
    if (threadIdx.x == 0)
        sh[kQmMaskWords] = __hip_atomic_fetch_add(&a.flags[kQmTicket], 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == (int)gridDim.x - 1 ? 1u : 0u;
    __syncthreads();
    if (!sh[kQmMaskWords]) return;
    // ---- the last block: OR of every block's words, then qm_decide
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");                   // every thread of the last block reads other blocks' slots
    __syncthreads();
    if (threadIdx.x < kQmMaskWords) sh[threadIdx.x] = 0u;
    __syncthreads();
    for (int k = threadIdx.x; k < nw * (int)gridDim.x; k += blockDim.x) {
        const unsigned int v = __hip_atomic_load(reinterpret_cast<unsigned int *>(&a.diff[k]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (v) atomicOr(&sh[k / (int)gridDim.x], v);
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    a.flags[kQmTicket] = 0;
    bool fluct = false;
    for (int w = 0; w < nw; w++) {
        const int nbits = (sweep - 32 * w) < 32 ? (sweep - 32 * w) : 32;
        const uint32_t all = nbits >= 32 ? 0xffffffffu : ((1u << nbits) - 1u);
        if ((sh[w] & all) != all) fluct = true;                          // some earlier mask equals the next one everywhere
    }
    if (fluct) { a.flags[kQmSweeps] = sweep - 1; a.flags[kQmDone] = 1; return; }   // :2263-2268 (NOI--)
    a.flags[kQmSweeps] = sweep;
    if (!a.flags[kQmAny] || sweep >= a.max_sweeps) a.flags[kQmDone] = 1;            // :2077
    a.flags[kQmAny] = 0;
}

static inline int64_t pad256(int64_t n) { return (n + 255) & ~255LL; }
static inline int64_t finish_blocks(int32_t n) { return ((int64_t)n + kQmFinishThreads - 1) / kQmFinishThreads; }
// flag words + one slot per finish block and mask word (every sweep rewrites the slots it reads: no clearing)
static inline int64_t head_words(int32_t n, int32_t max_sweeps) { return (kQmFlagWords + ((int64_t)max_sweeps + 32) / 32 * finish_blocks(n) + 63) & ~63LL; }

int64_t qm_workspace_bytes(int32_t n, int32_t max_sweeps)
{
    const int64_t N = pad256(n);
    // bx, by (f32) + bid (i32) + flags/diff words + mask[2][N] + stack[max_sweeps+1][N]
    return 12 * N + 4 * head_words(n, max_sweeps) + 2 * N + N * (int64_t)(max_sweeps + 1);
}

hipError_t launch_qm(QmArgs a, void *work, hipStream_t stream)
{
    if (a.N <= 0) return hipSuccess;
    if (a.max_sweeps >= 32 * kQmMaskWords) return hipErrorInvalidValue;     // (the reference stops at 101 sweeps)
    const int64_t N = pad256(a.N);
    unsigned char *w = static_cast<unsigned char *>(work);
    a.bx = reinterpret_cast<float *>(w); w += 4 * N;
    a.by = reinterpret_cast<float *>(w); w += 4 * N;
    a.bid = reinterpret_cast<int32_t *>(w); w += 4 * N;
    a.flags = reinterpret_cast<int32_t *>(w);
    a.diff = a.flags + kQmFlagWords;
    w += 4 * head_words(a.N, a.max_sweeps);
    a.mask[0] = w; w += N;
    a.mask[1] = w; w += N;
    a.stack = w;
    const int nb = (int)((a.N + kQmThreads - 1) / kQmThreads);
    const int nbf = (int)((a.N + kQmFinishThreads - 1) / kQmFinishThreads);
    const int nbi = (int)((((a.N > a.max_sweeps + 2) ? a.N : a.max_sweeps + 2) + kQmThreads - 1) / kQmThreads);
    hipLaunchKernelGGL(qm_init, dim3(nbi), dim3(kQmThreads), 0, stream, a);
    for (int s = 1; s <= a.max_sweeps; s++) {
        hipLaunchKernelGGL(qm_sweep, dim3(nb), dim3(kQmThreads), 0, stream, a, s);
        hipLaunchKernelGGL(qm_finish_sweep, dim3(nbf), dim3(kQmFinishThreads), 0, stream, a, s);
    }
    return hipGetLastError();
}

}  // namespace mimc3
