// Probe: the guarded fast finish of match_mx_kernel.hip -- (float)(num / sqrt(va * vb)) from v_rsq_f64 + one Newton step --
// against the reference's operations (correctly rounded f64 sqrt and division), on random integer sums of the ranges the
// matcher produces.  Reports: the largest relative error of v_rsq_f64 and of the refined reciprocal root, how many cells the
// guard sends to the exact path, how many UNGUARDED cells differ (must be 0), and how close to an f32 rounding boundary the
// nearest cell whose two results differ lay (the margin the guard of 2^13 f64 ulps has).
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off mx_finish.hip -o mx_finish
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)

struct Res { unsigned long long n, amb, bad, differ; unsigned int max_dist; double max_e0, max_e1; };

__device__ __forceinline__ uint64_t rnd(uint64_t &s) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }

__global__ void finish_probe(int iters, int npx, Res *out)
{
    uint64_t s = 0x9E3779B97F4A7C15ull * (blockIdx.x * blockDim.x + threadIdx.x + 1);
    unsigned long long n = 0, amb = 0, bad = 0, differ = 0;
    unsigned int max_dist = 0u;
    double max_e0 = 0, max_e1 = 0;
    for (int it = 0; it < iters; it++) {
        // sums of npx pixels in [0, 255]: pick the moments so that both variances are non-negative integers of the matcher's size
        const int nn = npx - (int)(rnd(s) % 70);
        const double dn = (double)nn;
        const int sx = (int)(rnd(s) % (255u * nn)), sy = (int)(rnd(s) % (255u * nn));
        const double mx = (double)sx / nn, my = (double)sy / nn;
        // sxx >= sx^2 / n etc.; spread = up to 2x the minimum
        const long long sxx_min = (long long)(mx * sx) + 1, syy_min = (long long)(my * sy) + 1;
        const int sxx = (int)(sxx_min + (long long)(rnd(s) % (unsigned long long)(sxx_min / 4 + 1000)));
        const int syy = (int)(syy_min + (long long)(rnd(s) % (unsigned long long)(syy_min / 4 + 1000)));
        const double va = dn * (double)sxx - (double)sx * (double)sx, vb = dn * (double)syy - (double)sy * (double)sy;
        if (!(va > 0) || !(vb > 0)) continue;
        // |num| <= sqrt(va vb)
        const double lim = sqrt(va * vb);
        const double num = (double)(long long)((((double)(rnd(s) >> 11) / 9007199254740992.0) * 2.0 - 1.0) * lim);
        const double P = va * vb;
        const float exact = (float)(num / sqrt(P));
        const double r = __builtin_amdgcn_rsq(P);
        const double g = P * r;
        const double e2 = __builtin_fma(-r, g, 1.0);
        const double r1 = __builtin_fma(0.5 * r, e2, r);
        const double q = num * r1;
        const float fast = (float)q;
        const uint32_t lowbits = (uint32_t)__double2loint(q) & 0x1fffffffu;
        const uint32_t low = lowbits - (0x10000000u - 0x2000u);
        const bool a = !(P > 0.0) || low <= 0x4000u;
        const double truth = 1.0 / sqrt(P);
        const double e0 = fabs(r / truth - 1.0), e1 = fabs(r1 / truth - 1.0);
        max_e0 = e0 > max_e0 ? e0 : max_e0; max_e1 = e1 > max_e1 ? e1 : max_e1;
        n++;
        if (a) amb++;
        if (__float_as_uint(fast) != __float_as_uint(exact)) {
            differ++;
            const uint32_t d = lowbits > 0x10000000u ? lowbits - 0x10000000u : 0x10000000u - lowbits;
            max_dist = d > max_dist ? d : max_dist;
            if (!a) bad++;
        }
    }
    atomicAdd(&out->n, n); atomicAdd(&out->amb, amb); atomicAdd(&out->bad, bad); atomicAdd(&out->differ, differ);
    atomicMax(&out->max_dist, max_dist);
    // (max of doubles through integer atomics: non-negative doubles order like their bit patterns)
    atomicMax(reinterpret_cast<unsigned long long *>(&out->max_e0), (unsigned long long)__double_as_longlong(max_e0));
    atomicMax(reinterpret_cast<unsigned long long *>(&out->max_e1), (unsigned long long)__double_as_longlong(max_e1));
}

int main(int argc, char **argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 4096;
    Res *d; CK(hipMalloc(&d, sizeof(Res)));
    for (int npx : {225, 961, 1089, 3721, 4225, 6561}) {
        Res h{};
        CK(hipMemcpy(d, &h, sizeof(Res), hipMemcpyHostToDevice));
        hipLaunchKernelGGL(finish_probe, dim3(4096), dim3(256), 0, 0, iters, npx, d);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(&h, d, sizeof(Res), hipMemcpyDeviceToHost));
        printf("chip of %4d px: %llu cells, guard took %llu (2^%.1f), fast != exact on %llu cells, of them UNGUARDED %llu; "
               "farthest differing cell %u f64 ulps from a boundary (guard 8192); rel. error rsq %.3g (2^%.1f), refined %.3g (2^%.1f)\n",
               npx, h.n, h.amb, h.n ? log2((double)h.amb / h.n) : 0.0, h.differ, h.bad, h.max_dist,
               h.max_e0, log2(h.max_e0), h.max_e1, log2(h.max_e1));
    }
    return 0;
}
