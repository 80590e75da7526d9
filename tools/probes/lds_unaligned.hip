// Probe: are byte-misaligned ds_read_b32 supported on gfx950, and at what cost?
// Also times v_dot4_u32_u8 throughput.  Build: hipcc --offload-arch=gfx950 -O3 lds_unaligned.hip -o lds_unaligned
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__global__ void probe(int shift, int iters, uint32_t *out, unsigned long long *cyc)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[8192];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = (unsigned char)(i * 7 + 3);
    __syncthreads();
    const int lane = threadIdx.x;
    uint32_t acc = 0;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const unsigned char *p = lds + ((lane * 4 + k * 264 + shift + it * 4) & 4095);
            uint32_t v;
            asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"((uint32_t)(uintptr_t)p) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            acc += v;
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

__global__ void check(int shift, uint32_t *out)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[1024];
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) lds[i] = (unsigned char)(i * 7 + 3);
    __syncthreads();
    const unsigned char *p = lds + threadIdx.x * 4 + shift;
    uint32_t v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"((uint32_t)(uintptr_t)p) : "memory");
    out[threadIdx.x] = v;
}

int main()
{
    uint32_t *d_out; unsigned long long *d_cyc;
    hipMalloc(&d_out, 1 << 20); hipMalloc(&d_cyc, 8192);
    for (int shift = 0; shift < 4; shift++) {
        hipLaunchKernelGGL(check, dim3(1), dim3(64), 0, 0, shift, d_out);
        std::vector<uint32_t> h(64);
        hipMemcpy(h.data(), d_out, 256, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int l = 0; l < 64; l++) {
            uint32_t want = 0;
            for (int b = 0; b < 4; b++) want |= (uint32_t)(unsigned char)((l * 4 + shift + b) * 7 + 3) << (8 * b);
            if (h[l] != want) bad++;
        }
        printf("shift %d: %s (%d bad) sample got=%08x\n", shift, bad ? "WRONG" : "correct", bad, h[1]);
    }
    for (int shift = 0; shift < 4; shift++) {
        hipLaunchKernelGGL(probe, dim3(256), dim3(256), 0, 0, shift, 200, d_out, d_cyc);
        hipDeviceSynchronize();
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipEventRecord(a);
        hipLaunchKernelGGL(probe, dim3(1024), dim3(256), 0, 0, shift, 2000, d_out, d_cyc);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        unsigned long long c; hipMemcpy(&c, d_cyc, 8, hipMemcpyDeviceToHost);
        printf("shift %d: %.3f ms, block0 cycles per read (dependent) = %.1f\n", shift, ms, (double)c / (2000.0 * 16));
    }
    return 0;
}
