// prio_bench.hip — does s_setprio change how a SIMD's VALU issue slots are shared on gfx950?
// One workgroup of 16 waves (four per SIMD) runs the same loop of independent FMAs; in the
// second launch the waves 12..15 (the youngest of each SIMD) raise their priority first.  Prints cycles per wave with the
// SIMD it ran on (HW_ID bits 4..5).
// Build: hipcc --offload-arch=gfx950 -O3 tools/prio_bench.hip -o tools/prio_bench
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void __launch_bounds__(1024) k(int iters, int raise, unsigned long long *cycles, unsigned *simd)
{
    const int wave = threadIdx.x >> 6;
    if (raise && wave >= 12) __builtin_amdgcn_s_setprio(3);
    float a = threadIdx.x * 1e-3f, b = 1.0001f, c = 0.5f, d = 0.25f, e = 0.125f, f = 2.0f, g = 3.0f, h = 4.0f;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) {
        asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                     "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                     : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h)
                     : "v"(0.999f), "v"(1e-6f));
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    __builtin_amdgcn_s_setprio(0);
    unsigned hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    if ((threadIdx.x & 63) == 0) {
        cycles[wave] = t1 - t0;
        simd[wave] = (hw >> 4) & 3u;
    }
    if (a + b + c + d + e + f + g + h == 12345.0f) cycles[0] = 0;
}

int main()
{
    unsigned long long *cyc;
    unsigned *simd;
    hipMalloc((void **)&cyc, 16 * sizeof *cyc);
    hipMalloc((void **)&simd, 16 * sizeof *simd);
    for (int raise = 0; raise < 2; raise++)
        for (int rep = 0; rep < 2; rep++) {
            hipLaunchKernelGGL(k, dim3(1), dim3(1024), 0, 0, 20000, raise, cyc, simd);
            unsigned long long hc[16];
            unsigned hs[16];
            hipMemcpy(hc, cyc, sizeof hc, hipMemcpyDeviceToHost);
            hipMemcpy(hs, simd, sizeof hs, hipMemcpyDeviceToHost);
            printf("waves 12..15 %s:", raise ? "at priority 3" : "at priority 0");
            for (int w = 0; w < 16; w++) printf(" w%d(simd%u)=%.2f", w, hs[w], (double)hc[w] / (20000.0 * 8.0));
            printf("  [clock ticks per FMA; 100 MHz counter]\n");
        }
    return 0;
}
