// valu_bench.hip — FP32 vector issue-rate microbenchmark for gfx950.
// Measures how many FP32 operations per clock per CU plain and packed VALU
// instructions sustain at 1..8 waves per SIMD; this decides whether the render
// kernels should pair rays into v_pk_* instructions.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/valu_bench.hip -o tools/valu_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float float2v __attribute__((ext_vector_type(2)));

#define ITERS 4096
#define NACC 8

template <int MODE>
__global__ void __launch_bounds__(256) k(float *out, float a, float b)
{
    float acc[NACC];
    float2v acc2[NACC];
#pragma unroll
    for (int i = 0; i < NACC; i++) {
        acc[i] = (float)threadIdx.x * 1e-3f + i;
        acc2[i] = float2v{acc[i], acc[i] + 0.5f};
    }
    float2v a2 = {a, a}, b2 = {b, b};
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < NACC; i++) {
            if (MODE == 0) acc[i] = __builtin_fmaf(acc[i], a, b);              // v_fma_f32
            if (MODE == 1) acc[i] = acc[i] * a + b;                            // v_mul + v_add (contract off)
            if (MODE == 2) acc2[i] = __builtin_elementwise_fma(acc2[i], a2, b2); // v_pk_fma_f32
            if (MODE == 3) acc2[i] = acc2[i] * a2 + b2;                        // v_pk_mul + v_pk_add
            if (MODE == 4) acc[i] = acc[i] * a;                                // v_mul_f32 only
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; i++) s += acc[i] + acc2[i].x + acc2[i].y;
    if (s == 12345.678f) out[0] = s;
}

template <int MODE>
double run(int blocks, float *d)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, 1e-6f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; r++) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, 1e-6f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / 5.0;
}

int main()
{
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    printf("device %s CUs %d clock %d kHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate);
    float *d;
    hipMalloc(&d, 4096);
    const char *names[] = {"v_fma_f32", "v_mul+v_add", "v_pk_fma_f32", "v_pk_mul+v_pk_add", "v_mul_f32"};
    const double flop_per_iter[] = {2, 2, 4, 4, 1};      // per lane per accumulator
    const double instr_per_iter[] = {1, 2, 1, 2, 1};
    for (int wps = 1; wps <= 8; wps *= 2) {
        int blocks = p.multiProcessorCount * wps;  // 4 waves per block -> wps waves per SIMD
        for (int m = 0; m < 5; m++) {
            double ms = m == 0 ? run<0>(blocks, d) : m == 1 ? run<1>(blocks, d) : m == 2 ? run<2>(blocks, d)
                        : m == 3 ? run<3>(blocks, d) : run<4>(blocks, d);
            double lanes = (double)blocks * 256;
            double flops = lanes * ITERS * NACC * flop_per_iter[m];
            double instr = (double)blocks * 4 * ITERS * NACC * instr_per_iter[m];
            printf("waves/SIMD %d  %-18s %8.3f ms  %7.2f TFLOP/s  %6.2f wave-instr/ns/chip  cyc/instr/SIMD@2.4GHz %.2f\n",
                   wps, names[m], ms, flops / ms / 1e9, instr / ms / 1e6,
                   (ms * 1e-3 * 2.4e9) / (instr / (p.multiProcessorCount * 4.0)));
        }
    }
    return 0;
}
