// filter_loop_bench.hip — runs generated GROUP loops (filter_asm.inc: the sequential walk over
// an array of balls, same instruction mix and LDS traffic as the sphere trips) alone on
// never-passing data and reports cycles per VALU instruction, per occupancy.
// Build: hipcc --offload-arch=gfx950 -O3 -I raytracer_amd/csrc tools/filter_loop_bench.hip -o tools/filter_loop_bench
//        (-DFA_INC='"other.inc"' to time an experimental variant of the generator's output)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#ifndef FA_INC
#define FA_INC "filter_asm.inc"
#endif
#include FA_INC
typedef float f4raw __attribute__((ext_vector_type(4)));
#define CH 1024

// LOOP: 0 = primary loop (4 rays), 2 / 3 = trace loop with 2 / 3 rays.
template <int LOOP>
__global__ void __launch_bounds__(256) k(const float4 *sph, int reps, float *out)
{
    __shared__ float4 lds[CH + FA_LDS_SLACK];
    __shared__ int cand[16 * 256];
    for (int i = threadIdx.x; i < CH + FA_LDS_SLACK; i += 256) lds[i] = sph[i];
    __syncthreads();
    const float t = 1e-3f * threadIdx.x;
    const f4raw r0 = {0.1f, 0.6f + t, 0.0f, 0.8f}, r1 = {0.2f, 0.6f, t, 0.8f}, r2 = {0.3f, 0.0f, 0.6f + t, 0.8f},
                r3 = {0.4f, t, 0.6f, 0.8f};
    const f4raw lane = {t, 1.0f, -t, 0.5f};
    int hits = 0;
    for (int r = 0; r < reps; r++) {
        int i = 0;
        unsigned addr = (unsigned)(unsigned long long)(const __attribute__((address_space(3))) char *)lds;
        const unsigned q0 = (unsigned)(unsigned long long)(const __attribute__((address_space(3))) char *)cand + 4u * threadIdx.x;
        unsigned qaddr = q0;
        // 16 trips per call: one queue slot per trip, as in the kernels
        for (int part = 0; part < CH / 128; part++) {
            const int cnt = (part + 1) * 128;
            qaddr = q0;
            if constexpr (LOOP == 0)
                asm volatile(FA_PGROUP4_ASM
                             : [i] "+s"(i), [addr] "+v"(addr), [qaddr] "+v"(qaddr)
                             : FA_DN0(r0), FA_DN1(r1), FA_DN2(r2), FA_DN3(r3), [cnt] "s"(cnt)
                             : FA_PGROUP4_CLOBBERS);
            if constexpr (LOOP == 2)
                asm volatile(FA_GROUP2_ASM
                             : [i] "+s"(i), [addr] "+v"(addr), [qaddr] "+v"(qaddr)
                             : FA_DN0(r0), FA_DN1(r1), FA_LANE(lane), [cnt] "s"(cnt)
                             : FA_GROUP2_CLOBBERS);
            if constexpr (LOOP == 3)
                asm volatile(FA_GROUP3_ASM
                             : [i] "+s"(i), [addr] "+v"(addr), [qaddr] "+v"(qaddr)
                             : FA_DN0(r0), FA_DN1(r1), FA_DN2(r2), FA_LANE(lane), [cnt] "s"(cnt)
                             : FA_GROUP3_CLOBBERS);
            hits += (int)(qaddr - q0);
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = (float)hits;
}

template <int LOOP>
void run(const char *what, int valu_per_trip, int blocks, int reps, const float4 *dsph, float *dout)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<LOOP>, dim3(blocks), dim3(256), 0, 0, dsph, reps, dout);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<LOOP>, dim3(blocks), dim3(256), 0, 0, dsph, reps, dout);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_wave = (double)reps * (CH / 8) * valu_per_trip;
    const double waves_per_simd = blocks * 4.0 / 1024.0;
    printf("%-10s blocks %5d reps %4d  %8.3f ms  %.2f cycles per VALU instr per SIMD @2.4GHz\n", what, blocks, reps,
           ms, ms * 1e-3 * 2.4e9 / (instr_per_wave * waves_per_simd));
}

int main()
{
    std::vector<float4> sph(CH + FA_LDS_SLACK);
    // w = -inf: neither loop ever passes such an entry.
    for (size_t i = 0; i < sph.size(); i++) sph[i] = make_float4(1.f + 0.001f * i, -2.f, 3.f, -INFINITY);
    float4 *dsph;
    float *dout;
    (void)hipMalloc(&dsph, sph.size() * sizeof(float4));
    (void)hipMalloc(&dout, 8192 * 256 * sizeof(float));
    (void)hipMemcpy(dsph, sph.data(), sph.size() * sizeof(float4), hipMemcpyHostToDevice);
    const int blocks[] = {512, 768, 1024, 2048, 3072, 4096};
    for (int b : blocks) {
        for (int reps : {80}) {
            run<0>("pgroup4", FA_PGROUP4_VALU, b, reps, dsph, dout);
            run<2>("group2", FA_GROUP2_VALU, b, reps, dsph, dout);
            run<3>("group3", FA_GROUP3_VALU, b, reps, dsph, dout);
        }
    }
    return 0;
}
