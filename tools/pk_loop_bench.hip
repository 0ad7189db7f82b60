// pk_loop_bench.hip — times the sphere-filter inner loop in its scalar and packed forms on
// synthetic data (no sphere ever passes), to see what the loop costs per ray-sphere test
// without the rest of the render kernels around it.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize tools/pk_loop_bench.hip -o tools/pk_loop_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(4))) f4 *cf4p;
__device__ __forceinline__ v2 splat(float s) { return v2{s, s}; }

// MODE 0: scalar, 1 item per lane, v_cmp per test (the v1 kernel's loop)
// MODE 1: packed, 2 items per lane, min/max3 flag
// MODE 2: packed, 2 items per lane, flag from v only (no x<=0 term): max3 of v halves
// MODE 3: scalar, 1 item per lane, min/max flag instead of v_cmp
// MODE 4: packed arithmetic only (flag from one half) - lower bound of the packed form
template <int MODE, int B>
__global__ void __launch_bounds__(256) k(const float4 *sphp, int n, const float *in, float *out)
{
    const int t = threadIdx.x + blockIdx.x * 256;
    cf4p sph = (cf4p)(unsigned long long)sphp;
    const v2 dx = {in[t * 8 + 0], in[t * 8 + 1]}, dy = {in[t * 8 + 2], in[t * 8 + 3]},
             dz = {in[t * 8 + 4], in[t * 8 + 5]}, a = {in[t * 8 + 6], in[t * 8 + 7]};
    float acc = 0.f;
    for (int i = 0; i < n; i += B) {
        float flag = -1.0f;
        bool any = false;
#pragma unroll
        for (int u = 0; u < B; u++) {
            const f4 s = sph[i + u];
            if (MODE == 0 || MODE == 3) {
                const float x = (dx[0] * s.x + dy[0] * s.y) + dz[0] * s.z;
                if (MODE == 0) {
                    const float m = x * fabsf(x);
                    any = any || ((-m) >= a[0] * s.w);
                } else {
                    const float v = x * x - a[0] * s.w;
                    flag = fmaxf(flag, fminf(v, -x));
                }
            } else {
                const v2 x = (dx * splat(s.x) + dy * splat(s.y)) + dz * splat(s.z);
                const v2 v = x * x - a * splat(s.w);
                if (MODE == 1) flag = fmaxf(flag, fmaxf(fminf(v[0], -x[0]), fminf(v[1], -x[1])));
                if (MODE == 2) flag = fmaxf(flag, fmaxf(v[0], v[1]));
                if (MODE == 4) flag = fmaxf(flag, v[0]);
            }
        }
        if (MODE == 0 ? any : (flag >= 0.0f)) acc += 1.0f;
    }
    out[t] = acc;
}

template <int MODE, int B>
double run(int blocks, const float4 *sph, int n, const float *in, float *out)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, B>), dim3(blocks), dim3(256), 0, 0, sph, n, in, out);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int r = 0; r < 5; r++) hipLaunchKernelGGL((k<MODE, B>), dim3(blocks), dim3(256), 0, 0, sph, n, in, out);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms / 5.0;
}

int main()
{
    const int n = 1000, items = 3840 * 2160;
    std::vector<float4> sph(n);
    for (int i = 0; i < n; i++) sph[i] = make_float4(0.01f * i, 0.5f, -0.25f, 1e30f);  // c huge: never passes
    std::vector<float> in((size_t)items * 8);
    for (size_t i = 0; i < in.size(); i++) in[i] = 0.001f * (float)(i % 977) - 0.4f;
    float4 *dsph;
    float *din, *dout;
    (void)hipMalloc(&dsph, n * sizeof(float4));
    (void)hipMalloc(&din, in.size() * sizeof(float));
    (void)hipMalloc(&dout, (size_t)items * sizeof(float));
    (void)hipMemcpy(dsph, sph.data(), n * sizeof(float4), hipMemcpyHostToDevice);
    (void)hipMemcpy(din, in.data(), in.size() * sizeof(float), hipMemcpyHostToDevice);
    const double tests = (double)items * n;
    struct { const char *name; double ms; int per_lane; } r[] = {
        {"scalar  v_cmp      B=4", run<0, 4>(items / 256, dsph, n, din, dout), 1},
        {"scalar  min/max    B=4", run<3, 4>(items / 256, dsph, n, din, dout), 1},
        {"packed  min/max3   B=4", run<1, 4>(items / 512, dsph, n, din, dout), 2},
        {"packed  min/max3   B=8", run<1, 8>(items / 512, dsph, n, din, dout), 2},
        {"packed  v-only     B=4", run<2, 4>(items / 512, dsph, n, din, dout), 2},
        {"packed  arith only B=4", run<4, 4>(items / 512, dsph, n, din, dout), 2},
    };
    for (auto &e : r)
        printf("%-26s %7.3f ms  %6.2f Gtests/s  %5.2f cycles/test/SIMD-lane-group@2.4GHz\n", e.name, e.ms,
               tests / e.ms / 1e6, e.ms * 1e-3 * 2.4e9 * 1024 / (tests / 64.0));
    return 0;
}
