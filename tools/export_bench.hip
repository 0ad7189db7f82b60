// export_bench.hip — what a device -> host copy kernel costs the kernels that run beside it (gfx950).
//
// A "victim" kernel (streams through device memory: reads, writes, or returning atomics) is timed alone and
// beside a copy kernel that writes page-locked host memory over PCIe, for several forms of the copy's stores
// (plain, nontemporal, write-through with sc0 sc1) and several grid sizes.  Prints the victim's duration and the
// copy's rate in each case.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                             \
    do {                                                                                     \
        hipError_t e_ = (x);                                                                 \
        if (e_ != hipSuccess) {                                                              \
            fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__);      \
            exit(1);                                                                         \
        }                                                                                    \
    } while (0)

typedef int word16 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ void __launch_bounds__(256) k_copy(word16 *__restrict__ dst, const word16 *__restrict__ src, size_t words)
{
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < words; i += stride) {
        const word16 v = __builtin_nontemporal_load(src + i);
        if (MODE == 0) dst[i] = v;
        else if (MODE == 1) __builtin_nontemporal_store(v, dst + i);
        else {
            asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 0" ::"v"(dst + i), "v"(v) : "memory");
        }
    }
}

// what: 0 reads (sum), 1 writes, 2 returning atomics on 4096 spread counters
__global__ void __launch_bounds__(256) k_victim(int *buf, size_t n, int what, int *sink)
{
    const size_t stride = (size_t)gridDim.x * 256;
    int acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        if (what == 0) acc += buf[i];
        else if (what == 1) buf[i] = (int)i;
        else acc += atomicAdd(&buf[(i * 64) & 0x3ffff], 1);
    }
    if (acc == 123456789) *sink = acc;
}

int main()
{
    const size_t copy_bytes = (size_t)3840 * 2160 * 4 / 2;   // half a 4K frame
    const size_t victim_ints = (size_t)256 << 20;              // 1 GiB
    int *vbuf, *sink;
    word16 *src, *host;
    CHECK(hipMalloc(&vbuf, victim_ints * 4));
    CHECK(hipMalloc(&sink, 4));
    CHECK(hipMalloc(&src, copy_bytes));
    CHECK(hipHostMalloc(&host, copy_bytes, hipHostMallocDefault));
    CHECK(hipMemset(vbuf, 0, victim_ints * 4));
    CHECK(hipMemset(src, 1, copy_bytes));
    hipStream_t s1, s2;
    CHECK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    CHECK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    hipEvent_t a0, a1, b0, b1;
    CHECK(hipEventCreate(&a0));
    CHECK(hipEventCreate(&a1));
    CHECK(hipEventCreate(&b0));
    CHECK(hipEventCreate(&b1));
    const char *names[3] = {"reads 1 GiB", "writes 1 GiB", "16M returning atomics"};
    const size_t vn[3] = {victim_ints, victim_ints, (size_t)16 << 20};
    for (int what = 0; what < 3; what++) {
        float alone = 0.f;
        for (int rep = 0; rep < 3; rep++) {
            CHECK(hipEventRecord(a0, s1));
            hipLaunchKernelGGL(k_victim, dim3(2048), dim3(256), 0, s1, vbuf, vn[what], what, sink);
            CHECK(hipEventRecord(a1, s1));
            CHECK(hipStreamSynchronize(s1));
            CHECK(hipEventElapsedTime(&alone, a0, a1));
        }
        printf("victim %-22s alone %.3f ms\n", names[what], alone);
        for (int mode = -1; mode < 3; mode++)
            for (int blocks : {2, 8, 32}) {
                if (mode < 0 && blocks != 2) continue;
                float tv = 0.f, tc = 0.f;
                for (int rep = 0; rep < 2; rep++) {
                    CHECK(hipEventRecord(b0, s2));
                    if (mode < 0) CHECK(hipMemcpyAsync(host, src, copy_bytes, hipMemcpyDeviceToHost, s2));
                    else if (mode == 0) hipLaunchKernelGGL(k_copy<0>, dim3(blocks), dim3(256), 0, s2, host, src, copy_bytes / 16);
                    else if (mode == 1) hipLaunchKernelGGL(k_copy<1>, dim3(blocks), dim3(256), 0, s2, host, src, copy_bytes / 16);
                    else hipLaunchKernelGGL(k_copy<2>, dim3(blocks), dim3(256), 0, s2, host, src, copy_bytes / 16);
                    CHECK(hipEventRecord(b1, s2));
                    CHECK(hipEventRecord(a0, s1));
                    hipLaunchKernelGGL(k_victim, dim3(2048), dim3(256), 0, s1, vbuf, vn[what], what, sink);
                    CHECK(hipEventRecord(a1, s1));
                    CHECK(hipStreamSynchronize(s1));
                    CHECK(hipStreamSynchronize(s2));
                    CHECK(hipEventElapsedTime(&tv, a0, a1));
                    CHECK(hipEventElapsedTime(&tc, b0, b1));
                }
                printf("   beside %-28s %2d blocks: victim %.3f ms, copy %.3f ms (%.1f GB/s)\n",
                       mode < 0 ? "hipMemcpyAsync" : mode == 0 ? "copy kernel, plain stores" : mode == 1 ? "copy kernel, nt stores" : "copy kernel, sc0 sc1 stores",
                       mode < 0 ? 0 : blocks, tv, tc, copy_bytes / (tc * 1e-3) / 1e9);
            }
    }
    // the copy alone
    for (int mode = 0; mode < 3; mode++)
        for (int blocks : {1, 2, 4, 8, 32}) {
            float tc = 0.f;
            for (int rep = 0; rep < 2; rep++) {
                CHECK(hipEventRecord(b0, s2));
                if (mode == 0) hipLaunchKernelGGL(k_copy<0>, dim3(blocks), dim3(256), 0, s2, host, src, copy_bytes / 16);
                else if (mode == 1) hipLaunchKernelGGL(k_copy<1>, dim3(blocks), dim3(256), 0, s2, host, src, copy_bytes / 16);
                else hipLaunchKernelGGL(k_copy<2>, dim3(blocks), dim3(256), 0, s2, host, src, copy_bytes / 16);
                CHECK(hipEventRecord(b1, s2));
                CHECK(hipStreamSynchronize(s2));
                CHECK(hipEventElapsedTime(&tc, b0, b1));
            }
            printf("copy alone, mode %d, %2d blocks: %.3f ms (%.1f GB/s)\n", mode, blocks, tc, copy_bytes / (tc * 1e-3) / 1e9);
        }
    return 0;
}
