// calib_d2h.hip -- device -> page-locked host: the copy engine (hipMemcpyAsync) against a kernel that stores into the host block,
// for one 4K RGBA frame (34.4 MB) and a few other sizes; and both directions at once (duplex).  tools/bin/calib_d2h
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ __launch_bounds__(256) void k_copy(uint4 *__restrict__ dst, const uint4 *__restrict__ src, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}
int main()
{
    const size_t sizes[] = { 1u << 20, 8u << 20, 34406400, 137625600 };
    hipStream_t st; CK(hipStreamCreate(&st));
    for (size_t bytes : sizes) {
        void *d = nullptr, *h = nullptr;
        CK(hipMalloc(&d, bytes)); CK(hipHostMalloc(&h, bytes, hipHostMallocDefault));
        CK(hipMemset(d, 0x5a, bytes)); std::memset(h, 0, bytes);
        auto timeit = [&](auto fn) { fn(); hipStreamSynchronize(st); auto t0 = std::chrono::steady_clock::now(); for (int i = 0; i < 10; i++) fn(); hipStreamSynchronize(st);
                                     return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 10; };
        const double a = timeit([&] { hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, st); });
        for (int blocks : { 64, 256, 1024 }) {
            const double b = timeit([&] { hipLaunchKernelGGL(k_copy, dim3(blocks), dim3(256), 0, st, (uint4 *)h, (const uint4 *)d, bytes / 16); });
            printf("{\"bytes\": %zu, \"memcpy_us\": %.1f, \"memcpy_GBs\": %.1f, \"kernel_blocks\": %d, \"kernel_us\": %.1f, \"kernel_GBs\": %.1f}\n", bytes, a, bytes / a / 1e3, blocks, b, bytes / b / 1e3);
        }
        const double c = timeit([&] { hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, st); });
        const double e = timeit([&] { hipLaunchKernelGGL(k_copy, dim3(256), dim3(256), 0, st, (uint4 *)d, (const uint4 *)h, bytes / 16); });
        printf("{\"bytes\": %zu, \"h2d_memcpy_us\": %.1f, \"h2d_memcpy_GBs\": %.1f, \"h2d_kernel_us\": %.1f, \"h2d_kernel_GBs\": %.1f}\n", bytes, c, bytes / c / 1e3, e, bytes / e / 1e3);
        {   // both directions at once on two streams (what a pipelined video loop asks of the link): upload into d2 while d comes down
            void *d2 = nullptr, *h2 = nullptr; hipStream_t st2;
            CK(hipMalloc(&d2, bytes)); CK(hipHostMalloc(&h2, bytes, hipHostMallocDefault)); CK(hipStreamCreate(&st2));
            std::memset(h2, 1, bytes);
            auto both = [&] { hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, st); hipMemcpyAsync(d2, h2, bytes, hipMemcpyHostToDevice, st2); };
            both(); hipStreamSynchronize(st); hipStreamSynchronize(st2);
            auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < 10; i++) both();
            hipStreamSynchronize(st); hipStreamSynchronize(st2);
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 10;
            printf("{\"bytes\": %zu, \"duplex_us_per_pair\": %.1f, \"duplex_GBs_each_way\": %.1f}\n", bytes, us, bytes / us / 1e3);
            hipFree(d2); hipHostFree(h2); hipStreamDestroy(st2);
        }
        hipFree(d); hipHostFree(h);
    }
    return 0;
}
