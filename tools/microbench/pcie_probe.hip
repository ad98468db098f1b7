// Micro-benchmark (gfx950 box): host<->device copy rates of page-locked buffers as the drop-in pair issues them --
// aligned and byte-misaligned hipMemcpy, and a copy kernel that reads / writes the page-locked host buffer directly.
// Build: hipcc --offload-arch=gfx950 -O3 -o pcie_probe pcie_probe.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// byte-granular copy: 16-byte words where both sides allow it after a common head
__global__ __launch_bounds__(256) void copy_kernel(uint8_t* __restrict__ dst, const uint8_t* __restrict__ src, size_t n)
{
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (size_t)gridDim.x * blockDim.x;
    if ((((uintptr_t)dst ^ (uintptr_t)src) & 15) == 0) {
        size_t head = (16 - ((uintptr_t)dst & 15)) & 15;
        if (head > n) head = n;
        if (tid < head) dst[tid] = src[tid];
        const size_t words = (n - head) / 16;
        const uint4* s = reinterpret_cast<const uint4*>(src + head);
        uint4* d = reinterpret_cast<uint4*>(dst + head);
        for (size_t i = tid; i < words; i += nth) d[i] = s[i];
        const size_t done = head + words * 16;
        if (tid < n - done) dst[done + tid] = src[done + tid];
    } else {
        for (size_t i = tid; i < n; i += nth) dst[i] = src[i];
    }
}

int main()
{
    const size_t n = (size_t)256 << 20;
    uint8_t *h, *d;
    CHECK(hipHostMalloc((void**)&h, n + 4096, hipHostMallocDefault));
    CHECK(hipMalloc((void**)&d, n + 4096));
    memset(h, 1, n + 4096);
    CHECK(hipMemset(d, 2, n + 4096));
    CHECK(hipDeviceSynchronize());
    auto time_it = [&](const char* name, auto fn) {
        double best = 1e30, first = 0;
        for (int r = 0; r < 4; ++r) {
            CHECK(hipDeviceSynchronize());
            const double t0 = now();
            fn();
            CHECK(hipDeviceSynchronize());
            const double t = now() - t0;
            best = t < best ? t : best;
            if (r == 0) first = t;
        }
        printf("%-52s %7.2f ms  %6.1f GB/s   (first call %7.2f ms)\n", name, best * 1e3, n / best / 1e9, first * 1e3);
        fflush(stdout);
    };
    {   // the drop-in's situation: every buffer is new (allocated, page-locked) and used exactly once
        for (int rep = 0; rep < 3; ++rep) {
            uint8_t *fh, *fd, *fh2;
            double t0 = now();
            CHECK(hipHostMalloc((void**)&fh, n, hipHostMallocDefault));
            CHECK(hipHostMalloc((void**)&fh2, n, hipHostMallocDefault));
            const double t_pin = now() - t0;
            memset(fh, 5, n);
            t0 = now();
            CHECK(hipMalloc((void**)&fd, n));
            const double t_alloc = now() - t0;
            t0 = now();
            CHECK(hipMemcpy(fd, fh, n, hipMemcpyHostToDevice));
            const double t_h2d = now() - t0;
            t0 = now();
            CHECK(hipMemcpy(fh2, fd, n, hipMemcpyDeviceToHost));
            const double t_d2h = now() - t0;
            t0 = now();
            CHECK(hipMemcpy(fh2, fd, n, hipMemcpyDeviceToHost));
            const double t_d2h2 = now() - t0;
            printf("fresh buffers: pin 2x256MiB %.2f ms, hipMalloc %.2f ms, first H2D %.2f ms, first D2H (untouched host dst) %.2f ms, second D2H %.2f ms\n",
                   t_pin * 1e3, t_alloc * 1e3, t_h2d * 1e3, t_d2h * 1e3, t_d2h2 * 1e3);
            CHECK(hipFree(fd));
            CHECK(hipHostFree(fh));
            CHECK(hipHostFree(fh2));
        }
    }
    time_it("H2D hipMemcpy aligned", [&] { CHECK(hipMemcpy(d, h, n, hipMemcpyHostToDevice)); });
    time_it("H2D hipMemcpy src+9", [&] { CHECK(hipMemcpy(d, h + 9, n, hipMemcpyHostToDevice)); });
    time_it("H2D hipMemcpy src+9 dst+9", [&] { CHECK(hipMemcpy(d + 9, h + 9, n, hipMemcpyHostToDevice)); });
    time_it("D2H hipMemcpy aligned", [&] { CHECK(hipMemcpy(h, d, n, hipMemcpyDeviceToHost)); });
    time_it("D2H hipMemcpy dst+9", [&] { CHECK(hipMemcpy(h + 9, d, n, hipMemcpyDeviceToHost)); });
    time_it("D2H hipMemcpy dst+9 src+9", [&] { CHECK(hipMemcpy(h + 9, d + 9, n, hipMemcpyDeviceToHost)); });
    for (int blocks : {256, 1024, 4096}) {
        char name[96];
        snprintf(name, sizeof name, "H2D copy kernel (%d x 256) aligned", blocks);
        time_it(name, [&] { hipLaunchKernelGGL(copy_kernel, dim3(blocks), dim3(256), 0, 0, d, h, n); });
        snprintf(name, sizeof name, "H2D copy kernel (%d x 256) src+9 dst+9", blocks);
        time_it(name, [&] { hipLaunchKernelGGL(copy_kernel, dim3(blocks), dim3(256), 0, 0, d + 9, h + 9, n); });
        snprintf(name, sizeof name, "D2H copy kernel (%d x 256) aligned", blocks);
        time_it(name, [&] { hipLaunchKernelGGL(copy_kernel, dim3(blocks), dim3(256), 0, 0, h, d, n); });
        snprintf(name, sizeof name, "D2H copy kernel (%d x 256) dst+9 src+9", blocks);
        time_it(name, [&] { hipLaunchKernelGGL(copy_kernel, dim3(blocks), dim3(256), 0, 0, h + 9, d + 9, n); });
    }
    // both directions at once (full duplex): H2D on one stream, D2H on another
    hipStream_t s1, s2;
    CHECK(hipStreamCreate(&s1));
    CHECK(hipStreamCreate(&s2));
    uint8_t *h2, *d2;
    CHECK(hipHostMalloc((void**)&h2, n, hipHostMallocDefault));
    CHECK(hipMalloc((void**)&d2, n));
    memset(h2, 3, n);
    time_it("duplex hipMemcpyAsync H2D + D2H (each 256 MiB)", [&] {
        CHECK(hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, s1));
        CHECK(hipMemcpyAsync(h2, d2, n, hipMemcpyDeviceToHost, s2));
    });
    // chunked H2D: 8 x 32 MiB async
    time_it("H2D 8 x 32 MiB hipMemcpyAsync one stream", [&] {
        for (int c = 0; c < 8; ++c) CHECK(hipMemcpyAsync(d + (size_t)c * (n / 8), h + (size_t)c * (n / 8), n / 8, hipMemcpyHostToDevice, s1));
    });
    // pageable source
    uint8_t* p = (uint8_t*)malloc(n);
    memset(p, 4, n);
    time_it("H2D hipMemcpy pageable", [&] { CHECK(hipMemcpy(d, p, n, hipMemcpyHostToDevice)); });
    time_it("D2H hipMemcpy pageable", [&] { CHECK(hipMemcpy(p, d, n, hipMemcpyDeviceToHost)); });
    return 0;
}
