// d2h_probe.hip -- how fast can finished frames leave the GPU?  (dev aid; not part of the library)
//   hipcc --offload-arch=gfx950 -O3 tools/d2h_probe.hip -o tools/d2h_probe
// Measures device -> pinned-host throughput for the ways kbe_render_video could deliver frames:
//   memcpy   hipMemcpyAsync on one stream (what round 1 shipped), by transfer size
//   memcpy2  the same, transfers alternating over two streams
//   kernel   a copy kernel storing straight into device-visible pinned memory, by grid size / store flavour
// Each with and without a bandwidth-hungry kernel running on another stream (the frame loop's stand-in).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef unsigned int u4 __attribute__((ext_vector_type(4)));

template <int FLAVOUR>
__global__ void __launch_bounds__(256) k_copy(const u4* __restrict__ src, u4* __restrict__ dst, size_t n16)
{
    const size_t stride = (size_t) gridDim.x * blockDim.x;
    for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) {
        const u4 v = __builtin_nontemporal_load(src + i);
        if (FLAVOUR == 0) dst[i] = v;
        else __builtin_nontemporal_store(v, dst + i);
    }
}

// stand-in for the frame loop: streams `bytes` through the chip, `iters` times
__global__ void __launch_bounds__(256) k_busy(const u4* __restrict__ a, u4* __restrict__ b, size_t n16, int iters)
{
    const size_t stride = (size_t) gridDim.x * blockDim.x;
    for (int it = 0; it < iters; it++)
        for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) {
            u4 v = a[i];
            v.x += it;
            b[i] = v;
        }
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv)
{
    const size_t FRAME = 1024 * 1024 * 3;
    const size_t TOTAL = 512 * FRAME;                     // 1.6 GB moved per measurement
    const bool busy = argc > 1 && !strcmp(argv[1], "busy");
    uint8_t *dev, *host, *bsrc, *bdst;
    CK(hipMalloc(&dev, 64 * FRAME));
    CK(hipMemset(dev, 7, 64 * FRAME));
    CK(hipHostMalloc(&host, 64 * FRAME, hipHostMallocDefault));
    memset(host, 1, 64 * FRAME);
    CK(hipMalloc(&bsrc, 256 << 20));
    CK(hipMalloc(&bdst, 256 << 20));
    uint8_t* host_dev;
    CK(hipHostGetDevicePointer((void**) &host_dev, host, 0));
    hipStream_t s[4], sb;
    for (int i = 0; i < 4; i++) CK(hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    CK(hipDeviceSynchronize());

    auto start_busy = [&]() {
        if (busy) hipLaunchKernelGGL(k_busy, dim3(2048), dim3(256), 0, sb, (const u4*) bsrc, (u4*) bdst, (size_t) (256 << 20) / 16, 400);
    };
    auto report = [&](const char* what, double dt) {
        CK(hipDeviceSynchronize());
        printf("%-44s %7.2f GB/s  (%6.0f frames/s at 3.1 MB)%s\n", what, TOTAL / dt / 1e9, TOTAL / dt / FRAME, busy ? "  [busy chip]" : "");
        fflush(stdout);
    };

    for (size_t frames : { (size_t) 1, (size_t) 4, (size_t) 16, (size_t) 64 }) {
        const size_t bytes = frames * FRAME;
        for (int ns = 1; ns <= 4; ns *= 2) {
            start_busy();
            // warm
            for (int i = 0; i < ns; i++) CK(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, s[i]));
            for (int i = 0; i < ns; i++) CK(hipStreamSynchronize(s[i]));
            const double t0 = now();
            const size_t n = TOTAL / bytes;
            for (size_t k = 0; k < n; k++) {
                const size_t slot = (k * frames) % 64;
                const size_t off = (slot + frames <= 64 ? slot : 0) * FRAME;
                CK(hipMemcpyAsync(host + off, dev + off, bytes, hipMemcpyDeviceToHost, s[k % ns]));
            }
            for (int i = 0; i < ns; i++) CK(hipStreamSynchronize(s[i]));
            const double dt = now() - t0;
            char what[128];
            snprintf(what, sizeof what, "hipMemcpyAsync %2zu frames/transfer, %d stream%s", frames, ns, ns > 1 ? "s" : "");
            report(what, dt);
        }
    }
    for (int flavour = 0; flavour < 2; flavour++)
        for (int grid : { 4, 8, 16, 32, 64, 128, 256, 1024 }) {
            start_busy();
            const size_t bytes = 16 * FRAME, n = TOTAL / bytes;
            auto launch = [&](size_t k) {
                const size_t off = ((k * 16) % 64) * FRAME;
                if (flavour == 0) hipLaunchKernelGGL(k_copy<0>, dim3(grid), dim3(256), 0, s[0], (const u4*) (dev + off), (u4*) (host_dev + off), bytes / 16);
                else hipLaunchKernelGGL(k_copy<1>, dim3(grid), dim3(256), 0, s[0], (const u4*) (dev + off), (u4*) (host_dev + off), bytes / 16);
            };
            launch(0);
            CK(hipStreamSynchronize(s[0]));
            const double t0 = now();
            for (size_t k = 0; k < n; k++) launch(k);
            CK(hipStreamSynchronize(s[0]));
            const double dt = now() - t0;
            char what[128];
            snprintf(what, sizeof what, "copy kernel, %4d workgroups, %s stores", grid, flavour ? "nontemporal" : "plain");
            report(what, dt);
        }
    // sanity: the kernel path really wrote host memory
    printf("host[0] = %d (expect 7)\n", host[0]);
    return 0;
}
