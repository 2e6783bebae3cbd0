// d2h_probe2.hip -- the lane structure of kbe_render_video in miniature (dev aid): per lane and frame, a render stand-in
// (streams ~100 MB through the chip) followed by a 64-workgroup copy kernel into pinned host memory, all in the lane's
// stream.  Prints host enqueue time and total time per frame, for 1..8 lanes, with and without the copy.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) k_copy(const u4* __restrict__ src, u4* __restrict__ dst, size_t n16)
{
    const size_t stride = (size_t) gridDim.x * blockDim.x;
    for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) dst[i] = __builtin_nontemporal_load(src + i);
}
__global__ void __launch_bounds__(256) k_busy(const u4* __restrict__ a, u4* __restrict__ b, size_t n16)
{
    const size_t stride = (size_t) gridDim.x * blockDim.x;
    for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) { u4 v = a[i]; v.x += 1; b[i] = v; }
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv)
{
    const size_t FRAME = 1024 * 1024 * 3;
    const int NF = 512;
    const int copy_blocks = argc > 1 ? atoi(argv[1]) : 64;
    uint8_t *dev, *host, *bsrc, *bdst;
    CK(hipMalloc(&dev, 8 * FRAME));
    CK(hipHostMalloc(&host, (size_t) NF * FRAME, hipHostMallocDefault));
    memset(host, 1, (size_t) NF * FRAME);
    CK(hipMalloc(&bsrc, 8 * (50u << 20)));
    CK(hipMalloc(&bdst, 8 * (50u << 20)));
    hipStream_t s[8];
    for (int i = 0; i < 8; i++) CK(hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking));
    for (int with_copy = 0; with_copy < 3; with_copy++)
        for (int lanes : { 1, 2, 4, 8 }) {
            for (int rep = 0; rep < 2; rep++) {
                CK(hipDeviceSynchronize());
                const double t0 = now();
                for (int f = 0; f < NF; f++) {
                    const int l = f % lanes;
                    for (int k = 0; k < 4; k++)     // four dependent launches of ~8 us each
                        hipLaunchKernelGGL(k_busy, dim3(2048), dim3(256), 0, s[l], (const u4*) (bsrc + (size_t) l * (50u << 20)), (u4*) (bdst + (size_t) l * (50u << 20)), (size_t) (12u << 20) / 16);
                    if (with_copy == 1) hipLaunchKernelGGL(k_copy, dim3(copy_blocks), dim3(256), 0, s[l], (const u4*) (dev + l * FRAME), (u4*) (host + (size_t) f * FRAME), FRAME / 16);
                    if (with_copy == 2) CK(hipMemcpyAsync(host + (size_t) f * FRAME, dev + l * FRAME, FRAME, hipMemcpyDeviceToHost, s[l]));
                }
                const double t1 = now();
                CK(hipDeviceSynchronize());
                const double t2 = now();
                if (rep) printf("%-28s lanes %d: enqueue %6.1f us/frame, total %6.1f us/frame (%5.1f GB/s)\n",
                                with_copy == 0 ? "render stand-in only" : with_copy == 1 ? "+ copy kernel per frame" : "+ hipMemcpyAsync per frame", lanes,
                                (t1 - t0) / NF * 1e6, (t2 - t0) / NF * 1e6, with_copy ? FRAME * NF / (t2 - t0) / 1e9 : 0.0);
            }
        }
    return 0;
}
