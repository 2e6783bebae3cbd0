// sdma_probe2.hip -- WHAT does a device-to-host copy slow down?  (dev aid; follows tools/sdma_probe.hip)
// The first probe found that a streaming kernel runs 3.7 x slower next to a device-to-host copy whether the copy is the runtime's
// blit kernel or an SDMA engine driven through HSA -- so it is not the CUs the copy takes.  This one separates the candidates:
// kernels of five kinds (tiny and cache-resident; HBM stream in + out; read only; write only; pure arithmetic), each timed with HIP
// events alone and while copies run back to back on an SDMA engine (device-to-host, host-to-device) and through hipMemcpyAsync.
// Build: hipcc --offload-arch=gfx950 -O2 tools/sdma_probe2.hip -o _variants/sdma_probe2 -lhsa-runtime64
#include <hip/hip_runtime.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
#define HK(x) do { hsa_status_t e_ = (x); if (e_ != HSA_STATUS_SUCCESS) { printf("%s: 0x%x\n", #x, (unsigned) e_); } } while (0)
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) k_rw(const u4* __restrict__ a, u4* __restrict__ b, size_t n16)
{
    const size_t stride = (size_t) gridDim.x * blockDim.x;
    for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) { u4 v = a[i]; v.x += 1; b[i] = v; }
}
__global__ void __launch_bounds__(256) k_read(const u4* __restrict__ a, u4* __restrict__ b, size_t n16)
{
    const size_t stride = (size_t) gridDim.x * blockDim.x;
    u4 acc = { 0, 0, 0, 0 };
    for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) { u4 v = a[i]; acc.x ^= v.x; acc.y += v.y; acc.z ^= v.z; acc.w += v.w; }
    if (acc.x == 0x12345u && acc.y == 7u) b[0] = acc;
}
__global__ void __launch_bounds__(256) k_write(u4* __restrict__ b, size_t n16, unsigned seed)
{
    const size_t stride = (size_t) gridDim.x * blockDim.x;
    for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) { u4 v = { seed, (unsigned) i, seed, seed }; b[i] = v; }
}
__global__ void __launch_bounds__(256) k_alu(float* out, int iters)
{
    float a = threadIdx.x * 0.001f, b = blockIdx.x * 0.002f + 1.0f, c = 0.5f, d = 0.25f;
    for (int i = 0; i < iters; i++) { a = __builtin_fmaf(a, b, c); b = __builtin_fmaf(b, c, d); c = __builtin_fmaf(c, d, a); d = __builtin_fmaf(d, a, b); }
    if (a + b + c + d == 12345.678f) out[0] = a;
}
__global__ void __launch_bounds__(256) k_lds(float* out, int iters)
{
    __shared__ float s[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) s[i] = i;
    __syncthreads();
    float acc = 0.0f;
    unsigned j = threadIdx.x;
    for (int i = 0; i < iters; i++) { acc += s[j & 4095]; j = j * 5u + 1u; }
    if (acc == 12345.678f) out[0] = acc;
}
static hsa_agent_t g_gpu, g_cpu;
static int g_have_gpu = 0, g_have_cpu = 0;
static hsa_status_t on_agent(hsa_agent_t a, void*)
{
    hsa_device_type_t t;
    hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t);
    if (t == HSA_DEVICE_TYPE_GPU && !g_have_gpu) { g_gpu = a; g_have_gpu = 1; }
    if (t == HSA_DEVICE_TYPE_CPU && !g_have_cpu) { g_cpu = a; g_have_cpu = 1; }
    return HSA_STATUS_SUCCESS;
}
int main()
{
    CK(hipSetDevice(0));
    const size_t COPY = 24u << 20, BIG = 512u << 20, SMALL = 12u << 20;
    uint8_t *dev, *host, *a, *b;
    float* out;
    CK(hipMalloc(&dev, 4 * COPY));
    CK(hipHostMalloc(&host, 4 * COPY, hipHostMallocDefault));
    memset(host, 1, 4 * COPY);
    CK(hipMalloc(&a, BIG)); CK(hipMalloc(&b, BIG)); CK(hipMalloc(&out, 4096));
    CK(hipMemset(a, 1, BIG)); CK(hipMemset(b, 2, BIG)); CK(hipMemset(dev, 3, 4 * COPY));
    HK(hsa_init());
    HK(hsa_iterate_agents(on_agent, nullptr));
    uint32_t mask = 0;
    hsa_amd_memory_copy_engine_status(g_cpu, g_gpu, &mask);
    const int eng = mask ? __builtin_ctz(mask) : 0;
    uint32_t mask_in = 0;
    hsa_amd_memory_copy_engine_status(g_gpu, g_cpu, &mask_in);
    const int eng_in = mask_in ? __builtin_ctz(mask_in & ~(1u << eng) ? mask_in & ~(1u << eng) : mask_in) : 0;
    hipStream_t sk, sc;
    CK(hipStreamCreateWithFlags(&sk, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&sc, hipStreamNonBlocking));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    struct Kind { const char* name; int launches; };
    const Kind kinds[] = { { "tiny rw, 12 MB, cache-resident (2048 wg)", 600 }, { "stream rw 512 MB -> 512 MB", 24 }, { "read 512 MB", 40 }, { "write 512 MB", 40 },
                           { "pure ALU (2048 wg x 256 thr)", 40 }, { "LDS reads only", 40 }, { "tiny rw, 12 MB, 256 wg", 600 } };
    auto launch = [&](int kind) {
        switch (kind) {
        case 0: hipLaunchKernelGGL(k_rw, dim3(2048), dim3(256), 0, sk, (const u4*) a, (u4*) b, SMALL / 16); break;
        case 1: hipLaunchKernelGGL(k_rw, dim3(4096), dim3(256), 0, sk, (const u4*) a, (u4*) b, BIG / 16); break;
        case 2: hipLaunchKernelGGL(k_read, dim3(4096), dim3(256), 0, sk, (const u4*) a, (u4*) b, BIG / 16); break;
        case 3: hipLaunchKernelGGL(k_write, dim3(4096), dim3(256), 0, sk, (u4*) b, BIG / 16, 5u); break;
        case 4: hipLaunchKernelGGL(k_alu, dim3(2048), dim3(256), 0, sk, out, 20000); break;
        case 5: hipLaunchKernelGGL(k_lds, dim3(2048), dim3(256), 0, sk, out, 20000); break;
        case 6: hipLaunchKernelGGL(k_rw, dim3(256), dim3(256), 0, sk, (const u4*) a, (u4*) b, SMALL / 16); break;
        }
    };
    const char* modes[] = { "alone", "+ SDMA device-to-host", "+ SDMA host-to-device", "+ hipMemcpyAsync device-to-host", "+ SDMA device-to-device" };
    for (int kind = 0; kind < 7; kind++) {
        double alone = 0.0;
        for (int mode = 0; mode < 5; mode++) {
            float best = 1e30f;
            double copy_rate = 0.0;
            for (int rep = 0; rep < 3; rep++) {
                CK(hipDeviceSynchronize());
                // warm the kernel's clocks
                for (int k = 0; k < 3; k++) launch(kind);
                CK(hipStreamSynchronize(sk));
                const int NC = 400;         // 400 x 24 MB: ~170 ms of link time -- the kernels finish long before; the rest is drained below
                std::vector<hsa_signal_t> sig;
                if (mode == 1 || mode == 2 || mode == 4) {
                    sig.resize(NC);
                    for (auto& x : sig) HK(hsa_signal_create(1, 0, nullptr, &x));
                    for (int c = 0; c < NC; c++) {
                        uint8_t* h = host + (size_t) (c & 3) * COPY; uint8_t* d = dev + (size_t) (c & 3) * COPY;
                        if (mode == 1) HK(hsa_amd_memory_async_copy_on_engine(h, g_cpu, d, g_gpu, COPY, c ? 1 : 0, c ? &sig[c - 1] : nullptr, sig[c], (hsa_amd_sdma_engine_id_t) (1u << eng), true));
                        else if (mode == 2) HK(hsa_amd_memory_async_copy_on_engine(d, g_gpu, h, g_cpu, COPY, c ? 1 : 0, c ? &sig[c - 1] : nullptr, sig[c], (hsa_amd_sdma_engine_id_t) (1u << eng_in), true));
                        else HK(hsa_amd_memory_async_copy_on_engine(dev + (size_t) ((c + 1) & 3) * COPY, g_gpu, d, g_gpu, COPY, c ? 1 : 0, c ? &sig[c - 1] : nullptr, sig[c], (hsa_amd_sdma_engine_id_t) (1u << eng), true));
                    }
                }
                if (mode == 3) for (int c = 0; c < 40; c++) CK(hipMemcpyAsync(host + (size_t) (c & 3) * COPY, dev + (size_t) (c & 3) * COPY, COPY, hipMemcpyDeviceToHost, sc));
                CK(hipEventRecord(e0, sk));
                for (int k = 0; k < kinds[kind].launches; k++) launch(kind);
                CK(hipEventRecord(e1, sk));
                CK(hipEventSynchronize(e1));
                float ms = 0.0f;
                CK(hipEventElapsedTime(&ms, e0, e1));
                // how far did the copies get while the kernels ran?
                if (!sig.empty()) {
                    int done = 0;
                    for (int c = 0; c < NC; c++) done += hsa_signal_load_relaxed(sig[c]) <= 0;
                    copy_rate = (double) done * COPY / (ms * 1e-3) / 1e9;
                    hsa_signal_wait_scacquire(sig[NC - 1], HSA_SIGNAL_CONDITION_LT, 1, 8000000000ull, HSA_WAIT_STATE_BLOCKED);
                    for (auto& x : sig) hsa_signal_destroy(x);
                }
                if (mode == 3) CK(hipStreamSynchronize(sc));
                if (ms < best) best = ms;
            }
            const double us = best * 1e3 / kinds[kind].launches;
            if (mode == 0) alone = us;
            printf("%-44s %-32s %9.1f us per launch  x %.2f%s", kinds[kind].name, modes[mode], us, us / alone, copy_rate > 0.0 ? "" : "\n");
            if (copy_rate > 0.0) printf("   (copies ran at %.1f GB/s meanwhile)\n", copy_rate);
            fflush(stdout);
        }
    }
    return 0;
}
