// sdma_probe.hip -- can the frame hand-off leave the CUs?  (dev aid)
// hipMemcpyAsync device-to-host is a blit KERNEL on this runtime (DESIGN.md "Short videos"): its PCIe-bound stores slow every
// kernel that stores next to it.  HSA offers the SDMA engines directly: hsa_amd_memory_async_copy_on_engine.  This probe asks
//   1. which engines the runtime reports for GPU -> host (hsa_amd_memory_copy_engine_status / _get_preferred_copy_engine);
//   2. what a device-to-host copy of 3 / 24 / 96 MB reaches on each of them, and through hsa_amd_memory_async_copy (engine chosen
//      by the runtime), against hipMemcpyAsync;
//   3. what a streaming kernel loses while such a copy runs (the blit kernel halves it);
//   4. whether a copy can be ordered against HIP streams from the GPU side alone: its dependency signal released by a one-wave
//      kernel that stores the signal's value, its completion signal awaited by a one-wave kernel that polls it (bounded).
// Build: hipcc --offload-arch=gfx950 -O2 tools/sdma_probe.hip -o _variants/sdma_probe -lhsa-runtime64
#include <hip/hip_runtime.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include <hsa/amd_hsa_signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
#define HK(x) do { hsa_status_t e_ = (x); if (e_ != HSA_STATUS_SUCCESS) { const char* m_ = 0; hsa_status_string(e_, &m_); printf("%s: 0x%x %s\n", #x, (unsigned) e_, m_ ? m_ : "?"); } } while (0)
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) k_busy(const u4* __restrict__ a, u4* __restrict__ b, size_t n16)
{
    const size_t stride = (size_t) gridDim.x * blockDim.x;
    for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) { u4 v = a[i]; v.x += 1; b[i] = v; }
}
__global__ void k_fill(uint32_t* p, size_t n, uint32_t seed)
{
    const size_t stride = (size_t) gridDim.x * blockDim.x;
    for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) p[i] = seed + (uint32_t) i * 2654435761u;
}
// one lane: store `v` into a signal's value (system scope, release): what lets an SDMA copy that polls the signal go
__global__ void k_signal_store(volatile int64_t* value, int64_t v)
{
    __threadfence_system();
    __hip_atomic_store((int64_t*) value, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// one lane: poll a signal's value until it is <= 0 or `max_spins` polls have gone by; out[0] = polls taken, out[1] = last value
__global__ void k_signal_wait(volatile int64_t* value, long long max_spins, long long* out)
{
    long long n = 0;
    int64_t v;
    while ((v = __hip_atomic_load((int64_t*) value, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM)) > 0 && n < max_spins) { n++; __builtin_amdgcn_s_sleep(8); }
    out[0] = n; out[1] = v;
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
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
static volatile int64_t* value_of(hsa_signal_t s) { return &((amd_signal_t*) s.handle)->value; }

int main(int argc, char** argv)
{
    const size_t FRAME = 1024 * 1024 * 3;
    CK(hipSetDevice(0));
    uint8_t *dev, *host, *bsrc, *bdst;
    const size_t BIG = 96 * FRAME / 3;        // 96 MB
    CK(hipMalloc(&dev, BIG));
    CK(hipHostMalloc(&host, BIG, hipHostMallocDefault));
    memset(host, 1, BIG);
    CK(hipMalloc(&bsrc, 64u << 20));
    CK(hipMalloc(&bdst, 64u << 20));
    hipLaunchKernelGGL(k_fill, dim3(1024), dim3(256), 0, 0, (uint32_t*) dev, BIG / 4, 7u);
    CK(hipDeviceSynchronize());
    HK(hsa_init());
    HK(hsa_iterate_agents(on_agent, nullptr));
    if (!g_have_gpu || !g_have_cpu) { printf("no GPU / CPU agent\n"); return 1; }
    char name[64] = {};
    hsa_agent_get_info(g_gpu, HSA_AGENT_INFO_NAME, name);
    printf("gpu agent %s\n", name);
    uint32_t mask = 0, pref = 0;
    hsa_status_t st = hsa_amd_memory_copy_engine_status(g_cpu, g_gpu, &mask);
    printf("copy_engine_status(dst cpu, src gpu): status 0x%x mask 0x%x\n", (unsigned) st, mask);
    st = hsa_amd_memory_get_preferred_copy_engine(g_cpu, g_gpu, &pref);
    printf("preferred_copy_engine(dst cpu, src gpu): status 0x%x mask 0x%x\n", (unsigned) st, pref);
    uint32_t mask_h2d = 0;
    st = hsa_amd_memory_copy_engine_status(g_gpu, g_cpu, &mask_h2d);
    printf("copy_engine_status(dst gpu, src cpu): status 0x%x mask 0x%x\n", (unsigned) st, mask_h2d);

    hsa_signal_t done;
    HK(hsa_signal_create(1, 0, nullptr, &done));
    auto check = [&](size_t bytes, const char* what) {
        size_t bad = 0;
        const uint32_t* h = (const uint32_t*) host;
        for (size_t i = 0; i < bytes / 4; i += 4099) bad += h[i] != 7u + (uint32_t) i * 2654435761u;
        if (bad) printf("   !! %s: %zu sampled words wrong\n", what, bad);
    };
    // 2. bandwidth per engine
    hipStream_t s0, s1;
    CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    for (size_t bytes : { FRAME, 8 * FRAME, BIG }) {
        for (int rep = 0; rep < 2; rep++) {
            memset(host, 0, 4096);
            const double t0 = now();
            CK(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, s0));
            CK(hipStreamSynchronize(s0));
            const double t1 = now();
            if (rep) printf("hipMemcpyAsync            %6.1f MB: %7.1f us  %5.1f GB/s\n", bytes / 1e6, (t1 - t0) * 1e6, bytes / (t1 - t0) / 1e9);
        }
        check(bytes, "hipMemcpyAsync");
        for (int rep = 0; rep < 2; rep++) {
            memset(host, 0, 4096);
            hsa_signal_store_relaxed(done, 1);
            const double t0 = now();
            hsa_status_t e = hsa_amd_memory_async_copy(host, g_cpu, dev, g_gpu, bytes, 0, nullptr, done);
            if (e != HSA_STATUS_SUCCESS) { printf("hsa_amd_memory_async_copy: 0x%x\n", (unsigned) e); break; }
            hsa_signal_wait_scacquire(done, HSA_SIGNAL_CONDITION_LT, 1, 2000000000ull, HSA_WAIT_STATE_ACTIVE);
            const double t1 = now();
            if (rep) printf("hsa_amd_memory_async_copy %6.1f MB: %7.1f us  %5.1f GB/s\n", bytes / 1e6, (t1 - t0) * 1e6, bytes / (t1 - t0) / 1e9);
        }
        check(bytes, "hsa async copy");
        for (int eng = 0; eng < 16; eng++) {
            if (!(mask & (1u << eng))) continue;
            bool ok = true;
            for (int rep = 0; rep < 2 && ok; rep++) {
                memset(host, 0, 4096);
                hsa_signal_store_relaxed(done, 1);
                const double t0 = now();
                hsa_status_t e = hsa_amd_memory_async_copy_on_engine(host, g_cpu, dev, g_gpu, bytes, 0, nullptr, done, (hsa_amd_sdma_engine_id_t) (1u << eng), true);
                if (e != HSA_STATUS_SUCCESS) { printf("copy_on_engine %d: 0x%x\n", eng, (unsigned) e); ok = false; break; }
                hsa_signal_wait_scacquire(done, HSA_SIGNAL_CONDITION_LT, 1, 2000000000ull, HSA_WAIT_STATE_ACTIVE);
                const double t1 = now();
                if (rep) printf("copy_on_engine %2d         %6.1f MB: %7.1f us  %5.1f GB/s\n", eng, bytes / 1e6, (t1 - t0) * 1e6, bytes / (t1 - t0) / 1e9);
            }
            if (ok) check(bytes, "copy_on_engine");
        }
    }
    // 3. what a streaming kernel loses next to a copy: 200 launches of k_busy (12 MB in, 12 MB out each) on s1, alone / next to
    // hipMemcpyAsync of 8 frames back to back on s0 / next to SDMA copies of 8 frames back to back
    const int eng0 = mask ? __builtin_ctz(mask) : -1;
    const int engp = pref ? __builtin_ctz(pref) : eng0;
    for (int mode = 0; mode < 4; mode++) {
        if ((mode == 2 && eng0 < 0) || (mode == 3 && (engp < 0 || engp == eng0))) continue;
        for (int rep = 0; rep < 2; rep++) {
            CK(hipDeviceSynchronize());
            const int NK = 400, NC = 12;
            std::vector<hsa_signal_t> sig(NC);
            for (auto& x : sig) HK(hsa_signal_create(1, 0, nullptr, &x));
            const double t0 = now();
            double tc = 0.0;
            if (mode == 1) for (int c = 0; c < NC; c++) CK(hipMemcpyAsync(host + (size_t) (c % 4) * 8 * FRAME, dev, 8 * FRAME, hipMemcpyDeviceToHost, s0));
            if (mode >= 2) for (int c = 0; c < NC; c++)
                HK(hsa_amd_memory_async_copy_on_engine(host + (size_t) (c % 4) * 8 * FRAME, g_cpu, dev, g_gpu, 8 * FRAME, c ? 1 : 0, c ? &sig[c - 1] : nullptr, sig[c],
                                                       (hsa_amd_sdma_engine_id_t) (1u << (mode == 2 ? eng0 : engp)), true));
            for (int k = 0; k < NK; k++) hipLaunchKernelGGL(k_busy, dim3(2048), dim3(256), 0, s1, (const u4*) bsrc, (u4*) bdst, (size_t) (12u << 20) / 16);
            if (mode == 1) { CK(hipStreamSynchronize(s0)); tc = now() - t0; }
            if (mode >= 2) { hsa_signal_wait_scacquire(sig[NC - 1], HSA_SIGNAL_CONDITION_LT, 1, 4000000000ull, HSA_WAIT_STATE_ACTIVE); tc = now() - t0; }
            CK(hipStreamSynchronize(s1));
            const double tk = now() - t0;
            if (rep) printf("%-34s kernels %7.1f us each (%d launches)%s", mode == 0 ? "k_busy alone" : mode == 1 ? "k_busy next to hipMemcpyAsync" : mode == 2 ? "k_busy next to SDMA (first engine)" : "k_busy next to SDMA (preferred)",
                            tk / NK * 1e6, NK, mode ? "" : "\n");
            if (rep && mode) printf(";  copies %5.1f GB/s (%d x 25 MB in %.0f us)\n", NC * 8.0 * FRAME / tc / 1e9, NC, tc * 1e6);
            for (auto& x : sig) hsa_signal_destroy(x);
        }
    }
    // 4. ordering from the GPU side: dependency released by a kernel's store, completion awaited by a polling kernel
    if (eng0 >= 0) {
        hsa_signal_t dep, fin;
        HK(hsa_signal_create(1, 0, nullptr, &dep));
        HK(hsa_signal_create(1, 0, nullptr, &fin));
        long long* out;
        CK(hipHostMalloc(&out, 64, hipHostMallocDefault));
        out[0] = out[1] = -7;
        memset(host, 0, 8 * FRAME);
        // can the GPU see the signals' values at all?  (hipHostRegister-free: the signal pool is fine-grained system memory)
        hipPointerAttribute_t attr;
        hipError_t pe = hipPointerGetAttributes(&attr, (const void*) value_of(dep));
        printf("signal value at %p: hipPointerGetAttributes -> %s\n", (void*) value_of(dep), pe == hipSuccess ? "known to HIP" : hipGetErrorString(pe));
        (void) hipGetLastError();
        const double t0 = now();
        HK(hsa_amd_memory_async_copy_on_engine(host, g_cpu, dev, g_gpu, 8 * FRAME, 1, &dep, fin, (hsa_amd_sdma_engine_id_t) (1u << eng0), true));
        // the copy must NOT have started: give it 2 ms and look
        while (now() - t0 < 2e-3) { }
        printf("before the release: completion signal %lld, host word %u (copy %s)\n", (long long) hsa_signal_load_relaxed(fin), ((uint32_t*) host)[1000],
               ((uint32_t*) host)[1000] == 0 ? "waiting, as it should" : "ALREADY RAN");
        for (int k = 0; k < 4; k++) hipLaunchKernelGGL(k_busy, dim3(2048), dim3(256), 0, s1, (const u4*) bsrc, (u4*) bdst, (size_t) (12u << 20) / 16);
        hipLaunchKernelGGL(k_signal_store, dim3(1), dim3(1), 0, s1, value_of(dep), (int64_t) 0);
        hipLaunchKernelGGL(k_signal_wait, dim3(1), dim3(1), 0, s1, value_of(fin), 4000000ll, out);
        hipError_t e = hipStreamSynchronize(s1);
        const double t1 = now();
        printf("release + poll from kernels: %s, %.0f us, polls %lld, last value %lld, completion signal now %lld\n", hipGetErrorString(e), (t1 - t0) * 1e6 - 2000.0, out[0], out[1],
               (long long) hsa_signal_load_relaxed(fin));
        if (hsa_signal_load_relaxed(fin) > 0) {
            // the engine never saw the store: release from the host so that nothing is left pending
            hsa_signal_store_screlease(dep, 0);
            hsa_signal_wait_scacquire(fin, HSA_SIGNAL_CONDITION_LT, 1, 2000000000ull, HSA_WAIT_STATE_BLOCKED);
            printf("   (released from the host instead: completion signal %lld)\n", (long long) hsa_signal_load_relaxed(fin));
        }
        check(8 * FRAME, "gpu-ordered copy");
    }
    printf("done\n");
    return 0;
}
