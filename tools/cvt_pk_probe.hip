// cvt_pk_probe.hip -- does v_cvt_pk_u8_f32 equal (uint8_t) (int) fmed3(v, 0, 255) -- common.py:255's clip and astype(np.uint8), i.e.
// truncation towards zero with saturation -- for EVERY fp32 bit pattern?  (dev aid, round 4: a candidate for the tile epilogue.)
//   hipcc --offload-arch=gfx950 -O2 tools/cvt_pk_probe.hip -o /tmp/cvt_pk_probe && /tmp/cvt_pk_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

__global__ void k_probe(unsigned long long* counts, uint32_t* first_bad)
{
    const uint32_t stride = gridDim.x * blockDim.x;
    unsigned long long bad = 0, bad_finite = 0;
    for (uint64_t b = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; b < (1ull << 32); b += stride) {
        const float v = __uint_as_float((uint32_t) b);
        const uint32_t want = (uint8_t) (int) __builtin_amdgcn_fmed3f(v, 0.0f, 255.0f);
        uint32_t got;
        asm volatile("v_cvt_pk_u8_f32 %0, %1, 0, 0" : "=v"(got) : "v"(v));
        if ((got & 0xFFu) != want) {
            bad++;
            if (v == v && fabsf(v) < 1.0e30f) { bad_finite++; atomicMin(first_bad, (uint32_t) b); }
        }
    }
    atomicAdd(&counts[0], bad);
    atomicAdd(&counts[1], bad_finite);
}

int main()
{
    unsigned long long* counts; uint32_t* first;
    hipMalloc(&counts, 16); hipMalloc(&first, 4);
    hipMemset(counts, 0, 16); hipMemset(first, 0xFF, 4);
    hipLaunchKernelGGL(k_probe, dim3(4096), dim3(256), 0, 0, counts, first);
    unsigned long long h[2]; uint32_t f;
    hipMemcpy(h, counts, 16, hipMemcpyDeviceToHost); hipMemcpy(&f, first, 4, hipMemcpyDeviceToHost);
    float fv; memcpy(&fv, &f, 4);
    printf("v_cvt_pk_u8_f32 against (uint8_t) (int) fmed3(v, 0, 255): %llu of 2^32 patterns differ (%llu finite); first finite difference at 0x%08x = %.9g\n", h[0], h[1], f, fv);
    return 0;
}
