// kbe_cloud.hip -- packs the resident point cloud for the fused frame kernel (layout and rationale: kbe_cloud.h).
// Runs once per video, after the set-up loop of process_kenburns has grown the cloud (common.py:175-220).
#include <hip/hip_runtime.h>
#include <math.h>
#include <string.h>

#include <rocprim/device/device_radix_sort.hpp>

#include "kbe.h"
#include "kbe_cloud.h"
#include "kbe_host.h"

using namespace kbe;

namespace {

constexpr uint32_t KEY_DEGENERATE = 0xFFFFFFFEu;    // sorts behind every ordinary point
constexpr uint32_t KEY_PADDING = 0xFFFFFFFFu;       // ... and the padding behind those

__device__ __forceinline__ uint32_t spread13(uint32_t v)      // 13 bits -> every second bit
{
    v &= 0x1FFFu;
    v = (v | (v << 8)) & 0x00FF00FFu;
    v = (v | (v << 4)) & 0x0F0F0F0Fu;
    v = (v | (v << 2)) & 0x33333333u;
    v = (v | (v << 1)) & 0x55555555u;
    return v;
}

// ordinary = finite, z >= 1 and a bounded p; everything else that is finite is "degenerate" (kept in camera space)
__device__ __forceinline__ int classify(float x, float y, float z, float fd, float& px, float& py)
{
    if (!(fabsf(x) < 1.0e30f) || !(fabsf(y) < 1.0e30f) || !(fabsf(z) < 1.0e30f)) return 0;      // non-finite: never rendered (kbe.h)
    if (z >= 1.0f) {
        const float s = fd / z;
        px = x * s;
        py = y * s;
        if (fabsf(px) < 1.0e7f && fabsf(py) < 1.0e7f) return 1;
    }
    return 2;
}

// Sort key: the Morton code of the 8 x 8-pixel cell a point projects to in the cloud's own view (focal fd, W x H raster).
// The first `raster_n` points are the pixels of the image raster (how process_kenburns builds the cloud,
// common.py:176-179): pixel (x, y) projects to x + 0.5, so its cell is its own 8 x 8 block of the raster and -- in a run
// of their own -- a block of 64 consecutive packed points IS one cell (when the raster's sides are multiples of 8; a
// denser raster gives several blocks per cell).  Appended points follow in a second run: mixed into the raster run they
// would shift its blocks off the cell grid (measured: 30 instead of ~23 candidate blocks per tile).  The hint only
// decides the run; the cell always comes from the coordinates, so a wrong hint costs speed at worst.
constexpr uint32_t KEY_APPENDED = 0x40000000u;
__global__ void __launch_bounds__(256) k_cloud_keys(const float* __restrict__ points, int N, int Np, int W, int H, float fd, int raster_n,
                                                    uint32_t* __restrict__ keys, uint32_t* __restrict__ idx)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Np) return;
    uint32_t key = KEY_PADDING;
    if (i < N) {
        float px, py;
        const int cls = classify(points[i], points[(size_t) N + i], points[2 * (size_t) N + i], fd, px, py);
        key = KEY_DEGENERATE;
        if (cls == 1) {
            // clamped to the raster's surroundings; + 1 keeps the cells left of / above the raster non-negative
            const float u = px + 0.5f * (float) W, v = py + 0.5f * (float) H;
            const int cx = (int) fminf(fmaxf(floorf(u * 0.125f) + 1.0f, 0.0f), 8190.0f);
            const int cy = (int) fminf(fmaxf(floorf(v * 0.125f) + 1.0f, 0.0f), 8190.0f);
            key = (i < raster_n ? 0u : KEY_APPENDED) | spread13((uint32_t) cx) | (spread13((uint32_t) cy) << 1);
        }
    }
    keys[i] = key;
    idx[i] = (uint32_t) i;
}

__global__ void __launch_bounds__(256) k_cloud_gather(const float* __restrict__ points, const float* __restrict__ image,
                                                      const float* __restrict__ depth, int N, int Np, const uint32_t* __restrict__ order,
                                                      CloudPoint* __restrict__ pd, CloudColour* __restrict__ col)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= Np) return;
    const uint32_t i = order[j];
    const bool real = i < (uint32_t) N;
    const size_t n = (size_t) N;
    CloudPoint p;
    p.x = real ? points[i] : 0.0f;
    p.y = real ? points[n + i] : 0.0f;
    p.z = real ? points[2 * n + i] : __builtin_nanf("");         // padding: z = NaN fails `z >= 0.001` (common.py:453)
    CloudColour c;
    c.r = real ? image[i] : 0.0f;
    c.g = real ? image[n + i] : 0.0f;
    c.b = real ? image[2 * n + i] : 0.0f;
    c.depth = real ? depth[i] : 0.0f;
    pd[j] = p;
    col[j] = c;
}

__device__ __forceinline__ void node_clear(CloudNode& n)
{
    n.px0 = n.py0 = n.z0 = n.X0 = n.Y0 = n.Z0 = INFINITY;
    n.px1 = n.py1 = n.z1 = n.X1 = n.Y1 = n.Z1 = -INFINITY;
    n.flags = 0;
    n.pad[0] = n.pad[1] = n.pad[2] = 0;
}

// level 0: one thread per block of 64 points (once per video: simplicity over speed)
__global__ void __launch_bounds__(256) k_cloud_blocks(const CloudPoint* __restrict__ pd, float fd, CloudNode* __restrict__ nodes, int n_blocks)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_blocks) return;
    CloudNode n;
    node_clear(n);
    for (int k = 0; k < kCloudBlock; k++) {
        const CloudPoint p = pd[(size_t) b * kCloudBlock + k];
        const float x = p.x, y = p.y, z = p.z;
        float px = 0.0f, py = 0.0f;
        const int cls = classify(x, y, z, fd, px, py);
        if (cls == 1) {
            n.flags |= 1u;
            n.px0 = fminf(n.px0, px); n.px1 = fmaxf(n.px1, px);
            n.py0 = fminf(n.py0, py); n.py1 = fmaxf(n.py1, py);
            n.z0 = fminf(n.z0, z); n.z1 = fmaxf(n.z1, z);
        } else if (cls == 2) {
            // process_shift scales x and y by z / (z + 1e-7) (common.py:105-106), a factor in [0, 1] for these points and
            // unbounded for z in (-1e-7, 0): the box holds 0 as well; the sliver of negative z is hopeless either way
            // (z + shift_z below the near plane unless the camera moves back by more than it moves at all)
            n.flags |= 2u;
            n.X0 = fminf(n.X0, fminf(x, 0.0f)); n.X1 = fmaxf(n.X1, fmaxf(x, 0.0f));
            n.Y0 = fminf(n.Y0, fminf(y, 0.0f)); n.Y1 = fmaxf(n.Y1, fmaxf(y, 0.0f));
            n.Z0 = fminf(n.Z0, z); n.Z1 = fmaxf(n.Z1, z);
        }
    }
    nodes[b] = n;
}

__global__ void __launch_bounds__(256) k_cloud_parents(const CloudNode* __restrict__ child, int n_child, CloudNode* __restrict__ parent, int n_parent)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_parent) return;
    CloudNode n;
    node_clear(n);
    for (int k = 0; k < kCloudFan; k++) {
        const int c = p * kCloudFan + k;
        if (c >= n_child) break;
        const CloudNode m = child[c];
        n.flags |= m.flags;
        n.px0 = fminf(n.px0, m.px0); n.px1 = fmaxf(n.px1, m.px1);
        n.py0 = fminf(n.py0, m.py0); n.py1 = fmaxf(n.py1, m.py1);
        n.z0 = fminf(n.z0, m.z0); n.z1 = fmaxf(n.z1, m.z1);
        n.X0 = fminf(n.X0, m.X0); n.X1 = fmaxf(n.X1, m.X1);
        n.Y0 = fminf(n.Y0, m.Y0); n.Y1 = fmaxf(n.Y1, m.Y1);
        n.Z0 = fminf(n.Z0, m.Z0); n.Z1 = fmaxf(n.Z1, m.Z1);
    }
    parent[p] = n;
}

CloudLayout cloud_layout(int N)
{
    CloudLayout L = cloud_layout_base(N);
    size_t tmp = 0;
    (void) rocprim::radix_sort_pairs(nullptr, tmp, (uint32_t*) nullptr, (uint32_t*) nullptr, (uint32_t*) nullptr, (uint32_t*) nullptr,
                                     (size_t) L.Np, 0, 32, (hipStream_t) nullptr);
    L.sort_tmp_bytes = tmp;
    L.total = L.sort_tmp + cloud_align(tmp);
    return L;
}

}  // namespace

extern "C" {

size_t kbe_cloud_pack_bytes(int N)
{
    return N < 0 ? 0 : cloud_layout(N).total;
}

int kbe_cloud_pack(const float* points, const float* image, const float* depth, int N, int W, int H, double focal, int raster_w,
                   int raster_n, void* packed, kbe_stream_t stream)
{
    if (!(raster_w > 0 && raster_n >= raster_w && raster_n <= N && raster_n % raster_w == 0)) raster_w = raster_n = 0;      // a hint, not a requirement
    KBE_REQUIRE(packed && N >= 0 && N <= (1 << 30) && W > 0 && H > 0 && focal > 0.0 && ((uintptr_t) packed & 255) == 0, "kbe_cloud_pack: bad arguments");
    KBE_REQUIRE(N == 0 || (points && image && depth), "kbe_cloud_pack: cloud pointers are NULL");
    const hipStream_t s = (hipStream_t) stream;
    const CloudLayout L = cloud_layout(N);
    char* b = (char*) packed;
    uint32_t* keys_in = (uint32_t*) (b + L.keys_in), *keys_out = (uint32_t*) (b + L.keys_out);
    uint32_t* idx_in = (uint32_t*) (b + L.idx_in), *idx_out = (uint32_t*) (b + L.idx_out);
    const unsigned grid = blocks_for((size_t) L.Np);
    hipLaunchKernelGGL(k_cloud_keys, dim3(grid), dim3(256), 0, s, points, N, L.Np, W, H, (float) focal, raster_n, keys_in, idx_in);
    size_t tmp = L.sort_tmp_bytes;
    const hipError_t e = rocprim::radix_sort_pairs((void*) (b + L.sort_tmp), tmp, keys_in, keys_out, idx_in, idx_out, (size_t) L.Np, 0, 32, s);
    if (e != hipSuccess) return fail(KBE_E_LAUNCH, "kbe_cloud_pack: radix_sort_pairs", e);
    hipLaunchKernelGGL(k_cloud_gather, dim3(grid), dim3(256), 0, s, points, image, depth, N, L.Np, idx_out, (CloudPoint*) (b + L.pd),
                       (CloudColour*) (b + L.col));
    hipLaunchKernelGGL(k_cloud_blocks, dim3(blocks_for((size_t) L.count[0])), dim3(256), 0, s, (const CloudPoint*) (b + L.pd), (float) focal,
                       (CloudNode*) (b + L.level[0]), L.count[0]);
    for (int l = 1; l < L.n_levels; l++)
        hipLaunchKernelGGL(k_cloud_parents, dim3(blocks_for((size_t) L.count[l])), dim3(256), 0, s, (const CloudNode*) (b + L.level[l - 1]),
                           L.count[l - 1], (CloudNode*) (b + L.level[l]), L.count[l]);
    return launched("kbe_cloud_pack");
}

}  // extern "C"

// shared with kbe_frame.hip: the view of a packed buffer
namespace kbe {
PackedCloud cloud_open(const void* packed, int N, double focal)
{
    return cloud_view(packed, cloud_layout_base(N), (float) focal);
}
}  // namespace kbe
