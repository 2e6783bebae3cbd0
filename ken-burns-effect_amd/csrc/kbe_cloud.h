// kbe_cloud.h -- the packed point cloud of the fused frame kernel: layout shared by kbe_cloud.hip (which builds it)
// and kbe_fused.hip (which renders from it).
//
// The reference keeps the cloud as three tensors in whatever order process_inpaint appended the points
// (common.py:176-179, :76-80) and scatters every point into the target raster with global atomics.  The fused
// kernel turns the scatter around: a target tile PULLS the points that can reach it.  For that the cloud is packed
// once per video (it is resident for all frames):
//   * points sorted by the Morton code of the 8 x 8-pixel cell they project to in the cloud's own view (the image
//     raster for the image pixels, their own view for the appended ones), cut into BLOCKS of 64 consecutive points:
//     a block is a spatially compact handful of neighbours -- for the image pixels one 8 x 8 cell of the raster, its four
//     SUB-BLOCKS of 16 points the cell's 4 x 4 quadrants.  Positions {x, y, z} and {r, g, b, depth} are kept as two arrays
//     of structures (12 + 16 bytes per point);
//   * per block a NODE: a box that bounds where its points can land in ANY view -- {p = (x, y) * Fd / z, z} for
//     ordinary points (the projection (p z + s Fd) F' / ((z + sz) Fd) is monotone in p and in z, so the four corners
//     of the box bound it), and an {x, y, z} box for degenerate points (z < 1, masked points at the origin, points
//     behind the camera);
//   * levels of nodes over 32 children each, up to a top level of at most 64 nodes, so that a tile finds its
//     ~28 candidate blocks among 18 k (1024^2) with four wave-wide rounds of node tests.
// Nothing of this changes a result: every point a tile's pixels can see passes the (conservative) node tests, and a
// point that passes without landing in the tile is dropped by the exact projection.
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace kbe {

constexpr int kCloudBlock = 64;         // points per block (one per lane)
#ifndef KBE_CLOUD_SUB
#define KBE_CLOUD_SUB 16
#endif
constexpr int kCloudSub = KBE_CLOUD_SUB; // points per SUB-BLOCK (a quarter of a block; 8 = an eighth, dev): the unit of a tile's candidate list (kbe_fused.hip)
constexpr int kCloudFan = 32;           // children per node
constexpr int kCloudTopMax = 64;        // nodes of the top level at most (one wave tests them in one go)
constexpr int kCloudMaxLevels = 6;      // 64 * 32^5 blocks: far beyond the 2^30-point limit of the frame loop

struct CloudNode {                      // 64 bytes
    float px0, px1, py0, py1, z0, z1;   // ordinary points: p = coordinate * Fd / z (pixels from the principal point), and z
    float X0, X1, Y0, Y1, Z0, Z1;       // degenerate points: camera-space box
    uint32_t flags;                     // bit 0: holds ordinary points, bit 1: holds degenerate points
    uint32_t pad[3];
};
static_assert(sizeof(CloudNode) == 64, "node size");

struct CloudPoint { float x, y, z; };            // 12 bytes, tightly packed: streamed once per frame by the placement launch
struct CloudColour { float r, g, b, depth; };   // 16 bytes: one 128-bit load per record in the tile launch
static_assert(sizeof(CloudPoint) == 12 && sizeof(CloudColour) == 16, "packed point layout");

struct PackedCloud {                    // passed to the kernels by value
    const CloudPoint* pd;               // [Np]  position
    const CloudColour* col;             // [Np]  colour and the depth channel (28 bytes per point in all: what the reference's tensors hold)
    int Np;                             // points incl. padding (a multiple of 64; padding has z = NaN and is culled)
    int n_levels;                       // level 0 = blocks
    int count[kCloudMaxLevels];
    const CloudNode* level[kCloudMaxLevels];
    float fd;                           // the focal length p is expressed in
};

struct CloudLayout {                    // byte offsets inside the caller's `packed` buffer, a pure function of N
    int Np, n_levels;
    int count[kCloudMaxLevels];
    size_t pd, col, level[kCloudMaxLevels], keys_in, keys_out, idx_in, idx_out, sort_tmp, sort_tmp_bytes, total;
};

inline size_t cloud_align(size_t v) { return (v + 255) & ~(size_t) 255; }

// everything but sort_tmp_bytes / total, which need rocprim (kbe_cloud.hip adds them)
inline CloudLayout cloud_layout_base(int N)
{
    CloudLayout L = {};
    const int n = N > 0 ? N : 0;
    L.Np = ((n + kCloudBlock - 1) / kCloudBlock) * kCloudBlock;
    if (L.Np == 0) L.Np = kCloudBlock;                      // an empty cloud is one block of padding
    int c = L.Np / kCloudBlock, lv = 0;
    L.count[0] = c;
    while (c > kCloudTopMax && lv + 1 < kCloudMaxLevels) {
        c = (c + kCloudFan - 1) / kCloudFan;
        L.count[++lv] = c;
    }
    L.n_levels = lv + 1;
    size_t o = 0;
    L.pd = o;    o += cloud_align(sizeof(CloudPoint) * (size_t) L.Np);
    L.col = o;   o += cloud_align(sizeof(CloudColour) * (size_t) L.Np);
    for (int l = 0; l < L.n_levels; l++) { L.level[l] = o; o += cloud_align(sizeof(CloudNode) * (size_t) L.count[l]); }
    L.keys_in = o;  o += cloud_align(4 * (size_t) L.Np);
    L.keys_out = o; o += cloud_align(4 * (size_t) L.Np);
    L.idx_in = o;   o += cloud_align(4 * (size_t) L.Np);
    L.idx_out = o;  o += cloud_align(4 * (size_t) L.Np);
    L.sort_tmp = o;
    return L;
}

inline PackedCloud cloud_view(const void* packed, const CloudLayout& L, float fd)
{
    PackedCloud pc = {};
    const char* b = (const char*) packed;
    pc.pd = (const CloudPoint*) (b + L.pd);
    pc.col = (const CloudColour*) (b + L.col);
    pc.Np = L.Np;
    pc.n_levels = L.n_levels;
    for (int l = 0; l < L.n_levels; l++) { pc.count[l] = L.count[l]; pc.level[l] = (const CloudNode*) (b + L.level[l]); }
    pc.fd = fd;
    return pc;
}

}  // namespace kbe
