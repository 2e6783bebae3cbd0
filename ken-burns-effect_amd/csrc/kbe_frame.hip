// kbe_frame.hip -- the per-frame hot path on the resident point cloud (include/kbe.h, "The frame
// loop of process_kenburns"): project + z-splat + bucket -> tile gather -> hole fill.
//
// Design (MI355X).  The reference scatters every point into a global z-buffer (float CAS loop)
// and then into 5 global accumulator planes, 20 float atomics per point (common.py:435-507,
// :586-669).  Measured on gfx950: L2 float atomics retire ~0.2 T/s (95 us per 1024^2 frame for
// the accumulation alone) and LDS float atomics cost ~49 cycles per wave-level ds_add_f32 (a
// tiled LDS-accumulator version kept every CU's LDS pipe busy for ~90 us).  So the scatter is
// turned into a gather:
//   k_project  one thread per point: shift (common.py:104-109), project (:447-484), ONE native
//              atomic umin on the order-preserving key of dblError into the z-buffer (:486-506),
//              and a 16-byte record {ox, oy, dblError, index} appended to the bucket of every
//              32x16 target tile one of its four corners lies in (appends are aggregated per
//              wave: one counter atomic per distinct tile, records stored coalesced);
//   k_tiles    one workgroup per tile: z-buffer tile + halo -> LDS, degrid (:525-568) in LDS,
//              records -> per-pixel linked lists in LDS (bin = north-west corner; one
//              ds_wrxchg per record), then every pixel walks the 4 bins that can reach it,
//              z-tests (:639) and accumulates (:641) in registers, normalises (:686), applies
//              the hole mask (:253), converts to uint8 (:255) and stores coalesced;
//   k_fill_holes  the hole list (:838-924) with an exact branch-and-bound over the 16 directions, one
//              half-wave per hole or (frames with very many holes) one lane per hole; also resets z-buffer
//              and bucket counters;
//   k_tiles_nc the same tile machinery for render_pointcloud with any channel count (4 channels at a time);
//   kbe_render_video  the whole loop enqueued from C, consecutive frames on several streams ("lanes").
// No accumulator or float render ever exists in HBM and no floating-point atomic is executed.
// What bounds these kernels is instruction issue, not bandwidth: the code below is
// written branch-free where lanes mostly agree and with wave-uniform work kept on the scalar unit.
//
// Numerics are those of oracle/kbe_oracle.c: the z-buffer is bit-exact (min commutes), degrid
// is the out-of-place schedule, accumulation order is bucket order (not point order).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>
#include <type_traits>

#include "kbe.h"
#include "kbe_cloud.h"
#include "kbe_device.h"
#include "kbe_fill.h"
#include "kbe_host.h"

#pragma clang fp contract(off)

using namespace kbe;

namespace kbe { PackedCloud cloud_open(const void* packed, int N, double focal); }      // kbe_cloud.hip

namespace {

#ifndef KBE_TILE_W
#define KBE_TILE_W 32
#endif
#ifndef KBE_TILE_H
#define KBE_TILE_H 16
#endif
#ifndef KBE_TILE_THREADS
#define KBE_TILE_THREADS 256
#endif
#ifndef KBE_TILE_CAP
#define KBE_TILE_CAP 768
#endif
#ifndef KBE_BUCKET_FACTOR
#define KBE_BUCKET_FACTOR 12
#endif
constexpr int TW = KBE_TILE_W, TH = KBE_TILE_H;     // target tile owned by one workgroup
// strip tables of the hole fill (k_hole_dist): per fill direction, W + H + 8 lines of (lo, hi); built from the extents of
// up to STRIP_TILES tile rows / columns
constexpr int STRIP_TILES = 512;
constexpr int KW = TW + 2, KH = TH + 2;             // tile + the 1-px halo whose z the degrid reads
constexpr int BW = TW + 1, BH = TH + 1;             // bins: north-west corners x0-1 .. x0+TW-1, y0-1 .. y0+TH-1
constexpr int TILE_THREADS = KBE_TILE_THREADS;
constexpr int PIX_PER_THREAD = TW * TH / TILE_THREADS;
constexpr int REC_CAP = KBE_TILE_CAP;               // records a tile holds in LDS at once (more: several rounds)
constexpr int BUCKET_CAP = KBE_BUCKET_FACTOR * TW * TH;     // records a tile's bucket holds in HBM (more: brute force)
#ifndef KBE_BUCKET_PAD
#define KBE_BUCKET_PAD 272
#endif
constexpr int BUCKET_STRIDE = BUCKET_CAP + KBE_BUCKET_PAD;  // records between two buckets: NOT a power-of-two multiple, or the
                                                    // live head of every bucket lands on the same few HBM channels
constexpr int CNT_STRIDE = 32;                      // ints between two bucket counters: one 128-byte line each, so that
                                                    // the counter atomics of neighbouring tiles do not serialise in L2
static_assert(TW * TH % TILE_THREADS == 0 && TILE_THREADS % 64 == 0 && REC_CAP >= TILE_THREADS && TW % 32 == 0, "tile geometry");

struct Scratch {                            // carve-out of the caller's scratch allocation
    uint32_t* zkeys;        // [H*W]  z-buffer as order-preserving keys; KBE_ZKEY_EMPTY between frames
    uint32_t* zkeys_b;      // [H*W]  second z-buffer: consecutive frames of a video alternate, each clearing the other's in its tile launch
    uint8_t* dist;          // [H*W]  Chebyshev distance to the nearest valid pixel, capped (frames with very many holes: k_hole_dist)
    float2* strips;         // [16][W + H + 8]  per fill direction and line across the image: where along it valid pixels can be (k_hole_dist)
    uint8_t* dist_blocks;   // [tiles_y * TH / 8][tiles_x * TW / 8]  the same distance between 8 x 8 blocks, in blocks
    int* tile_count;        // [n_tiles * CNT_STRIDE]  records appended to each bucket; 0 between frames
    int* hole_count;        // [1]
    int4* bbox;             // [n_tiles]: per tile, x0, y0, x1, y1 of its valid pixels (inclusive; empty: x0 > x1); plain stores
    uint32_t* coarse;       // [n_tiles]: bit (cy * (TW/8) + cx) = the 8x8 block (cx, cy) of the tile holds a valid pixel
    int* holes;             // [H*W]
    float* depth;           // [H*W]  render[3] * (existing > 0): the fill compares the two ends of a ray with it
    uint32_t* mask;         // [H][ceil(W/32)]  bit = depth > 0: what the fill walks on (32x smaller than the plane)
    float4* buckets;        // [n_tiles][BUCKET_STRIDE]  {ox, oy, dblError, point index}
    int tiles_x, tiles_y;
};

inline size_t align16(size_t v) { return (v + 15) & ~(size_t) 15; }

Scratch carve(void* base, int W, int H)
{
    char* p = (char*) base;
    const size_t hw = (size_t) W * H;
    Scratch s;
    s.tiles_x = (W + TW - 1) / TW;
    s.tiles_y = (H + TH - 1) / TH;
    const size_t n_tiles = (size_t) s.tiles_x * s.tiles_y;
    s.zkeys = (uint32_t*) p;      p += align16(4 * hw);
    s.tile_count = (int*) p;      p += align16(4 * n_tiles * CNT_STRIDE);
    s.hole_count = (int*) p;      p += 16;
    s.bbox = (int4*) p;           p += align16(16 * n_tiles);
    s.coarse = (uint32_t*) p;     p += align16(4 * n_tiles);
    s.holes = (int*) p;           p += align16(4 * hw);
    s.depth = (float*) p;         p += align16(4 * hw);
    s.mask = (uint32_t*) p;       p += align16(4 * (size_t) H * ((W + 31) / 32));
    s.zkeys_b = (uint32_t*) p;    p += align16(4 * hw);
    s.dist = (uint8_t*) p;        p += align16(hw);
    s.strips = (float2*) p;       p += align16(8 * 16 * (size_t) (W + H + 8));
    s.dist_blocks = (uint8_t*) p; p += align16(n_tiles * (TW / 8) * (TH / 8));
    s.buckets = (float4*) p;
    return s;
}

size_t scratch_bytes(int W, int H)
{
    const size_t hw = (size_t) W * H;
    const size_t n_tiles = (size_t) ((W + TW - 1) / TW) * ((H + TH - 1) / TH);
    return align16(4 * hw) + align16(4 * n_tiles * CNT_STRIDE) + 16 + align16(16 * n_tiles) + align16(4 * n_tiles) + align16(4 * hw) + align16(4 * hw) +
           align16(4 * (size_t) H * ((W + 31) / 32)) + align16(4 * hw) + align16(hw) + align16(8 * 16 * (size_t) (W + H + 8)) + align16(n_tiles * (TW / 8) * (TH / 8)) + n_tiles * BUCKET_STRIDE * sizeof(float4);
}


__global__ void k_scratch_init(uint32_t* zkeys, uint32_t* zkeys_b, size_t hw, int* tile_count, int n_tiles, int* hole_count)
{
    const size_t stride = (size_t) gridDim.x * blockDim.x, gtid = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    for (size_t i = gtid; i < hw; i += stride) { zkeys[i] = KBE_ZKEY_EMPTY; if (zkeys_b) zkeys_b[i] = KBE_ZKEY_EMPTY; }
    for (size_t i = gtid; i < (size_t) n_tiles; i += stride) tile_count[i * CNT_STRIDE] = 0;
    if (gtid == 0) { hole_count[0] = 0; hole_count[1] = 0; }
}

// ---------------------------------------------------------------------------------------
// launch 1: project every point once
// ---------------------------------------------------------------------------------------
struct ProjectArgs {
    const float* points;    // [3,N]
    int N;
    int raster_w, raster_n; // hint: the first raster_n points are a row-major raster raster_w wide (0: unknown)
    Camera cam;
    uint32_t* zkeys;
    int* tile_count;
    float4* buckets;
    int tiles_x, tiles_y;
    int* hole_count;
    int dense;              // more than two points per target pixel: pre-reduce the z-splat within the wave
    int buckets_32bit;      // every bucket ends below byte 2^32 of `buckets`
};

// Groups the lanes of a wave by target tile: for a lane that `want`s, `same` is the mask of the
// lanes wanting the same tile and `leader` its lowest lane.  Pure cross-lane work (ballots,
// shuffles), no memory traffic; one loop trip per distinct tile (1-3 for coherent points).
struct TileGroup { unsigned long long same; int leader; };

__device__ __forceinline__ TileGroup group_by_tile(bool want, int tile)
{
    TileGroup g = { 0ull, 0 };
    unsigned long long pending = __ballot(want);
    while (pending) {                                           // wave-uniform
        const int leader = __ffsll((long long) pending) - 1;
        const int t = __builtin_amdgcn_readlane(tile, leader);     // leader is wave-uniform: v_readlane, not an LDS round trip (ds_bpermute)
        const unsigned long long same = __ballot(want && tile == t);
        if (want && tile == t) { g.same = same; g.leader = leader; }
        pending &= ~same;
    }
    return g;
}

// The groups of the east spills follow from the groups of the own tiles: lanes that share an own tile share its
// east neighbour, so a group's spilling lanes are `same & ballot(spills)` and no second grouping loop is needed --
// except for lanes whose own tile is outside the image (corner at -1) but whose east tile is inside.
__device__ __forceinline__ TileGroup east_groups(const TileGroup& own, bool want_own, bool want_east, int east_tile)
{
    const unsigned long long sp = __ballot(want_east);
    TileGroup g = group_by_tile(want_east && !want_own, east_tile);            // normally no lane: the loop does not run
    if (want_east && want_own) {
        g.same = own.same & sp;
        g.leader = __ffsll((long long) g.same) - 1;
    }
    return g;
}

constexpr int UNIT = 64;                // points per wave unit: one per lane (4 per lane needed 98-118 VGPRs, halved the
                                        // occupancy and doubled the time of this kernel; 2 per lane measured 9 % slower)
constexpr int PATCH_ROWS = 2;           // a raster unit is a 32 x 2 patch

// One wave handles units of 64 points, one per lane.  Per unit: load, shift (common.py:104-109), project
// (:447-468), then -- as soon as the image position is known -- the bucket bookkeeping: the point goes to the
// bucket of the tile of its north-west corner (e = 0) and, when that corner sits in a tile's last column (or at
// -1, just outside), also to the east neighbour (e = 1); the lanes of a wave share very few target tiles, so they
// are grouped and one leader per tile bumps the counter for all of them.  A returning global atomic is a ~2 us
// round trip (probe: with the results unused this launch is 3.8 us shorter), so all counter atomics of the round
// are issued back to back and the rest of the point's work -- weights (:472-484), dblError (:470), winner corner
// and the z-splat atomic umin (:486-506) -- is done while they are in flight; only then are the results consumed
// and the 16-byte records stored.  Points whose corner also sits in a tile's last ROW need a second round
// (south, south-east); ~6 % of the waves of a raster.
#ifndef KBE_PROJECT_BLOCK
#define KBE_PROJECT_BLOCK 64        // one wave per workgroup: fits the gaps other lanes' kernels leave (29.7 vs 30.6 us per frame at 256)
#endif
__global__ void __launch_bounds__(KBE_PROJECT_BLOCK) k_project(ProjectArgs a)
{
    const int lane = threadIdx.x & 63;
    // wave-uniform values are made scalar explicitly (the unit -> point index arithmetic below then runs on the
    // scalar unit, once per wave, in 32 bits; as vector 64-bit arithmetic it was a sixth of this kernel's instructions)
    const int wave = __builtin_amdgcn_readfirstlane((int) ((blockIdx.x * blockDim.x + threadIdx.x) >> 6));
    const int n_waves = (int) ((gridDim.x * blockDim.x) >> 6);
    const Camera& cam = a.cam;
    if (blockIdx.x == 0 && threadIdx.x == 0) *a.hole_count = 0;
    const size_t N = (size_t) a.N;
    // Work units.  Where the cloud is known to start with a row-major raster (the image pixels), a unit is a
    // 32 x 2 patch of it rather than 64 consecutive pixels of a row: its points then fall into one or two target
    // tiles, and a bucket's records reference neighbouring points.  Pure speed hint.
    const unsigned patches_x = (a.raster_w > 0 && a.raster_w % 32 == 0) ? (unsigned) a.raster_w / 32u : 0u;
    const unsigned patch_rows = patches_x ? ((unsigned) a.raster_n / (unsigned) a.raster_w) / PATCH_ROWS : 0u;
    const unsigned n_patches = patches_x * patch_rows;                  // <= N / UNIT
    const unsigned lin0 = n_patches * UNIT;                             // points before lin0 are covered by patches
    const unsigned n_units = n_patches + ((unsigned) a.N - lin0 + UNIT - 1) / UNIT;
    for (unsigned unit = (unsigned) wave; unit < n_units; unit += (unsigned) n_waves) {
        unsigned i;
        if (unit < n_patches) {
            // the patch's first point on the scalar unit; a lane adds its row (0 or raster_w) and column
            const unsigned pyb = unit / patches_x, pxb = unit - pyb * patches_x;
            static_assert(PATCH_ROWS == 2, "a lane's patch row is lane >> 5");
            i = (pyb * PATCH_ROWS * (unsigned) a.raster_w + pxb * 32u) + ((lane >> 5) ? (unsigned) a.raster_w : 0u) + (unsigned) (lane & 31);
        } else {
            i = lin0 + (unit - n_patches) * UNIT + (unsigned) lane;
        }
        bool ok = i < (unsigned) a.N;
        float x = 0.0f, y = 0.0f, z = 0.0f, ox = 0.0f, oy = 0.0f;
        if (ok) {
            const uint32_t off = i << 2;                                // N <= 2^30: a 32-bit byte offset on three uniform bases
            x = *(const float*) ((const char*) a.points + off);
            y = *(const float*) ((const char*) (a.points + N) + off);
            z = *(const float*) ((const char*) (a.points + 2 * N) + off);
            apply_shift(cam, x, y, z);
            ok = project_xy(cam, x, y, z, ox, oy);
        }
        Proj p;
        p.nwx = (int) floorf(ox); p.nwy = (int) floorf(oy);
        ok = ok && ((unsigned) (p.nwx + 1) <= (unsigned) cam.W) & ((unsigned) (p.nwy + 1) <= (unsigned) cam.H);      // touches the image at all: -1 <= nw < size
        const bool spx = ok && ((p.nwx + 1) % TW == 0), spy = ok && ((p.nwy + 1) % TH == 0);
        static_assert((TW & (TW - 1)) == 0 && (TH & (TH - 1)) == 0, "tile sizes are powers of two");
        const int tx0 = p.nwx >> __builtin_ctz(TW), ty0 = p.nwy >> __builtin_ctz(TH);              // floor division: -1 for nw == -1

        // round 0: own tile and east neighbour; the counter atomics go out now
        TileGroup grp[2];
        int tgt[2], base[2];
        bool want[2];
#pragma unroll
        for (int e = 0; e < 2; e++) {
            const int tx = tx0 + e;
            want[e] = ok && (e == 0 || spx) && ((unsigned) tx < (unsigned) a.tiles_x) & ((unsigned) ty0 < (unsigned) a.tiles_y);
            tgt[e] = __mul24(ty0, a.tiles_x) + tx;                      // 24-bit multiply: full rate (the 32-bit one is quarter rate)
            base[e] = 0;
        }
        grp[0] = group_by_tile(want[0], tgt[0]);
        grp[1] = east_groups(grp[0], want[0], want[1], tgt[1]);
#pragma unroll
        for (int e = 0; e < 2; e++)
            if (want[e] && lane == grp[e].leader) base[e] = atomicAdd(&a.tile_count[(uint32_t) tgt[e] * CNT_STRIDE], __popcll(grp[e].same));

        // ... and while they are in flight: weights, dblError, winner corner, z-splat
        float err = 0.0f;
        int zidx = -1;
        if (ok) {
            project_weights(ox, oy, p);
            err = project_err_fast(cam, z);
            const int k = winner_corner(p);                             // common.py:486-506
            if (k >= 0) {
                const int cx = p.nwx + (k & 1), cy = p.nwy + (k >> 1);
                if (inside(cx, cy, cam.W, cam.H)) zidx = __mul24(cy, cam.W) + cx;
            }
        }
        if (a.dense) {
            // a cloud denser than the target raster (BASELINE configs[4]: 4 points per pixel): the 2 x 2 source
            // neighbours (lanes ^1, ^32, ^33 of a 32 x 2 patch) mostly splat onto the same pixel and their atomics
            // would serialise on one address (measured: 8x the time per atomic); the lowest lane of those that agree
            // issues one atomic with their minimum
            uint32_t key = zkey_encode(err);
            bool issue = zidx >= 0;
#pragma unroll
            for (int m = 0; m < 3; m++) {
                const int mask = m == 0 ? 1 : (m == 1 ? 32 : 33);
                const int pidx = __shfl_xor(zidx, mask);
                const uint32_t pkey = (uint32_t) __shfl_xor((int) key, mask);
                if (zidx >= 0 && pidx == zidx) {
                    key = min(key, pkey);
                    if ((lane ^ mask) < lane) issue = false;
                }
            }
            if (issue) atomicMin(&a.zkeys[(uint32_t) zidx], key);
        } else if (zidx >= 0) {
            atomicMin(&a.zkeys[(uint32_t) zidx], zkey_encode(err));
        }
        const float4 rec = make_float4(ox, oy, err, __int_as_float((int) i));
        // all buckets within 4 GB (frames up to 4096 x 4096): a 32-bit byte offset from a 24-bit multiply on the
        // uniform base; otherwise 64-bit arithmetic (a quarter-rate multiply-add)
        auto store_record = [&](int tile, int slot) {
            static_assert(BUCKET_STRIDE * 16 < (1 << 24), "the bucket stride in bytes is a 24-bit factor");
            if (a.buckets_32bit) *(float4*) ((char*) a.buckets + (__umul24((uint32_t) tile, (uint32_t) BUCKET_STRIDE * 16u) + ((uint32_t) slot << 4))) = rec;
            else a.buckets[(size_t) tile * BUCKET_STRIDE + slot] = rec;
        };
#pragma unroll
        for (int e = 0; e < 2; e++) {
            const int b0 = __shfl(base[e], grp[e].leader);
            if (want[e]) {
                const int slot = b0 + __popcll(grp[e].same & ((1ull << lane) - 1ull));
                if (slot < BUCKET_CAP) store_record(tgt[e], slot);      // beyond: the tile sees count > cap
            }
        }
        // round 1 (rare): south and south-east neighbours
        if (__ballot(spy) != 0ull) {                                    // wave-uniform
#pragma unroll
            for (int e = 0; e < 2; e++) {
                const int tx = tx0 + e, ty = ty0 + 1;
                want[e] = spy && (e == 0 || spx) && ((unsigned) tx < (unsigned) a.tiles_x) & ((unsigned) ty < (unsigned) a.tiles_y);
                tgt[e] = __mul24(ty, a.tiles_x) + tx;
                base[e] = 0;
            }
            grp[0] = group_by_tile(want[0], tgt[0]);
            grp[1] = east_groups(grp[0], want[0], want[1], tgt[1]);
#pragma unroll
            for (int e = 0; e < 2; e++)
                if (want[e] && lane == grp[e].leader) base[e] = atomicAdd(&a.tile_count[(uint32_t) tgt[e] * CNT_STRIDE], __popcll(grp[e].same));
#pragma unroll
            for (int e = 0; e < 2; e++) {
                const int b0 = __shfl(base[e], grp[e].leader);
                if (want[e]) {
                    const int slot = b0 + __popcll(grp[e].same & ((1ull << lane) - 1ull));
                    if (slot < BUCKET_CAP) store_record(tgt[e], slot);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// launch 2: the tile kernel
// ---------------------------------------------------------------------------------------
struct TileArgs {
    const float* points;    // [3,N]  (only the brute-force path of an overflowing bucket reads it)
    const float* image;     // [3,N]
    const float* depth_in;  // [N]
    int N;
    Camera cam;
    const uint32_t* zkeys;
    const int* tile_count;
    const float4* buckets;
    int tiles_x, tiles_y;
    uint32_t* zkeys_clear;  // optional: the OTHER z-buffer, whose pixels of this tile are reset here (and this tile's bucket counter)
    int* tile_count_clear;
    uint8_t* frame;         // [H,W,3]
    float* depth;           // [H*W]
    uint32_t* mask;         // [H][ceil(W/32)]
    int* holes;
    int* hole_count;
    int4* bbox;
    uint32_t* coarse;
    float* render;          // optional [4,H,W] (unfilled; the fill kernel patches the holes)
    float* existing;        // optional [H*W]
    float* zee;             // optional [H*W] degridded z-buffer
    float* zee_pre;         // optional [H*W] pre-degrid z-buffer
};

// blockIdx -> tile id such that each XCD (block b runs on XCD b % 8) owns a contiguous band of
// tile rows: the records of neighbouring tiles reference neighbouring points (shared L2 lines).
__device__ __forceinline__ int xcd_tile(int b, int n)
{
    const int xcd = b & 7, j = b >> 3, q = n >> 3, r = n & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
}

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
// a pixel's five accumulators; r|g and b|depth as pairs so that they update with packed fp32 instructions
struct PixAcc { f2 rg, bd; float w; };

struct TileLds {
    float4 rec[REC_CAP + 1];    // ox, oy, dblError, link to the next record of the bin; slot REC_DUMMY: see gather
    float4 rgbd[REC_CAP + 1];   // the point's r, g, b, depth, fetched once at insert time
    int head[BH * BW];          // link to the first record of each bin.  A link is the record's BYTE offset, REC_NULL = none
    float zpre[KH * KW];        // z-buffer before degrid, tile + halo; after the degrid: uint8 staging area + per-wave partials
    float zee[TH * TW];         // degridded z-buffer
    int nrec;
    int odd_z[TILE_THREADS / 64];   // per wave: some z of tile + halo is outside [2^19, 1e6] (then: the generic, fp64-capable code)
};

constexpr int REC_DUMMY = REC_CAP;          // what an exhausted list reads: dblError = +inf (fails every z test), colours 0, its own successor
constexpr int REC_NULL = REC_DUMMY * 16;    // "no record" as a link: the dummy's byte offset, so that every link can be read as it is

__device__ __forceinline__ void lds_dummy_record(TileLds& L)
{
    L.rec[REC_DUMMY] = make_float4(0.0f, 0.0f, __builtin_inff(), __int_as_float(REC_NULL));
    L.rgbd[REC_DUMMY] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
}

// threads one record into the list of its bin (bin = north-west corner relative to x0-1, y0-1)
__device__ __forceinline__ void lds_insert(TileLds& L, int idx, float ox, float oy, float err, const float4& rgbd, int x0, int y0)
{
    const int bx = (int) floorf(ox) - (x0 - 1), by = (int) floorf(oy) - (y0 - 1);
    L.rgbd[idx] = rgbd;
    const int next = atomicExch(&L.head[__mul24(by, BW) + bx], idx << 4);
    L.rec[idx] = make_float4(ox, oy, err, __int_as_float(next));
}

__device__ __forceinline__ float4 fetch_rgbd(const TileArgs& a, int id)
{
    // uniform plane bases + one 32-bit byte offset per record (N <= 2^30): the loads take the scalar-base form and
    // the lane computes a single shift instead of four 64-bit address additions
    const uint32_t off = (uint32_t) id << 2;
    const char* r = (const char*) a.image;
    const char* g = (const char*) (a.image + (size_t) a.N);
    const char* b = (const char*) (a.image + 2 * (size_t) a.N);
    const char* d = (const char*) a.depth_in;
    return make_float4(*(const float*) (r + off), *(const float*) (g + off), *(const float*) (b + off), *(const float*) (d + off));
}

// z-tested bilinear accumulation (common.py:586-669) of the records now in LDS, in registers.
// Everything the walk touches is in LDS (a variant that fetched r, g, b, depth from global memory per
// (pixel, record) pair spent ~13 us of the launch on those dependent loads).
// The launch is bound by instruction issue and LDS latency (PMC: the SIMDs issue ~85 % of the time, a wave
// waits ~46 % of its life), so the walk is branch-free and as parallel as the data allows: the heads of the four
// bins that can reach a pixel are read together, then one record of EACH bin together, and a record that
// fails the z test contributes with weight 0 -- adding c * 0 leaves the accumulator bits unchanged, so the sums
// are those of the branching loop.  A bin that has run out reads the dummy record, which fails the z test by
// itself and links to itself: no "is this a record" test per (pixel, record), and a link is the byte offset
// the LDS read takes as it is.  The
// trip count is the longest of the four lists, not their sum.  FAST: every z of the tile is in the band where
// `zee + 1.0` is exact in fp32 (plus_one_is_exact); otherwise the comparison runs in fp64 where it has to.
template <bool FAST, class Args>
__device__ __forceinline__ void gather(const Args& a, const TileLds& L, int tid, int x0, int y0,
                                       PixAcc (&acc)[PIX_PER_THREAD])
{
#pragma unroll
    for (int m = 0; m < PIX_PER_THREAD; m++) {
        const int q = tid + m * TILE_THREADS;
        const int ly = q / TW, lx = q - ly * TW;
        if (!inside(x0 + lx, y0 + ly, a.cam.W, a.cam.H)) continue;
        const float zee = L.zee[q];
        const bool exact = FAST || plus_one_is_exact(zee);                  // then zee + 1.0f IS the double sum
        const float zlimf = zee + 1.0f;
        const double zlim = (double) zee + 1.0;
        const float Xf = (float) (x0 + lx), Yf = (float) (y0 + ly);
        // corner k of a point is this pixel  <=>  its north-west corner is (X - (k & 1), Y - (k >> 1)).  That pins
        // floor(ox), floor(oy), so the bilinear weight of common.py:481-484 needs two subtractions and one
        // product: (ex - ox | ox - fx) * (ey - oy | oy - fy) with fx = (float) nwx, ex = (float) (nwx + 1).
        auto add = [&](int k, const float4& r, const float4& c) {
            const bool pass = exact ? (r.z <= zlimf) : ((double) r.z <= zlim);             // :639
            const float wx = (k & 1) ? (r.x - (Xf - 1.0f)) : ((Xf + 1.0f) - r.x);          // k & 1 ? ox - fx : ex - ox
            const float wy = (k >> 1) ? (r.y - (Yf - 1.0f)) : ((Yf + 1.0f) - r.y);
            const float w = pass ? wx * wy : 0.0f;
            const f4 cv = *(const f4*) &c;
            acc[m].rg += cv.xy * w;                                         // :641 product rounded, then added (v_pk_mul_f32, v_pk_add_f32)
            acc[m].bd += cv.zw * w;
            acc[m].w += w;                                                  // the `ones` channel (:429)
        };
        int nx[4];
#pragma unroll
        for (int k = 0; k < 4; k++) nx[k] = L.head[(ly + 1 - (k >> 1)) * BW + (lx + 1 - (k & 1))];
        do {
            float4 r[4], c[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                r[k] = *(const float4*) ((const char*) L.rec + nx[k]);
                c[k] = *(const float4*) ((const char*) L.rgbd + nx[k]);
            }
#pragma unroll
            for (int k = 0; k < 4; k++) {
                add(k, r[k], c[k]);
                nx[k] = __float_as_int(r[k].w);
            }
        } while (min(min(nx[0], nx[1]), min(nx[2], nx[3])) < REC_NULL);      // some list goes on
    }
}


// degrid (common.py:525-568), out of place: L.zpre (tile + halo, decoded) -> L.zee.  `fast`: every z of tile + halo in
// [2^19, 1e6] (any scene whose points are farther than F*B/475712 from the camera) -> fp32-only, branch-free
template <class Args>
__device__ __forceinline__ void tile_degrid(const Args& a, TileLds& L, int tid, int x0, int y0, bool fast)
{
    const int W = a.cam.W, H = a.cam.H;
    if (fast && !a.zee_pre) {
#pragma unroll
        for (int u = 0; u < PIX_PER_THREAD; u++) {
            const int i = tid + u * TILE_THREADS;
            const int ly = i / TW, lx = i - ly * TW;
            const float* z = &L.zpre[(ly + 1) * KW + (lx + 1)];
            const float nb_a[4] = { z[1], z[KW], z[KW + 1], z[1 - KW] };            // (+1, 0) (0, +1) (+1, +1) (+1, -1)
            const float nb_d[4] = { z[-1], z[-KW], z[-KW - 1], z[KW - 1] };         // their mirror images
            const float zd = degrid_pixel_fast(z[0], nb_a, nb_d);
            L.zee[i] = zd;                                                          // pixels past the image edge: never read
            if (a.zee && x0 + lx < W && y0 + ly < H) a.zee[(size_t) (y0 + ly) * W + x0 + lx] = zd;
        }
    } else {
        for (int i = tid; i < TH * TW; i += TILE_THREADS) {
            const int ly = i / TW, lx = i - ly * TW;
            const int x = x0 + lx, y = y0 + ly;
            if (x >= W || y >= H) continue;
            auto at = [&](int xx, int yy) { return L.zpre[(yy - y0 + 1) * KW + (xx - x0 + 1)]; };
            const float zd = degrid_pixel(x, y, W, H, at);
            L.zee[i] = zd;
            if (a.zee) a.zee[(size_t) y * W + x] = zd;
            if (a.zee_pre) a.zee_pre[(size_t) y * W + x] = at(x, y);
        }
    }

}

// resolve + store of a tile whose pixels hold their accumulated sums: normalise (common.py:686), hole mask (:253),
// uint8 (:255), validity bitmask / bounding box / coarse bits / hole list for the fill, coalesced stores
template <class Args>
__device__ __forceinline__ void tile_epilogue(const Args& a, TileLds& L, PixAcc (&acc)[PIX_PER_THREAD], int tile, int x0, int y0)
{
    const int tid = threadIdx.x, lane = tid & 63;
    const int W = a.cam.W, H = a.cam.H;
    // resolve: normalise (common.py:686), hole mask (:253), uint8 (:255)
    const size_t HW = (size_t) W * H;
    static_assert(sizeof(L.zpre) >= TW * TH * 3 + (64 + TILE_THREADS / 64) * sizeof(int), "uint8 staging + per-wave partials fit the z-buffer area");
    uint8_t* s_u8 = (uint8_t*) L.zpre;            // the pre-degrid z-buffer is dead since the barrier in front of the gather
    int* const s_part = (int*) L.zpre + TW * TH * 3 / 4;
    float res[PIX_PER_THREAD][4], dms[PIX_PER_THREAD];
    bool hole[PIX_PER_THREAD], valid[PIX_PER_THREAD];
    unsigned long long hm[PIX_PER_THREAD];
    int n_holes = 0;
#pragma unroll
    for (int m = 0; m < PIX_PER_THREAD; m++) {
        const int q = tid + m * TILE_THREADS;
        const int ly = q / TW, lx = q - ly * TW;
        const bool in = x0 + lx < W && y0 + ly < H;
        const float w = acc[m].w;
        const float den = w + 0.0000001f;
        // four numerators over one denominator: ONE IEEE division for the correctly rounded reciprocal, then
        // q = a * y, q' = fma(fma(-den, q, a), y, q) per channel -- the correctly rounded a / den (Markstein;
        // tests/markstein_div_check.c) unless an intermediate underflows, i.e. for |a| below ~2^-100, where the
        // last bit may differ (no colour or depth of a real cloud gets there)
        const float y = 1.0f / den;
        auto quot = [&](float a_) { const float q = a_ * y; return __builtin_fmaf(__builtin_fmaf(-den, q, a_), y, q); };
        res[m][0] = quot(acc[m].rg.x); res[m][1] = quot(acc[m].rg.y); res[m][2] = quot(acc[m].bd.x); res[m][3] = quot(acc[m].bd.y);
        dms[m] = res[m][3] * (w > 0.0f ? 1.0f : 0.0f);
        valid[m] = in && dms[m] > 0.0f;
        hole[m] = in && !(dms[m] > 0.0f);
        hm[m] = __ballot(hole[m]);
        n_holes += __popcll(hm[m]);
    }
    int vx0 = W, vy0 = H, vx1 = -1, vy1 = -1;
    uint32_t cbits = 0;
    const int wave_s = __builtin_amdgcn_readfirstlane(tid >> 6);
    static_assert((TW / 8) * (TH / 8) <= 32 && TH % 8 == 0, "the coarse block bits of a tile fit a word");
#pragma unroll
    for (int m = 0; m < PIX_PER_THREAD; m++) {
        const int q = tid + m * TILE_THREADS;
        const int ly = q / TW, lx = q - ly * TW;
        const int x = x0 + lx, y = y0 + ly;
        const bool in = x < W && y < H;
        s_u8[q * 3] = to_u8(res[m][0]); s_u8[q * 3 + 1] = to_u8(res[m][1]); s_u8[q * 3 + 2] = to_u8(res[m][2]);
        {   // validity bits: each 32-lane half of the wave holds 32 consecutive pixels of one row
            const unsigned long long vm = __ballot(valid[m]);
            if ((lane & 31) == 0 && in) a.mask[__umul24((uint32_t) y, (uint32_t) ((W + 31) >> 5)) + (uint32_t) (x >> 5)] = (uint32_t) (vm >> (lane & 32));
            // bounding box of the valid pixels (depth > 0), which lets the hole fill discard rays that can never hit
            // one: straight from the ballot, on the scalar unit (as a 6-step butterfly of 4 values it was 24
            // cross-lane operations per thread).  TW == 32: the low half of the wave is row `wrow`, the high half the next.
            static_assert(TW == 32, "a wave holds two tile rows");
            const uint32_t lo = (uint32_t) vm, hi = (uint32_t) (vm >> 32), any = lo | hi;
            if (any) {                                                  // wave-uniform
                const int wrow = y0 + (wave_s << 1) + m * (TILE_THREADS / TW);       // scalar: bounding box and block bits stay on the scalar unit
                vx0 = min(vx0, x0 + __builtin_ctz(any)); vx1 = max(vx1, x0 + 31 - __builtin_clz(any));
                vy0 = min(vy0, lo ? wrow : wrow + 1); vy1 = max(vy1, hi ? wrow + 1 : wrow);
                // which 8 x 8 blocks of the tile hold a valid pixel (the hole fill skips through blocks that do not)
                const uint32_t cols = (any & 0xFFu ? 1u : 0u) | (any & 0xFF00u ? 2u : 0u) | (any & 0xFF0000u ? 4u : 0u) | (any & 0xFF000000u ? 8u : 0u);
                cbits |= cols << ((TW / 8) * ((wrow - y0) >> 3));
            }
        }
        if (in) {
            // W * H <= 2^30: a 32-bit element index (24-bit multiply) and scalar plane bases instead of 64-bit vector arithmetic
            const uint32_t o = __umul24((uint32_t) y, (uint32_t) W) + (uint32_t) x;
            a.depth[o] = dms[m];
            if (a.render) { a.render[o] = res[m][0]; (a.render + HW)[o] = res[m][1]; (a.render + 2 * HW)[o] = res[m][2]; (a.render + 3 * HW)[o] = res[m][3]; }
            if (a.existing) a.existing[o] = acc[m].w;
        }
    }
    {
        // per-wave boxes meet in LDS, one plain 16-byte store per tile;
        // global atomics here -- even one cache line per tile row, even with a look first -- serialised so
        // badly across XCDs that they added 80-350 us per frame
        int* sb = s_part;
        if (lane == 0) { sb[4 * wave_s + 0] = vx0; sb[4 * wave_s + 1] = vy0; sb[4 * wave_s + 2] = vx1; sb[4 * wave_s + 3] = vy1; sb[64 + wave_s] = (int) cbits; }
    }
    __syncthreads();
    if (tid == 0) {
        const int* sb = s_part;
        int4 bb = make_int4(W, H, -1, -1);
        for (int w = 0; w < TILE_THREADS / 64; w++) {
            bb.x = min(bb.x, sb[4 * w]); bb.y = min(bb.y, sb[4 * w + 1]); bb.z = max(bb.z, sb[4 * w + 2]); bb.w = max(bb.w, sb[4 * w + 3]);
        }
        a.bbox[tile] = bb;
        uint32_t cb = 0;
        for (int w = 0; w < TILE_THREADS / 64; w++) cb |= (uint32_t) sb[64 + w];
        a.coarse[tile] = cb;
    }
    // uint8 rows leave as dwords when the row segment is 4-byte aligned and complete
    const bool dword_rows = (W & 3) == 0 && (TW * 3) % 4 == 0 && x0 + TW <= W;
    if (dword_rows) {
        constexpr int DW_PER_ROW = TW * 3 / 4;
        for (int i = tid; i < TH * DW_PER_ROW; i += TILE_THREADS) {
            const int ly = i / DW_PER_ROW, k = i - ly * DW_PER_ROW;
            if (y0 + ly >= H) continue;
            // byte offset < 3 * 2^30: 32 bits
            *(uint32_t*) (a.frame + ((__umul24((uint32_t) (y0 + ly), (uint32_t) W) + (uint32_t) x0) * 3u + 4u * (uint32_t) k)) = ((const uint32_t*) s_u8)[ly * DW_PER_ROW + k];
        }
    } else {
        for (int i = tid; i < TH * TW * 3; i += TILE_THREADS) {
            const int q = i / 3, ch = i - q * 3;
            const int ly = q / TW, lx = q - ly * TW;
            if (x0 + lx < W && y0 + ly < H) a.frame[((size_t) (y0 + ly) * W + x0 + lx) * 3 + ch] = s_u8[i];
        }
    }
    // The hole list last: ONE returning atomic per wave reserves its slots (a ~2 us round trip).  Anywhere earlier the
    // wave would sit in it in front of a barrier and hold up its whole workgroup; here it only delays its own exit.
    if (n_holes > 0) {                              // wave-uniform
        int base = 0;
        if (lane == 0) base = atomicAdd(a.hole_count, n_holes);
        base = __builtin_amdgcn_readfirstlane(base);
#pragma unroll
        for (int m = 0; m < PIX_PER_THREAD; m++) {
            const int q = tid + m * TILE_THREADS;
            const int ly = q / TW, lx = q - ly * TW;
            const int slot = base + __popcll(hm[m] & ((1ull << lane) - 1ull));
            // (the list holds W*H entries, enough for any one frame; the bound only matters when this launch is
            // repeated without the projection launch that zeroes the count, as bench.py does to time it alone)
            if (hole[m] && slot < W * H) a.holes[(uint32_t) slot] = (int) __umul24((uint32_t) (y0 + ly), (uint32_t) W) + x0 + lx;
            base += __popcll(hm[m]);
        }
    }
}

#ifndef KBE_TILE_WAVES
#define KBE_TILE_WAVES 4
#endif
#define KBE_TILE_ATTR amdgpu_waves_per_eu(KBE_TILE_WAVES, KBE_TILE_WAVES)
__global__ void __launch_bounds__(TILE_THREADS) __attribute__((KBE_TILE_ATTR)) k_tiles(TileArgs a)
{
    __shared__ TileLds L;

    const int tid = threadIdx.x, lane = tid & 63;
    const int tile = xcd_tile(blockIdx.x, gridDim.x);
    const int ty = tile / a.tiles_x, tx = tile - ty * a.tiles_x;
    const int x0 = tx * TW, y0 = ty * TH;
    const int W = a.cam.W, H = a.cam.H;

    // The launch is latency-bound, so the loads are ordered by what depends on them: the bucket count and this
    // thread's share of the first REC_CAP records first (the colour fetch needs the point indices in them),
    // then its share of the z-buffer tile; the colour loads are issued as soon as the records are in and fly
    // during the z-buffer decode, the first barrier and the degrid.  None of these loads sits under a branch:
    // the compiler's wait-count bookkeeping is per program point, and a load that MAY have been issued makes
    // every later wait on an older load a wait for everything (measured: the colour loads were waited for
    // in front of the degrid instead of behind it).
    const int count = a.tile_count[tile * CNT_STRIDE];
    const bool bucketed = count <= BUCKET_CAP;
    const float4* B = a.buckets + (size_t) tile * BUCKET_STRIDE;
    constexpr int ZPER = (KH * KW + TILE_THREADS - 1) / TILE_THREADS;
    constexpr int PER = (REC_CAP + TILE_THREADS - 1) / TILE_THREADS;
    float4 rr[PER], cc[PER];
    // the first REC_CAP records are loaded WITHOUT waiting for the count (the bucket is at least that
    // large, so the addresses are valid; slots past the count hold stale records and are masked below)
    static_assert(BUCKET_CAP >= ((REC_CAP + TILE_THREADS - 1) / TILE_THREADS) * TILE_THREADS, "speculative bucket loads stay in bounds");
#pragma unroll
    for (int u = 0; u < PER; u++) rr[u] = B[tid + u * TILE_THREADS];
    uint32_t zk[ZPER];
    bool zin[ZPER];
#pragma unroll
    for (int u = 0; u < ZPER; u++) {
        const int i = tid + u * TILE_THREADS;
        const int py = i / KW, pxl = i - py * KW;
        const int xr = x0 - 1 + pxl, yr = y0 - 1 + py;
        zin[u] = inside(xr, yr, W, H);
        const int x = min(max(xr, 0), W - 1), y = min(max(yr, 0), H - 1);       // clamped: always a valid address
        // W * H < 2^31 / 4: a 32-bit byte offset on the uniform base
        // (24-bit multiply: full rate, the 32-bit one is quarter rate; y, W < 2^24)
        zk[u] = *(const uint32_t*) ((const char*) a.zkeys + ((__umul24((uint32_t) y, (uint32_t) W) + (uint32_t) x) << 2));
    }
    for (int i = tid; i < BH * BW; i += TILE_THREADS) L.head[i] = REC_NULL;
    if (tid == 0) {
        L.nrec = 0;
        lds_dummy_record(L);
    }
    // colours of the records (slots past the count: point 0, discarded later; the host never passes a NULL cloud)
    const int n0 = bucketed ? min(REC_CAP, count) : 0;
#pragma unroll
    for (int u = 0; u < PER; u++) {
        const int i = tid + u * TILE_THREADS;
        cc[u] = fetch_rgbd(a, i < n0 ? __float_as_int(rr[u].w) : 0);
    }
    bool band = true;
#pragma unroll
    for (int u = 0; u < ZPER; u++) {
        const int i = tid + u * TILE_THREADS;
        if (i < KH * KW) {
            const float z = zkey_decode(zin[u] ? zk[u] : KBE_ZKEY_EMPTY);       // common.py:430 outside
            L.zpre[i] = z;
            band = band && degrid_fast_ok(z);
        }
    }
    {
        const unsigned long long odd = __ballot(!band);
        if (lane == 0) L.odd_z[tid >> 6] = odd != 0ull;
    }
    __syncthreads();
    // one decision per tile: every z of tile + halo in [2^19, 1e6] (any scene whose points are farther than
    // F*B/475712 from the camera) -> fp32-only, branch-free degrid and z test
    bool fast = true;
#pragma unroll
    for (int w = 0; w < TILE_THREADS / 64; w++) fast = fast && L.odd_z[w] == 0;
    fast = (bool) __builtin_amdgcn_readfirstlane((int) fast);
    tile_degrid(a, L, tid, x0, y0, fast);

    PixAcc acc[PIX_PER_THREAD];
#pragma unroll
    for (int m = 0; m < PIX_PER_THREAD; m++) { acc[m].rg = (f2) (0.0f); acc[m].bd = (f2) (0.0f); acc[m].w = 0.0f; }

    if (bucketed) {
        // the normal path: the tile's records, REC_CAP at a time (one round unless points pile up)
        for (int r0 = 0; r0 == 0 || r0 < count; r0 += REC_CAP) {
            const int n = min(REC_CAP, count - r0);
            if (r0 > 0) {
                __syncthreads();                                // the previous round's gather is done with the lists
                for (int i = tid; i < BH * BW; i += TILE_THREADS) L.head[i] = REC_NULL;
#pragma unroll
                for (int u = 0; u < PER; u++) {
                    const int i = tid + u * TILE_THREADS;
                    rr[u] = i < n ? B[r0 + i] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                }
#pragma unroll
                for (int u = 0; u < PER; u++) {
                    const int i = tid + u * TILE_THREADS;
                    cc[u] = i < n ? fetch_rgbd(a, __float_as_int(rr[u].w)) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                }
                __syncthreads();
            }
            {
                // all of this thread's list exchanges first, then the records with the links they returned (one after
                // the other each exchange was an LDS round trip in front of the next)
                int nxt[PER];
#pragma unroll
                for (int u = 0; u < PER; u++) {
                    const int i = tid + u * TILE_THREADS;
                    nxt[u] = REC_NULL;
                    if (i < n) {
                        const int bx = (int) floorf(rr[u].x) - (x0 - 1), by = (int) floorf(rr[u].y) - (y0 - 1);
                        L.rgbd[i] = cc[u];
                        nxt[u] = atomicExch(&L.head[__mul24(by, BW) + bx], i << 4);
                    }
                }
#pragma unroll
                for (int u = 0; u < PER; u++) {
                    const int i = tid + u * TILE_THREADS;
                    if (i < n) L.rec[i] = make_float4(rr[u].x, rr[u].y, rr[u].z, __int_as_float(nxt[u]));
                }
            }
            __syncthreads();
            if (fast) gather<true>(a, L, tid, x0, y0, acc);
            else gather<false>(a, L, tid, x0, y0, acc);
        }
        // no barrier here: what follows stages its bytes and per-wave partial results in the z-buffer area, dead since
        // the barrier in front of the gather, so a wave that is done resolves its pixels while others still walk
    } else {
        // the bucket overflowed (an extreme pile-up of points on this tile): re-derive the tile's
        // records from the whole cloud, REC_CAP at a time.  Slow, but any cloud renders correctly.
        __syncthreads();
        const int n_round = (a.N + TILE_THREADS - 1) / TILE_THREADS * TILE_THREADS;
        for (int i0 = 0; i0 < n_round; i0 += TILE_THREADS) {
            const int i = i0 + tid;
            bool ok = i < a.N;
            float ox = 0.0f, oy = 0.0f, z = 0.0f;
            if (ok) {
                float x = a.points[i], y = a.points[(size_t) a.N + i];
                z = a.points[2 * (size_t) a.N + i];
                apply_shift(a.cam, x, y, z);
                ok = project_xy(a.cam, x, y, z, ox, oy);
            }
            if (ok) {
                const int bx = (int) floorf(ox) - (x0 - 1), by = (int) floorf(oy) - (y0 - 1);
                ok = (bx >= 0) & (bx < BW) & (by >= 0) & (by < BH);
            }
            const unsigned long long m = __ballot(ok);
            if (m) {
                int base = 0;
                const int leader = __ffsll((long long) m) - 1;
                if (lane == leader) base = atomicAdd(&L.nrec, __popcll(m));
                base = __shfl(base, leader);
                if (ok) lds_insert(L, base + __popcll(m & ((1ull << lane) - 1ull)), ox, oy, project_err(a.cam, z), fetch_rgbd(a, i), x0, y0);
            }
            __syncthreads();
            if (L.nrec + TILE_THREADS > REC_CAP || i0 + TILE_THREADS >= n_round) {      // uniform
                gather<false>(a, L, tid, x0, y0, acc);
                __syncthreads();
                for (int j = tid; j < BH * BW; j += TILE_THREADS) L.head[j] = REC_NULL;
                if (tid == 0) L.nrec = 0;
                __syncthreads();
            }
        }
    }

    tile_epilogue(a, L, acc, tile, x0, y0);
    // Consecutive frames of a video alternate between two z-buffers: this launch leaves the OTHER one empty for the next
    // frame's projection (a tile's pixels of the buffer in use are still being read by its neighbours' halos, so a launch
    // cannot clear its own), and its own bucket counter (nobody else reads it).  That takes the z-buffer / bucket reset
    // -- a launch of its own riding in k_fill_holes for a frame rendered alone -- out of the scatter.
    if (a.zkeys_clear) {
        if (tid == 0) a.tile_count_clear[tile * CNT_STRIDE] = 0;
#pragma unroll
        for (int m = 0; m < PIX_PER_THREAD; m++) {
            const int q = tid + m * TILE_THREADS;
            const int ly = q / TW, lx = q - ly * TW;
            if (x0 + lx < W && y0 + ly < H) a.zkeys_clear[__umul24((uint32_t) (y0 + ly), (uint32_t) W) + (uint32_t) (x0 + lx)] = KBE_ZKEY_EMPTY;
        }
    }
}

// ---------------------------------------------------------------------------------------
// THE FUSED SCATTER: render_pointcloud (common.py:428-686) of one frame in ONE launch, from the packed cloud
// (kbe_cloud.h).  No global z-buffer, no bucket records, no global atomic: a tile PULLS its points.
//   cull     the tile walks the node hierarchy of the cloud (a node = a conservative box of where its points can land
//            in this view) down to its candidate blocks of 64 points: ~20 of 18 k at 1024^2, 2-3 node tests per thread;
//   splat    every candidate point is shifted (common.py:104-109) and projected (:447-468); a point whose north-west
//            corner lies in the tile or within two pixels of it min-splats the key of its dblError into the tile's
//            z-buffer IN LDS (tile + 1-pixel halo: one ds_min_u32 on the winner corner, :486-506), and a point whose
//            corner can colour a tile pixel becomes a record {ox, oy, dblError, index} in LDS, threaded into the
//            per-pixel lists at once;
//   then     exactly k_tiles: degrid (:525-568) in LDS, colours by point index, z-tested gather in registers
//            (:586-669), normalise (:686), hole mask (:253), uint8 (:255), coalesced stores.
// A halo pixel's z is the minimum over the points whose WINNER corner it is; those have their north-west corner
// within one more pixel, hence the two-pixel reach of the splat.  Neighbouring tiles project the blocks they share
// again (~2.3 tiles per block of an 8 x 8 patch): arithmetic that replaces 16-byte records written to and read back
// from HBM, the 4-byte z-buffer's atomics, its reset, and a kernel boundary.
// More than REC_CAP records on a tile (piled-up points, a cloud denser than the raster): the z-buffer is finished
// first, then the candidates are taken again in runs that fit (their record counts were noted on the first pass).
// More candidate blocks than the LDS list holds (MAXC: > 32 k points on one tile): the tile scans block ranges
// instead of a list, testing each block's node inline.  Slow paths, but any cloud renders correctly.
// ---------------------------------------------------------------------------------------
constexpr int MAXC = 256;                   // candidate blocks a tile lists in LDS at once
constexpr int RING = 128;                   // a wave's ring of waiting points: at most 63 left over + 64 new
static_assert(MAXC == TILE_THREADS, "one candidate per thread in the prefix scan of the slow path");

struct FrameArgs {
    PackedCloud pc;
    Camera cam;
    int tiles_x, tiles_y;
    uint8_t* frame;         // [H,W,3]
    float* depth;           // [H*W]
    uint32_t* mask;         // [H][ceil(W/32)]
    int* holes;
    int* hole_count;
    int4* bbox;
    uint32_t* coarse;
    float* render;          // optional [4,H,W] (unfilled; the fill kernel patches the holes)
    float* existing;        // optional [H*W]
    float* zee;             // optional [H*W] degridded z-buffer
    float* zee_pre;         // optional [H*W] pre-degrid z-buffer
    float4* spill;          // [n_tiles][BUCKET_STRIDE]: where a tile's records beyond REC_CAP wait for their round
};

struct FrameLds {
    TileLds T;
    int list[2][MAXC];      // node ids of the level being expanded / the candidate blocks
    int cnt[MAXC];          // records each candidate contributes (slow path: prefix sums)
    int n_at[kCloudMaxLevels];      // survivors per level
    int overflow;           // some level had more than MAXC survivors
    int n_ovf;              // records that did not fit the first round and went to the tile's spill area
    int ring[TILE_THREADS / 64][RING];      // per wave: indices of the points waiting for the exact work
    int wave_sum[TILE_THREADS / 64];
    int run_end;
};

struct CullView {           // the view, as the node tests need it
    float g, Sx, Sy;        // F' / Fd, shift_x * Fd, shift_y * Fd
    float focal, sx, sy, sz;
    float rx0, rx1, ry0, ry1;       // the tile's reach in (image position - principal point): [x0 - 2, x0 + TW + 1) etc.
};

// can a point of this node have its north-west corner within the tile's reach?  Conservative: the projection is
// monotone in each box coordinate (kbe_cloud.h), so the box corners bound it; a pixel of slack covers the rounding
// of these few operations and of the exact projection.
__device__ __forceinline__ bool node_hits(const CloudNode& n, const CullView& q)
{
    bool hit = false;
    if (n.flags & 1u) {
        const float d0 = n.z0 + q.sz, d1 = n.z1 + q.sz;
        if (d1 >= 0.001f) {                                     // else: all behind the near plane (common.py:453)
            if (d0 < 0.001f) {
                hit = true;                                     // straddles it: no bound
            } else {
                const float t0 = q.g * __builtin_amdgcn_rcpf(d0), t1 = q.g * __builtin_amdgcn_rcpf(d1);
                const float xa = __builtin_fmaf(n.px0, n.z0, q.Sx) * t0, xb = __builtin_fmaf(n.px1, n.z0, q.Sx) * t0;
                const float xc = __builtin_fmaf(n.px0, n.z1, q.Sx) * t1, xd = __builtin_fmaf(n.px1, n.z1, q.Sx) * t1;
                const float ya = __builtin_fmaf(n.py0, n.z0, q.Sy) * t0, yb = __builtin_fmaf(n.py1, n.z0, q.Sy) * t0;
                const float yc = __builtin_fmaf(n.py0, n.z1, q.Sy) * t1, yd = __builtin_fmaf(n.py1, n.z1, q.Sy) * t1;
                const float xlo = fminf(fminf(xa, xb), fminf(xc, xd)), xhi = fmaxf(fmaxf(xa, xb), fmaxf(xc, xd));
                const float ylo = fminf(fminf(ya, yb), fminf(yc, yd)), yhi = fmaxf(fmaxf(ya, yb), fmaxf(yc, yd));
                const float mx = 1.0f + 1.0e-4f * fmaxf(fabsf(xlo), fabsf(xhi)), my = 1.0f + 1.0e-4f * fmaxf(fabsf(ylo), fabsf(yhi));
                hit = (xhi + mx >= q.rx0) & (xlo - mx < q.rx1) & (yhi + my >= q.ry0) & (ylo - my < q.ry1);
            }
        }
    }
    if (!hit && (n.flags & 2u)) {
        const float d0 = n.Z0 + q.sz, d1 = n.Z1 + q.sz;
        if (d1 >= 0.001f) {
            if (d0 < 0.001f) {
                hit = true;
            } else {
                const float t0 = q.focal * __builtin_amdgcn_rcpf(d0), t1 = q.focal * __builtin_amdgcn_rcpf(d1);
                const float x0 = n.X0 + q.sx, x1 = n.X1 + q.sx, y0 = n.Y0 + q.sy, y1 = n.Y1 + q.sy;
                const float xlo = fminf(fminf(x0 * t0, x0 * t1), fminf(x1 * t0, x1 * t1)), xhi = fmaxf(fmaxf(x0 * t0, x0 * t1), fmaxf(x1 * t0, x1 * t1));
                const float ylo = fminf(fminf(y0 * t0, y0 * t1), fminf(y1 * t0, y1 * t1)), yhi = fmaxf(fmaxf(y0 * t0, y0 * t1), fmaxf(y1 * t0, y1 * t1));
                const float mx = 1.0f + 1.0e-4f * fmaxf(fabsf(xlo), fabsf(xhi)), my = 1.0f + 1.0e-4f * fmaxf(fabsf(ylo), fabsf(yhi));
                hit = (xhi + mx >= q.rx0) & (xlo - mx < q.rx1) & (yhi + my >= q.ry0) & (ylo - my < q.ry1);
            }
        }
    }
    return hit;
}

__device__ __forceinline__ float4 fetch_rgbd(const FrameArgs& a, int id)
{
#if defined(KBE_FRAME_STOP) && defined(KBE_FRAME_NO_RGBD)       // (dev) what do the colour loads cost?
    return make_float4(0.5f, 0.25f, 0.125f, 700.0f + (float) (id & 1));
#endif
    const uint32_t off = (uint32_t) id << 2;
    const char* r = (const char*) a.pc.rgb;
    const char* g = (const char*) (a.pc.rgb + (size_t) a.pc.Np);
    const char* b = (const char*) (a.pc.rgb + 2 * (size_t) a.pc.Np);
    const char* d = (const char*) a.pc.depth;
    return make_float4(*(const float*) (r + off), *(const float*) (g + off), *(const float*) (b + off), *(const float*) (d + off));
}

// what a pass over candidate blocks does with each point
enum : int { PASS_Z = 1, PASS_COUNT = 2, PASS_INSERT = 4, PASS_SPILL = 8 };

#if defined(KBE_FRAME_STATS)     // dev build only (tools/frame_stats.py): what the tiles of k_frame did, summed over launches
__device__ unsigned long long g_frame_stats[8];     // tiles, top-level survivors, candidate blocks, points in z reach, records, slow tiles, ranged tiles
#endif
#if defined(KBE_FRAME_STOP)      // dev build only (tools/gpu_variant_pmc.sh): the kernel ends after stage KBE_FRAME_STOP, to cost the stages
#define KBE_STOP_AFTER(n) do { if (KBE_FRAME_STOP == (n)) return; } while (0)
#else
#define KBE_STOP_AFTER(n) do { } while (0)
#endif

__global__ void __launch_bounds__(TILE_THREADS) __attribute__((KBE_TILE_ATTR)) k_frame(FrameArgs a)
{
    __shared__ FrameLds F;
    TileLds& L = F.T;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tile = xcd_tile(blockIdx.x, gridDim.x);
    const int ty = tile / a.tiles_x, tx = tile - ty * a.tiles_x;
    const int x0 = tx * TW, y0 = ty * TH;
    const int W = a.cam.W, H = a.cam.H;
    const Camera& cam = a.cam;
    const PackedCloud& pc = a.pc;
    uint32_t* const zk = (uint32_t*) L.zpre;            // the tile's z-buffer as keys until the splat is complete

    CullView q;
    q.g = cam.focal_f / pc.fd;
    q.sx = cam.has_shift ? cam.sx : 0.0f; q.sy = cam.has_shift ? cam.sy : 0.0f; q.sz = cam.has_shift ? cam.sz : 0.0f;
    q.Sx = q.sx * pc.fd; q.Sy = q.sy * pc.fd;
    q.focal = cam.focal_f;
    q.rx0 = (float) (x0 - 2) - cam.cx_f; q.rx1 = (float) (x0 + TW + 1) - cam.cx_f;
    q.ry0 = (float) (y0 - 2) - cam.cy_f; q.ry1 = (float) (y0 + TH + 1) - cam.cy_f;

    for (int i = tid; i < BH * BW; i += TILE_THREADS) L.head[i] = REC_NULL;
    for (int i = tid; i < KH * KW; i += TILE_THREADS) zk[i] = KBE_ZKEY_EMPTY;             // common.py:430
    if (tid < kCloudMaxLevels) F.n_at[tid] = 0;
    if (tid == 0) {
        L.nrec = 0;
        F.overflow = 0;
        F.n_ovf = 0;
        lds_dummy_record(L);
    }
    __syncthreads();

    // ---- cull: top level, then level by level down to the blocks
    auto append = [&](int* list, int* counter, bool hit, int id) {
        const unsigned long long m = __ballot(hit);
        if (m) {                                                // wave-uniform
            int base = 0;
            const int leader = __ffsll((long long) m) - 1;
            if (lane == leader) base = atomicAdd(counter, __popcll(m));
            base = __builtin_amdgcn_readlane(base, leader);
            const int pos = base + __popcll(m & ((1ull << lane) - 1ull));
            if (hit) {
                if (pos < MAXC) list[pos] = id;
                else F.overflow = 1;
            }
        }
    };
    const int top = pc.n_levels - 1;
    int cur = 0;
    for (int n0 = 0; n0 < pc.count[top]; n0 += TILE_THREADS) {
        const int n = n0 + tid;
        const bool hit = n < pc.count[top] && node_hits(pc.level[top][n], q);
        append(F.list[0], &F.n_at[top], hit, n);
    }
    __syncthreads();
    for (int lvl = top - 1; lvl >= 0 && !F.overflow; lvl--) {
        const int items = min(F.n_at[lvl + 1], MAXC) * kCloudFan;
        for (int it0 = 0; it0 < items; it0 += TILE_THREADS) {
            const int it = it0 + tid;
            int child = 0;
            bool hit = false;
            if (it < items) {
                child = F.list[cur][it / kCloudFan] * kCloudFan + (it % kCloudFan);
                hit = child < pc.count[lvl] && node_hits(pc.level[lvl][child], q);
            }
            append(F.list[cur ^ 1], &F.n_at[lvl], hit, child);
        }
        cur ^= 1;
        __syncthreads();
    }
    KBE_STOP_AFTER(1);                                          // (dev) the cull
    const bool ranged = F.overflow != 0;                        // uniform: scan block ranges instead of a list
    const int n_blocks = pc.count[0];
    const int* const cand = F.list[cur];

    // ---- the exact work on one point per lane: shift (common.py:104-109), projection (:447-468), then by `flags`
    // PASS_Z the min-splat of its dblError on the winner corner (:470-506), PASS_COUNT how many of the wave's points
    // become records (noted for candidate `c`), PASS_INSERT its record threaded into the per-pixel lists while there
    // is room (slots >= REC_CAP are dropped: the caller then knows from the total that the tile needs the slow path).
    auto exact_point = [&](int flags, float x, float y, float z, bool valid, int idx, int c, float4* spill) {
        float ox = 0.0f, oy = 0.0f;
        apply_shift(cam, x, y, z);
        const bool ok = project_xy(cam, x, y, z, ox, oy) && valid;
        Proj p;
        p.nwx = (int) floorf(ox); p.nwy = (int) floorf(oy);
        const int rx = p.nwx - (x0 - 2), ry = p.nwy - (y0 - 2);
        // north-west corner within [x0 - 2, x0 + TW] x [y0 - 2, y0 + TH]: its winner corner can be a pixel of tile + halo
        const bool in_z = ok && ((unsigned) rx <= (unsigned) (TW + 2)) & ((unsigned) ry <= (unsigned) (TH + 2));
        // ... within [x0 - 1, x0 + TW - 1] x [y0 - 1, y0 + TH - 1] and touching the image: it can colour a tile pixel
        const bool in_r = in_z && ((unsigned) (rx - 1) <= (unsigned) TW) & ((unsigned) (ry - 1) <= (unsigned) TH) &&
                          ((unsigned) (p.nwx + 1) <= (unsigned) W) & ((unsigned) (p.nwy + 1) <= (unsigned) H);
        float err = 0.0f;
#if defined(KBE_FRAME_STATS)
        { const unsigned long long mz = __ballot(in_z); if (lane == 0 && (flags & PASS_Z)) atomicAdd(&g_frame_stats[3], (unsigned long long) __popcll(mz)); }
#endif
        if (in_z) {
            err = project_err_fast(cam, z);
            if (flags & PASS_Z) {
                project_weights(ox, oy, p);
                const int k = winner_corner(p);                                     // common.py:486-506
                if (k >= 0) {
                    const int cx = p.nwx + (k & 1), cy = p.nwy + (k >> 1);
                    const int lx = cx - (x0 - 1), ly = cy - (y0 - 1);
                    if (inside(cx, cy, W, H) && ((unsigned) lx < (unsigned) KW) & ((unsigned) ly < (unsigned) KH))
                        atomicMin(&zk[__mul24(ly, KW) + lx], zkey_encode(err));
                }
            }
        }
        if (flags & (PASS_COUNT | PASS_INSERT)) {
            const unsigned long long m = __ballot(in_r);
            const int n_r = __popcll(m);
            if ((flags & PASS_COUNT) && lane == 0) F.cnt[c] = n_r;
            if ((flags & PASS_INSERT) && m) {                   // wave-uniform
                int base = 0;
                if (lane == 0) base = atomicAdd(&L.nrec, n_r);
                base = __builtin_amdgcn_readfirstlane(base);
                const int slot = base + __popcll(m & ((1ull << lane) - 1ull));
                if (in_r && slot < REC_CAP) {
                    const int next = atomicExch(&L.head[__mul24(ry - 1, BW) + (rx - 1)], slot << 4);
                    L.rec[slot] = make_float4(ox, oy, err, __int_as_float(next));
                    L.rgbd[slot].x = __int_as_float(idx);                           // the point, until its colours arrive
                }
                if ((flags & PASS_SPILL) && base + n_r > REC_CAP) {                 // wave-uniform; a few tiles in a hundred
                    const bool sp = in_r && slot >= REC_CAP;
                    const unsigned long long ms = __ballot(sp);
                    int sbase = 0;
                    if (lane == 0) sbase = atomicAdd(&F.n_ovf, __popcll(ms));
                    sbase = __builtin_amdgcn_readfirstlane(sbase);
                    const int o = sbase + __popcll(ms & ((1ull << lane) - 1ull));
                    if (sp && o < BUCKET_CAP) spill[o] = make_float4(ox, oy, err, __int_as_float(idx));
                }
            }
        }
    };

    constexpr int WAVES = TILE_THREADS / 64;
    auto load_block = [&](int b, float& x, float& y, float& z) {
        const uint32_t off = (uint32_t) ((b < 0 ? 0 : b) * kCloudBlock + lane) << 2;               // Np <= 2^30: 32-bit byte offsets
        x = *(const float*) ((const char*) pc.xyz + off);
        y = *(const float*) ((const char*) (pc.xyz + (size_t) pc.Np) + off);
        z = *(const float*) ((const char*) (pc.xyz + 2 * (size_t) pc.Np) + off);
    };

    // ---- slow path only: one exact pass over candidates [c0, c1) of the window starting at block `wbase` (list mode:
    // wbase unused).  A wave takes every fourth candidate; the coordinates of its next block are loaded before it
    // works on the current one.
    auto pass = [&](int flags, int c0, int c1, int wbase) {
        auto block_of = [&](int c) -> int {                     // wave-uniform
            if (c >= c1) return -1;
            if (!ranged) return cand[c];
            const int b = wbase + c;
            return (b < n_blocks && node_hits(pc.level[0][b], q)) ? b : -1;
        };
        int c = c0 + wave;
        int b_next = block_of(c);
        float xn, yn, zn;
        load_block(b_next, xn, yn, zn);
        for (; c < c1; c += WAVES) {                            // wave-uniform
            const int b = b_next;
            const float x = xn, y = yn, z = zn;
            b_next = block_of(c + WAVES);
            load_block(b_next, xn, yn, zn);
            if (b < 0) { if ((flags & PASS_COUNT) && lane == 0) F.cnt[c] = 0; continue; }
            exact_point(flags, x, y, z, true, b * kCloudBlock + lane, c, nullptr);
        }
    };

    constexpr int ZPER = (KH * KW + TILE_THREADS - 1) / TILE_THREADS;
    constexpr int PER = (REC_CAP + TILE_THREADS - 1) / TILE_THREADS;
    // keys -> floats in place (a pixel outside the image was never splatted: it reads 1e6 like common.py:430), and the
    // one decision per tile whether the fp32-only degrid and z test apply
    auto decode_z = [&]() {
        bool band = true;
#pragma unroll
        for (int u = 0; u < ZPER; u++) {
            const int i = tid + u * TILE_THREADS;
            if (i < KH * KW) {
                const float z = zkey_decode(zk[i]);
                L.zpre[i] = z;
                band = band && degrid_fast_ok(z);
            }
        }
        const unsigned long long odd = __ballot(!band);
        if (lane == 0) L.odd_z[tid >> 6] = odd != 0ull;
    };
    auto tile_is_fast = [&]() {
        bool fast = true;
#pragma unroll
        for (int w = 0; w < TILE_THREADS / 64; w++) fast = fast && L.odd_z[w] == 0;
        return (bool) __builtin_amdgcn_readfirstlane((int) fast);
    };

    PixAcc acc[PIX_PER_THREAD];
#pragma unroll
    for (int m = 0; m < PIX_PER_THREAD; m++) { acc[m].rg = (f2) (0.0f); acc[m].bd = (f2) (0.0f); acc[m].w = 0.0f; }

    // ---- the normal path, a stream per wave with no workgroup barrier inside.  Every candidate point gets an
    // APPROXIMATE position (one reciprocal, good to a thousandth of a pixel); two thirds of the candidates are near
    // misses that belong to neighbouring tiles and end here, after ~20 instructions instead of ~130.  The points
    // within a pixel of the tile's reach -- and every point nearer than z = 2, where process_shift's z / (z + 1e-7) is
    // not exactly 1 and the approximation does not hold -- are pushed onto the wave's ring (their indices); whenever 64
    // are waiting, the wave takes them off, reads their coordinates again (it has just read them: cache hits) and does
    // the EXACT work with every lane busy: z-splat into the LDS z-tile, record into the per-pixel lists.  Records
    // beyond REC_CAP (a few tiles in a hundred: two surfaces over one another at a depth edge) spill into the tile's
    // own area of the scratch in HBM and are gathered in further rounds.
    const int n_cand = ranged ? 0 : min(F.n_at[0], MAXC);
    float4* const spill = a.spill + (size_t) tile * BUCKET_STRIDE;
    if (!ranged) {
        int* const ring = F.ring[wave];
        int head = 0, tail = 0;                                 // wave-uniform
        int c = wave;
        int b_next = c < n_cand ? cand[c] : -1;
        float xn, yn, zn;
        load_block(b_next, xn, yn, zn);
        const float wx = (float) (TW + 3) + 1.0f, wy = (float) (TH + 3) + 1.0f;
        while (c < n_cand || tail > head) {                     // wave-uniform
            if (c < n_cand) {
                const int b = b_next;
                const float x = xn, y = yn, z = zn;
                c += WAVES;
                b_next = c < n_cand ? cand[c] : -1;
                load_block(b_next, xn, yn, zn);
                const float zs = z + q.sz;
                const float t = q.focal * __builtin_amdgcn_rcpf(zs);
                const float ax = (x + q.sx) * t - q.rx0, ay = (y + q.sy) * t - q.ry0;      // position relative to the start of the reach
                const bool take = (zs >= 0.0009f) && (!(z >= 2.0f) || ((ax >= -1.0f) & (ax < wx) & (ay >= -1.0f) & (ay < wy)));
                const unsigned long long m = __ballot(take);
                if (take) ring[(tail + __popcll(m & ((1ull << lane) - 1ull))) & (RING - 1)] = b * kCloudBlock + lane;
                tail += __popcll(m);
            }
            if (tail - head >= 64 || (c >= n_cand && tail > head)) {
                const int n = min(64, tail - head);
                const bool valid = lane < n;
                const int idx = valid ? ring[(head + lane) & (RING - 1)] : 0;
                head += n;
                const uint32_t off = (uint32_t) idx << 2;
                const float x = *(const float*) ((const char*) pc.xyz + off);
                const float y = *(const float*) ((const char*) (pc.xyz + (size_t) pc.Np) + off);
                const float z = *(const float*) ((const char*) (pc.xyz + 2 * (size_t) pc.Np) + off);
                exact_point(PASS_Z | PASS_INSERT | PASS_SPILL, x, y, valid ? z : 4.0f, valid, idx, 0, spill);
            }
        }
    }
    __syncthreads();
    KBE_STOP_AFTER(3);                                          // (dev) + the stream
    const int total = L.nrec;
    bool fast;
    const int n_spill = F.n_ovf;
    if (!ranged && n_spill <= BUCKET_CAP) {
        // ---- the first REC_CAP records are in LDS.  Colours by point index now (in flight during the degrid)
        const int n_first = min(total, REC_CAP);
        float4 cc[PER];
#pragma unroll
        for (int u = 0; u < PER; u++) {
            const int i = tid + u * TILE_THREADS;
            cc[u] = fetch_rgbd(a, i < n_first ? __float_as_int(L.rgbd[i].x) : 0);
        }
        decode_z();
        __syncthreads();
        fast = tile_is_fast();
        tile_degrid(a, L, tid, x0, y0, fast);
#pragma unroll
        for (int u = 0; u < PER; u++) {
            const int i = tid + u * TILE_THREADS;
            if (i < n_first) L.rgbd[i] = cc[u];
        }
        __syncthreads();
        KBE_STOP_AFTER(4);                                      // (dev) + colours, degrid
        if (fast) gather<true>(a, L, tid, x0, y0, acc);
        else gather<false>(a, L, tid, x0, y0, acc);
        KBE_STOP_AFTER(5);                                      // (dev) + gather
        // further rounds: the spilled records, REC_CAP at a time (already projected: only lists, colours and the walk)
        for (int r0 = 0; r0 < n_spill; r0 += REC_CAP) {         // uniform
            const int n = min(REC_CAP, n_spill - r0);
            __syncthreads();                                    // the previous gather is done with the lists
            for (int i = tid; i < BH * BW; i += TILE_THREADS) L.head[i] = REC_NULL;
            float4 rr[PER];
#pragma unroll
            for (int u = 0; u < PER; u++) {
                const int i = tid + u * TILE_THREADS;
                rr[u] = i < n ? spill[r0 + i] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            }
#pragma unroll
            for (int u = 0; u < PER; u++) {
                const int i = tid + u * TILE_THREADS;
                cc[u] = fetch_rgbd(a, i < n ? __float_as_int(rr[u].w) : 0);
            }
            __syncthreads();
#pragma unroll
            for (int u = 0; u < PER; u++) {
                const int i = tid + u * TILE_THREADS;
                if (i < n) lds_insert(L, i, rr[u].x, rr[u].y, rr[u].z, cc[u], x0, y0);
            }
            __syncthreads();
            if (fast) gather<true>(a, L, tid, x0, y0, acc);
            else gather<false>(a, L, tid, x0, y0, acc);
        }
    } else {
        // ---- the slow path, a small uniform state machine around ONE more copy of the pass: [ranged: the z-splat,
        // window by window;] degrid; then per window [ranged: count,] prefix sums of the counts and runs of at most
        // REC_CAP records: insert, colours, gather.
        enum { S_ZWIN, S_DEGRID, S_COUNT, S_SCAN, S_RUN, S_DONE };
        // (list mode gets here with a z-buffer that is complete only if phase B ran: it is simply done again, with the counts)
        int state = S_ZWIN, wb = 0, c0 = 0, done = 0;
#if defined(KBE_FRAME_STOP) && defined(KBE_FRAME_SKIP_SLOW)      // (dev) what would the launch cost without its slow tiles?
        state = S_DONE;
#endif
        fast = false;
        while (state != S_DONE) {                               // uniform
            const int n_win = ranged ? min(MAXC, n_blocks - wb) : n_cand;
            int flags = 0, p0 = 0, p1 = n_win;
            if (state == S_ZWIN) flags = ranged ? PASS_Z : (PASS_Z | PASS_COUNT);
            else if (state == S_COUNT) flags = PASS_COUNT;
            else if (state == S_RUN) {
                for (int i = tid; i < BH * BW; i += TILE_THREADS) L.head[i] = REC_NULL;
                if (tid == 0) { L.nrec = 0; F.run_end = n_win; }
                __syncthreads();
                // the run ends in front of the first candidate whose prefix sum exceeds done + REC_CAP
                for (int c = c0 + tid; c < n_win; c += TILE_THREADS)
                    if (F.cnt[c] - done > REC_CAP && (c == c0 || F.cnt[c - 1] - done <= REC_CAP)) F.run_end = c;
                __syncthreads();
                flags = PASS_INSERT; p0 = c0; p1 = F.run_end;
            }
            if (flags) pass(flags, p0, p1, wb);
            __syncthreads();
            if (state == S_ZWIN) {
                wb += MAXC;
                if (!ranged || wb >= n_blocks) state = S_DEGRID;
            } else if (state == S_DEGRID) {
                decode_z();
                __syncthreads();
                fast = tile_is_fast();
                tile_degrid(a, L, tid, x0, y0, fast);
                wb = 0;
                state = ranged ? S_COUNT : S_SCAN;
            } else if (state == S_COUNT) {
                state = S_SCAN;
            } else if (state == S_SCAN) {
                // inclusive prefix sums of the counts, one entry per thread
                const int e = tid < n_win ? F.cnt[tid] : 0;
                int v = e;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) { const int t = __shfl_up(v, off); if (lane >= off) v += t; }
                if (lane == 63) F.wave_sum[tid >> 6] = v;
                __syncthreads();
                for (int w = 0; w < (tid >> 6); w++) v += F.wave_sum[w];
                if (tid < n_win) F.cnt[tid] = v;
                c0 = 0; done = 0;
                state = n_win > 0 ? S_RUN : S_DONE;
                if (state == S_DONE && ranged && wb + MAXC < n_blocks) { wb += MAXC; state = S_COUNT; }
            } else if (state == S_RUN) {
                const int n = min(L.nrec, REC_CAP);
                for (int i = tid; i < n; i += TILE_THREADS) L.rgbd[i] = fetch_rgbd(a, __float_as_int(L.rgbd[i].x));
                __syncthreads();
                if (fast) gather<true>(a, L, tid, x0, y0, acc);
                else gather<false>(a, L, tid, x0, y0, acc);
                c0 = p1;
                done = c0 > 0 ? F.cnt[c0 - 1] : 0;
                if (c0 >= n_win) {
                    state = S_DONE;
                    if (ranged && wb + MAXC < n_blocks) { wb += MAXC; state = S_COUNT; }
                }
            }
            __syncthreads();
        }
    }
#if defined(KBE_FRAME_STATS)
    if (tid == 0) {
        atomicAdd(&g_frame_stats[0], 1ull);
        atomicAdd(&g_frame_stats[1], (unsigned long long) F.n_at[top]);
        atomicAdd(&g_frame_stats[2], (unsigned long long) n_cand);
        atomicAdd(&g_frame_stats[4], (unsigned long long) total);
        atomicAdd(&g_frame_stats[5], (unsigned long long) !(!ranged && n_spill <= BUCKET_CAP));
        atomicAdd(&g_frame_stats[6], (unsigned long long) (n_spill > 0));
        atomicAdd(&g_frame_stats[7], (unsigned long long) n_spill);
    }
#endif
    tile_epilogue(a, L, acc, tile, x0, y0);
}

// ---------------------------------------------------------------------------------------
// render_pointcloud for ANY channel count on the same machinery (the 68-channel forward warp of the inpaint
// set-up, pointcloud_inpainting.py:201: image, disparity and 64 context features): k_project fills the z-buffer
// and the buckets exactly as for a frame; this kernel degrids the tile once and then takes the data four
// channels at a time -- per chunk the records are threaded into the per-pixel lists again with their four values
// (the lists cost little next to the walk), every pixel walks its bins and the chunk leaves normalised
// (common.py:686).  No accumulator in HBM, no floating-point atomic; 20x faster than the global-atomic
// formulation at 68 channels (0.1 vs 2.2 ms at 1024^2).
// ---------------------------------------------------------------------------------------
struct TileNcArgs {
    const float* points;    // [3,N]  (only the brute-force path of an overflowing bucket reads it)
    const float* data;      // [C,N]
    int N, C;
    Camera cam;
    const uint32_t* zkeys;
    const int* tile_count;
    const float4* buckets;
    int tiles_x, tiles_y;
    float* render;          // [C,H,W] normalised
    float* existing;        // [H*W] weight sum
};

__device__ __forceinline__ float4 fetch_chunk(const TileNcArgs& a, int id, int c0)
{
    const size_t N = (size_t) a.N;
    const float* D = a.data + (size_t) c0 * N + id;
    float4 v;
    v.x = D[0];
    v.y = c0 + 1 < a.C ? D[N] : 0.0f;
    v.z = c0 + 2 < a.C ? D[2 * N] : 0.0f;
    v.w = c0 + 3 < a.C ? D[3 * N] : 0.0f;
    return v;
}

__global__ void __launch_bounds__(TILE_THREADS) __attribute__((amdgpu_waves_per_eu(4, 4))) k_tiles_nc(TileNcArgs a)
{
    __shared__ TileLds L;
    const int tid = threadIdx.x, lane = tid & 63;
    const int tile = xcd_tile(blockIdx.x, gridDim.x);
    const int ty = tile / a.tiles_x, tx = tile - ty * a.tiles_x;
    const int x0 = tx * TW, y0 = ty * TH;
    const int W = a.cam.W, H = a.cam.H;
    const size_t HW = (size_t) W * H;
    const int count = a.tile_count[tile * CNT_STRIDE];
    const bool bucketed = count <= BUCKET_CAP;
    const float4* B = a.buckets + (size_t) tile * BUCKET_STRIDE;
    constexpr int PER = (REC_CAP + TILE_THREADS - 1) / TILE_THREADS;

    // z tile + halo -> LDS, one decision per tile (see k_tiles), degrid
    bool band = true;
    for (int i = tid; i < KH * KW; i += TILE_THREADS) {
        const int py = i / KW, pxl = i - py * KW;
        const int x = x0 - 1 + pxl, y = y0 - 1 + py;
        const float z = zkey_decode(inside(x, y, W, H) ? a.zkeys[(size_t) y * W + x] : KBE_ZKEY_EMPTY);     // common.py:430 outside
        L.zpre[i] = z;
        band = band && degrid_fast_ok(z);
    }
    {
        const unsigned long long odd = __ballot(!band);
        if (lane == 0) L.odd_z[tid >> 6] = odd != 0ull;
    }
    if (tid == 0) lds_dummy_record(L);
    __syncthreads();
    bool fast = true;
#pragma unroll
    for (int w = 0; w < TILE_THREADS / 64; w++) fast = fast && L.odd_z[w] == 0;
    fast = (bool) __builtin_amdgcn_readfirstlane((int) fast);
    for (int i = tid; i < TH * TW; i += TILE_THREADS) {
        const int ly = i / TW, lx = i - ly * TW;
        const int x = x0 + lx, y = y0 + ly;
        if (x >= W || y >= H) continue;
        auto at = [&](int xx, int yy) { return L.zpre[(yy - y0 + 1) * KW + (xx - x0 + 1)]; };
        L.zee[i] = degrid_pixel(x, y, W, H, at);
    }

    for (int c0 = 0; c0 < a.C; c0 += 4) {
        PixAcc acc[PIX_PER_THREAD];
#pragma unroll
        for (int m = 0; m < PIX_PER_THREAD; m++) { acc[m].rg = (f2) (0.0f); acc[m].bd = (f2) (0.0f); acc[m].w = 0.0f; }
        if (bucketed) {
            for (int r0 = 0; r0 == 0 || r0 < count; r0 += REC_CAP) {
                const int n = min(REC_CAP, count - r0);
                float4 rr[PER], cc[PER];
#pragma unroll
                for (int u = 0; u < PER; u++) {
                    const int i = tid + u * TILE_THREADS;
                    rr[u] = B[min(r0 + i, BUCKET_CAP - 1)];                // unconditional (clamped) loads: see k_tiles
                }
#pragma unroll
                for (int u = 0; u < PER; u++) {
                    const int i = tid + u * TILE_THREADS;
                    cc[u] = fetch_chunk(a, i < n ? __float_as_int(rr[u].w) : 0, c0);
                }
                __syncthreads();                                        // zee written / the previous gather is done with the lists
                for (int i = tid; i < BH * BW; i += TILE_THREADS) L.head[i] = REC_NULL;
                __syncthreads();
#pragma unroll
                for (int u = 0; u < PER; u++) {
                    const int i = tid + u * TILE_THREADS;
                    if (i < n) lds_insert(L, i, rr[u].x, rr[u].y, rr[u].z, cc[u], x0, y0);
                }
                __syncthreads();
                if (fast) gather<true>(a, L, tid, x0, y0, acc);
                else gather<false>(a, L, tid, x0, y0, acc);
            }
        } else {
            // the bucket overflowed (an extreme pile-up of points on this tile): re-derive the tile's records from the
            // whole cloud, REC_CAP at a time.  Slow, but any cloud renders correctly.
            __syncthreads();
            for (int j = tid; j < BH * BW; j += TILE_THREADS) L.head[j] = REC_NULL;
            if (tid == 0) L.nrec = 0;
            __syncthreads();
            const int n_round = (a.N + TILE_THREADS - 1) / TILE_THREADS * TILE_THREADS;
            for (int i0 = 0; i0 < n_round; i0 += TILE_THREADS) {
                const int i = i0 + tid;
                bool ok = i < a.N;
                float ox = 0.0f, oy = 0.0f, z = 0.0f;
                if (ok) {
                    float x = a.points[i], y = a.points[(size_t) a.N + i];
                    z = a.points[2 * (size_t) a.N + i];
                    apply_shift(a.cam, x, y, z);
                    ok = project_xy(a.cam, x, y, z, ox, oy);
                }
                if (ok) {
                    const int bx = (int) floorf(ox) - (x0 - 1), by = (int) floorf(oy) - (y0 - 1);
                    ok = (bx >= 0) & (bx < BW) & (by >= 0) & (by < BH);
                }
                const unsigned long long m = __ballot(ok);
                if (m) {
                    int base = 0;
                    const int leader = __ffsll((long long) m) - 1;
                    if (lane == leader) base = atomicAdd(&L.nrec, __popcll(m));
                    base = __builtin_amdgcn_readlane(base, leader);
                    if (ok) lds_insert(L, base + __popcll(m & ((1ull << lane) - 1ull)), ox, oy, project_err(a.cam, z), fetch_chunk(a, i, c0), x0, y0);
                }
                __syncthreads();
                if (L.nrec + TILE_THREADS > REC_CAP || i0 + TILE_THREADS >= n_round) {      // uniform
                    gather<false>(a, L, tid, x0, y0, acc);
                    __syncthreads();
                    for (int j = tid; j < BH * BW; j += TILE_THREADS) L.head[j] = REC_NULL;
                    if (tid == 0) L.nrec = 0;
                    __syncthreads();
                }
            }
        }
        // the chunk leaves normalised (common.py:686); the weight sum with the first chunk
#pragma unroll
        for (int m = 0; m < PIX_PER_THREAD; m++) {
            const int q = tid + m * TILE_THREADS;
            const int ly = q / TW, lx = q - ly * TW;
            const int x = x0 + lx, y = y0 + ly;
            if (x >= W || y >= H) continue;
            const size_t o = (size_t) y * W + x;
            const float den = acc[m].w + 0.0000001f;
            a.render[(size_t) c0 * HW + o] = acc[m].rg.x / den;
            if (c0 + 1 < a.C) a.render[(size_t) (c0 + 1) * HW + o] = acc[m].rg.y / den;
            if (c0 + 2 < a.C) a.render[(size_t) (c0 + 2) * HW + o] = acc[m].bd.x / den;
            if (c0 + 3 < a.C) a.render[(size_t) (c0 + 3) * HW + o] = acc[m].bd.y / den;
            if (c0 == 0) a.existing[o] = acc[m].w;
        }
    }
}

// ---------------------------------------------------------------------------------------
// hole fill: one 32-lane group per hole; lane = direction * 2 + end (0: against, 1: along)
// (common.py:838-924; the direction loop and both ray walks run in parallel, then the
// "strictly shorter, first direction wins" reduction picks the same source pixel)
// ---------------------------------------------------------------------------------------
struct FillRect { int x0, y0, x1, y1; };    // inclusive; only holes inside are filled

#ifndef KBE_FILL_SERIAL_BATCH
#define KBE_FILL_SERIAL_BATCH 8
#endif
#ifndef KBE_FILL_BY_COUNT_MIN_LANES
#define KBE_FILL_BY_COUNT_MIN_LANES 2
#endif
constexpr int COARSE_WORDS = 2048;     // 8 x 8 blocks of images up to 2048 x 2048 (larger: the walks do not skip)
#ifndef KBE_FILL_SERIAL_MIN
#define KBE_FILL_SERIAL_MIN 49152       // holes per frame from which one lane per hole beats one half-wave per hole
#endif

// Frames with very many holes (dolly: no inpainting, common.py:217; hundreds of thousands of holes in wide
// disocclusion bands): ONE LANE PER HOLE, the 16 directions in the reference's order, both ends of a direction
// advancing together.  In the half-wave-per-hole scheme most lanes are pruned after the first batches and the wave
// then walks a few long rays at 3 % lane utilisation; here a lane always does useful work, and a later direction
// stops as soon as the distance between its two current positions reaches the best complete one (it would have to
// be STRICTLY shorter to win, common.py:900).  Same exact arithmetic (:876-898); neighbouring lanes hold
// neighbouring holes (the list is written tile by tile), so their walks have similar lengths.
__device__ __forceinline__ void fill_hole_serial(int px, const float* __restrict__ depth, const uint32_t* __restrict__ mask, int W, int H,
                                                 int wpr, const FillDirs& dirs, int bx0, int by0, int bx1, int by1,
                                                 uint8_t* __restrict__ frame, float* __restrict__ render,
                                                 const uint32_t* near, int c_wpr)
{
    constexpr int SB = KBE_FILL_SERIAL_BATCH;
    static_assert(SB <= 8, "a skipped batch must stay within one 8 x 8 block of where it ends");
    // true when no valid pixel lies within one block of the block of in-image position (px_, py_)
    auto far_from_valid = [&](int px_, int py_) { return !((near[(py_ >> 3) * c_wpr + (px_ >> 8)] >> ((px_ >> 3) & 31)) & 1u); };
    const int y = px / W, x = px - y * W;
    float best = 1000000.0f;                    // dblShortest (:854)
    float best_s = INFINITY;                    // ex^2 + ey^2 of the best direction (what `best` is the sqrtf of)
    int sax = -1, say = -1, sbx = -1, sby = -1;
    // A hole outside the box of valid pixels: of the two opposite ends of any direction at least one moves away from
    // the box or parallel to it and can never hit a valid pixel, so every direction is skipped (:884-885, :895-896)
    if (x < bx0 || x > bx1 || y < by0 || y > by1) return;
    for (int d = 0; d < 16; d++) {
        const float ddx = dirs.x[d], ddy = dirs.y[d];
        float fa_x = (float) x, fa_y = (float) y, fb_x = fa_x, fb_y = fa_y;
        int ax = x, ay = y, bx = x, by = y;
        bool hit_a = false, hit_b = false, dead = false;
        // phase 1: both ends together, until one of them has hit
        while (!dead && !hit_a && !hit_b) {
            if (near) {
                // both ends deep inside a hole?  then the next SB steps of both cannot hit anything: take them at once
                float ta_x = fa_x, ta_y = fa_y, tb_x = fb_x, tb_y = fb_y;
#pragma unroll
                for (int k = 0; k < SB; k++) { ta_x -= ddx; ta_y -= ddy; tb_x += ddx; tb_y += ddy; }       // the same fp32 sums
                const int eax = (int) roundf(ta_x), eay = (int) roundf(ta_y), ebx = (int) roundf(tb_x), eby = (int) roundf(tb_y);
                const bool in_both = ((unsigned) eax < (unsigned) W) & ((unsigned) eay < (unsigned) H) & ((unsigned) ebx < (unsigned) W) & ((unsigned) eby < (unsigned) H);
                if (in_both && far_from_valid(eax, eay) && far_from_valid(ebx, eby)) {
                    fa_x = ta_x; fa_y = ta_y; fb_x = tb_x; fb_y = tb_y;
                    ax = eax; ay = eay; bx = ebx; by = eby;
                    if ((ax < bx0 && ddx >= 0.0f) || (ax > bx1 && ddx <= 0.0f) || (ay < by0 && ddy >= 0.0f) || (ay > by1 && ddy <= 0.0f)) { dead = true; break; }
                    if ((bx < bx0 && ddx <= 0.0f) || (bx > bx1 && ddx >= 0.0f) || (by < by0 && ddy <= 0.0f) || (by > by1 && ddy >= 0.0f)) { dead = true; break; }
                    const float sx_ = (float) (bx - ax), sy_ = (float) (by - ay);
                    if (sx_ * sx_ + sy_ * sy_ >= best_s) { dead = true; break; }
                    continue;
                }
            }
            // a batch of SB steps per end: positions first (they do not depend on the data), loads together
            int pax[SB], pay[SB], pbx[SB], pby[SB];
            uint32_t wa[SB], wb[SB];
            bool ina[SB], inb[SB];
#pragma unroll
            for (int k = 0; k < SB; k++) {
                fa_x -= ddx; pax[k] = (int) roundf(fa_x);       // :876-877
                fa_y -= ddy; pay[k] = (int) roundf(fa_y);
                fb_x += ddx; pbx[k] = (int) roundf(fb_x);       // :887-888
                fb_y += ddy; pby[k] = (int) roundf(fb_y);
                ina[k] = ((unsigned) pax[k] < (unsigned) W) & ((unsigned) pay[k] < (unsigned) H);
                inb[k] = ((unsigned) pbx[k] < (unsigned) W) & ((unsigned) pby[k] < (unsigned) H);
                wa[k] = mask[(ina[k] && !hit_a) ? (unsigned) pay[k] * (unsigned) wpr + ((unsigned) pax[k] >> 5) : 0u];
                wb[k] = mask[(inb[k] && !hit_b) ? (unsigned) pby[k] * (unsigned) wpr + ((unsigned) pbx[k] >> 5) : 0u];
            }
#pragma unroll
            for (int k = 0; k < SB; k++) {
                if (!hit_a && !dead) {
                    ax = pax[k]; ay = pay[k];
                    if (!ina[k]) dead = true;
                    else if ((wa[k] >> (ax & 31)) & 1u) hit_a = true;
                }
                if (!hit_b && !dead) {
                    bx = pbx[k]; by = pby[k];
                    if (!inb[k]) dead = true;
                    else if ((wb[k] >> (bx & 31)) & 1u) hit_b = true;
                }
            }
            if (dead) break;
            // left the box of valid pixels for good?
            if (!hit_a && ((ax < bx0 && ddx >= 0.0f) || (ax > bx1 && ddx <= 0.0f) || (ay < by0 && ddy >= 0.0f) || (ay > by1 && ddy <= 0.0f))) { dead = true; break; }
            if (!hit_b && ((bx < bx0 && ddx <= 0.0f) || (bx > bx1 && ddx >= 0.0f) || (by < by0 && ddy <= 0.0f) || (by > by1 && ddy >= 0.0f))) { dead = true; break; }
            // bound: the ends only move apart
            const float ex = (float) (bx - ax), ey = (float) (by - ay);
            const float s_now = ex * ex + ey * ey;
            if (s_now >= best_s) { dead = true; break; }       // sqrtf is monotone: this direction cannot become STRICTLY shorter (:900)
        }
        // phase 2: the end that is still looking walks alone (half the arithmetic per step)
        if (!dead && hit_a != hit_b) {
            const bool is_a = !hit_a;
            const float sdx = is_a ? -ddx : ddx, sdy = is_a ? -ddy : ddy;
            float fx = is_a ? fa_x : fb_x, fy = is_a ? fa_y : fb_y;
            int cx = is_a ? ax : bx, cy = is_a ? ay : by;
            const int ox = is_a ? bx : ax, oy = is_a ? by : ay;            // the end that has hit stays put
            bool hit = false;
            while (!dead && !hit) {
                if (near) {
                    float tx = fx, ty = fy;
#pragma unroll
                    for (int k = 0; k < SB; k++) { tx += sdx; ty += sdy; }
                    const int ex_ = (int) roundf(tx), ey_ = (int) roundf(ty);
                    if (((unsigned) ex_ < (unsigned) W) & ((unsigned) ey_ < (unsigned) H) && far_from_valid(ex_, ey_)) {
                        fx = tx; fy = ty; cx = ex_; cy = ey_;
                        if ((cx < bx0 && sdx <= 0.0f) || (cx > bx1 && sdx >= 0.0f) || (cy < by0 && sdy <= 0.0f) || (cy > by1 && sdy >= 0.0f)) { dead = true; break; }
                        const float sx_ = (float) (cx - ox), sy_ = (float) (cy - oy);
                        if (sx_ * sx_ + sy_ * sy_ >= best_s) { dead = true; break; }
                        continue;
                    }
                }
                int px_[SB], py_[SB];
                uint32_t wv[SB];
                bool in_[SB];
#pragma unroll
                for (int k = 0; k < SB; k++) {
                    fx += sdx; px_[k] = (int) roundf(fx);
                    fy += sdy; py_[k] = (int) roundf(fy);
                    in_[k] = ((unsigned) px_[k] < (unsigned) W) & ((unsigned) py_[k] < (unsigned) H);
                    wv[k] = mask[in_[k] ? (unsigned) py_[k] * (unsigned) wpr + ((unsigned) px_[k] >> 5) : 0u];
                }
#pragma unroll
                for (int k = 0; k < SB; k++) {
                    if (!hit && !dead) {
                        cx = px_[k]; cy = py_[k];
                        if (!in_[k]) dead = true;
                        else if ((wv[k] >> (cx & 31)) & 1u) hit = true;
                    }
                }
                if (dead || hit) break;
                if ((cx < bx0 && sdx <= 0.0f) || (cx > bx1 && sdx >= 0.0f) || (cy < by0 && sdy <= 0.0f) || (cy > by1 && sdy >= 0.0f)) { dead = true; break; }
                const float ex = (float) (cx - ox), ey = (float) (cy - oy);
                if (ex * ex + ey * ey >= best_s) { dead = true; break; }
            }
            if (is_a) { ax = cx; ay = cy; } else { bx = cx; by = cy; }
        }
        if (dead) continue;
        const float ex = (float) (bx - ax), ey = (float) (by - ay);
        const float sq = ex * ex + ey * ey;
        const float dist = sqrtf(sq);                           // :898
        if (best > dist) { best = dist; best_s = sq; sax = ax; say = ay; sbx = bx; sby = by; }     // :900
    }
    if (sax < 0) return;                                        // unfillable: keeps the rendered value (:913-919)
    int sx = sax, sy = say;
    if (depth[(size_t) say * W + sax] < depth[(size_t) sby * W + sbx]) { sx = sbx; sy = sby; }     // :904 the farther (background) end
    const size_t s = (size_t) sy * W + sx, o = (size_t) px, HW = (size_t) W * H;
    frame[o * 3] = frame[s * 3]; frame[o * 3 + 1] = frame[s * 3 + 1]; frame[o * 3 + 2] = frame[s * 3 + 2];
    if (render) for (int c = 0; c < 4; c++) render[c * HW + o] = render[c * HW + s];
}

// ---------------------------------------------------------------------------------------
// Frames with very many holes (no inpainting: a dolly zoom, common.py:217, or a raw cloud): most of a ray's steps cross
// empty space, and with the block-level skips the cost was in the last 8-16 single steps of every ray in front of the
// rim (measured: 27 eight-step batches per hole).  k_hole_dist gives every pixel its Chebyshev distance D to the nearest
// valid pixel (0 = valid, capped); a ray at a hole with distance D can take D - 1 steps at once and look
// only at where it lands: a step moves at most 1 pixel per axis and rounding a position adds at most 1, so the first
// D - 2 positions are holes for sure.  Same fp32 sums (:876-889), same positions tested in the end, no mask look-ups.
// The table is D iterations of a 3 x 3 dilation of the validity bitmask in LDS (32 pixels per word), the distance =
// the number of iterations a pixel's bit stayed clear, counted in bit planes.
// ---------------------------------------------------------------------------------------
// The fill asks the pixel table only where the block table says "near" (the nearest block with a valid pixel is the pixel's
// own or a neighbour: the nearest valid pixel is then at most 15 away), so 15 dilations are all it needs; the block
// table carries the long jumps, and 15 blocks (jumps of up to 158 steps) serve as well as 31: the launch sits between the
// tile launch and the fill of every such frame, and its length is its number of dilations (dolly bench: 138.4 us per frame
// with 31 / 31, 137.5 with 15 / 31, 134.9 with 15 / 15).  (A capped entry is a lower bound of the distance: still safe.)
#ifndef KBE_DIST_CAP
#define KBE_DIST_CAP 15
#endif
#ifndef KBE_DIST_CAP_BLOCKS
#define KBE_DIST_CAP_BLOCKS 15
#endif
constexpr int DT_W = 64, DT_H = 32;                 // interior of one workgroup: 2 words x 32 rows
constexpr int DT_WORDS = 4;
static_assert(DT_W == 64, "the halo is one 32-pixel word on each side");

// Strip tables.  A ray of direction u through a hole p stays within 0.75 pixels of the line through p (positions are
// rounded per axis; the fp32 sums drift by < 0.03 over 1000 steps), so the only valid pixels it can ever meet lie in the
// strip of lines c in [b - 1, b + 2), b = floor(c(p)), c(q) = n . q the coordinate across the direction.  Per direction
// and b, (lo, hi) bound the coordinate t(q) = u . q along the direction over every valid pixel of that strip -- or rather
// over a superset of them: the x-extent of each tile row (the y-extent of each tile column for the flat directions), from
// the tiles' boxes.  The end walking towards -u meets nothing once lo > t + 1, the end towards +u once hi < t - 1: the
// direction is skipped (common.py:880-885, 891-896) without walking to the image border.  A zoomed-out frame is mostly
// border around a convex patch of valid pixels; outside a convex patch NO direction has valid pixels on both sides.
// Measured on the last frame of the dolly bench (266 k holes inside the box of valid pixels): 1.7 of a hole's 16 directions
// complete, 4.5 pass this test; pixel steps per hole 6811 -> 560 (tools/strip_proto.c, against brute-force walks: no
// direction that completes is ever skipped).
constexpr float STRIP_MARGIN = 1.0f;
__host__ __device__ __forceinline__ int strip_bins(int W, int H) { return W + H + 8; }
// c(q) = -uy x + ux y over the image starts at -(max(0, uy W) + max(0, -ux H)); + 2 keeps b - 1 non-negative
__device__ __forceinline__ int strip_offset(float ux, float uy, int W, int H)
{
    return (int) ceilf(fmaxf(0.0f, uy * (float) W) + fmaxf(0.0f, -ux * (float) H)) + 2;
}

__device__ void build_strips(const int4* __restrict__ bbox, int tiles_x, int tiles_y, int W, int H, float ux, float uy, int first_bin,
                             float2* __restrict__ out)
{
    __shared__ int s_ext[4][STRIP_TILES];           // per tile row: min x, max x; per tile column: min y, max y
    const int tid = threadIdx.x;
    for (int i = tid; i < STRIP_TILES; i += blockDim.x) { s_ext[0][i] = 1 << 30; s_ext[1][i] = -1; s_ext[2][i] = 1 << 30; s_ext[3][i] = -1; }
    __syncthreads();
    for (int t = tid; t < tiles_x * tiles_y; t += blockDim.x) {
        const int4 bb = bbox[t];
        if (bb.z < bb.x) continue;                              // a tile without a valid pixel
        const int ty = t / tiles_x, tx = t - ty * tiles_x;
        atomicMin(&s_ext[0][ty], bb.x); atomicMax(&s_ext[1][ty], bb.z);
        atomicMin(&s_ext[2][tx], bb.y); atomicMax(&s_ext[3][tx], bb.w);
    }
    __syncthreads();
    const int b = first_bin + tid;
    if (b >= strip_bins(W, H)) return;
    const float c0 = (float) (b - strip_offset(ux, uy, W, H)) - STRIP_MARGIN, c1 = c0 + 1.0f + 2.0f * STRIP_MARGIN;
    float lo = INFINITY, hi = -INFINITY;
    const bool steep = fabsf(uy) >= fabsf(ux);                  // the line crosses every row once: walk the tile rows
    const int n = steep ? tiles_y : tiles_x;
    const float ua = steep ? ux : uy, ub = steep ? uy : ux;     // a = the coordinate along a row (column), b = across
    const float inv = 1.0f / ub;
    for (int i = 0; i < n; i++) {
        const int e0 = s_ext[steep ? 0 : 2][i], e1 = s_ext[steep ? 1 : 3][i];
        if (e1 < 0) continue;
        const float b0 = (float) (i * (steep ? TH : TW)), b1 = b0 + (float) ((steep ? TH : TW) - 1);
        // steep: c = -uy x + ux y  =>  x = (ux y - c) / uy;   flat: y = (c + uy x) / ux
        const float v0 = steep ? (ua * b0 - c0) * inv : (c0 + ua * b0) * inv, v1 = steep ? (ua * b0 - c1) * inv : (c1 + ua * b0) * inv;
        const float v2 = steep ? (ua * b1 - c0) * inv : (c0 + ua * b1) * inv, v3 = steep ? (ua * b1 - c1) * inv : (c1 + ua * b1) * inv;
        float a0 = fminf(fminf(v0, v1), fminf(v2, v3)) - 0.01f, a1 = fmaxf(fmaxf(v0, v1), fmaxf(v2, v3)) + 0.01f;
        a0 = fmaxf(a0, (float) e0); a1 = fminf(a1, (float) e1);
        if (a0 > a1) continue;
        // t = ux x + uy y = ua a + ub b over [a0, a1] x [b0, b1]
        const float t0 = ua * a0 + ub * b0, t1 = ua * a0 + ub * b1, t2 = ua * a1 + ub * b0, t3 = ua * a1 + ub * b1;
        lo = fminf(lo, fminf(fminf(t0, t1), fminf(t2, t3)));
        hi = fmaxf(hi, fmaxf(fmaxf(t0, t1), fmaxf(t2, t3)));
    }
    out[b] = make_float2(lo, hi);
}

// One workgroup's share of a distance table: the 64 x 32 bits at (64 bx, 32 by) of a bit grid given by load(row, word)
// (0 outside the grid: nothing valid there), one byte per bit to out[row * pitch + col] for rows < n_rows, cols < n_cols.
template <int CAP, typename Load>
__device__ __forceinline__ void dilate_distances(Load load, int bx, int by, int pitch, int n_rows, int n_cols, uint8_t* __restrict__ out)
{
    constexpr int ROWS = DT_H + 2 * CAP;         // + halo: CAP rows above / below (one word left / right)
    static_assert(CAP <= 31, "five bit planes; the halo is one 32-pixel word on each side");
    __shared__ uint32_t buf[2][ROWS][DT_WORDS];
    const int tid = threadIdx.x;
    const int x0 = bx * DT_W, y0 = by * DT_H;
    const int w0 = (x0 >> 5) - 1, r0 = y0 - CAP;
    for (int i = tid; i < ROWS * DT_WORDS; i += 256) {
        const int r = i / DT_WORDS, w = i - r * DT_WORDS;
        buf[0][r][w] = load(r0 + r, w0 + w);
    }
    // the owner of an interior word counts, in five bit planes, for how many iterations each of its 32 bits stayed clear
    const bool owner = tid < DT_H * 2;
    const int orow = CAP + (tid >> 1), ow = 1 + (tid & 1);
    uint32_t p0 = 0, p1 = 0, p2 = 0, p3 = 0, p4 = 0;
    __syncthreads();
    int cur = 0;
    for (int k = 0; k < CAP; k++) {
        if (owner) {
            uint32_t c = ~buf[cur][orow][ow], t;
            t = p0 & c; p0 ^= c; c = t;
            t = p1 & c; p1 ^= c; c = t;
            t = p2 & c; p2 ^= c; c = t;
            t = p3 & c; p3 ^= c; c = t;
            p4 ^= c;
        }
        if (k + 1 < CAP) {
            for (int i = tid; i < ROWS * DT_WORDS; i += 256) {
                const int r = i / DT_WORDS, w = i - r * DT_WORDS;
                uint32_t v = 0;
#pragma unroll
                for (int dr = -1; dr <= 1; dr++) {
                    const int rr = r + dr;
                    if (rr < 0 || rr >= ROWS) continue;
                    const uint32_t m = buf[cur][rr][w];
                    const uint32_t l = w > 0 ? buf[cur][rr][w - 1] : 0u, rt = w + 1 < DT_WORDS ? buf[cur][rr][w + 1] : 0u;
                    v |= m | (m << 1) | (m >> 1) | (l >> 31) | (rt << 31);
                }
                buf[cur ^ 1][r][w] = v;
            }
            __syncthreads();
            cur ^= 1;
        }
    }
    if (owner) {
        const int y = y0 + (tid >> 1), xb = x0 + (tid & 1) * 32;
        if (y < n_rows && xb < n_cols) {
            uint8_t* o = out + (size_t) y * pitch + xb;
            const bool dwords = (pitch & 3) == 0 && xb + 32 <= n_cols;
            for (int j = 0; j < 32; j += 4) {
                uint32_t packed = 0;
#pragma unroll
                for (int b = 0; b < 4; b++) {
                    const int bit = j + b;
                    const uint32_t d = ((p0 >> bit) & 1u) | (((p1 >> bit) & 1u) << 1) | (((p2 >> bit) & 1u) << 2) | (((p3 >> bit) & 1u) << 3) | (((p4 >> bit) & 1u) << 4);
                    packed |= d << (8 * b);
                }
                if (dwords) *(uint32_t*) (o + j) = packed;
                else for (int b = 0; b < 4; b++) if (xb + j + b < n_cols) o[j + b] = (uint8_t) (packed >> (8 * b));
            }
        }
    }
}

// grid: the 64 x 32-pixel blocks of the image, then -- in block rows of their own -- the blocks of the strip tables and
// those of the coarse table (the same distance between 8 x 8-pixel blocks, from the tiles' `coarse` bits: where a
// block is k >= 2 blocks from the nearest block with a valid pixel, every pixel of it is at least 8 (k - 1) + 1 pixels
// from one -- jumps of up to 240 steps through empty space, looked up in LDS by the fill)
__global__ void __launch_bounds__(256) k_hole_dist(const uint32_t* __restrict__ mask, int W, int H, uint8_t* __restrict__ dist,
                                                   const int* __restrict__ hole_count, int min_holes,
                                                   const int4* __restrict__ bbox, int tiles_x, int tiles_y, float2* __restrict__ strips, FillDirs dirs,
                                                   const uint32_t* __restrict__ coarse, uint8_t* __restrict__ dist_blocks, int image_rows)
{
    if (*hole_count < min_holes) return;                        // few holes: the fill does not use the tables
    if ((int) blockIdx.y >= image_rows) {
        int si = ((int) blockIdx.y - image_rows) * (int) gridDim.x + (int) blockIdx.x;
        const int bins = strip_bins(W, H), per_dir = (bins + 255) / 256;
        if (si < 16 * per_dir) {
            const int d = si / per_dir;
            if (strips) build_strips(bbox, tiles_x, tiles_y, W, H, dirs.x[d], dirs.y[d], (si - d * per_dir) * 256, strips + (size_t) d * bins);
            return;
        }
        si -= 16 * per_dir;
        constexpr int CX = TW / 8, CY = TH / 8, TPW = 32 / CX;  // blocks per tile; tiles per 32-block word
        const int cw = tiles_x * CX, ch = tiles_y * CY;
        const int cgx = (cw + DT_W - 1) / DT_W, cgy = (ch + DT_H - 1) / DT_H;
        if (si >= cgx * cgy) return;
        const int cwpr = (cw + 31) >> 5;
        dilate_distances<KBE_DIST_CAP_BLOCKS>([=](int r, int wi) -> uint32_t {
            if (r < 0 || r >= ch || wi < 0 || wi >= cwpr) return 0u;
            const int ty = r / CY, sub = r - ty * CY;
            uint32_t word = 0;
            for (int t = 0; t < TPW; t++) {
                const int tx = wi * TPW + t;
                if (tx < tiles_x) word |= ((coarse[ty * tiles_x + tx] >> (CX * sub)) & ((1u << CX) - 1u)) << (CX * t);
            }
            return word;
        }, si % cgx, si / cgx, cw, ch, cw, dist_blocks);
        return;
    }
    const int wpr = (W + 31) >> 5;
    dilate_distances<KBE_DIST_CAP>([=](int y, int wi) -> uint32_t { return (y >= 0 && y < H && wi >= 0 && wi < wpr) ? mask[(size_t) y * wpr + wi] : 0u; },
                     (int) blockIdx.x, (int) blockIdx.y, W, H, W, dist);
}

// the contest's key holds an end's step count in 14 bits: a ray takes at most max(W, H) / 0.707 steps (larger frames
// fill with the other schedules)
inline bool fill_tables_fit(int W, int H) { return W <= 11000 && H <= 11000; }

// the extents the strip tables are built from: up to STRIP_TILES tile rows / columns
inline bool strips_fit(const Scratch& sc) { return sc.tiles_x <= STRIP_TILES && sc.tiles_y <= STRIP_TILES; }

void launch_hole_dist(hipStream_t s, const Scratch& sc, int W, int H, const int* hole_count, int min_holes, const FillDirs& dirs, const float2* strips)
{
    const int gx = (W + DT_W - 1) / DT_W, gy = (H + DT_H - 1) / DT_H;
    const int cw = sc.tiles_x * (TW / 8), ch = sc.tiles_y * (TH / 8);
    const int extra = 16 * ((strip_bins(W, H) + 255) / 256) + ((cw + DT_W - 1) / DT_W) * ((ch + DT_H - 1) / DT_H);
    hipLaunchKernelGGL(k_hole_dist, dim3(gx, gy + (extra + gx - 1) / gx), dim3(256), 0, s, sc.mask, W, H, sc.dist, hole_count, min_holes,
                       sc.bbox, sc.tiles_x, sc.tiles_y, (float2*) strips, dirs, sc.coarse, sc.dist_blocks, gy);
}

// m repeated fp32 additions a := a - u (or + u), exactly, in a few steps.  While a stays in one binade [2^e, 2^(e+1))
// every value of the chain is a multiple of q = 2^(e-23), and each rounded sum moves a by the SAME amount R = u rounded to
// a multiple of q: the exact sum lies between two neighbours of a's grid, and which one is nearer does not depend on a --
// unless u sits exactly half-way between two multiples of q (a tie: round-to-even looks at a).  j such sums are a -/+ j R,
// computed on the integer mantissa.  j is cut so that the chain, and one step beyond it on either side, stays inside the
// binade (no sum is rounded on a finer or a coarser grid); across a binade boundary, for ties, below 1 and for the last
// two steps the sums are added one at a time.  (tools/advance_check.c: against step-by-step sums, 24 M cases.)
// `limit`: positions below -1 or above limit + 1 are outside the image for good (the ray is monotone), where the value
// no longer matters: the direction is skipped (common.py:880-885).
__device__ __forceinline__ float advance_exact(float a, float u, int m, bool subtract, float limit)
{
    if (u == 0.0f) return a;
    while (m > 0) {
        const uint32_t bits = __float_as_uint(a);
        const int e = (int) (bits >> 23) - 127;
        if (m >= 3 && a >= 1.0f && e <= 23) {
            const float sc = ldexpf(u, 23 - e);                 // u / q, exact
            const float r = rintf(sc);
            if (fabsf(sc - r) != 0.5f) {
                const int step = (int) r, mag = abs(step);
                const int A = (int) ((bits & 0x7FFFFFu) | 0x800000u);       // a / q in [2^23, 2^24)
                const bool down = subtract ? step > 0 : step < 0;
                const int room_down = A - (1 << 23) - mag, room_up = (1 << 24) - 1 - mag - A;
                const int room = down ? room_down : room_up, other = down ? room_up : room_down;
                int j = (room > 0 && other >= 0 && mag > 0) ? (int) ((float) room / (float) mag) - 1 : 0;     // <= room / mag for sure
                j = min(j, m);
                if (j >= 1) {
                    const int end = A + j * (down ? -mag : mag);
                    a = __uint_as_float((bits & 0xFF800000u) | ((uint32_t) end & 0x7FFFFFu));
                    m -= j;
                    continue;
                }
            }
        }
        a = subtract ? a - u : a + u;
        m--;
        if (a < -1.0f || a > limit) break;
    }
    return a;
}

#if defined(KBE_FRAME_STATS)     // dev build only (tools/fill_stats.py)
__device__ unsigned long long g_fill_stats[8];      // holes walked, directions walked, fine look-ups, coarse look-ups, -, directions cut by the bound, directions skipped, skipped before a step
#define KBE_FILL_STAT(i, v) atomicAdd(&g_fill_stats[i], (unsigned long long) (v))
__device__ unsigned long long g_fill_hist[16];       // ray ends by the loop iterations they lived: [0] < 4, [1] < 8, ... doubling; [12] = the longest, [13] = steps of rays living >= 128 iterations, [14] = their iterations
#define KBE_FILL_RAY_DONE(iters, steps) do { int b_ = 0; while ((4 << b_) <= (iters) && b_ < 11) b_++; atomicAdd(&g_fill_hist[b_], 1ull); atomicMax(&g_fill_hist[12], (unsigned long long) (iters)); \
    if ((iters) >= 128) { atomicAdd(&g_fill_hist[13], (unsigned long long) (steps)); atomicAdd(&g_fill_hist[14], (unsigned long long) (iters)); } } while (0)
#else
#define KBE_FILL_STAT(i, v) ((void) 0)
#define KBE_FILL_RAY_DONE(iters, steps) ((void) 0)
#endif

// One coordinate of a ray end while it walks.  Fast mode (e >= 0): the coordinate is A 2^(e-23) with A in [2^23, 2^24),
// and one fp32 addition of -/+ u moves A by `step` (advance_exact's argument, kept as state): m additions are one
// multiply-add and one range test, the pixel a shift.  Invariant of the fast mode: A, and one step to either side of it,
// inside the binade.  Slow mode (e < 0; A holds the float's bits): below 32, next to a binade boundary, or a tie --
// single additions until the fast mode can be entered again.
struct Axis { int A, step, e; };

__device__ __forceinline__ bool axis_interior(int A, int mag) { return (unsigned) (A - (1 << 23) - mag) < (unsigned) ((1 << 23) - 2 * mag); }

__device__ __forceinline__ Axis axis_enter(float f, float u, bool subtract)
{
    const uint32_t bits = __float_as_uint(f);
    const int e = (int) (bits >> 23) - 127;
    if (f >= 32.0f && e <= 22) {                                // |step| <= 2^18: m * step cannot overflow, 2 |step| < 2^23
        const float sc = ldexpf(u, 23 - e);                     // u / q, exact
        const float r = rintf(sc);
        const int step = subtract ? -(int) r : (int) r;
        const int A = (int) ((bits & 0x7FFFFFu) | 0x800000u);
        if (fabsf(sc - r) != 0.5f && axis_interior(A, abs(step))) return Axis{ A, step, e };
    }
    return Axis{ (int) bits, 0, -1 };
}

__device__ __forceinline__ float axis_value(const Axis& ax)
{
    return ax.e >= 0 ? __uint_as_float(((uint32_t) (ax.e + 127) << 23) | ((uint32_t) ax.A & 0x7FFFFFu)) : __int_as_float(ax.A);
}

__device__ __forceinline__ int axis_pixel(const Axis& ax)      // (int) roundf(value): positive values round half up
{
    if (ax.e >= 0) { const int sh = 23 - ax.e; return (ax.A + (1 << (sh - 1))) >> sh; }
    return (int) roundf(__int_as_float(ax.A));
}

// r pending additions, all at once if they end inside the binade (and the invariant holds at the end: everything in between
// lies between two interior values)
__device__ __forceinline__ void axis_jump(Axis& ax, int& r)
{
    if (ax.e >= 0) {
        const int end = ax.A + r * ax.step;
        if (axis_interior(end, abs(ax.step))) { ax.A = end; r = 0; }
    }
}

// ... otherwise, typically in front of a binade boundary: as many as fit in front of it at once, four single additions in
// fp32 (that is across), whatever mode the value is in then, and the rest at once if they fit now.  What is left stays
// pending: the lane comes back in the next iteration of its loop.  Kept short on purpose -- in a wave of 64 some lane
// is here in almost every iteration (9 % of the advances: an image has a binade boundary in its middle), and the wave
// pays for its longest lane (a loop to completion here: 3/4 of the kernel's time).
__device__ __forceinline__ void axis_catch_up(Axis& ax, int& r, float u, bool subtract, float limit)
{
    if (u == 0.0f) { r = 0; return; }                           // a + 0 = a
    if (ax.e >= 0) {
        const int mag = max(1, abs(ax.step));
        const int room = ax.step < 0 ? ax.A - (1 << 23) - mag : (1 << 24) - 1 - mag - ax.A;
        const int j = min(r, (int) ((float) room * __builtin_amdgcn_rcpf((float) mag)) - 1);
        if (j >= 1 && axis_interior(ax.A + j * ax.step, mag)) { ax.A += j * ax.step; r -= j; }      // the test is what counts, j only a guess
    }
    float f = axis_value(ax);
#pragma unroll
    for (int i = 0; i < 4; i++) if (r > 0) { f = subtract ? f - u : f + u; r--; }       // :876-877 / :887-888
    if (f < -1.0f || f > limit) r = 0;                          // outside the image for good: the value no longer matters
    ax = axis_enter(f, u, subtract);
    if (r > 0) axis_jump(ax, r);
}

#ifndef KBE_FILL_BURST
#define KBE_FILL_BURST 8                // steps a creeping ray takes together ...
#endif
#ifndef KBE_FILL_BURST_LANES
#define KBE_FILL_BURST_LANES 16         // ... in a wave whose queue has run dry and of which no more lanes than this still walk
#endif
#ifndef KBE_FILL_REFILL_MIN
#define KBE_FILL_REFILL_MIN 16          // lanes of a wave that must be waiting before new work is fetched
#endif
#ifndef KBE_FILL_FINE_BELOW
#define KBE_FILL_FINE_BELOW 2           // coarse distances below this ask the fine table as well (longer jumps, one more load)
#endif
constexpr unsigned long long FILL_NO_ENTRY = ~0ull;
constexpr int FILL_MAX_STEPS = (1 << 14) - 1;
enum { END_IDLE = 0, END_WALK = 1, END_HIT = 2, END_DEAD = 3 };

__device__ __forceinline__ int swap_with_neighbour(int v)       // lanes 2i and 2i + 1 exchange v (all lanes active)
{
    return __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, false);       // quad_perm [1, 0, 3, 2]
}

// the box of all valid pixels from the tiles' boxes (every thread of the block gets it; s_bb: one int[4] per wave)
__device__ __forceinline__ void valid_box(const int4* __restrict__ bbox, int n_tiles, int W, int H, int (*s_bb)[4], int& bx0, int& by0, int& bx1, int& by1)
{
    bx0 = W; by0 = H; bx1 = -1; by1 = -1;
    for (int t = threadIdx.x; t < n_tiles; t += blockDim.x) {
        const int4 bb = bbox[t];
        bx0 = min(bx0, bb.x); by0 = min(by0, bb.y); bx1 = max(bx1, bb.z); by1 = max(by1, bb.w);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        bx0 = min(bx0, __shfl_xor(bx0, off)); by0 = min(by0, __shfl_xor(by0, off));
        bx1 = max(bx1, __shfl_xor(bx1, off)); by1 = max(by1, __shfl_xor(by1, off));
    }
    if ((threadIdx.x & 63) == 0) { s_bb[threadIdx.x >> 6][0] = bx0; s_bb[threadIdx.x >> 6][1] = by0; s_bb[threadIdx.x >> 6][2] = bx1; s_bb[threadIdx.x >> 6][3] = by1; }
    __syncthreads();
    for (int w = 0; w < (int) (blockDim.x >> 6); w++) { bx0 = min(bx0, s_bb[w][0]); by0 = min(by0, s_bb[w][1]); bx1 = max(bx1, s_bb[w][2]); by1 = max(by1, s_bb[w][3]); }
}

// The fill of a frame with very many holes, with the tables of k_hole_dist (launched in front of it; a kernel of its own so
// that its loop gets its own register allocation and code: inside k_fill_holes, next to the other two schedules, the same
// loop ran 20 % slower whenever code was added anywhere in that kernel).  `min_holes`: frames with fewer holes are left
// to k_fill_holes, which is launched behind this kernel in any case (housekeeping) and skips the frames filled here.
__global__ void __launch_bounds__(256) k_fill_tables(const int* __restrict__ holes, const int* __restrict__ hole_count, int min_holes,
                                                     const float* __restrict__ depth, int W, int H, FillDirs dirs, FillRect rect,
                                                     uint8_t* __restrict__ frame, float* __restrict__ render, int n_tiles,
                                                     const int4* __restrict__ bbox, int tiles_x, int tiles_y,
                                                     const uint8_t* __restrict__ dist, const float2* __restrict__ strips,
                                                     const uint8_t* __restrict__ dist_blocks)
{
    const int n = min(*hole_count, W * H);
    if (n < min_holes || (int) (blockIdx.x * blockDim.x) >= n) return;
    __shared__ int s_bb[4][4];
    // the block-distance table, two entries per byte (they are <= 15), if it fits: 8 KB hold a 1024 x 1024 frame's, and with
    // the queue and the slots a workgroup then needs < 20 KB, so that 8 of them share a CU
    static_assert(KBE_DIST_CAP_BLOCKS <= 15, "block distances are stored in 4 bits");
    __shared__ uint32_t s_pool[COARSE_WORDS];
    int bx0, by0, bx1, by1;
    valid_box(bbox, n_tiles, W, H, s_bb, bx0, by0, bx1, by1);
    // With the tables of k_hole_dist (launched in front of this kernel for frames expected to have very many holes).
    // A workgroup takes 256 holes at a time.
    // (1) One lane per hole: the strip test of its 16 directions; the directions that pass -- 4.5 of 16 on the
    //     dolly bench -- are queued in LDS.
    // (2) One lane per END of a queued (hole, direction), neighbouring lanes the two ends; ONE loop for everything:
    //     an iteration is one advance (exactly the fp32 sums of :876-889, on the integer mantissa: struct Axis) and
    //     one look-up -- the coarse table in LDS, and where that says "near" the fine table -- or, for lanes whose
    //     direction is decided, waiting until enough lanes wait to fetch new work together.  A direction is
    //     decided when one end leaves the image or its strip (skipped, :880-885 / :891-896), when both ends stand on
    //     valid pixels (it enters the hole's contest, :898-900, with an LDS atomicMin: the fp32 length of the span in
    //     the high word -- positive floats order like their bits -- then the direction: the reference keeps the
    //     FIRST direction of the shortest length, `best > dd` is strict; then the step counts of the two ends),
    //     or when its ends are already farther apart than a direction in the contest (they only move apart: it can
    //     neither win nor tie).
    // (3) One lane per hole: the winner's end points from its step counts, the fill.
    // One lane per hole for everything left 3/4 of the lanes idle in every direction and chained ~100 dependent
    // look-ups per lane (890 us per launch); loops nested per lane (per end, per jump) ran at ~20 % lane use.
    // Where the time still goes (tools/fill_stats.py, late dolly frames): while the queue has work 49 of 64 lanes
    // walk; after it has run dry the waves walk their last rays to the barrier with 6 lanes -- more than half of
    // all loop iterations, whatever the batch size: of 1.9 M ray ends 1.1 M live < 4 iterations and ~1 700 live
    // 128-335 (rays creeping through speckled regions at 1.3-1.9 steps per iteration).  Tried against that and
    // slower: the queue in HBM with persistent waves (a look-up per iteration at the hole's key in
    // L2 instead of LDS), batches of up to 1024 slots claimed from a cursor (3 instead of 5 workgroups per CU),
    // waves working on their own without any barrier (119 registers, 46 KB: occupancy 3), creeping rays taking 8
    // steps per iteration in sparse waves (the longest launch 894 -> 724 us, the average 385 -> 405), creeping
    // rays first in the queue, 6-8 waves per SIMD with the block table read from memory (no change).
    constexpr int FB = 256;
    static_assert(FB % 64 == 0 && FB * 16 <= 65536, "queue entries are 16 bits");
    __shared__ unsigned long long s_key[FB];
    __shared__ uint16_t s_queue[FB * 16];
    __shared__ int s_px[FB], s_wave_n[FB / 64], s_next;
    __shared__ uint8_t s_m0[FB];
    __shared__ float s_dir[2][16];
    __shared__ int s_off[16];
    const int cw = tiles_x * (TW / 8);
    const int c_bytes = cw * tiles_y * (TH / 8);
    const bool in_lds = c_bytes <= 2 * (int) sizeof(s_pool);
    if (in_lds)
        for (int i = threadIdx.x; i < (c_bytes + 7) / 8; i += blockDim.x) {
            const uint32_t lo = ((const uint32_t*) dist_blocks)[2 * i], hi = 8 * i + 4 < c_bytes ? ((const uint32_t*) dist_blocks)[2 * i + 1] : 0u;
            // bytes b0..b7 -> nibbles: entry 2j in the low half of byte j
            s_pool[i] = (lo & 0xFu) | ((lo >> 4) & 0xF0u) | ((lo >> 8) & 0xF00u) | ((lo >> 12) & 0xF000u) |
                        ((hi & 0xFu) << 16) | (((hi >> 4) & 0xF0u) << 16) | (((hi >> 8) & 0xF00u) << 16) | (((hi >> 12) & 0xF000u) << 16);
        }
    const auto block_distance = [&](int ci) -> int {           // blocks to the nearest block with a valid pixel
        return in_lds ? (((const uint8_t*) s_pool)[ci >> 1] >> ((ci & 1) << 2)) & 15 : (int) dist_blocks[ci];
    };
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < 16) { s_dir[0][tid] = dirs.x[tid]; s_dir[1][tid] = dirs.y[tid]; s_off[tid] = strip_offset(dirs.x[tid], dirs.y[tid], W, H); }
    const int bins = strip_bins(W, H);
    const size_t HW = (size_t) W * H;
    const bool is_b = lane & 1;                         // the end walking towards +u
    for (int base = blockIdx.x * FB; base < n; base += gridDim.x * FB) {
        __syncthreads();                                // the tables are loaded / the previous batch is done with the LDS arrays
        // (1)
        const int h = base + tid;
        int px = -1, x = 0, y = 0;
        uint32_t pass = 0;
        if (h < n) {
            px = holes[h];
            y = px / W; x = px - y * W;
            // outside the rectangle to be filled / outside the box of the valid pixels: every direction is skipped
            if (x < rect.x0 || x > rect.x1 || y < rect.y0 || y > rect.y1 || x < bx0 || x > bx1 || y < by0 || y > by1) px = -1;
        }
        if (px >= 0) {
            KBE_FILL_STAT(0, 1);
            pass = 0xFFFFu;
            if (strips) {
                pass = 0;
#pragma unroll
                for (int d = 0; d < 16; d++) {
                    const float ddx = s_dir[0][d], ddy = s_dir[1][d];
                    const float c = ddx * (float) y - ddy * (float) x, t = ddx * (float) x + ddy * (float) y;
                    const float2 lh = strips[(size_t) d * bins + ((int) floorf(c) + s_off[d])];
                    if (!(lh.x > t + STRIP_MARGIN || lh.y < t - STRIP_MARGIN)) pass |= 1u << d;     // valid pixels on both sides
                }
            }
            if (pass) {
                const int c_here = block_distance((y >> 3) * cw + (x >> 3));
                s_m0[tid] = (uint8_t) (c_here >= 2 ? 8 * (c_here - 1) : max(1, (int) dist[(uint32_t) px] - 1));
            }
        }
        s_px[tid] = px;
        s_key[tid] = FILL_NO_ENTRY;
        if (tid == 0) s_next = 0;
        const int mine = __popc(pass);
        int incl = mine;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const int o = __shfl_up(incl, off); if (lane >= off) incl += o; }
        if (lane == 63) s_wave_n[wave] = incl;
        __syncthreads();
        int at = incl - mine, total = 0;
        for (int w = 0; w < FB / 64; w++) { if (w < wave) at += s_wave_n[w]; total += s_wave_n[w]; }
        for (uint32_t m = pass; m; m &= m - 1) s_queue[at++] = (uint16_t) ((tid << 4) | (__ffs(m) - 1));
        __syncthreads();
        // (2)
        {
            int st = END_IDLE, rx = 0, ry = 0, k = 0, ix = 0, iy = 0, slot = 0, d = 0;
#if defined(KBE_FRAME_STATS)
            int iters = 0;
#endif
            float ux = 0.0f, uy = 0.0f, bound = 0.0f, inv_umax = 1.0f;
            Axis X = { 0, 0, -1 }, Y = { 0, 0, -1 };
            // the two ends of a direction look at each other
            const auto look = [&]() {
                    const int pst = swap_with_neighbour(st), pix = swap_with_neighbour(ix), piy = swap_with_neighbour(iy), pk = swap_with_neighbour(k);
                    if (st != END_IDLE) {
                        if (st == END_DEAD || pst == END_DEAD) st = END_IDLE;
                        else {
                            const float ex = (float) (ix - pix), ey = (float) (iy - piy);
                            const float ssq = ex * ex + ey * ey;                            // exact: small integers
                            // the best length in the contest, squared and rounded up a little: a span whose square is
                            // above that has a longer fp32 length (sqrtf is monotone and correctly rounded)
                            const float best = __uint_as_float(((const volatile uint32_t*) &s_key[slot])[1]);          // no entry yet: NaN
                            if (st == END_HIT && pst == END_HIT) {
                                const int ka = is_b ? pk : k, kb = is_b ? k : pk;
                                const float dd = sqrtf(ssq);                                // :898
                                if (!is_b && 1000000.0f > dd && ka <= FILL_MAX_STEPS && kb <= FILL_MAX_STEPS)      // :854, :900
                                    atomicMin(&s_key[slot], ((unsigned long long) __float_as_uint(dd) << 32) | ((unsigned long long) d << 28) |
                                                            ((unsigned long long) ka << 14) | (unsigned long long) kb);
                                st = END_IDLE;
                            } else if (ssq > best * best * 1.000001f) {                     // NaN: never true
                                KBE_FILL_STAT(5, is_b ? 0 : 1);
                                st = END_IDLE;
                            }
                        }
                    }
            };
            const auto step = [&]() {
                    // one advance, one look-up
                    if (st == END_WALK) {
                        axis_jump(X, rx);
                        axis_jump(Y, ry);
                        if (rx | ry) {                      // one of them did not get there: one catch-up, for one axis
                            const bool on_x = rx > 0;
                            Axis a = on_x ? X : Y;
                            int r = on_x ? rx : ry;
                            KBE_FILL_STAT(4, 1);
                            axis_catch_up(a, r, on_x ? ux : uy, !is_b, on_x ? (float) W : (float) H);
                            if (on_x) { X = a; rx = r; } else { Y = a; ry = r; }
                        }
                        if ((rx | ry) == 0) {
                            ix = axis_pixel(X); iy = axis_pixel(Y);
                            const float t = ux * (float) ix + uy * (float) iy;
                            int m = 0;
                            if (!(((unsigned) ix < (unsigned) W) & ((unsigned) iy < (unsigned) H))) st = END_DEAD;       // :880-885 / :891-896
                            else if (is_b ? bound < t - STRIP_MARGIN : bound > t + STRIP_MARGIN) st = END_DEAD;         // past every valid pixel of its strip
                            else {
                                const int ci = (iy >> 3) * cw + (ix >> 3);
                                const int c = block_distance(ci);
                                KBE_FILL_STAT(3, 1);
                                // With the nearest valid pixel D away (Chebyshev) from this one, the pixel j steps
                                // on is at most j max(|ux|, |uy|) + 1 away from this one (the steps; the rounding of
                                // both positions; < 0.03 of drift): a hole for sure while j umax + 1.03 < D.  The first
                                // position to look at is step ceil((D - 1.03) / umax).
                                if (c >= KBE_FILL_FINE_BELOW) m = (int) ceilf((float) (8 * (c - 1)) * inv_umax - 0.03f);   // D >= 8 (c - 1) + 1
                                else {
                                    const int dn = dist[(uint32_t) iy * (uint32_t) W + (uint32_t) ix];
                                    KBE_FILL_STAT(2, 1);
                                    if (dn == 0) st = END_HIT;          // depth > 0 (:882 / :893)
                                    else m = max(c >= 2 ? (int) ceilf((float) (8 * (c - 1)) * inv_umax - 0.03f) : 1, (int) ceilf(((float) dn - 1.03f) * inv_umax));
                                }
                            }
                            rx = ry = m;
                            k += m;
                        }
                    }
            };
            // A ray creeping through a speckled region (a valid pixel next to every position, none on the ray) takes one
            // step per look-up for a hundred iterations and more -- ~1 700 of the 1.9 M ray ends of a late dolly frame live
            // 128-335 iterations -- and its workgroup waits for it.  Once the queue has run dry and few lanes of the wave still
            // walk, such a ray takes its next KBE_FILL_BURST steps together: the positions do not depend on what is found
            // there, so their look-ups go out at once, and the first that ends the ray (a valid pixel, the image border, the
            // end of its strip) counts.  In a loop of its own: the same code inside the main loop made that one 18 % slower
            // without ever running.
            const auto creep = [&]() -> bool {
                if (!(st == END_WALK && rx == ry && rx >= 1 && rx <= 2)) return false;
                constexpr int B = KBE_FILL_BURST;
                float fx = axis_value(X), fy = axis_value(Y);
                int bpx[B], bpy[B], bdn[B];
                bool bstop[B];
#pragma unroll
                for (int j = 0; j < B; j++) {
                    fx = is_b ? fx + ux : fx - ux;                      // :876-877 / :887-888
                    fy = is_b ? fy + uy : fy - uy;
                    bpx[j] = (int) roundf(fx); bpy[j] = (int) roundf(fy);
                    const bool inb = ((unsigned) bpx[j] < (unsigned) W) & ((unsigned) bpy[j] < (unsigned) H);
                    const float t = ux * (float) bpx[j] + uy * (float) bpy[j];
                    bstop[j] = !inb || (is_b ? bound < t - STRIP_MARGIN : bound > t + STRIP_MARGIN);
                    bdn[j] = dist[inb ? (uint32_t) bpy[j] * (uint32_t) W + (uint32_t) bpx[j] : 0u];
                }
                KBE_FILL_STAT(2, B);
                k -= rx;                                                // the pending steps are among these
                bool decided = false;
#pragma unroll
                for (int j = 0; j < B; j++) {
                    if (decided) continue;
                    if (bstop[j]) { st = END_DEAD; decided = true; }
                    else if (bdn[j] == 0) { st = END_HIT; ix = bpx[j]; iy = bpy[j]; k += j + 1; decided = true; }
                }
                if (!decided) {
                    ix = bpx[B - 1]; iy = bpy[B - 1];
                    X = axis_enter(fx, ux, !is_b);
                    Y = axis_enter(fy, uy, !is_b);
                    const int m = max(1, (int) ceilf(((float) bdn[B - 1] - 1.03f) * inv_umax));
                    rx = ry = m;
                    k += B + m;
                }
                return true;
            };
            bool stragglers = false;
            for (;;) {
                look();
#if defined(KBE_FRAME_STATS)
                if (st == END_IDLE && iters > 0) { KBE_FILL_RAY_DONE(iters, k); iters = 0; }
                if (st == END_WALK) iters++;
#endif
                // new work, for a quarter of the wave at a time (fetching runs at the pace of its slowest lane)
                const unsigned long long idle = __ballot(st == END_IDLE);
                if (idle) {
                    const int next = *(const volatile int*) &s_next;
                    if (next >= total) {
                        if (idle == ~0ull) break;
                        if (64 - __popcll(idle) <= KBE_FILL_BURST_LANES) { stragglers = true; break; }
                    }
                    else if (__popcll(idle) >= KBE_FILL_REFILL_MIN || idle == ~0ull) {
                        const int n_pairs = __popcll(idle) >> 1;
                        int first = 0;
                        if (lane == (int) __ffsll((long long) idle) - 1) first = atomicAdd(&s_next, n_pairs);
                        first = __shfl(first, (int) __ffsll((long long) idle) - 1);
                        const int q = first + (__popcll(idle & ((1ull << lane) - 1ull)) >> 1);
                        if (st == END_IDLE && q < total) {
                            const int e = s_queue[q];
                            slot = e >> 4; d = e & 15;
                            const int qpx = s_px[slot];
                            iy = qpx / W; ix = qpx - iy * W;
                            ux = s_dir[0][d]; uy = s_dir[1][d];
                            inv_umax = 0.999999f / fmaxf(fabsf(ux), fabsf(uy));
                            bound = is_b ? INFINITY : -INFINITY;
                            if (strips) {
                                const float2 lh = strips[(size_t) d * bins + ((int) floorf(ux * (float) iy - uy * (float) ix) + s_off[d])];
                                bound = is_b ? lh.y : lh.x;
                            }
                            X = axis_enter((float) ix, ux, !is_b);
                            Y = axis_enter((float) iy, uy, !is_b);
                            rx = ry = k = s_m0[slot];
                            st = END_WALK;
                            KBE_FILL_STAT(1, is_b ? 0 : 1);
                        }
                    }
                }
#if defined(KBE_FRAME_STATS)
                { const unsigned long long w = __ballot(st == END_WALK), hw = __ballot(st == END_HIT);
                  const bool empty = *(const volatile int*) &s_next >= total;
                  if (lane == 0) { KBE_FILL_STAT(6, 1ull | (empty ? 1ull << 32 : 0ull)); KBE_FILL_STAT(7, (unsigned long long) __popcll(w) | (empty ? (unsigned long long) __popcll(w) << 32 : 0ull));
                                   KBE_FILL_STAT(0, (unsigned long long) __popcll(hw) << 32); } }
#endif
                step();
            }
            // the wave's last rays (the queue has run dry, <= KBE_FILL_BURST_LANES lanes still walk): creeping ones in bursts
            if (stragglers)
                for (;;) {
                    look();
                    if (__ballot(st != END_IDLE) == 0ull) break;
#if defined(KBE_FRAME_STATS)
                    if (st == END_IDLE && iters > 0) { KBE_FILL_RAY_DONE(iters, k); iters = 0; }
                    if (st == END_WALK) iters++;
                    { const unsigned long long w = __ballot(st == END_WALK); if (lane == 0) { KBE_FILL_STAT(6, 1ull | (1ull << 32)); KBE_FILL_STAT(7, (unsigned long long) __popcll(w) | ((unsigned long long) __popcll(w) << 32)); } }
#endif
                    if (!creep()) step();
                }
        }
        __syncthreads();
        // (3)
        const unsigned long long key = s_key[tid];
        if (px >= 0 && key != FILL_NO_ENTRY) {
            const int d = (int) (key >> 28) & 15, ka = (int) (key >> 14) & FILL_MAX_STEPS, kb = (int) key & FILL_MAX_STEPS;
            const float ddx = s_dir[0][d], ddy = s_dir[1][d];
            const int sax = (int) roundf(advance_exact((float) x, ddx, ka, true, INFINITY)), say = (int) roundf(advance_exact((float) y, ddy, ka, true, INFINITY));
            const int sbx = (int) roundf(advance_exact((float) x, ddx, kb, false, INFINITY)), sby = (int) roundf(advance_exact((float) y, ddy, kb, false, INFINITY));
            int sx = sax, sy = say;
            if (depth[(size_t) say * W + sax] < depth[(size_t) sby * W + sbx]) { sx = sbx; sy = sby; }     // :904 the farther (background) end
            const size_t src = (size_t) sy * W + sx, o = (size_t) px;
            frame[o * 3] = frame[src * 3]; frame[o * 3 + 1] = frame[src * 3 + 1]; frame[o * 3 + 2] = frame[src * 3 + 2];
            if (render) for (int c = 0; c < 4; c++) render[c * HW + o] = render[c * HW + src];
        }
    }
}

#ifndef KBE_FILL_BLOCK
#define KBE_FILL_BLOCK 256
#endif
#ifndef KBE_FILL_MAX_BLOCKS
#define KBE_FILL_MAX_BLOCKS 2048
#endif
__global__ void __launch_bounds__(KBE_FILL_BLOCK) k_fill_holes(const int* __restrict__ holes, const int* __restrict__ hole_count,
                                                    const float* __restrict__ depth, const uint32_t* __restrict__ mask, int W, int H,
                                                    FillDirs dirs, FillRect rect,
                                                    uint8_t* __restrict__ frame, float* __restrict__ render,
                                                    uint32_t* __restrict__ zkeys, int* __restrict__ tile_count, int n_tiles,
                                                    const int4* __restrict__ bbox, int fill_mode, const uint32_t* __restrict__ coarse,
                                                    int tiles_x, int tiles_y, int reset_scatter_scratch, int* __restrict__ next_hole_count,
                                                    int tables)
{
    // leave the scratch ready for the next frame.  Bucket path: empty z-buffer, empty buckets.  Fused path: it has
    // neither; its hole counters alternate between frames, and this launch zeroes the one the NEXT frame will count in
    // (nobody reads or writes that one while this launch runs).
    if (reset_scatter_scratch) {
        const int gtid = blockIdx.x * blockDim.x + threadIdx.x, gsz = gridDim.x * blockDim.x;
        for (int i = gtid; i < W * H; i += gsz) zkeys[i] = KBE_ZKEY_EMPTY;
        for (int i = gtid; i < n_tiles; i += gsz) tile_count[i * CNT_STRIDE] = 0;
    }
    if (next_hole_count && blockIdx.x == 0 && threadIdx.x == 0) *next_hole_count = 0;
    const int n = min(*hole_count, W * H);
    if (tables && n >= tables - 1) return;                      // k_fill_tables, launched in front of this kernel, filled this frame
    // A ray is a straight line, monotone in x and in y.  Once it is outside the bounding box of the valid
    // pixels on a side it is not moving back from, it can never meet one: its outcome is "left the image"
    // (common.py:880-885) without walking there.  Exact, and it is what makes a zoomed-out (dolly) frame,
    // where most of the image is empty border, cheap.
    __shared__ int s_bb[KBE_FILL_BLOCK / 64][4];
    if ((int) (blockIdx.x * (blockDim.x >> 5)) >= n) return;    // no hole for this block (whole block: uniform)
    int bx0, by0, bx1, by1;
    valid_box(bbox, n_tiles, W, H, s_bb, bx0, by0, bx1, by1);
    const int wpr = (W + 31) >> 5;              // mask words per row
    // fill_mode: 0 = by hole count (the multi-lane frame loop: the per-lane schedule does less work but has long
    // dependent chains, which only pays when other frames' kernels fill the chip meanwhile), 1 = one lane per hole,
    // 2 = one half-wave per hole (a frame rendered on its own)
    if (fill_mode == 1 || (fill_mode == 0 && n >= KBE_FILL_SERIAL_MIN)) {       // uniform over the launch
        // Coarse map for the walks (LDS): bit (cy, cx) = some 8 x 8 block within one block of (cx, cy) holds a valid
        // pixel.  Eight steps of a ray stay within 7 pixels of where they end, i.e. inside the 3 x 3 blocks around the end
        // position's block; if that neighbourhood has no valid pixel the eight steps cannot hit one and are taken at
        // once (16 additions, the same fp32 sums, no rounding of the positions in between, no mask look-ups).
        if ((int) (blockIdx.x * blockDim.x) >= n) return;       // no hole for this block in this schedule either
        __shared__ uint32_t s_pool[2 * COARSE_WORDS];           // the coarse maps of either schedule
        uint32_t* s_blk = s_pool, *s_near = s_pool + COARSE_WORDS;
        constexpr int CX = TW / 8, CY = TH / 8;                 // coarse blocks per tile
        const int c_rows = tiles_y * CY, c_wpr = (tiles_x * CX + 31) >> 5;
        const bool skip_ok = c_rows * c_wpr <= COARSE_WORDS;
        if (skip_ok) {
            for (int idx = threadIdx.x; idx < c_rows * c_wpr; idx += blockDim.x) {
                const int r = idx / c_wpr, wi = idx - r * c_wpr;
                const int ty = r / CY, sub = r - ty * CY;
                uint32_t word = 0;
                for (int t = 0; t < 32 / CX; t++) {
                    const int tx = wi * (32 / CX) + t;
                    if (tx < tiles_x) word |= ((coarse[ty * tiles_x + tx] >> (CX * sub)) & ((1u << CX) - 1u)) << (CX * t);
                }
                s_blk[idx] = word;
            }
            __syncthreads();
            for (int idx = threadIdx.x; idx < c_rows * c_wpr; idx += blockDim.x) {
                const int r = idx / c_wpr, wi = idx - r * c_wpr;
                uint32_t near = 0;
                for (int dr = -1; dr <= 1; dr++) {
                    const int rr = r + dr;
                    if (rr < 0 || rr >= c_rows) continue;
                    const uint32_t w0 = s_blk[rr * c_wpr + wi];
                    const uint32_t wl = wi > 0 ? s_blk[rr * c_wpr + wi - 1] : 0u, wr = wi + 1 < c_wpr ? s_blk[rr * c_wpr + wi + 1] : 0u;
                    near |= w0 | (w0 << 1) | (w0 >> 1) | (wl >> 31) | (wr << 31);
                }
                s_near[idx] = near;
            }
            __syncthreads();
        }
        const int gtid = blockIdx.x * blockDim.x + threadIdx.x, gsz = gridDim.x * blockDim.x;
        for (int h = gtid; h < n; h += gsz) {
            const int px = holes[h];
            const int y = px / W, x = px - y * W;
            if (x < rect.x0 || x > rect.x1 || y < rect.y0 || y > rect.y1) continue;
            fill_hole_serial(px, depth, mask, W, H, wpr, dirs, bx0, by0, bx1, by1, frame, render, skip_ok ? s_near : nullptr, c_wpr);
        }
        return;
    }
    const int lane = threadIdx.x & 31;
    const int group = (blockIdx.x * blockDim.x + threadIdx.x) >> 5, n_groups = (gridDim.x * blockDim.x) >> 5;
    const int d = lane >> 1, end = lane & 1;
    const float ddx = end ? dirs.x[d] : -dirs.x[d], ddy = end ? dirs.y[d] : -dirs.y[d];
    const size_t HW = (size_t) W * H;
    for (int h = group; h < n; h += n_groups) {
        const int px = holes[h];
        const int y = px / W, x = px - y * W;
        if (x < rect.x0 || x > rect.x1 || y < rect.y0 || y > rect.y1) continue;
        float fx = (float) x, fy = (float) y;
        int ix = 0, iy = 0;
        bool ok = false;
        bool done = (x < bx0 && ddx <= 0.0f) || (x > bx1 && ddx >= 0.0f) || (y < by0 && ddy <= 0.0f) || (y > by1 && ddy >= 0.0f);
        // if either end of a direction is hopeless the direction is skipped (:884-885, :895-896): do not walk the other end
        done = done || (bool) __shfl_xor((int) done, 1);
        // common.py:876-883 / :887-894.  The positions do not depend on the data, so the walk issues a batch
        // of mask loads at a time and then inspects them in order (the dependent-load chain of the textbook
        // loop is avoidable latency).  Batches of 8: larger ones (32 for rays still going) measured slower --
        // what bounds a frame with few, long rays (one running along a thin disocclusion strip for hundreds
        // of pixels) is the serial fp32 position update of a lone wave, not the loads.
        auto walk = [&](auto batch_tag) {
            constexpr int kBatch = decltype(batch_tag)::value;
            int bx[kBatch], by[kBatch];
            uint32_t bw[kBatch];
            bool bin[kBatch];
            // branch-free: a step outside the image reads word 0 and is flagged; nothing below sits under a branch
            // (this kernel is instruction-bound on frames with many holes: ~30 instructions per step instead of ~40)
#pragma unroll
            for (int k = 0; k < kBatch; k++) {
                fx += ddx; bx[k] = (int) roundf(fx);
                fy += ddy; by[k] = (int) roundf(fy);
                bin[k] = ((unsigned) bx[k] < (unsigned) W) & ((unsigned) by[k] < (unsigned) H);
                const unsigned mi = bin[k] ? (unsigned) by[k] * (unsigned) wpr + ((unsigned) bx[k] >> 5) : 0u;
                bw[k] = mask[mi];
            }
            bool stop = false;                                          // a step of this batch ended the walk
#pragma unroll
            for (int k = 0; k < kBatch; k++) {
                const bool hit = bin[k] && ((bw[k] >> (bx[k] & 31)) & 1u);      // depth > 0 (common.py:882 / :893)
                const bool take = !stop;                                // the first ending step fixes position and outcome
                ix = take ? bx[k] : ix; iy = take ? by[k] : iy;
                ok = ok || (take && hit);
                stop = stop || hit || !bin[k];
            }
            // ended, or left the box of valid pixels for good?
            done = stop || (ix < bx0 && ddx <= 0.0f) || (ix > bx1 && ddx >= 0.0f) || (iy < by0 && ddy <= 0.0f) || (iy > by1 && ddy >= 0.0f);
        };
        // Branch and bound over the 16 directions (exact).  The winner is the direction whose two hits are
        // STRICTLY closest (:900, first direction on ties).  The two ends of a direction move apart monotonically,
        // so the distance between their CURRENT positions, computed with the arithmetic of :898, bounds the
        // distance between their eventual hits from below (fp32 multiply, add and sqrt are monotone).  Once a
        // direction is complete, every direction whose bound already exceeds it can stop: it could never be
        // strictly shorter.  A hole in a thin disocclusion strip thus costs the 2-3 steps across the strip, not
        // the hundreds along it, and a wide hole the walk to its nearest rim, not to its farthest.
        int ox = 0, oy = 0;
        bool ook = false;
        float dist = INFINITY, best = INFINITY;
        for (;;) {
            if (!done) walk(std::integral_constant<int, 8>());
            ox = __shfl_xor(ix, 1); oy = __shfl_xor(iy, 1);     // the other end of my direction
            ook = (bool) __shfl_xor((int) ok, 1);
            const bool odone = (bool) __shfl_xor((int) done, 1);
            const float ex = (float) (ix - ox), ey = (float) (iy - oy);
            const float cur = sqrtf(ex * ex + ey * ey);         // :898 on the current positions
            dist = (ok && ook && 1000000.0f > cur) ? cur : INFINITY;    // :854 + :900 against the initial dblShortest
            best = dist;
#pragma unroll
            for (int off = 2; off < 32; off <<= 1) best = fminf(best, __shfl_xor(best, off));
            // stop: the other end is hopeless (:884-885, :895-896), or this direction can no longer win
            done = done || (odone && !ook) || cur > best;
            if (__all(done)) break;
        }
        if (best == INFINITY) continue;                         // unfillable: keeps the rendered value (:913-919)
        const unsigned long long m = __ballot(dist == best);
        const unsigned mine = (unsigned) (m >> (threadIdx.x & 32));     // my 32-lane half
        const int win = __ffs((int) mine) - 1;                  // lowest lane = lowest direction, its `from` end
        if (lane == win) {
            // lane `win` is the `from` end (even lane); partner values are the `to` end
            int sxp = ix, syp = iy;
            if (depth[(size_t) iy * W + ix] < depth[(size_t) oy * W + ox]) { sxp = ox; syp = oy; }     // :904 the farther (background) end
            const size_t s = (size_t) syp * W + sxp, o = (size_t) px;
            frame[o * 3] = frame[s * 3]; frame[o * 3 + 1] = frame[s * 3 + 1]; frame[o * 3 + 2] = frame[s * 3 + 2];
            if (render) for (int c = 0; c < 4; c++) render[c * HW + o] = render[c * HW + s];
        }
    }
}

// ---------------------------------------------------------------------------------------
// frame hand-off (common.py:255 `.cpu()`): finished uint8 frames go from the lane's device buffers into pinned host
// memory IN THE LANE'S OWN STREAM -- no copy stream, no event (cross-stream events cost ~25 us per frame on this
// stack: a dedicated copy stream measured 89-107 us per frame) -- either by the runtime's transfer engine, one
// hipMemcpyAsync per group of frames (default), or by k_deliver below, one frame at a time.
// Left alone, the lanes' copies share the PCIe link, finish together, and the lanes then render together: a convoy
// that leaves the link idle a quarter of the time (measured: 80 us per 1024^2 frame, 39 GB/s).  So the copies take
// TURNS: copy i waits (one lane polling, s_sleep in between) until copy i - 1 has finished and then has the link
// to itself; the lanes fall into a staggered pipeline -- with two lanes, one renders its next group while the other's
// group leaves -- and the link is busy back to back (59 us per frame = 53 GB/s of the ~57 the link gives a single
// large transfer; tools/d2h_probe*.hip, gpurun_out sweeps in DESIGN.md).  The turn is ADVISORY -- a performance
// ordering only: the wait is bounded (~4 ms) and a copy that gives up simply copies, so no mapping of streams onto
// hardware queues can deadlock it.
// k_deliver: a copy kernel with plain 16-byte stores into device-visible host memory.  64 unthrottled workgroups
// reach 55 GB/s alone, but PCIe-bound stores parked in the write queues stall every other kernel's stores; 16
// workgroups with 2 KB in flight per wave are the best compromise found (66 us per frame next to 3 rendering lanes).
// ---------------------------------------------------------------------------------------
constexpr int DELIVER_BLOCKS = 16, DELIVER_THREADS = 256, DELIVER_KB_PER_WAVE = 2;
constexpr int DELIVER_MAX_POLLS = 4096;     // x ~1 us
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
struct DeliverCtl { uint32_t serving; uint32_t pad[31]; uint32_t done[32]; };
static_assert(sizeof(DeliverCtl) <= 256, "kbe_video_stage_bytes reserves 256 bytes");      // serving: copies finished so far; done: per-copy workgroup count

// the turn of a runtime transfer: pass == 0 waits (bounded) until `ticket` is served, pass == 1 hands the turn on
__global__ void __launch_bounds__(64) k_turn(DeliverCtl* ctl, uint32_t ticket, int pass)
{
    if (threadIdx.x != 0) return;
    if (pass) { atomicMax(&ctl->serving, ticket + 1); return; }
    for (int polls = 0; polls < DELIVER_MAX_POLLS; polls++) {
        if (__hip_atomic_load(&ctl->serving, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= ticket) break;
        __builtin_amdgcn_s_sleep(32);
    }
}

__global__ void __launch_bounds__(DELIVER_THREADS) k_deliver(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, size_t bytes,
                                                             DeliverCtl* ctl, uint32_t ticket)
{
    if (ctl) {
        if (threadIdx.x == 0) {
            for (int polls = 0; polls < DELIVER_MAX_POLLS; polls++) {
                if (__hip_atomic_load(&ctl->serving, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= ticket) break;
                __builtin_amdgcn_s_sleep(32);
            }
        }
        __syncthreads();
    }
    const size_t gtid = (size_t) blockIdx.x * blockDim.x + threadIdx.x, gsz = (size_t) gridDim.x * blockDim.x;
    if ((((uintptr_t) src | (uintptr_t) dst) & 15) == 0) {          // uniform; the normal case (W*H*3 a multiple of 16)
        const size_t n16 = bytes >> 4;
        const u32x4* s16 = (const u32x4*) src;
        u32x4* d16 = (u32x4*) dst;
        // at most DELIVER_KB_PER_WAVE KB of stores in flight per wave: the link is fed (its bandwidth-delay product is
        // ~100 KB) without parking megabytes of PCIe-bound writes in the L2 / fabric write queues, where every
        // other kernel's stores wait behind them (measured: k_tiles 19 -> 89 us next to an unthrottled 64-workgroup copy)
        for (size_t i0 = gtid; i0 < n16; i0 += gsz * DELIVER_KB_PER_WAVE) {
#pragma unroll
            for (int k = 0; k < DELIVER_KB_PER_WAVE; k++) {
                const size_t i = i0 + (size_t) k * gsz;
                if (i < n16) d16[i] = __builtin_nontemporal_load(s16 + i);      // the frame is read once: no L2 allocation
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        for (size_t i = (n16 << 4) + gtid; i < bytes; i += gsz) dst[i] = src[i];
    } else {
        for (size_t i = gtid; i < bytes; i += gsz) dst[i] = src[i];
    }
    if (ctl) {
        // the last workgroup to get here passes the turn on (its own stores need not have landed: the next copy only
        // competes for the link, it does not read them)
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t* cnt = &ctl->done[ticket & 31];
            if (atomicAdd(cnt, 1u) == gridDim.x - 1) {
                __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                atomicMax(&ctl->serving, ticket + 1);
            }
        }
    }
}

// the hole fill of one frame: with KBE_STAGE_FILL_DIST the tables and the table-driven fill in front of k_fill_holes (each of
// them returns at once when the frame has fewer holes than the schedule asks for)
void launch_fill(hipStream_t s, const Scratch& sc, int W, int H, const int* hole_count, int stages, const FillDirs& dirs, const FillRect& rect,
                 unsigned fill_blocks, uint8_t* frame_u8, float* render_f32, int n_tiles, int reset_scatter_scratch, int* next_hole_count)
{
    int tables = 0;
    const int fill_mode = (stages & KBE_STAGE_FILL_PER_LANE) ? 1 : ((stages & KBE_STAGE_FILL_PER_HALFWAVE) || !(stages & KBE_STAGE_FILL_BY_COUNT) ? 2 : 0);
    if ((stages & KBE_STAGE_FILL_DIST) && (stages & (KBE_STAGE_FILL_PER_LANE | KBE_STAGE_FILL_BY_COUNT)) && fill_tables_fit(W, H)) {
        const int min_holes = (stages & KBE_STAGE_FILL_PER_LANE) ? 0 : KBE_FILL_SERIAL_MIN;
        const float2* strips = strips_fit(sc) ? sc.strips : nullptr;
        launch_hole_dist(s, sc, W, H, hole_count, min_holes, dirs, strips);
        const size_t hw = (size_t) W * H;
        const unsigned blocks = (unsigned) ((hw + 255) / 256 < KBE_FILL_MAX_BLOCKS ? (hw + 255) / 256 : KBE_FILL_MAX_BLOCKS);
        hipLaunchKernelGGL(k_fill_tables, dim3(blocks), dim3(256), 0, s, sc.holes, hole_count, min_holes, sc.depth, W, H, dirs, rect, frame_u8, render_f32,
                           n_tiles, sc.bbox, sc.tiles_x, sc.tiles_y, sc.dist, strips, sc.dist_blocks);
        tables = 1 + min_holes;                                 // k_fill_holes: the frame is done if it has >= tables - 1 holes
    }
    hipLaunchKernelGGL(k_fill_holes, dim3(fill_blocks), dim3(KBE_FILL_BLOCK), 0, s, sc.holes, hole_count, sc.depth, sc.mask, W, H, dirs, rect,
                       frame_u8, render_f32, sc.zkeys, sc.tile_count, n_tiles, sc.bbox, fill_mode,
                       sc.coarse, sc.tiles_x, sc.tiles_y, reset_scatter_scratch, next_hole_count, tables);
}

}  // namespace

extern "C" {

size_t kbe_frame_scratch_bytes(int W, int H)
{
    return (W <= 0 || H <= 0) ? 0 : scratch_bytes(W, H);
}

size_t kbe_video_scratch_stride(int W, int H)
{
    return (W <= 0 || H <= 0) ? 0 : ((scratch_bytes(W, H) + 255) & ~(size_t) 255);
}

// stage = [lanes raw frames][lanes * fin finished frames][ring half 0: batch frames][ring half 1: batch frames][turn counter]
static inline size_t stage_fin_per_lane(int batch) { return batch < -2 ? (size_t) -batch : 2; }
static inline size_t stage_ctl_offset(int W, int H, int lanes, int batch)
{
    const size_t fb = (size_t) W * H * 3;
    return (((size_t) lanes * (1 + stage_fin_per_lane(batch)) + 2 * (size_t) (batch > 0 ? batch : 0)) * fb + 255) & ~(size_t) 255;
}

size_t kbe_video_stage_bytes(int W, int H, int lanes, int batch)
{
    if (W <= 0 || H <= 0 || lanes < 1) return 0;
    return stage_ctl_offset(W, H, lanes, batch) + 256;      // frames + the hand-off's turn counter
}

int kbe_frame_scratch_init(void* scratch, int W, int H, kbe_stream_t stream)
{
    KBE_REQUIRE(scratch && W > 0 && H > 0 && ((uintptr_t) scratch & 15) == 0, "kbe_frame_scratch_init: bad arguments");
    const Scratch sc = carve(scratch, W, H);
    hipLaunchKernelGGL(k_scratch_init, dim3(1024), dim3(256), 0, (hipStream_t) stream, sc.zkeys, sc.zkeys_b, (size_t) W * H, sc.tile_count,
                       sc.tiles_x * sc.tiles_y, sc.hole_count);
    return launched("kbe_frame_scratch_init");
}

int kbe_render_frame_stages(const float* points, const float* image, const float* depth, int N, int W, int H, double focal,
                            double baseline, const float* shift3, void* scratch, uint8_t* frame_u8, float* render_f32,
                            float* existing_f32, float* zee_f32, float* zee_pre_f32, int stages, const int* fill_rect,
                            int raster_w, int raster_n, kbe_stream_t stream)
{
    KBE_REQUIRE(scratch && frame_u8 && N >= 0 && N <= (1 << 30) && W > 0 && H > 0 && (size_t) W * H <= (1u << 30) &&
                W < (1 << 24) && H < (1 << 24) && ((uintptr_t) scratch & 15) == 0, "kbe_render_frame: bad arguments");
    KBE_REQUIRE(N == 0 || (points && image && depth), "kbe_render_frame: cloud pointers are NULL");
    static const FillDirs dirs = make_fill_dirs();
    const hipStream_t s = (hipStream_t) stream;
    const Scratch sc = carve(scratch, W, H);
    const Camera cam = make_camera(W, H, focal, baseline, shift3);
    const int n_tiles = sc.tiles_x * sc.tiles_y;
    int rc = KBE_OK;

    // which z-buffer this frame splats into, and whether its tile launch clears the other one (include/kbe.h)
    const bool alternate = (stages & (KBE_STAGE_ZBUF_A | KBE_STAGE_ZBUF_B)) != 0;
    uint32_t* const zk_use = (stages & KBE_STAGE_ZBUF_B) ? sc.zkeys_b : sc.zkeys;
    uint32_t* const zk_other = (stages & KBE_STAGE_ZBUF_B) ? sc.zkeys : sc.zkeys_b;
    if (stages & KBE_STAGE_PROJECT) {
        ProjectArgs p;
        p.points = points; p.N = N; p.cam = cam; p.zkeys = zk_use; p.tile_count = sc.tile_count; p.buckets = sc.buckets;
        p.tiles_x = sc.tiles_x; p.tiles_y = sc.tiles_y; p.hole_count = sc.hole_count;
        p.raster_w = 0; p.raster_n = 0;
        p.dense = (size_t) N > 2 * (size_t) W * H;
        p.buckets_32bit = (size_t) n_tiles * BUCKET_STRIDE * sizeof(float4) <= ((size_t) 1 << 32);
        if (raster_w > 0 && raster_n >= raster_w && raster_n <= N && raster_n % raster_w == 0) { p.raster_w = raster_w; p.raster_n = raster_n; }
#ifndef KBE_PROJECT_MAX_BLOCKS
#define KBE_PROJECT_MAX_BLOCKS 1000000
#endif
        unsigned blocks = N > 0 ? blocks_for((size_t) N, KBE_PROJECT_BLOCK) + 2 : 1;
        if (blocks > KBE_PROJECT_MAX_BLOCKS) blocks = KBE_PROJECT_MAX_BLOCKS;
        hipLaunchKernelGGL(k_project, dim3(blocks), dim3(KBE_PROJECT_BLOCK), 0, s, p);
        if ((rc = launched("kbe_render_frame/project"))) return rc;
    }
    if (stages & KBE_STAGE_TILES) {
        TileArgs a;
        a.points = points; a.image = image; a.depth_in = depth; a.N = N; a.cam = cam;
        if (N == 0) a.points = a.image = a.depth_in = (const float*) sc.zkeys;     // never dereferenced for a record, but never NULL
        a.zkeys = zk_use; a.tile_count = sc.tile_count; a.buckets = sc.buckets; a.tiles_x = sc.tiles_x; a.tiles_y = sc.tiles_y;
        a.zkeys_clear = alternate ? zk_other : nullptr; a.tile_count_clear = sc.tile_count;
        a.frame = frame_u8; a.depth = sc.depth; a.mask = sc.mask; a.holes = sc.holes; a.hole_count = sc.hole_count; a.bbox = sc.bbox; a.coarse = sc.coarse;
        a.render = render_f32; a.existing = existing_f32; a.zee = zee_f32; a.zee_pre = zee_pre_f32;
        hipLaunchKernelGGL(k_tiles, dim3(n_tiles), dim3(TILE_THREADS), 0, s, a);
        if ((rc = launched("kbe_render_frame/tiles"))) return rc;
    }
    if (stages & KBE_STAGE_FILL) {
        const size_t hw = (size_t) W * H;
        const size_t want_fill = hw / 64, max_fill = (size_t) KBE_FILL_MAX_BLOCKS * 256 / KBE_FILL_BLOCK;       // the same number of threads
        const unsigned fill_blocks = (unsigned) (want_fill < max_fill ? (want_fill > 0 ? want_fill : 1) : max_fill);
        FillRect rect = { 0, 0, W - 1, H - 1 };
        if (fill_rect) { rect.x0 = fill_rect[0]; rect.y0 = fill_rect[1]; rect.x1 = fill_rect[2]; rect.y1 = fill_rect[3]; }
        launch_fill(s, sc, W, H, sc.hole_count, stages, dirs, rect, fill_blocks, frame_u8, render_f32, n_tiles, alternate ? 0 : 1, (int*) nullptr);
        rc = launched("kbe_render_frame/fill");
    }
    return rc;
}

int kbe_render_frame(const float* points, const float* image, const float* depth, int N, int W, int H, double focal,
                     double baseline, const float* shift3, void* scratch, uint8_t* frame_u8, float* render_f32,
                     float* existing_f32, float* zee_f32, float* zee_pre_f32, kbe_stream_t stream)
{
    return kbe_render_frame_stages(points, image, depth, N, W, H, focal, baseline, shift3, scratch, frame_u8, render_f32,
                                   existing_f32, zee_f32, zee_pre_f32, KBE_STAGE_PROJECT | KBE_STAGE_TILES | KBE_STAGE_FILL,
                                   nullptr, 0, 0, stream);
}

int kbe_render_frame_fused(const void* packed, int N, double cloud_focal, int W, int H, double focal, double baseline,
                           const float* shift3, void* scratch, uint8_t* frame_u8, float* render_f32, float* existing_f32,
                           float* zee_f32, float* zee_pre_f32, int stages, const int* fill_rect, int parity, kbe_stream_t stream)
{
    KBE_REQUIRE(packed && scratch && frame_u8 && N >= 0 && N <= (1 << 30) && W > 0 && H > 0 && (size_t) W * H <= (1u << 30) &&
                W < (1 << 24) && H < (1 << 24) && ((uintptr_t) scratch & 15) == 0 && cloud_focal > 0.0 && parity >= -1 && parity <= 1,
                "kbe_render_frame_fused: bad arguments");
    static const FillDirs dirs = make_fill_dirs();
    const hipStream_t s = (hipStream_t) stream;
    const Scratch sc = carve(scratch, W, H);
    const Camera cam = make_camera(W, H, focal, baseline, shift3);
    const int n_tiles = sc.tiles_x * sc.tiles_y;
    int* const count_now = sc.hole_count + (parity == 1 ? 1 : 0);
    int* const count_next = sc.hole_count + (parity == 1 ? 0 : 1);
    int rc = KBE_OK;
    if (parity < 0 && !(stages & KBE_STAGE_KEEP_HOLE_COUNT)) {
        // a frame on its own: the caller keeps no frame parity, so the hole counter is zeroed in front of the launch
        const hipError_t e = hipMemsetAsync(sc.hole_count, 0, 2 * sizeof(int), s);
        if (e != hipSuccess) return fail(KBE_E_LAUNCH, "kbe_render_frame_fused: hipMemsetAsync", e);
    }
    if (stages & KBE_STAGE_TILES) {
        FrameArgs a;
        a.pc = cloud_open(packed, N, cloud_focal);
        a.cam = cam;
        a.tiles_x = sc.tiles_x; a.tiles_y = sc.tiles_y;
        a.frame = frame_u8; a.depth = sc.depth; a.mask = sc.mask; a.holes = sc.holes; a.hole_count = count_now; a.bbox = sc.bbox; a.coarse = sc.coarse;
        a.render = render_f32; a.existing = existing_f32; a.zee = zee_f32; a.zee_pre = zee_pre_f32; a.spill = sc.buckets;
        hipLaunchKernelGGL(k_frame, dim3(n_tiles), dim3(TILE_THREADS), 0, s, a);
        if ((rc = launched("kbe_render_frame_fused/scatter"))) return rc;
    }
    if (stages & KBE_STAGE_FILL) {
        const size_t hw = (size_t) W * H;
        const size_t want_fill = hw / 64, max_fill = (size_t) KBE_FILL_MAX_BLOCKS * 256 / KBE_FILL_BLOCK;
        const unsigned fill_blocks = (unsigned) (want_fill < max_fill ? (want_fill > 0 ? want_fill : 1) : max_fill);
        FillRect rect = { 0, 0, W - 1, H - 1 };
        if (fill_rect) { rect.x0 = fill_rect[0]; rect.y0 = fill_rect[1]; rect.x1 = fill_rect[2]; rect.y1 = fill_rect[3]; }
        launch_fill(s, sc, W, H, count_now, stages, dirs, rect, fill_blocks, frame_u8, render_f32, n_tiles, 0, parity >= 0 ? count_next : (int*) nullptr);
        rc = launched("kbe_render_frame_fused/fill");
    }
    return rc;
}

#if defined(KBE_FRAME_STATS)
extern "C" __attribute__((visibility("default"))) int kbe_debug_fill_hist(unsigned long long* out16, int reset)
{
    hipError_t e = hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_fill_hist), 16 * sizeof(unsigned long long));
    if (e == hipSuccess && reset) { const unsigned long long z[16] = {}; e = hipMemcpyToSymbol(HIP_SYMBOL(g_fill_hist), z, sizeof(z)); }
    return e == hipSuccess ? 0 : -1;
}
extern "C" __attribute__((visibility("default"))) int kbe_debug_fill_stats(unsigned long long* out8, int reset)
{
    hipError_t e = hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_fill_stats), 8 * sizeof(unsigned long long));
    if (e == hipSuccess && reset) { const unsigned long long z[8] = {}; e = hipMemcpyToSymbol(HIP_SYMBOL(g_fill_stats), z, sizeof(z)); }
    return e == hipSuccess ? 0 : -1;
}
extern "C" __attribute__((visibility("default"))) int kbe_debug_frame_stats(unsigned long long* out8, int reset)
{
    hipError_t e = hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_frame_stats), 8 * sizeof(unsigned long long));
    if (e == hipSuccess && reset) { const unsigned long long z[8] = {}; e = hipMemcpyToSymbol(HIP_SYMBOL(g_frame_stats), z, sizeof(z)); }
    return e == hipSuccess ? 0 : -1;
}
#endif

int kbe_render_pointcloud_tiled(const float* points, const float* data, int N, int C, int W, int H, double focal,
                                double baseline, const float* shift3, void* scratch, float* render, float* existing,
                                kbe_stream_t stream)
{
    KBE_REQUIRE(scratch && render && existing && N >= 0 && N <= (1 << 30) && C > 0 && W > 0 && H > 0 && (size_t) W * H <= (1u << 30) &&
                W < (1 << 24) && H < (1 << 24) && ((uintptr_t) scratch & 15) == 0, "kbe_render_pointcloud_tiled: bad arguments");
    KBE_REQUIRE(N == 0 || (points && data), "kbe_render_pointcloud_tiled: cloud pointers are NULL");
    const hipStream_t s = (hipStream_t) stream;
    const Scratch sc = carve(scratch, W, H);
    const Camera cam = make_camera(W, H, focal, baseline, shift3);
    const int n_tiles = sc.tiles_x * sc.tiles_y;
    int rc = kbe_render_frame_stages(points, data, data, N, W, H, focal, baseline, shift3, scratch, (uint8_t*) sc.holes, nullptr, nullptr,
                                     nullptr, nullptr, KBE_STAGE_PROJECT, nullptr, 0, 0, stream);
    if (rc != KBE_OK) return rc;
    TileNcArgs a;
    a.points = points; a.data = data; a.N = N; a.C = C; a.cam = cam;
    if (N == 0) a.points = a.data = (const float*) sc.zkeys;            // never dereferenced for a record, but never NULL
    a.zkeys = sc.zkeys; a.tile_count = sc.tile_count; a.buckets = sc.buckets; a.tiles_x = sc.tiles_x; a.tiles_y = sc.tiles_y;
    a.render = render; a.existing = existing;
    hipLaunchKernelGGL(k_tiles_nc, dim3(n_tiles), dim3(TILE_THREADS), 0, s, a);
    if ((rc = launched("kbe_render_pointcloud_tiled/tiles"))) return rc;
    // leave the scratch clean (z-buffer, bucket counters)
    hipLaunchKernelGGL(k_scratch_init, dim3(1024), dim3(256), 0, s, sc.zkeys, (uint32_t*) nullptr, (size_t) W * H, sc.tile_count, n_tiles, sc.hole_count);
    return launched("kbe_render_pointcloud_tiled/reset");
}

constexpr int KBE_VIDEO_STAGES = KBE_STAGE_PROJECT | KBE_STAGE_TILES | KBE_STAGE_FILL;
int kbe_render_video(const float* points, const float* image, const float* depth, int N, int W, int H, double baseline,
                     int n_frames, const double* focals, const float* shifts, int crop_w, int crop_h, void* scratch,
                     uint8_t* stage, int batch, uint8_t* host_out, int raster_w, int raster_n, const void* packed,
                     double cloud_focal, int flags, kbe_stream_t stream, kbe_stream_t copy_stream, int lanes,
                     const kbe_stream_t* lane_streams)
{
    KBE_REQUIRE(n_frames >= 0 && focals && shifts && stage && host_out && W > 0 && H > 0 && batch >= -64 && (!packed || cloud_focal > 0.0),
                "kbe_render_video: bad arguments");
    KBE_REQUIRE((crop_w == 0 && crop_h == 0) || (crop_w > 0 && crop_h > 0 && crop_w <= W && crop_h <= H), "kbe_render_video: bad crop");
    KBE_REQUIRE(lanes >= 1 && lanes <= KBE_MAX_LANES && (lanes == 1 || lane_streams), "kbe_render_video: bad lanes");
    const hipStream_t cs = (hipStream_t) stream;
    const size_t fb = (size_t) W * H * 3;
    const size_t sb = (scratch_bytes(W, H) + 255) & ~(size_t) 255;      // == kbe_frame_scratch_bytes rounded: lane stride
    const bool crop = crop_w > 0;
    // Frames are independent, so consecutive frames go to `lanes` HIP streams, each with its own scratch and raw
    // frame: the fixed cost of a kernel boundary on this chip (launch ramp, tail, and the L2 write-back between
    // dependent kernels) is then paid while another frame's kernels run.
    // stage = [lanes raw frames][2 * lanes finished frames][ring half 0: batch frames][ring half 1: batch frames].
    hipStream_t ls[KBE_MAX_LANES], ds[1];
    for (int l = 0; l < lanes; l++) ls[l] = l == 0 ? cs : (hipStream_t) lane_streams[l];
    ds[0] = copy_stream ? (hipStream_t) copy_stream : cs;        // only the staged ring (batch > 0) uses it
    const int fin = (int) stage_fin_per_lane(batch);                    // finished-frame buffers per lane
    const int slots = fin * lanes;
    uint8_t* const finished = stage + (size_t) lanes * fb;
    uint8_t* ring[2] = { finished + (size_t) slots * fb, finished + ((size_t) slots + (size_t) (batch > 0 ? batch : 0)) * fb };
    // where do the frames go?  (a pointer the runtime does not know is taken for device memory, as before)
    uint8_t* host_dev = nullptr;                // host_out as the device sees it, when it is pinned host memory
    if (batch <= 0) {
        hipPointerAttribute_t attr;
        if (hipPointerGetAttributes(&attr, host_out) == hipSuccess && attr.type == hipMemoryTypeHost) {
            void* dp = nullptr;
            if (hipHostGetDevicePointer(&dp, host_out, 0) != hipSuccess || !dp)
                return fail(KBE_E_INVALID, "kbe_render_video: host_out is host memory the device cannot address (pin it with hipHostMalloc / hipHostRegister)");
            host_dev = (uint8_t*) dp;
        } else {
            (void) hipGetLastError();           // unknown to the runtime: not an error here
        }
    }
    const bool per_frame = host_dev != nullptr;                         // per-frame hand-off to pinned host memory
    // the hand-off's turn counter sits behind the frame buffers of `stage`, 256-byte aligned
    const size_t ctl_offset = stage_ctl_offset(W, H, lanes, batch);
    KBE_REQUIRE(((uintptr_t) stage & 255) == 0, "kbe_render_video: stage must be 256-byte aligned");
    const bool ringed = batch > 0;
    int rect[4] = { 0, 0, W - 1, H - 1 };
    if (crop) {
        // the pixels cv2.getRectSubPix reads (common.py:256), padded by one: see kbe_render_frame_stages
        const int x0 = (int) floor(W / 2.0 - (crop_w - 1) * 0.5) - 1, y0 = (int) floor(H / 2.0 - (crop_h - 1) * 0.5) - 1;
        rect[0] = x0 > 0 ? x0 : 0; rect[1] = y0 > 0 ? y0 : 0;
        rect[2] = x0 + crop_w + 2 < W - 1 ? x0 + crop_w + 2 : W - 1;
        rect[3] = y0 + crop_h + 2 < H - 1 ? y0 + crop_h + 2 : H - 1;
    }
    // events (created and destroyed per call): `start`, per slot / ring half `rendered` and `copied`, per stream `idle`
    constexpr int MAX_EV = 4 + 4 * KBE_MAX_LANES;
    hipEvent_t pool[MAX_EV];
    int n_ev = 0;
    bool ok = true;
    auto make = [&]() -> hipEvent_t {
        hipEvent_t e = nullptr;
        if (n_ev >= MAX_EV || hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { ok = false; return nullptr; }
        pool[n_ev++] = e;
        return e;
    };
    auto destroy = [&]() { for (int k = 0; k < n_ev; k++) (void) hipEventDestroy(pool[k]); };
    if (per_frame && lanes > 1) {
        // the turn counter of the hand-off starts at 0 for every call (on `stream`, before the other lanes start)
        const hipError_t e = hipMemsetAsync(stage + ctl_offset, 0, sizeof(DeliverCtl), cs);
        if (e != hipSuccess) return fail(KBE_E_LAUNCH, "kbe_render_video: hipMemsetAsync", e);
    }
    if (packed) {
        // every lane starts the call on hole counter 0: both of its counters are zeroed here, on `stream`, before the lanes start
        for (int l = 0; l < lanes; l++) {
            const hipError_t e = hipMemsetAsync(carve((char*) scratch + (size_t) l * sb, W, H).hole_count, 0, 2 * sizeof(int), cs);
            if (e != hipSuccess) return fail(KBE_E_LAUNCH, "kbe_render_video: hipMemsetAsync", e);
        }
    }
    hipEvent_t start = lanes > 1 || (ringed && ds[0] != cs) ? make() : nullptr;
    // the other streams start once everything enqueued on `stream` so far (the cloud) is done
    if (start && ok) {
        (void) hipEventRecord(start, cs);
        for (int l = 1; l < lanes; l++) (void) hipStreamWaitEvent(ls[l], start, 0);
        if (ringed && ds[0] != cs) (void) hipStreamWaitEvent(ds[0], start, 0);
    }
    int lane_frames[KBE_MAX_LANES] = {};
    auto render = [&](int i, int l, uint8_t* out) {
        uint8_t* raw = stage + (size_t) l * fb;
        const int fill_flags = lanes >= KBE_FILL_BY_COUNT_MIN_LANES ? (KBE_STAGE_FILL_BY_COUNT | ((flags & KBE_VIDEO_FILL_DIST) ? KBE_STAGE_FILL_DIST : 0)) : 0;
        int rc;
        if (packed)         // the fused scatter on the packed cloud; a lane's frames alternate between its two hole counters
            rc = kbe_render_frame_fused(packed, N, cloud_focal, W, H, focals[i], baseline, shifts + 3 * (size_t) i,
                                        (char*) scratch + (size_t) l * sb, crop ? raw : out, nullptr, nullptr, nullptr, nullptr,
                                        KBE_STAGE_TILES | KBE_STAGE_FILL | fill_flags, crop ? rect : nullptr, lane_frames[l]++ & 1,
                                        (kbe_stream_t) ls[l]);
        else {
            // a lane's frames alternate between the two z-buffers (A, B, A, ...), each clearing the other's in its tile
            // launch; a lane's LAST frame, if it falls on A, takes the stand-alone form (A cleared by its fill launch), so
            // that every call leaves A empty -- B is always cleared before it is used
            const int k = lane_frames[l]++;
            const bool last_of_lane = i + lanes >= n_frames;
            const int zflags = (k & 1) ? KBE_STAGE_ZBUF_B : (last_of_lane ? 0 : KBE_STAGE_ZBUF_A);
            rc = kbe_render_frame_stages(points, image, depth, N, W, H, focals[i], baseline, shifts + 3 * (size_t) i,
                                         (char*) scratch + (size_t) l * sb, crop ? raw : out, nullptr, nullptr, nullptr, nullptr,
                                         KBE_VIDEO_STAGES | fill_flags | zflags, crop ? rect : nullptr, raster_w, raster_n, (kbe_stream_t) ls[l]);
        }
        if (rc == KBE_OK && crop) rc = kbe_crop_resize_u8(raw, W, H, crop_w, crop_h, out, (kbe_stream_t) ls[l]);
        return rc;
    };
    // whoever synchronises `stream` afterwards also sees every frame delivered and every other stream idle
    auto join = [&]() {
        for (int l = 1; l < lanes && ok; l++) { hipEvent_t e = make(); if (e) { (void) hipEventRecord(e, ls[l]); (void) hipStreamWaitEvent(cs, e, 0); } }
        if (ringed && ds[0] != cs && ok) { hipEvent_t e = make(); if (e) { (void) hipEventRecord(e, ds[0]); (void) hipStreamWaitEvent(cs, e, 0); } }
    };
    int rc = KBE_OK;
    if (!ringed && !per_frame) {
        // host_out is device memory: the last kernel of every frame stores straight into it (the frames stay in HBM)
        for (int i = 0; i < n_frames && rc == KBE_OK; i++) rc = render(i, i % lanes, host_out + (size_t) i * fb);
    } else if (!ringed) {
        // Hand-off to pinned host memory in the lane's own stream (no event), the lanes taking turns:
        //   batch == 0   per frame, k_deliver (a lane alternates between two finished-frame slots);
        //   batch < 0    per group of G = -batch consecutive frames, rendered by ONE lane into its G slots and sent with one
        //                runtime transfer (hipMemcpyAsync) between a gate kernel that waits for the turn and one that
        //                passes it on.
        DeliverCtl* ctl = lanes > 1 ? (DeliverCtl*) (stage + ctl_offset) : nullptr;
        if (batch == 0) {
            for (int i = 0; i < n_frames && rc == KBE_OK; i++) {
                const int l = i % lanes, slot = i % slots;
                uint8_t* out = finished + (size_t) slot * fb;
                rc = render(i, l, out);
                if (rc != KBE_OK) break;
                hipLaunchKernelGGL(k_deliver, dim3(DELIVER_BLOCKS), dim3(DELIVER_THREADS), 0, ls[l], out, host_dev + (size_t) i * fb, fb,
                                   ctl, (uint32_t) i);
                rc = launched("kbe_render_video/deliver");
            }
        } else {
            const int G = -batch;
            for (int i0 = 0, g = 0; i0 < n_frames && rc == KBE_OK; i0 += G, g++) {
                const int l = g % lanes, nb = n_frames - i0 < G ? n_frames - i0 : G;
                uint8_t* base = finished + (size_t) l * fin * fb;
                for (int k = 0; k < nb && rc == KBE_OK; k++) rc = render(i0 + k, l, base + (size_t) k * fb);
                if (rc != KBE_OK) break;
                if (ctl) hipLaunchKernelGGL(k_turn, dim3(1), dim3(64), 0, ls[l], ctl, (uint32_t) g, 0);
                const hipError_t e = hipMemcpyAsync(host_out + (size_t) i0 * fb, base, (size_t) nb * fb, hipMemcpyDeviceToHost, ls[l]);
                if (e != hipSuccess) { rc = fail(KBE_E_LAUNCH, "kbe_render_video: hipMemcpyAsync", e); break; }
                if (ctl) hipLaunchKernelGGL(k_turn, dim3(1), dim3(64), 0, ls[l], ctl, (uint32_t) g, 1);
                rc = launched("kbe_render_video/turn");
            }
        }
    } else {
        // staged ring: a half is copied to the host with ONE runtime transfer while the other half is being rendered;
        // cross-stream events are per batch, not per frame
        const hipStream_t dc = ds[0];
        hipEvent_t rendered[2][KBE_MAX_LANES] = {}, copied[2] = { nullptr, nullptr };
        for (int h = 0; h < 2; h++) {
            copied[h] = make();
            for (int l = 0; l < lanes; l++) if (ls[l] != dc) rendered[h][l] = make();
        }
        int n_batches = 0;
        for (int i0 = 0; i0 < n_frames && rc == KBE_OK && ok; i0 += batch, n_batches++) {
            const int half = n_batches & 1;
            const int nb = n_frames - i0 < batch ? n_frames - i0 : batch;
            if (n_batches >= 2)
                for (int l = 0; l < lanes; l++) if (ls[l] != dc) (void) hipStreamWaitEvent(ls[l], copied[half], 0);    // the half is free again
            for (int k = 0; k < nb && rc == KBE_OK; k++) rc = render(i0 + k, (i0 + k) % lanes, ring[half] + (size_t) k * fb);
            if (rc != KBE_OK) break;
            for (int l = 0; l < lanes; l++) {
                if (ls[l] == dc) continue;                                  // same stream as the transfer: ordered anyway
                (void) hipEventRecord(rendered[half][l], ls[l]);
                (void) hipStreamWaitEvent(dc, rendered[half][l], 0);
            }
            const hipError_t e = hipMemcpyAsync(host_out + (size_t) i0 * fb, ring[half], (size_t) nb * fb, hipMemcpyDeviceToHost, dc);
            if (e != hipSuccess) { rc = fail(KBE_E_LAUNCH, "kbe_render_video: hipMemcpyAsync", e); break; }
            (void) hipEventRecord(copied[half], dc);
        }
    }
    join();
    destroy();
    if (rc == KBE_OK && !ok) rc = fail(KBE_E_LAUNCH, "kbe_render_video: hipEventCreate");
    return rc;
}

}  // extern "C"
