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
//   (kbe_holes.hip)  the hole list (:838-924) filled with an exact branch-and-bound over the 16 directions;
//   (kbe_fused.hip)  the other scatter route: k_frame, one launch on the packed cloud, z-tile in LDS;
//   k_tiles_nc the same tile machinery for render_pointcloud with any channel count (4 channels at a time);
//   kbe_render_video  the whole loop enqueued from C, consecutive frames on several streams ("lanes"), finished
//              frames handed to pinned host memory (k_turn, k_deliver).
// The tile machinery in LDS (record lists, degrid, z-tested gather, epilogue) that k_tiles, k_tiles_nc and k_frame
// share, the tile geometry and the per-view scratch are in kbe_tiles.h.
// No accumulator or float render ever exists in HBM and no floating-point atomic is executed.
// What bounds these kernels is instruction issue, not bandwidth: the code below is
// written branch-free where lanes mostly agree and with wave-uniform work kept on the scalar unit.
//
// Numerics are those of oracle/kbe_oracle.c: the z-buffer is bit-exact (min commutes), degrid
// is the out-of-place schedule, accumulation order is bucket order (not point order).
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <mutex>
#include <thread>
#include <vector>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>

#include "kbe_cloud.h"
#include "kbe_tiles.h"

using namespace kbe;

namespace {

__global__ void k_scratch_init(uint32_t* zkeys, uint32_t* zkeys_b, size_t hw, int* tile_count, int n_tiles, int* hole_count)
{
    const size_t stride = (size_t) gridDim.x * blockDim.x, gtid = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    for (size_t i = gtid; i < hw; i += stride) { zkeys[i] = KBE_ZKEY_EMPTY; if (zkeys_b) zkeys_b[i] = KBE_ZKEY_EMPTY; }
    for (size_t i = gtid; i < (size_t) n_tiles; i += stride) { tile_count[i * CNT_STRIDE] = 0; tile_count[i * CNT_STRIDE + 1] = 0; }      // (+ 1: the arrivals at a shared list, kbe_fused.hip)
    if (gtid < (size_t) HOLE_COUNT_INTS) hole_count[gtid] = 0;
}

// k_scratch_init for the sets `stride` bytes apart, one launch (blockIdx.y = the set; the offsets are those of carve())
__global__ void k_scratch_init_sets(char* base, size_t stride, size_t zkeys, size_t zkeys_b, size_t hw, size_t tile_count, int n_tiles, size_t hole_count)
{
    char* const set = base + (size_t) blockIdx.y * stride;
    uint32_t* const za = (uint32_t*) (set + zkeys), * const zb = (uint32_t*) (set + zkeys_b);
    const size_t step = (size_t) gridDim.x * blockDim.x, gtid = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    for (size_t i = gtid; i < hw; i += step) { za[i] = KBE_ZKEY_EMPTY; zb[i] = KBE_ZKEY_EMPTY; }
    for (size_t i = gtid; i < (size_t) n_tiles; i += step) { ((int*) (set + tile_count))[i * CNT_STRIDE] = 0; ((int*) (set + tile_count))[i * CNT_STRIDE + 1] = 0; }
    if (gtid < (size_t) HOLE_COUNT_INTS) ((int*) (set + hole_count))[gtid] = 0;
}

// the hole counters / list totals (HOLE_COUNT_INTS ints) of `n` scratch sets `stride` bytes apart
// ... and both banks of their per-tile list counters (`tile_ints` ints at `tile_first` of the first set; blockIdx.y = the set): a
// video that ended in an error after a launch that had placed ahead leaves the counters of one bank standing
__global__ void k_zero_counters(int* first, size_t stride, int n, int* tile_first, size_t tile_ints)
{
    static_assert(HOLE_COUNT_INTS == 8, "k >> 3, k & 7");
    if (blockIdx.x == 0 && blockIdx.y == 0)
        for (int k = threadIdx.x; k < HOLE_COUNT_INTS * n; k += blockDim.x) ((int*) ((char*) first + (size_t) (k >> 3) * stride))[k & 7] = 0;
    int* const t = (int*) ((char*) tile_first + (size_t) blockIdx.y * stride);
    for (size_t k = (size_t) blockIdx.x * blockDim.x + threadIdx.x; k < tile_ints; k += (size_t) gridDim.x * blockDim.x) t[k] = 0;
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
__device__ __forceinline__ void project_body(const ProjectArgs& a)
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

__device__ __forceinline__ void tiles_body(const TileArgs& a)
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
    lds_reset_heads(L, tid);
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
    const bool fast = lds_tile_is_fast(L);
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
                lds_reset_heads(L, tid);
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
                lds_reset_heads(L, tid);
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

// Both launches of the bucket route take up to KBE_SCATTER_JOBS frames (blockIdx.y = the frame; same cloud, same frame
// size, each frame with its own camera and scratch set).  Frames are independent, and a launch on its own is bound by
// its ramp and its latencies, not by the chip: k_project issues for 7 us of its 14, k_tiles for 8 of its 19.5.  Several
// frames per launch fill those gaps inside ONE stream (the video loop's lanes do the same across streams, but HIP maps
// a process's streams onto four hardware queues).
constexpr int KBE_SCATTER_JOBS = KBE_FILL_JOBS;
struct ProjectJobs { ProjectArgs a[KBE_SCATTER_JOBS]; };
struct TileJobs { TileArgs a[KBE_SCATTER_JOBS]; };

__global__ void __launch_bounds__(KBE_PROJECT_BLOCK) k_project(ProjectArgs a)
{
    project_body(a);
}

__global__ void __launch_bounds__(TILE_THREADS) __attribute__((KBE_TILE_ATTR)) k_tiles(TileArgs a)
{
    tiles_body(a);
}

// ... and the forms that take a group of frames (a frame on its own keeps the launches above: their arguments sit in the
// kernel-argument registers, while a group's are indexed by blockIdx.y and loaded by every wave -- 33.5 vs 34.8 us per frame
// for the scatter of a single frame)
__global__ void __launch_bounds__(KBE_PROJECT_BLOCK) k_project_group(ProjectJobs jobs)
{
    project_body(jobs.a[blockIdx.y]);
}

__global__ void __launch_bounds__(TILE_THREADS) __attribute__((KBE_TILE_ATTR)) k_tiles_group(TileJobs jobs)
{
    tiles_body(jobs.a[blockIdx.y]);
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
    const bool fast = lds_tile_is_fast(L);
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
                lds_reset_heads(L, tid);
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
            lds_reset_heads(L, tid);
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
                    lds_reset_heads(L, tid);
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

// the turn of a runtime transfer.  pass == 0: the group draws a ticket NOW, when its frames are ready -- first ready, first
// served: with turns in the order of the groups a lane whose group is ready waited for lanes still rendering earlier groups
// (dolly frames differ 5 x in cost along a video: passes of 109 and 145 ms alternated) -- and waits (bounded) until the
// transfers in front of it have finished; pass == 1 hands the turn on.  pad[0]: tickets drawn so far.
__global__ void __launch_bounds__(64) k_turn(DeliverCtl* ctl, int pass, int max_polls)
{
    if (threadIdx.x != 0) return;
    if (pass) { atomicAdd(&ctl->serving, 1u); return; }
    const uint32_t ticket = atomicAdd(&ctl->pad[0], 1u);
    for (int polls = 0; polls < max_polls; polls++) {
        if (__hip_atomic_load(&ctl->serving, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= ticket) break;
        __builtin_amdgcn_s_sleep(32);
    }
}

// ---------------------------------------------------------------------------------------
// The hand-off by an SDMA ENGINE (KBE_VIDEO_SDMA; round 5).  hipMemcpyAsync device-to-host is a blit KERNEL on this runtime:
// it takes wave slots next to the rendering and sits in the lane's stream, so the lane's next group cannot start before its
// last one has left.  HSA drives the DMA engines directly (hsa_amd_memory_async_copy_on_engine), and a copy can be ordered
// against HIP streams from the GPU side alone (tools/sdma_probe.hip): the host enqueues the copy of a group at once with a
// DEPENDENCY signal the engine polls, a one-lane kernel behind the group's last launch stores 0 into that signal's value, and
// whoever needs the copy done -- the lane before it renders into the same slots again, `stream` at the end of the call -- runs
// a one-lane kernel that polls the copy's COMPLETION signal.  The engine takes the copies in the order they were enqueued:
// the lanes need no turns.  The signals live in a process-wide pool (the one piece of state the library keeps: they must
// outlive the call, which returns before the copies run) and are reused once the call that used them has run to its end.
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) k_signal_release(volatile int64_t* value)
{
    if (threadIdx.x != 0) return;
    __threadfence_system();
    __hip_atomic_store((int64_t*) value, (int64_t) 0, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// bounded (SDMA_WAIT_SECONDS of the device's wall clock -- the rate is asked of the runtime, sdma_wait_ticks -- where a video's groups take
// milliseconds): an engine that never reports is a dead or a stalled device, and a kernel that polls for ever would hide that.  Giving up
// is an ERROR the caller must see -- the frames of that group are not in its memory -- but not one to kill the process for (a trap raises
// an HSA queue exception and the runtime aborts: ADVICE r5): the kernel stores 1 into the pool's host-visible error word and returns.
// kbe_video_handoff_status() reports it (sticky) and waits on the host for the copies still under way; calls after it keep off the engine.
__global__ void __launch_bounds__(64) k_signal_wait(volatile int64_t* value, unsigned long long max_ticks, int* gave_up)
{
    if (threadIdx.x != 0) return;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    do {
        if (__hip_atomic_load((int64_t*) value, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) <= 0) return;
        __builtin_amdgcn_s_sleep(16);
    } while (__builtin_amdgcn_s_memrealtime() - t0 < max_ticks);
    __hip_atomic_store(gave_up, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);      // (1 = not yet reported: kbe_video_handoff_status)
}
#ifndef KBE_SDMA_WAIT_SECONDS
#define KBE_SDMA_WAIT_SECONDS 4.0
#endif
#ifndef KBE_SDMA_TWO_ENGINES
#define KBE_SDMA_TWO_ENGINES 0      // (1: consecutive groups alternate between two engines -- measured slower end to end, see sdma_open)
#endif

struct SdmaPair {                       // the signals of one copy and where their values live
    hsa_signal_t dep, fin; volatile int64_t* dep_value; volatile int64_t* fin_value;
    bool released;                      // this use: the kernel that releases `dep` is known to be in a stream (else: the error path releases it from the host)
};
struct SdmaGeneration { hipEvent_t done; std::vector<SdmaPair> pairs; };
struct SdmaRoute { hsa_agent_t gpu, cpu; uint32_t engines; };          // the engines that copy from `gpu` to `cpu`'s memory (the preferred ones, or all)
struct SdmaPool {
    std::mutex mu;
    int state = 0;                              // 0 = not tried, 1 = HSA is up, -1 = it is not (or an engine stopped answering: no more copies through it)
    std::vector<SdmaPair> idle;
    std::vector<SdmaGeneration> running;
    std::vector<SdmaRoute> routes;
    int* gave_up = nullptr;                     // pinned host word, device-visible: a k_signal_wait that gave up stores 1 (sticky)
    unsigned long long wait_ticks = 0;          // KBE_SDMA_WAIT_SECONDS of the device's wall clock
};
static SdmaPool& sdma_pool() { static SdmaPool* p = new SdmaPool; return *p; }         // (never destroyed: the signals must not die before the runtime)
// A signal whose value device code may store to and poll: HSA asks for one that only GPUs consume (no host interrupt behind it -- nothing
// on the host ever waits for these) and hands out the value's address (hsa_amd_signal_value_pointer)
static bool sdma_signal(hsa_signal_t& sg, volatile int64_t*& value)
{
    if (hsa_amd_signal_create(1, 0, nullptr, HSA_AMD_SIGNAL_AMD_GPU_ONLY, &sg) != HSA_STATUS_SUCCESS) return false;
    volatile hsa_signal_value_t* p = nullptr;
    if (hsa_amd_signal_value_pointer(sg, &p) != HSA_STATUS_SUCCESS || !p) { (void) hsa_signal_destroy(sg); return false; }
    static_assert(sizeof(hsa_signal_value_t) == sizeof(int64_t), "a signal's value is 64 bits");
    value = (volatile int64_t*) p;
    return true;
}
struct SdmaCall {                               // one kbe_render_video call's use of the engine
    bool ok = false;
    hsa_agent_t gpu = {}, cpu = {};
    hsa_amd_sdma_engine_id_t engine[2] = { HSA_AMD_SDMA_ENGINE_0, HSA_AMD_SDMA_ENGINE_0 };      // consecutive groups alternate between two engines
    int sent = 0;
    std::vector<SdmaPair> used;
};
// which agents own the two buffers, and an engine that copies from the one to the other; false: no SDMA hand-off (the caller
// falls back to hipMemcpyAsync)
static bool sdma_open(SdmaCall& c, const void* device_buffer, const void* host_buffer)
{
    SdmaPool& pool = sdma_pool();
    std::lock_guard<std::mutex> lock(pool.mu);
    if (pool.state == 0) {
        pool.state = hsa_init() == HSA_STATUS_SUCCESS ? 1 : -1;       // (reference-counted: HIP's own runtime holds it up already)
        if (pool.state > 0) {
            // the word a polling kernel that gives up writes, and how many ticks of the wall clock it polls for (s_memrealtime counts at the
            // rate the runtime reports: 100 MHz on gfx950)
            int dev = 0, khz = 0;
            void* w = nullptr;
            if (hipHostMalloc(&w, 64, hipHostMallocMapped) != hipSuccess) { (void) hipGetLastError(); pool.state = -1; }
            else {
                pool.gave_up = (int*) w; *pool.gave_up = 0;
                if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess || khz <= 0) { (void) hipGetLastError(); khz = 100000; }
                pool.wait_ticks = (unsigned long long) (KBE_SDMA_WAIT_SECONDS * 1000.0 * (double) khz);
            }
        }
    }
    if (pool.state < 0) return false;
    if (__atomic_load_n(pool.gave_up, __ATOMIC_ACQUIRE) != 0) return false;      // an engine stopped answering: the runtime's transfers from here on
    hsa_amd_pointer_info_t dinfo = {}, hinfo = {};
    dinfo.size = hinfo.size = sizeof(hsa_amd_pointer_info_t);
    if (hsa_amd_pointer_info(device_buffer, &dinfo, nullptr, nullptr, nullptr) != HSA_STATUS_SUCCESS || dinfo.type == HSA_EXT_POINTER_TYPE_UNKNOWN) return false;
    if (hsa_amd_pointer_info(host_buffer, &hinfo, nullptr, nullptr, nullptr) != HSA_STATUS_SUCCESS || hinfo.type == HSA_EXT_POINTER_TYPE_UNKNOWN) return false;
    hsa_device_type_t dt, ht;
    if (hsa_agent_get_info(dinfo.agentOwner, HSA_AGENT_INFO_DEVICE, &dt) != HSA_STATUS_SUCCESS || dt != HSA_DEVICE_TYPE_GPU) return false;
    if (hsa_agent_get_info(hinfo.agentOwner, HSA_AGENT_INFO_DEVICE, &ht) != HSA_STATUS_SUCCESS || ht != HSA_DEVICE_TYPE_CPU) return false;
    // which engines copy from this GPU to this CPU agent is a property of the machine: asked once per pair of agents (the two queries cost
    // the host ~10 us of every call, between the launches of a video's first and second group)
    uint32_t pick = 0;
    for (const SdmaRoute& r : pool.routes)
        if (r.gpu.handle == dinfo.agentOwner.handle && r.cpu.handle == hinfo.agentOwner.handle) pick = r.engines;
    if (!pick) {
        uint32_t avail = 0, preferred = 0;
        if (hsa_amd_memory_copy_engine_status(hinfo.agentOwner, dinfo.agentOwner, &avail) != HSA_STATUS_SUCCESS || !avail) return false;
        if (hsa_amd_memory_get_preferred_copy_engine(hinfo.agentOwner, dinfo.agentOwner, &preferred) != HSA_STATUS_SUCCESS) preferred = 0;
        pick = (preferred & avail) ? (preferred & avail) : avail;
        pool.routes.push_back(SdmaRoute{ dinfo.agentOwner, hinfo.agentOwner, pick });
    }
    // its lowest engine.  (An engine takes ~9.4 us from the end of one copy to the start of the next even when that one has long been
    // released -- the device-side timeline of a 20-frame video, tools/sdma_timeline.py: five such gaps in 1.35 ms.  With the groups
    // alternating between TWO engines, KBE_SDMA_TWO_ENGINES, the copies overlap and the last one ends 60 us earlier on the device --
    // and the video is delivered SLOWER: 13.9 instead of 14.6 k frames/s for 20 frames, 16.15 instead of 16.6 k for 75: the host sees
    // the end ~170 us after the device instead of ~30.  One engine it is.)
    const uint32_t first = pick & (~pick + 1u), rest = pick & ~first, second = rest ? (rest & (~rest + 1u)) : first;
    c.engine[0] = (hsa_amd_sdma_engine_id_t) first;
    c.engine[1] = (hsa_amd_sdma_engine_id_t) (KBE_SDMA_TWO_ENGINES ? second : first);
    c.gpu = dinfo.agentOwner; c.cpu = hinfo.agentOwner;
    // the signals of calls that have run to their end are idle again
    for (size_t g = 0; g < pool.running.size(); ) {
        if (hipEventQuery(pool.running[g].done) == hipSuccess) {
            (void) hipEventDestroy(pool.running[g].done);
            pool.idle.insert(pool.idle.end(), pool.running[g].pairs.begin(), pool.running[g].pairs.end());
            pool.running[g] = std::move(pool.running.back());
            pool.running.pop_back();
        } else {
            (void) hipGetLastError();       // hipErrorNotReady is not an error here
            g++;
        }
    }
    return c.ok = true;
}
static bool sdma_pair(SdmaCall& c, SdmaPair& p)
{
    SdmaPool& pool = sdma_pool();
    std::lock_guard<std::mutex> lock(pool.mu);
    if (!pool.idle.empty()) { p = pool.idle.back(); pool.idle.pop_back(); }
    else {
        if (!sdma_signal(p.dep, p.dep_value)) return false;
        if (!sdma_signal(p.fin, p.fin_value)) { (void) hsa_signal_destroy(p.dep); return false; }
    }
    hsa_signal_store_relaxed(p.dep, 1);
    hsa_signal_store_relaxed(p.fin, 1);
    p.released = false;
    c.used.push_back(p);
    return true;
}
// The call ends in an ERROR with copies on the engine (a launch failed behind them, KBE_VIDEO_INJECT_FAULT): none of them may outlive
// the call -- the caller, told of the error, may free its buffer the moment we return -- and none may keep waiting for a release
// that never comes (the engine's queue would be wedged for the process).  Every copy whose release kernel is not known to be in a
// stream is released from the host (it moves whatever its slots hold: the frames of a failed call are not valid anyway); the lanes
// run dry; the host polls every completion signal; a pair whose copy has completed is idle again, one whose copy has not after
// seconds is never reused.
static void sdma_abort(SdmaCall& c, const hipStream_t* lanes_streams, int lanes)
{
    if (c.used.empty()) return;
    for (SdmaPair& p : c.used) if (!p.released) hsa_signal_store_screlease(p.dep, 0);
    for (int l = 0; l < lanes; l++) { (void) hipStreamSynchronize(lanes_streams[l]); (void) hipGetLastError(); }
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<SdmaPair> done;
    for (SdmaPair& p : c.used) {
        bool fin = false;
        while (!(fin = hsa_signal_load_scacquire(p.fin) <= 0) && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < KBE_SDMA_WAIT_SECONDS)
            std::this_thread::sleep_for(std::chrono::microseconds(20));
        if (fin) done.push_back(p);
    }
    SdmaPool& pool = sdma_pool();
    std::lock_guard<std::mutex> lock(pool.mu);
    if (done.size() != c.used.size()) { __atomic_store_n(pool.gave_up, 1, __ATOMIC_RELEASE); pool.state = -1; }      // an engine that does not answer
    pool.idle.insert(pool.idle.end(), done.begin(), done.end());
    c.used.clear();
    c.ok = false;
}
// the call is enqueued: its signals are idle again once `stream` has run past this point
static void sdma_close(SdmaCall& c, hipStream_t stream)
{
    if (c.used.empty()) return;
    SdmaPool& pool = sdma_pool();
    std::lock_guard<std::mutex> lock(pool.mu);
    SdmaGeneration g;
    g.pairs.swap(c.used);
    if (hipEventCreateWithFlags(&g.done, hipEventDisableTiming) == hipSuccess && hipEventRecord(g.done, stream) == hipSuccess) pool.running.push_back(std::move(g));
    // (else: the pairs are dropped -- a leak of a few signals, never a reuse that is too early)
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

}  // namespace

extern "C" {

size_t kbe_frame_scratch_bytes(int W, int H, int N)
{
    return (W <= 0 || H <= 0 || N < 0) ? 0 : scratch_set_bytes(W, H, N);
}

size_t kbe_video_scratch_stride(int W, int H, int N)
{
    return (W <= 0 || H <= 0 || N < 0) ? 0 : ((scratch_set_bytes(W, H, N) + 255) & ~(size_t) 255);
}

// stage = [lanes raw frames][lanes * fin finished frames][ring half 0: batch frames][ring half 1: batch frames][turn counter]
static inline size_t stage_fin_per_lane(int batch) { return batch < -2 ? (size_t) -batch : 2; }
static inline size_t stage_ctl_offset(int W, int H, int lanes, int batch)
{
    const size_t fb = (size_t) W * H * 3;
    return (((size_t) lanes * (KBE_FRAME_JOBS + stage_fin_per_lane(batch)) + 2 * (size_t) (batch > 0 ? batch : 0)) * fb + 255) & ~(size_t) 255;
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
    // (both banks of counters: 4 * n_tiles * CNT_STRIDE bytes is a multiple of 16, so the banks are contiguous)
    hipLaunchKernelGGL(k_scratch_init, dim3(1024), dim3(256), 0, (hipStream_t) stream, sc.zkeys, sc.zkeys_b, (size_t) W * H, sc.tile_count,
                       2 * sc.tiles_x * sc.tiles_y, sc.hole_count);
    return launched("kbe_frame_scratch_init");
}

int kbe_frame_scratch_init_sets(void* scratch, size_t stride, int n, int W, int H, kbe_stream_t stream)
{
    KBE_REQUIRE(scratch && n > 0 && n <= 65535 && W > 0 && H > 0 && ((uintptr_t) scratch & 15) == 0 && (stride & 15) == 0 && stride >= scratch_bytes(W, H),
                "kbe_frame_scratch_init_sets: bad arguments");
    const Scratch sc = carve(scratch, W, H);
    auto off = [&](const void* p) { return (size_t) ((const char*) p - (const char*) scratch); };
    hipLaunchKernelGGL(k_scratch_init_sets, dim3(256, n), dim3(256), 0, (hipStream_t) stream, (char*) scratch, stride, off(sc.zkeys), off(sc.zkeys_b), (size_t) W * H,
                       off(sc.tile_count), 2 * sc.tiles_x * sc.tiles_y, off(sc.hole_count));
    return launched("kbe_frame_scratch_init_sets");
}

}  // extern "C"

namespace {
// one frame of a group: its camera, its scratch set, where it goes, which of the set's z-buffers it uses (KBE_STAGE_ZBUF_*)
struct FrameJob {
    double focal;
    const float* shift3;
    void* scratch;
    uint8_t* frame_u8;
    float* render_f32; float* existing_f32; float* zee_f32; float* zee_pre_f32;
    int zflags;
};

// the launches of `n` frames of the same cloud and size, each launch taking all n frames (bucket route)
int render_jobs(const float* points, const float* image, const float* depth, int N, int W, int H, double baseline, int n, const FrameJob* jobs,
                int stages, const int* fill_rect, int raster_w, int raster_n, hipStream_t s)
{
    static const FillDirs dirs = make_fill_dirs();
    ProjectJobs pj;
    TileJobs tj;
    FillTarget targets[KBE_SCATTER_JOBS];
    int n_tiles = 0, rc = KBE_OK;
    for (int k = 0; k < KBE_SCATTER_JOBS; k++) {
        const FrameJob& job = jobs[k < n ? k : 0];
        const Scratch sc = carve(job.scratch, W, H);
        const Camera cam = make_camera(W, H, job.focal, baseline, job.shift3);
        n_tiles = sc.tiles_x * sc.tiles_y;
        // which z-buffer this frame splats into, and whether its tile launch clears the other one (include/kbe.h)
        const bool alternate = (job.zflags & (KBE_STAGE_ZBUF_A | KBE_STAGE_ZBUF_B)) != 0;
        uint32_t* const zk_use = (job.zflags & KBE_STAGE_ZBUF_B) ? sc.zkeys_b : sc.zkeys;
        uint32_t* const zk_other = (job.zflags & KBE_STAGE_ZBUF_B) ? sc.zkeys : sc.zkeys_b;
        ProjectArgs& p = pj.a[k];
        p.points = points; p.N = N; p.cam = cam; p.zkeys = zk_use; p.tile_count = sc.tile_count; p.buckets = sc.buckets;
        p.tiles_x = sc.tiles_x; p.tiles_y = sc.tiles_y; p.hole_count = sc.hole_count;
        p.raster_w = 0; p.raster_n = 0;
        p.dense = (size_t) N > 2 * (size_t) W * H;
        p.buckets_32bit = (size_t) n_tiles * BUCKET_STRIDE * sizeof(float4) <= ((size_t) 1 << 32);
        if (raster_w > 0 && raster_n >= raster_w && raster_n <= N && raster_n % raster_w == 0) { p.raster_w = raster_w; p.raster_n = raster_n; }
        TileArgs& a = tj.a[k];
        a.points = points; a.image = image; a.depth_in = depth; a.N = N; a.cam = cam;
        if (N == 0) a.points = a.image = a.depth_in = (const float*) sc.zkeys;     // never dereferenced for a record, but never NULL
        a.zkeys = zk_use; a.tile_count = sc.tile_count; a.buckets = sc.buckets; a.tiles_x = sc.tiles_x; a.tiles_y = sc.tiles_y;
        a.zkeys_clear = alternate ? zk_other : nullptr; a.tile_count_clear = sc.tile_count;
        a.frame = job.frame_u8; a.depth = sc.depth; a.mask = sc.mask; a.holes = sc.holes; a.hole_count = sc.hole_count; a.bbox = sc.bbox; a.coarse = sc.coarse;
        a.render = job.render_f32; a.existing = job.existing_f32; a.zee = job.zee_f32; a.zee_pre = job.zee_pre_f32;
        targets[k] = FillTarget{ sc, sc.hole_count, job.frame_u8, job.render_f32, alternate ? 0 : 1, nullptr };
    }
    if (stages & KBE_STAGE_PROJECT) {
#ifndef KBE_PROJECT_MAX_BLOCKS
#define KBE_PROJECT_MAX_BLOCKS 1000000
#endif
        unsigned blocks = N > 0 ? blocks_for((size_t) N, KBE_PROJECT_BLOCK) + 2 : 1;
        if (blocks > KBE_PROJECT_MAX_BLOCKS) blocks = KBE_PROJECT_MAX_BLOCKS;
        if (n == 1) hipLaunchKernelGGL(k_project, dim3(blocks), dim3(KBE_PROJECT_BLOCK), 0, s, pj.a[0]);
        else hipLaunchKernelGGL(k_project_group, dim3(blocks, n), dim3(KBE_PROJECT_BLOCK), 0, s, pj);
        if ((rc = launched("kbe_render_frame/project"))) return rc;
    }
    if (stages & KBE_STAGE_TILES) {
        if (n == 1) hipLaunchKernelGGL(k_tiles, dim3(n_tiles), dim3(TILE_THREADS), 0, s, tj.a[0]);
        else hipLaunchKernelGGL(k_tiles_group, dim3(n_tiles, n), dim3(TILE_THREADS), 0, s, tj);
        if ((rc = launched("kbe_render_frame/tiles"))) return rc;
    }
    if (stages & KBE_STAGE_FILL) {
        FillRect rect = { 0, 0, W - 1, H - 1 };
        if (fill_rect) { rect.x0 = fill_rect[0]; rect.y0 = fill_rect[1]; rect.x1 = fill_rect[2]; rect.y1 = fill_rect[3]; }
        launch_fill(s, n, targets, W, H, stages, dirs, rect, n_tiles);
        rc = launched("kbe_render_frame/fill");
    }
    return rc;
}
}  // namespace

extern "C" {

int kbe_render_frame_stages(const float* points, const float* image, const float* depth, int N, int W, int H, double focal,
                            double baseline, const float* shift3, void* scratch, uint8_t* frame_u8, float* render_f32,
                            float* existing_f32, float* zee_f32, float* zee_pre_f32, int stages, const int* fill_rect,
                            int raster_w, int raster_n, kbe_stream_t stream)
{
    KBE_REQUIRE(scratch && frame_u8 && N >= 0 && N <= (1 << 30) && W > 0 && H > 0 && (size_t) W * H <= (1u << 30) &&
                W < (1 << 24) && H < (1 << 24) && ((uintptr_t) scratch & 15) == 0, "kbe_render_frame: bad arguments");
    KBE_REQUIRE(N == 0 || (points && image && depth), "kbe_render_frame: cloud pointers are NULL");
    const FrameJob job = { focal, shift3, scratch, frame_u8, render_f32, existing_f32, zee_f32, zee_pre_f32, stages & (KBE_STAGE_ZBUF_A | KBE_STAGE_ZBUF_B) };
    return render_jobs(points, image, depth, N, W, H, baseline, 1, &job, stages, fill_rect, raster_w, raster_n, (hipStream_t) stream);
}

int kbe_render_frame_group(const float* points, const float* image, const float* depth, int N, int W, int H, double baseline, int n_frames,
                           const double* focals, const float* shifts, void* const* scratch, uint8_t* const* frames_u8, const int* zbuf_flags,
                           int stages, const int* fill_rect, int raster_w, int raster_n, kbe_stream_t stream)
{
    KBE_REQUIRE(n_frames >= 1 && n_frames <= KBE_SCATTER_JOBS && focals && shifts && scratch && frames_u8 && N >= 0 && N <= (1 << 30) && W > 0 && H > 0 &&
                (size_t) W * H <= (1u << 30) && W < (1 << 24) && H < (1 << 24), "kbe_render_frame_group: bad arguments");
    KBE_REQUIRE(N == 0 || (points && image && depth), "kbe_render_frame_group: cloud pointers are NULL");
    FrameJob jobs[KBE_SCATTER_JOBS];
    for (int k = 0; k < n_frames; k++) {
        KBE_REQUIRE(scratch[k] && frames_u8[k] && ((uintptr_t) scratch[k] & 15) == 0, "kbe_render_frame_group: bad scratch / frame pointer");
        for (int j = 0; j < k; j++) KBE_REQUIRE(scratch[j] != scratch[k], "kbe_render_frame_group: the frames of a group need scratch sets of their own");
        jobs[k] = FrameJob{ focals[k], shifts + 3 * (size_t) k, scratch[k], frames_u8[k], nullptr, nullptr, nullptr, nullptr,
                            zbuf_flags ? zbuf_flags[k] & (KBE_STAGE_ZBUF_A | KBE_STAGE_ZBUF_B) : 0 };
    }
    return render_jobs(points, image, depth, N, W, H, baseline, n_frames, jobs, stages & ~(KBE_STAGE_ZBUF_A | KBE_STAGE_ZBUF_B), fill_rect, raster_w, raster_n,
                       (hipStream_t) stream);
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
    KBE_REQUIRE(packed && scratch && frame_u8 && N >= 0 && N <= KBE_FUSED_MAX_POINTS && W > 0 && H > 0 && (size_t) W * H <= (1u << 30) &&
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
        // a frame on its own: the caller keeps no frame parity, so the hole counters (and the binning launch's flags behind
        // them) are zeroed in front of the launch
        const hipError_t e = hipMemsetAsync(sc.hole_count, 0, HOLE_COUNT_INTS * sizeof(int), s);
        if (e != hipSuccess) return fail(KBE_E_LAUNCH, "kbe_render_frame_fused: hipMemsetAsync", e);
    }
    if (stages & KBE_STAGE_TILES) {
        const FusedTarget t = { cam, sc, scratch_place(scratch, W, H), parity, frame_u8, render_f32, existing_f32, zee_f32, zee_pre_f32, -1 };
        launch_frames_fused(s, 1, packed, N, cloud_focal, &t, false, 0, nullptr, fused_build_of_stages(stages));
        if ((rc = launched("kbe_render_frame_fused/scatter"))) return rc;
    }
    if (stages & KBE_STAGE_FILL) {
        FillRect rect = { 0, 0, W - 1, H - 1 };
        if (fill_rect) { rect.x0 = fill_rect[0]; rect.y0 = fill_rect[1]; rect.x1 = fill_rect[2]; rect.y1 = fill_rect[3]; }
        const FillTarget target = { sc, count_now, frame_u8, render_f32, 0, parity >= 0 ? count_next : nullptr };
        launch_fill(s, 1, &target, W, H, stages, dirs, rect, n_tiles);
        rc = launched("kbe_render_frame_fused/fill");
    }
    return rc;
}


int kbe_render_frame_group_fused(const void* packed, int N, double cloud_focal, int W, int H, double baseline, int n_frames, const double* focals,
                                 const float* shifts, void* const* scratch, uint8_t* const* frames_u8, const int* parities, int stages,
                                 const int* fill_rect, kbe_stream_t stream)
{
    KBE_REQUIRE(packed && n_frames >= 1 && n_frames <= KBE_FRAME_JOBS && focals && shifts && scratch && frames_u8 && N >= 0 && N <= KBE_FUSED_MAX_POINTS && W > 0 && H > 0 &&
                (size_t) W * H <= (1u << 30) && W < (1 << 24) && H < (1 << 24) && cloud_focal > 0.0, "kbe_render_frame_group_fused: bad arguments");
    static const FillDirs dirs = make_fill_dirs();
    const hipStream_t s = (hipStream_t) stream;
    FusedTarget ft[KBE_FRAME_JOBS];
    FillTarget targets[KBE_FRAME_JOBS];
    int n_tiles = 0, rc = KBE_OK;
    for (int k = 0; k < n_frames; k++) {
        KBE_REQUIRE(scratch[k] && frames_u8[k] && ((uintptr_t) scratch[k] & 15) == 0, "kbe_render_frame_group_fused: bad scratch / frame pointer");
        for (int j = 0; j < k; j++) KBE_REQUIRE(scratch[j] != scratch[k], "kbe_render_frame_group_fused: the frames of a group need scratch sets of their own");
        const int par = parities ? parities[k] : -1;
        KBE_REQUIRE(par >= -1 && par <= 1, "kbe_render_frame_group_fused: parity is -1, 0 or 1");
        const Scratch sc = carve(scratch[k], W, H);
        n_tiles = sc.tiles_x * sc.tiles_y;
        if (par < 0 && !(stages & KBE_STAGE_KEEP_HOLE_COUNT)) {
            const hipError_t e = hipMemsetAsync(sc.hole_count, 0, HOLE_COUNT_INTS * sizeof(int), s);
            if (e != hipSuccess) return fail(KBE_E_LAUNCH, "kbe_render_frame_group_fused: hipMemsetAsync", e);
        }
        ft[k] = FusedTarget{ make_camera(W, H, focals[k], baseline, shifts + 3 * (size_t) k), sc, scratch_place(scratch[k], W, H), par, frames_u8[k], nullptr, nullptr, nullptr, nullptr, -1 };
        targets[k] = FillTarget{ sc, sc.hole_count + (par == 1 ? 1 : 0), frames_u8[k], nullptr, 0, par >= 0 ? sc.hole_count + (par == 1 ? 0 : 1) : nullptr };
    }
    if (stages & KBE_STAGE_TILES) {
        launch_frames_fused(s, n_frames, packed, N, cloud_focal, ft, false, 0, nullptr, fused_build_of_stages(stages));
        if ((rc = launched("kbe_render_frame_group_fused/scatter"))) return rc;
    }
    if (stages & KBE_STAGE_FILL) {
        FillRect rect = { 0, 0, W - 1, H - 1 };
        if (fill_rect) { rect.x0 = fill_rect[0]; rect.y0 = fill_rect[1]; rect.x1 = fill_rect[2]; rect.y1 = fill_rect[3]; }
        for (int k0 = 0; k0 < n_frames; k0 += KBE_FILL_JOBS)       // the fill takes KBE_FILL_JOBS frames per launch
            launch_fill(s, n_frames - k0 < KBE_FILL_JOBS ? n_frames - k0 : KBE_FILL_JOBS, targets + k0, W, H, stages, dirs, rect, n_tiles);
        rc = launched("kbe_render_frame_group_fused/fill");
    }
    return rc;
}

int kbe_render_frame_group_ahead_ok(int N, int W, int H, int n_frames, int n_next)
{
    return (N >= 0 && W > 0 && H > 0 && n_frames >= 1 && n_frames <= KBE_FRAME_JOBS && n_next >= 1 && n_next <= KBE_FRAME_JOBS &&
            fused_can_place_ahead(N, W, H, n_frames, n_next)) ? 1 : 0;
}

int kbe_render_frame_group_ahead(const void* packed, int N, double cloud_focal, int W, int H, double baseline, int n_frames, const double* focals,
                                 const float* shifts, void* const* scratch, uint8_t* const* frames_u8, const int* turns, int placed, int n_next,
                                 const double* next_focals, const float* next_shifts, void* const* next_scratch, const int* next_turns, int stages,
                                 const int* fill_rect, double near_depth, kbe_stream_t stream)
{
    KBE_REQUIRE(packed && n_frames >= 1 && n_frames <= KBE_FRAME_JOBS && focals && shifts && scratch && frames_u8 && turns && N >= 0 && N <= KBE_FUSED_MAX_POINTS && W > 0 && H > 0 &&
                (size_t) W * H <= (1u << 30) && W < (1 << 24) && H < (1 << 24) && cloud_focal > 0.0, "kbe_render_frame_group_ahead: bad arguments");
    KBE_REQUIRE(n_next >= 0 && n_next <= KBE_FRAME_JOBS && (n_next == 0 || (next_focals && next_shifts && next_scratch && next_turns)), "kbe_render_frame_group_ahead: bad next group");
    KBE_REQUIRE(near_depth >= 0.0 && near_depth < 1.0e30, "kbe_render_frame_group_ahead: near_depth is a depth (0: unknown)");
    KBE_REQUIRE(n_next == 0 || fused_can_place_ahead(N, W, H, n_frames, n_next), "kbe_render_frame_group_ahead: too many placements for the tile launch (kbe_render_frame_group_ahead_ok)");
    static const FillDirs dirs = make_fill_dirs();
    const hipStream_t s = (hipStream_t) stream;
    FusedTarget ft[KBE_FRAME_JOBS], nt[KBE_FRAME_JOBS];
    FillTarget targets[KBE_FRAME_JOBS];
    int n_tiles = 0, rc = KBE_OK;
    for (int k = 0; k < n_frames; k++) {
        KBE_REQUIRE(scratch[k] && frames_u8[k] && ((uintptr_t) scratch[k] & 15) == 0 && turns[k] >= 0, "kbe_render_frame_group_ahead: bad scratch / frame pointer / turn");
        for (int j = 0; j < k; j++) KBE_REQUIRE(scratch[j] != scratch[k], "kbe_render_frame_group_ahead: the frames of a group need scratch sets of their own");
        const Scratch sc = carve(scratch[k], W, H);
        n_tiles = sc.tiles_x * sc.tiles_y;
        const int par = turns[k] & 1;
        ft[k] = FusedTarget{ make_camera(W, H, focals[k], baseline, shifts + 3 * (size_t) k), sc, scratch_place(scratch[k], W, H), par, frames_u8[k], nullptr, nullptr, nullptr, nullptr, turns[k] };
        targets[k] = FillTarget{ sc, sc.hole_count + par, frames_u8[k], nullptr, 0, sc.hole_count + (par ^ 1) };
    }
    for (int k = 0; k < n_next; k++) {
        KBE_REQUIRE(next_scratch[k] && ((uintptr_t) next_scratch[k] & 15) == 0 && next_turns[k] >= 0, "kbe_render_frame_group_ahead: bad scratch / turn of the next group");
        for (int j = 0; j < k; j++) KBE_REQUIRE(next_scratch[j] != next_scratch[k], "kbe_render_frame_group_ahead: the frames of a group need scratch sets of their own");
        for (int j = 0; j < n_frames; j++)
            KBE_REQUIRE(next_scratch[k] != scratch[j] || next_turns[k] == turns[j] + 1, "kbe_render_frame_group_ahead: a set used by both groups takes consecutive turns");
        nt[k] = FusedTarget{ make_camera(W, H, next_focals[k], baseline, next_shifts + 3 * (size_t) k), carve(next_scratch[k], W, H), scratch_place(next_scratch[k], W, H),
                             next_turns[k] & 1, nullptr, nullptr, nullptr, nullptr, nullptr, next_turns[k] };
    }
    // (every argument has been checked: only now is anything enqueued)
    // A set's first turn -- in this group, without placements made ahead, or joining the sequence with the next group: its hole
    // counters and list totals start from zero (as a frame on its own zeroes them), and so do BOTH banks of its per-tile list
    // counters: a sequence that was abandoned (an error return, a caller that stopped after a launch that placed ahead) leaves
    // the counters of the bank it placed into standing, and placements appended behind stale counts list sub-blocks twice.
    auto start_set = [&](const Scratch& sc) -> hipError_t {
        hipError_t e = hipMemsetAsync(sc.hole_count, 0, HOLE_COUNT_INTS * sizeof(int), s);
        if (e == hipSuccess) e = hipMemsetAsync(sc.tile_count, 0, 2 * align16(4 * (size_t) sc.tiles_x * sc.tiles_y * CNT_STRIDE), s);
        return e;
    };
    for (int k = 0; k < n_frames; k++)
        if (turns[k] == 0 && !placed) {
            const hipError_t e = start_set(ft[k].sc);
            if (e != hipSuccess) return fail(KBE_E_LAUNCH, "kbe_render_frame_group_ahead: hipMemsetAsync", e);
        }
    for (int k = 0; k < n_next; k++)
        if (next_turns[k] == 0) {
            const hipError_t e = start_set(nt[k].sc);
            if (e != hipSuccess) return fail(KBE_E_LAUNCH, "kbe_render_frame_group_ahead: hipMemsetAsync", e);
        }
    if (stages & KBE_STAGE_TILES) {
        launch_frames_fused(s, n_frames, packed, N, cloud_focal, ft, placed != 0, n_next, nt, fused_build_of_stages(stages), near_depth);
        if ((rc = launched("kbe_render_frame_group_ahead/scatter"))) return rc;
    }
    if (stages & KBE_STAGE_FILL) {
        FillRect rect = { 0, 0, W - 1, H - 1 };
        if (fill_rect) { rect.x0 = fill_rect[0]; rect.y0 = fill_rect[1]; rect.x1 = fill_rect[2]; rect.y1 = fill_rect[3]; }
        for (int k0 = 0; k0 < n_frames; k0 += KBE_FILL_JOBS)
            launch_fill(s, n_frames - k0 < KBE_FILL_JOBS ? n_frames - k0 : KBE_FILL_JOBS, targets + k0, W, H, stages, dirs, rect, n_tiles);
        rc = launched("kbe_render_frame_group_ahead/fill");
    }
    return rc;
}

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

#if defined(KBE_VIDEO_TRACE)
#include <time.h>
static double trace_now() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }
#endif
constexpr int KBE_VIDEO_STAGES = KBE_STAGE_PROJECT | KBE_STAGE_TILES | KBE_STAGE_FILL;
#ifndef KBE_HBM_CONSECUTIVE
#define KBE_HBM_CONSECUTIVE 1
#endif
int kbe_render_video(const float* points, const float* image, const float* depth, int N, int W, int H, double baseline,
                     int n_frames, const double* focals, const float* shifts, int crop_w, int crop_h, void* scratch,
                     uint8_t* stage, int batch, uint8_t* host_out, int raster_w, int raster_n, const void* packed,
                     double cloud_focal, int flags, kbe_stream_t stream, kbe_stream_t copy_stream, int lanes,
                     const kbe_stream_t* lane_streams, double near_depth)
{
    KBE_REQUIRE(n_frames >= 0 && focals && shifts && stage && host_out && W > 0 && H > 0 && batch >= -64 && (!packed || cloud_focal > 0.0) &&
                (size_t) W * H <= (1u << 30) && W < (1 << 24) && H < (1 << 24), "kbe_render_video: bad arguments");
    KBE_REQUIRE((crop_w == 0 && crop_h == 0) || (crop_w > 0 && crop_h > 0 && crop_w <= W && crop_h <= H), "kbe_render_video: bad crop");
    KBE_REQUIRE(lanes >= 1 && lanes <= KBE_MAX_LANES && (lanes == 1 || lane_streams), "kbe_render_video: bad lanes");
    KBE_REQUIRE(near_depth >= 0.0 && near_depth < 1.0e30, "kbe_render_video: near_depth is a depth (0: unknown)");
    const hipStream_t cs = (hipStream_t) stream;
    const size_t fb = (size_t) W * H * 3;
    const size_t sb = (scratch_set_bytes(W, H, N) + 255) & ~(size_t) 255;      // == kbe_video_scratch_stride: lane stride
    const bool crop = crop_w > 0;
    // KBE_VIDEO_FILL_PAIRS (with _FILL_DIST; `scratch` then holds 2 * lanes sets): a lane renders TWO frames, each into a
    // scratch set of its own, and fills them in the same launches.  The table-driven fill is bound by its own chain of
    // dependent look-ups, not by the chip (272 us alone, 352 us with four of them overlapping), and more than four
    // streams do not overlap any better (the hardware queues): two frames per launch are the way to have eight in flight.
    // KBE_VIDEO_FILL_GROUP(n), n <= 4; KBE_VIDEO_GROUP(n), n <= KBE_FRAME_JOBS, for the fused route, whose scatter launches take up to
    // KBE_FRAME_JOBS frames (fill and crop launches then take them four at a time)
    const int wide_group = ((flags >> 5) & 15) + 1;
    const int group = batch <= 0 ? (wide_group > 1 ? wide_group : ((flags >> 1) & 3) + 1) : 1;
    KBE_REQUIRE(group <= (packed ? KBE_FRAME_JOBS : KBE_FILL_JOBS), "kbe_render_video: more frames per launch than the route's launches take");
    KBE_REQUIRE(!packed || N <= KBE_FUSED_MAX_POINTS, "kbe_render_video: the packed cloud's route takes up to 2^28 points (more: the plain cloud's route)");
    // the fused route always takes the group form (one frame per launch is a group of one): its tile launches also make the
    // placements of the lane's NEXT group (launch_frames_fused) unless KBE_VIDEO_NO_AHEAD says otherwise
    const bool pairs = group > 1 || (packed && batch <= 0);
    // Frames are independent, so consecutive frames go to `lanes` HIP streams, each with its own scratch and raw
    // frame: the fixed cost of a kernel boundary on this chip (launch ramp, tail, and the L2 write-back between
    // dependent kernels) is then paid while another frame's kernels run.
    // stage = [KBE_FRAME_JOBS * lanes raw frames][lanes * fin finished frames][ring half 0: batch frames][ring half 1: batch frames].
    hipStream_t ls[KBE_MAX_LANES], ds[1];
    for (int l = 0; l < lanes; l++) ls[l] = l == 0 ? cs : (hipStream_t) lane_streams[l];
    ds[0] = copy_stream ? (hipStream_t) copy_stream : cs;        // only the staged ring (batch > 0) uses it
#if defined(KBE_VIDEO_TRACE)
    const double t_call = trace_now();
#endif
    const int fin = (int) stage_fin_per_lane(batch);                    // finished-frame buffers per lane
    const int slots = fin * lanes;
    uint8_t* const finished = stage + (size_t) KBE_FRAME_JOBS * lanes * fb;
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
        // every scratch set starts the call on hole counter 0: its counters (and list totals) are zeroed here, on `stream`, before
        // the lanes start -- by ONE small launch for all sets (a memset per set was eight launches in front of a video's first frame)
        const Scratch sc0 = carve(scratch, W, H);
        hipLaunchKernelGGL(k_zero_counters, dim3(16, group * lanes), dim3(256), 0, cs, sc0.hole_count, sb, group * lanes, sc0.tile_count,
                           2 * align16(4 * (size_t) sc0.tiles_x * sc0.tiles_y * CNT_STRIDE) / sizeof(int));
        if (int rc0 = launched("kbe_render_video/counters")) return rc0;
    }
    hipEvent_t start = lanes > 1 || (ringed && ds[0] != cs) ? make() : nullptr;
    // the other streams start once everything enqueued on `stream` so far (the cloud) is done
    if (start && ok) {
        (void) hipEventRecord(start, cs);
        for (int l = 1; l < lanes; l++) (void) hipStreamWaitEvent(ls[l], start, 0);
        if (ringed && ds[0] != cs) (void) hipStreamWaitEvent(ds[0], start, 0);
    }
    // which lane renders frame i with one frame per launch (render() below): whole groups of G = -batch consecutive frames when
    // they are handed to pinned host memory per group, round-robin otherwise; a lane's frame count tells its LAST frame
    const bool by_groups = batch < 0 && host_dev != nullptr;
    // The transfer groups of the hand-off by groups: sizes 1, 2, 4, ... (KBE_VIDEO_FAST_RAMP: 1, 3, 7, ...) up to G = -batch, then G.  The link is idle until the
    // first group has been rendered, and a transfer costs ~25 us whatever its size: a video of 20 frames in groups of 2 ran at
    // 41 GB/s, in groups of 16 it would wait for 16 frames before the first byte moves.  Small groups first start the link
    // after one frame; the groups double while it is busy.  Group g goes to lane g % lanes.
    std::vector<int> group_start;                               // group_start[g] .. group_start[g + 1]: the frames of group g
    if (by_groups) {
        const int G = -batch;
        for (int i0 = 0, size = (flags & KBE_VIDEO_EVEN_GROUPS) ? G : 1; i0 < n_frames; ) {
            group_start.push_back(i0);
            i0 += size < n_frames - i0 ? size : n_frames - i0;
            const int grown = (flags & KBE_VIDEO_FAST_RAMP) ? size * 2 + 1 : size * 2;
            size = grown < G ? grown : G;
        }
        group_start.push_back(n_frames);
    }
    const int n_groups = by_groups ? (int) group_start.size() - 1 : 0;
    int lane_frames[KBE_MAX_LANES] = {}, lane_total[KBE_MAX_LANES] = {};
    if (by_groups) for (int g = 0; g < n_groups; g++) lane_total[g % lanes] += group_start[g + 1] - group_start[g];
    else for (int i = 0; i < n_frames; i++) lane_total[i % lanes]++;
    auto render = [&](int i, int l, uint8_t* out) {
        uint8_t* raw = stage + (size_t) l * fb;
        const int fill_flags = lanes >= KBE_FILL_BY_COUNT_MIN_LANES ? (KBE_STAGE_FILL_BY_COUNT | ((flags & KBE_VIDEO_FILL_DIST) ? KBE_STAGE_FILL_DIST : 0)) : 0;
        int rc;
        if (packed)         // the fused scatter on the packed cloud; a lane's frames alternate between its two hole counters
            rc = kbe_render_frame_fused(packed, N, cloud_focal, W, H, focals[i], baseline, shifts + 3 * (size_t) i,
                                        (char*) scratch + (size_t) l * sb, crop ? raw : out, nullptr, nullptr, nullptr, nullptr,
                                        KBE_STAGE_TILES | KBE_STAGE_FILL | fill_flags | ((flags & KBE_VIDEO_FUSED_LEAN) ? KBE_STAGE_FUSED_LEAN : 0) |
                                        ((flags & KBE_VIDEO_FUSED_ROOMY) ? KBE_STAGE_FUSED_ROOMY : 0), crop ? rect : nullptr, lane_frames[l]++ & 1,
                                        (kbe_stream_t) ls[l]);
        else {
            // a lane's frames alternate between the two z-buffers (A, B, A, ...), each clearing the other's in its tile
            // launch; a lane's LAST frame, if it falls on A, takes the stand-alone form (A cleared by its fill launch), so
            // that every call leaves A empty -- B is always cleared before it is used
            // (the lane's last frame by COUNT: with groups of G frames per lane `i + lanes >= n_frames` named frames in the
            // middle of a lane's share, which then ran stand-alone and left B dirty for the lane's next frame)
            const int k = lane_frames[l]++;
            const bool last_of_lane = k + 1 == lane_total[l];
            const int zflags = (k & 1) ? KBE_STAGE_ZBUF_B : (last_of_lane ? 0 : KBE_STAGE_ZBUF_A);
            rc = kbe_render_frame_stages(points, image, depth, N, W, H, focals[i], baseline, shifts + 3 * (size_t) i,
                                         (char*) scratch + (size_t) l * sb, crop ? raw : out, nullptr, nullptr, nullptr, nullptr,
                                         KBE_VIDEO_STAGES | fill_flags | zflags, crop ? rect : nullptr, raster_w, raster_n, (kbe_stream_t) ls[l]);
        }
        if (rc == KBE_OK && crop) rc = kbe_crop_resize_u8(raw, W, H, crop_w, crop_h, out, (kbe_stream_t) ls[l]);
        return rc;
    };
    // up to `group` frames of lane l: each scattered into a scratch set of its own, filled together, cropped
    static const FillDirs fill_dirs = make_fill_dirs();
    int set_frames[KBE_MAX_LANES][KBE_FRAME_JOBS] = {}, set_total[KBE_MAX_LANES][KBE_FRAME_JOBS] = {};
    bool counting = false;
    // the calls of every lane in order (noted by the counting pass): the fused route's tile launch of a group also makes the
    // placements of the lane's next group
    struct GroupCall { int count; int idx[KBE_FRAME_JOBS]; };
    std::vector<GroupCall> lane_calls[KBE_MAX_LANES];
    size_t lane_pos[KBE_MAX_LANES] = {};
    bool lane_placed[KBE_MAX_LANES] = {};
    auto render_group = [&](int l, int count, const int* idx, uint8_t* const* outs) {
        if (counting) {
            GroupCall c = {};
            c.count = count;
            for (int j = 0; j < count; j++) { set_total[l][j]++; c.idx[j] = idx[j]; }
            lane_calls[l].push_back(c);
            return (int) KBE_OK;
        }
        const int fill_flags = (lanes * group >= KBE_FILL_BY_COUNT_MIN_LANES ? KBE_STAGE_FILL_BY_COUNT : 0) | ((flags & KBE_VIDEO_FILL_DIST) ? KBE_STAGE_FILL_DIST : 0);
        uint8_t* raws[KBE_FRAME_JOBS];
        int rc = KBE_OK;
        if (!packed) {
            // bucket route: every launch (projection, tiles, fill) takes the whole group
            FrameJob jobs[KBE_FILL_JOBS];
            for (int j = 0; j < count; j++) {
                const int i = idx[j], k = set_frames[l][j]++;
                const bool last_of_set = k + 1 == set_total[l][j];
                raws[j] = stage + (size_t) (KBE_FRAME_JOBS * l + j) * fb;
                jobs[j] = FrameJob{ focals[i], shifts + 3 * (size_t) i, (char*) scratch + (size_t) (group * l + j) * sb, crop ? raws[j] : outs[j], nullptr, nullptr,
                                    nullptr, nullptr, (k & 1) ? KBE_STAGE_ZBUF_B : (last_of_set ? 0 : KBE_STAGE_ZBUF_A) };
            }
            rc = render_jobs(points, image, depth, N, W, H, baseline, count, jobs, KBE_VIDEO_STAGES | ((fill_flags & KBE_STAGE_FILL_BY_COUNT) ? fill_flags : 0),
                             crop ? rect : nullptr, raster_w, raster_n, ls[l]);
        } else {
            // fused route: the binning launch, the tile launch and the fill each take the whole group
            FillTarget targets[KBE_FRAME_JOBS];
            FusedTarget ft[KBE_FRAME_JOBS];
            for (int j = 0; j < count; j++) {
                const int i = idx[j], k = set_frames[l][j]++;
                char* const scr = (char*) scratch + (size_t) (group * l + j) * sb;
                raws[j] = stage + (size_t) (KBE_FRAME_JOBS * l + j) * fb;
                uint8_t* const target = crop ? raws[j] : outs[j];
                const Scratch sc = carve(scr, W, H);
                ft[j] = FusedTarget{ make_camera(W, H, focals[i], baseline, shifts + 3 * (size_t) i), sc, scratch_place(scr, W, H), k & 1, target, nullptr, nullptr, nullptr, nullptr, k };
                targets[j] = FillTarget{ sc, sc.hole_count + (k & 1), target, nullptr, 0, sc.hole_count + ((k & 1) ^ 1) };
            }
            if (count == 0) return rc;
            // the lane's next group: its placements ride in this group's tile launch (set j then takes its next turn)
            FusedTarget nt[KBE_FRAME_JOBS];
            int n_next = 0;
            const size_t pos = lane_pos[l]++;
            // (not for a next group that is LARGER: that is the ramp of the transfer groups at the start of a delivered video,
            // where the first frames should leave as early as they can -- a one-frame launch that also places four frames holds
            // the first transfer back: --steps 20 14.1 k -> 13.4 k frames/s with it)
            if (!(flags & KBE_VIDEO_NO_AHEAD) && pos + 1 < lane_calls[l].size() && lane_calls[l][pos + 1].count <= count &&
                fused_can_place_ahead(N, W, H, count, lane_calls[l][pos + 1].count)) {
                const GroupCall& nc = lane_calls[l][pos + 1];
                n_next = nc.count;
                for (int j = 0; j < n_next; j++) {
                    const int i = nc.idx[j], k = set_frames[l][j];
                    char* const scr = (char*) scratch + (size_t) (group * l + j) * sb;
                    nt[j] = FusedTarget{ make_camera(W, H, focals[i], baseline, shifts + 3 * (size_t) i), carve(scr, W, H), scratch_place(scr, W, H), k & 1, nullptr, nullptr, nullptr, nullptr, nullptr, k };
                }
            }
            launch_frames_fused(ls[l], count, packed, N, cloud_focal, ft, lane_placed[l], n_next, nt, (flags & KBE_VIDEO_FUSED_LEAN) ? 1 : ((flags & KBE_VIDEO_FUSED_ROOMY) ? 2 : 0), near_depth);
            lane_placed[l] = n_next > 0;
            if ((rc = launched("kbe_render_video/scatter"))) return rc;
            FillRect fr = { 0, 0, W - 1, H - 1 };
            if (crop) { fr.x0 = rect[0]; fr.y0 = rect[1]; fr.x1 = rect[2]; fr.y1 = rect[3]; }
            for (int j0 = 0; j0 < count; j0 += KBE_FILL_JOBS)
                launch_fill(ls[l], count - j0 < KBE_FILL_JOBS ? count - j0 : KBE_FILL_JOBS, targets + j0, W, H,
                            KBE_STAGE_FILL | ((fill_flags & KBE_STAGE_FILL_BY_COUNT) ? fill_flags : 0), fill_dirs, fr, targets[0].sc.tiles_x * targets[0].sc.tiles_y);
            rc = launched("kbe_render_video/fill");
        }
        for (int j0 = 0; j0 < count && rc == KBE_OK && crop; j0 += KBE_FILL_JOBS)
            rc = crop_resize_group(count - j0 < KBE_FILL_JOBS ? count - j0 : KBE_FILL_JOBS, raws + j0, W, H, crop_w, crop_h, outs + j0, ls[l]);
        return rc;
    };
    // whoever synchronises `stream` afterwards also sees every frame delivered and every other stream idle
    auto join = [&]() {
        for (int l = 1; l < lanes && ok; l++) { hipEvent_t e = make(); if (e) { (void) hipEventRecord(e, ls[l]); (void) hipStreamWaitEvent(cs, e, 0); } }
        if (ringed && ds[0] != cs && ok) { hipEvent_t e = make(); if (e) { (void) hipEventRecord(e, ds[0]); (void) hipStreamWaitEvent(cs, e, 0); } }
    };
    int rc = KBE_OK;
    SdmaCall sdma;
    if (!ringed && !per_frame) {
        // host_out is device memory: the last kernel of every frame stores straight into it (the frames stay in HBM)
        if (pairs) {
            // a chunk of group * lanes CONSECUTIVE frames at a time, lane l taking frames base + l * group .. base + l * group + group - 1 of
            // it (consecutive cameras share candidate lists, kbe_fused.hip: share_plan; until round 5 lane l took frames l, l + lanes,
            // ...: cameras `lanes` steps apart in every launch).  The last, partial chunk fills the lanes one after the other, `group`
            // frames each.  (ADVICE r5 proposed to deal it evenly, ceil(remaining / lanes) frames per lane, so that every lane has work:
            // measured in round 6 -- tools/batches/gpu_r06_tail.sh, frames/s left in HBM, lane by lane / evenly: 6 frames 26.5 / 26.4 k, 8: 29.5 /
            // 29.5, 12: 33.5 / 33.4, 20: 35.9 / 33.9, 24: 37.6 / 35.5, 30: 38.2 / 37.7, 40: 39.8 / 39.6 -- never faster, 5 % slower
            // where it turns one four-frame launch into four one-frame launches: a launch's ramp and tail cost more than the idle lanes.)
            // (A first pass counts the frames of every scratch set: the bucket route must know a set's last frame.)
            for (int pass = 0; pass < 2 && rc == KBE_OK; pass++) {
                counting = pass == 0;
                for (int base = 0; base < n_frames && rc == KBE_OK; base += group * lanes)
                    for (int l = 0; l < lanes && rc == KBE_OK; l++) {
                        int idx[KBE_FRAME_JOBS], count = 0;
                        uint8_t* outs[KBE_FRAME_JOBS];
                        for (int m = 0; m < group; m++) {
                            const int i = KBE_HBM_CONSECUTIVE ? base + l * group + m : base + m * lanes + l;
                            if (i < n_frames) { idx[count] = i; outs[count++] = host_out + (size_t) i * fb; }
                        }
                        if (count) rc = render_group(l, count, idx, outs);
                    }
            }
        } else
        for (int i = 0; i < n_frames && rc == KBE_OK; i++) rc = render(i, i % lanes, host_out + (size_t) i * fb);
    } else if (!ringed) {
        // Hand-off to pinned host memory in the lane's own stream (no event), the lanes taking turns:
        //   batch == 0   per frame, k_deliver (a lane alternates between two finished-frame slots);
        //   batch < 0    per group of G = -batch consecutive frames, rendered by ONE lane into its G slots and sent with one
        //                runtime transfer (hipMemcpyAsync) between a gate kernel that waits for the turn and one that
        //                passes it on.
        // (where the rendering binds, not the link, the transfers need no order: a lane that waits for its turn only idles)
        DeliverCtl* const ctl_turns = lanes > 1 && !(flags & KBE_VIDEO_FREE_TRANSFERS) ? (DeliverCtl*) (stage + ctl_offset) : nullptr;
        DeliverCtl* ctl = ctl_turns;
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
            // KBE_VIDEO_SDMA: the groups leave through an SDMA engine (above); whatever keeps HSA from it falls back to the runtime's
            // transfers, group by group
            // (opened behind the first group's launches, below: the queries cost the host 10-25 us, which would otherwise sit in front of
            // the video's first kernel)
            bool sdma_asked = !(flags & KBE_VIDEO_SDMA);
            volatile int64_t* lane_fin[KBE_MAX_LANES] = {};         // the completion signal of the group the lane's slots hold
            // how long a group may wait for its turn: the transfers of every other lane in front of it (a poll is ~1 us; the
            // link moves ~50 bytes per ns), three times over -- a fixed 4 ms was no margin for 16 frames of 2048^2 (3.8 ms each)
            const double group_us = (double) G * (double) fb / 50.0e3;
            const int turn_polls = (int) fmin(fmax(3.0 * lanes * group_us, (double) DELIVER_MAX_POLLS), 1.0e6);
            if (pairs) {                                                // the frames of every scratch set, counted first
                counting = true;
                for (int g = 0; g < n_groups; g++) {
                    const int l = g % lanes, nb = group_start[g + 1] - group_start[g];
                    for (int k = 0; k < nb; k += group) {
                        int idx[KBE_FRAME_JOBS], count = 0;
                        for (int m = 0; m < group && k + m < nb; m++) idx[count++] = group_start[g] + k + m;
                        (void) render_group(l, count, idx, nullptr);
                    }
                }
                counting = false;
            }
#if defined(KBE_VIDEO_GPU_TRACE)     // dev build only (tools/handoff_gpu_trace.py): when, on the device, a group's frames are ready, its gate opens, its transfer ends
            std::vector<hipEvent_t> ev_begin(n_groups), ev_ready(n_groups), ev_gate(n_groups), ev_copied(n_groups);
            hipEvent_t ev_start;
            (void) hipEventCreate(&ev_start);
            (void) hipEventRecord(ev_start, cs);
            for (int g = 0; g < n_groups; g++) { (void) hipEventCreate(&ev_begin[g]); (void) hipEventCreate(&ev_ready[g]); (void) hipEventCreate(&ev_gate[g]); (void) hipEventCreate(&ev_copied[g]); }
#endif
            for (int g = 0; g < n_groups && rc == KBE_OK; g++) {
                const int l = g % lanes, i0 = group_start[g], nb = group_start[g + 1] - i0;
#if defined(KBE_VIDEO_GPU_TRACE)
                (void) hipEventRecord(ev_begin[g], ls[l]);
#endif
#if defined(KBE_VIDEO_TRACE)
                const double t_g = trace_now();
#endif
                uint8_t* base = finished + (size_t) l * fin * fb;
                // (SDMA: the lane's slots hold the group it sent last -- until that has left.  Two sets of slots per lane, so that a
                // lane renders its next group while its last one leaves, measured SLOWER: 14.2 k instead of 14.6 k frames/s for 20
                // frames, 15.9 k instead of 16.6 k for 75 -- a lane then renders the group after next while the other lane still
                // renders the group the engine waits for)
                if (lane_fin[l]) { hipLaunchKernelGGL(k_signal_wait, dim3(1), dim3(64), 0, ls[l], lane_fin[l], sdma_pool().wait_ticks, sdma_pool().gave_up); lane_fin[l] = nullptr; }
                if (pairs) for (int k = 0; k < nb && rc == KBE_OK; k += group) {
                    int idx[KBE_FRAME_JOBS], count = 0;
                    uint8_t* outs[KBE_FRAME_JOBS];
                    for (int m = 0; m < group && k + m < nb; m++) { idx[count] = i0 + k + m; outs[count++] = base + (size_t) (k + m) * fb; }
                    rc = render_group(l, count, idx, outs);
                }
                else for (int k = 0; k < nb && rc == KBE_OK; k++) rc = render(i0 + k, l, base + (size_t) k * fb);
                if (rc != KBE_OK) break;
#if defined(KBE_VIDEO_TRACE)     // dev build only (tools/handoff_trace.py): where does the host spend the call?
                const double t_r = trace_now();
#endif
#if defined(KBE_VIDEO_GPU_TRACE)
                (void) hipEventRecord(ev_ready[g], ls[l]);
#endif
                if (!sdma_asked) {
                    sdma_asked = true;
                    (void) sdma_open(sdma, stage, host_out);
                }
                // the engine takes the copies in order: no turns -- while it takes them (a copy it refuses sends the rest of the video
                // through the runtime's transfers, which take turns again: ADVICE r5)
                ctl = sdma.ok ? nullptr : ctl_turns;
                if (ctl) hipLaunchKernelGGL(k_turn, dim3(1), dim3(64), 0, ls[l], ctl, 0, turn_polls);
#if defined(KBE_VIDEO_GPU_TRACE)
                (void) hipEventRecord(ev_gate[g], ls[l]);
#endif
#if defined(KBE_VIDEO_TRACE)
                const double t_t = trace_now();
#endif
                bool sent = false;
                if (sdma.ok) {
                    SdmaPair p;
                    const bool paired = sdma_pair(sdma, p);
                    if (paired && hsa_amd_memory_async_copy_on_engine(host_out + (size_t) i0 * fb, sdma.cpu, base, sdma.gpu, (size_t) nb * fb, 1, &p.dep, p.fin,
                                                                      sdma.engine[sdma.sent++ & 1], true) == HSA_STATUS_SUCCESS) {
                        // the copy is on the engine and waits for its release: from here on an error must go through sdma_abort
                        const bool inject = (flags & KBE_VIDEO_INJECT_FAULT) && g == (n_groups > 1 ? 1 : 0);
                        if (!inject) hipLaunchKernelGGL(k_signal_release, dim3(1), dim3(64), 0, ls[l], p.dep_value);
                        const hipError_t le = inject ? hipErrorLaunchFailure : hipGetLastError();
                        if (le != hipSuccess) { rc = fail(KBE_E_LAUNCH, inject ? "kbe_render_video: injected hand-off fault (KBE_VIDEO_INJECT_FAULT)" : "kbe_render_video/release", le); break; }
                        sdma.used.back().released = true;
                        lane_fin[l] = p.fin_value;
                        sent = true;
                    } else {
                        // this group and the rest: the runtime's transfers (the pair, if there is one, was never handed to the engine: idle again)
                        if (paired) { std::lock_guard<std::mutex> lock(sdma_pool().mu); sdma_pool().idle.push_back(sdma.used.back()); sdma.used.pop_back(); }
                        sdma.ok = false;
                        if (ctl_turns && !ctl) { ctl = ctl_turns; hipLaunchKernelGGL(k_turn, dim3(1), dim3(64), 0, ls[l], ctl, 0, turn_polls); }
                    }
                }
                const hipError_t e = sent ? hipSuccess : hipMemcpyAsync(host_out + (size_t) i0 * fb, base, (size_t) nb * fb, hipMemcpyDeviceToHost, ls[l]);
                if (e != hipSuccess) { rc = fail(KBE_E_LAUNCH, "kbe_render_video: hipMemcpyAsync", e); break; }
#if defined(KBE_VIDEO_GPU_TRACE)
                (void) hipEventRecord(ev_copied[g], ls[l]);
#endif
#if defined(KBE_VIDEO_TRACE)
                const double t_c = trace_now();
#endif
                if (ctl) hipLaunchKernelGGL(k_turn, dim3(1), dim3(64), 0, ls[l], ctl, 1, 0);
                rc = launched("kbe_render_video/turn");
#if defined(KBE_VIDEO_TRACE)
                fprintf(stderr, "group %3d lane %d frames %3d..%3d: t=%8.1f us  render-enqueue %7.1f  turn %6.1f  hipMemcpyAsync %7.1f  turn %6.1f\n", g, l, i0, i0 + nb - 1,
                        (t_g - t_call) * 1e6, (t_r - t_g) * 1e6, (t_t - t_r) * 1e6, (t_c - t_t) * 1e6, (trace_now() - t_c) * 1e6);
#endif
            }
            // (SDMA) a lane is done when its last group has left: whoever waits for the lanes (join) waits for the frames
            for (int l = 0; l < lanes; l++)
                if (lane_fin[l] && rc == KBE_OK)            // (KBE_VIDEO_INJECT_TIMEOUT: one tick of patience -- the give-up path, for its test)
                    hipLaunchKernelGGL(k_signal_wait, dim3(1), dim3(64), 0, ls[l], lane_fin[l], (flags & KBE_VIDEO_INJECT_TIMEOUT) ? 1ull : sdma_pool().wait_ticks, sdma_pool().gave_up);
#if defined(KBE_VIDEO_GPU_TRACE)
            for (int l = 0; l < lanes; l++) (void) hipStreamSynchronize(ls[l]);
            for (int g = 0; g < n_groups && rc == KBE_OK; g++) {
                float b = 0, r = 0, o = 0, c = 0;
                (void) hipEventElapsedTime(&b, ev_start, ev_begin[g]); (void) hipEventElapsedTime(&r, ev_start, ev_ready[g]);
                (void) hipEventElapsedTime(&o, ev_start, ev_gate[g]); (void) hipEventElapsedTime(&c, ev_start, ev_copied[g]);
                fprintf(stderr, "group %3d lane %d frames %3d..%3d: render %8.1f .. %8.1f us, gate open %8.1f, transfer ends %8.1f (%.1f GB/s from the gate)\n", g, g % lanes,
                        group_start[g], group_start[g + 1] - 1, b * 1e3, r * 1e3, o * 1e3, c * 1e3, (double) (group_start[g + 1] - group_start[g]) * fb / ((c - o) * 1e6));
            }
#endif
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
    if (rc != KBE_OK) sdma_abort(sdma, ls, lanes);       // no copy of a failed call outlives it, none keeps waiting for its release
    join();
    sdma_close(sdma, cs);
    destroy();
    if (rc == KBE_OK && !ok) rc = fail(KBE_E_LAUNCH, "kbe_render_video: hipEventCreate");
    return rc;
}

int kbe_video_handoff_status(void)
{
    SdmaPool& pool = sdma_pool();
    std::unique_lock<std::mutex> lock(pool.mu);
    // the word: 0 = nothing happened, 1 = a polling kernel gave up (not yet reported), 2 = reported -- the engine stays off either way
    // (sdma_open looks for != 0), but the error is this call's to report ONCE: the videos after it leave through the runtime's transfers
    // and are complete, their callers must not be told otherwise
    if (!pool.gave_up || __atomic_load_n(pool.gave_up, __ATOMIC_ACQUIRE) != 1) return KBE_OK;
    __atomic_store_n(pool.gave_up, 2, __ATOMIC_RELEASE);
    pool.state = -1;                                    // no more copies through the engine
    // the copies of calls that are still on record may yet complete: wait for them here, so that the caller may free its buffers
    std::vector<SdmaGeneration> gens;
    gens.swap(pool.running);
    lock.unlock();
    const auto t0 = std::chrono::steady_clock::now();
    for (SdmaGeneration& g : gens) {
        for (SdmaPair& p : g.pairs)
            while (hsa_signal_load_scacquire(p.fin) > 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < KBE_SDMA_WAIT_SECONDS)
                std::this_thread::sleep_for(std::chrono::microseconds(50));
        (void) hipEventDestroy(g.done);                 // (their signals are never reused)
    }
    return fail(KBE_E_LAUNCH, "kbe_render_video: an SDMA hand-off gave up waiting for its engine -- the frames of that video are not all in host memory; the engine is not used again");
}

}  // extern "C"
