// kbe_tiles.h -- what the kernels of the frame loop share (kbe_frame.hip: projection, tile renderers, the loop itself;
// kbe_fused.hip: the one-launch scatter; kbe_holes.hip: the hole fill): tile geometry, the per-view scratch, and the tile
// machinery in LDS -- per-pixel record lists, degrid, z-tested gather, epilogue -- that k_tiles, k_tiles_nc and k_frame run.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <type_traits>

#include "kbe.h"
#include "kbe_device.h"
#include "kbe_fill.h"
#include "kbe_host.h"
#include "kbe_cloud.h"

#pragma clang fp contract(off)

namespace kbe {

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
#define KBE_TILE_CAP 736     // with 736 (and not 768) k_frame's workgroup stays under 30 KB of LDS: five of them fit a CU
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
#ifndef KBE_CAND_CAP
#define KBE_CAND_CAP (2048 * 16 / KBE_CLOUD_SUB)     // (512 until round 5: from 5 points per pixel on the densest tiles' lists overflowed and those tiles scanned the
                                                    // cloud -- 2.5 ms per 1024^2 frame at 5 per pixel, 13 ms at 9; with 1024 entries 175 and 283 us, and the headline
                                                    // launch costs the same with 512, 768 or 1024: profiles/r05_density_sweep.txt)
#endif
constexpr int CNT_STRIDE = 32;                      // ints between two bucket counters: one 128-byte line each, so that
                                                    // the counter atomics of neighbouring tiles do not serialise in L2
static_assert(TW * TH % TILE_THREADS == 0 && TILE_THREADS % 64 == 0 && REC_CAP >= TILE_THREADS && TW % 32 == 0, "tile geometry");

struct Scratch {                            // carve-out of the caller's scratch allocation
    uint32_t* zkeys;        // [H*W]  z-buffer as order-preserving keys; KBE_ZKEY_EMPTY between frames
    uint32_t* zkeys_b;      // [H*W]  second z-buffer: consecutive frames of a video alternate, each clearing the other's in its tile launch
    uint8_t* dist;          // [H*W]  Chebyshev distance to the nearest valid pixel, capped (frames with very many holes: k_hole_dist)
    float2* strips;         // [16][W + H + 8]  per fill direction and line across the image: where along it valid pixels can be (k_hole_dist)
    uint8_t* dist_blocks;   // [tiles_y * TH / 8][tiles_x * TW / 8]  the same distance between 8 x 8 blocks, in blocks
    int* tile_count;        // [2][n_tiles * CNT_STRIDE]  records appended to each bucket; 0 between frames (the second bank: fused route, below)
    int* hole_count;        // [8]  two hole counters (the fused route alternates them), then the fused route's three "wide list entries" totals (FLAG_*)
    int4* bbox;             // [n_tiles]: per tile, x0, y0, x1, y1 of its valid pixels (inclusive; empty: x0 > x1); plain stores
    uint32_t* coarse;       // [n_tiles]: bit (cy * (TW/8) + cx) = the 8x8 block (cx, cy) of the tile holds a valid pixel
    int* holes;             // [H*W]
    float* depth;           // [H*W]  render[3] * (existing > 0): the fill compares the two ends of a ray with it
    uint32_t* mask;         // [H][ceil(W/32)]  bit = depth > 0: what the fill walks on (32x smaller than the plane)
    int* cand;              // [2][n_tiles][KBE_CAND_CAP]  fused route: the sub-blocks of the packed cloud that can reach each tile (k_place).
                            // TWO banks of lists, counters and placements: a tile launch can make the placements of the set's NEXT frame
                            // (the other bank) while it renders this one (kbe_fused.hip)
    float4* buckets;        // [n_tiles][BUCKET_STRIDE]  {ox, oy, dblError, point index}
    int tiles_x, tiles_y;
};

inline size_t align16(size_t v) { return (v + 15) & ~(size_t) 15; }

inline Scratch carve(void* base, int W, int H)
{
    char* p = (char*) base;
    const size_t hw = (size_t) W * H;
    Scratch s;
    s.tiles_x = (W + TW - 1) / TW;
    s.tiles_y = (H + TH - 1) / TH;
    const size_t n_tiles = (size_t) s.tiles_x * s.tiles_y;
    s.zkeys = (uint32_t*) p;      p += align16(4 * hw);
    s.tile_count = (int*) p;      p += 2 * align16(4 * n_tiles * CNT_STRIDE);
    s.hole_count = (int*) p;      p += 32;
    s.bbox = (int4*) p;           p += align16(16 * n_tiles);
    s.coarse = (uint32_t*) p;     p += align16(4 * n_tiles);
    s.holes = (int*) p;           p += align16(4 * hw);
    s.depth = (float*) p;         p += align16(4 * hw);
    s.mask = (uint32_t*) p;       p += align16(4 * (size_t) H * ((W + 31) / 32));
    s.zkeys_b = (uint32_t*) p;    p += align16(4 * hw);
    s.dist = (uint8_t*) p;        p += align16(hw);
    s.strips = (float2*) p;       p += align16(8 * 16 * (size_t) (W + H + 8));
    s.dist_blocks = (uint8_t*) p; p += align16(n_tiles * (TW / 8) * (TH / 8));
    s.cand = (int*) p;            p += 2 * align16(4 * n_tiles * KBE_CAND_CAP);
    s.buckets = (float4*) p;
    return s;
}

// the fixed part of a scratch set; a set used by the fused route is followed by its placement array (scratch_set_bytes)
inline size_t scratch_bytes(int W, int H)
{
    const size_t hw = (size_t) W * H;
    const size_t n_tiles = (size_t) ((W + TW - 1) / TW) * ((H + TH - 1) / TH);
    return align16(4 * hw) + 2 * align16(4 * n_tiles * CNT_STRIDE) + 32 + align16(16 * n_tiles) + align16(4 * n_tiles) + align16(4 * hw) + align16(4 * hw) +
           align16(4 * (size_t) H * ((W + 31) / 32)) + align16(4 * hw) + align16(hw) + align16(8 * 16 * (size_t) (W + H + 8)) + align16(n_tiles * (TW / 8) * (TH / 8)) + 2 * align16(4 * n_tiles * KBE_CAND_CAP) + n_tiles * BUCKET_STRIDE * sizeof(float4);
}


size_t fused_place_bytes(int N);    // kbe_fused.hip: 12 bytes per packed point, a frame's placements {ox, oy, dblError}
inline size_t scratch_set_bytes(int W, int H, int N) { return align16(scratch_bytes(W, H)) + (N >= 0 ? 2 * align16(fused_place_bytes(N)) : 0); }
inline void* scratch_place(void* base, int W, int H) { return (char*) base + align16(scratch_bytes(W, H)); }
// the fused route's banks of a scratch set (bank 0 / 1): counters, lists, placements
inline int* bank_tile_count(const Scratch& s, int bank) { return (int*) ((char*) s.tile_count + (size_t) bank * align16(4 * (size_t) s.tiles_x * s.tiles_y * CNT_STRIDE)); }
inline int* bank_cand(const Scratch& s, int bank) { return (int*) ((char*) s.cand + (size_t) bank * align16(4 * (size_t) s.tiles_x * s.tiles_y * KBE_CAND_CAP)); }
inline void* bank_place(void* place, int N, int bank) { return (char*) place + (size_t) bank * align16(fused_place_bytes(N)); }
constexpr int HOLE_COUNT_INTS = 8;          // ints of Scratch::hole_count: [0], [1] hole counters, [2..4] the fused route's list totals

struct FillRect { int x0, y0, x1, y1; };    // inclusive; only holes inside are filled

#ifndef KBE_FILL_BY_COUNT_MIN_LANES
#define KBE_FILL_BY_COUNT_MIN_LANES 2       // frames in flight from which the video loop lets the fill pick its schedule by the hole count
#endif

// kbe_holes.hip: the hole fill of one frame, or of two frames of the same size in the same launches -- with
// KBE_STAGE_FILL_DIST the tables (k_hole_dist) and the table-driven fill (k_fill_tables) in front of k_fill_holes; each
// returns at once when a frame has fewer holes than its schedule asks for
constexpr int KBE_FILL_JOBS = 4;
#ifndef KBE_FRAME_JOBS_MAX
#define KBE_FRAME_JOBS_MAX 12        // (12 x sizeof(FrameArgs) = 3.8 KB of kernel arguments: the launch takes 4 KB)
#endif
constexpr int KBE_FUSED_MAX_POINTS = 1 << 28;     // the packed cloud's route addresses a point's 16 bytes by a 32-bit byte offset (kbe_fused.hip: KBE_OFFSETS_32)
constexpr int KBE_FRAME_JOBS = KBE_FRAME_JOBS_MAX;      // frames a launch of the fused scatter (k_place, k_frame) takes at most
struct FillTarget {                 // a frame to be filled, on the host
    Scratch sc;
    const int* hole_count;
    uint8_t* frame_u8;
    float* render_f32;
    int reset_scatter_scratch;      // stand-alone frames of the bucket route: the fill launch empties z-buffer A and the bucket counters
    int* next_hole_count;           // fused route: the hole counter the NEXT frame counts in, zeroed here
};
struct FillJob {                    // ... and as the kernels see it
    const int* holes; const int* hole_count; const float* depth; const uint32_t* mask; uint8_t* frame; float* render;
    uint32_t* zkeys; int* tile_count; const int4* bbox; const uint32_t* coarse; uint8_t* dist; float2* strips; uint8_t* dist_blocks;
    int reset_scatter_scratch; int* next_hole_count;
};
struct FillJobs { FillJob j[KBE_FILL_JOBS]; };
void launch_fill(hipStream_t s, int n_jobs, const FillTarget* targets, int W, int H, int stages, const FillDirs& dirs, const FillRect& rect, int n_tiles);
// kbe_hip.hip: kbe_crop_resize_u8 for n <= 4 frames of the same size in one launch
int crop_resize_group(int n, const uint8_t* const* frames, int W, int H, int crop_w, int crop_h, uint8_t* const* outs, hipStream_t stream);
// kbe_fused.hip: the scatter of 1..KBE_FRAME_JOBS frames from the packed cloud (k_place + k_frame, each launch taking all the frames)
struct FusedTarget {
    Camera cam;
    Scratch sc;
    void* place;            // the scratch set's placement arrays (two banks of fused_place_bytes(N) bytes behind its fixed part)
    int parity;             // which of the scratch set's hole counters and banks the frame uses (-1 = 0)
    uint8_t* frame_u8;
    float* render_f32; float* existing_f32; float* zee_f32; float* zee_pre_f32;
    int turn;               // -1: a frame whose placements are made in front of it, list totals alternating with `parity`; k >= 0: the
                            // k-th use of the set in a sequence that may place AHEAD (list total k % 3; parity must be k & 1)
};
// `placed`: the frames' placements and lists exist (made ahead by the previous tile launch on the same sets); n_next > 0: this tile
// launch also makes those of `next` (frames that use the other bank of their sets; turn >= 0 everywhere)
// `build`: 0 = the lean or the roomy build of the tile launch by the cloud's density, 1 = lean, 2 = roomy (KBE_STAGE_FUSED_LEAN / _ROOMY)
// `near_depth`: the depth of the nearest point the caller knows of (objectDepthrange[0], common.py:88), > 0: consecutive frames placed ahead may share
// candidate lists (kbe_fused.hip: share_plan); 0: every frame keeps lists of its own
void launch_frames_fused(hipStream_t s, int n, const void* packed, int N, double cloud_focal, const FusedTarget* targets, bool placed = false, int n_next = 0,
                         const FusedTarget* next = nullptr, int build = 0, double near_depth = 0.0);
inline int fused_build_of_stages(int stages) { return (stages & 1024) ? 1 : ((stages & 2048) ? 2 : 0); }
bool fused_can_place_ahead(int N, int W, int H, int n, int n_next);

// blockIdx -> tile id such that each XCD (block b runs on XCD b % 8) owns a contiguous band of
// tile rows: the records of neighbouring tiles reference neighbouring points (shared L2 lines).
__device__ __forceinline__ int xcd_tile(int b, int n)
{
    const int xcd = b & 7, j = b >> 3, q = n >> 3, r = n & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
}
// The same for frame `job` of a launch that renders several (blockIdx.y = the frame; workgroups are dealt to the XCDs by their
// linear id, so with gridDim.x a multiple of 8 block b of EVERY frame runs on XCD b % 8): the bands rotate with the frame, so
// that an XCD whose band of one frame holds more work (the part of the image where surfaces overlap) gets a lighter band of
// the next -- the launch ends with its slowest XCD (a 4-frame launch spanned 66.0 - 76.2 M cycles by XCD with every frame's
// band fixed: profiles/r04_scatter_wave_timeline.txt).  Only when the bands are equal (n % 8 == 0): else they are not interchangeable.
__device__ __forceinline__ int xcd_tile_rot(int b, int n, int job)
{
#if defined(KBE_XCD_ROT) && !KBE_XCD_ROT
    (void) job;
    return xcd_tile(b, n);
#else
    if (n & 7) return xcd_tile(b, n);
    // (a launch of fewer than eight frames spreads its frames' bands over the eight: 4 frames -> every second band)
    const int stride = gridDim.y >= 8 ? 1 : 8 / (int) gridDim.y;
    return (((b + job * stride) & 7) * (n >> 3)) + (b >> 3);
#endif
}

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
// a pixel's five accumulators; r|g and b|depth as pairs so that they update with packed fp32 instructions
struct PixAcc { f2 rg, bd; float w; };

// CAP: records the tile holds at once.  The scratch, the bucket route and the fused route's launches for clouds denser than the
// raster use REC_CAP (TileLds); the fused route's launches for clouds of about a point per pixel use a smaller one and run six
// workgroups per CU instead of five (kbe_fused.hip: LEAN_CAP).
template <int CAP> struct TileLdsT {
    static constexpr int kCap = CAP;
    static constexpr int kDummy = CAP;          // what an exhausted list reads: dblError = +inf (fails every z test), colours 0, its own successor
    static constexpr int kNull = CAP * 16;      // "no record" as a link: the dummy's byte offset, so that every link can be read as it is
    float4 rec[CAP + 1];        // ox, oy, dblError, link to the next record of the bin; slot kDummy: see gather
    float4 rgbd[CAP + 1];       // the point's r, g, b, depth, fetched once at insert time
    int head[(BH * BW + 3) & ~3];   // link to the first record of each bin.  A link is the record's BYTE offset, kNull = none (rounded up to whole
                                // 16-byte words, so that this array and the next can be written four entries at a time: KBE_LDS_WIDE)
    float zpre[KH * KW];        // z-buffer before degrid, tile + halo; after the degrid: uint8 staging area + per-wave partials
    float zee[TH * TW];         // degridded z-buffer
    int odd_z[TILE_THREADS / 64];   // per wave: some z of tile + halo is outside [2^19, 1e6] (then: the generic, fp64-capable code); 16-byte aligned: read as one word
    int nrec;
};
typedef TileLdsT<REC_CAP> TileLds;
static_assert(TILE_THREADS / 64 == 4 && offsetof(TileLds, odd_z) % 16 == 0, "the four waves' flags are one 16-byte word");
// every z of tile + halo in the band (all four waves say so): wave-uniform
template <class LDS> __device__ __forceinline__ bool lds_tile_is_fast(const LDS& L)
{
    static_assert(offsetof(LDS, odd_z) % 16 == 0, "the four waves' flags are one 16-byte word");
    const int4 o = *(const int4*) L.odd_z;
    return (bool) __builtin_amdgcn_readfirstlane((int) ((o.x | o.y | o.z | o.w) == 0));
}

constexpr int REC_DUMMY = TileLds::kDummy;
constexpr int REC_NULL = TileLds::kNull;

template <class LDS> __device__ __forceinline__ void lds_dummy_record(LDS& L)
{
    L.rec[LDS::kDummy] = make_float4(0.0f, 0.0f, __builtin_inff(), __int_as_float(LDS::kNull));
    L.rgbd[LDS::kDummy] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
}

// every bin's list emptied: 141 16-byte stores by the workgroup's first threads (the array is padded to whole 16-byte words) instead
// of three trips of single stores with their bound tests for all 256
template <class LDS> __device__ __forceinline__ void lds_reset_heads(LDS& L, int tid)
{
    static_assert(sizeof(L.head) % 16 == 0 && offsetof(LDS, head) % 16 == 0 && sizeof(L.head) / 16 <= TILE_THREADS, "one 16-byte store per thread");
    if (tid < (int) (sizeof(L.head) / 16)) ((int4*) L.head)[tid] = make_int4(LDS::kNull, LDS::kNull, LDS::kNull, LDS::kNull);
}

// threads one record into the list of its bin (bin = north-west corner relative to x0-1, y0-1)
template <class LDS> __device__ __forceinline__ void lds_insert(LDS& L, int idx, float ox, float oy, float err, const float4& rgbd, int x0, int y0)
{
    const int bx = (int) floorf(ox) - (x0 - 1), by = (int) floorf(oy) - (y0 - 1);
    L.rgbd[idx] = rgbd;
    const int next = atomicExch(&L.head[__mul24(by, BW) + bx], idx << 4);
    L.rec[idx] = make_float4(ox, oy, err, __int_as_float(next));
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
#ifndef KBE_GATHER_PORT2
#define KBE_GATHER_PORT2 0
#endif
template <bool FAST, class Args, class LDS>
__device__ __forceinline__ void gather(const Args& a, const LDS& L, int tid, int x0, int y0,
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
        // KBE_GATHER_PORT2 (FAST tiles): the same arithmetic in instructions a SIMD issues through its SECOND port (DESIGN.md
        // section 4: plain two-operand fp32 / integer / logic instructions on vector registers co-issue with the first port's,
        // which the launch is bound by) -- 1: the four channels as separate v_mul_f32 + v_add_f32 instead of packed pairs
        // (packed fp32 takes the first port); 2: also the z test as a subtraction whose sign masks the weight instead of a
        // compare and a select.  r.z <= zlimf  <=>  r.z < the float above zlimf  <=>  r.z - that float is negative (both
        // finite, or r.z = +inf on the dummy record: the difference then is +inf; zlimf = zee + 1 with zee in [2^19, 1e6]).
        const float zl2 = __int_as_float(__float_as_int(zlimf) + 1);
        float ar = acc[m].rg.x, ag = acc[m].rg.y, ab = acc[m].bd.x, ad = acc[m].bd.y, aw = acc[m].w;
        auto add = [&](int k, const float4& r, const float4& c) {
            const float wx = (k & 1) ? (r.x - (Xf - 1.0f)) : ((Xf + 1.0f) - r.x);          // k & 1 ? ox - fx : ex - ox
            const float wy = (k >> 1) ? (r.y - (Yf - 1.0f)) : ((Yf + 1.0f) - r.y);
            float w;
            if (FAST && KBE_GATHER_PORT2 >= 2) {
                const float below = r.z - zl2;
                int sign;
                w = wx * wy;
                // (as the instructions: the compiler turns `(x >> 31) & y` back into a compare and a select)
                asm("v_ashrrev_i32 %0, 31, %1" : "=v"(sign) : "v"(below));
                asm("v_and_b32 %0, %1, %2" : "=v"(w) : "v"(w), "v"(sign));
            } else {
                const bool pass = exact ? (r.z <= zlimf) : ((double) r.z <= zlim);         // :639
                w = pass ? wx * wy : 0.0f;
            }
            if (FAST && KBE_GATHER_PORT2 >= 1) {
                ar += c.x * w; ag += c.y * w; ab += c.z * w; ad += c.w * w;                 // :641 product rounded, then added
                aw += w;
            } else {
                const f4 cv = *(const f4*) &c;
                acc[m].rg += cv.xy * w;                                     // :641 product rounded, then added (v_pk_mul_f32, v_pk_add_f32)
                acc[m].bd += cv.zw * w;
                acc[m].w += w;                                              // the `ones` channel (:429)
            }
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
        } while (min(min(nx[0], nx[1]), min(nx[2], nx[3])) < LDS::kNull);      // some list goes on
        if (FAST && KBE_GATHER_PORT2 >= 1) { acc[m].rg.x = ar; acc[m].rg.y = ag; acc[m].bd.x = ab; acc[m].bd.y = ad; acc[m].w = aw; }
    }
}


// degrid (common.py:525-568), out of place: L.zpre (tile + halo, decoded) -> L.zee.  `fast`: every z of tile + halo in
// [2^19, 1e6] (any scene whose points are farther than F*B/475712 from the camera) -> fp32-only, branch-free
template <class Args, class LDS>
__device__ __forceinline__ void tile_degrid(const Args& a, LDS& L, int tid, int x0, int y0, bool fast)
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
template <class Args, class LDS>
__device__ __forceinline__ void tile_epilogue(const Args& a, LDS& L, PixAcc (&acc)[PIX_PER_THREAD], int tile, int x0, int y0)
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
#if defined(KBE_DIV_FAST) && !KBE_DIV_FAST
        const float y = 1.0f / den;
#else
        const float y = div_unscaled(1.0f, den);          // den = w + 1e-7 with 0 <= w <= 4 N: in [2^-24, 2^27], nowhere near a rescaling
#endif
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
#ifndef KBE_EPILOGUE_WIDE
#define KBE_EPILOGUE_WIDE 1
#endif
    if (KBE_EPILOGUE_WIDE && (W & 15) == 0 && (TW * 3) % 16 == 0 && x0 + TW <= W && ((uintptr_t) a.frame & 15) == 0) {
        // ... as 16-byte words where the row segments are 16-byte aligned (a tile row is 96 bytes: six of them): 96 stores for the
        // tile, by the workgroup's LAST threads (the z decode in front of the degrid is the first threads' work) -- a trip for 96
        // threads instead of two trips with their index arithmetic for all 256
        constexpr int Q_PER_ROW = TW * 3 / 16;
        static_assert(TH * Q_PER_ROW <= TILE_THREADS, "one 16-byte store per thread");
        const int i = TILE_THREADS - 1 - tid;
        if (i < TH * Q_PER_ROW) {
            const int ly = i / Q_PER_ROW, k = i - ly * Q_PER_ROW;
            if (y0 + ly < H)
                *(uint4*) (a.frame + ((__umul24((uint32_t) (y0 + ly), (uint32_t) W) + (uint32_t) x0) * 3u + 16u * (uint32_t) k)) = ((const uint4*) s_u8)[ly * Q_PER_ROW + k];
        }
    } else if (dword_rows) {
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

}  // namespace kbe
