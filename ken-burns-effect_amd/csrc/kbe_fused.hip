// kbe_fused.hip -- the fused scatter: render_pointcloud (common.py:428-686) of one frame in ONE launch, from the packed
// cloud (kbe_cloud.h).  Host side: kbe_render_frame_fused / kbe_render_video in kbe_frame.hip, through launch_frame_fused.
#include <stdlib.h>
#include <string.h>
#include "kbe_cloud.h"
#include "kbe_tiles.h"

using namespace kbe;

namespace kbe { PackedCloud cloud_open(const void* packed, int N, double focal); }      // kbe_cloud.hip

namespace {

// ---------------------------------------------------------------------------------------
// THE FUSED SCATTER: render_pointcloud (common.py:428-686) of one frame from the packed cloud (kbe_cloud.h) in two
// launches.  No global z-buffer, no bucket of records, no per-point atomic in HBM: a tile PULLS its points.
//   k_place  one lane per point, in packed order: shift (common.py:104-109), projection (:447-468), dblError (:470); the
//            point's PLACEMENT {ox, oy, dblError} is stored at the point's own index (12 bytes, perfectly coalesced -- no
//            counter, no slot, no grouping of lanes by tile).  The 16 lanes of a SUB-BLOCK (a 4 x 4 quadrant of an 8 x 8
//            source cell) then reduce the bounding box of their north-west corners with four DPP steps, and the sub-block's
//            id is appended to the candidate list of every tile that box can reach: one or two tiles, rarely four, each
//            lane of the sub-block taking one of them (~2 atomics per 16 points).
//   k_frame  one workgroup per 32 x 16 target tile:
//   splat    every point of the listed sub-blocks (~1.6 x the points that actually reach the tile) is fetched with its
//            colours (a 96- and a 128-bit load); one whose north-west corner lies in the tile or within two pixels of it
//            min-splats the key of its dblError into the tile's z-buffer IN LDS (tile + 1-pixel halo: one ds_min_u32 on
//            the winner corner, :472-506), and one whose corner can colour a tile pixel becomes a record
//            {ox, oy, dblError} + {r, g, b, depth} in LDS, threaded into the per-pixel lists at once;
//   then     exactly k_tiles: degrid (:525-568) in LDS, z-tested gather in registers (:586-669), normalise (:686), hole
//            mask (:253), uint8 (:255), coalesced stores.
// A halo pixel's z is the minimum over the points whose WINNER corner it is; those have their north-west corner
// within one more pixel, hence the two-pixel reach of the splat.  Against the bucket route: every point is still projected
// exactly once, but what crosses HBM between the two launches is 12 bytes per point written and read in place instead of
// 16-byte records appended to per-tile buckets (with the counter atomics and the grouping of lanes that takes), the 4-byte
// z-buffer's atomics, its read-back and its reset.
// More than REC_CAP records on a tile (piled-up points, a cloud denser than the raster): the records beyond go to the
// tile's spill area in HBM and are gathered in further rounds.  A list that overflows, sub-blocks whose points scatter
// over the whole frame (an incoherent cloud), more spilled records than the spill area holds: the tile scans block ranges
// of the cloud with inline node tests instead and takes its records in runs that fit.  Slow paths, but any cloud renders
// correctly.
// ---------------------------------------------------------------------------------------
constexpr int MAXC = TILE_THREADS;      // candidate blocks per window of the slow path (one per thread in its prefix scan)
constexpr int SPILL_CAP = 4 * BUCKET_CAP;   // point indices a tile's spill area holds (the area is BUCKET_STRIDE records of 16 bytes)
constexpr int LIST_CAP = KBE_CAND_CAP;      // sub-blocks a tile's candidate list holds (kbe_tiles.h: the scratch is sized by it)
// A sub-block whose points reach more than BIN_WIDE_FAN tiles is "wide": it is still listed for every tile of its box, but
// the frame keeps a running total of such entries, and once that exceeds BIN_WIDE_BUDGET entries per tile of the frame (an
// incoherent cloud: every sub-block reaches every tile) the lists are abandoned and every tile scans the blocks itself.
constexpr int BIN_WIDE_FAN = 64;
constexpr int BIN_WIDE_BUDGET = 32;
__device__ __forceinline__ unsigned bin_budget(int tiles_x, int tiles_y) { return (unsigned) BIN_WIDE_BUDGET * (unsigned) (tiles_x * tiles_y) + 4096u; }
static_assert(MAXC == TILE_THREADS, "one candidate per thread in the prefix scan of the slow path");
static_assert(LIST_CAP % TILE_THREADS == 0 && kCloudBlock % kCloudSub == 0 && (kCloudSub == 16 || kCloudSub == 8), "list geometry: a sub-block is one DPP row, or half of one");

struct Placement { float ox, oy, err; };     // 12 bytes per point and frame; ox = PLACE_NONE: the point touches no pixel
static_assert(sizeof(Placement) == 12, "placement record");
constexpr float PLACE_NONE = -1.0e9f;        // no image position is that large (project_xy drops |ox| >= 1e9)

struct FrameArgs {          // a frame of a tile launch (the cloud is the launch's: FrameJobsT::pc)
    Camera cam;
    int tiles_x, tiles_y;
    const Placement* place; // [Np]  this frame's placements (k_place, or the previous tile launch on the set: place_ahead)
    int* tile_count;        // [n_tiles * CNT_STRIDE]: candidates listed per tile (the placement counts, k_frame zeroes its own)
    const int* cand;        // [n_tiles][LIST_CAP]
    const unsigned* bin_flag;       // this frame's total of wide list entries (beyond the budget the lists are not complete) ...
    unsigned* bin_flag_next;        // ... and the total a LATER frame of the set will count in, zeroed here
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

struct PlaceArgs {          // the placement of one frame: its camera, and where its placements and candidate lists go
    Camera cam;
    Placement* place;
    int* tile_count;
    int* cand;
    unsigned* bin_flag;     // running total of the wide sub-blocks' list entries (beyond the budget: the lists are abandoned)
};

// what a tile launch takes: the cloud, up to J frames to render, and up to J frames whose placements it makes AHEAD -- the
// work of k_place for the frames the NEXT tile launch on this stream renders (they use the other bank of their scratch sets)
// (KBE_SHARED_LISTS) consecutive frames of a launch share candidate lists in SUB-GROUPS: frame k reads the lists of its sub-group's
// lead (itself: lists of its own), and a sub-group's tiles zero the shared counter when the last of its frames has
// read it; of the frames placed ahead, frame k that leads a sub-group k .. last > k makes the lists for those frames
// (the box of a sub-block's corners under cameras k and nx_last[k], widened by `dev`: how far the cameras in between stray from
// the straight line between those two -- a Ken Burns path is a parabola in shift space, common.py:88-100), the others of its
// sub-group store their placements and list nothing.
template <int J> struct FrameJobsT {
    PackedCloud pc;
    int n_next, pad_;
    float dev[3];           // the largest distance, per axis, of a placed camera's shift from the chord of its sub-group
    uint32_t a_share[J];    // lead | size << 8 of the frame's sub-group among the frames this launch renders (dwords: a byte of the kernel
    uint32_t nx_share[J];   // lead | last << 8 among the frames it places     arguments is fetched by a VECTOR load, and the wait for it is a wait
                            //                                                 for every load the prologue has in flight: 15.0 -> 16.5 us per frame)
    FrameArgs a[J];
    PlaceArgs nx[J];
};
static_assert(sizeof(FrameJobsT<KBE_FRAME_JOBS>) <= 4096, "a launch takes 4 KB of kernel arguments");

template <int CAP> struct FrameLdsT {
    TileLdsT<CAP> T;
    int n_ovf;              // records that did not fit the first round and went to the tile's spill area
    int wave_sum[TILE_THREADS / 64];
    int run_end;
};

struct CullView {           // the view, as the node tests of the slow path need it
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

#if defined(KBE_FRAME_STATS)     // dev build only (tools/frame_stats.py): what k_place and the tiles of k_frame did, summed over launches
__device__ unsigned long long g_frame_stats[8];     // tiles, list entries written, candidate sub-blocks, points in z reach, records, slow tiles, spilling tiles, wide sub-blocks
#endif
#if defined(KBE_FRAME_PROBE)     // dev build only (tools/frame_probe.py): the shader clock at the phase boundaries of every wave of a tile launch
constexpr int PROBE_STAMPS = 14, PROBE_WAVES = 1 << 17;
__device__ unsigned long long g_frame_probe[PROBE_WAVES * PROBE_STAMPS];
#define KBE_PROBE(k) do { probe_t[k] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define KBE_PROBE(k) do { } while (0)
#endif

// ---------------------------------------------------------------------------------------
// k_place: the placements and the candidate lists of one frame.  A point matters to tile (tx, ty) when its north-west
// corner (nwx, nwy) lies in [tx TW - 2, tx TW + TW] x [ty TH - 2, ty TH + TH] (k_frame's z reach), i.e. for
// tx = (nwx - 1) >> log2 TW .. (nwx + 2) >> log2 TW; a sub-block is listed for the tiles of the box of its points' corners.
// Points that touch no pixel (behind the near plane, common.py:453; corner outside [-1, W - 1] x [-1, H - 1]) are marked and
// take no part in the box.  The order of a list's entries is the order the atomics retire in: it only decides the order of
// the fp32 sums, as the bucket order does on the other route.
// ---------------------------------------------------------------------------------------
struct PlaceJobs { PackedCloud pc; int tiles_x, tiles_y; PlaceArgs a[KBE_FRAME_JOBS]; };

// a * b + c on the low 24 bits of a and b, as the instruction: the compiler renders __umul24(x, constant) with a value whose
// range it cannot see as v_mul_lo_u32 -- a quarter-rate instruction, twice in every step of the splat
__device__ __forceinline__ uint32_t mad_u24(uint32_t a, uint32_t b, uint32_t c)
{
    uint32_t r;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(b), "v"(c));
    return r;
}

// minimum over the 16 lanes of a DPP row, left in every lane of the row: four v_min_i32 that read their second operand
// through the DPP cross-lane path (the compiler's own rendering of the same steps is a copy, a DPP copy and a min each).
// A DPP read needs two wait states behind the VALU write of its source; inline asm gets no hazard handling, hence the s_nop.
__device__ __forceinline__ int row_min(int v)
{
    // (s_nop 4 in front: should the compiler ever schedule a VALU write of EXEC just ahead, a DPP read needs five wait states behind
    // it.  EVERY lane of the row must be active: bound_ctrl is off, a disabled lane would keep its own value out of the others'
    // minimum -- the callers hold whole waves, Np is a multiple of 64 and every branch around a call is wave-uniform.  gfx9 wave64.)
    if (kCloudSub == 8)         // (dev: sub-blocks of 8 points -- the minimum over each half of the row)
        asm("s_nop 4\n\tv_min_i32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
            "s_nop 1\n\tv_min_i32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
            "s_nop 1\n\tv_min_i32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 1" : "+v"(v));
    else
    asm("s_nop 4\n\tv_min_i32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_min_i32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_min_i32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_min_i32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 1" : "+v"(v));
    return v;
}

// one point of a frame's placement, lane `lane` of a wave whose 64 lanes hold 64 consecutive points (four sub-blocks): point i = `p`.
// In two steps, so that a caller can do other work while the list atomic is under way: place_point_begin returns the list entry
// the lane owes -- the tile and the slot its atomic returned -- and place_point_end writes it.
struct ListSlot { int t, pos; };
// Whose candidate list a placement feeds (KBE_SHARED_LISTS; wave-uniform).  mode 0: the frame's own.  The frames a launch places
// ahead are consecutive cameras of a path -- same focal length, shifts on one line (the host checks: launch_frames_fused) -- and
// list almost the same sub-blocks for the same tiles: the group then shares ONE set of lists, made by the row of its first
// frame (mode 2) from the box of the sub-block's corners under the group's FIRST and LAST camera (a point's position is a
// ratio of two functions linear in the step: monotone between the two, so every frame in between stays inside; SHARE_MARGIN
// pixels cover the rounding of the frames' own fp32 arithmetic and of the shifts), and the other rows make none (mode 1).
#ifndef KBE_SHARED_LISTS
#define KBE_SHARED_LISTS 1
#endif
#ifndef KBE_SHARE_MAX_PX
#define KBE_SHARE_MAX_PX 13.0       // (launch_frames_fused: share_plan)
#endif
// 32-bit byte offsets on the launch's uniform bases (placements, points, colours): with a signed index every such address was a
// sign extension, a 64-bit multiply-add and a 64-bit add.  The route takes at most 2^28 points (KBE_FUSED_MAX_POINTS): x 16 fits.
#ifndef KBE_OFFSETS_32
#define KBE_OFFSETS_32 1
#endif
template <class T> __device__ __forceinline__ T* at_offset32(T* base, uint32_t index)
{
    return KBE_OFFSETS_32 ? (T*) ((char*) base + index * (uint32_t) sizeof(T)) : base + (int) index;
}
template <class T> __device__ __forceinline__ const T* at_offset32(const T* base, uint32_t index)
{
    return KBE_OFFSETS_32 ? (const T*) ((const char*) base + index * (uint32_t) sizeof(T)) : base + (int) index;
}
struct ShareMode { int mode; float dsx, dsy, dsz, dev_x, dev_y, dev_z; };      // ds: the sub-group's last camera's shift minus this (its first) one's; dev: FrameJobsT::dev
constexpr float SHARE_MARGIN = 0.25f;
__device__ __forceinline__ ListSlot place_point_begin(const CloudPoint& p, int i, int lane, const Camera& cam, int tiles_x, int tiles_y, Placement* place,
                                                      int* tile_count, int* cand_lists, unsigned* bin_flag, const ShareMode& share)
{
    ListSlot owed = { -1, 0 };
    float x = p.x, y = p.y, z = p.z, ox = 0.0f, oy = 0.0f;
    apply_shift(cam, x, y, z);
    bool ok = project_xy(cam, x, y, z, ox, oy);
    const bool seen = ok;                                       // in front of the near plane, at a finite position
    int nwx = (int) floorf(ox), nwy = (int) floorf(oy);
    ok = ok && ((unsigned) (nwx + 1) <= (unsigned) cam.W) & ((unsigned) (nwy + 1) <= (unsigned) cam.H);      // a corner inside the image: -1 <= nw < size
    Placement pl;
    pl.ox = ok ? ox : PLACE_NONE;
    pl.oy = ok ? oy : PLACE_NONE;
    pl.err = project_err_fast(cam, ok ? z : 1024.0f);
    *at_offset32(place, (uint32_t) i) = pl;
    int nex = nwx, ney = nwy;                                   // the corner's range over the frames the list serves: [nwx, nex] x [nwy, ney]
    if (KBE_SHARED_LISTS && share.mode == 1) return owed;       // uniform
    if (KBE_SHARED_LISTS && share.mode == 2) {
        // the point under the group's last camera, to the accuracy a box needs (a reciprocal instead of the division)
        const float xb = x + share.dsx, yb = y + share.dsy, zb = z + share.dsz;
        const bool seen_b = zb >= 0.001f;
        const float t = cam.focal_f * __builtin_amdgcn_rcpf(zb);
        const float oxb = __builtin_fmaf(xb, t, cam.cx_f), oyb = __builtin_fmaf(yb, t, cam.cy_f);
        // no bound where the point passes the near plane inside the group, comes closer to it than a thousandth of the focal
        // length (a shift's last bit then moves it by more than the margin), or lands nowhere finite: the whole image
        const float z_safe = 1.0e-3f * cam.focal_f;
        // a camera between the two sits within `dev` of a point of the chord, and a shift that is off by (ex, ey, ez) moves the
        // image position by (F ex - (ox - cx) ez) / (z + ez) exactly: the box grows by that much at the nearest depth it holds
        const float z_near = fminf(z, zb) - share.dev_z;
        const bool loose = (seen != seen_b) | (seen & !(z_near >= z_safe)) | (seen_b & !(z_near >= z_safe)) | (seen_b & !((fabsf(oxb) < 1.0e9f) & (fabsf(oyb) < 1.0e9f)));
        const float fw = (float) (cam.W - 1), fh = (float) (cam.H - 1);
        const float r_near = 1.001f * __builtin_amdgcn_rcpf(z_near);
        const float mx = __builtin_fmaf(__builtin_fmaf(fmaxf(fabsf(ox - cam.cx_f), fabsf(oxb - cam.cx_f)), share.dev_z, cam.focal_f * share.dev_x), r_near, SHARE_MARGIN);
        const float my = __builtin_fmaf(__builtin_fmaf(fmaxf(fabsf(oy - cam.cy_f), fabsf(oyb - cam.cy_f)), share.dev_z, cam.focal_f * share.dev_y), r_near, SHARE_MARGIN);
        float lox = fminf(ox, oxb) - mx, hix = fmaxf(ox, oxb) + mx;
        float loy = fminf(oy, oyb) - my, hiy = fmaxf(oy, oyb) + my;
        ok = seen & seen_b & (hix >= -1.0f) & (lox < fw + 1.0f) & (hiy >= -1.0f) & (loy < fh + 1.0f);
        lox = fmaxf(lox, -1.0f); hix = fminf(hix, fw); loy = fmaxf(loy, -1.0f); hiy = fminf(hiy, fh);
        if (loose) { ok = seen | seen_b; lox = -1.0f; hix = fw; loy = -1.0f; hiy = fh; }
        nwx = (int) floorf(lox); nex = (int) floorf(hix); nwy = (int) floorf(loy); ney = (int) floorf(hiy);
    }

    // the tiles the sub-block reaches: per lane the first and last tile its point matters to (monotone in the corner, so the
    // minimum / maximum over the row are those of the box); lanes that are out take no part
    static_assert((TW & (TW - 1)) == 0 && (TH & (TH - 1)) == 0, "tile sizes are powers of two");
    constexpr int BIG = 1 << 24;
    const int tx0 = max(row_min(ok ? (nwx - 1) >> __builtin_ctz(TW) : BIG), 0), tx1 = min(-row_min(ok ? -((nex + 2) >> __builtin_ctz(TW)) : BIG), tiles_x - 1);
    const int ty0 = max(row_min(ok ? (nwy - 1) >> __builtin_ctz(TH) : BIG), 0), ty1 = min(-row_min(ok ? -((ney + 2) >> __builtin_ctz(TH)) : BIG), tiles_y - 1);
    const int w = tx1 - tx0 + 1, h = ty1 - ty0 + 1;               // uniform over the row; no point in: tx0 = BIG, w < 0
    const bool some = w > 0 && h > 0;
    const int sub = i / kCloudSub, j = lane & (kCloudSub - 1);
    auto list_for = [&](int tx, int ty) {
        const int t = (int) mad_u24((uint32_t) ty, (uint32_t) tiles_x, (uint32_t) tx);
        const int pos = atomicAdd(&tile_count[(uint32_t) t * CNT_STRIDE], 1);
        if (pos < LIST_CAP) cand_lists[(size_t) t * LIST_CAP + pos] = sub;     // beyond: the tile sees count > LIST_CAP and scans
    };
    if (some && w <= 4 && h <= kCloudSub / 4) {
        // the usual box of one to four tiles (at most 4 x 4): lane j of the row takes tile (j & 3, j >> 2) of it -- one
        // atomic per lane, no loop, no division
#if defined(KBE_FRAME_STATS)
        if (j == 0) atomicAdd(&g_frame_stats[1], (unsigned long long) (w * h));
#endif
        if ((j & 3) < w && (j >> 2) < h) {
            owed.t = (int) mad_u24((uint32_t) (ty0 + (j >> 2)), (uint32_t) tiles_x, (uint32_t) (tx0 + (j & 3)));
            owed.pos = atomicAdd(&tile_count[(uint32_t) owed.t * CNT_STRIDE], 1);
        }
    } else if (some) {
        // a larger box: its points scatter.  Beyond BIN_WIDE_FAN tiles it is counted against the frame's budget by the row's
        // first lane (the look first keeps the total from running away once it is spent); beyond the budget the lists are
        // abandoned: every tile scans the blocks itself (k_frame's ranged path).  (All 16 lanes of the row are here together.)
        const int fan = w * h;
        bool listing = true;
        if (fan > BIN_WIDE_FAN) {
            const unsigned budget = bin_budget(tiles_x, tiles_y);
            int go = 0;
            if (j == 0 && __hip_atomic_load(bin_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) <= budget)
                go = atomicAdd(bin_flag, (unsigned) fan) + (unsigned) fan <= budget;
            listing = __shfl(go, lane & ~(kCloudSub - 1)) != 0;
        }
#if defined(KBE_FRAME_STATS)
        if (j == 0 && listing) { atomicAdd(&g_frame_stats[1], (unsigned long long) fan); if (fan > BIN_WIDE_FAN) atomicAdd(&g_frame_stats[7], 1ull); }
#endif
        if (listing)
            for (int k = j; k < fan; k += kCloudSub) { const int r = k / w; list_for(tx0 + (k - r * w), ty0 + r); }
    }
    return owed;
}
__device__ __forceinline__ void place_point_end(const ListSlot& owed, int i, int* cand_lists)
{
    if (owed.t >= 0 && owed.pos < LIST_CAP) cand_lists[(size_t) owed.t * LIST_CAP + owed.pos] = i / kCloudSub;     // beyond: the tile sees count > LIST_CAP and scans
}
__device__ __forceinline__ void place_point(const CloudPoint& p, int i, int lane, const Camera& cam, int tiles_x, int tiles_y, Placement* place,
                                            int* tile_count, int* cand_lists, unsigned* bin_flag, const ShareMode& share)
{
    place_point_end(place_point_begin(p, i, lane, cam, tiles_x, tiles_y, place, tile_count, cand_lists, bin_flag, share), i, cand_lists);
}

__global__ void __launch_bounds__(256) k_place(PlaceJobs jobs)
{
    const PlaceArgs& a = jobs.a[blockIdx.y];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;        // Np is a multiple of 64: whole waves only
    if (i >= jobs.pc.Np) return;
    place_point(jobs.pc.pd[i], i, threadIdx.x & 63, a.cam, jobs.tiles_x, jobs.tiles_y, a.place, a.tile_count, a.cand, a.bin_flag, ShareMode{ 0, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f });
}

// what a pass over candidate blocks does with each point
enum : int { PASS_Z = 1, PASS_COUNT = 2, PASS_INSERT = 4, PASS_SPILL = 8, PASS_COLOUR = 16, PASS_IN_IMAGE = 32 };

#if defined(KBE_FRAME_STOP)      // dev build only (tools/gpu_variant_pmc.sh): the kernel ends after stage KBE_FRAME_STOP, to cost the stages
#define KBE_STOP_AFTER(n) do { if (KBE_FRAME_STOP == (n)) return; } while (0)
#else
#define KBE_STOP_AFTER(n) do { } while (0)
#endif

// a wave-uniform LDS fetch-and-add: one lane's ds_add_rtn_u32, the old value in a scalar for all
__device__ __forceinline__ int lds_add_rtn_uniform(int* counter, int v)
{
    typedef __attribute__((address_space(3))) int* LdsPtr;
    int old = 0;
    if ((threadIdx.x & 63) == 0)
        asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(old) : "v"((uint32_t) (uintptr_t) (LdsPtr) counter), "v"(v) : "memory");
    return __builtin_amdgcn_readfirstlane(old);
}

// what the shared tile machinery (tile_degrid, gather, tile_epilogue: kbe_tiles.h) reads of a frame's arguments
struct TileOut {
    struct { int W, H; } cam;
    uint8_t* frame; float* depth; uint32_t* mask; int* holes; int* hole_count; int4* bbox; uint32_t* coarse;
    float* render; float* existing; float* zee; float* zee_pre;
};

// The arguments are read through a pointer into the kernel-argument segment rather than taken by value: the kernel has three
// phases with almost disjoint needs (splat: lists, placements, colours; resolve: the output planes; slow path: the cloud's
// hierarchy and the whole camera), and held in registers all at once they overflow the scalar file -- the compiler then
// parked sixteen of them in a vector register and fetched them back in every trip of the splat loop.  Making the pointer
// opaque between the phases has each phase load what it needs when it starts.
typedef const __attribute__((address_space(4))) FrameArgs* FrameArgsPtr;
typedef const __attribute__((address_space(4))) PlaceArgs* PlaceArgsPtr;
typedef const __attribute__((address_space(4))) PackedCloud* PackedCloudPtr;

// where in the tile's life its share of the NEXT frames' placements is made (place_ahead below): 0 = at the end, behind the
// epilogue's stores; 1 = behind the splat, in front of the barrier that ends it; 2, 3 = the first KBE_AHEAD_UNITS units of a
// wave up front -- their points requested with the tile's list, placed while the tile's own points are under way, the list
// entries written in front of (2) or behind (3) the splat -- and what is left at the end
#ifndef KBE_AHEAD_AT
#define KBE_AHEAD_AT 3
#endif
#ifndef KBE_AHEAD_UNITS
#define KBE_AHEAD_UNITS 3       // (a wave's share of an equal group is 2.17 units: with three up front nothing is left for the end; 17.9 -> 17.6 us per frame)
#endif
constexpr int AHEAD_UNITS = KBE_AHEAD_AT >= 2 ? KBE_AHEAD_UNITS : 0;
#ifndef KBE_LAZY_COLOURS
#define KBE_LAZY_COLOURS 2      // colours fetched for the records only, behind the splat (see frame_body): 0 never, 1 always,
#endif                          // 2 in the launch for dense clouds (measured: 1024^2 inpaint cloud +3 %, 2048^2 with 16.8 M points -4 %)

__device__ __forceinline__ Camera load_camera(const __attribute__((address_space(4))) Camera* c)
{
    Camera cam;
    cam.focal_f = c->focal_f; cam.fb_f = c->fb_f; cam.fb = c->fb; cam.half_w = c->half_w; cam.half_h = c->half_h;
    cam.cx_f = c->cx_f; cam.cy_f = c->cy_f; cam.fp32_centre = c->fp32_centre; cam.W = c->W; cam.H = c->H;
    cam.has_shift = c->has_shift; cam.sx = c->sx; cam.sy = c->sy; cam.sz = c->sz;
    return cam;
}

// The placements of the frames the NEXT tile launch renders -- k_place's work -- spread over the waves of this launch: row
// blockIdx.y takes frames blockIdx.y, blockIdx.y + gridDim.y, ... of `nx`, its waves the frames' units of 64 points in turn.
// k_place alone is a streaming launch that waits for memory 70 % of its life (the point, then its list slots); k_frame is bound
// by instruction issue: one launch lets the one's waits hide under the other's arithmetic whatever else the chip is doing.
// the mode of frame j of the `n_next` a launch places (shared: the launch's flag)
template <class Jobs> __device__ __forceinline__ ShareMode share_mode(const __attribute__((address_space(4))) Jobs* jp, int j, bool shared)
{
    ShareMode m = { 0, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f };
    if (KBE_SHARED_LISTS && shared) {
        const uint32_t sh = jp->nx_share[j];                            // uniform (a scalar load)
        const int lead = (int) (sh & 255u), last = (int) (sh >> 8);
        if (last > lead) {
            m.mode = j == lead ? 2 : 1;
            if (j == lead) {
                PlaceArgsPtr nx = (PlaceArgsPtr) jp->nx;
                m.dsx = nx[last].cam.sx - nx[lead].cam.sx; m.dsy = nx[last].cam.sy - nx[lead].cam.sy; m.dsz = nx[last].cam.sz - nx[lead].cam.sz;
                m.dev_x = jp->dev[0]; m.dev_y = jp->dev[1]; m.dev_z = jp->dev[2];
            }
        }
    }
    return m;
}

template <class Jobs> __device__ __forceinline__ void place_ahead(const __attribute__((address_space(4))) Jobs* jp, int n_next, int tiles_x, int tiles_y, int wave, int lane, int units_up_front, bool shared)
{
    PackedCloudPtr pcp = &jp->pc;
    PlaceArgsPtr nx = (PlaceArgsPtr) jp->nx;
    const CloudPoint* const pd = pcp->pd;
    const int n_units = pcp->Np / kCloudBlock;
    const int first = blockIdx.x * (TILE_THREADS / 64) + wave, step = gridDim.x * (TILE_THREADS / 64);
    for (int j = blockIdx.y; j < n_next; j += gridDim.y) {             // uniform
        PlaceArgsPtr a = nx + j;
        const Camera cam = load_camera(&a->cam);
        Placement* const place = a->place;
        int* const tile_count = a->tile_count;
        int* const cand = a->cand;
        unsigned* const bin_flag = a->bin_flag;
        const ShareMode share = share_mode(jp, j, shared);
        const int u0 = first + (j == (int) blockIdx.y ? units_up_front * step : 0);      // (the row's first frame: its first units were placed up front)
        if (u0 >= n_units) continue;
        CloudPoint p = *at_offset32(pd, (uint32_t) (u0 * kCloudBlock + lane));
        for (int u = u0; u < n_units; u += step) {                      // wave-uniform; the next unit's point requested before this one is worked on
            const CloudPoint q = p;
            const int un = u + step < n_units ? u + step : u;
            p = *at_offset32(pd, (uint32_t) (un * kCloudBlock + lane));
            place_point(q, u * kCloudBlock + lane, lane, cam, tiles_x, tiles_y, place, tile_count, cand, bin_flag, share);
        }
    }
}

// AHEAD: a launch that also makes placements (k_frame_ahead, k_frame_group_ahead: kernels of their own, so that a launch
// that places nothing carries none of it and profiles tell the two apart)
// UNITS: how many of a wave's units of the next frames' placement are requested with the tile's list and placed up front (the
// rest behind the epilogue): three for a cloud of about a point per pixel (a wave's share of an equal group is 2.2 units), eight
// for one much denser than the raster (configs[4]: 8 units per wave)
// CAP: the records the tile holds in LDS at once (REC_CAP, or LEAN_CAP with a sixth workgroup on the CU: below)
template <int J, bool AHEAD, int UNITS = AHEAD_UNITS, int CAP = REC_CAP>
__device__ __forceinline__ void frame_body(const __attribute__((address_space(4))) FrameJobsT<J>* jp, int job)
{
    constexpr bool LAZY = KBE_LAZY_COLOURS == 1 || (KBE_LAZY_COLOURS == 2 && UNITS > AHEAD_UNITS);
    FrameArgsPtr ap = (FrameArgsPtr) jp->a + job;
    PackedCloudPtr pcp = &jp->pc;
    typedef TileLdsT<CAP> Lds;
    __shared__ FrameLdsT<CAP> F;
    Lds& L = F.T;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tile = xcd_tile_rot(blockIdx.x, gridDim.x, job);
    const int tiles_x = ap->tiles_x, tiles_y = ap->tiles_y;
    const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const int x0 = tx * TW, y0 = ty * TH;
    const int W = ap->cam.W, H = ap->cam.H;
    uint32_t* const zk = (uint32_t*) L.zpre;            // the tile's z-buffer as keys until the splat is complete
    constexpr int WAVES = TILE_THREADS / 64;
#if defined(KBE_FRAME_PROBE)
    unsigned long long probe_t[PROBE_STAMPS] = {};
#endif
    KBE_PROBE(0);
#if defined(KBE_FRAME_PROBE)
    probe_t[13] = __builtin_amdgcn_s_memrealtime();     // 100 MHz: with stamp 9 - stamp 0 the shader clock this wave ran at
#endif

    // The candidate list first (everything else waits for it).  A wave takes four sub-blocks per step -- sixteen lanes each, a
    // lane one point -- and the steps go round the waves; the operands of FOUR steps are requested before the first is
    // worked on (a tile has ~14 steps, a wave three or four of them: usually all its loads are in flight at once, and it
    // waits for memory once instead of once per step -- with one step of look-ahead a step's ~100 instructions could not
    // cover a round trip to L2 / HBM under load).  Every lane reads its own list entries from global memory, and reads them
    // WITHOUT waiting for the count: entries past it are stale ids or whatever the scratch held, so they are clamped to the
    // cloud and the count decides later which steps exist.  None of these loads sits under a branch: a load under a branch
    // makes every later wait a wait for everything (kbe_frame.hip, k_tiles).
#ifndef KBE_SPLAT_DEPTH
#define KBE_SPLAT_DEPTH 4
#endif
    constexpr int DEPTH = KBE_SPLAT_DEPTH, SUBS_PER_STEP = 64 / kCloudSub;
    int* const tile_count = ap->tile_count;
    const int count = tile_count[tile * CNT_STRIDE];
    const bool wide = *ap->bin_flag > bin_budget(tiles_x, tiles_y);
    const int* const my_list = ap->cand + (size_t) tile * LIST_CAP;
    const Placement* const place = ap->place;
    const CloudColour* const colours = pcp->col;
    const uint32_t last_sub = (uint32_t) (pcp->Np / kCloudSub) - 1u;
    float4* const spill = ap->spill + (size_t) tile * BUCKET_STRIDE;
    Placement pl[DEPTH]; CloudColour cc[DEPTH]; int ix[DEPTH], ent[DEPTH];
    auto fetch_entries = [&](int st0) {
#pragma unroll
        for (int d = 0; d < DEPTH; d++) ent[d] = my_list[(uint32_t) min((st0 + d * WAVES) * SUBS_PER_STEP + lane / kCloudSub, LIST_CAP - 1)];
    };
    auto fetch_points = [&]() {
#pragma unroll
        for (int d = 0; d < DEPTH; d++) {
            ix[d] = (int) (min((uint32_t) ent[d], last_sub) * kCloudSub) + (lane & (kCloudSub - 1));
#if KBE_OFFSETS_32
            // (32-bit byte offsets on the launch's uniform bases: as `place[ix]` with a signed index every address was a sign extension,
            // a 64-bit multiply-add and a 64-bit add; the packed cloud holds at most 2^26 points -- cloud_open -- so x 16 fits)
            const uint32_t o16 = (uint32_t) ix[d] << 4;
            pl[d] = *(const Placement*) ((const char*) place + (o16 - (o16 >> 2)));
            if (!LAZY) cc[d] = *(const CloudColour*) ((const char*) colours + o16);
#else
            pl[d] = place[ix[d]];
            if (!LAZY) cc[d] = colours[ix[d]];
#endif
        }
    };
    fetch_entries(wave);
    // the points of this wave's first units of the NEXT frame's placement, requested with the list (unconditionally, the
    // addresses clamped: a row with no frame to place reads the cloud's first block)
    constexpr int NU = AHEAD ? UNITS : 0;
    const int n_next = AHEAD ? jp->n_next : 0;
    // (KBE_SHARED_LISTS) bit 0: this launch's frames read ONE set of lists, the first frame's; bit 1: so will the frames it places
    const int share_flags = KBE_SHARED_LISTS ? jp->pad_ : 0;
    const bool ahead = NU > 0 && (int) blockIdx.y < n_next;             // uniform: this row has a frame to place
    const int a_units = ahead ? pcp->Np / kCloudBlock : 1;
    const int a_first = blockIdx.x * WAVES + wave, a_step = gridDim.x * WAVES;
    CloudPoint a_pt[NU > 0 ? NU : 1];
    ListSlot a_owed[NU > 0 ? NU : 1];
#pragma unroll
    for (int d = 0; d < NU; d++) a_pt[d] = *at_offset32(pcp->pd, (uint32_t) (min(a_first + d * a_step, a_units - 1) * kCloudBlock + lane));
    auto ahead_finish = [&]() {
        if (ahead) {
            int* const cand_next = ((PlaceArgsPtr) jp->nx + blockIdx.y)->cand;
#pragma unroll
            for (int d = 0; d < NU; d++) place_point_end(a_owed[d], (a_first + d * a_step) * kCloudBlock + lane, cand_next);
        }
    };
#ifndef KBE_LDS_WIDE
#define KBE_LDS_WIDE 1
#endif
    // (KBE_LDS_WIDE: the bin heads and the z keys four entries per store -- 294 ds_write_b128 for the workgroup, one per thread and
    // a second one for 38 of them, instead of 1 173 single stores in two loops of three trips with their bound tests: a wave's
    // ~25 vector instructions become ~8, on a launch bound by vector issue)
    constexpr int HEAD4 = (int) (sizeof(L.head) / 16), ZK4 = KH * KW / 4;
    static_assert(!KBE_LDS_WIDE || ((KH * KW) % 4 == 0 && offsetof(Lds, head) % 16 == 0 && offsetof(Lds, zpre) % 16 == 0 && sizeof(L.head) % 16 == 0 &&
                                    HEAD4 <= TILE_THREADS && HEAD4 + ZK4 <= 2 * TILE_THREADS && ZK4 >= TILE_THREADS - HEAD4), "four entries per store");
    if (KBE_LDS_WIDE) {
        const int4 nul4 = make_int4(Lds::kNull, Lds::kNull, Lds::kNull, Lds::kNull);
        const uint4 emp4 = make_uint4(KBE_ZKEY_EMPTY, KBE_ZKEY_EMPTY, KBE_ZKEY_EMPTY, KBE_ZKEY_EMPTY);               // common.py:430
        if (tid < HEAD4) ((int4*) L.head)[tid] = nul4;
        else ((uint4*) zk)[tid - HEAD4] = emp4;
        if (tid < ZK4 - (TILE_THREADS - HEAD4)) ((uint4*) zk)[TILE_THREADS - HEAD4 + tid] = emp4;
    } else {
        lds_reset_heads(L, tid);
        for (int i = tid; i < KH * KW; i += TILE_THREADS) zk[i] = KBE_ZKEY_EMPTY;             // common.py:430
    }
    if (tid == 0) {
        L.nrec = 0;
        F.n_ovf = 0;
        lds_dummy_record(L);
        if (blockIdx.x == 0) *ap->bin_flag_next = 0;
    }
    KBE_PROBE(10);
    fetch_points();                                     // in flight across the barrier
    KBE_PROBE(11);
    // ... and placed while the tile's own points are under way; the list atomics return during the splat
#pragma unroll
    for (int d = 0; d < NU; d++) a_owed[d] = ListSlot{ -1, 0 };
    if (ahead) {
        PlaceArgsPtr na = (PlaceArgsPtr) jp->nx + blockIdx.y;
        const Camera ncam = load_camera(&na->cam);
        const ShareMode nshare = share_mode(jp, blockIdx.y, (share_flags & 2) != 0);
        KBE_PROBE(12);
#pragma unroll
        for (int d = 0; d < NU; d++)
            if (a_first + d * a_step < a_units)
                a_owed[d] = place_point_begin(a_pt[d], (a_first + d * a_step) * kCloudBlock + lane, lane, ncam, tiles_x, tiles_y, na->place, na->tile_count, na->cand, na->bin_flag, nshare);
    }
    const bool listed = !wide & (count <= LIST_CAP);            // uniform
    KBE_PROBE(1);
    __syncthreads();
    KBE_PROBE(2);
    if (KBE_AHEAD_AT == 2) ahead_finish();
    // (only now: every wave of the workgroup has its copy of the count)
    if (tid == 0) {                                     // ready for the next frame's k_place
        const int sharers = (KBE_SHARED_LISTS && (share_flags & 1)) ? (int) (jp->a_share[job] >> 8) : 1;
        if (sharers > 1) {
            // the list is the sub-group's: the last of its frames' workgroups to get here (each has read the count) zeroes it, and
            // the arrivals' own counter next to it
            if (atomicAdd(&tile_count[tile * CNT_STRIDE + 1], 1) == sharers - 1) { tile_count[tile * CNT_STRIDE + 1] = 0; tile_count[tile * CNT_STRIDE] = 0; }
        }
        else tile_count[tile * CNT_STRIDE] = 0;
    }
    KBE_STOP_AFTER(1);                                          // (dev) the list

    // ---- what a tile does with one placed point per lane {ox, oy, dblError}: by `flags` PASS_Z the min-splat of its dblError
    // on the winner corner (:472-506), PASS_COUNT how many of the wave's points become records (noted for candidate `c`),
    // PASS_INSERT its record threaded into the per-pixel lists while there is room (slots >= REC_CAP are dropped or, with
    // PASS_SPILL, go to the tile's spill area), PASS_COLOUR: the record's colours are `col` (else: the point's index waits
    // in their place until they are fetched).
    // (slow path) records each candidate block of a window contributes, then their prefix sums: in the tile's spill area, which
    // only the normal path's further rounds use -- 1 KB less LDS is what lets a fifth workgroup onto the CU
    int* const slow_cnt = (int*) spill;
    // (normal path) a record that finds no room in LDS waits in the spill area as its POINT INDEX (4 bytes; until round 5: its
    // 16-byte record {ox, oy, dblError, index}): the round that takes it reads the point's placement again, which the tile has just
    // pulled through the L2 -- 4 + 4 instead of 16 + 16 bytes of HBM traffic per spilled record, and the same area holds four times
    // as many (SPILL_CAP: 48 records per pixel of the tile)
    int* const spill_idx = (int*) spill;
    auto placed_point = [&](int flags, float ox, float oy, float err, bool ok, int idx, int c, const float4& col) {
        Proj p;
        p.nwx = (int) floorf(ox); p.nwy = (int) floorf(oy);
        const int rx = p.nwx - (x0 - 2), ry = p.nwy - (y0 - 2);
        // north-west corner within [x0 - 2, x0 + TW] x [y0 - 2, y0 + TH]: its winner corner can be a pixel of tile + halo
        const bool in_z = ok & ((unsigned) rx <= (unsigned) (TW + 2)) & ((unsigned) ry <= (unsigned) (TH + 2));
        // ... within [x0 - 1, x0 + TW - 1] x [y0 - 1, y0 + TH - 1] and touching the image: it can colour a tile pixel
        // (PASS_IN_IMAGE: a PLACED point -- k_place kept its position only if a corner lies inside the image and wrote PLACE_NONE
        // otherwise, which fails the range test: the two image tests are those of place_point_begin over again)
        const bool in_r = ok & ((unsigned) (rx - 1) <= (unsigned) TW) & ((unsigned) (ry - 1) <= (unsigned) TH) &
                          ((flags & PASS_IN_IMAGE) ? true : ((unsigned) (p.nwx + 1) <= (unsigned) W) & ((unsigned) (p.nwy + 1) <= (unsigned) H));
#if defined(KBE_FRAME_STATS)
        { const unsigned long long mz = __ballot(in_z); if (lane == 0 && (flags & PASS_Z)) atomicAdd(&g_frame_stats[3], (unsigned long long) __popcll(mz)); }
#endif
        if (in_z && (flags & PASS_Z)) {
            project_weights(ox, oy, p);
            const int k = winner_corner_finite(p);                                  // common.py:486-506 (ox, oy are finite)
            const int cx = p.nwx + (k & 1), cy = p.nwy + (k >> 1);
            const int lx = cx - (x0 - 1), ly = cy - (y0 - 1);
            // (byte offset from a 24-bit multiply-add: written on the element index the compiler folded the x 4 into a
            // quarter-rate 32-bit multiply)
            if (inside(cx, cy, W, H) & ((unsigned) lx < (unsigned) KW) & ((unsigned) ly < (unsigned) KH))
                atomicMin((uint32_t*) ((char*) zk + mad_u24((uint32_t) ly, KW * 4u, (uint32_t) lx << 2)), zkey_encode(err));
        }
        if (flags & (PASS_COUNT | PASS_INSERT)) {
            const unsigned long long m = __ballot(in_r);
            const int n_r = __popcll(m);
            if ((flags & PASS_COUNT) && lane == 0) slow_cnt[c] = n_r;
            if ((flags & PASS_INSERT) && m) {                   // wave-uniform
                // ONE LDS atomic for the wave (written as the instruction: around `if (lane == 0) atomicAdd(..)` the compiler's
                // atomic optimizer builds a dozen instructions of lane counting for a case that cannot occur here)
                int base = lds_add_rtn_uniform(&L.nrec, n_r);
                const int limit = CAP;
                // (the lanes of `m` below this one: v_mbcnt_lo / _hi on the ballot -- two instructions; written as a population count of
                // m & lanes-below the compiler makes two ands and two counts of it)
                const int slot = base + (int) __builtin_amdgcn_mbcnt_hi((uint32_t) (m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) m, 0u));
                if (in_r && slot < limit) {
                    const int next = atomicExch((int*) ((char*) L.head + mad_u24((uint32_t) (ry - 1), BW * 4u, (uint32_t) (rx - 1) << 2)), slot << 4);
                    L.rec[slot] = make_float4(ox, oy, err, __int_as_float(next));
                    if (flags & PASS_COLOUR) L.rgbd[slot] = col;
                    else L.rgbd[slot].x = __int_as_float(idx);                      // the point, until its colours arrive
                }
                if ((flags & PASS_SPILL) && base + n_r > limit) {                   // wave-uniform; a few tiles in a hundred
                    const bool sp = in_r && slot >= limit;
                    const unsigned long long ms = __ballot(sp);
                    int sbase = 0;
                    if (lane == 0) sbase = atomicAdd(&F.n_ovf, __popcll(ms));
                    sbase = __builtin_amdgcn_readfirstlane(sbase);
                    const int o = sbase + (int) __builtin_amdgcn_mbcnt_hi((uint32_t) (ms >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) ms, 0u));
                    if (sp && o < SPILL_CAP) spill_idx[o] = idx;
                }
            }
        }
    };

#ifndef KBE_PLACED_IN_IMAGE
#define KBE_PLACED_IN_IMAGE 1
#endif
    constexpr int KBE_PASS_PLACED = KBE_PLACED_IN_IMAGE ? PASS_IN_IMAGE : 0;
    // ---- the normal path: the points of the listed sub-blocks, their placements and colours requested above (a 96- and a
    // 128-bit load per point).  About a third of the points are near misses that belong to a neighbouring tile: they cost
    // a floor and a range test.  Records beyond REC_CAP (a few tiles in a hundred: two surfaces over one another at a depth
    // edge) spill into the tile's own area of the scratch in HBM and are gathered in further rounds.  A tile with more
    // than 64 candidate sub-blocks takes further trips of four steps per wave.
    const int n_cand = listed ? count : 0;
    {
        const int n_steps = (n_cand + SUBS_PER_STEP - 1) / SUBS_PER_STEP;
        for (int st0 = wave; st0 < n_steps; st0 += DEPTH * WAVES) {    // wave-uniform
            const bool more = st0 + DEPTH * WAVES < n_steps;
            if (more) fetch_entries(st0 + DEPTH * WAVES);       // the next trip's entries while this trip's points are worked on
#pragma unroll
            for (int d = 0; d < DEPTH; d++)
                if (st0 + d * WAVES < n_steps) {
                    const bool valid = (st0 + d * WAVES) * SUBS_PER_STEP + lane / kCloudSub < n_cand;
                    if (LAZY) placed_point(PASS_Z | PASS_INSERT | PASS_SPILL | KBE_PASS_PLACED, pl[d].ox, pl[d].oy, pl[d].err, valid, ix[d], 0, make_float4(0.0f, 0.0f, 0.0f, 0.0f));
                    else placed_point(PASS_Z | PASS_INSERT | PASS_SPILL | PASS_COLOUR | KBE_PASS_PLACED, pl[d].ox, pl[d].oy, pl[d].err, valid, ix[d], 0, make_float4(cc[d].r, cc[d].g, cc[d].b, cc[d].depth));
                }
            if (more) fetch_points();
        }
    }
    if (AHEAD && KBE_AHEAD_AT == 1 && n_next > 0) place_ahead(jp, n_next, tiles_x, tiles_y, wave, lane, NU, (share_flags & 2) != 0);
    KBE_PROBE(3);
    if (KBE_AHEAD_AT == 3) ahead_finish();
    __syncthreads();
    KBE_PROBE(4);
    KBE_STOP_AFTER(3);                                          // (dev) + the splat

    // ---- the second phase reads its arguments now
    // (the degrid and the gather need the frame's size and the two optional z planes; the planes the epilogue stores to are read
    // in front of it: loaded here, these sixteen scalars were parked in the lanes of a vector register across the degrid and the
    // gather -- a v_writelane each and a v_readlane to fetch it back, on a kernel bound by vector issue)
    asm volatile("" : "+s"(ap) :: "memory");
    TileOut a;
    a.cam.W = W; a.cam.H = H;
    a.zee = ap->zee; a.zee_pre = ap->zee_pre;
#if defined(KBE_LATE_ARGS) && !KBE_LATE_ARGS
    a.frame = ap->frame; a.depth = ap->depth; a.mask = ap->mask; a.holes = ap->holes; a.hole_count = ap->hole_count; a.bbox = ap->bbox; a.coarse = ap->coarse;
    a.render = ap->render; a.existing = ap->existing;
#endif

    constexpr int ZPER = (KH * KW + TILE_THREADS - 1) / TILE_THREADS;
    constexpr int PER = (CAP + TILE_THREADS - 1) / TILE_THREADS;
    // keys -> floats in place (a pixel outside the image was never splatted: it reads 1e6 like common.py:430), and the
    // one decision per tile whether the fp32-only degrid and z test apply
    auto decode_z = [&]() {
        bool band = true;
        if (KBE_LDS_WIDE) {
            // four keys per thread (the first 153 threads); the band test on the keys, which order as the floats do: smallest and
            // largest of the four against the keys of the band's ends
            if (tid < ZK4) {
                const uint4 k = ((const uint4*) zk)[tid];
                ((float4*) L.zpre)[tid] = make_float4(zkey_decode(k.x), zkey_decode(k.y), zkey_decode(k.z), zkey_decode(k.w));
                const uint32_t lo = min(min(k.x, k.y), min(k.z, k.w)), hi = max(max(k.x, k.y), max(k.z, k.w));
                band = (lo >= zkey_encode(524288.0f)) & (hi <= zkey_encode(1000000.0f));          // degrid_fast_ok of all four
            }
        } else {
#pragma unroll
        for (int u = 0; u < ZPER; u++) {
            const int i = tid + u * TILE_THREADS;
            if (i < KH * KW) {
                const float z = zkey_decode(zk[i]);
                L.zpre[i] = z;
                band = band && degrid_fast_ok(z);
            }
        }
        }
        const unsigned long long odd = __ballot(!band);
        if (lane == 0) L.odd_z[tid >> 6] = odd != 0ull;
    };
    auto tile_is_fast = [&]() { return lds_tile_is_fast(L); };
    auto fetch_rgbd = [&](int id) {
        const CloudColour c = colours[id];
        return make_float4(c.r, c.g, c.b, c.depth);
    };

    PixAcc acc[PIX_PER_THREAD];
#pragma unroll
    for (int m = 0; m < PIX_PER_THREAD; m++) { acc[m].rg = (f2) (0.0f); acc[m].bd = (f2) (0.0f); acc[m].w = 0.0f; }

    const int total = L.nrec;
    bool fast;
    const int n_spill = F.n_ovf;
    if (listed && n_spill <= SPILL_CAP) {
        // ---- the first REC_CAP records are in LDS with their colours
        // ... or (LAZY: the launch for dense clouds, which is as close to the memory's limit as to the issue rate's) with their
        // points: the colours of the RECORDS only -- the near misses of the candidate list, a third of it, never fetch
        // theirs -- requested here and stored behind the degrid, which needs none of them
        const int n_held = min(total, CAP);
        float4 lc[PER];
        if (LAZY) {
#pragma unroll
            for (int u = 0; u < PER; u++) {
                const int i = tid + u * TILE_THREADS;
                if (i < n_held) lc[u] = fetch_rgbd(__float_as_int(L.rgbd[i].x));
            }
        }
        decode_z();
        __syncthreads();
        fast = tile_is_fast();
        tile_degrid(a, L, tid, x0, y0, fast);
        if (LAZY) {
#pragma unroll
            for (int u = 0; u < PER; u++) {
                const int i = tid + u * TILE_THREADS;
                if (i < n_held) L.rgbd[i] = lc[u];
            }
        }
        KBE_PROBE(5);
        __syncthreads();                                        // the epilogue stages its bytes where the degrid still reads its neighbours' z
        KBE_PROBE(6);
        KBE_STOP_AFTER(4);                                      // (dev) + degrid
        if (fast) gather<true>(a, L, tid, x0, y0, acc);
        else gather<false>(a, L, tid, x0, y0, acc);
        KBE_PROBE(7);
        KBE_STOP_AFTER(5);                                      // (dev) + gather
        // further rounds: the spilled records, REC_CAP at a time (already projected: only lists, colours and the walk)
        for (int r0 = 0; r0 < n_spill; r0 += CAP) {             // uniform
            const int n = min(CAP, n_spill - r0);
            __syncthreads();                                    // the previous gather is done with the lists
            lds_reset_heads(L, tid);
            float4 rr[PER], cc[PER];
#pragma unroll
            for (int u = 0; u < PER; u++) {
                const int i = tid + u * TILE_THREADS;
                const int id = i < n ? spill_idx[r0 + i] : 0;
                const Placement q = *at_offset32(place, (uint32_t) id);
                rr[u] = make_float4(q.ox, q.oy, q.err, 0.0f);
                cc[u] = fetch_rgbd(id);
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
        // ---- the slow path, a small uniform state machine around ONE copy of an exact pass over blocks: the tile scans ALL
        // blocks of the cloud in windows of MAXC, each block's node tested inline: the z-splat, window by window (a tile
        // that got here from its list does it again: min is idempotent); degrid; then per window the record counts, their
        // prefix sums, and runs of at most REC_CAP records: insert, colours, gather.
        const FrameArgs* const gp = (const FrameArgs*) ap;      // (struct copies want a generic pointer)
        const Camera cam = gp->cam;
        const PackedCloud pc = *(const PackedCloud*) pcp;
        const int n_blocks = pc.count[0];
        CullView q;
        q.g = cam.focal_f / pc.fd;
        q.sx = cam.has_shift ? cam.sx : 0.0f; q.sy = cam.has_shift ? cam.sy : 0.0f; q.sz = cam.has_shift ? cam.sz : 0.0f;
        q.Sx = q.sx * pc.fd; q.Sy = q.sy * pc.fd;
        q.focal = cam.focal_f;
        q.rx0 = (float) (x0 - 2) - cam.cx_f; q.rx1 = (float) (x0 + TW + 1) - cam.cx_f;
        q.ry0 = (float) (y0 - 2) - cam.cy_f; q.ry1 = (float) (y0 + TH + 1) - cam.cy_f;
        const float4 no_colour = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        auto load_block = [&](int b, float& x, float& y, float& z) {
            const CloudPoint p = pc.pd[(b < 0 ? 0 : b) * kCloudBlock + lane];
            x = p.x; y = p.y; z = p.z;
        };
        // one exact pass over candidates [c0, c1) of the window of 64-point blocks starting at block `wbase`: shift
        // (common.py:104-109), projection (:447-468), dblError (:470), then as a placed point.  A wave takes every fourth
        // candidate; the coordinates of its next block are loaded before it works on the current one.
        auto pass = [&](int flags, int c0, int c1, int wbase) {
            auto block_of = [&](int c) -> int {                 // wave-uniform
                if (c >= c1) return -1;
                const int b = wbase + c;
                return (b < n_blocks && node_hits(pc.level[0][b], q)) ? b : -1;
            };
            int c = c0 + wave;
            int b_next = block_of(c);
            float xn, yn, zn;
            load_block(b_next, xn, yn, zn);
            for (; c < c1; c += WAVES) {                        // wave-uniform
                const int b = b_next;
                float x = xn, y = yn, z = zn;
                b_next = block_of(c + WAVES);
                load_block(b_next, xn, yn, zn);
                if (b < 0) { if ((flags & PASS_COUNT) && lane == 0) slow_cnt[c] = 0; continue; }
                float ox = 0.0f, oy = 0.0f;
                apply_shift(cam, x, y, z);
                const bool ok = project_xy(cam, x, y, z, ox, oy);
                placed_point(flags, ox, oy, project_err_fast(cam, ok ? z : 1024.0f), ok, b * kCloudBlock + lane, c, no_colour);
            }
        };
        enum { S_ZWIN, S_DEGRID, S_COUNT, S_SCAN, S_RUN, S_DONE };
        int state = S_ZWIN, wb = 0, c0 = 0, done = 0;
#if defined(KBE_FRAME_STOP) && defined(KBE_FRAME_SKIP_SLOW)      // (dev) what would the launch cost without its slow tiles?
        state = S_DONE;
#endif
        fast = false;
        while (state != S_DONE) {                               // uniform
            const int n_win = min(MAXC, n_blocks - wb);
            int flags = 0, p0 = 0, p1 = n_win;
            if (state == S_ZWIN) flags = PASS_Z;
            else if (state == S_COUNT) flags = PASS_COUNT;
            else if (state == S_RUN) {
                lds_reset_heads(L, tid);
                if (tid == 0) { L.nrec = 0; F.run_end = n_win; }
                __syncthreads();
                // the run ends in front of the first candidate whose prefix sum exceeds done + REC_CAP
                for (int c = c0 + tid; c < n_win; c += TILE_THREADS)
                    if (slow_cnt[c] - done > CAP && (c == c0 || slow_cnt[c - 1] - done <= CAP)) F.run_end = c;
                __syncthreads();
                flags = PASS_INSERT; p0 = c0; p1 = F.run_end;
            }
            if (flags) pass(flags, p0, p1, wb);
            __syncthreads();
            if (state == S_ZWIN) {
                wb += MAXC;
                if (wb >= n_blocks) state = S_DEGRID;
            } else if (state == S_DEGRID) {
                decode_z();
                __syncthreads();
                fast = tile_is_fast();
                tile_degrid(a, L, tid, x0, y0, fast);
                wb = 0;
                state = S_COUNT;
            } else if (state == S_COUNT) {
                state = S_SCAN;
            } else if (state == S_SCAN) {
                // inclusive prefix sums of the counts, one entry per thread
                const int e = tid < n_win ? slow_cnt[tid] : 0;
                int v = e;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) { const int t = __shfl_up(v, off); if (lane >= off) v += t; }
                if (lane == 63) F.wave_sum[tid >> 6] = v;
                __syncthreads();
                for (int w = 0; w < (tid >> 6); w++) v += F.wave_sum[w];
                if (tid < n_win) slow_cnt[tid] = v;
                c0 = 0; done = 0;
                state = n_win > 0 ? S_RUN : S_DONE;
                if (state == S_DONE && wb + MAXC < n_blocks) { wb += MAXC; state = S_COUNT; }
            } else if (state == S_RUN) {
                const int n = min(L.nrec, CAP);
                for (int i = tid; i < n; i += TILE_THREADS) L.rgbd[i] = fetch_rgbd(__float_as_int(L.rgbd[i].x));
                __syncthreads();
                if (fast) gather<true>(a, L, tid, x0, y0, acc);
                else gather<false>(a, L, tid, x0, y0, acc);
                c0 = p1;
                done = c0 > 0 ? slow_cnt[c0 - 1] : 0;
                if (c0 >= n_win) {
                    state = S_DONE;
                    if (wb + MAXC < n_blocks) { wb += MAXC; state = S_COUNT; }
                }
            }
            __syncthreads();
        }
    }
#if defined(KBE_FRAME_STATS)
    if (tid == 0) {
        atomicAdd(&g_frame_stats[0], 1ull);
        atomicAdd(&g_frame_stats[2], (unsigned long long) n_cand);
        atomicAdd(&g_frame_stats[4], (unsigned long long) total);
        atomicAdd(&g_frame_stats[5], (unsigned long long) !(listed && n_spill <= SPILL_CAP));
        atomicAdd(&g_frame_stats[6], (unsigned long long) (n_spill > 0));
    }
#else
    (void) total;
#endif
#if !defined(KBE_LATE_ARGS) || KBE_LATE_ARGS
    asm volatile("" : "+s"(ap) :: "memory");
    a.frame = ap->frame; a.depth = ap->depth; a.mask = ap->mask; a.holes = ap->holes; a.hole_count = ap->hole_count; a.bbox = ap->bbox; a.coarse = ap->coarse;
    a.render = ap->render; a.existing = ap->existing;
#endif
    tile_epilogue(a, L, acc, tile, x0, y0);
    KBE_PROBE(8);
    if (AHEAD && KBE_AHEAD_AT != 1) {
        // what the waves did not place up front: further units of the row's frame, further frames (groups that grow)
        asm volatile("" : "+s"(jp) :: "memory");
        const int n_left = jp->n_next;
        if (n_left > 0) place_ahead(jp, n_left, jp->a[job].tiles_x, jp->a[job].tiles_y, wave, lane, NU, KBE_SHARED_LISTS && (jp->pad_ & 2) != 0);
    }
    KBE_PROBE(9);
#if defined(KBE_FRAME_PROBE)
    probe_t[13] = __builtin_amdgcn_s_memrealtime() - probe_t[13];
    {
        const unsigned w = (blockIdx.y * gridDim.x + blockIdx.x) * WAVES + wave;
        if (lane < PROBE_STAMPS && w < (unsigned) PROBE_WAVES) {
            unsigned long long v = 0;
#pragma unroll
            for (int k = 0; k < PROBE_STAMPS; k++) if (lane == k) v = probe_t[k];
            g_frame_probe[(size_t) w * PROBE_STAMPS + lane] = v;
        }
    }
#endif
}

typedef FrameJobsT<1> FrameJob1;
typedef FrameJobsT<KBE_FRAME_JOBS> FrameJobs;

// Two builds of every tile launch.  ROOMY: REC_CAP records per tile, five workgroups per CU (30.4 KB of LDS each; 96 registers per
// lane, a few of the gather's spilled) -- with the tile's list read straight into registers the LDS allows it, and a fifth wave per
// SIMD covers more of the others' waits than the spills cost.  LEAN (round 4's last step; the names without suffix, what a cloud of
// about a point per pixel takes): LEAN_CAP = 608 records, 26.3 KB -- a SIXTH workgroup per CU at 80 registers per lane (32 spilled).
// Once the launch's time had become its instruction count, a sixth wave per SIMD paid where it had not
// before: 15.7 -> 15.0 us per frame at twelve frames per launch on the bench cloud (1.08 points per pixel: a tile reaches 608
// records on average, so about half of them send a few records through the spill area and take a second round -- and still);
// a raw cloud (1.0) 14.4 -> 13.4; caps of 576 / 624 measure 15.4 / 15.1, 636 loses the sixth slot (16.1).  Clouds denser than
// LEAN_MAX_DENSITY points per pixel keep the roomy build (16.8 M points on 2048^2: 287 us per frame roomy, 295 lean).
#ifndef KBE_FRAME_WAVES
#define KBE_FRAME_WAVES 5
#endif
#ifndef KBE_LEAN_CAP
#define KBE_LEAN_CAP 608
#endif
#ifndef KBE_LEAN_WAVES
#define KBE_LEAN_WAVES 6
#endif
#ifndef KBE_LEAN_MAX_DENSITY
#define KBE_LEAN_MAX_DENSITY 1.125
#endif
constexpr int LEAN_CAP = KBE_LEAN_CAP;
static_assert(LEAN_CAP >= TILE_THREADS && LEAN_CAP <= REC_CAP, "the lean build holds fewer records than the scratch's tiles are laid out for");
#define KBE_FRAME_ATTR amdgpu_waves_per_eu(KBE_FRAME_WAVES, KBE_FRAME_WAVES)
#define KBE_LEAN_ATTR amdgpu_waves_per_eu(KBE_LEAN_WAVES, KBE_LEAN_WAVES)
#define KBE_ARGS(T) ((const __attribute__((address_space(4))) T*) __builtin_amdgcn_kernarg_segment_ptr())      // the one argument, at offset 0

__global__ void __launch_bounds__(TILE_THREADS) __attribute__((KBE_LEAN_ATTR)) k_frame(FrameJob1) { frame_body<1, false, AHEAD_UNITS, LEAN_CAP>(KBE_ARGS(FrameJob1), 0); }
__global__ void __launch_bounds__(TILE_THREADS) __attribute__((KBE_FRAME_ATTR)) k_frame_roomy(FrameJob1) { frame_body<1, false>(KBE_ARGS(FrameJob1), 0); }
// AHEAD: ... that also make the placements of the frames the next tile launch renders
__global__ void __launch_bounds__(TILE_THREADS) __attribute__((KBE_LEAN_ATTR)) k_frame_ahead(FrameJob1) { frame_body<1, true, AHEAD_UNITS, LEAN_CAP>(KBE_ARGS(FrameJob1), 0); }
__global__ void __launch_bounds__(TILE_THREADS) __attribute__((KBE_FRAME_ATTR)) k_frame_ahead_roomy(FrameJob1) { frame_body<1, true>(KBE_ARGS(FrameJob1), 0); }
// several frames of the same cloud and size per launch (blockIdx.y = the frame), as the bucket route's grouped launches
__global__ void __launch_bounds__(TILE_THREADS) __attribute__((KBE_LEAN_ATTR)) k_frame_group(FrameJobs) { frame_body<KBE_FRAME_JOBS, false, AHEAD_UNITS, LEAN_CAP>(KBE_ARGS(FrameJobs), blockIdx.y); }
__global__ void __launch_bounds__(TILE_THREADS) __attribute__((KBE_FRAME_ATTR)) k_frame_group_roomy(FrameJobs) { frame_body<KBE_FRAME_JOBS, false>(KBE_ARGS(FrameJobs), blockIdx.y); }
__global__ void __launch_bounds__(TILE_THREADS) __attribute__((KBE_LEAN_ATTR)) k_frame_group_ahead(FrameJobs) { frame_body<KBE_FRAME_JOBS, true, AHEAD_UNITS, LEAN_CAP>(KBE_ARGS(FrameJobs), blockIdx.y); }
__global__ void __launch_bounds__(TILE_THREADS) __attribute__((KBE_FRAME_ATTR)) k_frame_group_ahead_roomy(FrameJobs) { frame_body<KBE_FRAME_JOBS, true>(KBE_ARGS(FrameJobs), blockIdx.y); }

// ... for a cloud much denser than the raster: eight units of a wave's placements up front (configs[4], 16.8 M points on a 2048^2
// raster: 8 units per wave; the placement launch of its own that such a cloud used to keep waits for memory 70 % of its life --
// 183 us per frame next to a tile launch of 208 -- and as part of the tile launch it hides: 401 -> 349 us per frame left in HBM)
constexpr int AHEAD_UNITS_DENSE = 8;
__global__ void __launch_bounds__(TILE_THREADS) __attribute__((KBE_FRAME_ATTR)) k_frame_group_ahead_dense(FrameJobs)
{
    frame_body<KBE_FRAME_JOBS, true, AHEAD_UNITS_DENSE>(KBE_ARGS(FrameJobs), blockIdx.y);
}

// (Round 6 built a tile launch WITHOUT records here -- two passes over the candidate list, the z-tested sums added to accumulator planes in
// LDS by ds_add_f32 instead of the register gather; commit 19f9093 has it.  Parity-green and 6.5 x slower: the LDS
// executes fp32 atomic adds about one lane at a time, 105.7 against 15.0 us per frame -- profiles/r06_lds_float_atomics.txt.)

}  // namespace

namespace kbe {
// Can the tile launch of n frames make the placements of n_next frames without outliving its own work?  Its waves share them:
// up to a few units of 64 points per wave (beyond AHEAD_UNITS per wave the launch is k_frame_group_ahead_dense).
#ifndef KBE_AHEAD_MAX_UNITS
#define KBE_AHEAD_MAX_UNITS 9
#endif
static size_t ahead_units_per_wave(int N, int W, int H, int n, int n_next)        // rounded up
{
    const size_t units = (size_t) cloud_layout_base(N).Np / kCloudBlock * (size_t) n_next;
    const size_t waves = (size_t) ((W + TW - 1) / TW) * ((H + TH - 1) / TH) * (TILE_THREADS / 64) * (size_t) n;
    return (units + waves - 1) / waves;
}
bool fused_can_place_ahead(int N, int W, int H, int n, int n_next)
{
    if (n < 1 || n_next < 1) return false;
    return ahead_units_per_wave(N, W, H, n, n_next) <= (size_t) KBE_AHEAD_MAX_UNITS;
}

// the scatter of n <= KBE_FRAME_JOBS frames of the same packed cloud and frame size: one placement launch (unless the frames are
// `placed`: the previous tile launch made their placements ahead) and one tile launch, each taking all n frames (frame k: its
// camera, scratch set, and by `parity` / `turn` the set's bank of placements and lists, hole counter and list total); the tile
// launch also makes the placements of `next` (n_next frames: the ones the next tile launch on this stream renders)
void launch_frames_fused(hipStream_t s, int n, const void* packed, int N, double cloud_focal, const FusedTarget* t, bool placed, int n_next, const FusedTarget* next, int build,
                         double near_depth)
{
    PlaceJobs pj;
    FrameJobs fj;
    const PackedCloud pc = cloud_open(packed, N, cloud_focal);
    unsigned n_tiles = 0;
    auto bank_of = [](const FusedTarget& f) { return f.parity == 1 ? 1 : 0; };
    // the set's list totals: frames with placements in front of them alternate between two (read [par], zero [par ^ 1] for the
    // next); in a sequence that places ahead the frame of turn k reads [k % 3], the placement for turn k + 1 counts in
    // [(k + 1) % 3] during the same launch, and [(k + 2) % 3] -- read last by turn k - 1 -- is zeroed
    auto flag_of = [](const FusedTarget& f) { return 2 + (f.turn >= 0 ? f.turn % 3 : (f.parity == 1 ? 1 : 0)); };
    auto flag_zeroed_by = [](const FusedTarget& f) { return 2 + (f.turn >= 0 ? (f.turn + 2) % 3 : (f.parity == 1 ? 0 : 1)); };
    auto place_args = [&](const FusedTarget& f) {
        PlaceArgs b;
        b.cam = f.cam; b.place = (Placement*) bank_place(f.place, N, bank_of(f)); b.tile_count = bank_tile_count(f.sc, bank_of(f)); b.cand = bank_cand(f.sc, bank_of(f));
        b.bin_flag = (unsigned*) f.sc.hole_count + flag_of(f);
        return b;
    };
    // (KBE_SHARED_LISTS) which consecutive frames of a group placed ahead share ONE set of candidate lists (FrameJobsT, ShareMode).
    // The group's cameras must differ in their shifts only.  The frames are cut into sub-groups of s consecutive ones -- the
    // largest s of 12, 8, 6, 4, 3, 2 for which the nearest point the caller knows of (`near_depth`: objectDepthrange's closest
    // depth, common.py:88) moves by at most KBE_SHARE_MAX_PX pixels between a sub-group's first and last camera: a sub-block is
    // listed for the box of its corners under those two cameras, and what the box gains in tiles must stay below what ONE list
    // for s frames saves (measured).  A camera between the two need not lie on the straight line between
    // them (a Ken Burns path is a parabola in shift space: shiftX = dU closestDepth(step) / F, common.py:88-100): how far the
    // sub-groups' cameras stray from their chords goes to the kernel as `dev` and widens the boxes.  Decided from the group's
    // cameras, the cloud and near_depth alone, so that the launch that places a group and the launch that renders it agree
    // without being told.
    struct SharePlan { bool any; float dev[3]; uint8_t lead[KBE_FRAME_JOBS], last[KBE_FRAME_JOBS], size[KBE_FRAME_JOBS]; };
    auto share_plan = [&](const FusedTarget* g, int m) {
        SharePlan P = {};
        for (int k = 0; k < KBE_FRAME_JOBS; k++) { P.lead[k] = P.last[k] = (uint8_t) k; P.size[k] = 1; }
        if (!KBE_SHARED_LISTS || m < 2 || !(near_depth > 0.0)) return P;
        // a sub-group's list is longer than a frame's own, and a list beyond LIST_CAP sends its tile down the slow path: only
        // clouds whose AVERAGE list (1.55 candidates per point of the tile's share, in sub-blocks) leaves a factor of four to
        // the capacity share (the bench cloud: 54 of 2048; 16.8 M points on 2048^2: 198, its densest tiles 480-500 -- with
        // lists of 512, until round 5, shared lists reached 515-555 there and eighteen tiles of a video scanned the whole cloud;
        // with 2048 such a cloud may share: measured the same with and without, 288.7 us per frame)
        const Scratch& sc = g[0].sc;
        if (1.55 * (double) pc.Np / kCloudSub / ((double) sc.tiles_x * sc.tiles_y) > LIST_CAP / 4.0) return P;
        const Camera& c0 = g[0].cam;
        double big = 1.0;
        for (int k = 0; k < m; k++) {
            const Camera& c = g[k].cam;
            if (c.focal_f != c0.focal_f || c.fb != c0.fb || c.half_w != c0.half_w || c.half_h != c0.half_h || c.W != c0.W || c.H != c0.H ||
                c.fp32_centre != c0.fp32_centre || !c.has_shift || !c0.fp32_centre) return P;
            big = fmax(big, fmax(fabs((double) c.sx), fmax(fabs((double) c.sy), fabs((double) c.sz))));
        }
        if (big > 100.0) return P;
        const double F = (double) c0.focal_f, half = 0.5 * (double) (c0.W > c0.H ? c0.W : c0.H);
        auto spread_px = [&](int a, int b) {            // how far the nearest point moves between cameras a and b, in pixels
            const Camera& ca = g[a].cam; const Camera& cb = g[b].cam;
            const double zn = near_depth + fmin((double) ca.sz, (double) cb.sz);
            if (!(zn > 0.01 * F)) return 1.0e30;
            return (hypot((double) cb.sx - ca.sx, (double) cb.sy - ca.sy) * F + half * fabs((double) cb.sz - ca.sz)) / zn;
        };
        static const int sizes[] = { 12, 8, 6, 4 };         // (sub-groups of 2 or 3 measured slower than lists of their own: 17.4-17.7 against 16.8-17.0 us per frame)
        int s_sub = 0;
        for (int q = 0; q < 4 && !s_sub; q++) {
            const int sz = sizes[q] < m ? sizes[q] : m;
            if (sz < 4) break;                              // (a group of two or three frames: lists of their own)
            bool fits = true;
            for (int a0 = 0; a0 < m && fits; a0 += sz) { const int b0 = (a0 + sz < m ? a0 + sz : m) - 1; fits = b0 == a0 || spread_px(a0, b0) <= (double) KBE_SHARE_MAX_PX; }
            if (fits) s_sub = sz;
        }
        if (!s_sub) return P;
        // how far the cameras of a sub-group stray from its chord, per axis (+ the shifts' own fp32 rounding)
        double dev[3] = { 0.0, 0.0, 0.0 };
        for (int a0 = 0; a0 < m; a0 += s_sub) {
            const int b0 = (a0 + s_sub < m ? a0 + s_sub : m) - 1;
            const Camera& ca = g[a0].cam; const Camera& cb = g[b0].cam;
            const double d[3] = { (double) cb.sx - ca.sx, (double) cb.sy - ca.sy, (double) cb.sz - ca.sz };
            const double dd = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
            for (int k = a0 + 1; k < b0; k++) {
                const Camera& c = g[k].cam;
                const double e[3] = { (double) c.sx - ca.sx, (double) c.sy - ca.sy, (double) c.sz - ca.sz };
                double lam = dd > 0.0 ? (e[0] * d[0] + e[1] * d[1] + e[2] * d[2]) / dd : 0.0;
                lam = lam < 0.0 ? 0.0 : (lam > 1.0 ? 1.0 : lam);
                for (int q = 0; q < 3; q++) dev[q] = fmax(dev[q], fabs(e[q] - lam * d[q]));
            }
        }
        for (int q = 0; q < 3; q++) dev[q] += 2.0e-6 * big;
        // (a path that strays from its chords by more than a pixel's worth at the nearest depth is no path to share lists on)
        const double zn0 = near_depth + fmin((double) c0.sz, (double) g[m - 1].cam.sz);
        if (!(zn0 > 0.01 * F) || (hypot(dev[0], dev[1]) * F + half * dev[2]) / zn0 > 2.0) return P;
        for (int a0 = 0; a0 < m; a0 += s_sub) {
            const int b0 = (a0 + s_sub < m ? a0 + s_sub : m) - 1;
            for (int k = a0; k <= b0; k++) { P.lead[k] = (uint8_t) a0; P.last[k] = (uint8_t) b0; P.size[k] = (uint8_t) (b0 - a0 + 1); }
            P.any = P.any || b0 > a0;
        }
        for (int q = 0; q < 3; q++) P.dev[q] = (float) (dev[q] * 1.0001);
        return P;
    };
    const SharePlan plan_now = placed ? share_plan(t, n) : share_plan(t, 0), plan_next = share_plan(next, n_next > 0 ? n_next : 0);
    const bool shared_now = plan_now.any, shared_next = plan_next.any;
    pj.pc = pc; fj.pc = pc; fj.n_next = n_next; fj.pad_ = (shared_now ? 1 : 0) | (shared_next ? 2 : 0);
    for (int q = 0; q < 3; q++) fj.dev[q] = plan_next.dev[q];
    for (int k = 0; k < KBE_FRAME_JOBS; k++) { fj.a_share[k] = plan_now.lead[k] | ((uint32_t) plan_now.size[k] << 8); fj.nx_share[k] = plan_next.lead[k] | ((uint32_t) plan_next.last[k] << 8); }
    for (int k = 0; k < KBE_FRAME_JOBS; k++) {
        const FusedTarget& f = t[k < n ? k : 0];
        const Scratch& sc = f.sc;
        n_tiles = (unsigned) (sc.tiles_x * sc.tiles_y);
        pj.tiles_x = sc.tiles_x; pj.tiles_y = sc.tiles_y;
        const int par = f.parity == 1 ? 1 : 0;
        pj.a[k] = place_args(f);
        FrameArgs& a = fj.a[k];
        a.cam = f.cam; a.tiles_x = sc.tiles_x; a.tiles_y = sc.tiles_y; a.place = pj.a[k].place;
        a.tile_count = pj.a[k].tile_count; a.cand = pj.a[k].cand; a.bin_flag = pj.a[k].bin_flag; a.bin_flag_next = (unsigned*) sc.hole_count + flag_zeroed_by(f);
        a.frame = f.frame_u8; a.depth = sc.depth; a.mask = sc.mask; a.holes = sc.holes; a.hole_count = sc.hole_count + par; a.bbox = sc.bbox; a.coarse = sc.coarse;
        a.render = f.render_f32; a.existing = f.existing_f32; a.zee = f.zee_f32; a.zee_pre = f.zee_pre_f32; a.spill = sc.buckets;
        fj.nx[k] = place_args(n_next > 0 ? next[k < n_next ? k : 0] : f);
    }
    if (shared_now)         // every frame reads the lists its sub-group's first frame's set holds
        // (... and still zeroes the list total its OWN set counts in two turns on: a set that joins an unshared or group-first turn
        // later must not count on top of what an earlier such turn left -- totals only grow, and past the budget every tile of the
        // frame scans the whole cloud: ADVICE r4)
        for (int k = 1; k < n; k++) {
            const int l = plan_now.lead[k];
            if (l != k) { fj.a[k].tile_count = fj.a[l].tile_count; fj.a[k].cand = fj.a[l].cand; fj.a[k].bin_flag = fj.a[l].bin_flag; }
        }
    if (!placed) hipLaunchKernelGGL(k_place, dim3(blocks_for((size_t) pc.Np), n), dim3(256), 0, s, pj);
    // the lean build (608 records per tile, six workgroups per CU) for clouds of about a point per pixel, the roomy one beyond
    // (`build`: KBE_STAGE_FUSED_LEAN / _ROOMY force one -- a switch for tests and measurements)
    const int forced = build;
    const Scratch& sc0 = t[0].sc;
    const bool lean = forced ? forced == 1 : (double) pc.Np <= KBE_LEAN_MAX_DENSITY * (double) t[0].cam.W * (double) t[0].cam.H;
    if (n == 1 && n_next <= 1) {
        FrameJob1 f1;
        f1.pc = pc; f1.n_next = n_next; f1.pad_ = 0; f1.a[0] = fj.a[0]; f1.nx[0] = fj.nx[0];
        f1.dev[0] = f1.dev[1] = f1.dev[2] = 0.0f; f1.a_share[0] = 1u << 8; f1.nx_share[0] = 0;
        if (n_next) hipLaunchKernelGGL(lean ? k_frame_ahead : k_frame_ahead_roomy, dim3(n_tiles), dim3(TILE_THREADS), 0, s, f1);
        else hipLaunchKernelGGL(lean ? k_frame : k_frame_roomy, dim3(n_tiles), dim3(TILE_THREADS), 0, s, f1);
    } else if (n_next) {                                                        // (also: one frame that places several)
        const bool dense = ahead_units_per_wave(N, sc0.tiles_x * TW, sc0.tiles_y * TH, n, n_next) > (size_t) AHEAD_UNITS + 1;
        if (dense) hipLaunchKernelGGL(k_frame_group_ahead_dense, dim3(n_tiles, n), dim3(TILE_THREADS), 0, s, fj);
        else hipLaunchKernelGGL(lean ? k_frame_group_ahead : k_frame_group_ahead_roomy, dim3(n_tiles, n), dim3(TILE_THREADS), 0, s, fj);
    }
    else hipLaunchKernelGGL(lean ? k_frame_group : k_frame_group_roomy, dim3(n_tiles, n), dim3(TILE_THREADS), 0, s, fj);
}
size_t fused_place_bytes(int N) { return (size_t) cloud_layout_base(N).Np * sizeof(Placement); }
}  // namespace kbe

#if defined(KBE_FRAME_PROBE)
extern "C" __attribute__((visibility("default"))) int kbe_debug_frame_probe(unsigned long long* out, size_t n_words)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_frame_probe), n_words * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
#endif
#if defined(KBE_FRAME_STATS)
extern "C" __attribute__((visibility("default"))) int kbe_debug_frame_stats(unsigned long long* out8, int reset)
{
    hipError_t e = hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_frame_stats), 8 * sizeof(unsigned long long));
    if (e == hipSuccess && reset) { const unsigned long long z[8] = {}; e = hipMemcpyToSymbol(HIP_SYMBOL(g_frame_stats), z, sizeof(z)); }
    return e == hipSuccess ? 0 : -1;
}
#endif
