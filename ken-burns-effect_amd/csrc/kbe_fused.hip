// kbe_fused.hip -- the fused scatter: render_pointcloud (common.py:428-686) of one frame in ONE launch, from the packed
// cloud (kbe_cloud.h).  Host side: kbe_render_frame_fused / kbe_render_video in kbe_frame.hip, through launch_frame_fused.
#include "kbe_cloud.h"
#include "kbe_tiles.h"

using namespace kbe;

namespace kbe { PackedCloud cloud_open(const void* packed, int N, double focal); }      // kbe_cloud.hip

namespace {

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


}  // namespace

namespace kbe {
void launch_frame_fused(hipStream_t s, unsigned n_tiles, const void* packed, int N, double cloud_focal, const Camera& cam, const Scratch& sc, int* hole_count,
                        uint8_t* frame_u8, float* render_f32, float* existing_f32, float* zee_f32, float* zee_pre_f32)
{
    FrameArgs a;
    a.pc = cloud_open(packed, N, cloud_focal);
    a.cam = cam;
    a.tiles_x = sc.tiles_x; a.tiles_y = sc.tiles_y;
    a.frame = frame_u8; a.depth = sc.depth; a.mask = sc.mask; a.holes = sc.holes; a.hole_count = hole_count; a.bbox = sc.bbox; a.coarse = sc.coarse;
    a.render = render_f32; a.existing = existing_f32; a.zee = zee_f32; a.zee_pre = zee_pre_f32; a.spill = sc.buckets;
    hipLaunchKernelGGL(k_frame, dim3(n_tiles), dim3(TILE_THREADS), 0, s, a);
}
}  // namespace kbe

#if defined(KBE_FRAME_STATS)
extern "C" __attribute__((visibility("default"))) int kbe_debug_frame_stats(unsigned long long* out8, int reset)
{
    hipError_t e = hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_frame_stats), 8 * sizeof(unsigned long long));
    if (e == hipSuccess && reset) { const unsigned long long z[8] = {}; e = hipMemcpyToSymbol(HIP_SYMBOL(g_frame_stats), z, sizeof(z)); }
    return e == hipSuccess ? 0 : -1;
}
#endif
