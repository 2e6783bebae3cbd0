// kbe_holes.hip -- fill_disocclusion (common.py:833-937) of a frame from its hole list: the three schedules of k_fill_holes /
// k_fill_tables (results identical, byte for byte) and the per-frame tables of the third (k_hole_dist).  DESIGN.md section 4.
#include "kbe_tiles.h"

using namespace kbe;

namespace {

// ---------------------------------------------------------------------------------------
// hole fill: one 32-lane group per hole; lane = direction * 2 + end (0: against, 1: along)
// (common.py:838-924; the direction loop and both ray walks run in parallel, then the
// "strictly shorter, first direction wins" reduction picks the same source pixel)
// ---------------------------------------------------------------------------------------

#ifndef KBE_FILL_SERIAL_BATCH
#define KBE_FILL_SERIAL_BATCH 8
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
// over exactly those (build_strips: the tiles' boxes first, then the bitmask's rows at either end).  The end walking towards -u meets nothing once lo > t + 1, the end towards +u once hi < t - 1: the
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

__device__ void build_strips(const int4* __restrict__ bbox, const uint32_t* __restrict__ mask, int tiles_x, int tiles_y, int W, int H, float ux, float uy,
                             int first_bin, float2* __restrict__ out)
{
    const int b = first_bin + (int) threadIdx.x;
    if (b >= strip_bins(W, H)) return;
    const float c0 = (float) (b - strip_offset(ux, uy, W, H)) - STRIP_MARGIN, c1 = c0 + 1.0f + 2.0f * STRIP_MARGIN;
    // Per tile row (tile column for a flat direction) the one to three tiles under the strip, each with the box of its own valid pixels
    // (the tile launch's bbox table), the strip clipped to the box; then the tile whose box reaches farthest towards either end of the
    // strip is looked at ROW BY ROW in the validity bitmask: its valid pixels of the strip give that end's bound, unless another tile's
    // box reaches farther than they do (then that box's reach does: still a superset).  Until round 5: the x-extent of each whole tile
    // ROW (the y-extent of each tile column) -- 4.46 of a late dolly frame's 16 directions per hole passed the test where 1.73
    // complete; the tiles' own boxes alone: 2.47; with the one tile looked at: 2.01; every tile looked at until nothing can improve
    // (exact): 1.91 -- but that walk's chain of dependent loads made k_hole_dist slower than the fill gained (tools/strip_proto.c on
    // the oracle's masks, the restatement of this function).  The directions that pass without completing are the expensive ones: they
    // walk to the end of their strip.
    const bool steep = fabsf(uy) >= fabsf(ux);                  // the line crosses every row once: walk the tile rows
    const int n = steep ? tiles_y : tiles_x, m = steep ? tiles_x : tiles_y;
    const float ua = steep ? ux : uy, ub = steep ? uy : ux;     // a = the coordinate along a row (column), b = across
    const float inv = 1.0f / ub;
    const int sa = steep ? TW : TH, sb = steep ? TH : TW;
    const int wpr = (W + 31) >> 5;
    const bool rows_cross = fabsf(uy) >= 1.0e-6f;               // else: a horizontal direction, a strip is whole rows
    const float inv_uy = rows_cross ? 1.0f / uy : 0.0f;
    static_assert(TW == 32 && TH <= 16, "a tile row is one word of the validity bitmask, a tile at most sixteen of them");
    // the strip's extent along a over the rows (columns) b0 .. b1 -- steep: c = -uy x + ux y => x = (ux y - c) / uy; flat: y = (c + uy x) / ux
    const auto along = [&](float b0, float b1, float& a0, float& a1) {
        const float v0 = steep ? (ua * b0 - c0) * inv : (c0 + ua * b0) * inv, v1 = steep ? (ua * b0 - c1) * inv : (c1 + ua * b0) * inv;
        const float v2 = steep ? (ua * b1 - c0) * inv : (c0 + ua * b1) * inv, v3 = steep ? (ua * b1 - c1) * inv : (c1 + ua * b1) * inv;
        a0 = fminf(fminf(v0, v1), fminf(v2, v3)) - 0.01f; a1 = fmaxf(fmaxf(v0, v1), fmaxf(v2, v3)) + 0.01f;
    };
    float lo1 = INFINITY, lo2 = INFINITY, hi1 = -INFINITY, hi2 = -INFINITY;        // the farthest and the second farthest reach of a box, either end
    int lo_tile = -1, hi_tile = -1;
    for (int i = 0; i < n; i++) {
        float a0, a1;
        along((float) (i * sb), (float) (i * sb + sb - 1), a0, a1);
        const int j0 = max((int) floorf(a0 / (float) sa), 0), j1 = min((int) floorf(a1 / (float) sa), m - 1);
        for (int j = j0; j <= j1; j++) {
            const int tile = steep ? i * tiles_x + j : j * tiles_x + i;
            const int4 bb = bbox[tile];
            if (bb.z < bb.x) continue;                          // a tile without a valid pixel
            const float q0 = (float) (steep ? bb.y : bb.x), q1 = (float) (steep ? bb.w : bb.z);        // the box across ...
            float p0, p1;
            along(q0, q1, p0, p1);
            p0 = fmaxf(p0, (float) (steep ? bb.x : bb.y)); p1 = fminf(p1, (float) (steep ? bb.z : bb.w));     // ... and along
            if (p0 > p1) continue;
            // t = ux x + uy y = ua a + ub b over [p0, p1] x [q0, q1]
            const float t0 = ua * p0 + ub * q0, t1 = ua * p0 + ub * q1, t2 = ua * p1 + ub * q0, t3 = ua * p1 + ub * q1;
            const float tmin = fminf(fminf(t0, t1), fminf(t2, t3)), tmax = fmaxf(fmaxf(t0, t1), fmaxf(t2, t3));
            if (tmin < lo1) { lo2 = lo1; lo1 = tmin; lo_tile = tile; } else lo2 = fminf(lo2, tmin);
            if (tmax > hi1) { hi2 = hi1; hi1 = tmax; hi_tile = tile; } else hi2 = fmaxf(hi2, tmax);
        }
    }
    // the valid pixels of the strip in one tile, row by row in the bitmask (the sixteen words requested together): the smallest
    // (`low`) or the largest t among them; none: +inf / -inf
    const auto in_tile = [&](int tile, bool low) -> float {
        float best = low ? INFINITY : -INFINITY;
        if (tile < 0) return best;
        const int4 bb = bbox[tile];
        const int ty = tile / tiles_x, tx = tile - ty * tiles_x, x0 = tx * TW;
        uint32_t words[TH];
#pragma unroll
        for (int r = 0; r < TH; r++) words[r] = mask[(size_t) min(bb.y + r, bb.w) * wpr + tx];
#pragma unroll
        for (int r = 0; r < TH; r++) {
            const int y = bb.y + r;
            if (y > bb.w) continue;
            float xa = -1.0e9f, xb = 1.0e9f;
            if (rows_cross) {
                const float e0 = (ux * (float) y - c0) * inv_uy, e1 = (ux * (float) y - c1) * inv_uy;
                xa = fminf(e0, e1) - 0.01f; xb = fmaxf(e0, e1) + 0.01f;
            } else {
                const float c = ux * (float) y;
                if (c < c0 - 0.01f || c > c1 + 0.01f) continue;
            }
            const int xl = (int) ceilf(fmaxf(xa, (float) bb.x)), xr = (int) floorf(fminf(xb, (float) bb.z));
            if (xl > xr) continue;
            const uint32_t w = words[r] & (0xFFFFFFFFu << (xl - x0)) & (0xFFFFFFFFu >> (31 - (xr - x0)));
            if (!w) continue;
            const float ta = ux * (float) (x0 + __builtin_ctz(w)) + uy * (float) y, tb = ux * (float) (x0 + 31 - __builtin_clz(w)) + uy * (float) y;
            best = low ? fminf(best, fminf(ta, tb)) : fmaxf(best, fmaxf(ta, tb));
        }
        return best;
    };
    const float lo = fminf(in_tile(lo_tile, true), lo2), hi = fmaxf(in_tile(hi_tile, false), hi2);
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
__device__ __forceinline__ void hole_dist_body(const uint32_t* __restrict__ mask, int W, int H, uint8_t* __restrict__ dist,
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
            if (strips) build_strips(bbox, mask, tiles_x, tiles_y, W, H, dirs.x[d], dirs.y[d], (si - d * per_dir) * 256, strips + (size_t) d * bins);
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

#ifndef KBE_FILL_TABLES_BLOCKS
#define KBE_FILL_TABLES_BLOCKS 768      // workgroups of k_fill_tables per frame (3 per CU; with four frames per launch and four lanes: 96.5 us per dolly frame, 2048: 99.5)
#endif
#ifndef KBE_FILL_BURST
#define KBE_FILL_BURST 4                // steps a creeping ray takes together (8: 102 us per dolly frame, 4: 99, with four frames per launch) ...
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
#if defined(KBE_FILL_TABLES_WAVES)
#define KBE_FILL_TABLES_ATTR __attribute__((amdgpu_waves_per_eu(KBE_FILL_TABLES_WAVES, KBE_FILL_TABLES_WAVES)))
#else
#define KBE_FILL_TABLES_ATTR
#endif
__device__ __forceinline__ void fill_tables_body(const int* __restrict__ holes, const int* __restrict__ hole_count, int min_holes,
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
            float ux = 0.0f, uy = 0.0f, inv_umax = 1.0f;
            int k_dead = FILL_MAX_STEPS + 1;           // the step count from which the end is past every valid pixel of its strip (below)
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
                            int m = 0;
                            if (!(((unsigned) ix < (unsigned) W) & ((unsigned) iy < (unsigned) H))) st = END_DEAD;       // :880-885 / :891-896
                            else if (k >= k_dead) st = END_DEAD;                                                        // past every valid pixel of its strip
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
                    bstop[j] = !inb || (k - rx + j + 1 >= k_dead);
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
            // (Round 5 tried more, twice: with few lanes of a wave still walking -- 4, 8, 16 -- a ray got the WHOLE WAVE, lane j looking at the
            // position j + 1 steps on, all 64 from the ray's integer mantissas; first for creeping rays only, then for any walking ray.
            // Same fills, no gain either time: 89.5 us per dolly frame with and without.  tools/fill_stats.py and a PMC pass say why: the
            // video loop is bound by its INSTRUCTION COUNT -- 38 M wave-level VALU per frame, 23 M of them this kernel's, = 67 us of issue
            // in an 89 us frame -- not by its longest rays (the longest ray of a late frame lives ~95 iterations, and sixteen frames are
            // in flight).  Removed again: git show 5750a8a:ken-burns-effect_amd/csrc/kbe_holes.hip.)
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
                            // The end is past every valid pixel of its strip once its pixel's coordinate along the direction, t = u . q, is
                            // more than STRIP_MARGIN beyond the strip's bound.  The pixel k steps on lies within 0.75 of the hole's t -/+ k
                            // (a unit direction; rounding per axis, < 0.03 of drift), so from k_dead = ceil(|bound - t| + 2.8) steps on that
                            // holds for sure -- one integer comparison per landing instead of the coordinate's two conversions, a
                            // multiply-add and a comparison (the ray may die two or three steps later than with the coordinate itself: it
                            // meets nothing there, that is what the bound says)
                            k_dead = FILL_MAX_STEPS + 1;
                            if (strips) {
                                const float2 lh = strips[(size_t) d * bins + ((int) floorf(ux * (float) iy - uy * (float) ix) + s_off[d])];
                                const float t_hole = ux * (float) ix + uy * (float) iy;
                                const float room = is_b ? lh.y - t_hole : t_hole - lh.x;            // -inf: nothing on that side at all
                                k_dead = room > (float) FILL_MAX_STEPS ? FILL_MAX_STEPS + 1 : (int) ceilf(fmaxf(room, -2.0f) + STRIP_MARGIN + 1.8f);
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
__device__ __forceinline__ void fill_holes_body(const int* __restrict__ holes, const int* __restrict__ hole_count,
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


}  // namespace

namespace {

// The kernels of a fill take up to KBE_FILL_JOBS frames per launch (blockIdx.z / .y = the frame): a lane of the video loop
// whose frames fill with the tables renders two frames and fills them TOGETHER -- the table-driven fill is bound by its
// own chain of dependent look-ups (272 us alone on the chip, 352 us with four of them overlapping), so two frames per
// launch take little longer than one.
__global__ void __launch_bounds__(256) k_hole_dist(FillJobs jobs, int W, int H, int min_holes, int tiles_x, int tiles_y, FillDirs dirs, int image_rows, int use_strips)
{
    const FillJob& J = jobs.j[blockIdx.z];
    hole_dist_body(J.mask, W, H, J.dist, J.hole_count, min_holes, J.bbox, tiles_x, tiles_y, use_strips ? J.strips : nullptr, dirs, J.coarse, J.dist_blocks, image_rows);
}

__global__ void __launch_bounds__(256) KBE_FILL_TABLES_ATTR k_fill_tables(FillJobs jobs, int min_holes, int W, int H, FillDirs dirs, FillRect rect, int n_tiles,
                                                                         int tiles_x, int tiles_y, int use_strips)
{
    const FillJob& J = jobs.j[blockIdx.y];
    fill_tables_body(J.holes, J.hole_count, min_holes, J.depth, W, H, dirs, rect, J.frame, J.render, n_tiles, J.bbox, tiles_x, tiles_y, J.dist,
                     use_strips ? J.strips : nullptr, J.dist_blocks);
}

__global__ void __launch_bounds__(KBE_FILL_BLOCK) k_fill_holes(FillJobs jobs, int W, int H, FillDirs dirs, FillRect rect, int n_tiles, int fill_mode,
                                                               int tiles_x, int tiles_y, int tables)
{
    const FillJob& J = jobs.j[blockIdx.y];
    fill_holes_body(J.holes, J.hole_count, J.depth, J.mask, W, H, dirs, rect, J.frame, J.render, J.zkeys, J.tile_count, n_tiles, J.bbox, fill_mode, J.coarse,
                    tiles_x, tiles_y, J.reset_scatter_scratch, J.next_hole_count, tables);
}

}  // namespace

namespace kbe {
// the hole fill of `n_jobs` frames of the same size (1 or 2): with KBE_STAGE_FILL_DIST the tables and the table-driven fill in
// front of k_fill_holes (each of them returns at once when a frame has fewer holes than the schedule asks for)
void launch_fill(hipStream_t s, int n_jobs, const FillTarget* targets, int W, int H, int stages, const FillDirs& dirs, const FillRect& rect, int n_tiles)
{
    FillJobs jobs;
    const Scratch& sc0 = targets[0].sc;
    for (int k = 0; k < KBE_FILL_JOBS; k++) {
        const FillTarget& t = targets[k < n_jobs ? k : 0];
        FillJob& j = jobs.j[k];
        j.holes = t.sc.holes; j.hole_count = t.hole_count; j.depth = t.sc.depth; j.mask = t.sc.mask; j.frame = t.frame_u8; j.render = t.render_f32;
        j.zkeys = t.sc.zkeys; j.tile_count = t.sc.tile_count; j.bbox = t.sc.bbox; j.coarse = t.sc.coarse; j.dist = t.sc.dist; j.strips = t.sc.strips;
        j.dist_blocks = t.sc.dist_blocks; j.reset_scatter_scratch = t.reset_scatter_scratch; j.next_hole_count = t.next_hole_count;
    }
    const size_t want_fill = (size_t) W * H / 64, max_fill = (size_t) KBE_FILL_MAX_BLOCKS * 256 / KBE_FILL_BLOCK;       // the same number of threads
    const unsigned fill_blocks = (unsigned) (want_fill < max_fill ? (want_fill > 0 ? want_fill : 1) : max_fill);
    int tables = 0;
    const int fill_mode = (stages & KBE_STAGE_FILL_PER_LANE) ? 1 : ((stages & KBE_STAGE_FILL_PER_HALFWAVE) || !(stages & KBE_STAGE_FILL_BY_COUNT) ? 2 : 0);
    if ((stages & KBE_STAGE_FILL_DIST) && (stages & (KBE_STAGE_FILL_PER_LANE | KBE_STAGE_FILL_BY_COUNT)) && fill_tables_fit(W, H)) {
        const int min_holes = (stages & KBE_STAGE_FILL_PER_LANE) ? 0 : KBE_FILL_SERIAL_MIN;
        const int use_strips = strips_fit(sc0) ? 1 : 0;
        const int gx = (W + DT_W - 1) / DT_W, gy = (H + DT_H - 1) / DT_H;
        const int cw = sc0.tiles_x * (TW / 8), ch = sc0.tiles_y * (TH / 8);
        const int extra = 16 * ((strip_bins(W, H) + 255) / 256) + ((cw + DT_W - 1) / DT_W) * ((ch + DT_H - 1) / DT_H);
        hipLaunchKernelGGL(k_hole_dist, dim3(gx, gy + (extra + gx - 1) / gx, n_jobs), dim3(256), 0, s, jobs, W, H, min_holes, sc0.tiles_x, sc0.tiles_y, dirs, gy, use_strips);
        const size_t hw = (size_t) W * H;
        const unsigned blocks = (unsigned) ((hw + 255) / 256 < KBE_FILL_TABLES_BLOCKS ? (hw + 255) / 256 : KBE_FILL_TABLES_BLOCKS);
        hipLaunchKernelGGL(k_fill_tables, dim3(blocks, n_jobs), dim3(256), 0, s, jobs, min_holes, W, H, dirs, rect, n_tiles, sc0.tiles_x, sc0.tiles_y, use_strips);
        tables = 1 + min_holes;                                 // k_fill_holes: a frame is done if it has >= tables - 1 holes
    }
    hipLaunchKernelGGL(k_fill_holes, dim3(fill_blocks, n_jobs), dim3(KBE_FILL_BLOCK), 0, s, jobs, W, H, dirs, rect, n_tiles, fill_mode, sc0.tiles_x, sc0.tiles_y, tables);
}

}  // namespace kbe

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
#endif
