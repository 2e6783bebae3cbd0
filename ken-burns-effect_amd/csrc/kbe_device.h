// kbe_device.h -- device-side building blocks shared by the render kernels (gfx950 only).
//
// Numerical contract: oracle/kbe_oracle.c (pinned to the reference kernel text).  This
// translation unit must be built with -ffp-contract=off: every fp32 operation below rounds
// once, the only fused multiply-adds are the explicit __builtin_fmaf calls, and the fp64
// sub-expressions are the ones the reference's double literals cause.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#pragma clang fp contract(off)

namespace kbe {

constexpr int kWave = 64;   // CDNA wavefront

// ---------------------------------------------------------------------------------------
// order-preserving fp32 <-> uint32 key (replaces the CAS-loop float atomicMin of
// /root/reference/utils/common.py:275-283 by one native global/LDS atomic umin)
// ---------------------------------------------------------------------------------------
// (written as shift / or / xor rather than compare-and-select: the same three instructions per key, but of the kinds a SIMD
// issues through its second port -- v_ashrrev_i32, v_or_b32, v_xor_b32 -- where v_cmp and v_cndmask take the first, which is
// the one the frame kernels are bound by: DESIGN.md section 4, profiles/r04_valu_rate.txt)
__device__ __forceinline__ uint32_t zkey_encode(float f)
{
    const uint32_t b = __float_as_uint(f);
#if defined(KBE_KEY_SELECT) && KBE_KEY_SELECT
    return b ^ ((b >> 31) ? 0xFFFFFFFFu : 0x80000000u);
#else
    return b ^ ((uint32_t) ((int32_t) b >> 31) | 0x80000000u);          // sign set: ^ 0xFFFFFFFF, clear: ^ 0x80000000
#endif
}

__device__ __forceinline__ float zkey_decode(uint32_t k)
{
#if defined(KBE_KEY_SELECT) && KBE_KEY_SELECT
    return __uint_as_float(k ^ ((k >> 31) ? 0x80000000u : 0xFFFFFFFFu));
#else
    return __uint_as_float(k ^ ~((uint32_t) ((int32_t) k >> 31) & 0x7FFFFFFFu));        // top bit set: ^ 0x80000000, clear: ^ 0xFFFFFFFF
#endif
}

// ---------------------------------------------------------------------------------------
// camera: everything the projection needs, precomputed once per launch on the host
// ---------------------------------------------------------------------------------------
struct Camera {
    float focal_f;      // (float) dblFocal             common.py:447 make_float3(0, 0, F)
    float fb_f;         // (float) fb, for the division-free fast path of dblError
    double fb;          // dblFocal * dblBaseline       common.py:470
    double half_w;      // 0.5 * W                      common.py:467
    double half_h;      // 0.5 * H                      common.py:468
    float cx_f, cy_f;   // (float) (0.5 * W - 0.5), (float) (0.5 * H - 0.5): the same position in one fp32 addition
    int fp32_centre;    // W, H >= 2: that addition is bit-identical to the fp64 form (tests/centre_offset_check.c)
    int W, H;
    int has_shift;      // apply process_shift (common.py:104-109) on the fly
    float sx, sy, sz;
};

// n / d, correctly rounded, as the instructions the compiler's own expansion of an fp32 division comes down to when neither
// v_div_scale_f32 rescales an operand nor v_div_fixup_f32 patches the result: the reciprocal estimate, one refinement of it,
// the quotient and two residual corrections -- eight instructions instead of twelve.  Bit-identical to `n / d` whenever
// |d| is in [2^-100, 2^100] and n is 0 or |n| in [2^-100, 2^100] (then no scaling condition of v_div_scale_f32 holds: neither operand
// nor 1 / d nor n / d is near the denormal range, the exponents are < 96 apart, and n's exponent is above 23; v_div_fixup_f32
// only restores the sign there, which the fmas carry anyway).  CALLERS GUARANTEE THE RANGE.  kbe_selftest_division runs it
// against `/` on the GPU (tests/test_hip_parity.py).
__device__ __forceinline__ float div_unscaled(float n, float d)
{
    float y = __builtin_amdgcn_rcpf(d);
    y = __builtin_fmaf(__builtin_fmaf(-d, y, 1.0f), y, y);
    float q = n * y;
    q = __builtin_fmaf(__builtin_fmaf(-d, q, n), y, q);
    return __builtin_fmaf(__builtin_fmaf(-d, q, n), y, q);
}

// process_shift's per-point arithmetic: x *= z / (z + 1e-7); y likewise; then += shift.
__device__ __forceinline__ void apply_shift(const Camera& cam, float& x, float& y, float& z)
{
    if (cam.has_shift) {
        // z + 1e-7f == z for every z >= 2 (half an ulp is then > 1e-7): the ratio is exactly 1 and the division
        // (11 instructions) and the two products with it (x * 1.0f is x) are skipped when that holds for the whole wave
        const float zz = z + 0.0000001f;
        if (__ballot(zz != z) != 0ull) {        // wave-uniform
            asm volatile("" ::: "memory");      // keeps the division inside the branch (it was if-converted into a select)
            const float r = z / zz;
            x = x * r + cam.sx;
            y = y * r + cam.sy;
        } else {
            x = x + cam.sx;
            y = y + cam.sy;
        }
        z = z + cam.sz;
    }
}

struct Proj {
    float err;          // dblError
    int nwx, nwy;       // north-west corner
    float w[4];         // NW, NE, SW, SE bilinear weights
};

// The projection is split in three so that callers can reject a point before paying for the
// fp64 division: (1) cull + image-plane position, (2) corner + weights, (3) dblError.

// common.py:453-468.  Returns false when the point touches no pixel at all.
__device__ __forceinline__ bool project_xy(const Camera& cam, float px, float py, float pz, float& ox, float& oy)
{
    // :453 (also covers :461).  `(double) pz >= 0.001` in fp32: 0.001f is the smallest float that is >= the double 0.001
    if (!(pz >= 0.001f)) return false;
    const float lvx = 0.0f - px, lvy = 0.0f - py, lvz = 0.0f - pz;
#if defined(KBE_DIV_FAST) && !KBE_DIV_FAST
    const float dist = (cam.focal_f - pz) / lvz;               // :457-459
#else
    // :457-459.  pz >= 0.001 here; F - pz is 0 or at least half an ulp of the smaller of the two (>= 2^-34): with pz and
    // F below 2^100 -- one test for the wave -- div_unscaled IS the division
    float dist;
    const float num = cam.focal_f - pz;
    const bool f_small = fabsf(cam.focal_f) < 1.0e30f;             // uniform; loop-invariant where a wave places several units
    if (f_small && __ballot(!(pz < 1.0e30f)) == 0ull) dist = div_unscaled(num, lvz);
    else dist = num / lvz;
#endif
    const float ix = __builtin_fmaf(dist, lvx, px);            // :465 as NVRTC (--fmad=true) emits it
    const float iy = __builtin_fmaf(dist, lvy, py);
    if (cam.fp32_centre) {                                     // wave-uniform; the normal case
        // both fp64 operations of :467-468 are exact whenever ix matters to the result, so the two roundings
        // collapse into the one of an fp32 addition (swept over every fp32 ix: tests/centre_offset_check.c)
        ox = ix + cam.cx_f;
        oy = iy + cam.cy_f;
    } else {
        ox = (float) (((double) ix + cam.half_w) - 0.5);       // :467
        oy = (float) (((double) iy + cam.half_h) - 0.5);       // :468
    }
    return (fabsf(ox) < 1.0e9f) && (fabsf(oy) < 1.0e9f);       // see kbe.h "Inputs must be finite"
}

// common.py:472-484
__device__ __forceinline__ void project_weights(float ox, float oy, Proj& p)
{
    const float fx = floorf(ox), fy = floorf(oy);
    p.nwx = (int) fx;
    p.nwy = (int) fy;
    const float ex = (float) (p.nwx + 1), ey = (float) (p.nwy + 1);      // east / south coordinates
    p.w[0] = (ex - ox) * (ey - oy);                             // :481 NW
    p.w[1] = (ox - fx) * (ey - oy);                             // :482 NE
    p.w[2] = (ex - ox) * (oy - fy);                             // :483 SW
    p.w[3] = (ox - fx) * (oy - fy);                             // :484 SE
}

// common.py:470: 1000000.0 - ((F * B) / (z + 0.0000001)), fp64 throughout, one final rounding
__device__ __forceinline__ float project_err(const Camera& cam, float pz)
{
    return (float) (1000000.0 - (cam.fb / ((double) pz + 0.0000001)));
}

// The same value WITHOUT the fp64 division on the common path.  With Q = F*B / (z + 1e-7) the result is
// fl32(1e6 - Q); for 1e6 - Q in [2^19, 2^20) that is (16e6 - round(16 Q)) / 16.  An fp32 estimate
// q = fb_f * rcp(z) is within 4e-7 relative of Q (v_rcp_f32 1 ulp, one multiply, the rounding of fb_f, and
// the dropped 1e-7 for z >= 16), so round(16 Q) is known for sure unless 16 q sits within that error of a
// half-integer; those rare lanes (< 0.1 %) take the exact fp64 route.  Bit-identical to project_err
// (tests/test_hip_parity.py::test_fast_dbl_error_is_exact sweeps the rounding boundaries).
__device__ __forceinline__ float project_err_fast(const Camera& cam, float pz)
{
    const float s = 16.0f * (cam.fb_f * __builtin_amdgcn_rcpf(pz));
    const float n = rintf(s);
    const bool sure = (pz >= 16.0f) & (s > 0.0f) & (s < 7600000.0f) & (fabsf(s - n) < 0.5f - (s * 4.0e-7f + 2.0e-6f));
    if (__builtin_expect(!sure, 0)) return project_err(cam, pz);
    return (float) (16000000 - (int) n) * 0.0625f;
}

// common.py:447-484 (== :599-636) in one piece
__device__ __forceinline__ bool project(const Camera& cam, float px, float py, float pz, Proj& p)
{
    float ox, oy;
    if (!project_xy(cam, px, py, pz, ox, oy)) return false;
    p.err = project_err(cam, pz);
    project_weights(ox, oy, p);
    return true;
}

// common.py:486-506: first of NW, NE, SW, SE whose weight is >= the other three; -1 = none
__device__ __forceinline__ int winner_corner(const Proj& p)
{
    const float nw = p.w[0], ne = p.w[1], sw = p.w[2], se = p.w[3];
    if ((nw >= ne) & (nw >= sw) & (nw >= se)) return 0;
    if ((ne >= nw) & (ne >= sw) & (ne >= se)) return 1;
    if ((sw >= nw) & (sw >= ne) & (sw >= se)) return 2;
    if ((se >= nw) & (se >= ne) & (se >= sw)) return 3;
    return -1;
}

// The same for FINITE weights (every point project_xy lets through: |ox|, |oy| < 1e9), without branches: the first
// corner whose weight is >= the other three is the first that attains their maximum -- an argmax that only a strictly
// larger weight displaces.  (With a NaN among the weights the reference picks none; callers that cannot rule NaNs out
// use winner_corner.)
__device__ __forceinline__ int winner_corner_finite(const Proj& p)
{
    float best = p.w[0];
    int k = 0;
    if (p.w[1] > best) { best = p.w[1]; k = 1; }
    if (p.w[2] > best) { best = p.w[2]; k = 2; }
    if (p.w[3] > best) { k = 3; }
    return k;
}

__device__ __forceinline__ bool inside(int x, int y, int W, int H)
{
    return ((unsigned) x < (unsigned) W) & ((unsigned) y < (unsigned) H);      // two compares instead of four
}

// `a + 1.0` evaluated in double (common.py:556-557, :639) equals the fp32 sum exactly when a lies in
// [2^19, 2^20 - 1): there fp32 has a spacing of 1/16, so adding 1.0 is exact.  dblError lives in that
// band for every point farther than F*B/475712 from the camera, and so does the empty value 1e6.
__device__ __forceinline__ bool plus_one_is_exact(float a)
{
    return (a >= 524288.0f) & (a < 1048575.0f);
}

// common.py:542-567 for one pixel; `at(x, y)` returns the pre-degrid value of an in-image pixel.
// The nine values are fetched first; when all of them allow it (wave-uniform test, the normal case)
// the `>= ... + 1.0` comparisons run in fp32, otherwise in fp64 as written in the reference.
template <class At>
__device__ __forceinline__ float degrid_pixel(int x, int y, int W, int H, At at)
{
    const float c = at(x, y);
    const int ox[4] = { 1, 0, 1, 1 };
    const int oy[4] = { 0, 1, 1, -1 };
    float a[4], d[4];
    bool use[4];
    bool exact = true;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int x1 = x + ox[k], y1 = y + oy[k], x2 = x - ox[k], y2 = y - oy[k];
        use[k] = inside(x1, y1, W, H) && inside(x2, y2, W, H);
        a[k] = use[k] ? at(x1, y1) : 1000000.0f;
        d[k] = use[k] ? at(x2, y2) : 1000000.0f;
        exact = exact && plus_one_is_exact(a[k]) && plus_one_is_exact(d[k]);
    }
    int count = 0;
    float sum = 0.0f;
    if (__all(exact)) {
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (use[k] && (c >= a[k] + 1.0f) && (c >= d[k] + 1.0f)) { count += 2; sum += a[k]; sum += d[k]; }
    } else {
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (use[k] && ((double) c >= (double) a[k] + 1.0) && ((double) c >= (double) d[k] + 1.0)) { count += 2; sum += a[k]; sum += d[k]; }
    }
    return count > 0 ? fminf(c, sum / (float) count) : c;
}

// The same, written exactly as the reference (fp64 comparisons throughout, no cross-lane test): for callers whose lanes
// diverge (the serial-schedule kernel, where only the lanes on the current wavefront take part).
template <class At>
__device__ __forceinline__ float degrid_pixel_one(int x, int y, int W, int H, At at)
{
    const float c = at(x, y);
    const int ox[4] = { 1, 0, 1, 1 };
    const int oy[4] = { 0, 1, 1, -1 };
    int count = 0;
    float sum = 0.0f;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int x1 = x + ox[k], y1 = y + oy[k], x2 = x - ox[k], y2 = y - oy[k];
        if (!(inside(x1, y1, W, H) && inside(x2, y2, W, H))) continue;                        // :548-553
        const float a = at(x1, y1), d = at(x2, y2);
        if (((double) c >= (double) a + 1.0) && ((double) c >= (double) d + 1.0)) { count += 2; sum += a; sum += d; }   // :556-561
    }
    return count > 0 ? fminf(c, sum / (float) count) : c;                                     // :564-566
}

// The same pixel when all nine values lie in [2^19, 1e6] (tested once per tile by the caller): the `+ 1.0`
// comparisons are exact in fp32 (plus_one_is_exact), a neighbour outside the image reads as the empty value 1e6
// and can then never pass `c >= 1e6 + 1`, which is how the reference's bounds test (:548-553) drops its pair,
// and the mean needs no division: with y = RN(1 / (2n)), q = s y, q' = fma(fma(-2n, q, s), y, q) IS the
// correctly rounded s / (2n) (Markstein; checked for every float s in [1, 1.7e7] and n = 1..4 by
// tests/markstein_check.c).  Bit-identical to degrid_pixel.
__device__ __forceinline__ float degrid_pixel_fast(float c, const float (&a)[4], const float (&d)[4])
{
    int n = 0;
    float s = 0.0f;
    // in the band both a + 1 and c - 1 are exact, so (c >= a + 1) & (c >= d + 1) is c - 1 >= max(a, d): one
    // subtraction for the pixel, one max and one comparison per pair -- on the bit patterns, which order positive
    // floats as integers (a float max would first canonicalise its operands)
    const int cm1 = __float_as_int(c - 1.0f);
    int m[4];
#pragma unroll
    for (int k = 0; k < 4; k++) m[k] = max(__float_as_int(a[k]), __float_as_int(d[k]));
    // most waves hold no pixel that any pair pulls down (a smooth surface without 1-pixel gaps): one test for the
    // whole wave, then the pixel keeps its value (wave-uniform branch)
    if (__ballot(cm1 >= min(min(m[0], m[1]), min(m[2], m[3]))) == 0ull) return c;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const bool t = cm1 >= m[k];
        n += t ? 1 : 0;
        s += t ? a[k] : 0.0f;                                   // s + 0.0f == s: same sum, same order as :559-560
        s += t ? d[k] : 0.0f;
    }
    // y = RN(1 / (2n)): 2^-1, 2^-2, 2^-3 for n = 1, 2, 4 by exponent arithmetic, the rounded sixth for n = 3 (the
    // select chain n == 3 ? .. : n == 1 ? .. compiled into divergent branches)
    const int e = n - (n >> 2);
    const float y = n == 3 ? 0.16666667f : __int_as_float(0x3F800000 - (e << 23));
    const float q = s * y;
    const float mean = __builtin_fmaf(__builtin_fmaf(-(float) (2 * n), q, s), y, q);
    return n > 0 ? fminf(c, mean) : c;
}

__device__ __forceinline__ bool degrid_fast_ok(float z) { return (z >= 524288.0f) & (z <= 1000000.0f); }

// common.py:255: (x * 255.0).clip(0.0, 255.0).astype(np.uint8)
__device__ __forceinline__ uint8_t to_u8(float v)
{
    // clip as one median-of-three (finite inputs: identical to the two comparisons; a NaN would become 0 either way
    // after the conversion)
    return (uint8_t) (int) __builtin_amdgcn_fmed3f(v * 255.0f, 0.0f, 255.0f);
}

// fire-and-forget fp32 add: the native global_atomic_add_f32 (the buffers are ordinary
// coarse-grained device allocations, so the hardware atomic is valid)
__device__ __forceinline__ void atomic_add_f32(float* p, float v)
{
    unsafeAtomicAdd(p, v);
}

}  // namespace kbe
