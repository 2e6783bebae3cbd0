// kbe_hip.hip -- gfx950 kernels and the extern "C" entry points declared in include/kbe.h.
//
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared
// Reference lines cited as common.py:NNN are /root/reference/utils/common.py.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <string.h>

#include "kbe.h"
#include "kbe_device.h"
#include "kbe_fill.h"
#include "kbe_host.h"

#pragma clang fp contract(off)

using namespace kbe;

thread_local char kbe::g_err[256] = "";

namespace {

// ---------------------------------------------------------------------------------------
// elementwise helpers
// ---------------------------------------------------------------------------------------
__global__ void k_fill_u32(uint32_t* p, size_t n, uint32_t v)
{
    const size_t stride = (size_t) gridDim.x * blockDim.x;
    for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) p[i] = v;
}

__global__ void k_zkeys_decode(const uint32_t* keys, size_t n, float* zee)
{
    const size_t stride = (size_t) gridDim.x * blockDim.x;
    for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) zee[i] = zkey_decode(keys[i]);
}

// self-test: dblError of each z through the exact fp64 expression and through the fast path
__global__ void k_selftest_err(const float* __restrict__ z, size_t n, Camera cam, float* __restrict__ fast, float* __restrict__ exact)
{
    const size_t stride = (size_t) gridDim.x * blockDim.x;
    for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        fast[i] = project_err_fast(cam, z[i]);
        exact[i] = project_err(cam, z[i]);
    }
}

// self-test: the frame loop's eight-instruction division (div_unscaled, kbe_device.h) against `/`
__global__ void k_selftest_division(const float* __restrict__ num, const float* __restrict__ den, size_t n, float* __restrict__ fast, float* __restrict__ ieee)
{
    const size_t stride = (size_t) gridDim.x * blockDim.x;
    for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        fast[i] = div_unscaled(num[i], den[i]);
        ieee[i] = num[i] / den[i];
    }
}

// ---------------------------------------------------------------------------------------
// kernel_pointrender_updateZee (common.py:435-507)
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock) k_zsplat(const float* __restrict__ points, int N, Camera cam,
                                                   uint32_t* __restrict__ zkeys, int32_t* __restrict__ winner)
{
    const int b = blockIdx.y;
    const float* P = points + (size_t) b * 3 * N;
    uint32_t* Z = zkeys + (size_t) b * cam.H * cam.W;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    float x = P[i], y = P[(size_t) N + i], z = P[2 * (size_t) N + i];
    apply_shift(cam, x, y, z);
    Proj p;
    int idx = -1;
    if (project(cam, x, y, z, p)) {
        const int c = winner_corner(p);
        if (c >= 0) {
            const int cx = p.nwx + (c & 1), cy = p.nwy + (c >> 1);
            if (inside(cx, cy, cam.W, cam.H)) {
                idx = cy * cam.W + cx;
                atomicMin(&Z[idx], zkey_encode(p.err));
            }
        }
    }
    if (winner) winner[(size_t) b * N + i] = idx;
}

// ---------------------------------------------------------------------------------------
// generate_mask (common.py:689-830): which point owns each pixel of the z-splat.
// The reference's single launch is a race (compare, float atomicMin, atomicExch of the owner,
// mask updates: four separate steps per point); its serial-order result -- the oracle's -- is
// "the owner of a pixel is the FIRST point, in index order, that attains the pixel's minimal
// dblError" (a later point only takes over when strictly nearer, :755), a point's mask is 1 iff it
// is a final owner, and point 0, once an owner, is never cleared (:759 tests `pid > 0`).
// That is one native 64-bit atomic umin on (order-preserving key of dblError) << 32 | index,
// followed by a look-up per point: deterministic, no compare-then-act window.
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock) k_mask_splat(const float* __restrict__ points, const float* __restrict__ shift, int N,
                                                       Camera cam, unsigned long long* __restrict__ keys, int32_t* __restrict__ winner)
{
    const int b = blockIdx.y;
    const float* P = points + (size_t) b * 3 * N;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    // common.py:690  tensorInput + tensorShift: one fp32 add per component
    const float x = P[i] + shift[3 * b], y = P[(size_t) N + i] + shift[3 * b + 1], z = P[2 * (size_t) N + i] + shift[3 * b + 2];
    Proj p;
    int idx = -1;
    if (project(cam, x, y, z, p)) {
        const int c = winner_corner(p);
        if (c >= 0) {
            const int cx = p.nwx + (c & 1), cy = p.nwy + (c >> 1);
            // :755 `zee > dblError` against the initial 1e6: a point at or beyond it never owns anything
            if (inside(cx, cy, cam.W, cam.H) && 1000000.0f > p.err) {
                idx = cy * cam.W + cx;
                atomicMin(&keys[(size_t) b * cam.H * cam.W + idx], ((unsigned long long) zkey_encode(p.err) << 32) | (unsigned) i);
            }
        }
    }
    winner[(size_t) b * N + i] = idx;
}

__global__ void __launch_bounds__(kBlock) k_mask_resolve(const unsigned long long* __restrict__ keys, const int32_t* __restrict__ winner,
                                                         int N, int HW, float* __restrict__ masks)
{
    const int b = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const int idx = winner[(size_t) b * N + i];
    const bool owner = idx >= 0 && (uint32_t) keys[(size_t) b * HW + idx] == (uint32_t) i;
    masks[(size_t) b * N + i] = (owner || (i == 0 && idx >= 0)) ? 1.0f : 0.0f;
}

// optional views of the key table in the reference's own formats: zee (:692) and the owner table (:694, a float
// tensor of -1 whose bits are used as int memory)
__global__ void __launch_bounds__(kBlock) k_mask_tables(const unsigned long long* __restrict__ keys, size_t n, float* __restrict__ zee,
                                                        int32_t* __restrict__ ids)
{
    const size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned long long k = keys[i];
    const bool hit = k != ~0ull;
    if (zee) zee[i] = hit ? zkey_decode((uint32_t) (k >> 32)) : 1000000.0f;
    if (ids) ids[i] = hit ? (int32_t) (uint32_t) k : (int32_t) 0xBF800000u;
}

__global__ void __launch_bounds__(kBlock) k_fill_u64(unsigned long long* p, size_t n, unsigned long long v)
{
    const size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

// ---------------------------------------------------------------------------------------
// kernel_pointrender_updateDegrid (common.py:525-568), out of place (Jacobi schedule)
// ---------------------------------------------------------------------------------------
template <bool FROM_KEYS>
__global__ void __launch_bounds__(kBlock) k_degrid(const uint32_t* __restrict__ keys, const float* __restrict__ zin,
                                                   int W, int H, float* __restrict__ zout)
{
    const int b = blockIdx.y;
    const size_t base = (size_t) b * H * W;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= W * H) return;
    const int y = i / W, x = i - y * W;
    auto at = [&](int xx, int yy) -> float {
        const size_t o = base + (size_t) yy * W + xx;
        return FROM_KEYS ? zkey_decode(keys[o]) : zin[o];
    };
    zout[base + i] = degrid_pixel(x, y, W, H, at);
}

// The reference kernel executed ONE PIXEL AFTER THE OTHER in index order (what the host shim that produced
// tests/golden/*_serial vectors does; Gauss-Seidel): pixel (x, y) sees the new values of (x-1, y), (x-1, y-1),
// (x, y-1), (x+1, y-1) and the old values of its other four neighbours (common.py:556-566 reads and writes the same
// buffer).  Proof-of-fidelity entry, not a production path: one workgroup per image walks the skewed wavefront
// t = x + 2 y -- every pixel of a front depends only on fronts t-1, t-2, t-3 and reads nothing a same-front or
// earlier-front pixel still has to overwrite -- so the result IS the serial one, in W + 2 H barrier steps.
__global__ void __launch_bounds__(1024) k_degrid_serial(int W, int H, float* __restrict__ z)
{
    float* Z = z + (size_t) blockIdx.x * H * W;
    // agent-scope relaxed accesses: served by L2, never by a stale line of this CU's vector L1
    auto at = [&](int xx, int yy) -> float { return __hip_atomic_load(&Z[(size_t) yy * W + xx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    for (int t = 0; t <= (W - 1) + 2 * (H - 1); t++) {
        for (int y = threadIdx.x; y < H; y += blockDim.x) {
            const int x = t - 2 * y;
            if (x >= 0 && x < W) {
                const float v = degrid_pixel_one(x, y, W, H, at);
                __hip_atomic_store(&Z[(size_t) y * W + x], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        __threadfence();
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------
// kernel_pointrender_updateOutput (common.py:586-669), any channel count
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock) k_accumulate(const float* __restrict__ points, const float* __restrict__ data,
                                                       int N, int C, Camera cam, const float* __restrict__ zee,
                                                       float* __restrict__ acc)
{
    const int b = blockIdx.y;
    const size_t HW = (size_t) cam.H * cam.W;
    const float* P = points + (size_t) b * 3 * N;
    const float* D = data + (size_t) b * C * N;
    const float* Z = zee + (size_t) b * HW;
    float* A = acc + (size_t) b * (C + 1) * HW;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    float x = P[i], y = P[(size_t) N + i], z = P[2 * (size_t) N + i];
    apply_shift(cam, x, y, z);
    Proj p;
    if (!project(cam, x, y, z, p)) return;
    bool take[4];
    size_t px[4];
    bool any = false;
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const int cx = p.nwx + (c & 1), cy = p.nwy + (c >> 1);
        take[c] = false;
        px[c] = 0;
        if (inside(cx, cy, cam.W, cam.H)) {
            px[c] = (size_t) cy * cam.W + cx;
            take[c] = (double) p.err <= (double) Z[px[c]] + 1.0;      // :639
        }
        any |= take[c];
    }
    if (!any) return;
    for (int ch = 0; ch < C; ch++) {
        const float v = D[(size_t) ch * N + i];
#pragma unroll
        for (int c = 0; c < 4; c++)
            if (take[c]) atomic_add_f32(&A[ch * HW + px[c]], v * p.w[c]);   // :641 product rounded, then added
    }
#pragma unroll
    for (int c = 0; c < 4; c++)
        if (take[c]) atomic_add_f32(&A[C * HW + px[c]], p.w[c]);            // the appended `ones` channel, :429
}

// common.py:686
__global__ void __launch_bounds__(kBlock) k_normalize(const float* __restrict__ acc, int C, int HW,
                                                      float* __restrict__ render, float* __restrict__ existing)
{
    const int b = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= HW) return;
    const float* A = acc + (size_t) b * (C + 1) * HW;
    const float wsum = A[(size_t) C * HW + i];
    const float den = wsum + 0.0000001f;
    for (int ch = 0; ch < C; ch++) render[((size_t) b * C + ch) * HW + i] = A[(size_t) ch * HW + i] / den;
    existing[(size_t) b * HW + i] = wsum;
}

// ---------------------------------------------------------------------------------------
// kernel_discfill_updateOutput (common.py:838-924)
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock) k_fill(const float* __restrict__ input, const float* __restrict__ depth,
                                                 int C, int W, int H, FillDirs dirs, float* __restrict__ output)
{
    const int b = blockIdx.y;
    const int HW = W * H;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= HW) return;
    const float* Dp = depth + (size_t) b * HW;
    const float* In = input + (size_t) b * C * HW;
    float* Out = output + (size_t) b * C * HW;
    int src = i;
    if (!(Dp[i] > 0.0f)) {                                      // :850
        const int y = i / W, x = i - y * W;
        const int s = fill_source(dirs, x, y, W, H, [&](int xx, int yy) { return Dp[yy * W + xx]; });
        if (s >= 0) src = s;
    }
    for (int ch = 0; ch < C; ch++) Out[(size_t) ch * HW + i] = In[(size_t) ch * HW + src];    // :834 clone + :921-923
}

__global__ void __launch_bounds__(kBlock) k_frame_u8(const float* __restrict__ render, int HW, uint8_t* __restrict__ frame)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= HW) return;
    frame[(size_t) i * 3 + 0] = to_u8(render[i]);
    frame[(size_t) i * 3 + 1] = to_u8(render[(size_t) HW + i]);
    frame[(size_t) i * 3 + 2] = to_u8(render[(size_t) 2 * HW + i]);
}

// ---------------------------------------------------------------------------------------
// cv2.getRectSubPix + cv2.resize(INTER_LINEAR) on 8-bit HWC (common.py:256-257).
// Restated from the OpenCV algorithms as described in SURVEY.md B.7; UNPINNED.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ int cv_round(float v) { return (int) rintf(v); }

// resize INTER_LINEAR 8u: source index pair and 11-bit coefficient pair of destination index d.  OpenCV (resize.cpp)
// treats the two directions differently: a COLUMN tap outside the row is clamped and its fraction zeroed (xmin / xmax),
// a ROW tap outside the image only has its index clipped -- both taps then read the same row with weights (1 - f, f),
// which rounds differently from (1, 0) by up to one count in the first and last output rows.
__device__ __forceinline__ void resize_coeff(int d, double scale, int src_n, bool horizontal, int& s0, int& s1, int& c0, int& c1)
{
    float f = (float) ((d + 0.5) * scale - 0.5);
    int s = (int) floorf(f);
    f -= (float) s;
    if (horizontal) {
        if (s < 0) { f = 0.f; s = 0; }
        if (s >= src_n - 1) { f = 0.f; s = src_n - 1; }
    }
    s0 = min(max(s, 0), src_n - 1);
    s1 = min(max(s + 1, 0), src_n - 1);
    c0 = cv_round((1.f - f) * 2048.f);
    c1 = cv_round(f * 2048.f);
}

// 24-bit multiplies as the instructions (a 32-bit integer multiply is a quarter-rate instruction, and the compiler cannot see that a
// byte times an 11- or 17-bit weight fits 24 bits); _s: the first factor is wave-uniform (a scalar register)
__device__ __forceinline__ uint32_t mul_u24_v(uint32_t a, uint32_t b) { uint32_t r; asm("v_mul_u32_u24 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ uint32_t mad_u24_v(uint32_t a, uint32_t b, uint32_t c) { uint32_t r; asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ uint32_t mul_u24_s(uint32_t a, uint32_t b) { uint32_t r; asm("v_mul_u32_u24 %0, %1, %2" : "=v"(r) : "s"(a), "v"(b)); return r; }
__device__ __forceinline__ uint32_t mul_hi_u24_s(uint32_t a, uint32_t b) { uint32_t r; asm("v_mul_hi_u32_u24 %0, %1, %2" : "=v"(r) : "s"(a), "v"(b)); return r; }      // bits 47..32 of the product
__device__ __forceinline__ uint32_t mad_u24_s(uint32_t a, uint32_t b, uint32_t c) { uint32_t r; asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "s"(a), "v"(b), "v"(c)); return r; }

// One workgroup = a 64 x CR_TH tile of the output, in the two steps OpenCV takes but without the intermediate
// image: (0) the raw-frame rows the tile needs are staged in LDS with coalesced dword loads, (1) the
// getRectSubPix patch pixels under the tile (each the rounded 16-bit fixed-point blend of 2 x 2 raw pixels) are
// computed ONCE into LDS, (2) every output pixel blends 2 x 2 of those.  (One thread per output pixel doing
// all 16 raw taps itself was instruction-bound: 400 VALU instructions per wave, 14.5 us per 1024^2 frame;
// before the LDS staging it issued 48 byte gathers per pixel and took 25 us.)  The resize coefficients of the
// tile's columns and rows are computed once, by 64 + CR_TH threads, not per pixel.
// Round 5: next to the other launches of the frame loop what counts is its instruction count (a 64 x 8 tile: ~450 vector
// instructions per wave, 7.1 us per 1024^2 frame = 0.11 of the HBM roofline by its 6 HW bytes).  The tile is 64 x 16 now -- the
// coefficients, the staging and the patch's two extra rows are paid once per 16 rows instead of once per 8 -- and a wave walks DOWN
// its four output rows: two consecutive
// output rows share a patch row (the scale is <= 1), so the horizontal pass of a patch row is taken once and kept in registers
// (every test about rows is wave-uniform).  A patch that starts on a whole pixel (an odd crop of an even frame) IS the raw
// rectangle: step (1) is skipped and step (2) reads the staged raw rows.
#ifndef KBE_CROP_TH
#define KBE_CROP_TH 16      // (64 x 8 / 16 / 32 / 64: 43.1 / 44.7 / 44.2 k frames/s left in HBM; a launch on its own is a chain of waits and takes longer with taller tiles)
#endif
constexpr int CR_TW = 64, CR_TH = KBE_CROP_TH, CR_THREADS = 256, CR_WAVES = CR_THREADS / 64;
constexpr int CR_ROWS_PER_WAVE = CR_TH / CR_WAVES;
constexpr int CR_PROWS = CR_TH + 2, CR_PCOLS = CR_TW + 2;              // patch pixels under a tile (scale <= 1)
constexpr int CR_PBYTES = CR_PCOLS * 3, CR_PSTRIDE = (CR_PBYTES + 3) / 4 + 1;      // dwords per patch row in LDS
constexpr int CR_RROWS = CR_PROWS + 1, CR_RDW = ((CR_PCOLS + 1) * 3 + 3 + 3) / 4 + 1;  // raw rows / dwords per raw row
constexpr int CR_STAGE = (CR_RROWS + CR_WAVES - 1) / CR_WAVES;        // raw rows a wave stages
static_assert(CR_TH % CR_WAVES == 0 && CR_RDW <= 64 && CR_PBYTES <= CR_THREADS && CR_TW + CR_TH <= CR_THREADS && CR_PROWS + 1 <= CR_THREADS, "crop tile geometry");

__device__ __forceinline__ void crop_resize_body(const uint8_t* __restrict__ img, int W, int H, int cw, int ch_, double scale_x, double scale_y, uint8_t* __restrict__ out)
{
    __shared__ uint32_t s_raw[CR_RROWS][CR_RDW];
    __shared__ uint32_t s_patch[CR_PROWS][CR_PSTRIDE];
    __shared__ __attribute__((aligned(16))) uint32_t s_out[CR_TH][CR_TW * 3 / 4];   // finished pixels leave as dwords / 16-byte words (byte stores are slow)
    __shared__ int s_cx[4][CR_TW], s_cy[4][CR_TH];              // per column / row of the tile: s0, s1, c0, c1
    __shared__ int s_ro[CR_PROWS + 1];
    const int tid = threadIdx.x;
    const int bx = blockIdx.x * CR_TW, by = blockIdx.y * CR_TH;
    // getRectSubPix: top-left sample position and 16-bit fixed-point bilinear weights
    const float cx = (float) W / 2.0f - (float) (cw - 1) * 0.5f, cy = (float) H / 2.0f - (float) (ch_ - 1) * 0.5f;
    const int ipx = (int) floorf(cx), ipy = (int) floorf(cy);
    const float fa = cx - (float) ipx, fb = cy - (float) ipy;
    const int a11 = cv_round((1.f - fa) * (1.f - fb) * 65536.f), a12 = cv_round(fa * (1.f - fb) * 65536.f);
    const int a21 = cv_round((1.f - fa) * fb * 65536.f), a22 = cv_round(fa * fb * 65536.f);
    // an odd crop of an even frame (or the reverse) starts on a whole pixel: a11 = 65536 and the patch IS the raw rectangle
    // ((t * 65536 + 32768) >> 16 == t); uniform over the launch
    const bool whole = (a12 | a21 | a22) == 0;
    if (tid < CR_TW + CR_TH) {
        const bool col = tid < CR_TW;
        const int k = col ? tid : tid - CR_TW;
        int s0, s1, c0, c1;
        if (col) resize_coeff(min(bx + k, W - 1), scale_x, cw, true, s0, s1, c0, c1);         // scale = (double) cw / W, divided on the host
        else resize_coeff(min(by + k, H - 1), scale_y, ch_, false, s0, s1, c0, c1);
        int* t = col ? &s_cx[0][k] : &s_cy[0][k];
        const int stride = col ? CR_TW : CR_TH;
        t[0] = s0; t[stride] = s1; t[2 * stride] = c0; t[3 * stride] = c1;
    }
    __syncthreads();
    // patch rectangle under the tile, and the raw rectangle under that (+1 sub-pixel tap), replicate border as OpenCV does
    const int px_lo = s_cx[0][0], px_hi = s_cx[1][CR_TW - 1], py_lo = s_cy[0][0], py_hi = s_cy[1][CR_TH - 1];
    const int n_pc3 = (px_hi - px_lo + 1) * 3, n_pr = py_hi - py_lo + 1;                    // <= CR_PBYTES, <= CR_PROWS
    const int rx0 = min(max(ipx + px_lo, 0), W - 1), rx1 = min(max(ipx + px_hi + 1, 0), W - 1);
    const int ry0 = min(max(ipy + py_lo, 0), H - 1), ry1 = min(max(ipy + py_hi + 1, 0), H - 1);
    const int n_rows = ry1 - ry0 + 1;                                                       // <= CR_RROWS
    const size_t frame_bytes = (size_t) W * H * 3;
    // (0) stage the raw rows: wave w takes rows w, w + 4, w + 8, ..., lane = dword of the row, so everything about a row
    // is wave-uniform (scalar unit) and a lane only adds its offset; every load is issued before the first LDS
    // store (a load-store-load-store loop serialises the memory latencies)
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), ln = tid & 63;
    uint32_t stage[CR_STAGE];
#pragma unroll
    for (int k = 0; k < CR_STAGE; k++) {
        const int r = wv + CR_WAVES * k;                                // wave-uniform
        stage[k] = 0;
        if (r < n_rows) {
            const size_t b0 = ((size_t) (ry0 + r) * W + rx0) * 3, b1 = ((size_t) (ry0 + r) * W + rx1) * 3 + 3;
            const size_t a0 = b0 & ~(size_t) 3;
            const int n_dw = (int) ((b1 - a0 + 3) >> 2);                // <= CR_RDW
            const uint8_t* row = img + a0;
            if (a0 + 4 * (size_t) n_dw <= frame_bytes) {                // wave-uniform: the whole row segment is inside the frame
                if (ln < n_dw) stage[k] = ((const uint32_t*) row)[ln];
            } else if (ln < n_dw) {                                     // the very last bytes of the frame
                for (int t = 0; t < 4; t++) if (a0 + 4 * (size_t) ln + t < frame_bytes) stage[k] |= (uint32_t) row[4 * ln + t] << (8 * t);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < CR_STAGE; k++) {
        const int r = wv + CR_WAVES * k;
        if (r < CR_RROWS && ln < CR_RDW) s_raw[r][ln] = stage[k];
    }
    if (tid <= CR_PROWS) {
        // byte offset in s_raw of the raw row under patch row `tid` (row CR_PROWS: the last bottom tap); row y starts
        // at byte ((y * W + rx0) * 3) & 3 of its staging row (only the low bits matter: 32-bit arithmetic is enough)
        const int y = min(max(ipy + py_lo + tid, 0), H - 1);
        s_ro[tid] = min(y - ry0, CR_RROWS - 1) * (int) sizeof(s_raw[0]) + (int) ((((uint32_t) y * (uint32_t) W + (uint32_t) rx0) * 3u) & 3u);
    }
    __syncthreads();
    // (1) patch pixels: a thread owns one byte column (pixel channel) of the patch and walks down its rows; the
    // bottom taps of one row are the top taps of the next, so a row costs 2 LDS reads, 4 multiply-adds and the
    // rounding of cast_8u.  (One (row, byte) item per step with its own index arithmetic: 400 instructions.)
    const uint8_t* raw = (const uint8_t*) s_raw;
    uint8_t* patch = (uint8_t*) s_patch;
    if (!whole) {                                                       // uniform
        if (tid < n_pc3) {
            const int pc = tid / 3, c = tid - pc * 3;
            const int x0 = (min(max(ipx + px_lo + pc, 0), W - 1) - rx0) * 3 + c, x1 = (min(max(ipx + px_lo + pc + 1, 0), W - 1) - rx0) * 3 + c;
            int o = s_ro[0];
            int t0 = raw[o + x0], t1 = raw[o + x1];
            for (int pr = 0; pr < n_pr; pr++) {
                o = s_ro[pr + 1];
                const int u0 = raw[o + x0], u1 = raw[o + x1];
                // (bytes times 17-bit weights that sum to 65536: 24-bit multiply-adds, the sum below 2^24 + 2^15)
                const uint32_t t = mad_u24_s((uint32_t) a22, (uint32_t) u1, mad_u24_s((uint32_t) a21, (uint32_t) u0, mad_u24_s((uint32_t) a12, (uint32_t) t1, mad_u24_s((uint32_t) a11, (uint32_t) t0, 1u << 15))));
                patch[pr * (int) sizeof(s_patch[0]) + tid] = (uint8_t) (t >> 16);
                t0 = u0; t1 = u1;
            }
        }
        __syncthreads();
    }
    // (2) resize INTER_LINEAR: 2 x 2 patch pixels per output pixel.  A thread owns a column of the tile and walks down its wave's
    // CR_ROWS_PER_WAVE output rows; the horizontal pass of a patch row (x 2048, its low four bits dropped as OpenCV's vertical pass drops them) is
    // kept for the two rows last used -- consecutive output rows mostly need one new patch row, not two.  What a row of the source
    // is called: its byte offset in `src` (the patch, or the staged raw rows when the patch is the raw rectangle).
    const int col = tid & (CR_TW - 1);
    const uint8_t* const src = whole ? raw : (const uint8_t*) patch;
    int sx, sx1;
    if (whole) {
        sx = (min(max(ipx + s_cx[0][col], 0), W - 1) - rx0) * 3;
        sx1 = (min(max(ipx + s_cx[1][col], 0), W - 1) - rx0) * 3;
    } else {
        sx = (s_cx[0][col] - px_lo) * 3;
        sx1 = (s_cx[1][col] - px_lo) * 3;
    }
    const int ax0 = s_cx[2][col], ax1 = s_cx[3][col];
    auto row_offset = [&](int pr) -> int {                              // wave-uniform
        return whole ? s_ro[pr] : pr * (int) sizeof(s_patch[0]);
    };
    auto horizontal = [&](int pr, int (&h)[3]) {
        const int o = row_offset(pr);
#pragma unroll
        for (int c = 0; c < 3; c++) h[c] = (int) (mad_u24_v(src[o + sx1 + c], (uint32_t) ax1, mul_u24_v(src[o + sx + c], (uint32_t) ax0)) & ~15u);       // (r >> 4) << 4: see the vertical pass
    };
    int ka = -1, kb = -1;                                               // the patch rows whose horizontal pass `ha`, `hb` hold (scalars)
    int ha[3] = { 0, 0, 0 }, hb[3] = { 0, 0, 0 };
    uint8_t px[CR_ROWS_PER_WAVE][3];
#pragma unroll
    for (int m = 0; m < CR_ROWS_PER_WAVE; m++) {
        const int row = wv * CR_ROWS_PER_WAVE + m;
        const int y0 = __builtin_amdgcn_readfirstlane(s_cy[0][row] - py_lo), y1 = __builtin_amdgcn_readfirstlane(s_cy[1][row] - py_lo);
        const int by0 = __builtin_amdgcn_readfirstlane(s_cy[2][row]), by1 = __builtin_amdgcn_readfirstlane(s_cy[3][row]);
        if (ka != y0) {                                                 // uniform
            if (kb == y0) {
#pragma unroll
                for (int c = 0; c < 3; c++) ha[c] = hb[c];
            } else horizontal(y0, ha);
            ka = y0;
        }
        if (y1 == y0) {                                                 // a row tap clipped at the image edge: both taps read the same row
#pragma unroll
            for (int c = 0; c < 3; c++) hb[c] = ha[c];
            kb = y0;
        } else if (kb != y1) { horizontal(y1, hb); kb = y1; }
#pragma unroll
        for (int c = 0; c < 3; c++) {
            // vertical pass: OpenCV's ((by * (r >> 4)) >> 16) is the high word of the 48-bit product (by << 12) * ((r >> 4) << 4) -- one
            // v_mul_hi_u32_u24 (by <= 2048 and r < 2^20: both factors fit 24 bits) instead of a multiply and a shift.  The two weights sum
            // to 2048 (2049 at most, where both roundings go up) and r >> 4 <= 32640, so the sum is at most 1020 and the pixel at most
            // 255: OpenCV's saturation never acts
            const uint32_t v = (mul_hi_u24_s((uint32_t) by0 << 12, (uint32_t) ha[c]) + mul_hi_u24_s((uint32_t) by1 << 12, (uint32_t) hb[c]) + 2u) >> 2;
            px[m][c] = (uint8_t) v;
        }
        uint8_t* so = (uint8_t*) s_out[row] + col * 3;
        so[0] = px[m][0]; so[1] = px[m][1]; so[2] = px[m][2];
    }
    __syncthreads();
    if ((W & 15) == 0 && bx + CR_TW <= W && ((uintptr_t) out & 15) == 0) {
        // a tile row is 192 bytes: twelve 16-byte stores where the frame allows it (as tile_epilogue, kbe_tiles.h)
        constexpr int Q = CR_TW * 3 / 16;
        static_assert((CR_TW * 3) % 16 == 0 && sizeof(s_out[0]) % 16 == 0, "16-byte row stores");
        for (int i = tid; i < CR_TH * Q; i += CR_THREADS) {
            const int r = i / Q, k = i - r * Q;
            if (by + r < H) ((uint4*) (out + ((size_t) (by + r) * W + bx) * 3))[k] = ((const uint4*) s_out[r])[k];
        }
    } else if ((W & 3) == 0 && bx + CR_TW <= W) {
        constexpr int DW = CR_TW * 3 / 4;
        for (int i = tid; i < CR_TH * DW; i += CR_THREADS) {
            const int r = i / DW, k = i - r * DW;
            if (by + r < H) ((uint32_t*) (out + ((size_t) (by + r) * W + bx) * 3))[k] = s_out[r][k];
        }
    } else {
#pragma unroll
        for (int m = 0; m < CR_ROWS_PER_WAVE; m++) {
            const int dx = bx + col, dy = by + wv * CR_ROWS_PER_WAVE + m;
            if (dx < W && dy < H) {
                const size_t o = ((size_t) dy * W + dx) * 3;
                out[o] = px[m][0]; out[o + 1] = px[m][1]; out[o + 2] = px[m][2];
            }
        }
    }
}

__global__ void __launch_bounds__(CR_THREADS) k_crop_resize_u8(const uint8_t* __restrict__ img, int W, int H, int cw, int ch_, double scale_x, double scale_y,
                                                               uint8_t* __restrict__ out)
{
    crop_resize_body(img, W, H, cw, ch_, scale_x, scale_y, out);
}

// the same for up to four frames of the same size in one launch (blockIdx.z = the frame): the video loop's groups
struct CropJobs { const uint8_t* img[4]; uint8_t* out[4]; };
__global__ void __launch_bounds__(CR_THREADS) k_crop_resize_u8_group(CropJobs jobs, int W, int H, int cw, int ch_, double scale_x, double scale_y)
{
    crop_resize_body(jobs.img[blockIdx.z], W, H, cw, ch_, scale_x, scale_y, jobs.out[blockIdx.z]);
}

// ---------------------------------------------------------------------------------------
// torch glue
// ---------------------------------------------------------------------------------------

// torch.linspace(-0.5 n + 0.5, 0.5 n - 0.5, n)[i] (fp32, evaluated from both ends like ATen)
__device__ __forceinline__ float linspace_centered(int n, int i)
{
    const float start = (float) ((-0.5 * n) + 0.5), end = (float) ((0.5 * n) - 0.5);
    if (n == 1) return start;
    const float step = (end - start) / (float) (n - 1);
    return (i < n / 2) ? start + step * (float) i : end - step * (float) (n - 1 - i);
}

// depth_to_points (common.py:382-392)
__global__ void __launch_bounds__(kBlock) k_depth_to_points(const float* __restrict__ depth, const float* __restrict__ valid,
                                                            int W, int H, float inv_focal, float* __restrict__ points)
{
    const int b = blockIdx.y;
    const int HW = W * H;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= HW) return;
    const int y = i / W, x = i - y * W;
    float d = depth[(size_t) b * HW + i];
    if (valid) d = d * valid[(size_t) b * HW + i];
    const float u = linspace_centered(W, x) * inv_focal;
    const float v = linspace_centered(H, y) * inv_focal;
    float* P = points + (size_t) b * 3 * HW;
    P[i] = d * u;
    P[(size_t) HW + i] = d * v;
    P[(size_t) 2 * HW + i] = d;
}

// process_shift, materialised (common.py:104-109)
__global__ void __launch_bounds__(kBlock) k_shift_points(const float* __restrict__ points, int N, Camera cam,
                                                         float* __restrict__ out)
{
    const int b = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const float* P = points + (size_t) b * 3 * N;
    float* O = out + (size_t) b * 3 * N;
    float x = P[i], y = P[(size_t) N + i], z = P[2 * (size_t) N + i];
    apply_shift(cam, x, y, z);
    O[i] = x;
    O[(size_t) N + i] = y;
    O[2 * (size_t) N + i] = z;
}

// spatial_filter 'laplacian' (common.py:397-409); SCALE: divide the input by *scale first and
// emit the (|lap| < thr) mask instead (common.py:70, pointcloud_inpainting.py:193)
template <bool MASK>
__global__ void __launch_bounds__(kBlock) k_laplacian(const float* __restrict__ in, const float* __restrict__ scale,
                                                      int W, int H, float thr, float* __restrict__ out)
{
    const int p = blockIdx.y;
    const int HW = W * H;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= HW) return;
    const float* I = in + (size_t) p * HW;
    const int y = i / W, x = i - y * W;
    const int ym = max(y - 1, 0), yp = min(y + 1, H - 1), xm = max(x - 1, 0), xp = min(x + 1, W - 1);
    const float s = MASK ? *scale : 1.0f;
    auto ld = [&](int yy, int xx) { const float v = I[yy * W + xx]; return MASK ? v / s : v; };
    float a = 0.0f;
    a = __builtin_fmaf(-1.0f, ld(ym, x), a);       // taps [0][1], [0][2], [1][0], [1][1], [2][0] (common.py:401-405)
    a = __builtin_fmaf(-1.0f, ld(ym, xp), a);
    a = __builtin_fmaf(-1.0f, ld(y, xm), a);
    a = __builtin_fmaf(4.0f, ld(y, x), a);
    a = __builtin_fmaf(-1.0f, ld(yp, xm), a);
    out[(size_t) p * HW + i] = MASK ? (fabsf(a) < thr ? 1.0f : 0.0f) : a;
}

// spatial_filter 'median-3' / 'median-5' (common.py:411-421): reflect pad, lower median by rank
template <int K>
__global__ void __launch_bounds__(kBlock) k_median(const float* __restrict__ in, int W, int H, float* __restrict__ out)
{
    constexpr int R = K / 2, NN = K * K;
    const int p = blockIdx.y;
    const int HW = W * H;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= HW) return;
    const float* I = in + (size_t) p * HW;
    const int y = i / W, x = i - y * W;
    float v[NN];
#pragma unroll
    for (int dy = -R; dy <= R; dy++) {
        int yy = y + dy;
        yy = yy < 0 ? -yy : (yy >= H ? 2 * (H - 1) - yy : yy);
#pragma unroll
        for (int dx = -R; dx <= R; dx++) {
            int xx = x + dx;
            xx = xx < 0 ? -xx : (xx >= W ? 2 * (W - 1) - xx : xx);
            v[(dy + R) * K + (dx + R)] = I[yy * W + xx];
        }
    }
    // element of rank (NN-1)/2 in a stable order (value, then index)
    float med = v[0];
#pragma unroll
    for (int a = 0; a < NN; a++) {
        int rank = 0;
#pragma unroll
        for (int b = 0; b < NN; b++) rank += (v[b] < v[a]) | ((v[b] == v[a]) & (b < a));
        if (rank == (NN - 1) / 2) med = v[a];
    }
    out[(size_t) p * HW + i] = med;
}

// PartialConv2d bookkeeping (utils/partial_conv.py:62-77).  mask has Cm channels: Cin (as the
// reference materialises it), 1 (every input channel carries the same mask: the sum over channels
// is Cin times the single-channel box sum, exact for 0/1 masks), or is NULL (no mask given: all ones).
__global__ void __launch_bounds__(kBlock) k_pconv_epilogue(const float* __restrict__ raw, const float* __restrict__ bias,
                                                           const float* __restrict__ mask, int Cm, int Cin, int H, int W, int Cout,
                                                           int Ho, int Wo, int k, int stride, int pad,
                                                           float* __restrict__ out, float* __restrict__ um_out,
                                                           const float* __restrict__ slope, const float* __restrict__ residual, int add_bias)
{
    const int b = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Ho * Wo) return;
    const int oy = i / Wo, ox = i - oy * Wo;
    float msum = 0.0f;
    const int planes = mask ? Cm : 1;
    for (int ci = 0; ci < planes; ci++) {
        const float* M = mask ? mask + ((size_t) b * Cm + ci) * H * W : nullptr;
        for (int ky = 0; ky < k; ky++) {
            const int iy = oy * stride - pad + ky;
            if (iy < 0 || iy >= H) continue;
            for (int kx = 0; kx < k; kx++) {
                const int ix = ox * stride - pad + kx;
                if (ix < 0 || ix >= W) continue;
                msum += M ? M[(size_t) iy * W + ix] : 1.0f;
            }
        }
    }
    if (planes != Cin) msum = (float) Cin * msum;
    float ratio = (float) (Cin * k * k) / (msum + 1e-8f);
    const float um = msum < 0.0f ? 0.0f : (msum > 1.0f ? 1.0f : msum);
    ratio = ratio * um;
    if (um_out) um_out[(size_t) b * Ho * Wo + i] = um;
    for (int co = 0; co < Cout; co++) {
        const size_t o = ((size_t) b * Cout + co) * Ho * Wo + i;
        const float r = raw[o];
        float v;
        if (bias) {
            const float bv = bias[co];
            // (add_bias: the convolution ran WITHOUT its bias -- MIOpen's Winograd kernels take none, PyTorch would add it in a
            // pass of its own -- and the sum the reference's formula starts from, rounding included, is formed here)
            const float rb = add_bias ? r + bv : r;
            v = ((rb - bv) * ratio + bv) * um;
        } else {
            v = r * ratio;
        }
        // what follows the layer in the GridNet's blocks, in the same pass: `+ skip` (partial_inpainting.py: Basic), or the next
        // layer's PReLU (its mask multiplication needs nothing: v is 0 wherever um -- the next layer's mask -- is)
        if (residual) v = v + residual[o];
        if (slope) v = v > 0.0f ? v : slope[co] * v;          // (torch.nn.functional.prelu, to the sign of a zero)
        out[o] = v;
    }
}

// prelu(x) * mask in one pass (partial_inpainting.py: `p_relu_1`, then `input * mask_in` of PartialConv2d.forward, :61): x, out
// [B,C,HW] (may alias), slope [C], mask [B,1,HW] or NULL
__global__ void __launch_bounds__(kBlock) k_prelu_mask(const float* __restrict__ x, const float* __restrict__ slope, const float* __restrict__ mask,
                                                       int C, size_t HW, float* __restrict__ out)
{
    const int b = blockIdx.y;
    const size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= HW) return;
    const float m = mask ? mask[(size_t) b * HW + i] : 1.0f;
    for (int c = 0; c < C; c++) {
        const size_t o = ((size_t) b * C + c) * HW + i;
        const float v = x[o];
        out[o] = (v > 0.0f ? v : slope[c] * v) * m;
    }
}

// What PyTorch runs around a convolution of the networks as separate element-wise passes -- the bias add (MIOpen's Winograd kernels
// take none), the PReLU that follows, the block's `+ skip` and the grid's `+ the stream from the neighbouring row`
// (models/pointcloud_inpainting.py:16-52, :133-172 of the reference) -- in ONE pass over the convolution's output:
//   out = act(x + bias[c]) + res1 + res2      act = PReLU with slope[c], or none; bias, slope, res1, res2 may each be NULL.
// x, out, res1, res2 [B,C,HW] (out may alias x); four pixels per thread when HW is a multiple of four.  The additions are in
// the order the separate passes make them ((x + bias) first, then the residuals left to right).
template <bool VEC>
__global__ void __launch_bounds__(kBlock) k_bias_act(const float* __restrict__ x, const float* __restrict__ bias, const float* __restrict__ slope,
                                                     const float* __restrict__ res1, const float* __restrict__ res2, size_t HW, int C, float* __restrict__ out)
{
    const int bc = blockIdx.y;
    const int c = bc % C;
    const float b = bias ? bias[c] : 0.0f, sl = slope ? slope[c] : 0.0f;
    const size_t base = (size_t) bc * HW;
    auto one = [&](float v, float r1, float r2) {
        if (bias) v = v + b;
        if (slope) v = v > 0.0f ? v : sl * v;
        if (res1) v = v + r1;
        if (res2) v = v + r2;
        return v;
    };
    if (VEC) {
        const size_t i = ((size_t) blockIdx.x * blockDim.x + threadIdx.x) * 4;
        if (i >= HW) return;
        const float4 v = *(const float4*) (x + base + i);
        const float4 r1 = res1 ? *(const float4*) (res1 + base + i) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        const float4 r2 = res2 ? *(const float4*) (res2 + base + i) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        *(float4*) (out + base + i) = make_float4(one(v.x, r1.x, r2.x), one(v.y, r1.y, r2.y), one(v.z, r1.z, r2.z), one(v.w, r1.w, r2.w));
    } else {
        const size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
        if (i >= HW) return;
        out[base + i] = one(x[base + i], res1 ? res1[base + i] : 0.0f, res2 ? res2[base + i] : 0.0f);
    }
}

// nn.Upsample(scale_factor=2, mode='bilinear', align_corners=False) and the PReLU behind it (the head of an `Upsample` block,
// models/pointcloud_inpainting.py:54-80) in one pass: x [BC,H,W] -> out [BC,2H,2W].  The source position and the weights as
// PyTorch's upsample_bilinear2d computes them (source = 0.5 (dst + 0.5) - 0.5, clamped at 0; the neighbour clamped at the edge;
// h0 (w0 v00 + w1 v01) + h1 (w0 v10 + w1 v11)); two output pixels of a row per thread.
__global__ void __launch_bounds__(kBlock) k_upsample2x_act(const float* __restrict__ x, const float* __restrict__ slope, int C, int H, int W, float* __restrict__ out)
{
    const int bc = blockIdx.z, y2 = blockIdx.y;
    const int xs = blockIdx.x * blockDim.x + threadIdx.x;       // source column: output columns 2 xs, 2 xs + 1
    if (xs >= W) return;
    const float hr = fmaxf(0.5f * ((float) y2 + 0.5f) - 0.5f, 0.0f);
    const int h1 = (int) hr, hp = h1 < H - 1 ? 1 : 0;
    const float hl1 = hr - (float) h1, hl0 = 1.0f - hl1;
    const float* r0 = x + ((size_t) bc * H + h1) * W;
    const float* r1 = r0 + (size_t) hp * W;
    const int xm = xs > 0 ? xs - 1 : 0, xp = xs < W - 1 ? xs + 1 : xs;
    const float a0 = r0[xm], a1 = r0[xs], a2 = r0[xp], b0 = r1[xm], b1 = r1[xs], b2 = r1[xp];
    const float sl = slope ? slope[bc % C] : 0.0f;
    float o[2];
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const int w2 = 2 * xs + k;
        const float wr = fmaxf(0.5f * ((float) w2 + 0.5f) - 0.5f, 0.0f);
        const int w1 = (int) wr;
        const float wl1 = wr - (float) w1, wl0 = 1.0f - wl1;
        // w1 is xs - 1 (k = 0, xs > 0) or xs; its neighbour w1 + 1 clamped at W - 1
        const bool left = w1 < xs;
        const float v00 = left ? a0 : a1, v01 = left ? a1 : a2, v10 = left ? b0 : b1, v11 = left ? b1 : b2;
        float v = hl0 * (wl0 * v00 + wl1 * v01) + hl1 * (wl0 * v10 + wl1 * v11);
        if (slope) v = v > 0.0f ? v : sl * v;
        o[k] = v;
    }
    *(float2*) (out + ((size_t) bc * 2 * H + y2) * 2 * W + 2 * xs) = make_float2(o[0], o[1]);
}

}  // namespace

// =======================================================================================
// extern "C" entry points (include/kbe.h)
// =======================================================================================

extern "C" {

int kbe_abi_version(void) { return KBE_ABI_VERSION; }

const char* kbe_last_error(void) { return g_err; }

int kbe_device_info(int device, char* name, int cap)
{
    hipDeviceProp_t prop;
    const hipError_t e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess) return fail(KBE_E_DEVICE, "hipGetDeviceProperties", e);
    if (name && cap > 0) { strncpy(name, prop.gcnArchName, (size_t) cap - 1); name[cap - 1] = 0; }
    return prop.multiProcessorCount;
}

int kbe_selftest_err(const float* z, size_t n, double focal, double baseline, float* fast, float* exact, kbe_stream_t stream)
{
    KBE_REQUIRE(z && fast && exact && n > 0, "kbe_selftest_err: bad arguments");
    const Camera cam = make_camera(1, 1, focal, baseline, nullptr);
    hipLaunchKernelGGL(k_selftest_err, dim3(2048), dim3(kBlock), 0, (hipStream_t) stream, z, n, cam, fast, exact);
    return launched("kbe_selftest_err");
}

int kbe_selftest_division(const float* num, const float* den, size_t n, float* fast, float* ieee, kbe_stream_t stream)
{
    KBE_REQUIRE(num && den && fast && ieee && n > 0, "kbe_selftest_division: bad arguments");
    hipLaunchKernelGGL(k_selftest_division, dim3(2048), dim3(kBlock), 0, (hipStream_t) stream, num, den, n, fast, ieee);
    return launched("kbe_selftest_division");
}

int kbe_zkeys_clear(uint32_t* zkeys, size_t n, kbe_stream_t stream)
{
    KBE_REQUIRE(zkeys && n > 0, "kbe_zkeys_clear: bad arguments");
    const unsigned grid = (unsigned) ((n + kBlock - 1) / kBlock < 4096 ? (n + kBlock - 1) / kBlock : 4096);
    hipLaunchKernelGGL(k_fill_u32, dim3(grid), dim3(kBlock), 0, (hipStream_t) stream, zkeys, n, KBE_ZKEY_EMPTY);
    return launched("kbe_zkeys_clear");
}

int kbe_zsplat(const float* points, int B, int N, int W, int H, double focal, double baseline,
               const float* shift3, uint32_t* zkeys, int32_t* winner, kbe_stream_t stream)
{
    KBE_REQUIRE(zkeys && B > 0 && N >= 0 && W > 0 && H > 0, "kbe_zsplat: bad arguments");
    if (N == 0) return KBE_OK;      // an empty cloud touches nothing (points may be NULL then)
    KBE_REQUIRE(points, "kbe_zsplat: points is NULL");
    const Camera cam = make_camera(W, H, focal, baseline, shift3);
    hipLaunchKernelGGL(k_zsplat, dim3(blocks_for(N), B), dim3(kBlock), 0, (hipStream_t) stream, points, N, cam, zkeys, winner);
    return launched("kbe_zsplat");
}

int kbe_generate_mask(const float* points, const float* shift, int B, int N, int W, int H, double focal, double baseline,
                      unsigned long long* keys, int32_t* winner, float* masks, float* zee, int32_t* ids, kbe_stream_t stream)
{
    KBE_REQUIRE(B > 0 && N >= 0 && W > 0 && H > 0 && keys && ((uintptr_t) keys & 7) == 0, "kbe_generate_mask: bad arguments");
    const hipStream_t s = (hipStream_t) stream;
    const size_t n_pix = (size_t) B * W * H;
    hipLaunchKernelGGL(k_fill_u64, dim3((unsigned) ((n_pix + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, keys, n_pix, ~0ull);
    if (N > 0) {
        KBE_REQUIRE(points && shift && winner && masks, "kbe_generate_mask: NULL buffer");
        const Camera cam = make_camera(W, H, focal, baseline, nullptr);
        hipLaunchKernelGGL(k_mask_splat, dim3(blocks_for(N), B), dim3(kBlock), 0, s, points, shift, N, cam, keys, winner);
        hipLaunchKernelGGL(k_mask_resolve, dim3(blocks_for(N), B), dim3(kBlock), 0, s, keys, winner, N, W * H, masks);
    }
    if (zee || ids)
        hipLaunchKernelGGL(k_mask_tables, dim3((unsigned) ((n_pix + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, keys, n_pix, zee, ids);
    return launched("kbe_generate_mask");
}

int kbe_zkeys_decode(const uint32_t* zkeys, size_t n, float* zee, kbe_stream_t stream)
{
    KBE_REQUIRE(zkeys && zee && n > 0, "kbe_zkeys_decode: bad arguments");
    const unsigned grid = (unsigned) ((n + kBlock - 1) / kBlock < 4096 ? (n + kBlock - 1) / kBlock : 4096);
    hipLaunchKernelGGL(k_zkeys_decode, dim3(grid), dim3(kBlock), 0, (hipStream_t) stream, zkeys, n, zee);
    return launched("kbe_zkeys_decode");
}

int kbe_degrid(const uint32_t* zkeys, const float* zee_in_f32, int B, int W, int H, float* zee_out, kbe_stream_t stream)
{
    KBE_REQUIRE((zkeys || zee_in_f32) && zee_out && B > 0 && W > 0 && H > 0, "kbe_degrid: bad arguments");
    const dim3 grid(blocks_for((size_t) W * H), B);
    if (zee_in_f32)
        hipLaunchKernelGGL(k_degrid<false>, grid, dim3(kBlock), 0, (hipStream_t) stream, zkeys, zee_in_f32, W, H, zee_out);
    else
        hipLaunchKernelGGL(k_degrid<true>, grid, dim3(kBlock), 0, (hipStream_t) stream, zkeys, zee_in_f32, W, H, zee_out);
    return launched("kbe_degrid");
}

int kbe_degrid_serial(const uint32_t* zkeys, const float* zee_in_f32, int B, int W, int H, float* zee_out, kbe_stream_t stream)
{
    KBE_REQUIRE((zkeys || zee_in_f32) && zee_out && B > 0 && W > 0 && H > 0, "kbe_degrid_serial: bad arguments");
    const size_t n = (size_t) B * W * H;
    const hipStream_t s = (hipStream_t) stream;
    if (zee_in_f32) {
        if (zee_in_f32 != zee_out) {
            const hipError_t e = hipMemcpyAsync(zee_out, zee_in_f32, n * sizeof(float), hipMemcpyDeviceToDevice, s);
            if (e != hipSuccess) return fail(KBE_E_LAUNCH, "kbe_degrid_serial: hipMemcpyAsync", e);
        }
    } else {
        int rc = kbe_zkeys_decode(zkeys, n, zee_out, stream);
        if (rc != KBE_OK) return rc;
    }
    hipLaunchKernelGGL(k_degrid_serial, dim3(B), dim3(H < 1024 ? ((H + 63) / 64) * 64 : 1024), 0, s, W, H, zee_out);
    return launched("kbe_degrid_serial");
}

int kbe_accumulate(const float* points, const float* data, int B, int N, int C, const float* zee, int W, int H,
                   double focal, double baseline, const float* shift3, float* acc, kbe_stream_t stream)
{
    KBE_REQUIRE(zee && acc && B > 0 && N >= 0 && C >= 0 && W > 0 && H > 0, "kbe_accumulate: bad arguments");
    if (N == 0) return KBE_OK;
    KBE_REQUIRE(points && (data || C == 0), "kbe_accumulate: points/data is NULL");
    const Camera cam = make_camera(W, H, focal, baseline, shift3);
    hipLaunchKernelGGL(k_accumulate, dim3(blocks_for(N), B), dim3(kBlock), 0, (hipStream_t) stream, points, data, N, C, cam,
                       zee, acc);
    return launched("kbe_accumulate");
}

int kbe_normalize(const float* acc, int B, int C, int W, int H, float* render, float* existing, kbe_stream_t stream)
{
    KBE_REQUIRE(acc && render && existing && B > 0 && C >= 0 && W > 0 && H > 0, "kbe_normalize: bad arguments");
    hipLaunchKernelGGL(k_normalize, dim3(blocks_for((size_t) W * H), B), dim3(kBlock), 0, (hipStream_t) stream, acc, C,
                       W * H, render, existing);
    return launched("kbe_normalize");
}

int kbe_render_pointcloud(const float* points, const float* data, int B, int N, int C, int W, int H, double focal,
                          double baseline, uint32_t* zkeys, float* zee, float* acc, float* render, float* existing,
                          kbe_stream_t stream)
{
    KBE_REQUIRE(zkeys && zee && acc, "kbe_render_pointcloud: scratch missing");
    int rc;
    if ((rc = kbe_zkeys_clear(zkeys, (size_t) B * H * W, stream))) return rc;
    const hipError_t e = hipMemsetAsync(acc, 0, sizeof(float) * (size_t) B * (C + 1) * H * W, (hipStream_t) stream);
    if (e != hipSuccess) return fail(KBE_E_LAUNCH, "hipMemsetAsync(acc)", e);
    if ((rc = kbe_zsplat(points, B, N, W, H, focal, baseline, nullptr, zkeys, nullptr, stream))) return rc;
    if ((rc = kbe_degrid(zkeys, nullptr, B, W, H, zee, stream))) return rc;
    if ((rc = kbe_accumulate(points, data, B, N, C, zee, W, H, focal, baseline, nullptr, acc, stream))) return rc;
    return kbe_normalize(acc, B, C, W, H, render, existing, stream);
}

int kbe_fill_disocclusion(const float* input, const float* depth, int B, int C, int W, int H, float* output,
                          kbe_stream_t stream)
{
    KBE_REQUIRE(input && depth && output && B > 0 && C > 0 && W > 0 && H > 0, "kbe_fill_disocclusion: bad arguments");
    static const FillDirs dirs = make_fill_dirs();
    hipLaunchKernelGGL(k_fill, dim3(blocks_for((size_t) W * H), B), dim3(kBlock), 0, (hipStream_t) stream, input, depth, C,
                       W, H, dirs, output);
    return launched("kbe_fill_disocclusion");
}

int kbe_frame_u8(const float* render_chw, int W, int H, uint8_t* frame_hwc, kbe_stream_t stream)
{
    KBE_REQUIRE(render_chw && frame_hwc && W > 0 && H > 0, "kbe_frame_u8: bad arguments");
    hipLaunchKernelGGL(k_frame_u8, dim3(blocks_for((size_t) W * H)), dim3(kBlock), 0, (hipStream_t) stream, render_chw,
                       W * H, frame_hwc);
    return launched("kbe_frame_u8");
}

int kbe_crop_resize_u8(const uint8_t* frame_hwc, int W, int H, int crop_w, int crop_h, uint8_t* out_hwc,
                       kbe_stream_t stream)
{
    KBE_REQUIRE(frame_hwc && out_hwc && W > 0 && H > 0 && crop_w > 0 && crop_h > 0 && crop_w <= W && crop_h <= H,
                "kbe_crop_resize_u8: bad arguments");
    hipLaunchKernelGGL(k_crop_resize_u8, dim3((W + CR_TW - 1) / CR_TW, (H + CR_TH - 1) / CR_TH), dim3(CR_THREADS), 0,
                       (hipStream_t) stream, frame_hwc, W, H, crop_w, crop_h, (double) crop_w / W, (double) crop_h / H, out_hwc);
    return launched("kbe_crop_resize_u8");
}

}  // extern "C"

namespace kbe {
// kbe_crop_resize_u8 for n <= 4 frames of the same size in one launch (kbe_render_video's groups)
int crop_resize_group(int n, const uint8_t* const* frames, int W, int H, int crop_w, int crop_h, uint8_t* const* outs, hipStream_t stream)
{
    if (n == 1) return kbe_crop_resize_u8(frames[0], W, H, crop_w, crop_h, outs[0], (kbe_stream_t) stream);
    CropJobs jobs;
    for (int k = 0; k < 4; k++) { jobs.img[k] = frames[k < n ? k : 0]; jobs.out[k] = outs[k < n ? k : 0]; }
    hipLaunchKernelGGL(k_crop_resize_u8_group, dim3((W + CR_TW - 1) / CR_TW, (H + CR_TH - 1) / CR_TH, n), dim3(CR_THREADS), 0, stream, jobs, W, H, crop_w, crop_h, (double) crop_w / W, (double) crop_h / H);
    return launched("kbe_crop_resize_u8 (group)");
}
}  // namespace kbe

extern "C" {

int kbe_depth_to_points(const float* depth, const float* valid, int B, int W, int H, double focal, float* points,
                        kbe_stream_t stream)
{
    KBE_REQUIRE(depth && points && B > 0 && W > 0 && H > 0, "kbe_depth_to_points: bad arguments");
    const float inv = (float) (1.0 / focal);
    hipLaunchKernelGGL(k_depth_to_points, dim3(blocks_for((size_t) W * H), B), dim3(kBlock), 0, (hipStream_t) stream, depth,
                       valid, W, H, inv, points);
    return launched("kbe_depth_to_points");
}

int kbe_shift_points(const float* points, int B, int N, const float* shift3, float* out, kbe_stream_t stream)
{
    KBE_REQUIRE(shift3 && B > 0 && N >= 0, "kbe_shift_points: bad arguments");
    if (N == 0) return KBE_OK;
    KBE_REQUIRE(points && out, "kbe_shift_points: points/out is NULL");
    const Camera cam = make_camera(1, 1, 1.0, 1.0, shift3);
    hipLaunchKernelGGL(k_shift_points, dim3(blocks_for(N), B), dim3(kBlock), 0, (hipStream_t) stream, points, N, cam, out);
    return launched("kbe_shift_points");
}

int kbe_spatial_filter(const float* in, int planes, int W, int H, int kind, float* out, kbe_stream_t stream)
{
    KBE_REQUIRE(in && out && planes > 0 && W > 0 && H > 0, "kbe_spatial_filter: bad arguments");
    const dim3 grid(blocks_for((size_t) W * H), planes);
    const hipStream_t s = (hipStream_t) stream;
    if (kind == 0) {
        hipLaunchKernelGGL(k_laplacian<false>, grid, dim3(kBlock), 0, s, in, (const float*) nullptr, W, H, 0.0f, out);
    } else if (kind == 3) {
        KBE_REQUIRE(W >= 2 && H >= 2, "median-3 needs W, H >= 2 (reflect pad)");
        hipLaunchKernelGGL(k_median<3>, grid, dim3(kBlock), 0, s, in, W, H, out);
    } else if (kind == 5) {
        KBE_REQUIRE(W >= 3 && H >= 3, "median-5 needs W, H >= 3 (reflect pad)");
        hipLaunchKernelGGL(k_median<5>, grid, dim3(kBlock), 0, s, in, W, H, out);
    } else {
        return fail(KBE_E_INVALID, "kbe_spatial_filter: kind must be 0, 3 or 5");
    }
    return launched("kbe_spatial_filter");
}

int kbe_laplacian_valid(const float* in, const float* scale_dev, int planes, int W, int H, float threshold, float* valid,
                        kbe_stream_t stream)
{
    KBE_REQUIRE(in && scale_dev && valid && planes > 0 && W > 0 && H > 0, "kbe_laplacian_valid: bad arguments");
    hipLaunchKernelGGL(k_laplacian<true>, dim3(blocks_for((size_t) W * H), planes), dim3(kBlock), 0, (hipStream_t) stream, in,
                       scale_dev, W, H, threshold, valid);
    return launched("kbe_laplacian_valid");
}

int kbe_pconv_epilogue(const float* raw, const float* bias, const float* mask, int mask_channels, int B, int Cin, int H, int W,
                       int Cout, int Ho, int Wo, int k, int stride, int pad, float* out, float* um, const float* prelu_slope,
                       const float* residual, int raw_without_bias, kbe_stream_t stream)
{
    KBE_REQUIRE(raw && out && B > 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0 && k > 0 && stride > 0 && pad >= 0,
                "kbe_pconv_epilogue: bad arguments");
    KBE_REQUIRE(!mask || mask_channels == 1 || mask_channels == Cin, "kbe_pconv_epilogue: mask must have 1 or Cin channels");
    hipLaunchKernelGGL(k_pconv_epilogue, dim3(blocks_for((size_t) Ho * Wo), B), dim3(kBlock), 0, (hipStream_t) stream, raw,
                       bias, mask, mask_channels, Cin, H, W, Cout, Ho, Wo, k, stride, pad, out, um, prelu_slope, residual, raw_without_bias);
    return launched("kbe_pconv_epilogue");
}

int kbe_prelu_mask(const float* x, const float* slope, const float* mask, int B, int C, int H, int W, float* out, kbe_stream_t stream)
{
    KBE_REQUIRE(x && slope && out && B > 0 && C > 0 && H > 0 && W > 0, "kbe_prelu_mask: bad arguments");
    hipLaunchKernelGGL(k_prelu_mask, dim3(blocks_for((size_t) H * W), B), dim3(kBlock), 0, (hipStream_t) stream, x, slope, mask, C, (size_t) H * W, out);
    return launched("kbe_prelu_mask");
}

int kbe_bias_act(const float* x, const float* bias, const float* slope, const float* res1, const float* res2, int B, int C, int H, int W, float* out,
                 kbe_stream_t stream)
{
    KBE_REQUIRE(x && out && B > 0 && C > 0 && H > 0 && W > 0 && (size_t) B * C <= 65535, "kbe_bias_act: bad arguments");
    const size_t HW = (size_t) H * W;
    const bool vec = HW % 4 == 0 && (((uintptr_t) x | (uintptr_t) out | (uintptr_t) res1 | (uintptr_t) res2) & 15) == 0;
    if (vec) hipLaunchKernelGGL(k_bias_act<true>, dim3(blocks_for(HW / 4), B * C), dim3(kBlock), 0, (hipStream_t) stream, x, bias, slope, res1, res2, HW, C, out);
    else hipLaunchKernelGGL(k_bias_act<false>, dim3(blocks_for(HW), B * C), dim3(kBlock), 0, (hipStream_t) stream, x, bias, slope, res1, res2, HW, C, out);
    return launched("kbe_bias_act");
}

int kbe_upsample2x_act(const float* x, const float* slope, int B, int C, int H, int W, float* out, kbe_stream_t stream)
{
    KBE_REQUIRE(x && out && x != out && B > 0 && C > 0 && H > 0 && W > 0 && (size_t) B * C <= 65535 && 2 * H <= 65535, "kbe_upsample2x_act: bad arguments");
    hipLaunchKernelGGL(k_upsample2x_act, dim3(blocks_for((size_t) W), 2 * H, B * C), dim3(kBlock), 0, (hipStream_t) stream, x, slope, C, H, W, out);
    return launched("kbe_upsample2x_act");
}

}  // extern "C"
