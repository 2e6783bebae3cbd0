// kbe_fill.h -- the ray search of kernel_discfill_updateOutput (common.py:838-924), shared by the
// per-pixel generic kernel and the per-hole cooperative kernel.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

#pragma clang fp contract(off)

namespace kbe {

struct FillDirs { float x[16], y[16]; };

inline FillDirs make_fill_dirs()
{
    // common.py:859-867: the direction table, normalised in fp32 on the host with the same
    // IEEE operations (sqrtf, divide) the kernel text performs per thread
    const float dx[16] = { -1, 0, 1, 1, -1, 1, 2, 2, -2, -1, 1, 2, 3, 3, 3, 3 };
    const float dy[16] = { 1, 1, 1, 0, 2, 2, 1, -1, 3, 3, 3, 3, 2, 1, -1, -2 };
    FillDirs d;
    for (int i = 0; i < 16; i++) {
        volatile float n = sqrtf((dx[i] * dx[i]) + (dy[i] * dy[i]));
        d.x[i] = dx[i] / n;
        d.y[i] = dy[i] / n;
    }
    return d;
}

// Finds the fill source of hole pixel (x, y): returns the linear index of the pixel to copy
// from, or -1 when every direction leaves the image on one side (common.py:913-919).
template <class DepthAt>
__device__ __forceinline__ int fill_source(const FillDirs& dirs, int x, int y, int W, int H, DepthAt depth_at)
{
    float shortest = 1000000.0f;
    int fx = -1, fy = -1;
    for (int d = 0; d < 16; d++) {
        const float ddx = dirs.x[d], ddy = dirs.y[d];
        float ax = (float) x, ay = (float) y;
        int iax, iay;
        float da = 0.0f;
        for (;;) {                                              // :876-883
            ax -= ddx; iax = (int) roundf(ax);
            ay -= ddy; iay = (int) roundf(ay);
            if ((iax < 0) | (iax >= W) | (iay < 0) | (iay >= H)) break;
            da = depth_at(iax, iay);
            if (da > 0.0f) break;
        }
        if ((iax < 0) | (iax >= W) | (iay < 0) | (iay >= H)) continue;      // :884-885
        float bx = (float) x, by = (float) y;
        int ibx, iby;
        float db = 0.0f;
        for (;;) {                                              // :887-894
            bx += ddx; ibx = (int) roundf(bx);
            by += ddy; iby = (int) roundf(by);
            if ((ibx < 0) | (ibx >= W) | (iby < 0) | (iby >= H)) break;
            db = depth_at(ibx, iby);
            if (db > 0.0f) break;
        }
        if ((ibx < 0) | (ibx >= W) | (iby < 0) | (iby >= H)) continue;      // :895-896
        const float ex = (float) (ibx - iax), ey = (float) (iby - iay);
        const float dist = sqrtf(ex * ex + ey * ey);            // :898 (exact small integers)
        if (shortest > dist) {                                  // :900
            fx = iax; fy = iay;
            if (da < db) { fx = ibx; fy = iby; }                // :904 the farther (background) end
            shortest = dist;
        }
    }
    return (fx < 0 || fy < 0) ? -1 : fy * W + fx;
}


}  // namespace kbe
