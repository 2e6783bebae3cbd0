// kbe_host.h -- host-side helpers shared by the translation units of libkbe_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdio.h>

#include "kbe.h"
#include "kbe_device.h"

namespace kbe {

extern thread_local char g_err[256];      // defined in kbe_hip.hip; read through kbe_last_error()

inline int fail(int code, const char* what, hipError_t e = hipSuccess)
{
    snprintf(g_err, sizeof(g_err), "%s%s%s", what, e != hipSuccess ? ": " : "", e != hipSuccess ? hipGetErrorString(e) : "");
    return code;
}

inline int launched(const char* what)
{
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? KBE_OK : fail(KBE_E_LAUNCH, what, e);
}

inline Camera make_camera(int W, int H, double focal, double baseline, const float* shift3)
{
    Camera c;
    c.focal_f = (float) focal;
    c.fb = focal * baseline;
    c.fb_f = (float) c.fb;
    c.half_w = 0.5 * (double) W;
    c.half_h = 0.5 * (double) H;
    c.cx_f = (float) (c.half_w - 0.5);
    c.cy_f = (float) (c.half_h - 0.5);
    c.fp32_centre = W >= 2 && H >= 2 && (double) c.cx_f == c.half_w - 0.5 && (double) c.cy_f == c.half_h - 0.5;
    c.W = W;
    c.H = H;
    c.has_shift = shift3 != nullptr;
    c.sx = shift3 ? shift3[0] : 0.0f;
    c.sy = shift3 ? shift3[1] : 0.0f;
    c.sz = shift3 ? shift3[2] : 0.0f;
    return c;
}

constexpr int kBlock = 256;

inline unsigned blocks_for(size_t n, int per_block = kBlock)
{
    return (unsigned) ((n + per_block - 1) / per_block);
}

}  // namespace kbe

#define KBE_REQUIRE(cond, what) do { if (!(cond)) return kbe::fail(KBE_E_INVALID, what); } while (0)
