/* A SECOND, independent restatement of the two OpenCV calls at /root/reference/utils/common.py:256-257 on 8-bit,
 * 3-channel images -- cv2.getRectSubPix(frame, (pw, ph), (W/2, H/2)) and cv2.resize(patch, (W, H), INTER_LINEAR) --
 * written from the structure of OpenCV's published implementation (modules/imgproc/src/samplers.cpp:
 * getRectSubPix_Cn_<uchar, uchar, int, scale_fixpt, cast_8u> with adjustRect; modules/imgproc/src/resize.cpp:
 * resizeGeneric_ with HResizeLinear<uchar, int, short, 2048> and the 8-bit VResizeLinear), scalar loops, row
 * buffers and all -- not from the prose in SURVEY.md B.7 that the oracle's numpy version and the HIP kernel follow.
 * Test infrastructure only (tests/test_opencv_restatement.py cross-checks the three).  OpenCV itself is not in the
 * image: agreement among restatements is NOT parity with cv2 -- that row stays "parity unpinned".
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static int cv_round(double v) { return (int) lrint(v); }       /* cvRound: round half to even */
static int cv_floor(double v) { int i = (int) v; return i - (i > v); }

/* samplers.cpp: adjustRect -- the part of the window that lies inside the image, and where reading starts */
typedef struct { int x, y, width, height; } Rect;
static const uint8_t* adjust_rect(const uint8_t* src, size_t src_step, int pix_size, int src_w, int src_h, int win_w, int win_h,
                                  int ipx, int ipy, Rect* r)
{
    Rect rect;
    if (ipx >= 0) { src += (size_t) ipx * pix_size; rect.x = 0; } else { rect.x = -ipx; if (rect.x > win_w) rect.x = win_w; }
    if (ipx < src_w - win_w) rect.width = win_w;
    else { rect.width = src_w - ipx - 1; if (rect.width < 0) { src += (size_t) rect.width * pix_size; rect.width = 0; } }
    if (ipy >= 0) { src += (size_t) ipy * src_step; rect.y = 0; } else rect.y = -ipy;
    if (ipy < src_h - win_h) rect.height = win_h;
    else { rect.height = src_h - ipy - 1; if (rect.height < 0) { src += (size_t) rect.height * src_step; rect.height = 0; } }
    *r = rect;
    return src - (size_t) rect.x * pix_size;
}

/* samplers.cpp: getRectSubPix_Cn_ with scale_fixpt (x * 2^16, rounded) and cast_8u ((x + 2^15) >> 16) */
static void get_rect_sub_pix_8u_c3(const uint8_t* src, int src_w, int src_h, uint8_t* dst, int win_w, int win_h, float cx, float cy)
{
    const int cn = 3;
    const size_t src_step = (size_t) src_w * cn, dst_step = (size_t) win_w * cn;
    cx -= (win_w - 1) * 0.5f;
    cy -= (win_h - 1) * 0.5f;
    const int ipx = cv_floor(cx), ipy = cv_floor(cy);
    const float a = cx - ipx, b = cy - ipy;
    const int a11 = cv_round((1.f - a) * (1.f - b) * 65536.f), a12 = cv_round(a * (1.f - b) * 65536.f);
    const int a21 = cv_round((1.f - a) * b * 65536.f), a22 = cv_round(a * b * 65536.f);
    if (0 <= ipx && ipx < src_w - win_w && 0 <= ipy && ipy < src_h - win_h) {
        /* the window is totally inside the image */
        const uint8_t* s = src + (size_t) ipy * src_step + (size_t) ipx * cn;
        for (int i = 0; i < win_h; i++, s += src_step, dst += dst_step)
            for (int j = 0; j < win_w * cn; j++)
                dst[j] = (uint8_t) ((s[j] * a11 + s[j + cn] * a12 + s[j + src_step] * a21 + s[j + src_step + cn] * a22 + (1 << 15)) >> 16);
        return;
    }
    Rect r;
    const uint8_t* s = adjust_rect(src, src_step, cn, src_w, src_h, win_w, win_h, ipx, ipy, &r);
    for (int i = 0; i < win_h; i++) {
        const uint8_t* s2 = s + src_step;
        if (i < r.y || i >= r.height) s2 -= src_step;           /* above / below the image: the row is used twice */
        for (int c = 0; c < cn; c++) {
            const int s0 = s[r.x * cn + c] * (a11 + a12) + s2[r.x * cn + c] * (a21 + a22);        /* left of the image */
            for (int j = 0; j < r.x; j++) dst[j * cn + c] = (uint8_t) ((s0 + (1 << 15)) >> 16);
        }
        int j = r.x;
        for (; j < r.width; j++)
            for (int c = 0; c < cn; c++) {
                const int k = j * cn + c;
                dst[k] = (uint8_t) ((s[k] * a11 + s[k + cn] * a12 + s2[k] * a21 + s2[k + cn] * a22 + (1 << 15)) >> 16);
            }
        for (int c = 0; c < cn; c++) {
            const int s0 = s[r.width * cn + c] * (a11 + a12) + s2[r.width * cn + c] * (a21 + a22);      /* right of the image */
            for (int jj = r.width; jj < win_w; jj++) dst[jj * cn + c] = (uint8_t) ((s0 + (1 << 15)) >> 16);
        }
        if (i < r.height) s = s2;
        dst += dst_step;
    }
}

/* resize.cpp: resizeGeneric_ for INTER_LINEAR on 8UC3: xofs / alpha / yofs / beta tables, two row buffers that are
 * reused while the source row pair stays the same, HResizeLinear (plain copy * 2048 from xmax on), the 8-bit VResizeLinear */
static void resize_linear_8u_c3(const uint8_t* src, int sw, int sh, uint8_t* dst, int dw, int dh)
{
    const int cn = 3, ONE = 2048;
    const double scale_x = (double) sw / dw, scale_y = (double) sh / dh;
    int* xofs = (int*) malloc(sizeof(int) * dw * cn);
    short* alpha = (short*) malloc(sizeof(short) * dw * cn * 2);
    int* yofs = (int*) malloc(sizeof(int) * dh);
    short* beta = (short*) malloc(sizeof(short) * dh * 2);
    int xmin = 0, xmax = dw;
    for (int dx = 0; dx < dw; dx++) {
        float fx = (float) ((dx + 0.5) * scale_x - 0.5);
        int sx = cv_floor(fx);
        fx -= sx;
        if (sx < 0) { xmin = dx + 1; fx = 0; sx = 0; }
        if (sx >= sw - 1) { if (xmax > dx) xmax = dx; fx = 0; sx = sw - 1; }
        for (int k = 0; k < cn; k++) {
            xofs[dx * cn + k] = sx * cn + k;
            alpha[(dx * cn + k) * 2] = (short) cv_round((1.f - fx) * ONE);
            alpha[(dx * cn + k) * 2 + 1] = (short) cv_round(fx * ONE);
        }
    }
    for (int dy = 0; dy < dh; dy++) {
        float fy = (float) ((dy + 0.5) * scale_y - 0.5);
        int sy = cv_floor(fy);
        fy -= sy;
        yofs[dy] = sy;
        beta[dy * 2] = (short) cv_round((1.f - fy) * ONE);
        beta[dy * 2 + 1] = (short) cv_round(fy * ONE);
    }
    (void) xmin;
    xmax *= cn;
    const int width = dw * cn;
    int* rows[2] = { (int*) malloc(sizeof(int) * width), (int*) malloc(sizeof(int) * width) };
    int prev_sy[2] = { -1, -1 };
    for (int dy = 0; dy < dh; dy++) {
        int sy0 = yofs[dy], k0 = 2, k1 = 0;
        const uint8_t* srows[2];
        for (int k = 0; k < 2; k++) {
            int sy = sy0 + k;
            if (sy < 0) sy = 0;
            if (sy > sh - 1) sy = sh - 1;                       /* clip(sy0 - ksize2 + 1 + k, 0, ssize.height) */
            for (k1 = k1 > k ? k1 : k; k1 < 2; k1++)
                if (k1 < 2 && sy == prev_sy[k1]) {              /* the row is already in a buffer: move it into place */
                    if (k1 > k) memcpy(rows[k], rows[k1], sizeof(int) * width);
                    break;
                }
            if (k1 == 2) k0 = k0 < k ? k0 : k;                  /* first row that has to be computed */
            srows[k] = src + (size_t) sy * sw * cn;
            prev_sy[k] = sy;
        }
        if (k0 < 2)
            for (int k = k0; k < 2; k++) {                      /* HResizeLinear */
                const uint8_t* S = srows[k];
                int* D = rows[k];
                int dx = 0;
                for (; dx < xmax; dx++) { const int sx = xofs[dx]; D[dx] = S[sx] * alpha[dx * 2] + S[sx + cn] * alpha[dx * 2 + 1]; }
                for (; dx < width; dx++) D[dx] = S[xofs[dx]] * ONE;
            }
        const short b0 = beta[dy * 2], b1 = beta[dy * 2 + 1];  /* VResizeLinear<uchar, int, short, FixedPtCast<int, uchar, 22>> */
        uint8_t* D = dst + (size_t) dy * width;
        for (int x = 0; x < width; x++) {
            const int v = (((b0 * (rows[0][x] >> 4)) >> 16) + ((b1 * (rows[1][x] >> 4)) >> 16) + 2) >> 2;
            D[x] = (uint8_t) (v < 0 ? 0 : (v > 255 ? 255 : v));
        }
    }
    free(xofs); free(alpha); free(yofs); free(beta); free(rows[0]); free(rows[1]);
}

/* common.py:256-257 for one frame [H, W, 3] */
void cvr_crop_resize_u8(const uint8_t* frame, int W, int H, int crop_w, int crop_h, uint8_t* out)
{
    uint8_t* patch = (uint8_t*) malloc((size_t) crop_w * crop_h * 3);
    get_rect_sub_pix_8u_c3(frame, W, H, patch, crop_w, crop_h, (float) (W / 2.0), (float) (H / 2.0));
    resize_linear_8u_c3(patch, crop_w, crop_h, out, W, H);
    free(patch);
}

/* the two steps on their own, for the border cases the frame loop never produces (windows that stick out of the image) */
void cvr_get_rect_sub_pix(const uint8_t* src, int W, int H, uint8_t* dst, int win_w, int win_h, float cx, float cy)
{
    get_rect_sub_pix_8u_c3(src, W, H, dst, win_w, win_h, cx, cy);
}

void cvr_resize_linear(const uint8_t* src, int sw, int sh, uint8_t* dst, int dw, int dh)
{
    resize_linear_8u_c3(src, sw, sh, dst, dw, dh);
}
