/*
 * kbe_jpeg.h -- C ABI of libkbe_jpeg.so: the frame writers' baseline-JPEG encoder (HOST code; ken-burns-effect_amd/csrc/kbe_jpeg.c).
 *
 * What it replaces.  The reference hands its finished uint8 frames to moviepy, which pipes them into an ffmpeg process
 * (`ImageSequenceClip(...).write_videofile(..., codec='mpeg4')`, /root/reference/utils/pipeline.py:130-134).  Where there is
 * no ffmpeg binary this package writes the video itself as Motion-JPEG (ken-burns-effect_amd/pipeline.py: write_mjpeg_mp4 /
 * write_mjpeg_avi); these entry points encode that stream's frames -- independent of one another -- on host threads.
 * Not part of the GPU hot path and not needed by it: pipeline.py falls back to Pillow (one frame at a time) when the library
 * is missing, and says so.
 *
 * Output: baseline sequential DCT (ISO/IEC 10918-1), 8-bit, JFIF YCbCr 4:2:0, Annex K.1 tables scaled by the IJG quality
 * rule, Annex K.3 Huffman tables -- what Pillow's save(format='JPEG', quality=q) writes by default, table for table.
 * Frames are interleaved RGB rows of `stride_bytes` bytes (>= 3 w).  Thread-safe, no global state.
 */
#ifndef KBE_JPEG_H
#define KBE_JPEG_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__)
#define KBE_JPEG_API __attribute__((visibility("default")))
#else
#define KBE_JPEG_API
#endif

enum { KBE_JPEG_OK = 0, KBE_JPEG_E_INVALID = -1, KBE_JPEG_E_SPACE = -2 /* `cap` too small: ask kbe_jpeg_bound */ };

/* bytes that hold any w x h frame's JPEG */
KBE_JPEG_API size_t kbe_jpeg_bound(int w, int h);
/* one frame -> out[0 .. *size) */
KBE_JPEG_API int kbe_jpeg_encode(const uint8_t* rgb, int w, int h, int stride_bytes, int quality, uint8_t* out, size_t cap, size_t* size);
/* n frames of one size -> outs[i][0 .. sizes[i]), each buffer `cap` bytes, on `threads` host threads (the caller's among them) */
KBE_JPEG_API int kbe_jpeg_encode_batch(const uint8_t* const* rgb, int n, int w, int h, int stride_bytes, int quality, uint8_t* const* outs, size_t cap,
                                       size_t* sizes, int threads);

#ifdef __cplusplus
}
#endif
#endif
