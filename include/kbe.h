/*
 * kbe.h -- C ABI of libkbe_hip.so, the MI355X (gfx950) novel-view render kernels.
 *
 * What this replaces.  The reference has no FFI layer; its de-facto kernel boundary is
 *     launch_kernel(name, preprocess_kernel(src, vars))(grid, block, args=[n, ptr...], stream)
 * (/root/reference/utils/common.py:267-380, call sites :516-521, :578-582, :680-684, :930-934):
 * CUDA source strings JIT-compiled by CuPy/NVRTC per (shape, focal, baseline) and launched on
 * raw `tensor.data_ptr()` device pointers on torch's current stream.  Every entry point below
 * is the compiled-once equivalent of one such launch (or of the torch glue between two of
 * them): plain device pointers, sizes as runtime ints, focal/baseline as `double` (they are
 * pasted into the reference kernels as double literals and take part in fp64 sub-expressions,
 * SURVEY.md Appendix B), and the HIP stream the work is enqueued on.
 *
 * Conventions
 *   - All pointers are DEVICE pointers to contiguous fp32 (unless typed otherwise) in the
 *     reference's layouts: points [B,3,N], data [B,C,N], z-buffer [B,1,H,W], images [B,C,H,W].
 *   - The caller owns every buffer, including scratch; nothing here allocates or frees device memory or
 *     synchronises.  All launches are asynchronous on `stream` (a hipStream_t; NULL = default).
 *     ONE exception, stated here so that no caller is surprised: kbe_render_video with lanes > 1 creates
 *     (and destroys before it returns) the HIP events that fork its lane streams from `stream` and join
 *     them back -- host-side objects, no device memory, no blocking wait.
 *   - Return value: KBE_OK, or a negative KBE_E_* for invalid arguments / launch failures
 *     (hipGetLastError after the launch).  Nothing throws across this boundary.
 *   - Re-entrant: safe with one process per GPU or one stream per thread.  No global state but one item: the pool of HSA
 *     signals kbe_render_video's KBE_VIDEO_SDMA hand-off draws from (mutex-guarded; see the flag).  No environment variable is read.
 *   - Inputs must be finite.  A point whose projection overflows int range is dropped (the
 *     reference's behaviour there is platform-defined; see DESIGN.md "Deviations").
 *   - Sizes: W*H < 2^31 pixels (2^30 for the frame loop); the frame loop (kbe_render_frame*, kbe_render_video, kbe_render_pointcloud_tiled)
 *     takes clouds of up to 2^30 points (2^28 on the packed cloud's route: 32-bit byte offsets) and rasters with W, H < 2^24 (its index
 *     arithmetic uses 24-bit multiplies).
 *
 * Numerical contract: identical to oracle/kbe_oracle.c (which is pinned bit-for-bit to the
 * reference kernel text): z-buffer and winner indices bit-exact; degrid uses the out-of-place
 * (Jacobi) schedule; accumulation order is the hardware's atomic order (last-ulp differences).
 */
#ifndef KBE_H
#define KBE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KBE_ABI_VERSION 11

/* the library is built with -fvisibility=hidden; only these entry points are exported */
#if defined(__GNUC__)
#define KBE_API __attribute__((visibility("default")))
#else
#define KBE_API
#endif

enum {
    KBE_OK = 0,
    KBE_E_INVALID = -1,   /* null pointer, non-positive size, unsupported kernel size ...   */
    KBE_E_LAUNCH = -2,    /* hipGetLastError() != hipSuccess after a launch                  */
    KBE_E_DEVICE = -3     /* not a gfx950 device / runtime query failed                      */
};

typedef void* kbe_stream_t;   /* hipStream_t */

/* z-buffer cells are kept as order-preserving uint32 keys of the fp32 `dblError` so that the
   reference's CAS-loop float atomicMin (common.py:275-283) becomes one native atomic. */
#define KBE_ZKEY_EMPTY 0xC9742400u   /* key of 1000000.0f, the z-buffer's initial value (:430) */

KBE_API int kbe_abi_version(void);
/* static string describing the last failing HIP call of the calling thread ("" if none) */
KBE_API const char* kbe_last_error(void);
/* fills name (<= cap bytes) with the device's gcnArchName, returns CU count or KBE_E_DEVICE */
KBE_API int kbe_device_info(int device, char* name, int cap);

/* Self-test hook: dblError = (float)(1e6 - F*B / (z + 1e-7)) (common.py:470) of n depths, once through the
   literal fp64 expression (`exact`) and once through the division-free fast path the frame loop uses
   (`fast`); the two must agree bit for bit. */
KBE_API int kbe_selftest_err(const float* z, size_t n, double focal, double baseline, float* fast, float* exact,
                             kbe_stream_t stream);

/* Self-test hook: n quotients num[i] / den[i], once as the compiler's IEEE division (`ieee`) and once through the
   eight-instruction sequence the frame loop uses where it knows the operands' range (`fast`: the projection's
   (F - z) / -z of common.py:457-459 and the reciprocal of the weight sum of :686).  For operands with |den| in
   [2^-100, 2^100] and num = 0 or |num| in the same range the two must agree bit for bit. */
KBE_API int kbe_selftest_division(const float* num, const float* den, size_t n, float* fast, float* ieee,
                                  kbe_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * render_pointcloud, stage by stage  (common.py:428-686)
 * ------------------------------------------------------------------------------------- */

/* zkeys[n] = KBE_ZKEY_EMPTY  -- replaces `new_zeros(...).fill_(1000000.0)` (common.py:430) */
KBE_API int kbe_zkeys_clear(uint32_t* zkeys, size_t n, kbe_stream_t stream);

/* kernel_pointrender_updateZee (common.py:435-507): project every point, pick the corner
   with the largest bilinear weight, min-splat its dblError.  If `shift3` is non-NULL (HOST
   pointer to 3 floats) the camera shift of process_shift (common.py:104-109) is applied to
   each point on the fly: x' = x * (z / (z + 1e-7f)) + sx, ... so the shifted cloud is never
   materialised.  `winner` (optional, [B,N] int32) receives each point's target pixel index
   y*W+x or -1 -- the "z-buffer index" of the parity contract. */
KBE_API int kbe_zsplat(const float* points, int B, int N, int W, int H, double focal, double baseline,
               const float* shift3, uint32_t* zkeys, int32_t* winner, kbe_stream_t stream);

/* zkeys -> fp32 z-buffer without degrid (test/debug view of the pre-degrid buffer) */
KBE_API int kbe_zkeys_decode(const uint32_t* zkeys, size_t n, float* zee, kbe_stream_t stream);

/* kernel_pointrender_updateDegrid (common.py:525-568), out of place: reads keys, writes the
   degridded fp32 z-buffer.  `zee_in_f32`, if non-NULL, is used instead of the keys. */
KBE_API int kbe_degrid(const uint32_t* zkeys, const float* zee_in_f32, int B, int W, int H, float* zee_out,
               kbe_stream_t stream);

/* The same kernel under the SERIAL schedule (pixels one after the other in index order, in place: what the
   reference kernel text does when executed by a host shim, and what tests/golden's `*_serial` vectors hold).  The
   reference's in-place update (common.py:556-566) is a race on a GPU; kbe_degrid above is the normative,
   deterministic out-of-place schedule.  This entry exists to tie the HIP path to reference-run vectors bit for bit
   (tests); one workgroup per image, W + 2 H barrier steps: not a production path.  zee_out may alias zee_in_f32. */
KBE_API int kbe_degrid_serial(const uint32_t* zkeys, const float* zee_in_f32, int B, int W, int H, float* zee_out,
                              kbe_stream_t stream);

/* kernel_pointrender_updateOutput (common.py:586-669): z-tested bilinear accumulation of
   C data channels plus the weight channel into acc [B,C+1,H,W] (zeroed by the caller,
   common.py:431).  data may be NULL only when C == 0. */
KBE_API int kbe_accumulate(const float* points, const float* data, int B, int N, int C, const float* zee,
                   int W, int H, double focal, double baseline, const float* shift3, float* acc,
                   kbe_stream_t stream);

/* common.py:686: render = acc[:, :C] / (acc[:, C:] + 1e-7f), existing = acc[:, C:] */
KBE_API int kbe_normalize(const float* acc, int B, int C, int W, int H, float* render, float* existing,
                  kbe_stream_t stream);

/* The whole of render_pointcloud.  scratch: zkeys [B*H*W] u32, zee [B*H*W] f32,
   acc [B*(C+1)*H*W] f32 (contents on entry are irrelevant). */
KBE_API int kbe_render_pointcloud(const float* points, const float* data, int B, int N, int C, int W, int H,
                          double focal, double baseline, uint32_t* zkeys, float* zee, float* acc,
                          float* render, float* existing, kbe_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * fill_disocclusion  (common.py:833-937)
 * ------------------------------------------------------------------------------------- */
KBE_API int kbe_fill_disocclusion(const float* input, const float* depth, int B, int C, int W, int H,
                          float* output, kbe_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * The frame loop of process_kenburns (common.py:222-260) on the RESIDENT point cloud.
 *
 * One output frame = process_shift -> render_pointcloud([image; depth]) -> fill_disocclusion
 * -> uint8 HWC (common.py:238-255), as three launches and no floating-point atomics:
 *   1. project: every point is shifted and projected ONCE; its dblError is min-splatted into
 *      the z-buffer (one native global atomic umin on an order-preserving key) and a 16-byte
 *      record {ox, oy, dblError, point index} is appended to the bucket of each 32x16 target
 *      tile whose pixels it can colour (wave-aggregated appends);
 *   2. tiles: one workgroup per target tile loads the tile's z-buffer (+1 px halo), degrids it
 *      in LDS, threads the tile's records into per-pixel lists in LDS and lets every pixel
 *      gather, z-test and accumulate its contributions in registers; normalise, hole mask and
 *      uint8 conversion follow in the same thread; pixels leave with plain coalesced stores;
 *   3. fill: one half-wave per hole pixel (16 directions x 2 ends in parallel); the same
 *      launch resets the z-buffer and the bucket counters for the next frame.
 * The cloud is used in place, in the reference's layout: points [3,N] (tensorInpaPoints),
 * image [3,N] (tensorInpaImage), depth [N] (tensorInpaDepth).  Any point order is correct;
 * spatially coherent order (e.g. raster) is faster.
 *
 * scratch: kbe_frame_scratch_bytes(W, H, N) bytes (N = points of the largest cloud rendered with it: a scratch set ends
 *          with 12 bytes per point for kbe_render_frame_fused; 0 when only the bucket route is used), initialised ONCE with
 *          kbe_frame_scratch_init;
 *          every kbe_render_frame call leaves it ready for the next one.
 *   frame_u8      [H,W,3]   out
 *   render_f32    [4,H,W]   optional: the filled float render (parity checks)
 *   existing_f32  [H*W]     optional: the accumulated weight (tensorExisting)
 *   zee_f32       [H*W]     optional: the degridded z-buffer the accumulation tested against
 *   zee_pre_f32   [H*W]     optional: the z-buffer before degrid (bit-exact contract)
 * ------------------------------------------------------------------------------------- */
KBE_API size_t kbe_frame_scratch_bytes(int W, int H, int N);
KBE_API int kbe_frame_scratch_init(void* scratch, int W, int H, kbe_stream_t stream);
/* The same for `n` sets `stride` bytes apart (stride >= kbe_frame_scratch_bytes(W, H), a multiple of 16), in ONE launch: a video on a
   new cloud initialises 16-32 of them. */
KBE_API int kbe_frame_scratch_init_sets(void* scratch, size_t stride, int n, int W, int H, kbe_stream_t stream);

KBE_API int kbe_render_frame(const float* points, const float* image, const float* depth, int N, int W, int H,
                             double focal, double baseline, const float* shift3, void* scratch,
                             uint8_t* frame_u8, float* render_f32, float* existing_f32, float* zee_f32,
                             float* zee_pre_f32, kbe_stream_t stream);

/* The same with (a) a subset of the three launches (measurement aid: bench.py times each launch
 * by differencing; only KBE_STAGE_PROJECT | KBE_STAGE_TILES | KBE_STAGE_FILL leaves the scratch
 * clean -- never use a subset for real frames) and (b) `fill_rect`: HOST pointer to
 * {x0, y0, x1, y1} (inclusive) or NULL.  Holes outside the rectangle keep their rendered value
 * instead of being filled.  A hole is never the source of another pixel's fill (sources are valid
 * pixels of the un-filled render, common.py:921-923), so when the caller is going to crop the
 * frame (common.py:256-257) skipping the holes the crop discards changes nothing inside the
 * crop window; and (c) a layout hint: the first `raster_n` points of the cloud are a row-major
 * raster `raster_w` wide (the image pixels, as process_kenburns builds tensorInpaPoints); 0, 0 if
 * unknown.  The hint only changes how points are assigned to waves (32x8 patches), never the
 * result. */
#define KBE_STAGE_PROJECT 1
#define KBE_STAGE_TILES 2
#define KBE_STAGE_FILL 4
/* with KBE_STAGE_FILL: the hole-fill schedule.  Default: one half-wave per hole.  _BY_COUNT: one lane per hole for
 * frames with more than ~49 k holes (less work, longer dependent chains: what kbe_render_video uses when frames of
 * several lanes overlap).  _PER_LANE / _PER_HALFWAVE force one.  The results are identical. */
#define KBE_STAGE_FILL_PER_LANE 8
#define KBE_STAGE_FILL_PER_HALFWAVE 16
#define KBE_STAGE_FILL_BY_COUNT 32
/* with _PER_LANE or _BY_COUNT: frames with very many holes (>= ~49 k; no inpainting: a dolly zoom, a raw cloud) first get
 * tables (one more launch, k_hole_dist; it returns at once when the frame has fewer holes): the Chebyshev distance of
 * every pixel, and of every 8 x 8 block, to the nearest valid one -- a ray at distance D takes ~D - 1 steps at once, the
 * reference's fp32 sums taken on the integer mantissa -- and per direction and line across the image where along the
 * line valid pixels can be -- a direction with nothing on one side of the hole is skipped without walking to the image
 * border.  The same positions are tested in the end, the result is identical.  Frames with a side above 11 000 pixels
 * ignore the flag. */
#define KBE_STAGE_FILL_DIST 512
/* kbe_render_frame_stages: the scratch holds two z-buffers, A and B.  Without these flags a frame stands alone: it splats
 * into A and its fill launch resets A and the bucket counters.  Consecutive frames of a video alternate instead:
 * _ZBUF_A = splat into A, the TILE launch clears B (and the bucket counters); _ZBUF_B = splat into B, the tile launch
 * clears A -- the reset then costs no launch and no pass over the z-buffer.  Rules: A must be empty when a stand-alone
 * or _ZBUF_A frame starts (it is after kbe_frame_scratch_init, after a stand-alone frame and after a _ZBUF_B frame), B
 * when a _ZBUF_B frame starts (it is after a _ZBUF_A frame); so a sequence is A, B, A, B, ... and must not END on
 * _ZBUF_A (render the last frame stand-alone instead).  kbe_render_video does all of this itself. */
#define KBE_STAGE_ZBUF_A 128
#define KBE_STAGE_ZBUF_B 256
/* kbe_render_frame_fused with parity -1 only: do not zero the hole counters first (timing aid; the frames of such a run
 * are not valid) */
#define KBE_STAGE_KEEP_HOLE_COUNT 64
/* the fused route's tile launches exist in two builds (kbe_fused.hip): LEAN (608 records per tile in LDS, six workgroups per CU)
 * and ROOMY (736, five).  Default: by the cloud's density (lean up to 1.125 points per pixel).  These force one -- for tests and
 * measurements; same frames.  (Until ABI 9 the library read an environment variable at every launch instead.) */
#define KBE_STAGE_FUSED_LEAN 1024
#define KBE_STAGE_FUSED_ROOMY 2048
KBE_API int kbe_render_frame_stages(const float* points, const float* image, const float* depth, int N, int W,
                                    int H, double focal, double baseline, const float* shift3, void* scratch,
                                    uint8_t* frame_u8, float* render_f32, float* existing_f32, float* zee_f32,
                                    float* zee_pre_f32, int stages, const int* fill_rect, int raster_w, int raster_n,
                                    kbe_stream_t stream);
/* The same launches for a GROUP of 1..4 frames of the same cloud and size: every launch (projection, tiles, fill) takes all
 * the frames of the group (one grid dimension is the frame).  Frames are independent, and a launch on its own is bound by
 * its ramp and its latencies rather than by the chip, so several frames per launch cost much less than as many launches
 * (what kbe_render_video does with KBE_VIDEO_FILL_GROUP).  focals [n], shifts [3 n]; scratch [n]: one initialised scratch
 * set PER FRAME (kbe_frame_scratch_bytes each); frames_u8 [n]; zbuf_flags [n] or NULL: KBE_STAGE_ZBUF_A / _B / 0 per frame
 * (each scratch set follows the A, B, A ... rule of kbe_render_frame_stages on its own); `stages` as there. */
KBE_API int kbe_render_frame_group(const float* points, const float* image, const float* depth, int N, int W, int H,
                                   double baseline, int n_frames, const double* focals, const float* shifts,
                                   void* const* scratch, uint8_t* const* frames_u8, const int* zbuf_flags, int stages,
                                   const int* fill_rect, int raster_w, int raster_n, kbe_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * The same frame from the PACKED cloud: render_pointcloud as two launches with nothing but 12 bytes per point between them
 * (the default route of the Python host side).
 *
 * kbe_cloud_pack, once per video after the set-up loop has grown the cloud: sorts the points by where they lie in
 * the cloud's own view (Morton order of 8 x 8-pixel cells for points projected with `focal` onto a W x H raster; any
 * order is correct, this one is fast; raster_w / raster_n: the layout hint of kbe_render_frame_stages, 0, 0 if unknown:
 * the first raster_n points then keep their own 8 x 8 cells), stores them as {x, y, z} and {r, g, b, depth} records, cuts
 * them into blocks of 64 (sub-blocks of 16), and builds over the blocks a hierarchy of boxes that bound where a block's
 * points can land in any view (only the slow paths use it).  `packed`: DEVICE buffer of kbe_cloud_pack_bytes(N) bytes,
 * 256-byte aligned, owned by the caller; the original three tensors are not needed afterwards.
 *
 * kbe_render_frame_fused: one frame.  `stages`: KBE_STAGE_TILES = the scatter -- k_place: every point shifted and projected
 * once, its {ox, oy, dblError} stored at its own index (the scratch set's last 12 N bytes), every sub-block listed for the
 * one to four tiles its points reach; k_frame: a tile pulls the points of its list, z-splat into an LDS z-tile, degrid,
 * z-tested gather, normalise, uint8 -- no z-buffer, bucket or accumulator in HBM, no per-point global atomic;
 * KBE_STAGE_FILL (+ schedule flags) = the hole fill.
 * `cloud_focal` = the focal given to kbe_cloud_pack.  `parity`: the scratch holds two hole counters; consecutive
 * frames on one scratch alternate 0, 1, 0, ... (the fill of a frame zeroes the counter of the next), starting from
 * a scratch whose counters are zero (kbe_frame_scratch_init, or any frame rendered with parity -1); -1 = a frame on
 * its own: the counters are zeroed by a memset in front of the scatter.  Outputs, scratch (kbe_frame_scratch_bytes(W, H, N)
 * bytes) and fill_rect as kbe_render_frame_stages; results equal the bucket path's up to the order of the fp32 sums.
 * ------------------------------------------------------------------------------------- */
KBE_API size_t kbe_cloud_pack_bytes(int N);
KBE_API int kbe_cloud_pack(const float* points, const float* image, const float* depth, int N, int W, int H, double focal,
                           int raster_w, int raster_n, void* packed, kbe_stream_t stream);
KBE_API int kbe_render_frame_fused(const void* packed, int N, double cloud_focal, int W, int H, double focal, double baseline,
                                   const float* shift3, void* scratch, uint8_t* frame_u8, float* render_f32,
                                   float* existing_f32, float* zee_f32, float* zee_pre_f32, int stages,
                                   const int* fill_rect, int parity, kbe_stream_t stream);

/* ... and for a GROUP of 1..12 frames of the same packed cloud and size: the placement launch, the tile launch and the fill
 * each take all the frames (as kbe_render_frame_group on the other route).  scratch [n]: one initialised scratch set per
 * frame (kbe_frame_scratch_bytes(W, H, N) each); parities [n] (or NULL: all -1): as kbe_render_frame_fused, per scratch set. */
KBE_API int kbe_render_frame_group_fused(const void* packed, int N, double cloud_focal, int W, int H, double baseline, int n_frames,
                                         const double* focals, const float* shifts, void* const* scratch, uint8_t* const* frames_u8,
                                         const int* parities, int stages, const int* fill_rect, kbe_stream_t stream);

/* ... and PIPELINED: a sequence of groups on one stream in which the tile launch of a group also makes the placements of the
 * NEXT group -- k_place's work spread over the tile launch's waves, whose waits for memory it fills -- so that the scatter
 * of a group is ONE launch.  A scratch set holds two banks of placements, lists and counters; the k-th use of a set in the
 * sequence (turns [n]: k >= 0, counted per set; a set on turn 0 has its counters zeroed in front of the launch that first
 * names it -- the hole counters, the list totals and BOTH banks of list counters, so a sequence that was abandoned after a call
 * that placed ahead, or that ended in an error, costs the next one nothing but that reset) renders from bank k & 1 and uses
 * hole counter k & 1; a sequence should END with a call that places nothing ahead (the banks are then empty again; a set used
 * with parity -1 / by kbe_render_frame_fused afterwards relies on that).  Every argument is validated before anything is enqueued.  placed != 0: the group's placements exist (the previous call named these frames, sets and turns as its
 * `next`), else a placement launch is made in front of the tile launch.  n_next > 0: the tile launch makes the placements
 * of the frames next_* (the cameras, sets and turns the next call will render with; a set that both groups use has
 * next turn = turn + 1); only where kbe_render_frame_group_ahead_ok(N, W, H, n_frames, n_next) != 0 (a cloud much denser
 * than the raster keeps its placement launch).  near_depth > 0 (the depth of the nearest point the caller knows of: the
 * reference's objectDepthrange[0], common.py:88; 0 = unknown): consecutive frames of a group placed ahead whose cameras differ
 * in their shifts only share candidate lists in sub-groups of 12, 8, 6 or 4 frames -- as many as keep the nearest point
 * within 13 pixels between a sub-group's first and last camera -- kept in the scratch set of each sub-group's first frame: the
 * library decides that from the cameras and near_depth alone in both calls, so the `next` cameras of one call must be the
 * cameras of the next call (they already must) and near_depth the same.  Everything else as kbe_render_frame_group_fused; same
 * results. */
KBE_API int kbe_render_frame_group_ahead_ok(int N, int W, int H, int n_frames, int n_next);
KBE_API int kbe_render_frame_group_ahead(const void* packed, int N, double cloud_focal, int W, int H, double baseline, int n_frames,
                                         const double* focals, const float* shifts, void* const* scratch, uint8_t* const* frames_u8,
                                         const int* turns, int placed, int n_next, const double* next_focals, const float* next_shifts,
                                         void* const* next_scratch, const int* next_turns, int stages, const int* fill_rect,
                                         double near_depth, kbe_stream_t stream);

/* render_pointcloud (common.py:428-686) for one sample and ANY channel count on the tile machinery of the frame loop
 * (no accumulator in HBM, no floating-point atomic): z-splat + buckets, then per 32x16 tile degrid and a
 * z-tested gather four channels at a time.  data [C,N]; render [C,H,W] normalised (:686), existing [H*W] the
 * weight sum; shift3 as in kbe_zsplat (NULL: none); scratch as kbe_render_frame (left clean).  Same results
 * as kbe_render_pointcloud up to the order of the fp32 sums. */
KBE_API int kbe_render_pointcloud_tiled(const float* points, const float* data, int N, int C, int W, int H, double focal,
                                        double baseline, const float* shift3, void* scratch, float* render,
                                        float* existing, kbe_stream_t stream);

/* The whole frame loop of process_kenburns (common.py:222-260) for `n_frames` cameras, enqueued
 * from native code (no per-frame host-language work): per frame kbe_render_frame_stages, then the
 * device-side crop + resize of common.py:256-257 when crop_w/crop_h > 0 (holes the crop discards are
 * not filled), then the hand-off of the finished uint8 frame to host_out[i] ([n_frames,H,W,3]; the
 * `.cpu()` of common.py:255).  focals [n_frames] and shifts [n_frames][3] are HOST arrays (shift as
 * the fp32 values process_shift produces).
 * `packed` / `cloud_focal`: the packed cloud of kbe_cloud_pack (then every frame is kbe_render_frame_fused and points /
 * image / depth / raster_* are not used), or NULL / 0 for the bucket path.
 * Frames are independent, so consecutive frames are enqueued on `lanes` (1..KBE_MAX_LANES) streams,
 * lane 0 on `stream`, lane l on lane_streams[l] (may be NULL when lanes == 1); each lane has its
 * own scratch, raw frame and finished frame.  Synchronising `stream` afterwards guarantees every
 * frame has landed and every lane is idle.
 * Hand-off to pinned (device-visible) host memory happens in the lanes' own streams (no copy stream, no event), the
 * lanes taking turns on the PCIe link (a bounded, advisory device-side wait), by `batch`:
 *   batch < 0: groups of up to G = -batch consecutive frames (the first ones smaller: 1, 2, 4, ... or, with KBE_VIDEO_FAST_RAMP, 1, 3, 7, ...; KBE_VIDEO_EVEN_GROUPS: all of size G) are
 *       rendered by one lane into its own G buffers and leave with ONE hipMemcpyAsync per group (the runtime's transfer engine).  The default of the Python host side:
 *       G = 16 on 2 lanes keeps the link busy back to back (58 us per 1024^2 frame, 54 GB/s of PCIe Gen5 x16; G = 8: 59 us).
 *   batch == 0: per frame, by a small copy kernel (k_deliver: 16 workgroups, 16-byte stores, throttled).
 *   batch <= 0 with host_out = DEVICE memory: every frame's last kernel stores straight into host_out[i]
 *       (the frames stay in HBM).
 *   batch > 0: round 1's scheme, kept for comparison and for host memory the device cannot address: frames are
 *       staged on the device in two halves of `batch` frames and leave with one hipMemcpyAsync per half on
 *       copy_stream (NULL: on `stream`), cross-stream events per half.
 *   scratch: lanes * kbe_video_scratch_stride(W, H, N) bytes (n times that with KBE_VIDEO_FILL_GROUP(n)), each lane's part
 *            initialised with kbe_frame_scratch_init;
 *   stage:   DEVICE buffer, 256-byte aligned, of kbe_video_stage_bytes(W, H, lanes, batch) bytes.
 * The call creates and destroys its HIP events (the one exception to "nothing is allocated": conventions at the top), and on the
 * fused route starts by zeroing every scratch set's counters -- hole counters, list totals, both banks of list counters -- so a
 * previous call that ended in an error leaves nothing behind. */
#define KBE_MAX_LANES 8
KBE_API size_t kbe_video_scratch_stride(int W, int H, int N);
KBE_API size_t kbe_video_stage_bytes(int W, int H, int lanes, int batch);
#define KBE_VIDEO_FILL_DIST 1        /* kbe_render_video flags: KBE_STAGE_FILL_DIST for every frame */
/* batch <= 0: a lane renders n = 2..4 frames at a time, each into a scratch set of its own, and every launch of the bucket
 * route (projection, tiles, fill, crop) takes all n (the fused route: its fill and crop).  For videos that are bound by
 * their lanes' chains of launches rather than by the chip -- small frames, frames that fill with the tables -- n frames per
 * launch take much less than n times as long; a 1024 x 1024 video of an inpainted cloud on four lanes is bound by the chip
 * and loses 3-7 %.  `scratch` must then hold n * lanes sets (n * lanes * kbe_video_scratch_stride bytes, every set
 * initialised with kbe_frame_scratch_init); same frames. */
#define KBE_VIDEO_FILL_GROUP(n) (((n) - 1) << 1)
/* the same for n = 1..12 frames at a time on the fused route (`packed`): the placement and tile launches take all n, the fill and
 * the crop four at a time.  A launch on its own pays its ramp and its tail once whatever it holds: the fused scatter of a
 * 1024 x 1024 frame costs 35 / 27 / 23 / 20.8 / 20.1 us per frame with 1 / 2 / 4 / 8 / 12 frames per launch (bits 5-8 of `flags`). */
#define KBE_VIDEO_GROUP(n) (((n) - 1) << 5)
/* batch <= 0, frames to host memory: the lanes do not take turns on the PCIe link (for videos whose rendering binds, not the
 * link: a lane waiting for its turn would only idle) */
#define KBE_VIDEO_FREE_TRANSFERS 8
/* batch < 0: every transfer group has G = -batch frames.  Default: the groups ramp -- 1, 2, 4, ... frames up to G, then G -- so
 * that the link starts after the first frame and a short video does not wait for G frames before its first byte moves. */
#define KBE_VIDEO_EVEN_GROUPS 16
/* fused route (`packed`): every group keeps a placement launch of its own in front of its tile launch.  Default: the tile launch
 * of a lane's group also makes the placements of the lane's NEXT group (kbe_render_frame_group_ahead) unless that group is
 * larger (the ramp at the start of a delivered video), so that the scatter of a group is one launch. */
#define KBE_VIDEO_NO_AHEAD 512
/* the transfer groups of the hand-off by groups grow 1, 3, 7, 15, 31, ... (each twice the last plus one) instead of 1, 2, 4, 8, ...:
 * the link needs 25 us + 57 us per 1024^2 frame for a group while the next group renders at 20-30 us per frame, so a group may
 * be a little more than twice the one in flight -- and every transfer saved is ~25 us of link time (a 20-frame video: 4 transfers
 * instead of 6, a 75-frame video 6 instead of 8).  The last group takes what is left.  (Measured: no gain on MI355X -- the larger
 * groups render next to the other lane's transfer and are slowed by it; the Python host side leaves it off.) */
#define KBE_VIDEO_FAST_RAMP 1024
/* batch < 0, frames to pinned host memory: the groups leave through an SDMA engine driven through HSA
 * (hsa_amd_memory_async_copy_on_engine; the copy's dependency signal is released and its completion signal awaited by one-lane
 * kernels in the lanes' streams, so the call stays asynchronous and `stream` still sees every frame delivered) instead of the
 * runtime's hipMemcpyAsync, which is a blit kernel on the compute units.  The engine takes the groups in order: no turns.
 * Falls back to hipMemcpyAsync when HSA does not report an engine for the two buffers (or refuses a copy: the rest of the video then
 * leaves through the runtime, the lanes taking turns again).  A completion signal that has not fired after 4 s of polling is a dead or
 * a stalled engine: the polling kernel gives up, stores into a host-visible error word and RETURNS (nothing traps, the process lives) --
 * kbe_video_handoff_status() reports it, and a caller that needs to know its frames have landed asks it after synchronising `stream`.
 * An error INSIDE the call (a launch that fails behind a copy already on the engine) returns only after every such copy has been
 * released and has completed: no copy of a failed call outlives it or keeps waiting.  The one piece of process-wide state the library
 * keeps belongs to this flag: a pool of HSA signals, reused once the call that used them has run to its end, and the error word.
 * Needs an HSA runtime with hsa_amd_memory_async_copy_on_engine (ROCm >= 6.0; libkbe_hip.so links libhsa-runtime64). */
#define KBE_VIDEO_SDMA 8192
/* fused route: force the lean / the roomy build of the tile launches (KBE_STAGE_FUSED_LEAN / _ROOMY for every frame) */
#define KBE_VIDEO_FUSED_LEAN 2048
#define KBE_VIDEO_FUSED_ROOMY 4096
#define KBE_VIDEO_FILL_PAIRS KBE_VIDEO_FILL_GROUP(2)
/* TEST HOOK (with KBE_VIDEO_SDMA): the hand-off of the video's second transfer group (its only one, if it has one) fails AFTER its copy
 * has been enqueued on the engine, the way a failed launch of the releasing kernel would: the call returns KBE_E_LAUNCH, through the same
 * clean-up a real failure takes (tests/test_hip_parity.py: the next video in the process is delivered intact, nothing is written into
 * the failed call's buffer after it has returned).  Costs nothing when not set. */
#define KBE_VIDEO_INJECT_FAULT 32768
/* TEST HOOK (with KBE_VIDEO_SDMA): the kernels that wait for the lanes' LAST groups to have left give up after one tick of the wall clock
 * instead of seconds -- the path an engine that stops answering takes: the error word is set, kbe_video_handoff_status() reports it once
 * (after waiting on the host for the copies, which here do complete), the engine stays off for the process. */
#define KBE_VIDEO_INJECT_TIMEOUT 65536
KBE_API int kbe_render_video(const float* points, const float* image, const float* depth, int N, int W, int H,
                             double baseline, int n_frames, const double* focals, const float* shifts, int crop_w,
                             int crop_h, void* scratch, uint8_t* stage, int batch, uint8_t* host_out,
                             int raster_w, int raster_n, const void* packed, double cloud_focal, int flags,
                             kbe_stream_t stream, kbe_stream_t copy_stream, int lanes, const kbe_stream_t* lane_streams,
                             double near_depth);
/* KBE_OK, or -- ONCE -- KBE_E_LAUNCH after a KBE_VIDEO_SDMA hand-off of this process has given up waiting for its engine (kbe_last_error
 * says so; the engine is not used again: later videos leave through hipMemcpyAsync, complete, and this call says KBE_OK for them).  Call it
 * after synchronising the stream a delivered video was enqueued on: on the error it first waits on the host, for seconds at most, for the
 * copies that may still be under way, so that the caller can free its buffer afterwards. */
KBE_API int kbe_video_handoff_status(void);

/* generate_mask's kernel (common.py:689-817; the median-5 of :829 is kbe_spatial_filter): masks[B,N] = 1 where
 * point i of points[B,3,N] + shift[B,3] (a DEVICE array, the tensorShift of :690) owns the pixel its z-splat
 * winner corner falls on, else 0.  The reference launch is a race; this computes its serial-order result
 * deterministically (owner = first point in index order attaining the pixel's minimal dblError; point 0, once an
 * owner, keeps its 1: `pid > 0`, :759).  Scratch: keys [B*H*W] 64-bit (8-byte aligned), winner [B,N] int32.
 * Optional outputs in the reference's formats: zee [B,H,W] (:692) and ids [B,H,W] (:694: owner index, or the int
 * bits of -1.0f where no point landed). */
KBE_API int kbe_generate_mask(const float* points, const float* shift, int B, int N, int W, int H, double focal,
                              double baseline, unsigned long long* keys, int32_t* winner, float* masks, float* zee,
                              int32_t* ids, kbe_stream_t stream);

/* common.py:255: (render[0:3] * 255).clip(0, 255).astype(uint8), CHW fp32 -> HWC u8 */
KBE_API int kbe_frame_u8(const float* render_chw, int W, int H, uint8_t* frame_hwc, kbe_stream_t stream);

/* common.py:256-257 equivalent on device: centred crop of (crop_w x crop_h) as cv2.getRectSubPix
   samples it, then bilinear resize back to (W x H) as cv2.resize(INTER_LINEAR) does on 8-bit
   images (fixed-point coefficients).  Parity with OpenCV is UNPINNED (OpenCV is not in the
   image). */
KBE_API int kbe_crop_resize_u8(const uint8_t* frame_hwc, int W, int H, int crop_w, int crop_h,
                       uint8_t* out_hwc, kbe_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * torch glue on the path
 * ------------------------------------------------------------------------------------- */

/* depth_to_points (common.py:382-392): depth [B,1,H,W] -> points [B,3,H,W].
   `valid` (optional, [B,1,H,W]) multiplies the depth first (common.py:71). */
KBE_API int kbe_depth_to_points(const float* depth, const float* valid, int B, int W, int H, double focal,
                        float* points, kbe_stream_t stream);

/* process_shift's tensor part (common.py:104-109), materialised (API compatibility) */
KBE_API int kbe_shift_points(const float* points, int B, int N, const float* shift3, float* out,
                     kbe_stream_t stream);

/* spatial_filter (common.py:394-426) over `planes` = B*C independent [H,W] planes.
   kind: 0 = 'laplacian' (replicate pad, the reference's asymmetric taps),
         3 = 'median-3', 5 = 'median-5' (reflect pad, lower median). */
KBE_API int kbe_spatial_filter(const float* in, int planes, int W, int H, int kind, float* out,
                       kbe_stream_t stream);

/* (|laplacian(x / *scale_dev)| < threshold) as 0/1 floats -- the validity mask of
   common.py:70 and pointcloud_inpainting.py:193.  scale_dev: DEVICE pointer to one float
   (tensor.max()), so no host sync is needed. */
KBE_API int kbe_laplacian_valid(const float* in, const float* scale_dev, int planes, int W, int H,
                        float threshold, float* valid, kbe_stream_t stream);

/* PartialConv2d mask bookkeeping fused into one pass (utils/partial_conv.py:62-77,
   multi_channel=True):  msum = box-sum of mask over Cin*k*k (zero pad);
   um = clamp(msum, 0, 1); ratio = Cin*k*k / (msum + 1e-8) * um;
   out = ((raw - bias) * ratio + bias) * um   (bias NULL: out = raw * ratio).
   raw/out [B,Cout,Ho,Wo] (may alias); mask [B,mask_channels,H,W] with mask_channels = Cin, or 1
   when every input channel carries the same mask (msum = Cin * single-channel box sum), or
   NULL = no mask given (all ones, :49-56); um [B,1,Ho,Wo] optional (the reference materialises
   Cout identical copies).
   What follows a layer inside the partial-convolution GridNet's blocks (models/partial_inpainting.py:16-52) can ride in the
   same pass (round 4; both optional, NULL = as before): residual [B,Cout,Ho,Wo] -- out += residual, the block's `+ skip` --
   and then prelu_slope [Cout] -- out = prelu(out), the next layer's activation, whose `input * mask_in` needs no pass of its
   own because out is already 0 wherever um, the next layer's mask, is 0.
   raw_without_bias != 0: `raw` is the convolution WITHOUT its bias (MIOpen's Winograd kernels take none: PyTorch adds it in a
   broadcasting pass of its own, 21 us per layer at 1024^2) and the epilogue forms raw + bias first, with the rounding that
   pass would have made -- the same values, one pass less. */
KBE_API int kbe_pconv_epilogue(const float* raw, const float* bias, const float* mask, int mask_channels, int B,
                               int Cin, int H, int W, int Cout, int Ho, int Wo, int k, int stride, int pad,
                               float* out, float* um, const float* prelu_slope, const float* residual, int raw_without_bias,
                               kbe_stream_t stream);

/* out = prelu(x, slope) * mask in one pass: a block's first activation and the `input * mask_in` of PartialConv2d.forward
   (utils/partial_conv.py:61).  x, out [B,C,H,W] (may alias); slope [C]; mask [B,1,H,W] or NULL (no multiplication). */
KBE_API int kbe_prelu_mask(const float* x, const float* slope, const float* mask, int B, int C, int H, int W, float* out,
                           kbe_stream_t stream);

/* The element-wise passes PyTorch runs around a convolution of the (plain) networks, in ONE pass over the convolution's output
   (round 4; the blocks of models/pointcloud_inpainting.py:16-80, disparity_estimation.py, disparity_refinement.py of the reference:
   `nn.Conv2d`'s bias add -- MIOpen's Winograd kernels take none --, the `nn.PReLU` behind it, `moduleMain(x) + skip`, and the
   GridNet's `+ the stream from the neighbouring row`, :133-172):
       out = act(x + bias[c]) + res1 + res2        act = PReLU with slope[c] or none
   x, out, res1, res2 [B,C,H,W] (out may alias x); bias, slope [C]; bias, slope, res1, res2 may each be NULL.  The additions are
   made in that order, as the separate passes make them.  B C <= 65535. */
KBE_API int kbe_bias_act(const float* x, const float* bias, const float* slope, const float* res1, const float* res2, int B, int C,
                         int H, int W, float* out, kbe_stream_t stream);

/* nn.Upsample(scale_factor=2, mode='bilinear', align_corners=False) and the nn.PReLU behind it, the head of an `Upsample` block
   (models/pointcloud_inpainting.py:54-80), in one pass: x [B,C,H,W] -> out [B,C,2H,2W]; slope [C] or NULL (no activation).
   Source positions and weights as PyTorch's upsample_bilinear2d computes them (0.5 (dst + 0.5) - 0.5 clamped at 0, the
   neighbour clamped at the edge). */
KBE_API int kbe_upsample2x_act(const float* x, const float* slope, int B, int C, int H, int W, float* out, kbe_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* KBE_H */
