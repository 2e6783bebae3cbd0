#!/usr/bin/env python3
"""Benchmark of the novel-view frame loop (BASELINE.json: frames/sec at 1024x1024, scatter HBM GB/s).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one output frame of the reference's process_kenburns loop body
(/root/reference/utils/common.py:222-260): camera shift -> forward-warp of the resident point
cloud (z-splat, degrid, z-tested bilinear accumulate) -> disocclusion fill -> uint8 ->
centred crop + resize -> the finished frame landing in pinned host memory (the `.cpu()` of
common.py:255; SURVEY.md 8d counts it in the step).  The point cloud is resident in HBM when the
timed region starts; frames are sharded over ranks (rank r renders steps r, r+N, ... of an N*K-step
path; --video-frames: of ONE video, strong scaling).  Synthetic seeded RGBD input (no datasets / checkpoints are reachable offline).

Timing: W untimed warm-up frames, then passes of EXACTLY K frames, each bracketed by barrier +
synchronize on both sides (max over ranks), repeated until >= 0.5 s of timed wall; `value` is K * N /
the MEDIAN pass.  `device_only` is the same loop with the frames left in HBM (no PCIe), reported beside
it.  The cloud broadcast (multi-GPU) is scene set-up, timed separately as `cloud_broadcast_ms`.

Prints ONE JSON line (rank 0) with the driver's contract plus `roofline` (the scatter =
render_pointcloud: SURVEY.md 8d's 28 N + 20 HW bytes over the HIP-event time of its launches, measured
live) and `cpu_baseline` (the CPU oracle timed on this host, rank 0 at N=1 only -- the oracle is
used here as the baseline, never as the product path).
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault('MIOPEN_FIND_MODE', 'FAST')     # the (untimed) Inpaint set-up: no exhaustive conv search

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s achievable


def build_scene(size, device, inpaint, settings=None, upsample=1):
    """objectCommon for a seeded synthetic size x size RGBD image.  With `inpaint` the point
    cloud is grown exactly as process_kenburns' set-up loop does (common.py:181-219): two end poses, each
    inpainted by the Inpaint network (seeded random weights: checkpoints are a network download) and the
    pixels that pose cannot see appended as new points."""
    from ken_burns_effect_amd import _native, common, synthetic
    K = _native.kernels()
    image, disp = synthetic.make_rgbd(size, size, seed=0)
    depth = (synthetic.FOCAL * synthetic.BASELINE) / (disp + 1e-7)
    oc = {'dblFocal': synthetic.FOCAL, 'dblBaseline': synthetic.BASELINE, 'intWidth': size, 'intHeight': size,
          'dblDispmin': float(disp.min()), 'dblDispmax': float(disp.max()),
          'objectDepthrange': synthetic.depthrange_of(depth), 'tensorRawImage': image.to(device),
          'tensorRawDisparity': disp.to(device), 'tensorRawDepth': depth.to(device)}
    oc['tensorRawPoints'] = K.depth_to_points(oc['tensorRawDepth'], synthetic.FOCAL).view(1, 3, -1)
    common._reset_inpa(oc)
    if upsample > 1:
        # BASELINE.json configs[4]: a point cloud `upsample`^2 times denser than the target raster -- the RGBD of an
        # (upsample * size)^2 image unprojected with focal upsample * F lands on sub-pixel positions of the size^2 view
        up = size * upsample
        image_u, disp_u = synthetic.make_rgbd(up, up, seed=0)
        depth_u = ((synthetic.FOCAL * synthetic.BASELINE) / (disp_u + 1e-7)).to(device)
        oc['tensorInpaPoints'] = K.depth_to_points(depth_u, synthetic.FOCAL * upsample).view(1, 3, -1)
        oc['tensorInpaImage'] = image_u.to(device).reshape(1, 3, -1)
        oc['tensorInpaDepth'] = depth_u.reshape(1, 1, -1)
        oc['tensorInpaDisparity'] = disp_u.to(device).reshape(1, 1, -1)
        oc['_kbeCloudRaster'] = (up, up * up)       # layout hint for the projection kernel (speed only)
    if inpaint:
        from ken_burns_effect_amd.pointcloud_inpainting import Inpaint
        net = synthetic.seeded_fill_(Inpaint(), 3).to(device).eval()
        with torch.no_grad():
            common.build_pointcloud(settings, oc, net)
        del net
        torch.cuda.empty_cache()
    return oc


KERNEL_SOURCES = ('kbe_fused.hip', 'kbe_frame.hip', 'kbe_tiles.h', 'kbe_device.h', 'kbe_cloud.h', 'kbe_cloud.hip')


def kernel_sources_stamp():
    """sha256 (16 hex digits) of the scatter's kernel sources as they lie in the tree -- their CODE: comments and white space are taken
    out first, so that a reworded comment does not disown a measurement -- what a committed PMC file says it was measured on
    (`sources_sha16`, written by the tools/pmc_*report.py that produce it).  The GPU box has no .git: a content hash travels."""
    import hashlib
    import re
    h = hashlib.sha256()
    for name in KERNEL_SOURCES:
        with open(os.path.join(ROOT, 'ken-burns-effect_amd', 'csrc', name), 'r', encoding='utf-8', errors='replace') as f:
            text = f.read()
        text = re.sub(r'/\*.*?\*/', ' ', text, flags=re.S)          # block comments
        text = re.sub(r'//[^\n]*', ' ', text)                       # line comments (no string literal of these files holds "//")
        h.update(re.sub(r'\s+', ' ', text).encode())
    return h.hexdigest()[:16]


def _fresh(doc, path):
    """A committed PMC file counts while the kernel sources are the ones it was measured on (the constant went stale
    silently when the kernel changed).  Files from before the stamp existed (rounds 1-5) carry none and are stale by definition
    once the sources have changed since -- which they have."""
    stamp = doc.get('sources_sha16')
    now = kernel_sources_stamp()
    if stamp == now:
        return True
    sys.stderr.write('bench.py: %s was measured on kernel sources %s, the tree holds %s: its figures are left out of the line\n' % (os.path.relpath(path, ROOT), stamp or '(unstamped)', now))
    return False


def measured_traffic(workload=''):
    """Per-frame HBM bytes of the frame kernels from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE /
    WRITE_SIZE in separate runs, calibrated and corrected by tools/pmc_report.py as MI355X_MICROARCH.md
    prescribes).  PMC counters cannot be collected from inside this process, so the figures are read from the newest
    profiles/r*_hbm_traffic.json (the default workload) or r*_hbm_traffic_config4.json (`workload` = '_config4': 2048^2 from
    16.8 M points, tools/pmc_config4_report.py); ({}, None) when there is none."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_hbm_traffic%s.json' % workload)))
    if not files:
        return {}, None
    try:
        doc = json.load(open(files[-1]))
        if not _fresh(doc, files[-1]):
            return {}, None
        per = {name: v['hbm_bytes'] for name, v in doc['kernels'].items()}
        # the group launches by frames per launch (tools/pmc_group_report.py): {'k_frame_group_ahead': {'12': {...}}}
        per['by_frames_per_launch'] = {k: v for k, v in doc.get('by_frames_per_launch', {}).items() if isinstance(v, dict)}
        return per, os.path.relpath(files[-1], ROOT)
    except Exception:
        return {}, None


def measured_instructions():
    """Per-launch wave-level VALU instruction counts of the scatter kernels from the committed PMC pass (SQ_INSTS_VALU;
    tools/gpu_pmc_scatter.sh -> profiles/r*_scatter_insts.json); {} when there is none."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_scatter_insts.json')))
    if not files:
        return {}, None
    try:
        doc = json.load(open(files[-1]))
        if not _fresh(doc, files[-1]):
            return {}, None
        per = dict(doc['kernels'])
        per['by_frames_per_launch'] = {k: v for k, v in doc.get('by_frames_per_launch', {}).items() if isinstance(v, dict)}
        return per, os.path.relpath(files[-1], ROOT)
    except Exception:
        return {}, None


def consecutive_groups(path, n):
    """The launch groups of a video whose cameras are `path`: group k = cameras [n k, n k + n); where the path does not divide, the
    last group is the path's LAST n cameras (a 20-step path in groups of twelve: [0..11], [8..19]) -- every launch holds n consecutive
    cameras of the path, as every launch of a real video does (until late in round 5 the last group wrapped round to the path's start:
    a jump from the last camera to the first inside one launch, which no video makes and which took that launch's shared lists away).
    What the scatter's launch is priced on since round 5: the frames of a group share candidate lists built for the box
    between a sub-group's first and last camera, so a launch of n copies of ONE camera is that scheme's best case."""
    if len(path) <= n:
        return [list(path)]
    groups = [list(path[k:k + n]) for k in range(0, len(path) - n + 1, n)]
    if len(path) % n:
        groups.append(list(path[-n:]))
    return groups


PRODUCT_STEPS = 75      # /root/reference/kbe.py:104: a video of the product is np.linspace(0, 1, 75)
DRIVER_STEPS = 20       # the driver's `--steps 20`
GROUP_FRAMES = 4        # frames per launch of the grouped launches (kbe_render_frame_group / _fused) when the timed region uses one
BUCKET_GROUP_MAX = 4    # kbe_render_frame_group takes up to four frames, kbe_render_frame_group_fused up to eight
# Wave-level VALU instructions the chip issues per us through ONE issue port per SIMD -- measured (tools/valu_rate.hip,
# profiles/r04_valu_rate.txt, r04_valu_pairs.txt): 557-571 per ns chip-wide for every instruction kind outside the co-issue class
# (fused multiply-adds with a separate addend register, min / max, compares, conversions, shifts left, 24-bit multiplies, DPP, every
# three-operand integer form ...), from two waves per SIMD up -- 1024 SIMDs x ~2.38 GHz / 4.26 cycles.  Plain fp32 add / sub / mul,
# integer add / sub, and / or / xor, mov, shifts right WITH VECTOR-REGISTER OPERANDS ONLY go through a second port alongside
# (up to 1 070 per ns when nothing else issues; a 50 : 50 mix with one-port instructions: 890-1 000); in the scatter a fifth of
# the instructions do (SQ_ACTIVE_INST_VALU2 / SQ_INSTS_VALU, profiles/r04_scatter_busy.json).
VALU_ISSUE_PER_US = 570.0e3


def time_kernels(oc, cams, route, reps=40, fill_rect=None, group_frames=GROUP_FRAMES, fill_flags=0, paths=None, fill_group=1):
    """Average GPU time of the frame launches from HIP events on the launch stream (torch's
    current stream is the one the C ABI launches on).  Each figure is `reps` back-to-back
    launches between two events; the tile kernel is timed alone (back to back on a prepared scratch) --
    rocprofv3's per-kernel average (profiles/) is the cross-check.  Returns {name: seconds}."""
    from ken_burns_effect_amd import _native
    K = _native.kernels()
    W, H = oc['intWidth'], oc['intHeight']
    state = K.prepare_cloud(oc['tensorInpaPoints'], oc['tensorInpaImage'], oc['tensorInpaDepth'], W, H, oc['dblFocal'], raster=oc.get('_kbeCloudRaster'),
                            near_depth=oc['objectDepthrange'][0] if oc.get('objectDepthrange') else None)
    focal, shift3 = cams[len(cams) // 2]
    Bl = oc['dblBaseline']

    def timed(fn):
        # (as many launches untimed first: this runs behind the host-side frame check, i.e. on a chip that has idled for
        # milliseconds, and the clocks need a few ms of load -- the first 40 launches of a kind read 3-4 % slower than the next 40)
        for _ in range(reps + 1):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / reps * 1e-3

    empty = (1, 1, 0, 0)
    out = {}
    # the fused route: the scatter = k_place + k_frame; back to back on the stream, consecutive frames alternating between the
    # scratch's two hole counters as in a video (no fill here to zero them: the hole list is bounded, the frames not used)
    fpar = [0]

    def fused_scatter():
        K.render_frame(state, shift3, focal, Bl, stages=2, fused=True, parity=fpar[0] & 1)
        fpar[0] += 1
    out['fused:scatter'] = timed(fused_scatter)
    K.render_frame(state, shift3, focal, Bl, fused=True)       # a frame on its own zeroes the counters the run above left
    # the same two launches (k_place, k_frame) taking `group_frames` frames each (kbe_render_frame_group_fused), per launch pair
    fgroup_out = torch.empty(group_frames, H, W, 3, dtype=torch.uint8, device=oc['tensorInpaPoints'].device)
    gpar = [0]

    def fused_grouped():
        K.render_frame_group_fused(state, [(focal, shift3)] * group_frames, Bl, fgroup_out, stages=2, parities=[gpar[0] & 1] * group_frames)
        gpar[0] += 1
    out['fused:scatter_group'] = timed(fused_grouped)
    K.render_frame_group_fused(state, [(focal, shift3)] * group_frames, Bl, fgroup_out, stages=6, fill_rect=empty)    # parity -1: the sets' counters zeroed in front
    # the scatter as the video loop launches it (kbe_render_frame_group_ahead): ONE launch per group in steady state -- the tile
    # launch of a group (k_frame_group_ahead) also makes the placements of the next group, here the same frames again; the sets
    # take turn after turn (banks and counters alternate)
    for key, n_ahead in (('fused:scatter_ahead', 1), ('fused:scatter_group_ahead', group_frames)):
        if not K.lib.kbe_render_frame_group_ahead_ok(state['N'], W, H, n_ahead, n_ahead):
            continue            # a cloud much denser than the raster keeps its placement launch (the video loop decides the same way)
        group = [(focal, shift3)] * n_ahead
        turn = [0]
        # (the argument arrays built once: through the ordinary wrapper the host needs about as long per call as the GPU per launch,
        # and the figure then follows the host's speed -- 0.31 on one box, 0.35 on the next)
        launch = K.prepared_group_ahead(state, group, Bl, fgroup_out[:n_ahead], group, stages=2)

        def fused_ahead():
            launch(turn[0], turn[0] > 0)
            turn[0] += 1
        # (the figure `roofline` is priced on: five rounds of `reps` launches, the median round -- single rounds scatter by +-2.5 %
        # with the chip's clocks; every round is kept in `out` for the line)
        rounds = sorted(timed(fused_ahead) for _ in range(5))
        out[key] = rounds[len(rounds) // 2]
        out[key + ':rounds'] = rounds
        K.render_frame_group_ahead(state, group, Bl, fgroup_out[:n_ahead], turn=turn[0], placed=True, next_cameras=None, stages=6, fill_rect=empty)  # the sequence ends
    # ... and on a video's OWN groups: consecutive cameras of a path (`paths`: {label: cameras}), the launch of group k placing group
    # k + 1 ahead, the path's groups taken round and round.  Timed in whole cycles of the path's groups (>= reps launches), five rounds,
    # the median round.
    n = group_frames
    if K.lib.kbe_render_frame_group_ahead_ok(state['N'], W, H, n, n):
        for label, path in (paths or {}).items():
            groups = consecutive_groups(path, n)
            launches = [K.prepared_group_ahead(state, g, Bl, fgroup_out[:n], groups[(k + 1) % len(groups)], stages=2) for k, g in enumerate(groups)]
            # turn 0 places group 0 (a placement launch in front); from then on turn t renders group (t - 1) % G and places group t % G
            K.render_frame_group_ahead(state, groups[-1], Bl, fgroup_out[:n], turn=0, placed=False, next_cameras=groups[0], stages=2)
            turn = [1]

            def consecutive():
                launches[(turn[0] - 1) % len(launches)](turn[0], True)
                turn[0] += 1
            count = len(groups) * max(1, (reps + len(groups) - 1) // len(groups))
            rounds = []
            for _ in range(5):
                for _ in range(count):
                    consecutive()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(count):
                    consecutive()
                e1.record()
                e1.synchronize()
                rounds.append(e0.elapsed_time(e1) / count * 1e-3)
            rounds.sort()
            out['fused:scatter_group_ahead:consecutive:' + label] = rounds[len(rounds) // 2]
            out['fused:scatter_group_ahead:consecutive:' + label + ':rounds'] = rounds
            K.render_frame_group_ahead(state, groups[(turn[0] - 1) % len(groups)], Bl, fgroup_out[:n], turn=turn[0], placed=True, next_cameras=None, stages=6, fill_rect=empty)
            del launches
    del fgroup_out
    out['fused:scatter+fill'] = timed(lambda: K.render_frame(state, shift3, focal, Bl, stages=6, fill_rect=fill_rect, fused=True))   # + the 8-byte memset of a frame on its own
    # the bucket route: k_project -> k_tiles, z-buffer and bucket records in HBM.  In a video consecutive frames alternate
    # between two z-buffers and each tile launch clears the other one (stage flags 128 / 256), so the scatter is these two
    # launches; a frame on its own has the reset riding in its fill launch (`bucket:scatter_alone`, empty fill rectangle)
    b = dict(fused=False)
    flip = [0]

    def alternating(stages, **kw):
        def run():
            K.render_frame(state, shift3, focal, Bl, stages=stages | (256 if flip[0] & 1 else 128), **b, **kw)
            flip[0] += 1
        return run

    def settle():           # a sequence must not end on z-buffer A (include/kbe.h): one more frame on B
        if flip[0] & 1:
            alternating(3)()
    out['bucket:scatter'] = timed(alternating(3))
    # the same two launches taking FOUR frames each (kbe_render_frame_group: what videos with KBE_VIDEO_FILL_GROUP use), per launch pair
    bucket_frames = min(group_frames, BUCKET_GROUP_MAX)
    group_out = torch.empty(bucket_frames, H, W, 3, dtype=torch.uint8, device=oc['tensorInpaPoints'].device)
    gflip = [0]

    def grouped():
        K.render_frame_group(state, [(focal, shift3)] * bucket_frames, Bl, group_out, stages=3, zbuf_flags=[256 if gflip[0] & 1 else 128] * bucket_frames)
        gflip[0] += 1
    out['bucket:scatter_group'] = timed(grouped)
    if gflip[0] & 1:
        grouped()
    K.render_frame_group(state, [(focal, shift3)] * bucket_frames, Bl, group_out, stages=4, fill_rect=empty)       # leaves the sets clean
    del group_out
    out['bucket:scatter+fill'] = timed(alternating(7, fill_rect=fill_rect))
    settle()
    out['bucket:reset'] = timed(lambda: K.render_frame(state, shift3, focal, Bl, stages=4, fill_rect=empty, **b))
    out['bucket:project+reset'] = timed(lambda: K.render_frame(state, shift3, focal, Bl, stages=5, fill_rect=empty, **b))
    out['bucket:scatter_alone'] = timed(lambda: K.render_frame(state, shift3, focal, Bl, stages=7, fill_rect=empty, **b))      # project + tiles + reset
    K.render_frame(state, shift3, focal, Bl, stages=1, **b)
    out['bucket:tiles'] = timed(lambda: K.render_frame(state, shift3, focal, Bl, stages=2, **b))       # k_tiles alone on a prepared scratch
    K.render_frame(state, shift3, focal, Bl, stages=4, fill_rect=empty, **b)           # leave the scratch clean
    for k in ('scatter', 'scatter+fill'):
        out[k] = out[route + ':' + k]
    out['fill'] = out['scatter+fill'] - out['scatter']
    # the hole fill along the camera path (its cost follows the number of holes: a dolly zoom's late frames have 700 k of them):
    # eight cameras spread over the video, each the frame's launches with and without the fill, with the schedule the video loop
    # uses (`fill_flags`: KBE_STAGE_FILL_BY_COUNT, and KBE_STAGE_FILL_DIST where the loop fills with the tables)
    fills = []
    fkw = dict(fused=(route == 'fused'))
    for cam_f, cam_s in [cams[(2 * i + 1) * len(cams) // 16] for i in range(8)]:
        if route == 'fused':
            t_all = timed(lambda: K.render_frame(state, cam_s, cam_f, Bl, stages=6 | fill_flags, fill_rect=fill_rect, **fkw))
            t_scatter = timed(lambda: K.render_frame(state, cam_s, cam_f, Bl, stages=2, parity=-1, **fkw))
        else:
            t_all = timed(lambda: K.render_frame(state, cam_s, cam_f, Bl, stages=7 | fill_flags, fill_rect=fill_rect, **fkw))
            t_scatter = timed(lambda: K.render_frame(state, cam_s, cam_f, Bl, stages=7, fill_rect=empty, **fkw))    # the fill launch with nothing to fill: its reset
        fills.append(max(t_all - t_scatter, 0.0))
    out['fill_along_path'] = sum(fills) / len(fills)
    out['fill_along_path_max'] = max(fills)
    # ... and with the frames per launch the video loop gives the fill where it groups them (KBE_VIDEO_FILL_GROUP: the table-driven fill of a
    # dolly zoom takes four frames per launch on the bucket route -- its launches are bound by their own chains of dependent look-ups, and
    # four frames in one launch take much less than four times one): four consecutive cameras at the same eight places of the path
    fill_group = min(fill_group, BUCKET_GROUP_MAX)           # (the fill's launches take up to four frames whatever the scatter's take)
    if fill_group > 1:
        gout = torch.empty(fill_group, H, W, 3, dtype=torch.uint8, device=oc['tensorInpaPoints'].device)
        gfills = []
        for i in range(8):
            at = min((2 * i + 1) * len(cams) // 16, max(0, len(cams) - fill_group))
            group = cams[at:at + fill_group]
            if len(group) < fill_group:
                break
            if route == 'fused':
                t_all = timed(lambda: K.render_frame_group_fused(state, group, Bl, gout, stages=6 | fill_flags, fill_rect=fill_rect))
                t_scatter = timed(lambda: K.render_frame_group_fused(state, group, Bl, gout, stages=2))
            else:
                t_all = timed(lambda: K.render_frame_group(state, group, Bl, gout, stages=7 | fill_flags, fill_rect=fill_rect))
                t_scatter = timed(lambda: K.render_frame_group(state, group, Bl, gout, stages=7, fill_rect=empty))  # the fill launch with nothing to fill: its reset
            gfills.append(max(t_all - t_scatter, 0.0))
        if gfills:
            out['fill_group_along_path'] = sum(gfills) / len(gfills)
            out['fill_group_along_path_max'] = max(gfills)
            out['fill_group_frames'] = fill_group
        del gout
    frame = K.render_frame(state, shift3, focal, Bl)
    cw, ch = int(0.9 * W), int(0.9 * H)
    out['crop_resize'] = timed(lambda: K.crop_resize_u8(frame, cw, ch))
    # the generic stage-by-stage path (global atomics), for comparison
    pts = oc['tensorInpaPoints']
    data = torch.cat([oc['tensorInpaImage'], oc['tensorInpaDepth']], 1)
    zkeys, _ = K.zsplat(pts, W, H, focal, Bl, shift3)
    zee = K.degrid(zkeys=zkeys)
    out['generic_zsplat'] = timed(lambda: K.zsplat(pts, W, H, focal, Bl, shift3))
    out['generic_accumulate'] = timed(lambda: K.accumulate(pts, data, zee, focal, Bl, shift3))
    return out


def cpu_baseline(oc, cams, crop, budget_s=20.0):
    """The CPU oracle (oracle/kbe_oracle.c) rendering the same frames on this host: single thread, then one frame per
    host thread (`all_cores`)."""
    from oracle import kbe_oracle
    ok = kbe_oracle.OracleKernels(schedule='jacobi')
    W, H = oc['intWidth'], oc['intHeight']
    state = ok.prepare_cloud(oc['tensorInpaPoints'].cpu(), oc['tensorInpaImage'].cpu(), oc['tensorInpaDepth'].cpu(), W, H)
    torch.set_num_threads(1)
    t0 = time.perf_counter()
    n = 0
    while n < len(cams) and (n < 2 or time.perf_counter() - t0 < budget_s):
        focal, shift3 = cams[(n * 7) % len(cams)]
        frame = ok.render_frame(state, shift3, focal, oc['dblBaseline'])
        if crop is not None:
            ok.crop_resize_u8(frame, crop[0], crop[1])
        n += 1
    dt = time.perf_counter() - t0
    out = {'value': n / dt, 'unit': 'frames/s', 'cores': 1, 'kind': 'port',
           'sample': '%d frames of the same %dx%d workload, oracle/kbe_oracle.c single-threaded (host has %d cores)'
                     % (n, W, H, os.cpu_count())}
    # frames are independent, so the same port also runs one frame per host thread (the C calls release the GIL);
    # reported beside the single-thread figure, bounded the same way
    threads = max(1, int(os.environ.get('KBE_CPU_BASELINE_THREADS', os.cpu_count() or 1)))      # all host cores
    if threads > 1:
        from concurrent.futures import ThreadPoolExecutor
        deadline = time.perf_counter() + 0.5 * budget_s

        def work(t):
            k = 0
            while k == 0 or time.perf_counter() < deadline:
                focal, shift3 = cams[((t * 131 + k) * 7) % len(cams)]
                frame = ok.render_frame(state, shift3, focal, oc['dblBaseline'])
                if crop is not None:
                    ok.crop_resize_u8(frame, crop[0], crop[1])
                k += 1
            return k
        t0 = time.perf_counter()
        with ThreadPoolExecutor(threads) as pool:
            done = sum(pool.map(work, range(threads)))
        dt = time.perf_counter() - t0
        out['all_cores'] = {'value': done / dt, 'unit': 'frames/s', 'cores': threads,
                            'sample': '%d frames in %.1f s, %d threads (os.cpu_count() = %d) each rendering whole frames'
                                      % (done, dt, threads, os.cpu_count() or 0)}
    return out


def partial_conv_reference_forward(self, input, mask_in=None):
    """PartialConv2d.forward in the reference's formulation (/root/reference/utils/partial_conv.py:39-83: a second convolution
    on the mask with the all-ones weight_maskUpdater, then five element-wise passes) on this package's module object -- what the
    fused epilogue (kbe_pconv_epilogue) is timed against here and compared with at 1024^2 in tests/test_hip_reference.py."""
    import torch.nn.functional as F
    fresh = mask_in is not None or self.last_size != tuple(input.shape)
    if fresh:
        self.last_size = tuple(input.shape)
        mask = mask_in if mask_in is not None else torch.ones(1, 1, *input.shape[2:], device=input.device)
        if self.multi_channel and mask.shape[1] == 1:
            mask = mask.expand(-1, self.in_channels, -1, -1)
        ones = self.weight_maskUpdater.to(input)
        self.update_mask = F.conv2d(mask, ones, None, self.stride, self.padding)
        self.mask_ratio = self.slide_winsize / (self.update_mask + 1e-8)
        self.update_mask = torch.clamp(self.update_mask, 0, 1)
        self.mask_ratio = self.mask_ratio * self.update_mask
    raw = F.conv2d(input * mask_in if mask_in is not None else input, self.weight, self.bias, self.stride, self.padding)
    if self.bias is not None:
        b = self.bias.view(1, -1, 1, 1)
        out = ((raw - b) * self.mask_ratio + b) * self.update_mask
    else:
        out = raw * self.mask_ratio
    return (out, self.update_mask) if self.return_mask else out


def pipeline_bench(args, device):
    """`bench.py --pipeline`: BASELINE.json configs[1] as ONE number -- a 512 x 512 image through the whole Pipeline
    (/root/reference/utils/pipeline.py:59-116: resize, Semantics + Disparity + Refine, point cloud, two inpaint passes, 64 frames
    delivered to host memory), networks with seeded random weights (checkpoints are a network download) -- and SURVEY.md 8d's
    "4b": the partial-convolution Inpaint forward at 1024^2 with PartialConv2d's mask bookkeeping fused into one HIP pass
    (partial_conv.py) against the reference's formulation (a second convolution on the mask + five element-wise passes,
    /root/reference/utils/partial_conv.py:62-77).  Prints one JSON line."""
    import warnings

    import torch.nn.functional as F
    from torch.utils.flop_counter import FlopCounterMode

    from ken_burns_effect_amd import common, kbe, partial_conv, synthetic
    from ken_burns_effect_amd.partial_inpainting import Inpaint as PartialInpaint
    from ken_burns_effect_amd.pipeline import Pipeline
    size = args.size if args.size != 1024 else 512
    frames_per_video = 64
    K, Wm = max(1, min(args.steps, 50)), max(1, min(args.warmup, 5))
    image, _ = synthetic.make_rgbd(size, size, 9)
    if args.miopen_find:
        torch.backends.cudnn.benchmark = True           # MIOpen's find step with a workspace (the default immediate mode gets none and falls back)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        pipe = Pipeline(model_paths=None, allow_random_weights=True, device=str(device), steps=frames_per_video)
    zoom = kbe.windows_for(size, size, dict.fromkeys(('startU', 'startV', 'startW', 'startH', 'endU', 'endV', 'endW', 'endH')), False)

    def timed(fn, reps, warm):
        for _ in range(warm):
            fn()
        ts = []
        for _ in range(reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        return float(np.median(ts)), ts

    call_s, call_ts = timed(lambda: pipe(image, zoom), K, Wm)
    # where the call's time goes: its three parts on their own (each synchronised: their sum exceeds the call by the overlap it loses)
    oc = pipe.objectCommon
    est_s, _ = timed(lambda: pipe.estimate(image), 5, 1)
    settings = {'dblSteps': np.linspace(0.0, 1.0, frames_per_video).tolist(), 'objectFrom': zoom['objectFrom'], 'objectTo': zoom['objectTo'], 'boolInpaint': True, 'dolly': False}

    def grow():
        common._reset_inpa(oc)
        with torch.no_grad():
            common.build_pointcloud(settings, oc, pipe.moduleInpaint)
    grow_s, _ = timed(grow, 5, 1)
    cams = common.frame_cameras(settings, oc)
    crop = common.crop_size(settings)
    host = torch.zeros(frames_per_video, size, size, 3, dtype=torch.uint8, pin_memory=True)
    loop_s, _ = timed(lambda: common.render_frames(cams, oc, crop, host_out=host), 10, 2)

    # the writers (/root/reference/utils/pipeline.py:120-134: cv2.imwrite per frame, moviepy -> ffmpeg mpeg4 for frames +
    # reversed[1:]): host work behind the delivered frames, timed as legs of their own -- the video file the package
    # writes where there is no ffmpeg binary (Motion-JPEG in an ISO base media file, each distinct frame encoded once by libkbe_jpeg.so
    # on host threads), round 5's writer beside it (Pillow: one frame at a time, every frame encoded), and the PNG frames (zlib on threads)
    import shutil
    import tempfile

    from ken_burns_effect_amd import pipeline as pipeline_mod
    frames_np = [f for f in host.numpy()]
    video = frames_np + frames_np[-2::-1]
    tmp = tempfile.mkdtemp(prefix='kbe_writers_')
    try:
        mp4 = os.path.join(tmp, '3d_kbe.mp4')
        video_s, _ = timed(lambda: pipeline_mod.write_video(mp4, video, fps=25), 5, 1)
        video_bytes = os.path.getsize(mp4)
        encoder = pipeline_mod.jpeg_encoder()[0]
        codec = 'mpeg4 (ffmpeg pipe)' if shutil.which('ffmpeg') else 'Motion-JPEG (quality 92, %s) in an ISO base media file' % ('libkbe_jpeg.so' if encoder == 'native' else 'Pillow')
        threads = pipeline_mod._writer_pool_size(len(frames_np))
        encode_s, _ = timed(lambda: pipeline_mod._jpegs(video, 92), 5, 1)
        os.environ['KBE_JPEG'] = 'pillow'               # round 5's writer: Pillow, one frame at a time, every frame of the way back encoded again
        try:
            serial_s, _ = timed(lambda: pipeline_mod.write_video(mp4, [f.copy() for f in video], fps=25), 2, 0)
        finally:
            del os.environ['KBE_JPEG']
        png_s, _ = timed(lambda: pipeline_mod.write_frames(os.path.join(tmp, 'frames'), frames_np), 3, 1)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)

    # 4b: the partial-convolution Inpaint forward at 1024^2, fused epilogue against the reference's formulation
    reference_forward = partial_conv_reference_forward
    big = 1024
    net = synthetic.seeded_fill_(PartialInpaint(), 5).to(device).eval()
    data = torch.randn(1, 68, big, big, device=device)
    mask = (torch.rand(1, 1, big, big, device=device) > 0.2).float()
    with torch.no_grad():
        net.normalize_images_disp(torch.rand(1, 3, big, big, device=device), torch.rand(1, 1, big, big, device=device), not_normed=True)
        # (tuned once per machine like the pipeline's own networks: the first run here pays MIOpen's find step for these shapes)
        from ken_burns_effect_amd.utils import miopen_tuned_once
        with miopen_tuned_once('bench-partial-inpaint-%d' % big, device, enabled=pipe.miopen_find == 'auto'):
            net.forward(tensorData=data, tensorMasks=mask)
        fused_s, _ = timed(lambda: net.forward(tensorData=data, tensorMasks=mask), 5, 2)
        with FlopCounterMode(display=False) as fc:
            net.forward(tensorData=data, tensorMasks=mask)
        fused_flops = fc.get_total_flops()
        fused_forward = partial_conv.PartialConv2d.forward
        partial_conv.PartialConv2d.forward = reference_forward
        for m in net.modules():                                 # the layers' cached mask statistics are the other formulation's
            if isinstance(m, partial_conv.PartialConv2d):
                m.last_size = (None, None, None, None)
        try:
            ref_s, _ = timed(lambda: net.forward(tensorData=data, tensorMasks=mask), 5, 2)
            with FlopCounterMode(display=False) as fc:
                net.forward(tensorData=data, tensorMasks=mask)
            ref_flops = fc.get_total_flops()
        finally:
            partial_conv.PartialConv2d.forward = fused_forward
    line = {
        'metric': 'pipeline_frames_per_sec_%dx%d' % (size, size), 'value': frames_per_video / call_s, 'unit': 'frames/s', 'n_gpus': 1, 'steps': K, 'warmup': Wm,
        'ms_per_step': call_s * 1e3, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': 'BASELINE configs[1]: %dx%d image -> Semantics + Disparity + Refine -> point cloud -> 2 inpaint passes -> %d frames in host memory; '
                               'a step = one whole video; seeded random weights' % (size, size, frames_per_video),
                   # ('auto': the find step ran in the first warm-up call if this machine had not tuned this size)
                   'miopen_find': True if args.miopen_find else pipe.miopen_find, 'call_ms': {'median': round(call_s * 1e3, 2), 'min': round(min(call_ts) * 1e3, 2), 'max': round(max(call_ts) * 1e3, 2)}},
        'stages_ms': {'estimate (resize + 3 networks + unprojection)': round(est_s * 1e3, 2), 'point cloud growth (2 x context net, 68-channel warp, Inpaint forward)': round(grow_s * 1e3, 2),
                      'frame loop (%d frames delivered)' % frames_per_video: round(loop_s * 1e3, 2)},
        'writers_ms': {'video file (%d frames = forth and back, %s)' % (len(video), codec): round(video_s * 1e3, 2), 'host_threads': threads, 'video_bytes': video_bytes,
                       'of which encoding the %d distinct frames' % frames_per_video: round(encode_s * 1e3, 2),
                       "round 5's writer (Pillow, one thread, all %d frames encoded)" % len(video): round(serial_s * 1e3, 2),
                       'png frames (%d, zlib level 1 on the writer threads)' % frames_per_video: round(png_s * 1e3, 2),
                       'image_to_video_file_ms': round((call_s + video_s) * 1e3, 2),
                       'note': 'host-side legs behind the delivered frames, not part of `value` (the reference: cv2.imwrite + moviepy/ffmpeg, /root/reference/utils/pipeline.py:120-134); '
                               'networks (estimate + growth) take %.1f ms of the call' % ((est_s + grow_s) * 1e3)},
        'partial_inpaint_1024': {'what': 'SURVEY 8d "4b": partial-convolution Inpaint.forward at 1024x1024, fp32 on MIOpen',
                                 'fused_epilogue_ms': round(fused_s * 1e3, 2), 'reference_formulation_ms': round(ref_s * 1e3, 2),
                                 'fused_conv_tflop': round(fused_flops / 1e12, 3), 'reference_conv_tflop': round(ref_flops / 1e12, 3),
                                 'fused_tflops': round(fused_flops / fused_s / 1e12, 1), 'reference_tflops': round(ref_flops / ref_s / 1e12, 1),
                                 'peak_fp32_matrix_tflops': 157.3},
    }
    print(json.dumps(line), flush=True)


def launch_ranks(n):
    """`python bench.py --gpus N` (N > 1, no WORLD_SIZE): start N ranks of this script under torch.distributed.run, one process per
    GPU, rendezvous on 127.0.0.1 -- the same command line the driver uses when it starts the ranks itself.  Rank 0's JSON line
    passes through on stdout; the exit status is the launcher's."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n), '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stderr.write('bench.py: starting %d ranks: %s\n' % (n, ' '.join(cmd)))
    return subprocess.call(cmd, env=dict(os.environ, OMP_NUM_THREADS=os.environ.get('OMP_NUM_THREADS', '8')))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=1024, help='timed frames per rank (a 128-frame video repeated is the same work; long enough for the queue ramp to vanish)')
    ap.add_argument('--warmup', type=int, default=128, help='untimed frames first (the clocks need a few ms of load: 8 frames read 6 %% slower than 256)')
    ap.add_argument('--size', type=int, default=1024)
    ap.add_argument('--dolly', action='store_true')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-crop', action='store_true')
    ap.add_argument('--device-only', action='store_true',
                    help='leave the finished frames in HBM: `value` is then NOT the SURVEY 8d step (dev comparison; config.delivery says so)')
    ap.add_argument('--no-overlap', action='store_true', help='with --batch: transfers on the compute stream (dev comparison)')
    ap.add_argument('--batch', type=int, default=None,
                    help='hand-off (include/kbe.h): default groups of 8 frames per hipMemcpyAsync, the lanes taking turns; 0: per-frame copy kernel; '
                         '> 0: staged ring, one hipMemcpyAsync per BATCH frames on a copy stream')
    ap.add_argument('--pipeline', action='store_true', help='BASELINE configs[1] as a whole (512^2 image -> nets -> 64 delivered frames) and the 1024^2 partial-conv '
                                                            'Inpaint forward; a separate line, not the headline metric')
    ap.add_argument('--miopen-find', action='store_true', help='with --pipeline: torch.backends.cudnn.benchmark = True (MIOpen find step with a workspace)')
    ap.add_argument('--upsample', type=int, default=1, help='cloud of (upsample * size)^2 points (BASELINE configs[4]: 2; implies --cloud raw)')
    ap.add_argument('--video-frames', type=int, default=0,
                    help='STRONG scaling: one video of this many frames in all, sharded over the ranks (BASELINE configs[2]: 128); a step is still a frame, '
                         'a pass renders the whole video once, --steps is ignored')
    ap.add_argument('--cloud', choices=['inpaint', 'raw'], default='inpaint',
                    help='inpaint: grow the cloud with the (seeded) Inpaint network as the pipeline does; raw: image pixels only')
    args = ap.parse_args()

    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')     # before the first HIP call: RCCL needs dmabuf IPC on this stack
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(launch_ranks(args.gpus))       # `python bench.py --gpus N`: one process per GPU, started here
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world_size = int(os.environ.get('WORLD_SIZE', '1'))
    if world_size != args.gpus:
        sys.exit('bench.py: --gpus %d but WORLD_SIZE=%d (start it as `python bench.py --gpus N`, or under torch.distributed.run with '
                 '--nproc-per-node N)' % (args.gpus, world_size))
    launch_only = os.environ.get('KBE_BENCH_LAUNCH_ONLY') == '1'        # tests/test_bench_launcher.py: rendezvous + collectives, no rendering
    if not launch_only and not torch.cuda.is_available():
        sys.exit('bench.py needs a GPU (the product path has no CPU fallback)')
    # one process per GPU.  (KBE_DIST_BACKEND=gloo lets the multi-rank code path be exercised on a box with fewer
    # GPUs than ranks -- ranks then share devices; such a line carries "scaling_valid": false.)
    backend = os.environ.get('KBE_DIST_BACKEND', 'nccl')
    n_dev = torch.cuda.device_count() if torch.cuda.is_available() else 0
    scaling_valid, scaling_note = True, None
    if world_size > 1 and backend == 'nccl' and n_dev < world_size and not launch_only:
        sys.exit('bench.py: --gpus %d but this box shows %d GPU(s): one process per GPU over RCCL needs %d devices '
                 '(KBE_DIST_BACKEND=gloo shares devices between ranks for a functional check -- not a measurement)' % (world_size, n_dev, world_size))
    if n_dev:
        dev_index = local_rank if backend == 'nccl' else local_rank % n_dev
        torch.cuda.set_device(dev_index)
        device = torch.device('cuda', dev_index)
    else:
        dev_index, device = 0, torch.device('cpu')
    ranks_seen = 1
    if world_size > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if backend == 'nccl' and n_dev:
            # RCCL over xGMI.  Should the communicator fail to come up -- on every rank, as configuration errors do -- the run
            # goes on with gloo for its three collectives (one cloud broadcast, barriers, a max over ranks), and the line says
            # so LOUDLY: "scaling_valid": false, config.collectives names the failure, stderr carries the exception
            try:
                dist.init_process_group('nccl', rank=rank, world_size=world_size, device_id=device)
                probe = torch.ones(1, device=device)
                dist.all_reduce(probe)
                torch.cuda.synchronize()
                ranks_seen = int(probe.item())
            except Exception as exc:                            # noqa: BLE001
                sys.stderr.write('bench.py: RCCL DID NOT COME UP (%s: %s); continuing on gloo -- this is NOT an xGMI measurement\n' % (type(exc).__name__, exc))
                if dist.is_initialized():
                    dist.destroy_process_group()
                backend = 'gloo (RCCL failed: %s)' % type(exc).__name__
                scaling_valid, scaling_note = False, 'RCCL failed to initialise; collectives ran on gloo over TCP'
                dist.init_process_group('gloo', rank=rank, world_size=world_size)
        else:
            if backend == 'nccl':
                backend = 'gloo'                                # launch-only check on a box without GPUs
            dist.init_process_group(backend.split()[0], rank=rank, world_size=world_size)
            scaling_valid, scaling_note = False, 'KBE_DIST_BACKEND=%s: ranks may share devices; functional check only' % backend
        if not backend.startswith('nccl'):
            probe = torch.ones(1)
            dist.all_reduce(probe)
            ranks_seen = int(probe.item())
        if ranks_seen != world_size:
            sys.exit('bench.py: the all-reduce of ones over %s returned %d, expected %d ranks' % (backend, ranks_seen, world_size))
    if launch_only:
        if rank == 0:
            print(json.dumps({'launcher': 'ok', 'world_size': world_size, 'ranks_seen': ranks_seen, 'collectives': backend if world_size > 1 else None,
                              'scaling_valid': scaling_valid}), flush=True)
        if world_size > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    if args.pipeline:
        if world_size > 1:
            sys.exit('bench.py --pipeline times one video on one GPU')
        return pipeline_bench(args, device)

    from ken_burns_effect_amd import common, sharding, synthetic
    # multi-GPU: every rank's frames land in pinned memory of the NUMA node its GPU hangs off (best effort; the CPU
    # baseline of a single-GPU run wants all cores, so N = 1 is left alone)
    numa_node = sharding.bind_to_gpu_numa_node(dev_index) if world_size > 1 else None

    size = args.size
    ofrom, oto = synthetic.default_windows(size, size, args.dolly)
    strong = args.video_frames > 0
    total_steps = args.video_frames if strong else args.steps * world_size
    settings = {'dblSteps': np.linspace(0.0, 1.0, max(total_steps, 2)).tolist()[:total_steps], 'objectFrom': ofrom,
                'objectTo': oto, 'boolInpaint': False, 'dolly': args.dolly}
    crop = None if args.no_crop else common.crop_size(settings)

    # rank 0 owns the scene; other ranks receive the cloud through the broadcast below
    oc = build_scene(size, device, args.cloud == 'inpaint' and not args.dolly and args.upsample == 1, settings, args.upsample) if rank == 0 else {}
    if world_size > 1:
        if rank == 0 and not args.device_only:
            # the lanes a delivered video uses are measured ONCE, here, and travel with the cloud's header: the other ranks do not
            # run the timing probe against each other's host traffic (sharding.measure_delivery_lanes)
            sharding.measure_delivery_lanes(dict(settings, dblSteps=sharding.shard_steps(settings['dblSteps'], 0, world_size)[1], boolCrop=not args.no_crop), oc)
        sharding.broadcast_cloud(oc, device)          # untimed warm-up of the communicator + fills `oc` everywhere
    n_points = oc['tensorInpaPoints'].shape[-1]
    _, my_steps = sharding.shard_steps(settings['dblSteps'], rank, world_size)
    cams = common.frame_cameras(dict(settings, dblSteps=my_steps), oc)
    n_mine = len(cams)                      # args.steps, or this rank's share of --video-frames

    # landing buffers of the timed runs are allocated (and every page touched) here, not in the loop
    delivery = 'hbm' if args.device_only else 'pinned_host'
    host_out = None if args.device_only else torch.zeros(max(n_mine, 1), size, size, 3, dtype=torch.uint8, pin_memory=True)
    dev_out = torch.zeros(max(n_mine, 1), size, size, 3, dtype=torch.uint8, device=device)

    def run_host():
        return common.render_frames(cams, oc, crop, host_out=host_out, overlap=not args.no_overlap, batch=args.batch)

    def run_device():
        return common.render_frames(cams, oc, crop, keep_on_device=True, host_out=dev_out)

    # warm-up (untimed): W frames on both routes (clocks, staging buffers, streams, first-touch of the pinned pages)
    nw = max(min(args.warmup, n_mine), 1)
    common.render_frames(cams[:nw], oc, crop, keep_on_device=True, host_out=dev_out[:nw])
    if host_out is not None:
        common.render_frames(cams[:nw], oc, crop, host_out=host_out[:nw], overlap=not args.no_overlap, batch=args.batch)

    def sync():
        torch.cuda.synchronize()
        if world_size > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # the one exchange step of a video: the finished cloud goes from rank 0 to every rank (RCCL broadcast).  It is
    # scene set-up like build_scene above, not part of a frame step, so it is timed on its own and reported as
    # `cloud_broadcast_ms` beside the frame rate (a video pays it once, whatever its length)
    broadcast_ms = None
    if world_size > 1:
        sync()
        t0 = time.perf_counter()
        sharding.broadcast_cloud(oc, device)
        sync()
        broadcast_ms = (time.perf_counter() - t0) * 1e3

    own_times = []

    def timed_pass(run):
        """EXACTLY K frames between barrier + synchronize on both sides; max over ranks."""
        sync()
        t0 = time.perf_counter()
        run()
        sync()
        dt = time.perf_counter() - t0
        own_times.append(dt)            # (this rank's own clock, before the max over the ranks: `ranks` in the line)
        if world_size > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    def timed(run, min_wall=0.5, min_passes=3, max_passes=200):
        """Passes of K frames until >= min_wall seconds of timed wall (a single short pass mostly measures the ramp of
        the lanes' queues); every rank takes the same number of passes (the max-reduced times are identical on all)."""
        times = []
        while len(times) < min_passes or (sum(times) < min_wall and len(times) < max_passes):
            times.append(timed_pass(run))
        return times

    # THE timed region: passes of K frames, cloud resident in HBM, finished uint8 frames delivered to pinned host memory
    # (--device-only: left in HBM); the HBM-resident rate is measured the same way and reported beside it
    times = timed(run_device if args.device_only else run_host)
    elapsed = float(np.median(times))
    own_elapsed = float(np.median(own_times))
    times_dev = times if args.device_only else timed(run_device)
    elapsed_dev = float(np.median(times_dev))

    # the delivered frames against the frames left in HBM (two runs of the same cameras; the accumulation order may differ in the
    # last bit): the first and the last frame of the pass, every value
    frames_check = None
    if host_out is not None:
        worst = 0
        for k in (0, n_mine // 2, n_mine - 1):
            d = (host_out[k].to(torch.int16) - dev_out[k].cpu().to(torch.int16)).abs()
            worst = max(worst, int(d.max()))
            frames_check = max(frames_check or 0.0, float((d > 0).float().mean()))
        frames_check = {'max_abs_diff': worst, 'worst_fraction_differing': frames_check, 'frames': [0, n_mine // 2, n_mine - 1],
                        'ok': worst <= 1 and frames_check < 1e-3}
        if not frames_check['ok']:
            sys.stderr.write('bench.py: DELIVERED FRAMES DIFFER from the frames left in HBM: %s\n' % frames_check)

    # every rank's own account of the timed region (multi-rank runs): its device, the NUMA node it bound to (or why it did not), its
    # own time per pass and what that is on its PCIe link -- a node where one rank's link or socket lags shows here, not in the max
    rank_rows = None
    if world_size > 1:
        # (one all_gather of six numbers per rank on the collectives' own device -- the kind of call the timed region's max-reduction makes --
        # rather than a pickled object: the path must hold under RCCL on a node nobody has run it on)
        mine = torch.tensor([rank, dev_index, -1 if numa_node is None else numa_node, sharding.NUMA_BIND['code'], n_mine, own_elapsed], dtype=torch.float64, device=device)
        rows = [torch.zeros_like(mine) for _ in range(world_size)]
        dist.all_gather(rows, mine)
        rank_rows = []
        for row in rows:
            r, d, node, code, frames, secs = row.cpu().tolist()
            rank_rows.append({'rank': int(r), 'device': int(d), 'numa_node': None if node < 0 else int(node), 'numa': sharding.NUMA_CODES.get(int(code), '?'), 'frames': int(frames),
                              'ms_per_pass': round(secs * 1e3, 4), 'pcie_GBs': None if args.device_only else round(frames * size * size * 3 / max(secs, 1e-12) / 1e9, 2)})

    if rank == 0:
        from ken_burns_effect_amd import _native
        lanes = max(1, min(_native.MAX_LANES, int(os.environ.get('KBE_LANES', _native.DEFAULT_LANES))))
        host_lanes = _native.kernels().delivery_lanes(common._prepared_cloud(_native.kernels(), oc), cams, oc['dblBaseline'], crop)   # measured once per cloud
        # the route and the frames per launch of the timed region (_native.video_launch_shape: the fused route with four frames
        # per launch where the frames are delivered to host memory; a zoom-out or a cloud denser than the raster: the bucket route)
        state = common._prepared_cloud(_native.kernels(), oc)
        video_flags, group_used, fused_used = _native.kernels().video_launch_shape(state, cams, args.batch, to_host=not args.device_only)
        route = 'fused' if fused_used else 'bucket'
        group_frames = group_used if group_used > 1 else GROUP_FRAMES
        fill_flags = 32 | (512 if video_flags & 1 else 0)           # KBE_STAGE_FILL_BY_COUNT, + KBE_STAGE_FILL_DIST where the loop fills with the tables
        # the camera paths the scatter's launch is priced on: this run's own, the product's 75 steps, the driver's 20 (consecutive
        # cameras in groups of `group_frames`: time_kernels)
        paths = {}
        if not args.dolly:
            for n_steps in dict.fromkeys((total_steps, PRODUCT_STEPS, DRIVER_STEPS)):
                paths[str(n_steps)] = common.frame_cameras(dict(settings, dblSteps=np.linspace(0.0, 1.0, max(n_steps, 2)).tolist()[:n_steps]), oc)
        kt = time_kernels(oc, cams, route, fill_rect=None if crop is None else common.crop_window(size, size, crop[0], crop[1]), group_frames=group_frames,
                          fill_flags=fill_flags, paths=paths, fill_group=group_used if (video_flags & 1) else 1)
        HW = size * size
        # roofline = the scatter (render_pointcloud, common.py:428-686: z-buffer clear + z-splat + degrid + accumulate +
        # normalise), SURVEY.md 8d: algorithmic bytes 28 N + 20 HW (every input once, every output once, no scratch) per frame
        # over the HIP-event time of the launches that implement it, back to back alone on a stream -- for the route and the
        # number of frames per launch the timed region took (fused: k_place + k_frame; bucket: k_project + k_tiles, the tile
        # launch clearing the other frame's z-buffer), with the one-frame launches and the other route beside it.
        scatter_bytes = 28 * n_points + 20 * HW
        # (the fused route's tile launch also makes the next group's placements -- k_frame_ahead / k_frame_group_ahead -- unless KBE_AHEAD=0)
        ahead = os.environ.get('KBE_AHEAD') != '0' and 'fused:scatter_group_ahead' in kt and 'fused:scatter_ahead' in kt
        route_launches = {'fused': ['k_frame_ahead'] if ahead else ['k_place', 'k_frame'], 'bucket': ['k_project', 'k_tiles'], 'fused:two_launches': ['k_place', 'k_frame']}
        # the committed PMC passes are of the default workload and of configs[4] (2048^2 from 16.8 M points) only
        default_workload = size == 1024 and args.cloud == 'inpaint' and not args.dolly and args.upsample == 1
        config4_workload = size == 2048 and args.upsample == 2 and not args.dolly
        per_kernel, traffic_src = measured_traffic() if default_workload else (measured_traffic('_config4') if config4_workload else ({}, None))
        insts, insts_src = measured_instructions() if default_workload else ({}, None)

        def roof(r, frames):
            if r == 'bucket':
                frames = min(frames, BUCKET_GROUP_MAX)
            names = route_launches[r]
            one_launch = names == ['k_frame_ahead']
            r = r.split(':')[0]
            t = kt[r + (':scatter_group' if frames > 1 else ':scatter') + ('_ahead' if one_launch else '')]
            tr = sum(per_kernel[k] for k in names) if all(k in per_kernel for k in names) else None
            tr_launch = None if tr is None else frames * tr
            # a group launch measured as such (the frames of a group share their candidate lists: the one-frame launch's figures
            # times the frames overstate it)
            group_name = 'k_frame_group_ahead' if one_launch else ('k_frame_group' if names == ['k_place', 'k_frame'] else None)
            g_traffic = per_kernel.get('by_frames_per_launch', {}).get(group_name, {}).get(str(frames)) if frames > 1 and one_launch else None
            g_insts = insts.get('by_frames_per_launch', {}).get(group_name, {}).get(str(frames)) if frames > 1 and one_launch else None
            if g_traffic:
                tr_launch = g_traffic['hbm_bytes']
            out = {'route': r, 'kernel': ' + '.join(n.replace('k_frame', 'k_frame_group').replace('k_tiles', 'k_tiles_group').replace('k_project', 'k_project_group')
                                                    if frames > 1 else n for n in names), 'frames_per_launch': frames,
                   'us': round(t * 1e6, 2), 'us_per_frame': round(t * 1e6 / frames, 2), 'algorithmic_bytes': frames * scatter_bytes,
                   'achieved': frames * scatter_bytes / t / 1e9, 'frac': frames * scatter_bytes / t / 1e9 / HBM_PEAK_GBS,
                   'traffic': tr_launch}
            if g_traffic:
                out['traffic_note'] = 'PMC on launches of %d frames' % frames
            if g_insts or all(k in insts for k in names):
                # the other roofline these launches live under: wave-level VALU instructions per frame over the chip's issue rate
                n_valu = g_insts['valu'] / frames if g_insts else sum(insts[k]['valu'] for k in names)
                out['valu_issue'] = {'insts_per_frame': n_valu, 'us_at_peak': round(n_valu / VALU_ISSUE_PER_US, 2),
                                     'frac': n_valu / VALU_ISSUE_PER_US / (t * 1e6 / frames)}
            return out
        frames_per_launch = group_used
        main = roof(route, frames_per_launch)
        single = roof(route, 1)
        grouped = roof(route, group_frames)
        other = roof('bucket' if route == 'fused' else 'fused', group_frames)
        two_launches = roof('fused:two_launches', group_frames) if route == 'fused' and ahead else None
        # the group launch on consecutive cameras of real paths (the frames of a group share candidate lists built for the box
        # between the group's first and last camera: identical cameras are the best case).  The line's `roofline` is priced on the
        # PRODUCT's path (75 steps, /root/reference/kbe.py:104) when the timed region launches groups; this run's own path, the
        # driver's 20 steps and the identical-camera figure are kept beside it.
        by_path = {}
        for label in paths:
            t = kt.get('fused:scatter_group_ahead:consecutive:' + label)
            if t:
                fpl = min(group_frames, len(paths[label]))      # (a path shorter than a launch: one group of all its frames)
                by_path[label] = {'steps': int(label), 'groups': len(consecutive_groups(paths[label], group_frames)), 'frames_per_launch': fpl,
                                  'us': round(t * 1e6, 2), 'us_per_frame': round(t * 1e6 / fpl, 2),
                                  'achieved': fpl * scatter_bytes / t / 1e9, 'frac': fpl * scatter_bytes / t / 1e9 / HBM_PEAK_GBS,
                                  'rounds_us': [round(x * 1e6, 2) for x in kt['fused:scatter_group_ahead:consecutive:' + label + ':rounds']]}
        cameras_note, identical = 'identical (one camera, %d copies per launch)' % frames_per_launch, None
        if route == 'fused' and ahead and frames_per_launch == group_frames and str(PRODUCT_STEPS) in by_path:
            c = by_path[str(PRODUCT_STEPS)]
            identical = {k: main[k] for k in ('us', 'us_per_frame', 'achieved', 'frac')}
            main = dict(main, us=c['us'], us_per_frame=c['us_per_frame'], achieved=c['achieved'], frac=c['frac'])
            if 'valu_issue' in main:
                main['valu_issue'] = dict(main['valu_issue'], frac=main['valu_issue']['us_at_peak'] / c['us_per_frame'])
            cameras_note = 'consecutive: groups of %d consecutive cameras of a %d-step path (the last group: the path\'s last %d), group k placing group k + 1 ahead' % (group_frames, PRODUCT_STEPS, group_frames)
        cloud = ('raw' if args.dolly else args.cloud) if args.upsample == 1 else '%dx-upsampled' % args.upsample ** 2
        line = {
            'metric': 'novel_view_frames_per_sec_%dx%d' % (size, size), 'value': total_steps / elapsed,
            'unit': 'frames/s', 'n_gpus': world_size, 'steps': total_steps if strong else args.steps, 'warmup': args.warmup,
            # (weak: K frames per rank and pass; strong, --video-frames: ONE video of K frames per pass, sharded over the ranks)
            'ms_per_step': elapsed / (total_steps if strong else args.steps) * 1e3, 'higher_is_better': True, 'scaling': 'strong' if strong else 'weak', 'vs_baseline': None,
            'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': '%dx%d %s path, %d pts (%s cloud), frame=shift+scatter+fill+u8%s' % (
                           size, size, 'dolly' if args.dolly else 'KBE', n_points, cloud, '' if crop is None else '+crop/resize'),
                       'delivery': delivery, 'scatter_route': route, 'frames_per_rank': n_mine, 'lanes': lanes if args.device_only else host_lanes,
                       'device_only_lanes': lanes, 'passes': len(times),
                       'pass_ms': {'median': round(elapsed * 1e3, 3), 'min': round(min(times) * 1e3, 3), 'max': round(max(times) * 1e3, 3)},
                       'sharding': 'frames %s over ranks (sharding.shard_indices); one cloud broadcast (set-up, cloud_broadcast_ms)' % os.environ.get('KBE_SHARD_SHAPE', sharding.SHARD_SHAPE),
                       'handoff': ('sdma' if _native.handoff_by_sdma() else 'blit (hipMemcpyAsync)') if not args.device_only else None,
                       'collectives': backend if world_size > 1 else None, 'ranks_seen': ranks_seen},
            'device_only': {'value': total_steps / elapsed_dev, 'unit': 'frames/s', 'ms_per_step': elapsed_dev / (total_steps if strong else args.steps) * 1e3,
                            'passes': len(times_dev), 'note': 'same K frames left in HBM (no PCIe hand-off)'},
            'roofline': {'bound': 'hbm', 'kernel': main['kernel'] + ' (the scatter = render_pointcloud, %s route)' % route,
                         'achieved': main['achieved'], 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': main['frac'],
                         'traffic': main['traffic'], 'traffic_source': traffic_src, 'algorithmic_bytes': main['algorithmic_bytes'],
                         'frames_per_launch': frames_per_launch, 'us': main['us'], 'us_per_frame': main['us_per_frame'],
                         'cameras': cameras_note, 'by_path': by_path, 'identical_cameras': identical,
                         'formula': '28 N + 20 HW per frame (SURVEY.md 8d)',
                         'valu_issue': main.get('valu_issue'), 'valu_issue_source': insts_src,
                         'one_frame_per_launch': single, 'grouped': grouped, 'other_route': other, 'placement_launch_in_front': two_launches,
                         'note': 'launches timed alone on one stream, back to back (HIP events, 40 repetitions), with the number of frames per '
                                 'launch the timed region uses; fused route: ONE launch per group -- the tile launch of a group also makes the placements '
                                 'of the next group, so a launch holds all of the scatter\'s work for its frames (`cameras` says which cameras a group holds); '
                                 'the matching rocprofv3 --stats summary is profiles/*scatter_group*_kernel_stats.csv '
                                 '(in the timed region the kernels of several lanes overlap and per-kernel durations stretch)',
                         'kernel_us': {k: ([round(x * 1e6, 2) for x in v] if isinstance(v, list) else round(v * 1e6, 2)) for k, v in kt.items()}},
        }
        # the hole fill (fill_disocclusion, common.py:833-937) under the same roofline: SURVEY.md 8d's 36 HW bytes per frame (the render
        # and the depth plane in, the render out) over the HIP-event time of its launches (k_hole_dist + k_fill_tables + k_fill_holes
        # where the loop fills with the tables, else k_fill_holes), mean over eight cameras along the path.  Where it takes longer per
        # frame than the scatter -- a dolly zoom without inpainting -- it is the line's `roofline`, the scatter's figures beside it.
        fill_t = kt['fill_along_path']
        fill_obj = {'bound': 'hbm', 'kernel': ('k_hole_dist + k_fill_tables + k_fill_holes' if video_flags & 1 else 'k_fill_holes') + ' (fill_disocclusion)',
                    'achieved': 36 * HW / max(fill_t, 1e-9) / 1e9, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': 36 * HW / max(fill_t, 1e-9) / 1e9 / HBM_PEAK_GBS,
                    'traffic': None, 'algorithmic_bytes': 36 * HW, 'formula': '36 HW per frame (SURVEY.md 8d)', 'us_per_frame': round(fill_t * 1e6, 2),
                    'us_worst_of_eight_cameras': round(kt['fill_along_path_max'] * 1e6, 2),
                    'note': 'a frame on its own on one stream (HIP events, 40 repetitions), with and without its fill, eight cameras along the path; '
                            'the walk is bound by instruction issue and the number of holes, not by these bytes'}
        if 'fill_group_along_path' in kt:
            # the fill's launches as the video loop makes them: `frames_per_launch` frames each (the single-frame figure stays beside it)
            g_t, g_n = kt['fill_group_along_path'], kt['fill_group_frames']
            fill_obj['one_frame_per_launch'] = {k: fill_obj[k] for k in ('achieved', 'frac', 'us_per_frame', 'us_worst_of_eight_cameras')}
            fill_obj.update({'frames_per_launch': g_n, 'us': round(g_t * 1e6, 2), 'us_per_frame': round(g_t * 1e6 / g_n, 2), 'algorithmic_bytes': g_n * 36 * HW,
                             'achieved': g_n * 36 * HW / max(g_t, 1e-9) / 1e9, 'frac': g_n * 36 * HW / max(g_t, 1e-9) / 1e9 / HBM_PEAK_GBS,
                             'us_worst_of_eight_cameras': round(kt['fill_group_along_path_max'] * 1e6 / g_n, 2),
                             'note': 'the fill launches of %d consecutive frames (what the video loop launches: KBE_VIDEO_FILL_GROUP) alone on one stream (HIP events, 40 repetitions), with and '
                                     'without their fill, at eight places along the path; `one_frame_per_launch`: a frame on its own; the walk is bound by instruction issue, '
                                     'dependent look-ups and the number of holes, not by these bytes' % g_n})
        if fill_t * 1e6 > single['us_per_frame']:          # (like with like: both a frame on its own)
            fill_obj['scatter'] = line['roofline']
            line['roofline'] = fill_obj
        else:
            line['roofline']['fill'] = fill_obj
        if frames_check is not None:
            line['frames_check'] = frames_check
            if not frames_check['ok']:          # a rate measured on wrong frames is not a measurement: no headline number, exit status 3
                line['invalid'] = 'delivered frames differ from the frames left in HBM'
                line['value_of_the_wrong_frames'], line['value'] = line['value'], None
        if not args.device_only:
            line['pcie'] = {'achieved': n_mine * size * size * 3 / elapsed / 1e9, 'unit': 'GB/s per GPU (rank 0)', 'peak': 63.0,
                            'note': 'uint8 frames of %.2f MB over PCIe Gen5 x16 (63 GB/s spec, ~57 measured with hipMemcpyAsync on an idle chip)'
                                    % (size * size * 3 / 1e6)}
        if world_size > 1:
            line['ranks'] = rank_rows
            line['scaling_valid'] = scaling_valid
            if scaling_note:
                line['scaling_note'] = scaling_note
        if broadcast_ms is not None:
            line['cloud_broadcast_ms'] = broadcast_ms
            line['config']['rank0_numa_node'] = numa_node
        if world_size == 1 and not args.no_cpu_baseline:
            line['cpu_baseline'] = cpu_baseline(oc, cams, crop)
        print(json.dumps(line), flush=True)
    if world_size > 1:
        dist.barrier()                  # rank 0 is still timing single kernels: leave together
        dist.destroy_process_group()
    if frames_check is not None and not frames_check['ok']:
        sys.exit(3)


if __name__ == '__main__':
    main()
