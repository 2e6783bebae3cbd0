#!/usr/bin/env python3
"""The fused scatter alone on a stream, us per frame by frames per launch: the two launches of a group (k_place, k_frame_group:
kbe_render_frame_group_fused) against the pipelined form (kbe_render_frame_group_ahead: ONE launch per group in steady state,
the tile launch making the next group's placements), and a check that both render the same frames.  KBE_LIB_PATH: a variant
build (dev aid)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from ken_burns_effect_amd import _native, common, synthetic  # noqa: E402

if os.environ.get('KBE_LIB_PATH'):
    _native._lib, _native._kernels, _native.LIB_PATH = None, None, os.environ['KBE_LIB_PATH']
# (dev) PAD_MB: device memory taken before anything else -- shifts every later allocation (is the launch's time a matter of addresses?)
_pad = torch.empty(int(float(os.environ.get('PAD_MB', '0')) * 1e6), dtype=torch.uint8, device='cuda') if os.environ.get('PAD_MB') else None
size = int(os.environ.get('SIZE', '1024'))
reps = int(os.environ.get('REPS', '40'))
ofrom, oto = synthetic.default_windows(size, size, False)
settings = {'dblSteps': [i / 63 for i in range(64)], 'objectFrom': ofrom, 'objectTo': oto, 'boolInpaint': True, 'dolly': False}
oc = bench.build_scene(size, torch.device('cuda:0'), os.environ.get('CLOUD', 'inpaint') == 'inpaint', settings, 1)
cams = common.frame_cameras(settings, oc)
K = _native.kernels()
state = common._prepared_cloud(K, oc)
K._pack(state)
Bl = oc['dblBaseline']
out = torch.empty(12, size, size, 3, dtype=torch.uint8, device='cuda')
ref = torch.empty_like(out)
K.group_scratch(state, 12)


def timed(fn, n):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps / n


# correctness: a sequence of four different groups, pipelined, against the same groups with their placement launches in front
for n in (() if os.environ.get('SKIP_CHECK') == '1' else (1, 5, 12)):
    groups = [[cams[(7 * g + k) % len(cams)] for k in range(n)] for g in range(4)]
    want = []
    for group in groups:
        K.render_frame_group_fused(state, group, Bl, ref[:n], stages=6)          # parity -1: counters zeroed in front, left zeroed
        want.append(ref[:n].clone())
    # control: the same classic call again (the order of the list atomics differs from run to run)
    K.render_frame_group_fused(state, groups[0], Bl, ref[:n], stages=6)
    dc = (ref[:n].int() - want[0].int()).abs()
    print('frames per launch %2d: control (placed-in-front twice), max |diff| %d, differing values %.2e' % (n, int(dc.max()), float((dc > 0).float().mean())))
    worst, frac = 0, 0.0
    for g, group in enumerate(groups):
        K.render_frame_group_ahead(state, group, Bl, out[:n], turn=g, placed=g > 0, next_cameras=groups[g + 1] if g + 1 < len(groups) else None, stages=6)
        d = (out[:n].int() - want[g].int()).abs()
        worst = max(worst, int(d.max()))
        frac = max(frac, float((d > 0).float().mean()))
    print('frames per launch %2d: pipelined vs placed-in-front, max |diff| %d, differing values at most %.2e' % (n, worst, frac))
    del want
torch.cuda.synchronize()

focal, shift3 = cams[len(cams) // 2]
for n in ((1, 2, 4, 8, 12) if os.environ.get('IDENTICAL', '1') == '1' else ((12,) if os.environ.get('IDENTICAL') == '12' else ())):
    group = [(focal, shift3)] * n
    par = [0]

    def classic():
        K.render_frame_group_fused(state, group, Bl, out[:n], stages=2, parities=[par[0] & 1] * n)
        par[0] += 1
    t_classic = timed(classic, n)
    K.render_frame_group_fused(state, group, Bl, out[:n], stages=6)        # parity -1: leaves the sets' counters zeroed
    turn = [0]

    def ahead():
        K.render_frame_group_ahead(state, group, Bl, out[:n], turn=turn[0], placed=turn[0] > 0, next_cameras=group, stages=2)
        turn[0] += 1
    t_ahead = timed(ahead, n)
    # ... and with the argument arrays built once (bench.py's way: the wrapper above costs the host ~100 us per call)
    launch = K.prepared_group_ahead(state, group, Bl, out[:n], group, stages=2)

    def ahead_prepared():
        launch(turn[0], True)
        turn[0] += 1
    t_prepared = timed(ahead_prepared, n)
    K.render_frame_group_ahead(state, group, Bl, out[:n], turn=turn[0], placed=True, next_cameras=None, stages=6)      # the sequence ends: nothing placed ahead
    torch.cuda.synchronize()
    K.render_frame_group_fused(state, group, Bl, out[:n], stages=6)
    print('%2d frame(s) per launch: k_place + k_frame %.2f us per frame, pipelined (one launch) %.2f us per frame (arguments prepared: %.2f)' % (n, t_classic, t_ahead, t_prepared))


# ---- the launch on a video's OWN groups: group k = cameras [n k, n k + n) of a path of PATHS steps (the last group: the path's last n), the
# launch of group k placing group k + 1 ahead -- the frames of a group share their candidate lists, built for the box between the
# group's first and last camera, so twelve copies of one camera (above) are that scheme's best case
import ctypes  # noqa: E402
stats = (ctypes.c_ulonglong * 8)() if hasattr(K.lib, 'kbe_debug_frame_stats') else None
for steps in [int(v) for v in os.environ.get('PATHS', '1024,75,20').split(',') if v]:
    path = common.frame_cameras(dict(settings, dblSteps=[i / max(steps - 1, 1) for i in range(steps)]), oc)
    for n in [int(v) for v in os.environ.get('LAUNCH_FRAMES', '12,8,4').split(',') if v]:
        groups = bench.consecutive_groups(path, n)
        launches = [K.prepared_group_ahead(state, g, Bl, out[:n], groups[(k + 1) % len(groups)], stages=2) for k, g in enumerate(groups)]
        turn = [0]

        def run():
            launches[turn[0] % len(launches)](turn[0], True)
            turn[0] += 1
        K.render_frame_group_ahead(state, groups[-1], Bl, out[:n], turn=0, placed=False, next_cameras=groups[0], stages=2)     # group 0's placements
        turn[0] = 1
        launches = launches[-1:] + launches[:-1]                    # launch index = turn: turn 1 renders group 0
        # whole cycles of the path's groups per timed round (at least `reps` launches)
        cycles = max(1, (reps + len(groups) - 1) // len(groups))
        per = []
        for _ in range(5):
            for _ in range(cycles * len(groups)):
                run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(cycles * len(groups)):
                run()
            e1.record()
            torch.cuda.synchronize()
            per.append(e0.elapsed_time(e1) * 1e3 / (cycles * len(groups)) / n)
        note = ''
        if stats is not None:
            K.lib.kbe_debug_frame_stats(stats, 1)
            for _ in range(len(groups)):
                run()
            torch.cuda.synchronize()
            K.lib.kbe_debug_frame_stats(stats, 1)
            t = max(1, stats[0])
            note = ' | per tile: list entries %.1f, candidates %.1f, in z reach %.0f, records %.0f; slow tiles %d, second round %d of %d' % (
                stats[1] / t * n, stats[2] / t, stats[3] / t, stats[4] / t, stats[5], stats[6], stats[0])
        # leave the sets clean: the sequence ends with a launch that places nothing
        g_last = groups[(turn[0] - 1) % len(groups)]
        K.render_frame_group_ahead(state, g_last, Bl, out[:n], turn=turn[0], placed=True, next_cameras=None, stages=6)
        torch.cuda.synchronize()
        K.render_frame_group_fused(state, g_last, Bl, out[:n], stages=6)
        per.sort()
        print('consecutive cameras, %4d-step path, %2d frames per launch (%d groups): %.2f us per frame (rounds %s)%s'
              % (steps, n, len(groups), per[len(per) // 2], ' '.join('%.2f' % v for v in per), note))


# ---- what a frame costs ALONG the path: twelve copies of the camera at POSITIONS of a 75-step path (identical cameras: the lists'
# best case everywhere, so what differs between positions is the frame's own work -- records per tile, second rounds)
for pos in [float(v) for v in os.environ.get('POSITIONS', '').split(',') if v]:
    cam = common.frame_cameras(dict(settings, dblSteps=[pos]), oc)[0]
    n = 12
    group = [cam] * n
    launch = K.prepared_group_ahead(state, group, Bl, out[:n], group, stages=2)
    K.render_frame_group_ahead(state, group, Bl, out[:n], turn=0, placed=False, next_cameras=group, stages=2)
    turn = [1]

    def at_pos():
        launch(turn[0], True)
        turn[0] += 1
    t = sorted(timed(at_pos, n) for _ in range(3))[1]
    note = ''
    if stats is not None:
        K.lib.kbe_debug_frame_stats(stats, 1)
        at_pos()
        torch.cuda.synchronize()
        K.lib.kbe_debug_frame_stats(stats, 1)
        tl = max(1, stats[0])
        note = ' | per tile: candidates %.1f, in z reach %.0f, records %.0f; slow tiles %d, second round %d of %d' % (stats[2] / tl, stats[3] / tl, stats[4] / tl, stats[5], stats[6], stats[0])
    K.render_frame_group_ahead(state, group, Bl, out[:n], turn=turn[0], placed=True, next_cameras=None, stages=6)
    torch.cuda.synchronize()
    K.render_frame_group_fused(state, group, Bl, out[:n], stages=6)
    print('camera at %.3f of the path, twelve copies per launch: %.2f us per frame%s' % (pos, t, note))
