#!/usr/bin/env python3
"""Times the frame launches for compile-time variants of the tile kernel (GPU box only; dev aid).

    python tools/tile_probe.py "-DKBE_TILE_H=16" "-DKBE_TILE_THREADS=256 -DKBE_PROBE_SKIP_B" ...
Each argument is a set of extra hipcc flags; the library is rebuilt into /tmp per variant.
"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

CSRC = os.path.join(ROOT, 'ken-burns-effect_amd', 'csrc')


def main():
    from ken_burns_effect_amd import _native, common, synthetic
    import bench
    size = int(os.environ.get('SIZE', '1024'))
    variants = sys.argv[1:] or ['']
    for i, flags in enumerate(variants):
        so = '/tmp/libkbe_probe_%d.so' % i
        cmd = ['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-fPIC', '-shared', '-fvisibility=hidden',
               '-I' + os.path.join(ROOT, 'include'), '-I' + CSRC] + flags.split() + \
              [os.path.join(CSRC, 'kbe_hip.hip'), os.path.join(CSRC, 'kbe_frame.hip'), '-o', so]
        subprocess.check_call(cmd)
        _native._lib, _native._kernels, _native.LIB_PATH = None, None, so
        dev = torch.device('cuda:0')
        oc = bench.build_scene(size, dev, os.environ.get('CLOUD', 'inpaint') == 'inpaint', dict(dolly=False, objectFrom=synthetic.default_windows(size, size, False)[0], objectTo=synthetic.default_windows(size, size, False)[1]))
        ofrom, oto = synthetic.default_windows(size, size, False)
        settings = {'dblSteps': [0.0, 0.25, 0.5, 0.75, 1.0], 'objectFrom': ofrom, 'objectTo': oto, 'boolInpaint': False, 'dolly': False}
        cams = common.frame_cameras(settings, oc)
        res = {}
        for ci in (0, 2, 4):
            cw, ch = common.crop_size(settings)
            kt = bench.time_kernels(oc, [cams[ci]], reps=int(os.environ.get('REPS', '60')), fill_rect=common.crop_window(size, size, cw, ch))
            res[ci] = kt
        if 'KBE_PROBE_TIMING' in flags:
            K = _native.kernels()
            state = K.prepare_cloud(oc['tensorInpaPoints'], oc['tensorInpaImage'], oc['tensorInpaDepth'], size, size)
            dbg = torch.zeros(4, size, size, device=dev)
            focal, shift3 = cams[2]
            K.render_frame(state, shift3, focal, oc['dblBaseline'], render_f32=dbg, stages=3)
            torch.cuda.synchronize()
            t = dbg.view(torch.int64).reshape(-1)[:(size // 32) ** 2 * 16].reshape(-1, 16).cpu().numpy().astype('float64')
            d = t[:, 1:11] - t[:, 0:10]
            names = ['loads->lds', 'barrier1', 'degrid', 'insert', 'barrier2', 'gather', 'barrier3', 'resolve', 'barrier4', 'store']
            print('   phases (cycles, mean over tiles): ' + ' '.join('%s=%.0f' % (n, v) for n, v in zip(names, d.mean(0))), ' total=%.0f' % (t[:, 10] - t[:, 0]).mean(), flush=True)
        print('variant %-50s' % (flags or '(default)'), ' | '.join(
            'step%d reset %.1f proj %.1f tiles %.1f fill %.1f frame %.1f' % (ci, r['reset'] * 1e6, (r['project+reset'] - r['reset']) * 1e6, r['tiles'] * 1e6, r['fill'] * 1e6, r['frame'] * 1e6) for ci, r in res.items()),
            flush=True)


if __name__ == '__main__':
    main()
