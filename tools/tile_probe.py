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
import numpy as np  # noqa: E402
import torch  # noqa: E402

CSRC = os.path.join(ROOT, 'ken-burns-effect_amd', 'csrc')


def main():
    from ken_burns_effect_amd import _native, common, synthetic
    import bench
    size = int(os.environ.get('SIZE', '1024'))
    variants = sys.argv[1:] or ['']
    for i, flags in enumerate(variants):
        so = '/tmp/libkbe_probe_%d.so' % i
        cmd = ['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-fno-slp-vectorize', '-fPIC', '-shared', '-fvisibility=hidden',
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
            span = t[:, 10].max() - t[:, 0].min()
            starts = np.sort(t[:, 0] - t[:, 0].min())
            print('   span=%.0f cycles; WG start quartiles %s; WG duration min/med/max %.0f %.0f %.0f' % (
                span, np.percentile(starts, [0, 25, 50, 75, 100]).round().tolist(),
                (t[:, 10] - t[:, 0]).min(), np.median(t[:, 10] - t[:, 0]), (t[:, 10] - t[:, 0]).max()), flush=True)
            nt = t.shape[0]
            xcd = np.arange(nt) & 7
            spans = [t[xcd == k, 10].max() - t[xcd == k, 0].min() for k in range(8)]
            print('   per-XCD span (cycles):', [int(v) for v in spans], flush=True)
            cnt = state['scratch'].view(torch.int32)[size * size:size * size + nt * 32:32].cpu().numpy()
            b = np.arange(nt); q, r = nt >> 3, nt & 7
            tile_of = np.where((b & 7) < r, (b & 7) * (q + 1), r * (q + 1) + ((b & 7) - r) * q) + (b >> 3)
            c = cnt[tile_of]; dur = t[:, 10] - t[:, 0]
            for lo, hi in ((0, 600), (600, 900), (900, 1100), (1100, 1300), (1300, 1536), (1536, 3072), (3072, 1 << 30)):
                m = (c >= lo) & (c < hi)
                if m.any():
                    print('   count [%d,%d): %d tiles, duration mean %.0f max %.0f; gather %.0f insert %.0f degrid %.0f' % (
                        lo, hi, m.sum(), dur[m].mean(), dur[m].max(), d[m, 5].mean(), d[m, 3].mean(), d[m, 2].mean()), flush=True)
            w0 = t[:, 11] - t[:, 11].min(); w1 = t[:, 12] - t[:, 11].min()
            print('   wall clock (10 ns ticks): kernel span %d; WG start percentiles %s; end percentiles %s' % (
                w1.max(), np.percentile(w0, [0, 10, 25, 50, 75, 90, 100]).astype(int).tolist(),
                np.percentile(w1, [0, 10, 25, 50, 75, 90, 100]).astype(int).tolist()), flush=True)
            print('   sorted WG starts (every 32nd):', np.sort(w0).astype(int)[::32].tolist(), flush=True)
            print('   sorted WG ends   (every 32nd):', np.sort(w1).astype(int)[::32].tolist(), flush=True)
            print('   XCD0 (start,end) in block order:', [(int(w0[i]), int(w1[i])) for i in range(0, nt, 8)][:40], flush=True)
            late = np.argsort(-w1)[:12]
            print('   last to finish: ' + ' '.join('b%d(xcd%d,cnt%d,start%d,end%d)' % (i, i & 7, c[i], w0[i], w1[i]) for i in late), flush=True)
            k0 = xcd == 0
            st = t[k0, 0] - t[k0, 0].min()
            print('   XCD0 starts:', np.sort(st).astype(int)[::8].tolist(), flush=True)
            print('   XCD0 ends  :', np.sort(t[k0, 10] - t[k0, 0].min()).astype(int)[::8].tolist(), flush=True)
            print('   phases (cycles, mean over tiles): ' + ' '.join('%s=%.0f' % (n, v) for n, v in zip(names, d.mean(0))), ' total=%.0f' % (t[:, 10] - t[:, 0]).mean(), flush=True)
        print('variant %-50s' % (flags or '(default)'), ' | '.join(
            'step%d reset %.1f proj %.1f tiles %.1f fill %.1f frame %.1f' % (ci, r['reset'] * 1e6, (r['project+reset'] - r['reset']) * 1e6, r['tiles'] * 1e6, r['fill'] * 1e6, r['frame'] * 1e6) for ci, r in res.items()),
            flush=True)


if __name__ == '__main__':
    main()
