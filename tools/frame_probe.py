#!/usr/bin/env python3
"""Where a wave of the one-launch scatter (k_frame_group_ahead) spends its life: a variant library built with -DKBE_FRAME_PROBE
under /tmp stamps the shader clock (s_memtime) at the phase boundaries of every wave; this prints the phases' durations and the
launch's timeline (dev aid; the stamps cost a few per cent themselves).  FRAMES = frames per launch (default 8)."""
import ctypes
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
so = '/tmp/libkbe_probe.so'
src = os.path.join(ROOT, 'ken-burns-effect_amd', 'csrc')
subprocess.check_call(['make', '-s', '-B', '-C', src, 'EXTRA=-DKBE_FRAME_PROBE ' + os.environ.get('EXTRA', ''), 'OUT=' + so])
import torch  # noqa: E402

import bench  # noqa: E402
from ken_burns_effect_amd import _native, common, synthetic  # noqa: E402

_native._lib, _native._kernels, _native.LIB_PATH = None, None, so
size = int(os.environ.get('SIZE', '1024'))
n = int(os.environ.get('FRAMES', '8'))
ofrom, oto = synthetic.default_windows(size, size, False)
settings = {'dblSteps': [i / 63 for i in range(64)], 'objectFrom': ofrom, 'objectTo': oto, 'boolInpaint': True, 'dolly': False}
oc = bench.build_scene(size, torch.device('cuda:0'), True, settings, 1)
cams = common.frame_cameras(settings, oc)
K = _native.kernels()
state = common._prepared_cloud(K, oc)
K._pack(state)
Bl = oc['dblBaseline']
out = torch.empty(12, size, size, 3, dtype=torch.uint8, device='cuda')
K.group_scratch(state, 12)
group = [cams[len(cams) // 2]] * n
K.render_frame_group_fused(state, group, Bl, out[:n], stages=6)
launch = K.prepared_group_ahead(state, group, Bl, out[:n], group, stages=2)
for turn in range(6):
    launch(turn, turn > 0)
torch.cuda.synchronize()
STAMPS = 14
tiles = ((size + 31) // 32) * ((size + 15) // 16)
waves = tiles * 4 * n
buf = (ctypes.c_ulonglong * (waves * STAMPS))()
assert K.lib.kbe_debug_frame_probe(buf, ctypes.c_size_t(waves * STAMPS)) == 0
t = np.frombuffer(buf, dtype=np.uint64).reshape(waves, STAMPS).astype(np.int64)
ok = (t[:, 5] > 0) & (t[:, 7] > 0)
print('%d waves, %d on the normal path' % (waves, int(ok.sum())))
t = t[ok]
t0 = t[:, 0].min()
names = ['entry -> list count known, points requested, placements ahead begun', 'wait at barrier 1', 'splat', 'ahead_finish + wait at barrier 2',
         'decode z, barrier, degrid', 'wait at barrier 4', 'gather', 'epilogue (incl. its barrier, stores, hole list)', 'placements left over']
tot = (t[:, 9] - t[:, 0]).astype(np.float64)
print('a wave lives %.0f cycles on average (median %.0f, p90 %.0f); the launch spans %.0f cycles' % (tot.mean(), np.median(tot), np.percentile(tot, 90), float(t[:, 9].max() - t0)))
real = t[:, 13].astype(np.float64)          # ticks of the 100 MHz s_memrealtime over the wave's life
good = real > 0
print('shader clock while the waves ran (s_memtime ticks per 10 ns of s_memrealtime): mean %.3f GHz, median %.3f, p10 %.3f, p90 %.3f; a wave lives %.2f us' % (
    (tot[good] / (real[good] * 10.0)).mean(), np.median(tot[good] / (real[good] * 10.0)), np.percentile(tot[good] / (real[good] * 10.0), 10),
    np.percentile(tot[good] / (real[good] * 10.0), 90), (real[good] * 0.01).mean()))
for k, name in enumerate(names):
    d = (t[:, k + 1] - t[:, k]).astype(np.float64)
    print('  %5.1f %%  mean %6.0f  median %6.0f  p90 %6.0f   %s' % (100.0 * d.sum() / tot.sum(), d.mean(), np.median(d), np.percentile(d, 90), name))
sub = [('entry -> LDS initialised, list entries and ahead points requested', 0, 10), ('list entries arrive, the points requested', 10, 11),
       ('next camera loaded', 11, 12), ('two units placed ahead (begin)', 12, 1)]
for name, a, b in sub:
    d = (t[:, b] - t[:, a]).astype(np.float64)
    print('        of the first phase: mean %6.0f  median %6.0f  p90 %6.0f   %s' % (d.mean(), np.median(d), np.percentile(d, 90), name))
# the launch's timeline: waves resident (entered, not yet left) at 17 instants; the XCDs' clocks are not synchronised, so each XCD's waves are
# placed on the XCD's own time axis (block b runs on XCD b % 8)
idx = np.nonzero(ok)[0]
xcd = ((idx // 4) % tiles) % 8
resident = np.zeros(17)
spans = []
for x in range(8):
    tx = t[xcd == x]
    x0 = tx[:, 0].min()
    span = float(tx[:, 9].max() - x0)
    spans.append(span)
    resident += np.array([int(((tx[:, 0] - x0 <= f * span) & (tx[:, 9] - x0 > f * span)).sum()) for f in np.linspace(0.02, 0.98, 17)])
print('the launch spans %.0f cycles (per XCD: %s)' % (np.mean(spans), ' '.join('%.0f' % v for v in spans)))
print('waves resident over the launch (of %d slots at 5 per SIMD):' % (1024 * 5), ' '.join('%d' % v for v in resident))
print('wave slots occupied on average: %.2f per SIMD' % (tot.sum() / np.mean(spans) / 1024.0))
