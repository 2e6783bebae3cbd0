#!/usr/bin/env python3
"""Would the Disparity GridNet (6 rows x 4 columns; 256^2 input for a 512^2 image) gain from running its independent blocks on
parallel streams?  The lateral blocks of a column do not depend on each other; only the vertical chain does.  Serial forward
against a wavefront schedule (one stream per row + the main stream for the chains), eager and as a captured graph; results must be
bit-identical (same kernels, same operands).  Every intermediate is kept alive to the end, so that the caching allocator never
reuses a block another stream may still read (dev aid)."""
import os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch
from ken_burns_effect_amd import synthetic
from ken_burns_effect_amd.disparity_estimation import Disparity, Semantics, ROW_FEATURES
from ken_burns_effect_amd.pointcloud_inpainting import _match
from ken_burns_effect_amd.utils import resize_image

dev = torch.device('cuda:0')
sem = synthetic.seeded_fill_(Semantics(), 1).to(dev).eval()
dis = synthetic.seeded_fill_(Disparity(), 2).to(dev).eval()
rows = len(ROW_FEATURES)
side = [torch.cuda.Stream() for _ in range(rows)]


def forward_parallel(self, tensorImage, tensorSemantics):
    main = torch.cuda.current_stream()
    keep = []

    def on(stream, fn, *deps):
        for e in deps:
            stream.wait_event(e)
        with torch.cuda.stream(stream):
            out = fn()
            ev = torch.cuda.Event()
            ev.record(stream)
        keep.append(out)
        return out, ev

    level = [self.moduleImage(tensorImage)]
    for r in range(1, rows):
        nxt = self._run(r - 1, 0, r, 0, level[r - 1])
        if r == 3:
            nxt = nxt + self.moduleSemantics(tensorSemantics)
        level.append(nxt)
    keep.extend(level)
    ready = torch.cuda.Event(); ready.record(main)
    # column 1: laterals side by side, the chain of down-sampling blocks on the main stream
    lat = [on(side[r], lambda r=r: self._run(r, 0, r, 1, level[r]), ready) for r in range(rows)]
    new, evs = [], []
    for r in range(rows):
        main.wait_event(lat[r][1])
        v = lat[r][0] if r == 0 else lat[r][0] + self._run(r - 1, 1, r, 1, new[r - 1])
        new.append(v); keep.append(v)
        e = torch.cuda.Event(); e.record(main); evs.append(e)
    level = new
    for col in (2, 3):
        lat = [on(side[r], lambda r=r, col=col: self._run(r, col - 1, r, col, level[r]), evs[r]) for r in range(rows)]
        new, nevs = [None] * rows, [None] * rows
        for r in range(rows - 1, -1, -1):
            main.wait_event(lat[r][1])
            v = lat[r][0]
            if r != rows - 1:
                v = v + _match(self._run(r + 1, col, r, col, new[r + 1]), v)
            new[r] = v; keep.append(v)
            e = torch.cuda.Event(); e.record(main); nevs[r] = e
        level, evs = new, nevs
    out = self.moduleDisparity(level[0])
    self._keep = keep            # alive until the next call
    return out


def timed(run, n=20):
    for _ in range(3): run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): out = run()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, out


size = int(os.environ.get('SIZE', '512'))
image = torch.rand(1, 3, size, size, device=dev)
with torch.no_grad():
    resized = resize_image(image, max_size=size // 2)
    feat = sem(resized)
    serial_ms, ref = timed(lambda: dis(resized, feat))
    par_ms, out = timed(lambda: forward_parallel(dis, resized, feat))
    print('Disparity on a %d^2 image: serial %.2f ms, rows in parallel (eager) %.2f ms, bit-identical %s' % (size, serial_ms, par_ms, bool(torch.equal(ref, out))), flush=True)
    for name, fn in (('serial', lambda: dis(resized, feat)), ('parallel', lambda: forward_parallel(dis, resized, feat))):
        s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(3): fn()
        torch.cuda.current_stream().wait_stream(s)
        g = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(g):
                gout = fn()
            g.replay(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20): g.replay()
            torch.cuda.synchronize()
            print('  captured graph, %s schedule: %.2f ms per replay, bit-identical %s' % (name, (time.perf_counter() - t0) / 20 * 1e3, bool(torch.equal(gout, ref))), flush=True)
        except Exception as e:
            print('  capture of the %s schedule failed: %s' % (name, str(e).split(chr(10))[0][:200]), flush=True)
            torch.cuda.synchronize()
