#!/usr/bin/env python3
"""Golden-vector generator -- runs ONLY in the build container.

Produces ``tests/golden/*.npz`` by executing the reference implementation that is
mounted read-only at ``/root/reference`` on seeded synthetic inputs.  Nothing of
the reference is copied: the fixtures hold inputs and expected outputs only, and
this script (our own code) documents exactly how they were made.

How the reference is executed here:

* torch-level functions (``depth_to_points``, ``spatial_filter``, ``process_shift``,
  ``process_inpaint``, ``process_kenburns``, ``Inpaint``, ``PartialConv2d``) are
  imported from ``/root/reference`` as they are.  ``cupy``, ``cv2`` and
  ``torchvision`` are absent from this image and are only needed for the import
  statements at the top of ``utils/common.py`` / ``models/*.py``, so empty module
  objects stand in for them; ``Tensor.cuda`` is the identity (no GPU here).
* the four CUDA kernels live as source strings inside ``render_pointcloud`` /
  ``fill_disocclusion``.  The reference's own ``preprocess_kernel`` expands them;
  the resulting text is compiled UNMODIFIED with g++ against a small header that
  declares the CUDA built-ins it uses (``blockIdx`` & co, ``float3``, serial
  ``atomicCAS`` / ``atomicAdd``), and a generated driver calls the kernel once
  per element in index order.  That pins the kernels' arithmetic with the
  *serial point-index schedule* (the GPU schedule of the original is not
  deterministic: in-place degrid, atomicAdd order -- SURVEY.md Appendix B.3/B.4).
  Two builds: ``-ffp-contract=off`` ("nofma") and ``-mfma -ffp-contract=fast``
  ("fma", what NVRTC's default --fmad=true does to ``x + dist * (-x)``); the
  shim's ``atomicAdd`` is noinline so that ``old + data * w`` is never fused,
  exactly like a hardware atomic.

Because the kernel text needs NVRTC/CuPy/a CUDA GPU to run natively, this
host-executed form is the strongest pin available; DESIGN.md says so.

Usage:  python tests/golden/make_golden.py          (writes next to this file)
"""
import sys

sys.dont_write_bytecode = True  # never write __pycache__ into the read-only tree

import ctypes
import hashlib
import os
import re
import subprocess
import types

import numpy as np
import torch

REF = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
WORK = os.environ.get('KBE_GOLDEN_WORK', '/tmp/kbe_golden_work')

sys.path.insert(0, ROOT)
from ken_burns_effect_amd import synthetic  # noqa: E402

# --------------------------------------------------------------------------------------
# import-time stand-ins (only so that `import utils.common` succeeds; none is called
# on the paths exercised below except the two cv2 no-ops noted in trace_kenburns)
# --------------------------------------------------------------------------------------

SHIM = r'''
#pragma once
#define __CUDACC__ 1
#include <math.h>
#include <assert.h>
#include <string.h>
#define __host__
#define __device__
#define __global__
#define __forceinline__ inline
#define KBE_VEC(T, N2, N3, N4) \
  struct N2 { T x, y; }; struct N3 { T x, y, z; }; struct N4 { T x, y, z, w; }; \
  static inline N2 make_##N2(T x, T y) { N2 v = {x, y}; return v; } \
  static inline N3 make_##N3(T x, T y, T z) { N3 v = {x, y, z}; return v; } \
  static inline N4 make_##N4(T x, T y, T z, T w) { N4 v = {x, y, z, w}; return v; }
KBE_VEC(float, float2, float3, float4)
KBE_VEC(int, int2, int3, int4)
KBE_VEC(unsigned int, uint2, uint3, uint4)
struct kbe_dim3 { unsigned int x, y, z; };
static kbe_dim3 blockIdx, threadIdx, blockDim, gridDim;
static inline int min(int a, int b) { return a < b ? a : b; }
static inline int max(int a, int b) { return a > b ? a : b; }
static inline unsigned int min(unsigned int a, unsigned int b) { return a < b ? a : b; }
static inline unsigned int max(unsigned int a, unsigned int b) { return a > b ? a : b; }
static inline float min(float a, float b) { return fminf(a, b); }
static inline float max(float a, float b) { return fmaxf(a, b); }
static inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
static inline int __float_as_int(float f) { int i; memcpy(&i, &f, 4); return i; }
static inline float __int_as_float(int i) { float f; memcpy(&f, &i, 4); return f; }
static inline int atomicCAS(int* p, int cmp, int val) { int old = *p; if (old == cmp) *p = val; return old; }
__attribute__((noinline)) static float atomicAdd(float* p, float v) { float old = *p; *p = old + v; return old; }
static inline int atomicExch(int* p, int v) { int old = *p; *p = v; return old; }
static inline float atomicExch(float* p, float v) { float old = *p; *p = v; return old; }
'''

FLAGS = {
    'nofma': ['-O2', '-ffp-contract=off'],
    'fma': ['-O2', '-mfma', '-ffp-contract=fast'],
}


class Harness:
    """Loads the reference under stand-ins and routes its kernel launches to g++ builds."""

    def __init__(self):
        os.makedirs(os.path.join(WORK, 'shim'), exist_ok=True)
        with open(os.path.join(WORK, 'shim', 'cuda_runtime.h'), 'w') as f:
            f.write(SHIM)
        self.mode = 'fma'
        self.trace = []        # (kernel name, {tensor name: snapshot}) per launch
        self.tracing = False
        self._vars = None
        self._cache = {}

        cupy = types.ModuleType('cupy')
        cupy.util = types.SimpleNamespace(memoize=lambda **kw: (lambda fn: fn))
        cupy.cuda = types.SimpleNamespace(compile_with_cache=None)
        cv2 = types.ModuleType('cv2')
        cv2.INTER_LINEAR = 1
        # identity stand-ins: process_kenburns' trace then returns the PRE-crop uint8 frames
        cv2.getRectSubPix = lambda image, patchSize, center: image
        cv2.resize = lambda src, dsize, fx=0.0, fy=0.0, interpolation=1: src
        tv = types.ModuleType('torchvision')
        tv.models = types.ModuleType('torchvision.models')
        sys.modules.update({'cupy': cupy, 'cv2': cv2, 'torchvision': tv, 'torchvision.models': tv.models})
        torch.cuda.current_stream = lambda *a, **k: types.SimpleNamespace(cuda_stream=0)
        torch.Tensor.cuda = lambda self, *a, **k: self
        sys.path.insert(0, REF)
        import utils.common as C
        import models.pointcloud_inpainting as PI
        import models.partial_inpainting as PPI
        import utils.partial_conv as PC
        self.C, self.PI, self.PPI, self.PC = C, PI, PPI, PC
        C.path_to_math_helper = os.path.join(REF, 'utils', 'helper_math.h')
        orig_pre = C.preprocess_kernel

        def pre(src, variables):
            self._vars = variables
            return orig_pre(src, variables)

        C.preprocess_kernel = pre
        C.launch_kernel = self._launch

    # -- kernel text -> shared object ---------------------------------------------------
    def _build(self, name, src):
        key = hashlib.sha1((self.mode + name + src).encode()).hexdigest()[:20]
        if key in self._cache:
            return self._cache[key]
        sig = re.search(r'void\s+' + name + r'\s*\((.*?)\)\s*\{', src, re.S).group(1)
        params = [p.strip() for p in sig.split(',')]
        names = [p.split()[-1].lstrip('*') for p in params]
        driver = ('\nextern "C" void run_%s(%s) {\n  blockDim.x = 1; gridDim.x = (unsigned) n;\n'
                  '  for (int i = 0; i < n; i++) { blockIdx.x = (unsigned) i; threadIdx.x = 0; %s(%s); }\n}\n'
                  % (name, ', '.join(params), name, ', '.join(names)))
        cpp = os.path.join(WORK, key + '.cpp')
        so = os.path.join(WORK, key + '.so')
        with open(cpp, 'w') as f:
            f.write(src + driver)
        subprocess.check_call(['g++', '-shared', '-fPIC', '-w', '-I' + os.path.join(WORK, 'shim')]
                              + FLAGS[self.mode] + [cpp, '-o', so])
        fn = getattr(ctypes.CDLL(so), 'run_' + name)
        fn.restype = None
        self._cache[key] = fn
        return fn

    def _launch(self, name, src):
        fn = self._build(name, src)
        variables = self._vars

        def call(grid, block, args, stream):
            fn(ctypes.c_int(args[0]), *[ctypes.c_void_p(a) for a in args[1:]])
            if self.tracing:
                self.trace.append((name, {k: v.detach().clone() for k, v in variables.items()
                                          if torch.is_tensor(v) and k in ('zee', 'output', 'masks', 'id_memory')}))
        return call

    def traced(self, fn, *args):
        self.trace, self.tracing = [], True
        try:
            out = fn(*args)
        finally:
            self.tracing = False
        return out, self.trace


def npy(t):
    return t.detach().cpu().numpy()


def save(name, **arrays):
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **arrays)
    print('%-28s %7.1f KB  %s' % (name + '.npz', os.path.getsize(path) / 1024.0, sorted(arrays)))


# --------------------------------------------------------------------------------------
# scenes
# --------------------------------------------------------------------------------------

def scene_points(H, W, seed, kind, n_extra, focal=synthetic.FOCAL, baseline=synthetic.BASELINE, shift=None):
    """Point cloud of a synthetic RGBD image plus `n_extra` adversarial points."""
    h = HARNESS
    image, disp = synthetic.make_rgbd(H, W, seed, kind, baseline)
    depth = (focal * baseline) / (disp + 0.0000001)
    pts = h.C.depth_to_points(depth, focal).view(1, 3, -1)
    data = torch.cat([image.view(1, 3, -1), depth.view(1, 1, -1)], 1)
    rng = np.random.default_rng(seed + 1000)
    if shift is not None:   # camera motion applies to the scene; the adversarial extras below stay as given
        pts = pts + torch.tensor(shift, dtype=torch.float32).view(1, 3, 1)
    if n_extra:
        zmin, zmax = float(depth.min()), float(depth.max())
        ex = np.zeros((1, 3, n_extra), np.float32)
        ex[0, 2] = rng.uniform(zmin * 0.5, zmax * 1.2, n_extra)
        ex[0, 0] = rng.uniform(-0.7 * W, 0.7 * W, n_extra) * ex[0, 2] / focal   # some out of view
        ex[0, 1] = rng.uniform(-0.7 * H, 0.7 * H, n_extra) * ex[0, 2] / focal
        # culled / degenerate depths: <0.001, 0, negative, and the [0.001, F*B/1e6) band whose
        # dblError is NEGATIVE
        special = [0.0, -5.0, 0.0005, 0.001, 0.0011, 0.01, 0.05, 0.0614, 0.07, 1.0]
        for i, z in enumerate(special[:n_extra]):
            ex[0, 2, i] = z
            ex[0, 0, i] = (i - 4.5) * z / focal * 3.0
            ex[0, 1, i] = (4.5 - i) * z / focal * 2.0
        # exact pixel-centre and exact half-way projections (winner tie-breaks NW,NE,SW,SE)
        for j, (u, v) in enumerate([(0.0, 0.0), (0.5, 0.0), (0.0, 0.5), (0.5, 0.5), (-0.5, -0.5)]):
            k = len(special) + j
            if k < n_extra:
                z = np.float32(640.0)
                ex[0, 2, k] = z
                ex[0, 0, k] = np.float32(u) * z / np.float32(focal)
                ex[0, 1, k] = np.float32(v) * z / np.float32(focal)
        exd = rng.random((1, 4, n_extra), dtype=np.float32)
        pts = torch.cat([pts, torch.from_numpy(ex)], 2)
        data = torch.cat([data, torch.from_numpy(exd)], 2)
    return image, disp, depth, pts.contiguous(), data.contiguous()


def winners_of(points, W, H, focal, baseline):
    """Per-point winner pixel (linear index, -1 = none) by splatting points one at a time."""
    h = HARNESS
    N = points.size(2)
    out = np.full((points.size(0), N), -1, np.int32)
    dummy = torch.zeros(1, 1, 1)
    for b in range(points.size(0)):
        for i in range(N):
            _, tr = h.traced(h.C.render_pointcloud, points[b:b + 1, :, i:i + 1].contiguous(), dummy, W, H, focal, baseline)
            zee = npy(tr[0][1]['zee']).reshape(-1)
            hit = np.nonzero(zee != np.float32(1000000.0))[0]
            assert hit.size <= 1
            if hit.size:
                out[b, i] = hit[0]
    return out


def render_case(name, H, W, seed, kind, n_extra, focal, baseline, shift, with_winners, batch=1, C=None):
    h = HARNESS
    pts_l, data_l = [], []
    for b in range(batch):
        _, _, _, pts, data = scene_points(H, W, seed + 17 * b, kind, n_extra,
                                          shift=tuple(s * (1.0 + 0.5 * b) for s in shift))
        if C is not None:   # arbitrary channel count: tile / crop the 4 channels and perturb
            reps = (C + 3) // 4
            data = torch.cat([data * (1.0 + 0.125 * r) for r in range(reps)], 1)[:, :C].contiguous()
        pts_l.append(pts)
        data_l.append(data)
    pts, data = torch.cat(pts_l, 0).contiguous(), torch.cat(data_l, 0).contiguous()
    arrays = dict(points=npy(pts), data=npy(data), W=np.int32(W), H=np.int32(H),
                  focal=np.float64(focal), baseline=np.float64(baseline),
                  baseline_is_int=np.bool_(isinstance(baseline, int)))
    for mode in ('fma', 'nofma'):
        h.mode = mode
        (render, existing), tr = h.traced(h.C.render_pointcloud, pts, data, W, H, focal, baseline)
        assert [t[0] for t in tr] == ['kernel_pointrender_updateZee', 'kernel_pointrender_updateDegrid',
                                      'kernel_pointrender_updateOutput']
        arrays['zee_pre_' + mode] = npy(tr[0][1]['zee'])
        arrays['zee_serial_' + mode] = npy(tr[1][1]['zee'])
        arrays['acc_' + mode] = npy(tr[2][1]['output'])
        if mode == 'fma':   # render/existing are acc[:C] / (acc[C] + 1e-7) and acc[C]: kept once
            arrays['render_fma'] = npy(render)
            arrays['existing_fma'] = npy(existing)
        if with_winners:
            arrays['winner_' + mode] = winners_of(pts, W, H, focal, baseline)
    h.mode = 'fma'
    save(name, **arrays)
    return arrays


# --------------------------------------------------------------------------------------
# fixture groups
# --------------------------------------------------------------------------------------

def gen_render():
    # the three focal literals of SURVEY B.1 (512.0, 409.6 and a dolly value with a long
    # decimal expansion), int vs float baseline literal, batch > 1, odd channel counts
    render_case('render_f512', 48, 64, 1, 'smooth', 40, 512.0, 120, (-9.0, 4.0, -25.0), True)
    render_case('render_f409', 48, 64, 2, 'smooth', 40, 409.6, 120, (6.0, -3.0, -40.0), True)
    render_case('render_f153', 40, 56, 3, 'smooth', 24, 153.60000000000002, 40.0, (2.0, 1.0, 10.0), False)
    render_case('render_noise', 48, 48, 4, 'noise', 0, 512.0, 120, (3.0, 2.0, -15.0), False)
    render_case('render_b2c7', 24, 40, 5, 'smooth', 16, 512.0, 120, (-4.0, 2.0, -20.0), False, batch=2, C=7)


def gen_fill():
    h = HARNESS
    arrays = {}
    for tag, (H, W, seed, shift, B) in {'a': (48, 64, 11, (-14.0, 6.0, -60.0), 1),
                                        'b': (40, 72, 12, (10.0, -8.0, -45.0), 2)}.items():
        rs, ds = [], []
        for b in range(B):
            _, _, _, pts, data = scene_points(H, W, seed + b, 'smooth', 0)
            pts = pts + torch.tensor(shift, dtype=torch.float32).view(1, 3, 1)
            render, existing = h.C.render_pointcloud(pts, data, W, H, 512.0, 120)
            depth = render[:, 3:4] * (existing > 0.0).float()
            rng = np.random.default_rng(seed + 50 + b)
            # extra holes: rectangles inside, one touching the border (rays that leave the image),
            # a one-pixel-wide slit and isolated pixels
            depth[0, 0, 5:14, 20:31] = 0.0
            depth[0, 0, H - 6:H, 0:9] = 0.0
            depth[0, 0, 0:H, W // 2] = 0.0
            for _ in range(30):
                depth[0, 0, int(rng.integers(0, H)), int(rng.integers(0, W))] = 0.0
            depth[0, 0, 2, 3] = -1.0                       # negative depth counts as a hole too
            rs.append(render)
            ds.append(depth)
        render, depth = torch.cat(rs, 0).contiguous(), torch.cat(ds, 0).contiguous()
        if tag == 'b':
            render = render[:, 0:3].contiguous()           # C = 3
        arrays['input_' + tag] = npy(render)
        arrays['depth_' + tag] = npy(depth)
        outs = {}
        for mode in ('fma', 'nofma'):
            h.mode = mode
            outs[mode] = npy(h.C.fill_disocclusion(render, depth))
        assert np.array_equal(outs['fma'], outs['nofma'])   # fill has no contractible expression
        arrays['output_' + tag] = outs['fma']
        h.mode = 'fma'
    # an image that is ALL holes and one with no holes
    z = torch.zeros(1, 1, 6, 7)
    x = torch.arange(2 * 42, dtype=torch.float32).view(1, 2, 6, 7)
    arrays['input_allholes'], arrays['depth_allholes'] = npy(x), npy(z)
    arrays['output_allholes'] = npy(h.C.fill_disocclusion(x, z))
    arrays['output_noholes'] = npy(h.C.fill_disocclusion(x, z + 1.0))
    save('fill', **arrays)


def gen_torch_helpers():
    h = HARNESS
    rng = np.random.default_rng(21)
    arrays = {}
    for tag, (B, H, W, F) in {'a': (2, 5, 7, 512.0), 'b': (1, 12, 16, 409.6), 'c': (1, 6, 9, 153.60000000000002)}.items():
        d = torch.from_numpy(rng.uniform(0.0, 900.0, (B, 1, H, W)).astype(np.float32))
        d[0, 0, 0, 0] = 0.0
        arrays['d2p_depth_' + tag] = npy(d)
        arrays['d2p_focal_' + tag] = np.float64(F)
        arrays['d2p_points_' + tag] = npy(h.C.depth_to_points(d, F))
    for tag, (B, Cn, H, W) in {'a': (2, 1, 9, 11), 'b': (1, 3, 8, 8)}.items():
        x = torch.from_numpy(rng.normal(0, 1, (B, Cn, H, W)).astype(np.float32))
        arrays['sf_input_' + tag] = npy(x)
        for kind in ('laplacian', 'median-3', 'median-5'):
            arrays['sf_%s_%s' % (kind, tag)] = npy(h.C.spatial_filter(x, kind))
    m = torch.from_numpy((rng.random((1, 1, 17, 19)) > 0.35).astype(np.float32))
    arrays['sf_input_mask'] = npy(m)
    arrays['sf_median-5_mask'] = npy(h.C.spatial_filter(m, 'median-5'))
    disp = synthetic.make_rgbd(20, 28, 23, 'smooth')[1]
    arrays['sf_input_disp'] = npy(disp)
    arrays['sf_valid_disp'] = npy((h.C.spatial_filter(disp / disp.max(), 'laplacian').abs() < 0.03).float())
    # process_shift
    _, _, depth, pts, _ = scene_points(24, 32, 24, 'smooth', 6)
    common = {'dblFocal': 512.0, 'intWidth': 32, 'intHeight': 24, 'objectDepthrange': synthetic.depthrange_of(depth)}
    arrays['ps_points'] = npy(pts)
    arrays['ps_depthrange'] = np.array([common['objectDepthrange'][0], common['objectDepthrange'][1],
                                        common['objectDepthrange'][2][0], common['objectDepthrange'][2][1],
                                        common['objectDepthrange'][3][0], common['objectDepthrange'][3][1]], np.float64)
    settings = [(-1.7, 0.9, 0.93, None), (2.25, -3.5, 0.85, 409.6), (0.0, 0.0, 1.0, None)]
    arrays['ps_settings'] = np.array([[s[0], s[1], s[2], -1.0 if s[3] is None else s[3]] for s in settings], np.float64)
    for i, (su, sv, ratio, focal) in enumerate(settings):
        dfrom = common['objectDepthrange'][0]
        st = {'tensorPoints': pts, 'dblShiftU': su, 'dblShiftV': sv, 'dblDepthFrom': dfrom, 'dblDepthTo': dfrom * ratio}
        out, shift = h.C.process_shift(st, common, focal) if focal is not None else h.C.process_shift(st, common)
        arrays['ps_out_%d' % i] = npy(out)
        arrays['ps_shift_%d' % i] = npy(shift)
    save('torch_helpers', **arrays)


def gen_partial_conv():
    h = HARNESS
    rng = np.random.default_rng(31)
    arrays = {}
    for tag, (cin, cout, k, s, p) in {'a': (5, 4, 3, 1, 1), 'b': (3, 6, 3, 2, 1), 'c': (4, 2, 1, 1, 0)}.items():
        conv = h.PC.PartialConv2d(cin, cout, kernel_size=k, stride=s, padding=p, bias=True, multi_channel=True, return_mask=True)
        synthetic.seeded_fill_(conv, 7)
        x = torch.from_numpy(rng.normal(0, 1, (2, cin, 10, 12)).astype(np.float32))
        m = torch.from_numpy((rng.random((2, cin, 10, 12)) > 0.5).astype(np.float32))
        m[:, :, 0:4, 0:5] = 0.0      # a window with no valid input at all (update_mask == 0)
        with torch.no_grad():
            out, um = conv(x, m)
        arrays.update({'x_' + tag: npy(x), 'm_' + tag: npy(m), 'w_' + tag: npy(conv.weight), 'b_' + tag: npy(conv.bias),
                       'cfg_' + tag: np.array([cin, cout, k, s, p], np.int32), 'out_' + tag: npy(out), 'mask_' + tag: npy(um)})
    save('partial_conv', **arrays)


def gen_inpaint():
    h = HARNESS
    torch.manual_seed(0)
    net = h.PI.Inpaint().eval()
    synthetic.seeded_fill_(net, 3)
    H, W = 40, 48
    image, disp = synthetic.make_rgbd(H, W, 41, 'smooth')
    rng = np.random.default_rng(42)
    arrays = dict(image=npy(image), disparity=npy(disp), n_state=np.int32(len(net.state_dict())),
                  n_params=np.int64(sum(p.numel() for p in net.parameters())),
                  state_names=np.array(sorted(net.state_dict().keys())))
    with torch.no_grad():
        # (1) forward on explicit 68-channel data + mask
        data = torch.from_numpy(rng.normal(0, 1, (1, 68, H, W)).astype(np.float32))
        mask = torch.from_numpy((rng.random((1, 1, H, W)) > 0.2).astype(np.float32))
        net.normalize_images_disp(image, disp, not_normed=True)   # sets the (de)normalisation state
        out = net(tensorData=data * mask, tensorMasks=mask)
        arrays.update(fw_data=npy(data * mask), fw_mask=npy(mask), fw_image=npy(out['tensorImage']),
                      fw_disparity=npy(out['tensorDisparity']))
        # (2) forward from image + disparity (context extractor inside)
        out = net(tensorMasks=mask, tensorImage=image.clone(), tensorDisparity=disp.clone())
        arrays.update(fi_image=npy(out['tensorImage']), fi_disparity=npy(out['tensorDisparity']))
        # (3) pointcloud_inpainting end to end (render of 68 channels + median-5 mask + forward)
        depth = (512.0 * 120) / (disp + 0.0000001)
        common = {'dblFocal': 512.0, 'dblBaseline': 120, 'intWidth': W, 'intHeight': H}
        shift = torch.tensor([-6.0, 2.5, -35.0]).view(1, 3, 1)
        (out), tr = h.traced(net.pointcloud_inpainting, image, disp, shift, common)
        arrays.update(pi_shift=npy(shift), pi_existing=npy(out['tensorExisting']), pi_image=npy(out['tensorImage']),
                      pi_disparity=npy(out['tensorDisparity']), pi_zee=npy(tr[1][1]['zee']))
    save('inpaint', **arrays)

    # partial-conv variant: forward only (unreachable from kbe.py, IndexError in process_inpaint)
    import contextlib
    import io
    pnet = h.PPI.Inpaint().eval()
    synthetic.seeded_fill_(pnet, 5)
    arrays = dict(n_state=np.int32(len(pnet.state_dict())), n_params=np.int64(sum(p.numel() for p in pnet.parameters())),
                  state_names=np.array(sorted(pnet.state_dict().keys())))
    with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
        data = torch.from_numpy(rng.normal(0, 1, (1, 68, 32, 40)).astype(np.float32))
        mask = torch.from_numpy((rng.random((1, 1, 32, 40)) > 0.3).astype(np.float32))
        image, disp = synthetic.make_rgbd(32, 40, 43, 'smooth')
        pnet.normalize_images_disp(image, disp, not_normed=True)
        try:
            out = pnet(tensorData=data * mask, tensorMasks=mask)
            arrays.update(fw_data=npy(data * mask), fw_mask=npy(mask), fw_image=npy(out['tensorImage']),
                          fw_disparity=npy(out['tensorDisparity']), fw_existing_shape=np.array(out['tensorExisting'].shape))
        except Exception as e:   # recorded, not hidden
            arrays['fw_error'] = np.array(repr(e))
    save('partial_inpaint', **arrays)


def gen_disparity():
    """Disparity (6x4 GridNet) and the two Refine variants with name-seeded weights.  Semantics wraps
    torchvision's VGG19-bn (absent here), so Disparity is fed a random semantics tensor."""
    import models.disparity_estimation as DE
    import models.disparity_refinement as DR
    import models.disparity_refinement_pretrained as DRP
    rng = np.random.default_rng(71)
    arrays = {}
    image = torch.from_numpy(rng.random((1, 3, 64, 96), dtype=np.float32))
    sem = torch.from_numpy(rng.normal(0, 1, (1, 512, 4, 6)).astype(np.float32))
    net = synthetic.seeded_fill_(DE.Disparity().eval(), 11)
    arrays.update(image=npy(image), semantics=npy(sem), disp_out=npy(net(image, sem)),
                  disp_names=np.array(sorted(net.state_dict().keys())), disp_params=np.int64(sum(p.numel() for p in net.parameters())))
    coarse = torch.from_numpy((rng.random((1, 1, 16, 24), dtype=np.float32) * 50 + 5).astype(np.float32))
    arrays['coarse'] = npy(coarse)
    for tag, mod in (('refine', DR), ('refinep', DRP)):
        r = synthetic.seeded_fill_(mod.Refine().eval(), 13)
        arrays[tag + '_out'] = npy(r(image, coarse))
        arrays[tag + '_names'] = np.array(sorted(r.state_dict().keys()))
        arrays[tag + '_params'] = np.int64(sum(p.numel() for p in r.parameters()))
    save('disparity', **arrays)


class RecordedInpaint:
    """Wraps the reference Inpaint and records what pointcloud_inpainting returned."""

    def __init__(self, net):
        self.net, self.calls = net, []

    def pointcloud_inpainting(self, *a, **k):
        out = self.net.pointcloud_inpainting(*a, **k)
        self.calls.append({kk: vv.detach().clone() for kk, vv in out.items()})
        return out


def gen_kenburns():
    h = HARNESS
    net = h.PI.Inpaint().eval()
    synthetic.seeded_fill_(net, 3)
    for tag, (H, W, seed, dolly, steps) in {'kbe': (48, 64, 51, False, [0.0, 0.3, 0.7, 1.0]),
                                            'dolly': (40, 56, 52, True, [0.0, 0.5, 1.0])}.items():
        image, disp = synthetic.make_rgbd(H, W, seed, 'smooth')
        depth = (512.0 * 120) / (disp + 1e-7)
        pts = h.C.depth_to_points(depth, 512.0)
        common = {'dblFocal': 512.0, 'dblBaseline': 120, 'intWidth': W, 'intHeight': H,
                  'dblDispmin': disp.min().item(), 'dblDispmax': disp.max().item(),
                  'objectDepthrange': synthetic.depthrange_of(depth),
                  'tensorRawPoints': pts.view(1, 3, -1), 'tensorRawImage': image,
                  'tensorRawDisparity': disp, 'tensorRawDepth': depth}
        ofrom, oto = synthetic.default_windows(H, W, dolly)
        settings = {'dblSteps': steps, 'objectFrom': ofrom, 'objectTo': oto, 'boolInpaint': True, 'dolly': dolly}
        rec = RecordedInpaint(net)
        dr = common['objectDepthrange']
        arrays = dict(image=npy(image), disparity=npy(disp), steps=np.array(steps, np.float64), dolly=np.bool_(dolly),
                      depthrange=np.array([dr[0], dr[1], dr[2][0], dr[2][1], dr[3][0], dr[3][1]], np.float64))
        with torch.no_grad():
            frames, tr = h.traced(h.C.process_kenburns, settings, common, rec)
        arrays['frames'] = np.stack(frames)            # pre-crop uint8 (cv2 stand-ins are identities)
        arrays['inpa_points'] = npy(common['tensorInpaPoints'])
        arrays['inpa_image'] = npy(common['tensorInpaImage'])
        arrays['inpa_depth'] = npy(common['tensorInpaDepth'])
        arrays['inpa_disparity'] = npy(common['tensorInpaDisparity'])
        for i, call in enumerate(rec.calls):
            for k, v in call.items():
                arrays['inpaint%d_%s' % (i, k)] = npy(v)
        # per-frame z-buffers (post-degrid, serial schedule) of the frame loop: the last
        # len(steps) render_pointcloud calls, 3 launches each + 1 fill launch
        per = [t for t in tr if t[0] == 'kernel_pointrender_updateDegrid'][-len(steps):]
        arrays['frame_zee_serial'] = np.stack([npy(t[1]['zee'])[0, 0] for t in per])
        save('kenburns_' + tag, **arrays)


def gen_autozoom():
    """process_autozoom of the reference (common.py:114-170).  The function is dead code there and cannot run as written: it calls
    process_shift without the objectCommon argument that function takes (:146-152).  Its BODY runs here unmodified; the one missing
    argument is supplied by wrapping the module's process_shift for the duration of the call.  Kept: inputs, settings, the window it returns."""
    h = HARNESS
    cases = {}
    for tag, (H, W, seed, zoom, shift) in {'a': (48, 64, 71, 1.25, 8.0), 'b': (40, 56, 72, 1.5, 10.0)}.items():
        image, disp = synthetic.make_rgbd(H, W, seed, 'smooth')
        depth = (512.0 * 120) / (disp + 1e-7)
        pts = h.C.depth_to_points(depth, 512.0)
        common = {'dblFocal': 512.0, 'dblBaseline': 120, 'intWidth': W, 'intHeight': H, 'objectDepthrange': synthetic.depthrange_of(depth),
                  'tensorRawPoints': pts.view(1, 3, -1), 'tensorRawImage': image, 'tensorRawDisparity': disp, 'tensorRawDepth': depth}
        settings = {'dblShift': shift, 'dblZoom': zoom, 'objectFrom': {'dblCenterU': W / 2.0, 'dblCenterV': H / 2.0, 'intCropWidth': W, 'intCropHeight': H}}
        real = h.C.process_shift
        h.C.process_shift = lambda s, _c=common, _r=real: _r(s, _c)
        try:
            window = h.C.process_autozoom(settings, common)
        finally:
            h.C.process_shift = real
        dr = common['objectDepthrange']
        cases[tag] = dict(image=npy(image), disparity=npy(disp), zoom=np.float64(zoom), shift=np.float64(shift),
                          depthrange=np.array([dr[0], dr[1], dr[2][0], dr[2][1], dr[3][0], dr[3][1]], np.float64),
                          window=np.array([window['dblCenterU'], window['dblCenterV'], window['intCropWidth'], window['intCropHeight']], np.float64))
    save('autozoom', **{'%s_%s' % (t, k): v for t, c in cases.items() for k, v in c.items()})


def gen_kenburns_at_size():
    """process_kenburns of the reference at 256 x 320 on a smooth scene (KBE path with two Inpaint passes, and a dolly zoom):
    only the seed, the steps and the reference's pre-crop uint8 FRAMES are kept (the inputs are regenerated from the seed by
    ken_burns_effect_amd.synthetic; the Inpaint weights from seeded_fill_).  What the product route -- Jacobi degrid, its own
    accumulation order, its own MIOpen Inpaint -- delivers is measured against these frames (tests/test_hip_reference.py)."""
    h = HARNESS
    net = h.PI.Inpaint().eval()
    synthetic.seeded_fill_(net, 3)
    H, W = 256, 320
    # (kbe_photo: the same scene kind with photograph-like colours -- synthetic.photo_like -- where a pixel that takes another source
    # point moves by a few counts instead of up to 255: the PSNR between two legal degrid schedules means something there)
    for tag, (seed, dolly, steps, colours) in {'kbe': (61, False, [0.0, 0.35, 0.7, 1.0], 'noise'), 'dolly': (62, True, [0.0, 0.4, 0.8], 'noise'),
                                               'kbe_photo': (63, False, [0.0, 0.35, 0.7, 1.0], 'photo'), 'dolly_photo': (64, True, [0.0, 0.4, 0.8], 'photo')}.items():
        image, disp = synthetic.make_rgbd(H, W, seed, 'smooth', colours=colours)
        depth = (512.0 * 120) / (disp + 1e-7)
        pts = h.C.depth_to_points(depth, 512.0)
        common = {'dblFocal': 512.0, 'dblBaseline': 120, 'intWidth': W, 'intHeight': H,
                  'dblDispmin': disp.min().item(), 'dblDispmax': disp.max().item(),
                  'objectDepthrange': synthetic.depthrange_of(depth),
                  'tensorRawPoints': pts.view(1, 3, -1), 'tensorRawImage': image,
                  'tensorRawDisparity': disp, 'tensorRawDepth': depth}
        ofrom, oto = synthetic.default_windows(H, W, dolly)
        settings = {'dblSteps': steps, 'objectFrom': ofrom, 'objectTo': oto, 'boolInpaint': True, 'dolly': dolly}
        with torch.no_grad():
            frames, _ = h.traced(h.C.process_kenburns, settings, common, net)
        extra = {} if colours == 'noise' else {'colours': np.array(colours)}         # (the two noise fixtures keep their bytes)
        save('kenburns_at_size_' + tag, frames=np.stack(frames), steps=np.array(steps, np.float64), dolly=np.bool_(dolly), seed=np.int64(seed),
             H=np.int64(H), W=np.int64(W), n_points=np.int64(common['tensorInpaPoints'].shape[-1]), **extra)


def gen_generate_mask():
    """generate_mask (common.py:689-830): per-point ownership mask of the z-splat, serial point order.
    Inputs are image rasters (N == H*W, the mask is viewed as an image for the median-5), batch 2, with
    laplacian-invalid points at depth 0 (culled), a shift that moves part of the cloud out of view, and a
    second case where several points per pixel compete (zoom: shift towards the camera)."""
    h = HARNESS
    arrays = {}
    for tag, (H, W, focal, baseline, shifts) in {
            'a': (40, 56, 512.0, 120, [(-6.0, 3.0, -20.0), (4.0, -2.0, 35.0)]),
            'b': (32, 48, 409.6, 40.0, [(1.5, 0.5, 120.0), (-30.0, 12.0, -60.0)]),
            'c': (12, 16, 512.0, 120, [(0.0, 0.0, 0.0), (0.5, -0.25, 5.0)])}.items():
        pts_l = []
        for b in range(2):
            image, disp = synthetic.make_rgbd(H, W, 20 + b, 'smooth', baseline)
            depth = (focal * baseline) / (disp + 0.0000001)
            valid = (h.C.spatial_filter(disp / disp.max(), 'laplacian').abs() < 0.03).float()
            pts_l.append(h.C.depth_to_points(depth * valid if tag != 'c' else depth, focal).view(1, 3, -1))
        pts = torch.cat(pts_l, 0).contiguous()
        shift = torch.tensor(shifts, dtype=torch.float32).view(2, 3, 1)
        if tag == 'c':
            # point 0 first owns its pixel and is then displaced by nearer points on the same ray (the reference
            # only clears a displaced owner when its index is > 0: point 0 keeps mask 1); also a displaced point > 0
            pts[:, :, 0] = pts[:, :, 27]
            pts[:, :, 5] = pts[:, :, 27] * 0.5
            pts[:, :, 9] = pts[:, :, 27] * 0.25
            pts[0, :, 13] = pts[0, :, 30] * 0.5
        arrays.update({tag + '_points': npy(pts), tag + '_shift': npy(shift), tag + '_W': np.int32(W), tag + '_H': np.int32(H),
                       tag + '_focal': np.float64(focal), tag + '_baseline': np.float64(baseline),
                       tag + '_baseline_is_int': np.bool_(isinstance(baseline, int))})
        for mode in ('fma', 'nofma'):
            h.mode = mode
            out, tr = h.traced(h.C.generate_mask, pts, shift, W, H, focal, baseline)
            assert [t[0] for t in tr] == ['kernel_pointrender_updateZee']
            arrays[tag + '_masks_raw_' + mode] = npy(tr[0][1]['masks'])
            arrays[tag + '_zee_' + mode] = npy(tr[0][1]['zee'])
            arrays[tag + '_ids_' + mode] = npy(tr[0][1]['id_memory'].view(torch.int32))
            arrays[tag + '_masks_' + mode] = npy(out)
    h.mode = 'fma'
    save('generate_mask', **arrays)


if __name__ == '__main__':
    HARNESS = Harness()
    torch.set_grad_enabled(False)
    torch.set_num_threads(1)   # bit-stable conv results
    which = sys.argv[1:] or ['render', 'fill', 'torch_helpers', 'partial_conv', 'inpaint', 'kenburns', 'kenburns_at_size', 'disparity', 'generate_mask', 'autozoom']
    for w in which:
        globals()['gen_' + w]()
