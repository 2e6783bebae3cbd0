"""GPU parity against REFERENCE-RUN vectors for the rows round 1 only covered on the CPU (a3, a4, a9,
a11, a13, a14, a15; 8f: f2): the set-up half of process_kenburns, the networks on MIOpen, and the reference's own
frames reproduced by the HIP kernels.

Two kernel sets are used:
  * the product set (`_native.kernels()`): tile renderer, out-of-place (Jacobi) degrid;
  * the product's serial-schedule route (`KBE_DEGRID=serial`: `_native.HipSerialScheduleKernels`, since round 6 a documented
    option of the package instead of a class in this file): the library's stage-by-stage HIP kernels with the degrid run under
    the SERIAL schedule (`kbe_degrid_serial`).  The golden vectors were produced by the reference's kernel text
    executed one element after the other (tests/golden/make_golden.py), so this is the schedule under which a HIP
    run can be compared with a reference run bit for bit (z-buffers) and count for count (frames).

Tolerances for MIOpen fp32 convolutions (different algorithms and summation orders than the CPU run that produced the
fixtures) are written at each assertion; index / mask / byte work is exact.
"""
import warnings

import numpy as np
import pytest
import torch

from conftest import assert_bits_equal, at_size_scene, load_golden, psnr_u8

pytestmark = pytest.mark.gpu

# MIOpen fp32 vs the fixture's CPU fp32.  Measured on MI355X (ROCm 7.2, MIOPEN_FIND_MODE=FAST): image <= 4.3e-6,
# disparity <= 1e-6 of its range.  The bars leave ~5-20x for other solver choices; 2e-5 is 1/200 of a uint8 count.
TOL_IMAGE = 2e-5
TOL_DISPARITY_REL = 1e-5


@pytest.fixture(scope='module')
def K():
    from ken_burns_effect_amd import _native
    assert torch.cuda.is_available(), 'GPU tests need a GPU'
    return _native.kernels()          # raises if libkbe_hip.so is missing: no fallback


def g(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def c(t):
    return t.detach().cpu().numpy()


def _depthrange(v):
    return (float(v[0]), float(v[1]), (int(v[2]), int(v[3])), (int(v[4]), int(v[5])))


def _scene(z, K):
    from ken_burns_effect_amd import synthetic
    image, disp = g(z['image']), g(z['disparity'])
    H, W = image.shape[2:]
    depth = (512.0 * 120) / (disp + 1e-7)
    oc = {'dblFocal': 512.0, 'dblBaseline': 120, 'intWidth': W, 'intHeight': H,
          'dblDispmin': disp.min().item(), 'dblDispmax': disp.max().item(), 'objectDepthrange': _depthrange(z['depthrange']),
          'tensorRawPoints': K.depth_to_points(depth, 512.0).view(1, 3, -1), 'tensorRawImage': image,
          'tensorRawDisparity': disp, 'tensorRawDepth': depth}
    ofrom, oto = synthetic.default_windows(H, W, bool(z['dolly']))
    settings = {'dblSteps': [float(s) for s in z['steps']], 'objectFrom': ofrom, 'objectTo': oto, 'boolInpaint': True,
                'dolly': bool(z['dolly']), 'boolCrop': False}
    return settings, oc


class ReplayInpaint:
    """Returns what the reference's Inpaint returned for the same call (recorded in the fixture), on the GPU."""

    def __init__(self, z):
        self.z, self.i = z, 0

    def pointcloud_inpainting(self, tensorImage, tensorDisparity, tensorShift, objectCommon, dblFocal=None):
        out = {k: g(self.z['inpaint%d_%s' % (self.i, k)]) for k in ('tensorExisting', 'tensorImage', 'tensorDisparity')}
        self.i += 1
        return out


# ---------------------------------------------------------------------------------------
# the reference's own frames on the HIP kernels (VERDICT r1 "what's weak" 1b / 1c)
# ---------------------------------------------------------------------------------------

@pytest.mark.parametrize('name', ['kenburns_kbe', 'kenburns_dolly'])
def test_reference_run_frames_on_the_hip_kernels_serial_schedule(K, monkeypatch, name):
    """process_kenburns of the reference (kernel text executed serially) vs the same call on the HIP kernels with the
    serial degrid: per-frame z-buffers bit for bit, the appended point cloud bit for bit, frames within one uint8 count
    (the accumulation order of atomicAdd is the only freedom left; the reference's own is not defined either)."""
    from ken_burns_effect_amd import _native, common
    z = load_golden(name)
    # the PRODUCT's serial-schedule route: KBE_DEGRID=serial makes _native.kernels() -- what common uses when
    # nobody injects a kernel set -- the stage-by-stage kernels with kbe_degrid_serial; its z-buffers are logged here to be compared
    monkeypatch.setenv('KBE_DEGRID', 'serial')
    ks = _native.kernels()
    assert isinstance(ks, _native.HipSerialScheduleKernels) and common._K() is ks
    monkeypatch.setattr(ks, 'zee_log', [])
    settings, oc = _scene(z, K)
    frames = common.process_kenburns(settings, oc, ReplayInpaint(z) if not z['dolly'] else None)
    for key, ref in (('tensorInpaPoints', 'inpa_points'), ('tensorInpaImage', 'inpa_image'), ('tensorInpaDepth', 'inpa_depth'),
                     ('tensorInpaDisparity', 'inpa_disparity')):
        assert_bits_equal(c(oc[key]), z[ref], key)
    assert len(frames) == len(z['frames']) == len(ks.zee_log)
    for i, (f, ref) in enumerate(zip(frames, z['frames'])):
        assert_bits_equal(c(ks.zee_log[i])[0, 0], z['frame_zee_serial'][i], 'frame %d: z-buffer after the serial degrid' % i)
        d = np.abs(f.astype(np.int32) - ref.astype(np.int32))
        assert d.max() <= 1 and (d > 0).mean() < 2e-3, 'frame %d: %d values differ, max %d' % (i, int((d > 0).sum()), int(d.max()))


# floors for the shipped route against the reference-run frames (tests/test_host_logic.py measures the same on the CPU oracle:
# photograph-like colours 44-52 dB on the KBE path -- what is left are pixels one degrid schedule hands to the seeded-random
# Inpaint network's colours and the other to the surface -- and 50-59 dB on the dolly zoom)
PRODUCT_FLOOR_DB = {'kbe': 30.0, 'dolly': 30.0, 'kbe_photo': 43.0, 'dolly_photo': 49.0}


@pytest.mark.parametrize('tag', ['kbe', 'dolly', 'kbe_photo', 'dolly_photo'])
def test_product_route_against_reference_run_frames_at_size(K, tag):
    """The SHIPPED route -- the tile kernels with their out-of-place (Jacobi) degrid, their own accumulation order, the Inpaint
    network on MIOpen -- against the frames the reference's process_kenburns produced at 256 x 320 (kernel text run serially;
    tests/golden/kenburns_at_size_*.npz).  tests/test_host_logic.py shows on the CPU that the serial schedule reproduces those
    frames byte for byte and that the Jacobi schedule moves only pixels whose degridded z differs, and filled holes.  The scene's
    colours are white noise, so a pixel that takes another source point moves by up to 255: the PSNR below is a floor, not what a
    photograph would show.  Measured on MI355X: KBE 32-37 dB with 0.5-0.65 % of the pixels moved, dolly 34-44 dB with 0.1-0.2 %."""
    from ken_burns_effect_amd import common, synthetic
    from ken_burns_effect_amd.pointcloud_inpainting import Inpaint
    z = load_golden('kenburns_at_size_' + tag)
    settings, oc = at_size_scene(z, 'cuda', K.depth_to_points)
    net = synthetic.seeded_fill_(Inpaint(), 3).cuda().eval()
    with torch.no_grad():
        frames = common.process_kenburns(settings, oc, net)
    assert len(frames) == len(z['frames'])
    # the grown cloud: the set-up's forward warp degrids under the Jacobi schedule too, so its hole mask -- and with it the number
    # of appended points -- differs slightly from the reference run's (measured: 89 321 against 89 622 points)
    assert abs(oc['tensorInpaPoints'].shape[-1] - int(z['n_points'])) <= 0.01 * int(z['n_points'])
    for i, (f, ref) in enumerate(zip(frames, z['frames'])):
        moved = (np.abs(f.astype(np.int32) - ref.astype(np.int32)).max(axis=2) > 1).mean()
        db = psnr_u8(f, ref)
        print('%s frame %d: %.2f dB against the reference-run frame, %.3f %% of the pixels moved by more than one count' % (tag, i, db, 100 * moved))
        assert db > PRODUCT_FLOOR_DB[tag] and moved < 0.01, 'frame %d: %.2f dB against the reference-run frame, %.3f %% of the pixels moved by more than one count' % (i, db, 100 * moved)


@pytest.mark.parametrize('case', ['render_f512', 'render_f409', 'render_f153', 'render_noise', 'render_b2c7'])
def test_serial_degrid_entry_bit_exact_against_reference_vectors(K, case):
    z = load_golden(case)
    pts = g(z['points'])
    W, H = int(z['W']), int(z['H'])
    baseline = int(z['baseline']) if bool(z['baseline_is_int']) else float(z['baseline'])
    zkeys, _ = K.zsplat(pts, W, H, float(z['focal']), baseline)
    assert_bits_equal(c(K.degrid_serial(zkeys=zkeys)), z['zee_serial_fma'], 'serial-schedule degrid from keys')
    assert_bits_equal(c(K.degrid_serial(zee=g(z['zee_pre_fma']))), z['zee_serial_fma'], 'serial-schedule degrid from floats')


def test_serial_degrid_entry_at_size_equals_the_oracle(K, oracle):
    """512 x 384, white-noise depth (long cascades: the schedules differ in most pixels), batch 2."""
    rng = np.random.default_rng(5)
    zee = (1e6 - 61440.0 / (rng.random((2, 1, 384, 512)) * 900 + 100)).astype(np.float32)
    zee[rng.random(zee.shape) < 0.1] = 1e6
    ser = oracle.degrid(torch.from_numpy(zee), 'serial').numpy()
    assert (ser != oracle.degrid(torch.from_numpy(zee), 'jacobi').numpy()).mean() > 0.05
    assert_bits_equal(c(K.degrid_serial(zee=g(zee))), ser, 'serial degrid at size')


# ---------------------------------------------------------------------------------------
# a3 process_inpaint / build_pointcloud on the product kernel set
# ---------------------------------------------------------------------------------------

def test_process_inpaint_appends_the_reference_points(K):
    """common.py:47-81 on the GPU (laplacian validity, unprojection, hole gather, append): the grown cloud equals the
    reference's bit for bit.  The Inpaint outputs are replayed from the fixture, so nothing but process_inpaint's own
    arithmetic is under test."""
    from ken_burns_effect_amd import common
    z = load_golden('kenburns_kbe')
    settings, oc = _scene(z, K)
    common.build_pointcloud(settings, oc, ReplayInpaint(z))
    assert oc['tensorInpaPoints'].is_cuda and oc['tensorInpaPoints'].shape[2] == z['inpa_points'].shape[2] > oc['intWidth'] * oc['intHeight']
    for key, ref in (('tensorInpaPoints', 'inpa_points'), ('tensorInpaImage', 'inpa_image'), ('tensorInpaDepth', 'inpa_depth'),
                     ('tensorInpaDisparity', 'inpa_disparity')):
        assert_bits_equal(c(oc[key]), z[ref], key)


# ---------------------------------------------------------------------------------------
# a4 process_load
# ---------------------------------------------------------------------------------------

def test_process_load_on_the_gpu_equals_the_cpu_restatement(K, oracle, monkeypatch):
    from ken_burns_effect_amd import common, synthetic
    image, disp = synthetic.make_rgbd(72, 88, 9)
    img8 = (image[0].permute(1, 2, 0).numpy() * 255).astype(np.uint8)
    gpu = {}
    common.process_load(img8, {'tensorDisparity': disp, 'device': 'cuda:0'}, gpu)
    monkeypatch.setattr(common, '_kernel_set', oracle.OracleKernels(schedule='serial'))
    cpu = {}
    common.process_load(img8, {'tensorDisparity': disp, 'device': 'cpu'}, cpu)
    assert set(gpu) == set(cpu)
    for k, v in cpu.items():
        if torch.is_tensor(v):
            assert gpu[k].is_cuda and gpu[k].shape == v.shape
            assert_bits_equal(c(gpu[k]), v.numpy(), k)
        else:
            assert gpu[k] == v, k
    assert gpu['dblBaseline'] == 40.0 and gpu['dblFocal'] == 512.0 and gpu['tensorRawPoints'].shape == (1, 3, 72 * 88)


# ---------------------------------------------------------------------------------------
# a9 Inpaint (MIOpen) and pointcloud_inpainting
# ---------------------------------------------------------------------------------------

@pytest.fixture(scope='module')
def net():
    from ken_burns_effect_amd import synthetic
    from ken_burns_effect_amd.pointcloud_inpainting import Inpaint
    return synthetic.seeded_fill_(Inpaint().eval(), 3).cuda()


def _close(out, ref, tol, what):
    err = float(np.abs(c(out) - ref).max())
    print('%s: max abs error %.3g (bar %.3g)' % (what, err, tol))
    assert err <= tol, '%s: max abs error %.3g > %.3g' % (what, err, tol)


def test_inpaint_forward_on_miopen_matches_reference(net):
    z = load_golden('inpaint')
    with torch.no_grad():
        net.normalize_images_disp(g(z['image']), g(z['disparity']), not_normed=True)
        out = net(tensorData=g(z['fw_data']), tensorMasks=g(z['fw_mask']))
    assert_bits_equal(c(out['tensorExisting']), z['fw_mask'], 'tensorExisting is the input mask')
    _close(out['tensorImage'], z['fw_image'], TOL_IMAGE, 'Inpaint.forward image')
    _close(out['tensorDisparity'], z['fw_disparity'], TOL_DISPARITY_REL * max(1.0, float(np.abs(z['fw_disparity']).max())), 'Inpaint.forward disparity')
    with torch.no_grad():
        out = net(tensorMasks=g(z['fw_mask']), tensorImage=g(z['image']), tensorDisparity=g(z['disparity']))
    _close(out['tensorImage'], z['fi_image'], TOL_IMAGE, 'Inpaint.forward(image, disparity) image')
    _close(out['tensorDisparity'], z['fi_disparity'], TOL_DISPARITY_REL * max(1.0, float(np.abs(z['fi_disparity']).max())), 'Inpaint.forward(image, disparity) disparity')


def test_pointcloud_inpainting_on_the_hip_kernels_matches_reference(net, K, monkeypatch):
    """pointcloud_inpainting.py:185-213 end to end on the GPU: unprojection, validity mask, context net, 68-channel
    forward warp, median-5 hole dilation, GridNet.  Serial-schedule kernel set: z-buffer and hole mask bit for bit
    against the reference run; then the product set (tile renderer, Jacobi): same holes wherever the two legal schedules
    agree, colours within the bars."""
    from ken_burns_effect_amd import common
    z = load_golden('inpaint')
    image, disp = g(z['image']), g(z['disparity'])
    H, W = image.shape[2:]
    oc = {'dblFocal': 512.0, 'dblBaseline': 120, 'intWidth': W, 'intHeight': H}
    from ken_burns_effect_amd import _native
    monkeypatch.setenv('KBE_DEGRID', 'serial')
    ks = _native.kernels()
    monkeypatch.setattr(ks, 'zee_log', [])
    with torch.no_grad():
        out = net.pointcloud_inpainting(image, disp, g(z['pi_shift']), oc)
    assert_bits_equal(c(ks.zee_log[-1]), z['pi_zee'], 'z-buffer of the 68-channel forward warp (serial schedule)')
    assert_bits_equal(c(out['tensorExisting']), z['pi_existing'], 'existing mask after median-5 dilation')
    _close(out['tensorImage'], z['pi_image'], TOL_IMAGE, 'pointcloud_inpainting image')
    _close(out['tensorDisparity'], z['pi_disparity'], TOL_DISPARITY_REL * max(1.0, float(np.abs(z['pi_disparity']).max())), 'pointcloud_inpainting disparity')
    monkeypatch.delenv('KBE_DEGRID')
    with torch.no_grad():
        prod = net.pointcloud_inpainting(image, disp, g(z['pi_shift']), oc)
    same = c(prod['tensorExisting']) == z['pi_existing']
    print('product (Jacobi) vs reference-run (serial) hole mask: %.4f of pixels agree' % same.mean())
    assert same.mean() > 0.97
    assert_bits_equal(c(image), z['image'], 'inputs are not modified')


# ---------------------------------------------------------------------------------------
# a10 / a11 partial convolution and the partial-conv GridNet
# ---------------------------------------------------------------------------------------

def test_partial_conv_layers_on_the_gpu_match_reference(K):
    from ken_burns_effect_amd.partial_conv import PartialConv2d
    z = load_golden('partial_conv')
    for tag in 'abc':
        cin, cout, k, s, p = [int(v) for v in z['cfg_' + tag]]
        conv = PartialConv2d(cin, cout, kernel_size=k, stride=s, padding=p, bias=True, multi_channel=True, return_mask=True).cuda()
        with torch.no_grad():
            conv.weight.copy_(g(z['w_' + tag]))
            conv.bias.copy_(g(z['b_' + tag]))
            out, um = conv(g(z['x_' + tag]), g(z['m_' + tag]))
        assert_bits_equal(c(um.contiguous()), z['mask_' + tag], 'update_mask ' + tag)
        _close(out, z['out_' + tag], 1e-5 * max(1.0, float(np.abs(z['out_' + tag]).max())), 'PartialConv2d ' + tag)


def test_partial_inpaint_forward_on_the_gpu_matches_reference(K):
    from ken_burns_effect_amd import synthetic
    from ken_burns_effect_amd.partial_inpainting import Inpaint
    z = load_golden('partial_inpaint')
    pnet = synthetic.seeded_fill_(Inpaint().eval(), 5).cuda()
    image, disp = synthetic.make_rgbd(32, 40, 43, 'smooth')
    with torch.no_grad():
        pnet.normalize_images_disp(image.cuda(), disp.cuda(), not_normed=True)
        out = pnet(tensorData=g(z['fw_data']), tensorMasks=g(z['fw_mask']))
    assert list(out['tensorMaskOut'].shape) == [int(v) for v in z['fw_existing_shape']] and out['tensorExisting'].shape[1] == 1
    _close(out['tensorImage'], z['fw_image'], 2 * TOL_IMAGE, 'partial Inpaint image')
    _close(out['tensorDisparity'], z['fw_disparity'], 2 * TOL_DISPARITY_REL * max(1.0, float(np.abs(z['fw_disparity']).max())), 'partial Inpaint disparity')


def test_partial_inpaint_forward_with_a_fractional_mask_takes_the_unfused_pairs(K):
    """The fused pairs of the partial GridNet skip the second `input * mask` (/root/reference/utils/partial_conv.py:61), which is
    right for masks of 0 / 1 only.  Inpaint.forward(tensorMasks=...) is a public entry point: with a FRACTIONAL mask the forward
    must still equal the reference's formulation (bench.partial_conv_reference_forward on every layer) -- it does because such
    a call takes the unfused pairs -- and a mask that is not [B,1,H,W] is refused by kbe_prelu_mask's wrapper."""
    import bench
    from ken_burns_effect_amd import _native, partial_conv, partial_inpainting, synthetic
    from ken_burns_effect_amd.partial_inpainting import Inpaint
    net = synthetic.seeded_fill_(Inpaint().eval(), 5).cuda()
    gen = torch.Generator(device='cpu').manual_seed(4)
    H, W = 64, 96
    data = torch.randn(1, 68, H, W, generator=gen).cuda()
    mask = (torch.rand(1, 1, H, W, generator=gen) * (torch.rand(1, 1, H, W, generator=gen) > 0.2).float()).cuda()      # zeros and fractions
    image, disp = synthetic.make_rgbd(H, W, 45, 'smooth')
    seen = []
    real = partial_inpainting._Pair._fused
    with torch.no_grad():
        net.normalize_images_disp(image.cuda(), disp.cuda(), not_normed=True)
        partial_inpainting._Pair._fused = lambda self: (seen.append(real(self)), seen[-1])[1]
        try:
            got = {k: v.clone() for k, v in net(tensorData=data, tensorMasks=mask).items() if torch.is_tensor(v)}
            assert seen and not any(seen), 'a fractional mask must not take the fused pairs'
            del seen[:]
            net(tensorData=data, tensorMasks=(mask > 0).float())
            assert seen and all(seen), 'a 0 / 1 mask takes them'
        finally:
            partial_inpainting._Pair._fused = real
        assert getattr(partial_inpainting._Pair._call, 'binary_masks', True) is True        # the call's switch is restored (per thread since round 6)
        fused_forward = partial_conv.PartialConv2d.forward
        partial_conv.PartialConv2d.forward = bench.partial_conv_reference_forward
        for m in net.modules():
            if isinstance(m, partial_conv.PartialConv2d):
                m.last_size = (None, None, None, None)
        try:
            ref = {k: v.clone() for k, v in net(tensorData=data, tensorMasks=mask).items() if torch.is_tensor(v)}
        finally:
            partial_conv.PartialConv2d.forward = fused_forward
            for m in net.modules():
                if isinstance(m, partial_conv.PartialConv2d):
                    m.last_size = (None, None, None, None)
    assert torch.equal(got['tensorMaskOut'], ref['tensorMaskOut'])
    _close(got['tensorImage'], ref['tensorImage'].cpu().numpy(), 2 * TOL_IMAGE, 'partial Inpaint image, fractional mask')
    _close(got['tensorDisparity'], ref['tensorDisparity'].cpu().numpy(), 2 * TOL_DISPARITY_REL * max(1.0, float(ref['tensorDisparity'].abs().max())), 'partial Inpaint disparity, fractional mask')
    with pytest.raises(_native.KbeError):
        K.prelu_mask(torch.randn(1, 4, 8, 8, device='cuda'), torch.full((4,), 0.25, device='cuda'), torch.ones(1, 4, 8, 8, device='cuda'))


def test_partial_inpaint_forward_at_1024_fused_epilogue_against_the_reference_formulation(K):
    """BASELINE configs[3] "4b" at its size: the partial-convolution Inpaint.forward on a 1024 x 1024 input (68 channels, a mask
    with a fifth of the pixels missing, in blobs and single pixels), every PartialConv2d through the fused epilogue
    (kbe_pconv_epilogue) against the same network with every layer in the reference's formulation (the mask's own convolution
    and five element-wise passes, /root/reference/utils/partial_conv.py:58-77; bench.partial_conv_reference_forward): the
    propagated masks equal; image and disparity: the root mean square of the difference within the tolerance of the 32 x 40
    fixture test above, the worst of the million pixels within four times it (+ MIOpen's run-to-run noise, measured here)."""
    import bench
    from ken_burns_effect_amd import partial_conv, synthetic
    from ken_burns_effect_amd.partial_inpainting import Inpaint
    big = 1024
    net = synthetic.seeded_fill_(Inpaint().eval(), 5).cuda()
    gen = torch.Generator(device='cpu').manual_seed(21)
    data = torch.randn(1, 68, big, big, generator=gen).cuda()
    coarse = (torch.rand(1, 1, big // 16, big // 16, generator=gen) > 0.15).float()
    mask = (torch.nn.functional.interpolate(coarse, size=(big, big), mode='nearest') * (torch.rand(1, 1, big, big, generator=gen) > 0.05).float()).cuda()
    image, disp = synthetic.make_rgbd(big, big, 44, 'smooth')
    with torch.no_grad():
        net.normalize_images_disp(image.cuda(), disp.cuda(), not_normed=True)
        fused = net(tensorData=data, tensorMasks=mask)
        fused = {k: v.clone() for k, v in fused.items() if torch.is_tensor(v)}
        fused_again = {k: v.clone() for k, v in net(tensorData=data, tensorMasks=mask).items() if torch.is_tensor(v)}
        fused_forward = partial_conv.PartialConv2d.forward
        partial_conv.PartialConv2d.forward = bench.partial_conv_reference_forward
        for m in net.modules():
            if isinstance(m, partial_conv.PartialConv2d):
                m.last_size = (None, None, None, None)
        try:
            ref = net(tensorData=data, tensorMasks=mask)
            ref = {k: v.clone() for k, v in ref.items() if torch.is_tensor(v)}
            # the same formulation twice more: what MIOpen's own run-to-run noise is
            again = [{k: v.clone() for k, v in net(tensorData=data, tensorMasks=mask).items() if torch.is_tensor(v)} for _ in range(2)]
        finally:
            partial_conv.PartialConv2d.forward = fused_forward
    assert torch.equal(fused['tensorMaskOut'], ref['tensorMaskOut']) and torch.equal(fused['tensorExisting'], ref['tensorExisting'])
    assert 0.05 < float(1 - mask.mean()) < 0.4
    # MIOpen's fp32 solvers at this size are not bit-reproducible from run to run (split-K accumulation by atomics): two runs of the
    # SAME formulation differ by up to ~5e-5 in the image after thirty layers (measured: 3e-5 - 5e-5), which is where the two
    # formulations sit as well -- so the bar is twice the fixture test's tolerance plus twice that noise, measured here
    # (the largest of four pairs of runs of one formulation -- a single pair's maximum over three million pixels scatters by a
    # factor of two, and the test flaked once on a pair that happened to agree well)
    pairs = [(again[0], ref), (again[1], ref), (again[0], again[1]), (fused_again, fused)]
    noise_i = max(float((a['tensorImage'] - b['tensorImage']).abs().max()) for a, b in pairs)
    noise_d = max(float((a['tensorDisparity'] - b['tensorDisparity']).abs().max()) for a, b in pairs)
    print('run-to-run noise of the reference formulation: image %.3g, disparity %.3g' % (noise_i, noise_d))
    # Two bars.  (1) The TYPICAL pixel: the root mean square of the difference stays below the fixture test's tolerance -- a
    # systematic difference between the formulations would show here whatever the extremes do.  (2) The WORST of a million
    # pixels: four times that tolerance (+ twice the measured noise).  The 32 x 40 fixture's bar of 2 TOL is a maximum over 1 280
    # pixels; the maximum over 820 times as many samples of the same distribution sits higher, and where the box's MIOpen picks
    # deterministic solvers (noise 0: the round's last collection) the two formulations still call DIFFERENT convolutions -- with
    # and without the bias, on masked and unmasked inputs -- whose accumulation orders differ: measured maxima 3e-5 - 4.9e-5.
    d_i = fused['tensorImage'] - ref['tensorImage']
    rms_i = float(d_i.double().pow(2).mean().sqrt())
    print('partial Inpaint image at 1024^2: rms difference %.3g (bar %.3g)' % (rms_i, TOL_IMAGE))
    assert rms_i <= TOL_IMAGE
    _close(fused['tensorImage'], c(ref['tensorImage']), 4 * TOL_IMAGE + 2 * noise_i, 'partial Inpaint image at 1024^2')
    dref = c(ref['tensorDisparity'])
    tol_d = TOL_DISPARITY_REL * max(1.0, float(np.abs(dref).max()))
    rms_d = float((fused['tensorDisparity'] - ref['tensorDisparity']).double().pow(2).mean().sqrt())
    print('partial Inpaint disparity at 1024^2: rms difference %.3g (bar %.3g)' % (rms_d, tol_d))
    assert rms_d <= tol_d
    _close(fused['tensorDisparity'], dref, 4 * tol_d + 2 * noise_d, 'partial Inpaint disparity at 1024^2')


# ---------------------------------------------------------------------------------------
# a13 / a14 Disparity, Refine
# ---------------------------------------------------------------------------------------

def test_disparity_and_refine_networks_on_miopen_match_reference():
    from ken_burns_effect_amd import synthetic
    from ken_burns_effect_amd.disparity_estimation import Disparity
    from ken_burns_effect_amd.disparity_refinement import Refine, RefinePretrained
    z = load_golden('disparity')
    dnet = synthetic.seeded_fill_(Disparity().eval(), 11).cuda()
    with torch.no_grad():
        out = dnet(g(z['image']), g(z['semantics']))
    assert out.shape == (1, 1, 32, 48)
    _close(out, z['disp_out'], 2e-5 * max(1.0, float(np.abs(z['disp_out']).max())), 'Disparity')
    for tag, cls in (('refine', Refine), ('refinep', RefinePretrained)):
        rnet = synthetic.seeded_fill_(cls().eval(), 13).cuda()
        coarse = g(z['coarse'])
        with torch.no_grad():
            out = rnet(g(z['image']), coarse)
        assert out.shape == (1, 1, 64, 96)
        _close(out, z[tag + '_out'], 2e-5 * max(1.0, float(np.abs(z[tag + '_out']).max())), tag)
        assert_bits_equal(c(coarse), z['coarse'], 'inputs untouched')


def test_networks_with_their_element_wise_passes_fused_against_the_stock_modules(monkeypatch):
    """The plain networks' blocks run their bias adds, activations, residual adds and x2 upsamplings in fused HIP passes
    (pointcloud_inpainting._main_fused; the tests above put that path against the reference's fixtures).  Here the same networks
    with KBE_FUSED_LAYERS=0 -- the stock nn.Sequential modules -- on the same inputs, at sizes whose rows are odd further down
    the grid (the cropped stream from below): the outputs must agree to the last few bits of the upsampling's blend."""
    from ken_burns_effect_amd import synthetic
    from ken_burns_effect_amd.disparity_estimation import Disparity
    from ken_burns_effect_amd.disparity_refinement import Refine, RefinePretrained
    from ken_burns_effect_amd.pointcloud_inpainting import Inpaint
    gen = torch.Generator().manual_seed(5)

    def both(fn):
        with torch.no_grad():
            monkeypatch.setenv('KBE_FUSED_LAYERS', '1')
            fused = fn()
            monkeypatch.setenv('KBE_FUSED_LAYERS', '0')
            stock = fn()
        monkeypatch.delenv('KBE_FUSED_LAYERS')
        return fused, stock

    for (H, W) in [(96, 128), (100, 140)]:                      # 100 x 140: rows of 50 x 70, 25 x 35, 13 x 18 in the Inpaint grid
        inp = synthetic.seeded_fill_(Inpaint().eval(), 3).cuda()
        image, disp = torch.rand(1, 3, H, W, generator=gen).cuda(), (torch.rand(1, 1, H, W, generator=gen) * 50 + 5).cuda()
        mask = (torch.rand(1, 1, H, W, generator=gen) > 0.2).float().cuda()
        fused, stock = both(lambda: inp(tensorMasks=mask, tensorImage=image, tensorDisparity=disp))
        for key in ('tensorImage', 'tensorDisparity'):
            err, scale = float((fused[key] - stock[key]).abs().max()), max(1.0, float(stock[key].abs().max()))
            print('Inpaint %dx%d %s: fused against stock %.3g (scale %.3g)' % (H, W, key, err, scale))
            assert err <= 1e-5 * scale
    for (H, W) in [(64, 96), (72, 104)]:
        dnet = synthetic.seeded_fill_(Disparity().eval(), 11).cuda()
        image, sem = torch.rand(1, 3, H, W, generator=gen).cuda(), torch.rand(1, 512, (H + 15) // 16, (W + 15) // 16, generator=gen).cuda()
        fused, stock = both(lambda: dnet(image, sem))
        err, scale = float((fused - stock).abs().max()), max(1.0, float(stock.abs().max()))
        print('Disparity %dx%d: fused against stock %.3g (scale %.3g)' % (H, W, err, scale))
        assert fused.shape == stock.shape and err <= 1e-5 * scale
        for cls in (Refine, RefinePretrained):
            rnet = synthetic.seeded_fill_(cls().eval(), 13).cuda()
            coarse = torch.rand(1, 1, H // 4, W // 4, generator=gen).cuda() * 30
            fused, stock = both(lambda: rnet(image, coarse))
            err, scale = float((fused - stock).abs().max()), max(1.0, float(stock.abs().max()))
            print('%s %dx%d: fused against stock %.3g (scale %.3g)' % (cls.__name__, H, W, err, scale))
            assert err <= 1e-5 * scale


# ---------------------------------------------------------------------------------------
# a15 / f2: Pipeline at BASELINE configs[1] (512 x 512, 64 frames, seeded weights)
# ---------------------------------------------------------------------------------------

def test_pipeline_config1_full_size_front_half_and_a_frame_against_the_oracle(K, oracle):
    """The whole Pipeline call of BASELINE.json configs[1] on the GPU.  Front half (resize, Semantics + Disparity,
    Refine, normalisation, depth, unprojection; pipeline.py:61-100) against the same modules run on the CPU (which
    test_models.py pins to the reference); then frame 37 of the 64 against the oracle rendering the pipeline's own
    final cloud, crop + resize included."""
    from ken_burns_effect_amd import common, kbe, synthetic
    from ken_burns_effect_amd.pipeline import Pipeline
    size, steps = 512, 64
    image, _ = synthetic.make_rgbd(size, size, 21)
    zoom = kbe.windows_for(size, size, dict.fromkeys(('startU', 'startV', 'startW', 'startH', 'endU', 'endV', 'endW', 'endH')), False)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        pipe = Pipeline(model_paths=None, allow_random_weights=True, device='cuda:0', steps=steps)
        frames = pipe(image, zoom)
        cpu = Pipeline(model_paths=None, allow_random_weights=True, device='cpu', steps=steps)
    assert len(frames) == steps and frames[0].shape == (size, size, 3) and frames[0].dtype == np.uint8
    oc = pipe.objectCommon
    n = oc['tensorInpaPoints'].shape[2]
    assert n > size * size and oc['tensorInpaImage'].shape == (1, 3, n) and oc['tensorInpaDepth'].shape == (1, 1, n)
    # front half: the GPU's disparity against the CPU's (seeded weights amplify rounding: the bar is relative to the range)
    torch.set_num_threads(8)
    common_saved = common._kernel_set
    try:
        common._kernel_set = oracle.OracleKernels(schedule='jacobi')      # the CPU twin's depth_to_points
        ref = cpu.estimate(image)
    finally:
        common._kernel_set = common_saved
    scale = float(ref['tensorRawDisparity'].abs().max())
    err = float((oc['tensorRawDisparity'].cpu() - ref['tensorRawDisparity']).abs().max())
    print('Pipeline front half: disparity max abs error %.3g of a range of %.3g' % (err, scale))
    assert err <= 2e-5 * scale          # measured 1e-4 of a range of 120
    assert abs(oc['dblDispmax'] - ref['dblDispmax']) <= 2e-5 * scale and oc['objectDepthrange'][2] == ref['objectDepthrange'][2]
    # a frame from the middle of the path against the oracle on the SAME cloud (Jacobi schedule, as the product)
    k = 37
    settings = {'dblSteps': np.linspace(0.0, 1.0, steps).tolist(), 'objectFrom': zoom['objectFrom'], 'objectTo': zoom['objectTo'], 'dolly': False}
    focal, shift3 = common.frame_cameras(settings, oc)[k]
    ok = oracle.OracleKernels('jacobi')
    state = ok.prepare_cloud(oc['tensorInpaPoints'].cpu(), oc['tensorInpaImage'].cpu(), oc['tensorInpaDepth'].cpu(), size, size)
    crop = common.crop_size(settings)
    want = ok.crop_resize_u8(ok.render_frame(state, shift3, focal, oc['dblBaseline']), crop[0], crop[1]).numpy()
    d = np.abs(frames[k].astype(np.int32) - want.astype(np.int32))
    assert d.max() <= 2 and (d > 0).mean() < 2e-3, 'frame %d: %.5f of values differ, max %d' % (k, (d > 0).mean(), int(d.max()))
