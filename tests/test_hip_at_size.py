"""GPU parity at BASELINE.json's large sizes: configs[4] at size -- a 2048 x 2048 frame from a
16.8 M-point cloud, four points per pixel -- and rasters beyond 4096^2 (the 64-bit bucket-offset switch of
k_project), for BOTH scatter routes.  What is compared:
  * the z-buffer before the degrid against an independent torch scatter-min over the winner pixels, bit for bit;
  * a 256 x 256 window of the un-filled render against the CPU oracle run on the points that can reach the window;
  * the two hole-fill schedules against each other; the tile renderers against the stage-by-stage atomic kernels.
"""
import numpy as np
import pytest
import torch

from conftest import assert_bits_equal

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def K():
    from ken_burns_effect_amd import _native
    assert torch.cuda.is_available(), 'GPU tests need a GPU'
    return _native.kernels()


def c(t):
    return t.detach().cpu().numpy()


@pytest.fixture(scope='module')
def dense_cloud(K):
    """BASELINE.json configs[4]: the RGBD of a 4096^2 image unprojected with focal 2 F lands on sub-pixel positions of
    the 2048^2 view: 16 777 216 points, four per target pixel."""
    from ken_burns_effect_amd import synthetic
    size, up = 2048, 2
    image, disp = synthetic.make_rgbd(size * up, size * up, 0)
    depth = ((synthetic.FOCAL * synthetic.BASELINE) / (disp + 1e-7)).cuda()
    pts = K.depth_to_points(depth, synthetic.FOCAL * up).view(1, 3, -1)
    return size, pts, image.cuda().reshape(1, 3, -1), depth.reshape(1, 1, -1)


@pytest.mark.parametrize('fused', [False, True], ids=['bucket', 'fused'])
def test_config4_2048_frame_from_16m_points(K, oracle, dense_cloud, fused, monkeypatch):
    monkeypatch.setenv('KBE_LANES', '1')                # one scratch: 0.9 GB of buckets at this size
    from ken_burns_effect_amd import synthetic
    size, pts, img, dep = dense_cloud
    n = pts.shape[2]
    assert n == 16777216
    focal, Bl = synthetic.FOCAL, synthetic.BASELINE
    shift3 = (14.0, -9.0, -31.0)
    state = K.prepare_cloud(pts, img, dep, size, size, focal, raster=(size * 2, n))
    rf = torch.empty(4, size, size, device='cuda')
    ex, zp, zd = (torch.empty(size * size, device='cuda') for _ in range(3))
    K.render_frame(state, shift3, focal, Bl, render_f32=rf, existing_f32=ex, zee_f32=zd, zee_pre_f32=zp, stages=3, fused=fused)
    # (1) z-buffer before the degrid == scatter-min of dblError over each point's winner pixel (independent kernels + torch)
    _, winner = K.zsplat(pts, size, size, focal, Bl, shift3=shift3, want_winner=True)
    shifted = K.shift_points(pts, shift3)
    err = (1000000.0 - (focal * Bl) / (shifted[0, 2].double() + 0.0000001)).float()
    w = winner[0].long()
    keep = w >= 0
    want = torch.full((size * size,), 1000000.0, device='cuda')
    want.scatter_reduce_(0, w[keep], err[keep], reduce='amin')
    assert torch.equal(zp.view(torch.int32), want.view(torch.int32)), 'pre-degrid z-buffer at 2048^2 / 16.8 M points'
    assert int(keep.sum()) > 15000000
    del err, want, keep
    # (2) a window of the un-filled render against the oracle on the points that can reach it (+ 8 px of context; the
    # comparison stays 4 px inside so that nothing outside the subset can matter: degrid reads 1 px, a point colours 2 x 2)
    wx0, wy0, ws = 1100, 700, 256
    py, px = w // size, w % size
    sel = (w >= 0) & (px >= wx0 - 8) & (px < wx0 + ws + 8) & (py >= wy0 - 8) & (py < wy0 + ws + 8)
    sub = torch.nonzero(sel).view(-1)
    assert 200000 < sub.numel() < 400000
    sp = shifted[:, :, sub].cpu()
    data = torch.cat([img[:, :, sub], dep[:, :, sub]], 1).cpu()
    ref, ref_ex = oracle.render_pointcloud(sp, data, size, size, focal, Bl, 'jacobi')
    win = (slice(wy0 + 4, wy0 + ws - 4), slice(wx0 + 4, wx0 + ws - 4))
    got_ex = c(ex).reshape(size, size)[win]
    assert np.array_equal(got_ex > 0, ref_ex.numpy()[0, 0][win] > 0), 'same pixels covered'
    assert np.abs(got_ex - ref_ex.numpy()[0, 0][win]).max() <= 1e-4 * float(ref_ex.max())
    for ch in range(4):
        a, b = c(rf[ch])[win], ref.numpy()[0, ch][win]
        assert np.abs(a - b).max() <= 1e-4 * max(1.0, float(np.abs(b).max())), 'channel %d' % ch
    zref = oracle.degrid(oracle.zsplat(sp, size, size, focal, Bl)[0], 'jacobi').numpy()[0, 0]
    assert_bits_equal(c(zd).reshape(size, size)[win], zref[win], 'degridded z-buffer in the window')
    # (3) the hole-fill schedules give the same frame: half-wave, lane, lane with the tables of k_hole_dist (at this size the
    # block-distance table is read from memory, not from LDS)
    frames = [c(K.render_frame(state, shift3, focal, Bl, stages=7 | mode, fused=fused)) for mode in (8, 16, 8 | 512)]
    for other in frames[1:]:
        d = np.abs(frames[0].astype(np.int32) - other.astype(np.int32))
        assert d.max() <= 1 and (d > 0).mean() < 1e-3
    assert int((c(ex) <= 0).sum()) > 0, 'the frame has holes to fill'


@pytest.mark.parametrize('size', [4608, 5120])
def test_rasters_beyond_4096_tile_renderers_against_the_atomic_kernels(K, size, monkeypatch):
    """Above 4096^2 the buckets no longer end below byte 2^32 (k_project switches to 64-bit offsets); 5120^2 is also past
    the coarse skip map of the per-lane hole fill.  Bucket route and fused route against the stage-by-stage kernels."""
    monkeypatch.setenv('KBE_LANES', '1')
    from ken_burns_effect_amd import synthetic
    image, disp = synthetic.make_rgbd(size, size, seed=0)
    depth = ((synthetic.FOCAL * synthetic.BASELINE) / (disp + 1e-7)).cuda()
    pts = K.depth_to_points(depth, synthetic.FOCAL).view(1, 3, -1)
    img, dep = image.cuda().reshape(1, 3, -1), depth.reshape(1, 1, -1)
    shift3 = (size * 0.004, -size * 0.003, -size * 0.02)
    p2 = K.shift_points(pts, shift3)
    render, existing = K.render_pointcloud(p2, torch.cat([img, dep], 1), size, size, synthetic.FOCAL, synthetic.BASELINE, tiled=False)
    filled = K.fill_disocclusion(render, render[:, 3:4] * (existing > 0.0).float())
    want_u8 = K.frame_u8(filled)
    assert int((existing[0, 0] <= 0).sum()) > 1000
    state = K.prepare_cloud(pts, img, dep, size, size, synthetic.FOCAL, raster=(size, size * size))
    for fused in (False, True):
        rf = torch.empty(4, size, size, device='cuda')
        ex = torch.empty(size * size, device='cuda')
        frame = K.render_frame(state, shift3, synthetic.FOCAL, synthetic.BASELINE, render_f32=rf, existing_f32=ex, fused=fused)
        assert torch.equal(ex.view(size, size) > 0, existing[0, 0] > 0), 'validity masks (fused=%s)' % fused
        assert float((rf - filled[0]).abs().max()) <= 1e-4 * max(1.0, float(filled.abs().max()))
        assert int((frame.int() - want_u8.int()).abs().max()) <= 1
        # the fill with the tables of k_hole_dist on the same un-filled frame: byte-identical to the per-lane schedule
        unfilled = K.render_frame(state, shift3, synthetic.FOCAL, synthetic.BASELINE, stages=3, fused=fused).clone()
        filled_by = []
        for mode in (8, 8 | 512):
            buf = unfilled.clone()
            K.render_frame(state, shift3, synthetic.FOCAL, synthetic.BASELINE, out=buf, stages=4 | mode, fused=fused)
            filled_by.append(buf)
        assert torch.equal(filled_by[0], filled_by[1]), 'table-driven fill at %d^2 (fused=%s)' % (size, fused)
        del rf, ex, unfilled, filled_by


def test_one_launch_scatter_at_1024_equals_the_two_launches(K, dense_cloud):
    """kbe_render_frame_group_ahead at BASELINE's frame size: a sequence of 8-frame groups along a camera path, every tile launch
    also making the next group's placements, against the same groups with their placement launches in front (frames within the
    accumulation order); and the rule that keeps a cloud much denser than the raster on its placement launch."""
    from ken_burns_effect_amd import synthetic
    size = 1024
    image, disp = synthetic.make_rgbd(size, size, 3)
    depth = ((synthetic.FOCAL * synthetic.BASELINE) / (disp + 1e-7)).cuda()
    pts = K.depth_to_points(depth, synthetic.FOCAL).view(1, 3, -1)
    state = K.prepare_cloud(pts, image.cuda().reshape(1, 3, -1), depth.reshape(1, 1, -1), size, size, synthetic.FOCAL, raster=(size, size * size))
    K._pack(state)
    Bl = synthetic.BASELINE
    cams = [(synthetic.FOCAL * (1.0 + 0.002 * i), (0.9 * i - 10.0, 4.0 - 0.4 * i, -1.1 * i)) for i in range(24)]
    groups = [cams[0:8], cams[8:16], cams[16:24]]
    want = []
    buf = torch.zeros(8, size, size, 3, dtype=torch.uint8, device='cuda')
    for g in groups:
        K.render_frame_group_fused(state, g, Bl, buf)
        want.append(buf.clone())
    assert K.lib.kbe_render_frame_group_ahead_ok(state['N'], size, size, 8, 8) == 1
    for i, g in enumerate(groups):
        K.render_frame_group_ahead(state, g, Bl, buf, turn=i, placed=i > 0, next_cameras=groups[i + 1] if i + 1 < len(groups) else None)
        d = (buf.int() - want[i].int()).abs()
        assert int(d.max()) <= 1 and float((d > 0).float().mean()) < 1e-4, 'group %d: max %d, %.2e differ' % (i, int(d.max()), float((d > 0).float().mean()))
        assert float((buf == 0).all(dim=3).float().mean()) < 0.2, 'frames are rendered'
    # 16.8 M points on a 2048^2 raster: eight units of 64 points per wave of the tile launch -- they ride along in the launch's
    # dense form (k_frame_group_ahead_dense: test below); on a 1024^2 raster the same cloud would be 32 units per wave: too many
    assert K.lib.kbe_render_frame_group_ahead_ok(dense_cloud[1].shape[2], 2048, 2048, 4, 4) == 1
    assert K.lib.kbe_render_frame_group_ahead_ok(dense_cloud[1].shape[2], 1024, 1024, 4, 4) == 0
    assert K.lib.kbe_render_frame_group_ahead_ok(state['N'], size, size, 1, 12) == 0 and K.lib.kbe_render_frame_group_ahead_ok(state['N'], size, size, 8, 0) == 0


def test_one_launch_scatter_of_the_dense_cloud_equals_the_two_launches(K, dense_cloud, monkeypatch):
    """configs[4] pipelined: groups of frames of the 16.8 M-point cloud at 2048^2 whose tile launch also makes the next group's
    placements -- eight units of 64 points per wave, the dense form of the launch -- against the same groups with their placement
    launches in front."""
    monkeypatch.setenv('KBE_LANES', '1')
    from ken_burns_effect_amd import synthetic
    size, pts, img, dep = dense_cloud
    state = K.prepare_cloud(pts, img, dep, size, size, synthetic.FOCAL, raster=(size * 2, pts.shape[2]))
    K._pack(state)
    Bl = synthetic.BASELINE
    cams = [(synthetic.FOCAL * (1.0 + 0.004 * i), (2.0 * i - 5.0, 3.0 - 0.7 * i, -2.5 * i)) for i in range(6)]
    groups = [cams[0:2], cams[2:4], cams[4:6]]
    buf = torch.zeros(2, size, size, 3, dtype=torch.uint8, device='cuda')
    want = []
    for g in groups:
        K.render_frame_group_fused(state, g, Bl, buf)
        want.append(buf.clone())
    for i, g in enumerate(groups):
        K.render_frame_group_ahead(state, g, Bl, buf, turn=i, placed=i > 0, next_cameras=groups[i + 1] if i + 1 < len(groups) else None)
        d = (buf.int() - want[i].int()).abs()
        assert int(d.max()) <= 1 and float((d > 0).float().mean()) < 1e-4, 'group %d: max %d, %.2e differ' % (i, int(d.max()), float((d > 0).float().mean()))
        assert float((buf == 0).all(dim=3).float().mean()) < 0.2, 'frames are rendered'


def test_config4_2048_multi_pass_inpaint_then_frames_against_the_oracle(K, oracle):
    """BASELINE.json configs[4]'s "multi-pass inpaint" at 2048 x 2048: process_kenburns with boolInpaint=True (common.py:181-219 on a
    2048^2 image: two end poses, each through the 68-channel warp of kbe_render_pointcloud_tiled and the Inpaint network --
    seeded weights -- with the pixels that pose cannot see appended), then frames of the loop (:222-255) compared value for value
    with the oracle rendering the SAME grown cloud with the same cameras (Jacobi schedule, the product's)."""
    from ken_burns_effect_amd import common, synthetic
    from ken_burns_effect_amd.pointcloud_inpainting import Inpaint
    size = 2048
    image, disp = synthetic.make_rgbd(size, size, seed=5, colours='photo')
    depth = (synthetic.FOCAL * 2 * synthetic.BASELINE) / (disp + 1e-7)            # focal 2 F: the 1024^2 camera at twice the resolution
    focal = synthetic.FOCAL * 2
    oc = {'dblFocal': focal, 'dblBaseline': synthetic.BASELINE, 'intWidth': size, 'intHeight': size, 'dblDispmin': float(disp.min()), 'dblDispmax': float(disp.max()),
          'objectDepthrange': synthetic.depthrange_of(depth), 'tensorRawImage': image.cuda(), 'tensorRawDisparity': disp.cuda(), 'tensorRawDepth': depth.cuda()}
    oc['tensorRawPoints'] = K.depth_to_points(oc['tensorRawDepth'], focal).view(1, 3, -1)
    ofrom, oto = synthetic.default_windows(size, size, False)
    steps = [0.0, 0.45, 1.0]
    settings = {'dblSteps': steps, 'objectFrom': ofrom, 'objectTo': oto, 'boolInpaint': True, 'dolly': False, 'boolCrop': False}
    net = synthetic.seeded_fill_(Inpaint(), 3).cuda().eval()
    with torch.no_grad():
        frames = common.process_kenburns(settings, oc, net)
    n = oc['tensorInpaPoints'].shape[-1]
    assert len(frames) == len(steps) and frames[0].shape == (size, size, 3)
    assert size * size * 1.005 < n < size * size * 1.25, 'both passes appended points: %d' % n
    ok = oracle.OracleKernels(schedule='jacobi')
    state = ok.prepare_cloud(oc['tensorInpaPoints'].cpu(), oc['tensorInpaImage'].cpu(), oc['tensorInpaDepth'].cpu(), size, size)
    for k in (0, 2):
        focal_k, shift3 = common.frame_cameras(settings, oc)[k]
        ref = ok.render_frame(state, shift3, focal_k, oc['dblBaseline']).numpy()
        d = np.abs(frames[k].astype(np.int32) - ref.astype(np.int32))
        assert d.max() <= 1 and (d > 0).mean() < 1e-3, 'frame %d of the 2048^2 video: %d values differ from the oracle, max %d' % (k, int((d > 0).sum()), int(d.max()))
