"""Pins the CPU oracle to the golden vectors made from the reference (tests/golden/make_golden.py).

Everything here is bit-exact unless a tolerance is written next to the assertion.
"""
import numpy as np
import pytest
import torch

from conftest import assert_bits_equal, load_golden

RENDER_CASES = ['render_f512', 'render_f409', 'render_f153', 'render_noise', 'render_b2c7']


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def _baseline(z):
    b = float(z['baseline'])
    return int(b) if bool(z['baseline_is_int']) else b


@pytest.mark.parametrize('case', RENDER_CASES)
@pytest.mark.parametrize('mode', ['fma', 'nofma'])
def test_render_stages_match_reference(oracle, case, mode):
    z = load_golden(case)
    pts, data = _t(z['points']), _t(z['data'])
    W, H, F, Bl = int(z['W']), int(z['H']), float(z['focal']), _baseline(z)
    fma = mode == 'fma'
    zee, winner = oracle.zsplat(pts, W, H, F, Bl, use_fma=fma, want_winner=True)
    assert_bits_equal(zee.numpy(), z['zee_pre_' + mode], 'pre-degrid z-buffer')
    if 'winner_' + mode in z.files:
        assert np.array_equal(winner.numpy(), z['winner_' + mode]), 'per-point winner pixel index'
    zs = oracle.degrid(zee, 'serial')
    assert_bits_equal(zs.numpy(), z['zee_serial_' + mode], 'degrid, serial schedule')
    acc = oracle.accumulate(pts, data, zs, F, Bl, use_fma=fma)
    assert_bits_equal(acc.numpy(), z['acc_' + mode], 'accumulated output')
    if fma:
        render, existing = oracle.normalize(acc)
        assert_bits_equal(render.numpy(), z['render_fma'], 'normalised render')
        assert_bits_equal(existing.numpy(), z['existing_fma'], 'existing')
        r2, e2 = oracle.render_pointcloud(pts, data, W, H, F, Bl, schedule='serial', use_fma=True)
        assert_bits_equal(r2.numpy(), z['render_fma'], 'render_pointcloud')


def test_winner_is_the_pixel_that_gets_the_min(oracle):
    z = load_golden('render_f512')
    pts = _t(z['points'])
    zee, winner = oracle.zsplat(pts, int(z['W']), int(z['H']), float(z['focal']), _baseline(z), want_winner=True)
    w = winner.numpy()[0]
    touched = np.zeros(zee.numel(), bool)
    touched[w[w >= 0]] = True
    assert np.array_equal(touched, (zee.numpy().reshape(-1) != np.float32(1e6)))
    assert (zee.numpy() < 0).sum() > 0, 'fixture must exercise negative dblError (SURVEY B.2)'


def test_jacobi_differs_from_serial_only_where_cascades_happen(oracle):
    z = load_golden('render_noise')
    zee = _t(z['zee_pre_fma'])
    jac, ser = oracle.degrid(zee, 'jacobi'), oracle.degrid(zee, 'serial')
    # both schedules only ever lower a value, never below the neighbourhood mean
    assert (jac <= zee).all() and (ser <= zee).all()
    assert (jac != ser).sum() > 0     # the white-noise scene is the cascade stress case
    # on pixels whose 3x3 neighbourhood the serial sweep has not modified yet, both agree
    assert_bits_equal(jac[0, 0, 0, :2].numpy(), ser[0, 0, 0, :2].numpy(), 'first pixels')


def test_fill_matches_reference(oracle):
    z = load_golden('fill')
    for tag in ('a', 'b', 'allholes'):
        out = oracle.fill_disocclusion(_t(z['input_' + tag]), _t(z['depth_' + tag]))
        assert_bits_equal(out.numpy(), z['output_' + tag], 'fill ' + tag)
    out = oracle.fill_disocclusion(_t(z['input_allholes']), _t(z['depth_allholes']) + 1.0)
    assert_bits_equal(out.numpy(), z['output_noholes'], 'fill noholes')
    assert_bits_equal(z['output_allholes'], z['input_allholes'], 'unfillable pixels keep their value')


def test_depth_to_points_matches_reference(oracle):
    z = load_golden('torch_helpers')
    for tag in 'abc':
        out = oracle.depth_to_points(_t(z['d2p_depth_' + tag]), float(z['d2p_focal_' + tag]))
        assert_bits_equal(out.numpy(), z['d2p_points_' + tag], 'depth_to_points ' + tag)


def test_shift_points_matches_reference(oracle):
    z = load_golden('torch_helpers')
    pts = _t(z['ps_points'])
    for i in range(3):
        out = oracle.shift_points(pts, _t(z['ps_shift_%d' % i]))
        assert_bits_equal(out.numpy(), z['ps_out_%d' % i], 'process_shift points %d' % i)


def test_median_matches_reference(oracle):
    z = load_golden('torch_helpers')
    for tag in 'ab':
        for kind in ('median-3', 'median-5'):
            out = oracle.spatial_filter(_t(z['sf_input_' + tag]), kind)
            assert_bits_equal(out.numpy(), z['sf_%s_%s' % (kind, tag)], kind + tag)
    out = oracle.spatial_filter(_t(z['sf_input_mask']), 'median-5')
    assert_bits_equal(out.numpy(), z['sf_median-5_mask'], 'median-5 on a binary mask')


def test_laplacian_matches_reference_to_rounding(oracle):
    # torch's conv2d summation order is unspecified; ours is fixed (row-major taps, fmaf).
    # tolerance: 4 ulp of the largest tap product.
    z = load_golden('torch_helpers')
    for tag in 'ab':
        x = z['sf_input_' + tag]
        out = oracle.spatial_filter(_t(x), 'laplacian').numpy()
        tol = 4 * np.finfo(np.float32).eps * 4.0 * np.abs(x).max()
        assert np.abs(out - z['sf_laplacian_' + tag]).max() <= tol
    disp = _t(z['sf_input_disp'])
    valid = (oracle.spatial_filter(disp / disp.max(), 'laplacian').abs() < 0.03).float()
    assert (valid.numpy() != z['sf_valid_disp']).mean() <= 0.002   # threshold-borderline pixels only


def test_pconv_epilogue_matches_reference(oracle):
    z = load_golden('partial_conv')
    for tag in 'abc':
        cin, cout, k, s, p = [int(v) for v in z['cfg_' + tag]]
        x, m = _t(z['x_' + tag]), _t(z['m_' + tag])
        raw = torch.nn.functional.conv2d(x * m, _t(z['w_' + tag]), _t(z['b_' + tag]), stride=s, padding=p)
        out, um = oracle.pconv_epilogue(raw, _t(z['b_' + tag]), m, k, s, p)
        assert_bits_equal(um.expand(-1, cout, -1, -1).numpy(), z['mask_' + tag], 'update_mask ' + tag)
        # raw comes from this machine's conv2d; the epilogue itself is exact given raw
        assert np.abs(out.numpy() - z['out_' + tag]).max() <= 1e-5 * max(1.0, np.abs(z['out_' + tag]).max())


@pytest.mark.parametrize('tag', ['a', 'b', 'c'])
@pytest.mark.parametrize('mode', ['fma', 'nofma'])
def test_generate_mask_matches_reference(oracle, tag, mode):
    """generate_mask (common.py:689-830): z-buffer, owner table, per-point mask and the median-filtered mask
    against the reference's own kernel run in serial point order (tests/golden/make_golden.py)."""
    z = load_golden('generate_mask')
    pts, shift = torch.from_numpy(z[tag + '_points']), torch.from_numpy(z[tag + '_shift'])
    W, H = int(z[tag + '_W']), int(z[tag + '_H'])
    baseline = int(z[tag + '_baseline']) if bool(z[tag + '_baseline_is_int']) else float(z[tag + '_baseline'])
    masks, zee, ids = oracle.generate_mask_raw(pts, shift, W, H, float(z[tag + '_focal']), baseline, use_fma=(mode == 'fma'))
    assert_bits_equal(zee.numpy(), z[tag + '_zee_' + mode], 'z-buffer')
    assert np.array_equal(ids.numpy(), z[tag + '_ids_' + mode]), 'owner table'
    assert np.array_equal(masks.numpy(), z[tag + '_masks_raw_' + mode]), 'per-point mask'
    out = oracle.generate_mask(pts, shift, W, H, float(z[tag + '_focal']), baseline, use_fma=(mode == 'fma'))
    assert np.array_equal(out.numpy(), z[tag + '_masks_' + mode]), 'median-5 of the mask image'
