import os
import sys

import numpy as np
import pytest

os.environ.setdefault('KBE_MIOPEN_FIND', '0')      # no MIOpen find step (20-50 s per process) inside the test suite
os.environ.setdefault('MIOPEN_FIND_MODE', 'FAST')   # ... and no solver search behind PyTorch's immediate mode either (the GPU suite: 45 s instead of 220)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + '.npz'))


@pytest.fixture(scope='session')
def oracle():
    """The CPU oracle front-end (test infrastructure)."""
    from oracle import kbe_oracle
    kbe_oracle.lib()
    return kbe_oracle


def bits(a):
    """fp32 array -> its bit patterns, so that comparisons are exact (and NaN-safe)."""
    return np.ascontiguousarray(np.asarray(a, dtype=np.float32)).view(np.uint32)


def assert_bits_equal(a, b, what=''):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    bad = int((bits(a) != bits(b)).sum())
    assert bad == 0, '%s: %d of %d elements differ bitwise (max abs %g)' % (
        what, bad, a.size, float(np.nanmax(np.abs(a.astype(np.float64) - b.astype(np.float64)))))


def at_size_scene(z, device, depth_to_points):
    """(settings, objectCommon) of a tests/golden/kenburns_at_size_*.npz fixture: the inputs are regenerated from its seed."""
    from ken_burns_effect_amd import synthetic
    H, W, dolly = int(z['H']), int(z['W']), bool(z['dolly'])
    image, disp = synthetic.make_rgbd(H, W, int(z['seed']), 'smooth', colours=str(z['colours']) if 'colours' in z else 'noise')
    depth = (512.0 * 120) / (disp + 1e-7)
    oc = {'dblFocal': 512.0, 'dblBaseline': 120, 'intWidth': W, 'intHeight': H, 'dblDispmin': disp.min().item(), 'dblDispmax': disp.max().item(),
          'objectDepthrange': synthetic.depthrange_of(depth), 'tensorRawImage': image.to(device), 'tensorRawDisparity': disp.to(device),
          'tensorRawDepth': depth.to(device)}
    oc['tensorRawPoints'] = depth_to_points(oc['tensorRawDepth'], 512.0).view(1, 3, -1)
    ofrom, oto = synthetic.default_windows(H, W, dolly)
    settings = {'dblSteps': [float(s) for s in z['steps']], 'objectFrom': ofrom, 'objectTo': oto, 'boolInpaint': True, 'dolly': dolly, 'boolCrop': False}
    return settings, oc


def psnr_u8(a, b):
    mse = float(np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2))
    return 200.0 if mse == 0 else 10.0 * np.log10(255.0 * 255.0 / mse)
