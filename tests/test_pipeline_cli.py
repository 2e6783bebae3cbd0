"""Pipeline front half + CLI plumbing on CPU (kernel set = oracle, injected; tiny image, seeded weights)."""
import numpy as np
import pytest
import torch


def test_cli_options_and_default_windows():
    from ken_burns_effect_amd import kbe
    cfg, window = kbe.parse(['--in', 'a.png', '--out', 'o', '--dolly', '--startU', '100', '--2d', '--write-frames'])
    assert cfg['in'] == 'a.png' and cfg['out'] == 'o' and cfg['dolly'] and cfg['2d'] and cfg['write-frames'] and window['startU'] == 100
    z = kbe.windows_for(1024, 768, dict.fromkeys(window), False)
    assert z['objectFrom'] == {'dblCenterU': 1024 / 2.15, 'dblCenterV': 768 / 2.15, 'intCropWidth': 921, 'intCropHeight': 691}
    assert z['objectTo']['intCropWidth'] == 870
    zd = kbe.windows_for(1024, 768, dict.fromkeys(window), True)
    assert zd['objectTo']['intCropWidth'] == 307 and zd['objectFrom']['dblCenterU'] == 512
    with pytest.raises(AssertionError):
        kbe.windows_for(100, 100, dict(startU=10, startV=50, startW=80, startH=80, endU=50, endV=50, endW=40, endH=40), False)


def test_pipeline_end_to_end_on_cpu(oracle, monkeypatch, tmp_path, recwarn):
    from ken_burns_effect_amd import common as C, kbe, synthetic
    from ken_burns_effect_amd.pipeline import Pipeline
    monkeypatch.setattr(C, '_kernel_set', oracle.OracleKernels('jacobi'))
    torch.set_num_threads(2)
    image, _ = synthetic.make_rgbd(64, 96, 5)
    pipe = Pipeline(model_paths=None, dolly=False, output_frames=True, device='cpu', steps=3)
    zoom = kbe.windows_for(96, 64, dict.fromkeys(('startU', 'startV', 'startW', 'startH', 'endU', 'endV', 'endW', 'endH')), False)
    frames = pipe(image, zoom, str(tmp_path))
    assert len(frames) == 3 and frames[0].shape == (64, 96, 3) and frames[0].dtype == np.uint8
    oc = pipe.objectCommon
    assert abs(oc['dblDispmax'] - 120) < 1e-3 and oc['dblDispmin'] >= 0                      # pipeline.py:79-81
    assert oc['tensorInpaPoints'].shape[2] >= 64 * 96                                         # inpainting appended points
    assert (tmp_path / 'frames' / '2.png').exists()
    assert (tmp_path / '3d_kbe.mp4').exists() or (tmp_path / '3d_kbe.npy').exists()
    assert any('seeded random weights' in str(w.message) for w in recwarn.list)
