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
    pipe = Pipeline(model_paths=None, allow_random_weights=True, dolly=False, output_frames=True, device='cpu', steps=3)
    zoom = kbe.windows_for(96, 64, dict.fromkeys(('startU', 'startV', 'startW', 'startH', 'endU', 'endV', 'endW', 'endH')), False)
    frames = pipe(image, zoom, str(tmp_path))
    assert len(frames) == 3 and frames[0].shape == (64, 96, 3) and frames[0].dtype == np.uint8
    oc = pipe.objectCommon
    assert abs(oc['dblDispmax'] - 120) < 1e-3 and oc['dblDispmin'] >= 0                      # pipeline.py:79-81
    assert oc['tensorInpaPoints'].shape[2] >= 64 * 96                                         # inpainting appended points
    assert (tmp_path / 'frames' / '2.png').exists()
    assert (tmp_path / '3d_kbe.mp4').exists()
    assert any('seeded random weights' in str(w.message) for w in recwarn.list)
    assert any('semantics (VGG19-bn)' in str(w.message) for w in recwarn.list)


def test_semantics_weights_come_from_a_file_when_given(tmp_path, recwarn):
    """the VGG19-bn weights must be loadable (torchvision state-dict layout), and only their absence may fall
    back to seeded weights -- loudly."""
    from ken_burns_effect_amd import kbe
    from ken_burns_effect_amd.disparity_estimation import Semantics
    from ken_burns_effect_amd.pipeline import Pipeline
    ref = Semantics()
    state = {'features.' + k.split('.', 2)[2]: torch.full_like(v, 0.125) if v.is_floating_point() else v for k, v in ref.state_dict().items()}
    path = str(tmp_path / 'vgg19_bn.pth')
    torch.save(state, path)
    pipe = Pipeline(model_paths=None, allow_random_weights=True, device='cpu', steps=2, semantics_path=path)
    got = pipe.moduleSemantics.state_dict()
    assert all(bool((v == 0.125).all()) for v in got.values() if v.is_floating_point())
    assert not any('semantics (VGG19-bn)' in str(w.message) for w in recwarn.list)
    cfg, _ = kbe.parse(['--semantics-path', path])
    assert cfg['semantics-path'] == path


def test_writers_frame_order_and_channel_flips(tmp_path, monkeypatch):
    """pipeline.py:120-134 of the reference: PNG frames are the BGR->RGB flip of what process_kenburns returned (cv2.imwrite
    of BGR data), unless --pretrained-estim (RGB input: no flip); the video holds frames + reversed(frames)[1:]."""
    from PIL import Image
    from ken_burns_effect_amd import pipeline as P
    frames = [np.full((4, 6, 3), i, np.uint8) for i in range(4)]
    for f in frames:
        f[..., 0] += 100                                            # channel 0 is distinguishable
    seen = {}
    monkeypatch.setattr(P, 'write_video', lambda path, fr, fps=25: seen.update(path=path, frames=[f.copy() for f in fr], fps=fps))

    class Stub(P.Pipeline):
        def __init__(self):
            self.output_frames, self.dolly, self.steps, self.objectCommon, self.moduleInpaint = True, False, 4, {}, None

        def estimate(self, tensorImage):
            return self.objectCommon

    monkeypatch.setattr(P.common, 'process_kenburns', lambda *a, **k: [f.copy() for f in frames])
    for pretrained_estim in (False, True):
        out = tmp_path / ('rgb' if pretrained_estim else 'bgr')
        Stub()(None, {'objectFrom': {}, 'objectTo': {}}, str(out), pretrained_estim=pretrained_estim)
        assert [int(f[0, 0, 0 if pretrained_estim else 2]) for f in seen['frames']] == [100, 101, 102, 103, 102, 101, 100] and seen['fps'] == 25
        png = np.asarray(Image.open(out / 'frames' / '3.png'))
        assert png.shape == (4, 6, 3) and int(png[0, 0, 0 if pretrained_estim else 2]) == 103 and int(png[0, 0, 1]) == 3


def test_missing_checkpoints_fail_unless_random_weights_are_allowed(tmp_path, monkeypatch):
    """the reference fails on a missing checkpoint (torch.load, utils.py:206); so does this package, for the three
    network checkpoints and for the VGG weights, unless the caller asks for seeded weights."""
    from ken_burns_effect_amd import kbe
    from ken_burns_effect_amd.pipeline import Pipeline
    monkeypatch.delenv('KBE_ALLOW_RANDOM_WEIGHTS', raising=False)
    monkeypatch.delenv('KBE_SEMANTICS_PATH', raising=False)
    with pytest.raises(FileNotFoundError, match='disparity'):
        Pipeline(model_paths=[str(tmp_path / 'nope.tar')] * 3, device='cpu', steps=2)
    with pytest.raises(FileNotFoundError):
        Pipeline(model_paths=None, device='cpu', steps=2)
    cfg, _ = kbe.parse(['--allow-random-weights'])
    assert cfg['allow-random-weights'] is True and kbe.parse([])[0]['allow-random-weights'] is False
    monkeypatch.setenv('KBE_ALLOW_RANDOM_WEIGHTS', '1')
    with pytest.warns(UserWarning, match='seeded random weights'):
        Pipeline(model_paths=None, device='cpu', steps=2)


def test_miopen_find_runs_once_per_machine_and_image_size(tmp_path, monkeypatch, capsys):
    """Pipeline(miopen_find='auto'), the default: the first call for an image size this machine has not tuned runs under MIOpen's
    find step (torch.backends.cudnn.benchmark) and leaves a marker; later calls -- and other sizes' markers -- do not switch it on
    again.  (Stubbed: no network runs here; the modes' timings are in profiles/r04_networks.txt.)"""
    from ken_burns_effect_amd import pipeline as P
    monkeypatch.setenv('KBE_CACHE_DIR', str(tmp_path))
    monkeypatch.delenv('KBE_MIOPEN_FIND', raising=False)
    monkeypatch.delenv('MIOPEN_DISABLE_CACHE', raising=False)
    db = tmp_path / 'userdb'
    db.mkdir()
    monkeypatch.setenv('MIOPEN_USER_DB_PATH', str(db))
    seen = []
    keeps = [True]          # does "MIOpen" keep what its find step measured?  (a read-only / disabled find-db keeps nothing)

    class Stub(P.Pipeline):
        def __init__(self):
            self.output_frames, self.dolly, self.steps, self.objectCommon, self.moduleInpaint = False, False, 2, {}, None
            self.partial_inpainting, self.moduleRefine, self.miopen_find = False, None, 'auto'
            self.device = torch.device('cuda:0')

        def tuning_tag(self, width, height):
            return '%dx%d' % (width, height)

        def estimate(self, tensorImage):
            seen.append(torch.backends.cudnn.benchmark)
            if torch.backends.cudnn.benchmark and keeps[0]:
                with open(db / 'gfx950.ufdb.txt', 'a') as f:
                    f.write('a solver measured\n')
            return self.objectCommon

    monkeypatch.setattr(torch.cuda, 'is_available', lambda: True)
    monkeypatch.setattr(torch.cuda, 'get_device_name', lambda *a: 'AMD Instinct MI355X')
    monkeypatch.setattr(P.common, 'process_kenburns', lambda *a, **k: [])
    monkeypatch.setattr(P.common, 'on_device_of', lambda *_: __import__('contextlib').nullcontext())
    before = torch.backends.cudnn.benchmark
    pipe = Stub()
    zoom = {'objectFrom': {}, 'objectTo': {}}
    pipe(torch.zeros(1, 3, 64, 96), zoom)
    pipe(torch.zeros(1, 3, 64, 96), zoom)
    pipe(torch.zeros(1, 3, 32, 48), zoom)
    assert seen == [True, False, True] and torch.backends.cudnn.benchmark == before
    import glob
    assert len(glob.glob(str(tmp_path / 'miopen-tuned' / '96x64-AMD_Instinct_MI355X-*'))) == 1 and len(glob.glob(str(tmp_path / 'miopen-tuned' / '*'))) == 2
    assert 'measures its convolution solvers once' in capsys.readouterr().err
    # a find-db that keeps nothing (read-only, MIOPEN_DISABLE_CACHE ...): the size is NOT marked tuned -- marking it would leave it
    # untuned for good -- and the next call measures again
    keeps[0] = False
    del seen[:]
    pipe(torch.zeros(1, 3, 16, 24), zoom)
    pipe(torch.zeros(1, 3, 16, 24), zoom)
    assert seen == [True, True] and len(glob.glob(str(tmp_path / 'miopen-tuned' / '*'))) == 2
    assert 'NOT marked tuned' in capsys.readouterr().err
    keeps[0] = True
    pipe(torch.zeros(1, 3, 16, 24), zoom)
    pipe(torch.zeros(1, 3, 16, 24), zoom)
    assert seen[2:] == [True, False] and len(glob.glob(str(tmp_path / 'miopen-tuned' / '*'))) == 3
    # explicit settings: never / always
    assert P.Pipeline.__init__.__defaults__ is not None
    monkeypatch.setenv('KBE_MIOPEN_FIND', '0')
    real = P.Pipeline(model_paths=None, allow_random_weights=True, device='cpu', steps=2)
    assert real.miopen_find is False and real.tuning_tag(96, 64) == '96x64-plain'


def _boxes(blob, start, end):
    """The boxes of an ISO base media file between two offsets: (type, body offset, body end)."""
    import struct
    out, pos = [], start
    while pos < end:
        size, tag = struct.unpack('>I4s', blob[pos:pos + 8])
        assert size >= 8 and pos + size <= end, (tag, size)
        out.append((tag, pos + 8, pos + size))
        pos += size
    assert pos == end
    return out


def _child(blob, boxes, tag):
    hit = [b for b in boxes if b[0] == tag]
    assert len(hit) == 1, (tag, [b[0] for b in boxes])
    return hit[0]


def test_video_without_ffmpeg_is_an_mp4_of_motion_jpeg_that_decodes_back(tmp_path, monkeypatch):
    """No ffmpeg binary (either image): write_video still leaves the .mp4 it was asked for -- an ISO base media file (ftyp, mdat,
    moov) with one video track of JPEG samples.  Walked box by box here: the sample table (stsz sizes, the one chunk's stco offset,
    stts, stsc) must lead to every frame, which must decode (PIL) to the frame written, in order, within JPEG's loss; the track and
    the sample entry carry the size; the esds names JPEG (objectTypeIndication 0x6C); durations are frames / fps."""
    import io
    import struct
    from PIL import Image
    from ken_burns_effect_amd import pipeline as P
    monkeypatch.setattr(P.shutil, 'which', lambda name: None)
    yy, xx = np.mgrid[0:48, 0:64]
    frames = [np.stack([(xx * 3 + 10 * i) % 256, (yy * 4) % 256, np.full_like(xx, 40 * i)], axis=2).astype(np.uint8) for i in range(5)]
    assert P.write_video(str(tmp_path / '3d_kbe.mp4'), frames, fps=25) is False
    blob = (tmp_path / '3d_kbe.mp4').read_bytes()
    top = _boxes(blob, 0, len(blob))
    assert [b[0] for b in top] == [b'ftyp', b'mdat', b'moov'] and blob[8:12] == b'isom'
    moov = _boxes(blob, *_child(blob, top, b'moov')[1:])
    mvhd = _child(blob, moov, b'mvhd')
    timescale, duration = struct.unpack('>II', blob[mvhd[1] + 12:mvhd[1] + 20])
    assert duration / timescale == 5 / 25
    trak = _boxes(blob, *_child(blob, moov, b'trak')[1:])
    tkhd = _child(blob, trak, b'tkhd')
    assert tkhd[2] - tkhd[1] == 84 and struct.unpack('>II', blob[tkhd[2] - 8:tkhd[2]]) == (64 << 16, 48 << 16)
    mdia = _boxes(blob, *_child(blob, trak, b'mdia')[1:])
    hdlr = _child(blob, mdia, b'hdlr')
    assert blob[hdlr[1] + 8:hdlr[1] + 12] == b'vide'
    mdhd = _child(blob, mdia, b'mdhd')
    assert struct.unpack('>II', blob[mdhd[1] + 12:mdhd[1] + 20]) == (timescale, duration)
    minf = _boxes(blob, *_child(blob, mdia, b'minf')[1:])
    assert {b[0] for b in minf} == {b'vmhd', b'dinf', b'stbl'}
    stbl = _boxes(blob, *_child(blob, minf, b'stbl')[1:])
    stsd = _child(blob, stbl, b'stsd')
    assert struct.unpack('>I', blob[stsd[1] + 4:stsd[1] + 8])[0] == 1
    entry = _boxes(blob, stsd[1] + 8, stsd[2])
    assert [b[0] for b in entry] == [b'mp4v']
    e0 = entry[0][1]
    assert struct.unpack('>HH', blob[e0 + 24:e0 + 28]) == (64, 48) and struct.unpack('>H', blob[e0 + 74:e0 + 76])[0] == 24
    esds = _boxes(blob, e0 + 78, entry[0][2])
    assert [b[0] for b in esds] == [b'esds']
    d = esds[0][1] + 4
    assert blob[d] == 0x03 and blob[d + 5] == 0x04 and blob[d + 7] == 0x6C and blob[d + 8] == 0x11         # ES, DecoderConfig: JPEG, visual stream
    stts = _child(blob, stbl, b'stts')
    assert struct.unpack('>III', blob[stts[1] + 4:stts[1] + 16]) == (1, 5, duration // 5)
    stsc = _child(blob, stbl, b'stsc')
    assert struct.unpack('>IIII', blob[stsc[1] + 4:stsc[1] + 20]) == (1, 1, 5, 1)
    stsz = _child(blob, stbl, b'stsz')
    uniform, count = struct.unpack('>II', blob[stsz[1] + 4:stsz[1] + 12])
    sizes = struct.unpack('>5I', blob[stsz[1] + 12:stsz[1] + 32])
    stco = _child(blob, stbl, b'stco')
    n_chunks, offset = struct.unpack('>II', blob[stco[1] + 4:stco[1] + 12])
    mdat = _child(blob, top, b'mdat')
    assert (uniform, count, n_chunks) == (0, 5, 1) and offset == mdat[1] and offset + sum(sizes) == mdat[2]
    assert not [b for b in stbl if b[0] == b'stss']                 # every sample a sync sample
    for want, size in zip(frames, sizes):
        sample = blob[offset:offset + size]
        assert sample[:2] == b'\xff\xd8' and sample[-2:] == b'\xff\xd9'
        got = np.asarray(Image.open(io.BytesIO(sample)).convert('RGB'))
        assert got.shape == want.shape and np.abs(got.astype(np.int32) - want.astype(np.int32)).mean() < 6.0
        offset += size


def test_an_avi_without_ffmpeg_is_motion_jpeg_that_decodes_back(tmp_path, monkeypatch):
    """No ffmpeg binary and an .avi asked for: a RIFF file whose frames decode back (PIL) to the frames written, in order, within
    JPEG's loss; its header carries the size, the rate and the frame count."""
    import io
    import struct
    from PIL import Image
    from ken_burns_effect_amd import pipeline as P
    monkeypatch.setattr(P.shutil, 'which', lambda name: None)
    yy, xx = np.mgrid[0:48, 0:64]
    frames = [np.stack([(xx * 3 + 10 * i) % 256, (yy * 4) % 256, np.full_like(xx, 40 * i)], axis=2).astype(np.uint8) for i in range(5)]
    assert P.write_video(str(tmp_path / '3d_kbe.avi'), frames, fps=25) is False
    blob = (tmp_path / '3d_kbe.avi').read_bytes()
    assert blob[:4] == b'RIFF' and blob[8:12] == b'AVI ' and struct.unpack('<I', blob[4:8])[0] == len(blob) - 8
    avih = blob.index(b'avih') + 8
    us_per_frame, _, _, flags, total = struct.unpack('<5I', blob[avih:avih + 20])
    width, height = struct.unpack('<2I', blob[avih + 32:avih + 40])
    assert (us_per_frame, total, width, height) == (40000, 5, 64, 48) and flags & 0x10
    pos, decoded = blob.index(b'movi') + 4, []
    while blob[pos:pos + 4] == b'00dc':
        size = struct.unpack('<I', blob[pos + 4:pos + 8])[0]
        decoded.append(np.asarray(Image.open(io.BytesIO(blob[pos + 8:pos + 8 + size])).convert('RGB')))
        pos += 8 + size + (size & 1)
    assert len(decoded) == 5 and blob[pos:pos + 4] == b'idx1'
    for want, got in zip(frames, decoded):
        assert got.shape == want.shape and np.abs(got.astype(np.int32) - want.astype(np.int32)).mean() < 6.0
