"""The frame writers' native pieces on CPU (/root/reference/utils/pipeline.py:120-134): libkbe_jpeg.so -- the baseline
JPEG encoder the Motion-JPEG video writers use, a batch of frames on host threads (include/kbe_jpeg.h) -- against Pillow, and the PNG
writer (zlib directly).  Pillow is the CHECKER here (its decoder reads every stream; its encoder's tables and quality are the bar)."""
import ctypes
import io
import os
import re
import struct

import numpy as np
import pytest
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, 'ken-burns-effect_amd', 'csrc', 'libkbe_jpeg.so')


@pytest.fixture(scope='module')
def lib():
    assert os.path.exists(LIB), 'libkbe_jpeg.so is not built: python -c "import __graft_entry__ as g; g.build()"'
    so = ctypes.CDLL(LIB)
    so.kbe_jpeg_bound.restype = ctypes.c_size_t
    so.kbe_jpeg_bound.argtypes = [ctypes.c_int, ctypes.c_int]
    return so


def encode(lib, frame, quality=92, cap=None):
    frame = np.ascontiguousarray(frame, dtype=np.uint8)
    h, w = frame.shape[:2]
    cap = int(lib.kbe_jpeg_bound(w, h)) if cap is None else cap
    out = np.empty(max(cap, 1), np.uint8)
    size = ctypes.c_size_t(0)
    rc = lib.kbe_jpeg_encode(ctypes.c_void_p(frame.ctypes.data), w, h, 3 * w, quality, ctypes.c_void_p(out.ctypes.data), ctypes.c_size_t(cap), ctypes.byref(size))
    return rc, out[:size.value].tobytes()


def pillow(frame, quality=92):
    buf = io.BytesIO()
    Image.fromarray(np.ascontiguousarray(frame)).save(buf, format='JPEG', quality=quality)
    return buf.getvalue()


def decode(data):
    return np.asarray(Image.open(io.BytesIO(data)).convert('RGB'))


def psnr(a, b):
    return 10.0 * np.log10(255.0 ** 2 / max(np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2), 1e-12))


def segments(data):
    """{marker: [payloads]} of a JPEG's header segments up to the start of scan."""
    out, i = {}, 2
    assert data[:2] == b'\xff\xd8'
    while i < len(data):
        assert data[i] == 0xFF
        marker = data[i + 1]
        n = struct.unpack('>H', data[i + 2:i + 4])[0]
        out.setdefault(marker, []).append(data[i + 4:i + 2 + n])
        i += 2 + n
        if marker == 0xDA:
            break
    return out


def tables(data):
    seg = segments(data)
    dqt, dht = {}, {}
    for body in seg.get(0xDB, []):
        while body:
            assert body[0] >> 4 == 0                    # 8-bit entries
            dqt[body[0] & 15] = body[1:65]
            body = body[65:]
    for body in seg.get(0xC4, []):
        while body:
            n = sum(body[1:17])
            dht[body[0]] = body[1:17 + n]
            body = body[17 + n:]
    return dqt, dht, seg[0xC0][0]


def photo_like(h, w, seed):
    g = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.stack([128 + 100 * np.sin(xx / 23.0 + seed) * np.cos(yy / 31.0), 128 + 90 * np.sin((xx + yy) / 41.0), 255.0 * xx / max(w - 1, 1)], -1)
    img[h // 4:h // 2, w // 3:2 * w // 3] = (220, 40, 60)                       # an edge or two
    return np.clip(img + g.normal(0, 4, img.shape), 0, 255).astype(np.uint8)


def test_the_library_exports_what_its_header_declares(lib):
    header = open(os.path.join(ROOT, 'include', 'kbe_jpeg.h')).read()
    declared = re.findall(r'KBE_JPEG_API\s+[\w\s\*]+?\b(kbe_jpeg_\w+)\s*\(', header)
    assert sorted(declared) == ['kbe_jpeg_bound', 'kbe_jpeg_encode', 'kbe_jpeg_encode_batch']
    for name in declared:
        assert hasattr(lib, name), name


@pytest.mark.parametrize('quality', [10, 50, 75, 92, 100])
def test_quantisation_and_huffman_tables_are_pillows(lib, quality):
    """Annex K.1 tables scaled by the IJG quality rule, Annex K.3 Huffman tables, 4:2:0: segment for segment what Pillow writes by default."""
    frame = photo_like(48, 64, 1)
    rc, mine = encode(lib, frame, quality)
    assert rc == 0
    q_mine, h_mine, sof_mine = tables(mine)
    q_pil, h_pil, sof_pil = tables(pillow(frame, quality))
    assert q_mine == q_pil and set(q_mine) == {0, 1}
    assert h_mine == h_pil and set(h_mine) == {0x00, 0x10, 0x01, 0x11}
    assert sof_mine == sof_pil                              # 8 bits, the size, Y 2x2 / Cb 1x1 / Cr 1x1 with tables 0 / 1 / 1


@pytest.mark.parametrize('size', [(96, 128), (50, 37), (17, 16), (16, 17), (1, 1), (3, 200)])
def test_streams_decode_to_the_picture_as_well_as_pillows_do(lib, size):
    h, w = size
    frame = photo_like(h, w, 3)
    rc, mine = encode(lib, frame, 92)
    assert rc == 0 and mine[:2] == b'\xff\xd8' and mine[-2:] == b'\xff\xd9'
    got = decode(mine)
    assert got.shape == (h, w, 3)
    ours, theirs = psnr(got, frame), psnr(decode(pillow(frame, 92)), frame)
    assert ours > theirs - 0.5, 'this encoder %.2f dB, Pillow %.2f dB' % (ours, theirs)
    if h * w >= 1000:
        assert abs(len(mine) - len(pillow(frame, 92))) < 0.05 * len(mine) + 64


def test_noise_and_flat_frames(lib):
    g = np.random.default_rng(5)
    noise = g.integers(0, 256, (64, 80, 3), dtype=np.uint8)                         # every coefficient non-zero, 0xFF bytes in the stream: stuffing
    rc, data = encode(lib, noise, 100)
    assert rc == 0 and psnr(decode(data), noise) > psnr(decode(pillow(noise, 100)), noise) - 0.5
    assert b'\xff\x00' in data[data.index(b'\xff\xda'):]
    for value in (0, 255, 128):
        flat = np.full((40, 40, 3), value, np.uint8)
        rc, data = encode(lib, flat, 92)
        assert rc == 0 and np.abs(decode(data).astype(int) - value).max() <= 1


def test_a_batch_on_threads_equals_the_frames_one_by_one(lib):
    frames = [photo_like(72, 88, s) for s in range(9)]
    n, (h, w) = len(frames), frames[0].shape[:2]
    cap = int(lib.kbe_jpeg_bound(w, h))
    outs = [np.empty(cap, np.uint8) for _ in range(n)]
    sizes = (ctypes.c_size_t * n)()
    for threads in (1, 4, 64):
        rc = lib.kbe_jpeg_encode_batch((ctypes.c_void_p * n)(*[f.ctypes.data for f in frames]), n, w, h, 3 * w, 92, (ctypes.c_void_p * n)(*[o.ctypes.data for o in outs]),
                                       ctypes.c_size_t(cap), sizes, threads)
        assert rc == 0
        for i, f in enumerate(frames):
            assert outs[i][:sizes[i]].tobytes() == encode(lib, f, 92)[1], 'frame %d on %d threads' % (i, threads)
    assert lib.kbe_jpeg_encode_batch(None, 0, w, h, 3 * w, 92, None, ctypes.c_size_t(cap), None, 4) == 0      # an empty batch is no error


def test_errors_are_codes(lib):
    frame = photo_like(32, 32, 2)
    assert encode(lib, frame, 92, cap=100)[0] == -2                                 # KBE_JPEG_E_SPACE
    size = ctypes.c_size_t(0)
    out = np.empty(4096, np.uint8)
    assert lib.kbe_jpeg_encode(None, 32, 32, 96, 92, ctypes.c_void_p(out.ctypes.data), ctypes.c_size_t(4096), ctypes.byref(size)) == -1
    assert lib.kbe_jpeg_encode(ctypes.c_void_p(frame.ctypes.data), 32, 32, 95, 92, ctypes.c_void_p(out.ctypes.data), ctypes.c_size_t(4096), ctypes.byref(size)) == -1      # stride < 3 w
    assert lib.kbe_jpeg_encode(ctypes.c_void_p(frame.ctypes.data), 0, 32, 96, 92, ctypes.c_void_p(out.ctypes.data), ctypes.c_size_t(4096), ctypes.byref(size)) == -1


def test_the_video_writers_encode_each_distinct_frame_once_on_the_native_encoder(lib, monkeypatch, tmp_path):
    import sys
    sys.path.insert(0, ROOT)
    from ken_burns_effect_amd import pipeline
    assert pipeline.jpeg_encoder()[0] == 'native'
    frames = [photo_like(64, 96, s) for s in range(5)]
    video = frames + frames[-2::-1]                                                 # forth and back: the same objects again
    jpegs = pipeline._jpegs(video, 92)
    assert len(jpegs) == 9 and [jpegs.index(j) for j in jpegs] == [0, 1, 2, 3, 4, 3, 2, 1, 0]
    assert all(jpegs[i] == encode(lib, frames[i], 92)[1] for i in range(5))
    # non-contiguous frames (the BGR -> RGB view of pipeline._run) go through as well
    flipped = [f[:, :, ::-1] for f in frames]
    assert pipeline._jpegs(flipped, 92)[2] == encode(lib, np.ascontiguousarray(flipped[2]), 92)[1]
    # ... and KBE_JPEG=pillow is Pillow's stream
    monkeypatch.setenv('KBE_JPEG', 'pillow')
    assert pipeline._jpegs(frames[:1], 92)[0] == pillow(frames[0], 92)
    monkeypatch.delenv('KBE_JPEG')
    path = pipeline.write_mjpeg_mp4(str(tmp_path / 'v.mp4'), video, fps=25)
    data = open(path, 'rb').read()
    assert data.count(b'\xff\xd8\xff\xe0') == 9                                     # nine samples, each a JFIF stream


@pytest.mark.parametrize('size', [(40, 56), (1, 1), (7, 3), (33, 129)])
def test_png_frames_are_lossless(size, tmp_path):
    import sys
    sys.path.insert(0, ROOT)
    from ken_burns_effect_amd import pipeline
    h, w = size
    g = np.random.default_rng(7)
    for frame in (photo_like(h, w, 4), g.integers(0, 256, (h, w, 3), dtype=np.uint8), photo_like(h, w, 5)[:, :, ::-1]):
        data = pipeline.png_bytes(frame)
        assert np.array_equal(np.asarray(Image.open(io.BytesIO(data)).convert('RGB')), frame)
    frames = [photo_like(h, w, s) for s in range(6)]
    pipeline.write_frames(str(tmp_path / 'frames'), frames)
    for i, f in enumerate(frames):
        assert np.array_equal(np.asarray(Image.open(str(tmp_path / 'frames' / ('%d.png' % i))).convert('RGB')), f)
