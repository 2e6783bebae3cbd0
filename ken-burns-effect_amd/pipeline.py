"""Image -> disparity -> point cloud -> Ken Burns frames.

Drop-in for ``/root/reference/utils/pipeline.py`` (``Pipeline`` :23-134): same constructor and
``__call__`` arguments.  The front half (:61-100: resize, Semantics + Disparity, Refine, disparity
normalisation, depth, unprojection, ``objectCommon``) runs as stock PyTorch-ROCm modules plus the
HIP unprojection; the frame loop is :func:`ken_burns_effect_amd.common.process_kenburns`.

Differences, on purpose: the unused Mask-RCNN of the reference (:36, deleted at :90) is not built; the ImageNet
weights of ``Semantics`` come from a file (``semantics_path``) instead of a torchvision download;
``cv2.minMaxLoc`` is replaced by :func:`synthetic.depthrange_of`; frames are written with PIL and the
video through an ``ffmpeg`` binary if one is on PATH (OpenCV / moviepy are not dependencies) --
otherwise the ``.mp4`` holds Motion-JPEG (ISO base media file written here, no external encoder).  Returns the frame list.
"""
import os
import shutil
import subprocess

import numpy as np
import torch

from . import common, synthetic
from .disparity_estimation import Disparity, Semantics
from .disparity_refinement import Refine, RefinePretrained
from .partial_inpainting import Inpaint as PartialInpaint
from .pointcloud_inpainting import Inpaint
from .utils import load_models, resize_image


class Pipeline():
    def __init__(self, model_paths=None, partial_inpainting=False, dolly=False, output_frames=False, pretrain=False, d2=False,
                 device='cuda:0', steps=75, inpaint_dtype=None, semantics_path=None, miopen_find=None, allow_random_weights=None):
        self.objectCommon = {'dblFocal': 1024.0 / 2, 'dblBaseline': 120}       # pipeline.py:26-27
        # The networks are four fifths of a video's time, and how fast MIOpen runs their convolutions depends on whether this
        # MACHINE has measured them before (measured, MI355X, profiles/r04_networks.txt): PyTorch's default (immediate) mode on a
        # machine with an empty MIOpen find-db picks by heuristics -- the 1024^2 Inpaint forward takes 21.4 ms, the partial one
        # 22.1; MIOpen's find step (torch.backends.cudnn.benchmark) measures the solvers -- 15.0 / 19.3 ms -- but takes 20-50 s in
        # EVERY process that asks for it; yet once one process has run it, its results sit in MIOpen's user find-db on disk and the
        # immediate mode of every later process picks the measured solvers too: 15.0 / 19.3 ms with a first call of 0.1 s.
        # (channels_last is NOT the answer on this stack: 36.3 instead of 21.4 ms immediate, 17.0 instead of 15.0 with the find.)
        # So ``miopen_find`` = 'auto' (the default; env KBE_MIOPEN_FIND=auto): the first call for an image size this machine
        # has not tuned runs under the find step -- it says so on stderr -- and leaves a marker next to the find-db
        # (~/.cache/kbe: utils.miopen_tuned_once); every later call and process runs in immediate mode on the tuned db.  True / '1':
        # always find (a long-lived server that sees many sizes); False / '0': never (tests, one-off runs that cannot wait).
        if miopen_find is None:
            miopen_find = {'1': True, '0': False}.get(os.environ.get('KBE_MIOPEN_FIND', 'auto'), 'auto')
        self.miopen_find = miopen_find
        if miopen_find is True:
            torch.backends.cudnn.benchmark = True
        self.partial_inpainting, self.dolly, self.output_frames, self.d2 = partial_inpainting, dolly, output_frames, d2
        self.device, self.steps = torch.device(device), steps
        self.moduleSemantics = Semantics().to(self.device).eval()
        self.moduleDisparity = Disparity().to(self.device).eval()
        self.moduleRefine = (RefinePretrained() if pretrain else Refine()).to(self.device).eval()
        self.moduleInpaint = (PartialInpaint() if partial_inpainting else Inpaint()).to(self.device).eval()
        models_list = [{'model': self.moduleDisparity, 'type': 'disparity'}, {'model': self.moduleRefine, 'type': 'refine'},
                       {'model': self.moduleInpaint, 'type': 'inpaint'}]
        paths = list(model_paths) if model_paths else [None, None, None]
        if len(paths) == 4:
            # pipeline.py:53-55 builds a second Inpaint for depth; process_inpaint's two-network branch is
            # broken in the reference (common.py:50-69), so the fourth checkpoint is loaded but unused
            self.moduleInpaintDepth = Inpaint().to(self.device).eval()
            models_list.append({'model': self.moduleInpaintDepth, 'type': 'inpaint'})
        # A missing checkpoint is an ERROR, as in the reference (torch.load of utils.py:206 raises): a video rendered from random
        # weights is garbage that looks like a result.  Benches, tests and smoke runs, which only need the arithmetic, say so:
        # ``allow_random_weights=True`` / --allow-random-weights / env KBE_ALLOW_RANDOM_WEIGHTS=1 -> seeded weights and a warning.
        if allow_random_weights is None:
            allow_random_weights = os.environ.get('KBE_ALLOW_RANDOM_WEIGHTS') == '1'
        load_models(models_list, paths, seed_missing=bool(allow_random_weights))
        # optional reduced-precision GridNet (default fp32 = the reference's arithmetic).  ``inpaint_dtype``
        # torch.bfloat16 / torch.float16, or env KBE_INPAINT_DTYPE=bf16|fp16: the two inpaint passes are 80 % of a
        # video's time and run ~4x faster in bf16; the frames then differ visibly in the last bits of the inpainted
        # regions only (not covered by the parity tests)
        if inpaint_dtype is None:
            inpaint_dtype = {'bf16': torch.bfloat16, 'fp16': torch.float16}.get(os.environ.get('KBE_INPAINT_DTYPE', ''))
        if inpaint_dtype is not None and hasattr(self.moduleInpaint, 'compute_dtype'):
            self.moduleInpaint.compute_dtype = inpaint_dtype
        # Semantics = torchvision's vgg19_bn(pretrained=True) in the reference (disparity_estimation.py:86): an ImageNet
        # checkpoint that torchvision downloads.  Neither torchvision nor the network is a dependency here, so the weights
        # come from a file: ``semantics_path`` / env KBE_SEMANTICS_PATH = a torchvision vgg19_bn state dict
        # (torch.save(torchvision.models.vgg19_bn(weights='IMAGENET1K_V1').state_dict(), path) on any machine).  Without
        # it the module gets seeded random weights -- and says so: a trained Disparity checkpoint fed random semantic
        # features gives a meaningless disparity.
        semantics_path = semantics_path or os.environ.get('KBE_SEMANTICS_PATH')
        if semantics_path and os.path.exists(semantics_path):
            state = torch.load(semantics_path, map_location='cpu')
            self.moduleSemantics.load_torchvision_state_dict(state.get('model_state_dict', state) if isinstance(state, dict) else state)
        elif not allow_random_weights:
            raise FileNotFoundError('semantics (VGG19-bn) weights %r not found: pass semantics_path / --semantics-path / KBE_SEMANTICS_PATH (a torchvision '
                                    'vgg19_bn state dict), or allow_random_weights=True / --allow-random-weights for runs that only need the arithmetic'
                                    % (semantics_path,))
        else:
            import warnings
            warnings.warn('checkpoint %r not found: semantics (VGG19-bn) network runs with seeded random weights; pass '
                          'semantics_path / --semantics-path / KBE_SEMANTICS_PATH (a torchvision vgg19_bn state dict)' % (semantics_path,))
            synthetic.seeded_fill_(self.moduleSemantics, 999)

    @torch.no_grad()
    def estimate(self, tensorImage):
        """pipeline.py:61-100: fills ``self.objectCommon`` from an image [1,3,H,W] in 0..1."""
        oc = self.objectCommon
        tensorImage = tensorImage.to(self.device).contiguous()
        oc['intWidth'], oc['intHeight'] = tensorImage.size(3), tensorImage.size(2)
        resized = resize_image(tensorImage, max_size=int(max(oc['intWidth'], oc['intHeight']) / 2))
        disparity = self.moduleDisparity(resized, self.moduleSemantics(resized))
        if self.d2:
            disparity = torch.ones_like(disparity)
        disparity = self.moduleRefine(tensorImage, disparity)
        low = disparity.min()
        if low < 0.0:
            disparity = disparity - low
        disparity = disparity / disparity.max() * oc['dblBaseline']
        depth = (oc['dblFocal'] * oc['dblBaseline']) / (disparity + 1e-7)
        points = common.depth_to_points(depth, oc['dblFocal'])
        oc['dblDispmin'], oc['dblDispmax'] = disparity.min().item(), disparity.max().item()
        oc['objectDepthrange'] = synthetic.depthrange_of(depth)
        oc['tensorRawPoints'] = points.view(1, 3, -1)
        oc['tensorRawImage'], oc['tensorRawDisparity'], oc['tensorRawDepth'] = tensorImage, disparity, depth
        return oc

    @torch.no_grad()
    def __call__(self, tensorImage, zoom_settings, output_path=None, inpaint_depth=False, pretrained_estim=False):
        with common.on_device_of(getattr(self, 'device', torch.device('cpu'))):      # the C ABI launches on the current device's stream
            return self._run(tensorImage, zoom_settings, output_path, inpaint_depth, pretrained_estim)

    def tuning_tag(self, width, height):
        """What MIOpen is tuned for, once per machine (utils.miopen_tuned_once): this pipeline's networks at this image size."""
        return '%dx%d-%s%s%s' % (width, height, 'partial' if self.partial_inpainting else 'plain', '-dolly' if self.dolly else '',
                                 '-pretrain' if isinstance(self.moduleRefine, RefinePretrained) else '')

    def _run(self, tensorImage, zoom_settings, output_path, inpaint_depth, pretrained_estim):
        from .utils import miopen_tuned_once
        auto = getattr(self, 'miopen_find', False) == 'auto'
        with miopen_tuned_once(self.tuning_tag(tensorImage.size(3), tensorImage.size(2)) if auto else '', getattr(self, 'device', 'cpu'), enabled=auto):
            self.estimate(tensorImage)
            if inpaint_depth:
                raise NotImplementedError('two-network depth inpainting is broken in the reference (common.py:50-69)')
            frames = common.process_kenburns({'dblSteps': np.linspace(0.0, 1.0, self.steps).tolist(),
                                              'objectFrom': zoom_settings['objectFrom'], 'objectTo': zoom_settings['objectTo'],
                                              'boolInpaint': True, 'dolly': self.dolly}, self.objectCommon, self.moduleInpaint)
        if output_path is not None:
            os.makedirs(output_path, exist_ok=True)
            # channel order on disk as the reference produces it: frames are in the INPUT's channel order
            # (BGR from cv2.imread unless --pretrained-estim); cv2.imwrite / moviepy want BGR / RGB (:125-134)
            to_rgb = (lambda f: f) if pretrained_estim else (lambda f: f[:, :, ::-1])
            rgb = [to_rgb(f) for f in frames]
            if self.output_frames:
                write_frames(os.path.join(output_path, 'frames'), rgb)
            # forth and back (:131): the SAME array objects twice -- an intra-frame encoder (the Motion-JPEG writers below) encodes each once
            write_video(os.path.join(output_path, '3d_kbe.mp4'), rgb + rgb[-2::-1], fps=25)
        return frames


WRITER_THREADS = None        # host threads the writers encode on (None: one per core, at most 32; KBE_WRITER_THREADS overrides; 1: the caller's thread only)
_JPEG_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'csrc', 'libkbe_jpeg.so')
_jpeg_lib = None


def _writer_pool_size(n_jobs):
    want = os.environ.get('KBE_WRITER_THREADS') or WRITER_THREADS or min(32, os.cpu_count() or 1)
    return max(1, min(int(want), max(n_jobs, 1)))


def jpeg_encoder():
    """'native' (libkbe_jpeg.so, include/kbe_jpeg.h: a batch of frames on host threads) or 'pillow' (one frame at a time: Pillow's encoder
    holds the interpreter lock -- measured, 8 threads: 201 ms against 171 ms on one).  KBE_JPEG=pillow forces the latter; a missing
    library falls back to it with a warning (the writers are host-side conveniences, not the render path: that one has no fallback)."""
    global _jpeg_lib
    if os.environ.get('KBE_JPEG', 'native') == 'pillow':
        return 'pillow', None
    if _jpeg_lib is None:
        import ctypes
        if not os.path.exists(_JPEG_LIB_PATH):
            import warnings
            warnings.warn('%s is missing (python -c "import __graft_entry__ as g; g.build()"): the video writer encodes with Pillow, one frame at a time' % _JPEG_LIB_PATH)
            return 'pillow', None
        lib = ctypes.CDLL(_JPEG_LIB_PATH)
        lib.kbe_jpeg_bound.restype = ctypes.c_size_t
        lib.kbe_jpeg_bound.argtypes = [ctypes.c_int, ctypes.c_int]
        lib.kbe_jpeg_encode_batch.restype = ctypes.c_int
        _jpeg_lib = lib
    return 'native', _jpeg_lib


def _on_threads(jobs, work):
    """work(job) for every job on the writers' pool of host threads (for work that releases the interpreter lock: zlib does)."""
    workers = _writer_pool_size(len(jobs))
    if workers <= 1:
        return [work(j) for j in jobs]
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=workers) as pool:
        return list(pool.map(work, jobs))


def png_bytes(frame_rgb, level=1):
    """One uint8 HxWx3 frame as a PNG (8-bit RGB, every row with the Sub filter, ONE zlib stream at `level`; default 1 = cv2.imwrite's
    default, /root/reference/utils/pipeline.py:125).  Written here rather than through Pillow because zlib.compress releases the
    interpreter lock and Pillow's PNG encoder does not: write_frames encodes a video's frames side by side."""
    import struct
    import zlib
    a = np.ascontiguousarray(frame_rgb)
    h, w = a.shape[:2]
    raw = np.empty((h, 1 + 3 * w), np.uint8)
    raw[:, 0] = 1                                                   # filter type 1 (Sub): each byte minus the byte three to its left
    flat = a.reshape(h, 3 * w)
    raw[:, 1:4] = flat[:, :3]
    raw[:, 4:] = flat[:, 3:] - flat[:, :-3]                         # (uint8 arithmetic wraps: exactly the filter's modulo 256)

    def chunk(tag, body):
        return struct.pack('>I', len(body)) + tag + body + struct.pack('>I', zlib.crc32(tag + body) & 0xFFFFFFFF)
    return (b'\x89PNG\r\n\x1a\n' + chunk(b'IHDR', struct.pack('>IIBBBBB', w, h, 8, 2, 0, 0, 0)) + chunk(b'IDAT', zlib.compress(raw.tobytes(), level)) + chunk(b'IEND', b''))


def write_frames(frames_dir, frames_rgb):
    """%d.png per frame (pipeline.py:122-126), encoded on the writers' threads."""
    os.makedirs(frames_dir, exist_ok=True)

    def save(job):
        idx, frame = job
        with open(os.path.join(frames_dir, '%d.png' % idx), 'wb') as f:
            f.write(png_bytes(frame))
    _on_threads(list(enumerate(frames_rgb)), save)


def write_mjpeg_avi(path, frames_rgb, fps=25, quality=92):
    """A playable video without any external encoder: Motion-JPEG in an AVI container (RIFF 'AVI ' with one 'vids' / 'MJPG' stream,
    an 'idx1' index; every frame a baseline JPEG from PIL).  What write_video writes for an ``.avi`` where there is no ffmpeg binary."""
    import struct
    h, w = frames_rgb[0].shape[:2]
    jpegs = _jpegs(frames_rgb, quality)
    n = len(jpegs)
    biggest = max(len(j) for j in jpegs)

    def chunk(tag, data):
        return tag + struct.pack('<I', len(data)) + data + (b'\0' if len(data) & 1 else b'')

    def lst(tag, data):
        return b'LIST' + struct.pack('<I', len(data) + 4) + tag + data

    avih = struct.pack('<14I', int(1e6 / fps), biggest * fps, 0, 0x10, n, 0, 1, biggest, w, h, 0, 0, 0, 0)       # AVIF_HASINDEX
    strh = b'vids' + b'MJPG' + struct.pack('<IHHIIIIIIII', 0, 0, 0, 0, 1, fps, 0, n, biggest, 0xFFFFFFFF, 0) + struct.pack('<4h', 0, 0, w, h)
    strf = struct.pack('<IiiHH4sIiiII', 40, w, h, 1, 24, b'MJPG', w * h * 3, 0, 0, 0, 0)
    hdrl = lst(b'hdrl', chunk(b'avih', avih) + lst(b'strl', chunk(b'strh', strh) + chunk(b'strf', strf)))
    index, offset = [], 4
    for j in jpegs:
        index.append(b'00dc' + struct.pack('<III', 0x10, offset, len(j)))               # AVIIF_KEYFRAME
        offset += 8 + len(j) + (len(j) & 1)
    movi_size = offset                                                                  # 'movi' + the chunks
    idx1 = chunk(b'idx1', b''.join(index))
    riff_size = 4 + len(hdrl) + 8 + movi_size + len(idx1)
    with open(path, 'wb') as f:
        f.write(b'RIFF' + struct.pack('<I', riff_size) + b'AVI ' + hdrl + b'LIST' + struct.pack('<I', movi_size) + b'movi')
        for j in jpegs:                                                                 # (streamed: no copy of the whole video in between)
            f.write(b'00dc' + struct.pack('<I', len(j)))
            f.write(j)
            if len(j) & 1:
                f.write(b'\0')
        f.write(idx1)
    return path


def _jpegs(frames_rgb, quality):
    """One baseline JPEG per frame.  A frame OBJECT that occurs several times in the list (the way back of a forth-and-back video,
    pipeline.py:131) is encoded once; the distinct frames go to libkbe_jpeg.so as one batch on the writers' host threads."""
    distinct, first = [], {}
    for frame in frames_rgb:
        if id(frame) not in first:
            first[id(frame)] = len(distinct)
            distinct.append(frame)
    kind, lib = jpeg_encoder()
    sizes = {f.shape for f in distinct}
    if kind == 'native' and len(sizes) == 1:
        import ctypes
        arrays = [np.ascontiguousarray(f, dtype=np.uint8) for f in distinct]
        h, w = arrays[0].shape[:2]
        cap = int(lib.kbe_jpeg_bound(w, h))
        # (the output buffers are worst-case sized -- 6 MB for a 1024^2 frame -- and reused: batches of at most ~256 MB of them)
        per = max(1, min(len(arrays), (256 << 20) // cap))
        outs = [np.empty(cap, np.uint8) for _ in range(per)]
        encoded = []
        for at in range(0, len(arrays), per):
            part = arrays[at:at + per]
            n = len(part)
            got = (ctypes.c_size_t * n)()
            rc = lib.kbe_jpeg_encode_batch((ctypes.c_void_p * n)(*[a.ctypes.data for a in part]), n, w, h, 3 * w, int(quality),
                                           (ctypes.c_void_p * n)(*[o.ctypes.data for o in outs[:n]]), ctypes.c_size_t(cap), got, _writer_pool_size(n))
            if rc != 0:
                raise RuntimeError('kbe_jpeg_encode_batch: %d' % rc)
            encoded += [outs[i][:got[i]].tobytes() for i in range(n)]
    else:
        import io
        from PIL import Image

        def encode(frame):
            buf = io.BytesIO()
            Image.fromarray(np.ascontiguousarray(frame)).save(buf, format='JPEG', quality=quality)
            return buf.getvalue()
        encoded = [encode(f) for f in distinct]
    return [encoded[first[id(frame)]] for frame in frames_rgb]


def write_mjpeg_mp4(path, frames_rgb, fps=25, quality=92):
    """An ``.mp4`` without any external encoder: an ISO base media file (ISO/IEC 14496-12) with one video track whose samples are
    baseline JPEGs (PIL) -- sample entry ``mp4v`` with an ``esds`` whose objectTypeIndication is 0x6C, "Visual ISO/IEC 10918-1
    (JPEG)", the registered way of carrying Motion-JPEG in MP4 (what ffmpeg's mp4 muxer writes for ``-c:v mjpeg``; ffmpeg-based
    players, VLC and mpv decode it).  Layout: ftyp, mdat (the samples, one chunk), moov (every sample a sync sample: no stss)."""
    import struct
    jpegs = _jpegs(frames_rgb, quality)
    h, w = frames_rgb[0].shape[:2]
    n, delta = len(jpegs), 512
    timescale = int(fps) * delta
    duration = n * delta

    def box(tag, body):
        return struct.pack('>I', len(body) + 8) + tag + body

    def full(tag, version, flags, body):
        return box(tag, struct.pack('>I', (version << 24) | flags) + body)

    def descriptor(tag, body):
        assert len(body) < 128
        return bytes([tag, len(body)]) + body

    matrix = struct.pack('>9I', 0x10000, 0, 0, 0, 0x10000, 0, 0, 0, 0x40000000)
    ftyp = box(b'ftyp', b'isom' + struct.pack('>I', 0x200) + b'isomiso2mp41')
    biggest, total = max(len(j) for j in jpegs), sum(len(j) for j in jpegs)
    first_sample = len(ftyp) + 8
    assert first_sample + total + 8 < (1 << 32), 'a 32-bit chunk offset: the video is too long for this writer'
    decoder = descriptor(0x04, bytes([0x6C, 0x11]) + struct.pack('>I', biggest)[1:] + struct.pack('>II', biggest * 8 * int(fps), total * 8 * int(fps) // n))
    esds = full(b'esds', 0, 0, descriptor(0x03, struct.pack('>HB', 1, 0) + decoder + descriptor(0x06, b'\x02')))
    name = b'Motion-JPEG'
    mp4v = box(b'mp4v', b'\0' * 6 + struct.pack('>H', 1) + b'\0' * 16 + struct.pack('>HHIIIH', w, h, 0x480000, 0x480000, 0, 1) +
               bytes([len(name)]) + name.ljust(31, b'\0') + struct.pack('>Hh', 24, -1) + esds)
    stbl = box(b'stbl', full(b'stsd', 0, 0, struct.pack('>I', 1) + mp4v) +
               full(b'stts', 0, 0, struct.pack('>III', 1, n, delta)) +
               full(b'stsc', 0, 0, struct.pack('>IIII', 1, 1, n, 1)) +
               full(b'stsz', 0, 0, struct.pack('>II', 0, n) + b''.join(struct.pack('>I', len(j)) for j in jpegs)) +
               full(b'stco', 0, 0, struct.pack('>II', 1, first_sample)))
    minf = box(b'minf', full(b'vmhd', 0, 1, struct.pack('>4H', 0, 0, 0, 0)) +
               box(b'dinf', full(b'dref', 0, 0, struct.pack('>I', 1) + full(b'url ', 0, 1, b''))) + stbl)
    mdia = box(b'mdia', full(b'mdhd', 0, 0, struct.pack('>IIIIHH', 0, 0, timescale, duration, 0x55C4, 0)) +
               full(b'hdlr', 0, 0, struct.pack('>I', 0) + b'vide' + b'\0' * 12 + b'VideoHandler\0') + minf)
    tkhd = full(b'tkhd', 0, 3, struct.pack('>IIIII', 0, 0, 1, 0, duration) + b'\0' * 8 + struct.pack('>hhhH', 0, 0, 0, 0) + matrix +
                struct.pack('>II', w << 16, h << 16))
    mvhd = full(b'mvhd', 0, 0, struct.pack('>IIIIIH', 0, 0, timescale, duration, 0x10000, 0x100) + b'\0' * 10 + matrix + b'\0' * 24 + struct.pack('>I', 2))
    with open(path, 'wb') as f:
        # (the samples go out one by one: joined into one `mdat` body first, then into one file image, the video's 20 MB were copied three
        # times before they reached the file -- more than half of what the writer took once the encoding ran on threads)
        f.write(ftyp + struct.pack('>I', total + 8) + b'mdat')
        for j in jpegs:
            f.write(j)
        f.write(box(b'moov', mvhd + box(b'trak', tkhd + mdia)))
    return path


def write_video(path, frames_rgb, fps=25):
    """mpeg4 through an ffmpeg pipe when the binary exists (what moviepy does, pipeline.py:132-134).  Without one the video is still
    written, under the name asked for, with the encoder this package carries: Motion-JPEG -- in an MP4 container for an ``.mp4``
    (write_mjpeg_mp4), in an AVI for an ``.avi`` (write_mjpeg_avi) -- and the function says so and returns False."""
    ffmpeg = shutil.which('ffmpeg')
    h, w = frames_rgb[0].shape[:2]
    if ffmpeg is None:
        writer = write_mjpeg_avi if path.lower().endswith('.avi') else write_mjpeg_mp4
        writer(path, frames_rgb, fps)
        print('ffmpeg not found: %s holds Motion-JPEG (%d frames at %d fps) instead of mpeg4' % (path, len(frames_rgb), fps))
        return False
    proc = subprocess.Popen([ffmpeg, '-y', '-loglevel', 'error', '-f', 'rawvideo', '-pix_fmt', 'rgb24', '-s', '%dx%d' % (w, h),
                             '-r', str(fps), '-i', '-', '-c:v', 'mpeg4', path], stdin=subprocess.PIPE)
    for frame in frames_rgb:
        proc.stdin.write(np.ascontiguousarray(frame).tobytes())
    proc.stdin.close()
    return proc.wait() == 0
