"""Colour + disparity inpainting network of the 3D Ken Burns pipeline.

Drop-in for ``/root/reference/models/pointcloud_inpainting.py`` (``Inpaint``, :83-236): same
constructor, same ``forward`` / ``pointcloud_inpainting`` / ``normalize_images_disp``
signatures and return dictionaries, and the same 171 state-dict entries (names such as
``moduleContext.0.weight`` or ``0x0 - 1x0.moduleMain.1.weight``), so a
checkpoint written by the reference loads unchanged.

The convolutions are stock PyTorch-ROCm modules (MIOpen); what is MI355X-specific here is the
point-cloud side of :meth:`Inpaint.pointcloud_inpainting`: validity mask, unprojection,
68-channel forward-warp and the median-5 hole dilation run through the HIP library
(``include/kbe.h``) via :mod:`ken_burns_effect_amd.common`.

The network is a 4-row x 4-column GridNet: rows carry 32/64/128/256 features at full, 1/2,
1/4, 1/8 resolution; columns 0-1 stream downwards, columns 2-3 stream upwards.
"""
import contextlib
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import common

ROW_FEATURES = (32, 64, 128, 256)
N_COLUMNS = 4


def _conv3(cin, cout, stride=1):
    return nn.Conv2d(cin, cout, kernel_size=3, stride=stride, padding=1)


def _act(channels):
    return nn.PReLU(num_parameters=channels, init=0.25)


def _fused_layers(x):
    """The kernel set when the element-wise passes around this block's convolutions can ride in fused HIP passes (kbe_bias_act,
    kbe_upsample2x_act; include/kbe.h), else None: inference in fp32 on the GPU only -- under autograd, autocast or on the CPU the
    blocks run as the stock modules they are made of.  ``KBE_FUSED_LAYERS=0`` turns it off (measurements, the parity test)."""
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4) or torch.is_grad_enabled() or torch.is_autocast_enabled():
        return None
    if os.environ.get('KBE_FUSED_LAYERS', '1') == '0' or x.device.index != torch.cuda.current_device():
        return None                 # (the C ABI launches on the CURRENT device's stream: a tensor elsewhere takes the stock modules)
    K = common._K()
    return K if hasattr(K, 'bias_act') else None


def _conv_act(K, conv, act, x, res1=None, res2=None):
    """act(conv(x)) + res1 + res2 with the bias add, the activation and the residuals in ONE pass over the convolution's output
    (MIOpen's Winograd kernels take no bias: PyTorch adds it -- and everything behind it -- in passes of their own)."""
    y = F.conv2d(x, conv.weight, None, conv.stride, conv.padding, conv.dilation, conv.groups)
    return K.bias_act(y, conv.bias, None if act is None else act.weight, res1, res2, out=y)


def _main_fused(K, seq, x, res1=None, res2=None):
    """A block's ``moduleMain`` -- [bilinear x2] [act] conv act conv -- then ``+ res1 + res2``: three or four passes besides the
    convolutions instead of five to eight.  A ``res2`` of another size (the stream from an odd-sized row below) is added behind
    the crop, as before."""
    mods = list(seq)
    x = x.contiguous()
    if isinstance(mods[0], nn.Upsample):
        x = K.upsample2x_act(x, mods[1].weight)
        mods = mods[2:]
    elif isinstance(mods[0], nn.PReLU):
        x = K.bias_act(x, None, mods[0].weight)
        mods = mods[1:]
    conv1, act, conv2 = mods
    y = _conv_act(K, conv1, act, x)
    if res2 is not None and res2.shape[2:] != (
            (y.size(2) + 2 * conv2.padding[0] - conv2.kernel_size[0]) // conv2.stride[0] + 1, (y.size(3) + 2 * conv2.padding[1] - conv2.kernel_size[1]) // conv2.stride[1] + 1):
        return res2 + _match(_conv_act(K, conv2, None, y, res1), res2)
    return _conv_act(K, conv2, None, y, res1, res2)


class Basic(nn.Module):
    """Residual pair of 3x3 convolutions; ``strType`` picks whether an activation comes first.
    A 1x1 ``moduleShortcut`` exists only when input and output widths differ."""

    def __init__(self, strType, intChannels):
        super().__init__()
        cin, cmid, cout = intChannels
        layers = [_conv3(cin, cmid), _act(cmid), _conv3(cmid, cout)]
        if strType == 'relu-conv-relu-conv':
            layers.insert(0, _act(cin))
        elif strType != 'conv-relu-conv':
            raise ValueError(strType)
        self.moduleMain = nn.Sequential(*layers)
        self.moduleShortcut = None if cin == cout else nn.Conv2d(cin, cout, kernel_size=1, stride=1, padding=0)

    def forward(self, tensorInput, extra=None):
        """``extra``: what the caller would add to the result (the GridNet's stream from the neighbouring row)."""
        skip = tensorInput if self.moduleShortcut is None else self.moduleShortcut(tensorInput)
        K = _fused_layers(tensorInput)
        if K is not None:
            return _main_fused(K, self.moduleMain, tensorInput, skip.contiguous(), extra)
        out = self.moduleMain(tensorInput) + skip
        return out if extra is None else extra + out


class Downsample(nn.Module):
    """act - conv(stride 2) - act - conv"""

    def __init__(self, intChannels):
        super().__init__()
        cin, cmid, cout = intChannels
        self.moduleMain = nn.Sequential(_act(cin), _conv3(cin, cmid, stride=2), _act(cmid), _conv3(cmid, cout))

    def forward(self, tensorInput, extra=None):
        K = _fused_layers(tensorInput)
        if K is not None:
            return _main_fused(K, self.moduleMain, tensorInput, None, extra)
        out = self.moduleMain(tensorInput)
        return out if extra is None else extra + out


class Upsample(nn.Module):
    """bilinear x2 - act - conv - act - conv"""

    def __init__(self, intChannels):
        super().__init__()
        cin, cmid, cout = intChannels
        self.moduleMain = nn.Sequential(nn.Upsample(scale_factor=2, mode='bilinear', align_corners=False),
                                        _act(cin), _conv3(cin, cmid), _act(cmid), _conv3(cmid, cout))

    def forward(self, tensorInput, extra=None):
        """``extra`` (the lateral stream of the row above): added to the result, which is cropped to its size first (_match)."""
        K = _fused_layers(tensorInput)
        if K is not None:
            return _main_fused(K, self.moduleMain, tensorInput, None, extra)
        out = self.moduleMain(tensorInput)
        return out if extra is None else extra + _match(out, extra)


def _edge(r0, c0, r1, c1):
    """Name of the grid edge from node (r0, c0) to node (r1, c1), as the checkpoints spell it."""
    return '%dx%d - %dx%d' % (r0, c0, r1, c1)


def _match(tensorUp, like):
    """An odd-sized level comes back one row / column too large from the x2 upsampling: crop it."""
    if tensorUp.size(2) != like.size(2):
        tensorUp = tensorUp[:, :, :-1, :]
    if tensorUp.size(3) != like.size(3):
        tensorUp = tensorUp[:, :, :, :-1]
    return tensorUp


class Inpaint(nn.Module):
    def __init__(self):
        super().__init__()
        self.spectral_norm = False
        self.tensorMean = None
        self.tensorStd = None
        # optional reduced-precision GridNet (torch.bfloat16 / torch.float16 under autocast, MIOpen); None = fp32 as the
        # reference, the only setting the parity tests cover
        self.compute_dtype = None

        self.moduleContext = nn.Sequential(_conv3(4, 64), _act(64), _conv3(64, 64), _act(64))
        # image(3) :: disparity(1) :: context(64) :: mask(1)
        self.moduleInput = Basic('conv-relu-conv', [3 + 1 + 64 + 1, ROW_FEATURES[0], ROW_FEATURES[0]])

        for row, feat in enumerate(ROW_FEATURES):
            for col in range(N_COLUMNS - 1):
                self.add_module(_edge(row, col, row, col + 1), Basic('relu-conv-relu-conv', [feat, feat, feat]))
        for col in (0, 1):
            for row in range(len(ROW_FEATURES) - 1):
                lo, hi = ROW_FEATURES[row], ROW_FEATURES[row + 1]
                self.add_module(_edge(row, col, row + 1, col), Downsample([lo, hi, hi]))
        for col in (2, 3):
            for row in range(len(ROW_FEATURES) - 1, 0, -1):
                hi, lo = ROW_FEATURES[row], ROW_FEATURES[row - 1]
                self.add_module(_edge(row, col, row - 1, col), Upsample([hi, lo, lo]))

        self.moduleImage = Basic('conv-relu-conv', [ROW_FEATURES[0], ROW_FEATURES[0], 3])
        self.moduleDisparity = Basic('conv-relu-conv', [ROW_FEATURES[0], ROW_FEATURES[0], 1])

    def _run(self, r0, c0, r1, c1, x, extra=None):
        return self._modules[_edge(r0, c0, r1, c1)](x, extra)

    def _context(self, x):
        """``moduleContext`` (conv - act - conv - act), each convolution's bias add and activation in one pass where they can be."""
        K = _fused_layers(x)
        if K is None:
            return self.moduleContext(x)
        c = self.moduleContext
        return _conv_act(K, c[2], c[3], _conv_act(K, c[0], c[1], x.contiguous()))

    def _grid(self, tensorFirst):
        """The GridNet body (pointcloud_inpainting.py:133-172): returns the top-row features of the last column."""
        rows = len(ROW_FEATURES)
        level = [tensorFirst]
        for r in range(1, rows):                                    # column 0, downwards
            level.append(self._run(r - 1, 0, r, 0, level[r - 1]))
        for r in range(rows):                                       # column 1: lateral, plus the stream from above
            lateral = self._run(r, 0, r, 1, level[r])
            level[r] = lateral if r == 0 else self._run(r - 1, 1, r, 1, level[r - 1], extra=lateral)       # lateral + the stream from above
        for col in (2, 3):                                          # columns 2, 3: lateral, plus the stream from below
            for r in range(rows - 1, -1, -1):
                lateral = self._run(r, col - 1, r, col, level[r])
                if r != rows - 1:
                    lateral = self._run(r + 1, col, r, col, level[r + 1], extra=lateral)       # lateral + _match(the stream from below, lateral)
                level[r] = lateral
        return level[0]

    def forward(self, tensorMasks, tensorImage=None, tensorDisparity=None, tensorData=None, tensorContext=None):
        """tensorMasks [B,1,H,W] (1 = known).  Either ``tensorData`` [B,68,H,W] (normalised image ::
        disparity :: context, already masked) or ``tensorImage`` + ``tensorDisparity`` (+ optional
        ``tensorContext``).  Returns {'tensorExisting', 'tensorImage', 'tensorDisparity'}."""
        if tensorImage is not None and tensorContext is None:
            tensorImage, tensorDisparity = self.normalize_images_disp(tensorImage, tensorDisparity, not_normed=True)
        if tensorData is None:
            if tensorContext is None:
                tensorContext = self._context(torch.cat([tensorImage, tensorDisparity], 1))
            tensorData = torch.cat([tensorImage, tensorDisparity, tensorContext], 1)

        fast = self.compute_dtype is not None and tensorMasks.is_cuda and not self.training
        with (torch.autocast('cuda', dtype=self.compute_dtype) if fast else contextlib.nullcontext()):
            top = self._grid(self.moduleInput(torch.cat([tensorData, tensorMasks], 1)))
            outImage, outDisparity = self.moduleImage(top), self.moduleDisparity(top)
        tensorImage, tensorDisparity = self.normalize_images_disp(outImage.float(), outDisparity.float(), not_normed=False)
        return {
            'tensorExisting': tensorMasks,
            'tensorImage': tensorImage if self.training else tensorImage.clamp(0.0, 1.0),
            'tensorDisparity': F.threshold(input=tensorDisparity, threshold=0.0, value=0.0),
        }

    @contextlib.contextmanager
    def keeping_source(self):
        """Inside: consecutive pointcloud_inpainting calls on the SAME image and disparity tensors share what depends on those alone
        (see there); on the way out -- also by an exception -- what was kept is released."""
        self._keep_source, self._kept_source = True, None
        try:
            yield self
        finally:
            self._keep_source, self._kept_source = False, None

    def pointcloud_inpainting(self, tensorImage, tensorDisparity, tensorShift, objectCommon, dblFocal=None):
        """Warps image, disparity and context features into the view displaced by ``tensorShift`` and
        inpaints what that view cannot see (pointcloud_inpainting.py:185-213)."""
        if dblFocal is None:
            dblFocal = objectCommon['dblFocal']
        assert tensorImage.shape[0] == 1, 'Please process one image at a time.'
        K = common._K()
        width, height = objectCommon['intWidth'], objectCommon['intHeight']

        # process_kenburns' set-up calls this twice with the SAME image and disparity (common.py:181-219: one pass per end pose):
        # what depends on them alone -- the points, the normalisation, the context features -- is kept from the first call, but only
        # INSIDE `with self.keeping_source():` (common.build_pointcloud): the entry holds the two inputs, the points and 68 feature
        # planes (0.3 GB at 1024^2) and must not outlive the pair of calls.  It holds the two input tensors themselves
        # (so that their addresses cannot be handed to other tensors) and their versions (an in-place change makes it stale); the
        # context network's parameters by version, storage, dtype and device (`.to()` / `.half()` replace the storage without
        # bumping the version); and the fused-layers switch.
        keeping = getattr(self, '_keep_source', False) and not torch.is_grad_enabled()
        kept = getattr(self, '_kept_source', None) if keeping else None
        stamp = (tensorImage._version, tensorDisparity._version, tensorImage.dtype, tensorImage.device, float(dblFocal), float(objectCommon['dblBaseline']),
                 os.environ.get('KBE_FUSED_LAYERS', '1')) + tuple((p._version, p.data_ptr(), p.dtype, p.device) for p in self.moduleContext.parameters())
        if kept is not None and kept[0] is tensorImage and kept[1] is tensorDisparity and kept[2] == stamp:
            tensorPoints, features, self.tensorMean, self.tensorStd = kept[3]
        else:
            tensorDepth = (dblFocal * objectCommon['dblBaseline']) / (tensorDisparity + 0.0000001)
            tensorValid = K.laplacian_valid(tensorDisparity, tensorDisparity.max(), 0.03)
            tensorPoints = K.depth_to_points(tensorDepth, dblFocal, valid=tensorValid).view(1, 3, -1)

            imageIn, disparityIn = tensorImage, tensorDisparity
            tensorImage, tensorDisparity = self.normalize_images_disp(tensorImage, tensorDisparity, not_normed=True)
            tensorContext = self._context(torch.cat([tensorImage, tensorDisparity], 1))
            features = torch.cat([tensorImage, tensorDisparity, tensorContext], 1).view(1, 68, -1)
            self._kept_source = (imageIn, disparityIn, stamp, (tensorPoints, features, self.tensorMean, self.tensorStd)) if keeping else None

        tensorRender, tensorExisting = K.render_pointcloud(tensorPoints + tensorShift, features, width, height, dblFocal,
                                                           objectCommon['dblBaseline'])
        tensorExisting = (tensorExisting > 0.0).float()
        tensorExisting = tensorExisting * K.spatial_filter(tensorExisting, 'median-5')     # hole dilation
        tensorRender = tensorRender * tensorExisting
        return self.forward(tensorData=tensorRender, tensorMasks=tensorExisting)

    def normalize_images_disp(self, tensorImage, tensorDisparity, not_normed=True):
        """``not_normed=True``: standardise both tensors with their own per-sample mean / unbiased std and
        remember the statistics on ``self``; ``False``: undo it with the remembered statistics
        (pointcloud_inpainting.py:217-236)."""
        if not_normed:
            def stats(t):
                flat = t.reshape(t.size(0), -1)
                return flat.mean(1, True).view(-1, 1, 1, 1), flat.std(1, True).view(-1, 1, 1, 1)
            meanI, stdI = stats(tensorImage)
            meanD, stdD = stats(tensorDisparity)
            self.tensorMean, self.tensorStd = [meanI, meanD], [stdI, stdD]
            tensorImage = (tensorImage - meanI) / (stdI + 0.0000001)
            tensorDisparity = (tensorDisparity - meanD) / (stdD + 0.0000001)
        else:
            tensorImage = tensorImage * (self.tensorStd[0] + 0.0000001) + self.tensorMean[0]
            tensorDisparity = tensorDisparity * (self.tensorStd[1] + 0.0000001) + self.tensorMean[1]
        return tensorImage, tensorDisparity
