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

    def forward(self, tensorInput):
        skip = tensorInput if self.moduleShortcut is None else self.moduleShortcut(tensorInput)
        return self.moduleMain(tensorInput) + skip


class Downsample(nn.Module):
    """act - conv(stride 2) - act - conv"""

    def __init__(self, intChannels):
        super().__init__()
        cin, cmid, cout = intChannels
        self.moduleMain = nn.Sequential(_act(cin), _conv3(cin, cmid, stride=2), _act(cmid), _conv3(cmid, cout))

    def forward(self, tensorInput):
        return self.moduleMain(tensorInput)


class Upsample(nn.Module):
    """bilinear x2 - act - conv - act - conv"""

    def __init__(self, intChannels):
        super().__init__()
        cin, cmid, cout = intChannels
        self.moduleMain = nn.Sequential(nn.Upsample(scale_factor=2, mode='bilinear', align_corners=False),
                                        _act(cin), _conv3(cin, cmid), _act(cmid), _conv3(cmid, cout))

    def forward(self, tensorInput):
        return self.moduleMain(tensorInput)


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

    def _run(self, r0, c0, r1, c1, x):
        return self._modules[_edge(r0, c0, r1, c1)](x)

    def _grid(self, tensorFirst):
        """The GridNet body (pointcloud_inpainting.py:133-172): returns the top-row features of the last column."""
        rows = len(ROW_FEATURES)
        level = [tensorFirst]
        for r in range(1, rows):                                    # column 0, downwards
            level.append(self._run(r - 1, 0, r, 0, level[r - 1]))
        for r in range(rows):                                       # column 1: lateral, plus the stream from above
            lateral = self._run(r, 0, r, 1, level[r])
            level[r] = lateral if r == 0 else lateral + self._run(r - 1, 1, r, 1, level[r - 1])
        for col in (2, 3):                                          # columns 2, 3: lateral, plus the stream from below
            for r in range(rows - 1, -1, -1):
                lateral = self._run(r, col - 1, r, col, level[r])
                if r != rows - 1:
                    lateral = lateral + _match(self._run(r + 1, col, r, col, level[r + 1]), lateral)
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
                tensorContext = self.moduleContext(torch.cat([tensorImage, tensorDisparity], 1))
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

    def pointcloud_inpainting(self, tensorImage, tensorDisparity, tensorShift, objectCommon, dblFocal=None):
        """Warps image, disparity and context features into the view displaced by ``tensorShift`` and
        inpaints what that view cannot see (pointcloud_inpainting.py:185-213)."""
        if dblFocal is None:
            dblFocal = objectCommon['dblFocal']
        assert tensorImage.shape[0] == 1, 'Please process one image at a time.'
        K = common._K()
        width, height = objectCommon['intWidth'], objectCommon['intHeight']

        tensorDepth = (dblFocal * objectCommon['dblBaseline']) / (tensorDisparity + 0.0000001)
        tensorValid = K.laplacian_valid(tensorDisparity, tensorDisparity.max(), 0.03)
        tensorPoints = K.depth_to_points(tensorDepth, dblFocal, valid=tensorValid).view(1, 3, -1)

        tensorImage, tensorDisparity = self.normalize_images_disp(tensorImage, tensorDisparity, not_normed=True)
        tensorContext = self.moduleContext(torch.cat([tensorImage, tensorDisparity], 1))
        features = torch.cat([tensorImage, tensorDisparity, tensorContext], 1).view(1, 68, -1)

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
