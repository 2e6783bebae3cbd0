"""Partial-convolution variant of the inpainting GridNet.

Drop-in for ``/root/reference/models/partial_inpainting.py`` (``Inpaint``, :99-279): same module
names (``p_relu_1 / conv1 / p_relu_2 / conv2 / moduleShortcut`` inside the grid edges), hence the
same 171-entry state dict (input conv ``[32,68,3,3]``: no mask channel).

Masks ride along with the features and are merged with ``torch.min`` wherever two streams meet
(:167, :187, :209); after every x2 upsampling the mask is re-binarised at 0.5 (:90).  All masks
are identical across channels, so they are carried with ONE channel here and expanded only at
the API boundary.

Inference contract.  The reference returns its 32-channel propagated mask as
``'tensorExisting'``, which makes ``process_inpaint`` raise (it indexes a 3-channel image with
it, common.py:75-77); that path is unreachable from ``kbe.py``.  Here ``'tensorExisting'`` is the
1-channel INPUT hole mask -- what the plain ``Inpaint`` returns and what ``process_inpaint``
needs -- and the propagated mask is available as ``'tensorMaskOut'`` ([B,32,H,W] view).
"""
import threading

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import common
from .partial_conv import PartialConv2d
from .pointcloud_inpainting import ROW_FEATURES, N_COLUMNS, _edge, _act


def _pconv(cin, cout, k=3, stride=1, with_mask=True):
    return PartialConv2d(in_channels=cin, out_channels=cout, kernel_size=k, stride=stride, padding=k // 2,
                         multi_channel=True, return_mask=with_mask)


def _one(mask):
    """[B,C,H,W] channel-identical mask -> its single channel"""
    return mask[:, 0:1]


class _Pair(nn.Module):
    """[act] - pconv - act - pconv, masks threaded through; the building block of every edge."""

    def __init__(self, intChannels, first_act, stride=1):
        super().__init__()
        cin, cmid, cout = intChannels
        if first_act:
            self.p_relu_1 = _act(cin)
        self.conv1 = _pconv(cin, cmid, stride=stride)
        self.p_relu_2 = _act(cmid)
        self.conv2 = _pconv(cmid, cout)
        self._first_act = first_act

    # The fused form skips the second `input * mask` (utils/partial_conv.py:61): right when the masks are 0 / 1 -- the first layer's
    # update mask is then 0 / 1 and its output 0 outside it -- and wrong for fractional masks.  Inpaint.forward looks at the mask it
    # is given and turns the fused form off for a call with a fractional one -- for THAT call: the switch is a per-thread
    # value the call sets and restores, not an attribute of the class that two modules on two threads would flip under one another.
    _call = threading.local()

    def _fused(self):
        """The element-wise passes around the two layers ride in the layers' own passes (include/kbe.h: kbe_prelu_mask,
        kbe_pconv_epilogue's prelu_slope / residual) when the kernel set has them and PartialConv2d.forward is this package's:
        per pair three passes over the feature maps instead of seven -- [prelu * mask] conv [renormalise + prelu] conv
        [renormalise + skip] against prelu, * mask, conv, renormalise, prelu, * mask, conv, renormalise, + skip.  Same values
        (a PReLU of the 0 the renormalisation leaves outside the mask is 0: the second multiplication has nothing to do)."""
        return getattr(_Pair._call, 'binary_masks', True) and hasattr(common._K(), 'prelu_mask') and getattr(type(self.conv1).forward, 'fuses_neighbours', False) and not torch.is_grad_enabled()

    def _pair(self, x, mask, skip=None):
        """-> (conv2(act(conv1([act] x))) [+ skip], mask)"""
        if self._fused():
            pre = False
            if self._first_act:
                x = common._K().prelu_mask(x, self.p_relu_1.weight, mask)
                pre = mask is not None
            x, mask = self.conv1(x, mask_in=mask, premasked=pre, act_slope=self.p_relu_2.weight)
            x, mask = self.conv2(x, mask_in=_one(mask), premasked=True, residual=skip)
            return x, _one(mask)
        if self._first_act:
            x = self.p_relu_1(x)
        x, mask = self.conv1(x, mask_in=mask)
        x, mask = self.conv2(self.p_relu_2(x), mask_in=_one(mask))
        return (x if skip is None else x + skip), _one(mask)


class Basic(_Pair):
    def __init__(self, strType, intChannels):
        if strType not in ('relu-conv-relu-conv', 'conv-relu-conv'):
            raise ValueError(strType)
        super().__init__(intChannels, strType == 'relu-conv-relu-conv')
        self.strType = strType
        cin, _, cout = intChannels
        self.moduleShortcut = None if cin == cout else _pconv(cin, cout, k=1, with_mask=False)

    def forward(self, tensorInput, mask_in=None):
        skip = tensorInput if self.moduleShortcut is None else self.moduleShortcut(tensorInput)   # no mask: :44, :53
        return self._pair(tensorInput, mask_in, skip)


class Downsample(_Pair):
    def __init__(self, intChannels):
        super().__init__(intChannels, True, stride=2)

    def forward(self, tensorInput, mask_in=None):
        return self._pair(tensorInput, mask_in)


class Upsample(_Pair):
    def __init__(self, intChannels):
        super().__init__(intChannels, True)
        self.upsample = nn.Upsample(scale_factor=2, mode='bilinear', align_corners=False)

    def forward(self, tensorInput, mask_in=None):
        mask = (self.upsample(mask_in) > 0.5).float()
        # (the feature maps' x2 bilinear upsampling through kbe_upsample2x_act where the fused passes are in use: PyTorch's
        # upsample_bilinear2d runs at a quarter of the memory's rate)
        K = common._K() if (tensorInput.is_cuda and tensorInput.dtype == torch.float32 and tensorInput.device.index == torch.cuda.current_device() and self._fused()) else None
        up = K.upsample2x_act(tensorInput.contiguous()) if K is not None and hasattr(K, 'upsample2x_act') else self.upsample(tensorInput)
        return self._pair(up, mask)


def _crop_like(x, like, value=None):
    if x.size(2) != like.size(2):
        x = x[:, :, :-1, :]
    if x.size(3) != like.size(3):
        x = x[:, :, :, :-1]
    return x


class Inpaint(nn.Module):
    def __init__(self):
        super().__init__()
        self.spectral_norm = False
        self.tensorMean = None
        self.tensorStd = None
        self.moduleContext = nn.Sequential(nn.Conv2d(4, 64, 3, 1, 1), _act(64), nn.Conv2d(64, 64, 3, 1, 1), _act(64))
        self.moduleInput = Basic('conv-relu-conv', [3 + 1 + 64, ROW_FEATURES[0], ROW_FEATURES[0]])
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

    def _run(self, r0, c0, r1, c1, x, m):
        return self._modules[_edge(r0, c0, r1, c1)](x, m)

    def forward(self, tensorMasks, tensorImage=None, tensorDisparity=None, tensorData=None, tensorContext=None):
        if tensorImage is not None and tensorContext is None:
            tensorImage, tensorDisparity = self.normalize_images_disp(tensorImage, tensorDisparity, not_normed=True)
        if tensorData is None:
            if tensorContext is None:
                tensorContext = self.moduleContext(torch.cat([tensorImage, tensorDisparity], 1))
            tensorData = torch.cat([tensorImage, tensorDisparity, tensorContext], 1)

        # (one host look at the mask per forward -- a device-to-host synchronisation, so only where the answer matters: when the fused
        # form could run at all.  A fractional mask takes the unfused pairs, which multiply by the mask twice as the reference does.)
        could_fuse = hasattr(common._K(), 'prelu_mask') and not torch.is_grad_enabled()
        binary = bool(((tensorMasks == 0) | (tensorMasks == 1)).all()) if could_fuse else True
        was = getattr(_Pair._call, 'binary_masks', True)
        _Pair._call.binary_masks = binary
        try:
            return self._forward(tensorData, tensorMasks)
        finally:
            _Pair._call.binary_masks = was

    def _forward(self, tensorData, tensorMasks):
        rows = len(ROW_FEATURES)
        feat, mask = [None] * rows, [None] * rows
        feat[0], mask[0] = self.moduleInput(tensorData, mask_in=_one(tensorMasks))
        for r in range(1, rows):
            feat[r], mask[r] = self._run(r - 1, 0, r, 0, feat[r - 1], mask[r - 1])
        for r in range(rows):                                       # column 1
            feat[r], mask[r] = self._run(r, 0, r, 1, feat[r], mask[r])
            if r != 0:
                down, dmask = self._run(r - 1, 1, r, 1, feat[r - 1], mask[r - 1])
                feat[r] = feat[r] + down
                mask[r] = torch.min(mask[r], dmask)
        for col in (2, 3):
            for r in range(rows - 1, -1, -1):
                feat[r], mask[r] = self._run(r, col - 1, r, col, feat[r], mask[r])
                if r != rows - 1:
                    up, umask = self._run(r + 1, col, r, col, feat[r + 1], mask[r + 1])
                    feat[r] = feat[r] + _crop_like(up, feat[r])
                    mask[r] = torch.min(mask[r], _crop_like(umask, mask[r]))
        tensorImage, _ = self.moduleImage(feat[0], mask_in=None)
        tensorDisparity, _ = self.moduleDisparity(feat[0], mask_in=None)
        tensorImage, tensorDisparity = self.normalize_images_disp(tensorImage, tensorDisparity, not_normed=False)
        return {
            'tensorExisting': tensorMasks,
            'tensorMaskOut': mask[0].expand(-1, ROW_FEATURES[0], -1, -1),
            'tensorImage': tensorImage if self.training else tensorImage.clamp(0.0, 1.0),
            'tensorDisparity': F.threshold(input=tensorDisparity, threshold=0.0, value=0.0),
        }

    def pointcloud_inpainting(self, tensorImage, tensorDisparity, tensorShift, objectCommon, dblFocal=None):
        """partial_inpainting.py:226-259 (identical to the plain network's point-cloud side)."""
        if dblFocal is None:
            dblFocal = objectCommon['dblFocal']
        assert tensorImage.shape[0] == 1, 'Please process one image at a time.'
        K = common._K()
        tensorDepth = (dblFocal * objectCommon['dblBaseline']) / (tensorDisparity + 0.0000001)
        tensorValid = K.laplacian_valid(tensorDisparity, tensorDisparity.max(), 0.03)
        tensorPoints = K.depth_to_points(tensorDepth, dblFocal, valid=tensorValid).view(1, 3, -1)
        tensorImage, tensorDisparity = self.normalize_images_disp(tensorImage, tensorDisparity, not_normed=True)
        tensorContext = self.moduleContext(torch.cat([tensorImage, tensorDisparity], 1))
        features = torch.cat([tensorImage, tensorDisparity, tensorContext], 1).view(1, 68, -1)
        tensorRender, tensorExisting = K.render_pointcloud(tensorPoints + tensorShift, features, objectCommon['intWidth'],
                                                           objectCommon['intHeight'], dblFocal, objectCommon['dblBaseline'])
        tensorExisting = (tensorExisting > 0.0).float()
        tensorExisting = tensorExisting * K.spatial_filter(tensorExisting, 'median-5')
        return self.forward(tensorData=tensorRender * tensorExisting, tensorMasks=tensorExisting)

    def normalize_images_disp(self, tensorImage, tensorDisparity, not_normed=True):
        from .pointcloud_inpainting import Inpaint as _Plain
        return _Plain.normalize_images_disp(self, tensorImage, tensorDisparity, not_normed)
