"""Disparity refinement: upsamples the coarse disparity x4, guided by the image.

Drop-in for ``/root/reference/models/disparity_refinement.py`` (``Refine`` :65-112) and, with
``Refine(pretrained=True)`` / the alias :class:`RefinePretrained`, for
``models/disparity_refinement_pretrained.py`` (:80-127), whose only difference is that its
``Basic`` blocks carry residual shortcuts (1x1 ``moduleShortcut`` where widths differ).  Same
state-dict entries as the respective reference module.  Stock PyTorch-ROCm convolutions;
runs once per image; H and W must be multiples of 4 (``kbe.py:108-114`` crops for that).
"""
import torch
import torch.nn as nn

from .pointcloud_inpainting import Basic as _ResidualBasic, Downsample, Upsample, _act, _conv3, _fused_layers, _main_fused


class _PlainBasic(nn.Module):
    """conv - act - conv without any shortcut (disparity_refinement.py:6-27)"""

    def __init__(self, strType, intChannels):
        super().__init__()
        cin, cmid, cout = intChannels
        layers = [_conv3(cin, cmid), _act(cmid), _conv3(cmid, cout)]
        if strType == 'relu-conv-relu-conv':
            layers.insert(0, _act(cin))
        self.moduleMain = nn.Sequential(*layers)

    def forward(self, tensorInput):
        K = _fused_layers(tensorInput)
        return self.moduleMain(tensorInput) if K is None else _main_fused(K, self.moduleMain, tensorInput)


def _standardise(t):
    flat = t.reshape(t.size(0), -1)
    mean, std = flat.mean(1, True).view(-1, 1, 1, 1), flat.std(1, True).view(-1, 1, 1, 1)
    return (t - mean) / (std + 0.0000001), mean, std


class Refine(nn.Module):
    def __init__(self, pretrained=False):
        super().__init__()
        block = _ResidualBasic if pretrained else _PlainBasic
        self.spectral_norm = False
        self.moduleImageOne = block('conv-relu-conv', [3, 24, 24])
        self.moduleImageTwo = Downsample([24, 48, 48])
        self.moduleImageThr = Downsample([48, 96, 96])
        self.moduleDisparityOne = block('conv-relu-conv', [1, 96, 96])
        self.moduleDisparityTwo = Upsample([192, 96, 96])
        self.moduleDisparityThr = Upsample([144, 48, 48])
        self.moduleDisparityFou = block('conv-relu-conv', [72, 24, 24])
        self.moduleRefine = block('conv-relu-conv', [24, 24, 1])

    def forward(self, tensorImage, tensorDisparity):
        """image [B,3,H,W], coarse disparity [B,1,H/4,W/4] -> refined disparity [B,1,H,W]"""
        image, _, _ = _standardise(tensorImage)
        disparity, mean, std = _standardise(tensorDisparity)
        one = self.moduleImageOne(image)
        two = self.moduleImageTwo(one)
        thr = self.moduleImageThr(two)
        up = self.moduleDisparityOne(disparity)
        up = self.moduleDisparityTwo(torch.cat([thr, up], 1))
        up = self.moduleDisparityThr(torch.cat([two, up], 1))
        up = self.moduleDisparityFou(torch.cat([one, up], 1))
        return self.moduleRefine(up) * (std + 0.0000001) + mean


class RefinePretrained(Refine):
    """models/disparity_refinement_pretrained.py:Refine"""

    def __init__(self):
        super().__init__(pretrained=True)
