"""Monocular disparity estimation: VGG19-bn semantics + a 6-row x 4-column GridNet.

Drop-in for ``/root/reference/models/disparity_estimation.py`` (``Semantics`` :82-117,
``Disparity`` :119-198): same call signatures and, for ``Disparity``, the same state-dict
entries (``moduleImage``, ``moduleSemantics``, grid edges ``'{r}x{c} - {r'}x{c'}'``,
``moduleDisparity``) so the reference's checkpoints load unchanged.

``Semantics`` in the reference wraps ``torchvision.models.vgg19_bn(pretrained=True)``;
torchvision is not a dependency here, so the VGG19-bn feature stack (up to the 4th pool, with
the reference's ``ceil_mode`` pools) is defined explicitly and
:meth:`Semantics.load_torchvision_state_dict` accepts a torchvision ``vgg19_bn`` state dict.
All convolutions are stock PyTorch-ROCm (MIOpen); this stage runs once per image.
"""
from collections import OrderedDict

import torch
import torch.nn as nn

from .pointcloud_inpainting import Basic, Downsample, Upsample, _edge, _match

# torchvision vgg19_bn `features` up to (excluding) the 5th block: conv widths, 'M' = 2x2 max-pool
_VGG19_CFG = (64, 64, 'M', 128, 128, 'M', 256, 256, 256, 256, 'M', 512, 512, 512, 512, 'M')

ROW_FEATURES = (32, 48, 64, 512, 512, 512)
N_COLUMNS = 4


class Semantics(nn.Module):
    def __init__(self):
        super().__init__()
        # blocks are keyed by their index in torchvision's `features` Sequential so that weights map 1:1
        blocks, cin, idx = [], 3, 0
        for v in _VGG19_CFG:
            if v == 'M':
                blocks.append(nn.MaxPool2d(kernel_size=2, stride=2, ceil_mode=True))       # disparity_estimation.py:91
                idx += 1
            else:
                blocks.append(nn.Sequential(OrderedDict([(str(idx), nn.Conv2d(cin, v, kernel_size=3, padding=1)),
                                                         (str(idx + 1), nn.BatchNorm2d(v)),
                                                         (str(idx + 2), nn.ReLU(inplace=True))])))
                cin = v
                idx += 3
        self.moduleVgg = nn.Sequential(*blocks)
        # the ImageNet statistics of :109-113 as (non-persistent: not in the state dict) buffers: built from Python lists in every
        # forward they were two synchronous host-to-device copies per image
        self.register_buffer('_mean', torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1), persistent=False)
        self.register_buffer('_std', torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1), persistent=False)

    def load_torchvision_state_dict(self, state):
        """`state`: torchvision vgg19_bn().state_dict() (keys 'features.N.weight' ...)."""
        own = self.state_dict()
        mapped = {}
        for key in own:
            n = key.split('.', 2)[2]                # 'moduleVgg.<block>.<N>.<param>' -> '<N>.<param>'
            mapped[key] = state['features.' + n]
        self.load_state_dict(mapped)

    def forward(self, tensorInput):
        # BGR -> RGB and ImageNet statistics (:109-113), without modifying the caller's tensor
        mean, std = self._mean.to(tensorInput.dtype), self._std.to(tensorInput.dtype)
        return self.moduleVgg((tensorInput.flip(1) - mean) / std)       # [:, [2, 1, 0]]: a flip of the three channels, without an index tensor from the host


class Disparity(nn.Module):
    def __init__(self):
        super().__init__()
        self.spectral_norm = False
        self.moduleImage = nn.Conv2d(3, ROW_FEATURES[0], kernel_size=7, stride=2, padding=3)
        self.moduleSemantics = nn.Conv2d(512, 512, kernel_size=3, stride=1, padding=1)
        rows = len(ROW_FEATURES)
        for row, feat in enumerate(ROW_FEATURES):
            for col in range(N_COLUMNS - 1):
                self.add_module(_edge(row, col, row, col + 1), Basic('relu-conv-relu-conv', [feat, feat, feat]))
        for col in (0, 1):
            for row in range(rows - 1):
                lo, hi = ROW_FEATURES[row], ROW_FEATURES[row + 1]
                self.add_module(_edge(row, col, row + 1, col), Downsample([lo, hi, hi]))
        for col in (2, 3):
            for row in range(rows - 1, 0, -1):
                hi, lo = ROW_FEATURES[row], ROW_FEATURES[row - 1]
                self.add_module(_edge(row, col, row - 1, col), Upsample([hi, lo, lo]))
        self.moduleDisparity = Basic('conv-relu-conv', [ROW_FEATURES[0], ROW_FEATURES[0], 1])

    def _run(self, r0, c0, r1, c1, x, extra=None):
        return self._modules[_edge(r0, c0, r1, c1)](x, extra)

    def forward(self, tensorImage, tensorSemantics):
        """image [B,3,H,W], semantics [B,512,H/16,W/16] -> disparity [B,1,H/2,W/2]"""
        rows = len(ROW_FEATURES)
        level = [self.moduleImage(tensorImage)]
        for r in range(1, rows):
            nxt = self._run(r - 1, 0, r, 0, level[r - 1])
            if r == 3:
                nxt = nxt + self.moduleSemantics(tensorSemantics)           # :159
            level.append(nxt)
        for r in range(rows):
            lateral = self._run(r, 0, r, 1, level[r])
            level[r] = lateral if r == 0 else self._run(r - 1, 1, r, 1, level[r - 1], extra=lateral)       # lateral + the stream from above
        for col in (2, 3):
            for r in range(rows - 1, -1, -1):
                lateral = self._run(r, col - 1, r, col, level[r])
                if r != rows - 1:
                    lateral = self._run(r + 1, col, r, col, level[r + 1], extra=lateral)       # lateral + _match(the stream from below, lateral)
                level[r] = lateral
        return self.moduleDisparity(level[0])
