"""Partial convolution with the mask bookkeeping fused into one HIP pass.

Drop-in for ``/root/reference/utils/partial_conv.py`` (NVIDIA's ``PartialConv2d``, :14-84):
same constructor keywords (``multi_channel``, ``return_mask``), same attributes
(``weight_maskUpdater``, ``slide_winsize``, ``update_mask``, ``mask_ratio``, ``last_size``) and
the same caching rule (the mask statistics are recomputed when a mask is passed or the input
size changes, :39-40).

The reference spends a second full convolution on the mask (``conv2d(mask, ones)``, :58) plus
five element-wise passes (:62-77).  Because ``weight_maskUpdater`` is all ones, that convolution
is a box sum that is identical for every output channel, so here ONE kernel
(``kbe_pconv_epilogue``, include/kbe.h) computes the box sum, the clamp, the ratio and the
renormalised output from the raw convolution result.  A mask that is the same on every input
channel (how ``partial_inpainting.Inpaint`` uses it) may be passed with a single channel.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import common


class PartialConv2d(nn.Conv2d):
    def __init__(self, *args, **kwargs):
        self.multi_channel = kwargs.pop('multi_channel', False)
        self.return_mask = kwargs.pop('return_mask', False)
        super().__init__(*args, **kwargs)
        kh, kw = self.kernel_size
        shape = (self.out_channels, self.in_channels, kh, kw) if self.multi_channel else (1, 1, kh, kw)
        self.weight_maskUpdater = torch.ones(*shape)        # kept for API compatibility; never convolved with
        self.slide_winsize = shape[1] * shape[2] * shape[3]
        self.last_size = (None, None, None, None)
        self.update_mask = None     # [B,1,Ho,Wo] (broadcasts like the reference's Cout identical channels)
        self.mask_ratio = None      # not materialised by the fused pass; kept for attribute compatibility
        if kh != kw or self.stride[0] != self.stride[1] or self.padding[0] != self.padding[1] or self.dilation != (1, 1) \
                or self.groups != 1:
            raise NotImplementedError('fused partial convolution supports square kernels, dilation 1, groups 1')

    def forward(self, input, mask_in=None, premasked=False, act_slope=None, residual=None):
        """As the reference's forward (utils/partial_conv.py:39-83).  The three keyword arguments are this package's own, for callers
        that fuse what surrounds the layer (partial_inpainting._Pair; only on a kernel set that has them, `fuses_neighbours`):
        ``premasked``: ``input`` is already 0 wherever ``mask_in`` is -- skip ``input * mask_in`` (:61); ``residual``: added to the
        output, and ``act_slope`` [Cout]: the output then goes through PReLU -- both in the epilogue's pass (include/kbe.h)."""
        assert len(input.shape) == 4
        if torch.is_grad_enabled() and (input.requires_grad or self.weight.requires_grad):
            raise NotImplementedError('the fused partial convolution is inference-only: call it under torch.no_grad()')
        fresh = mask_in is not None or self.last_size != tuple(input.shape)
        # (the convolution without its bias where the epilogue can add it: MIOpen's Winograd kernels take none and PyTorch would
        # add it in a pass of its own)
        K = common._K()
        late_bias = self.bias is not None and getattr(K, 'pconv_epilogue_adds_bias', False)
        raw = F.conv2d(input * mask_in if mask_in is not None and not premasked else input, self.weight, None if late_bias else self.bias, self.stride, self.padding)
        if fresh:
            self.last_size = tuple(input.shape)
            self._mask = mask_in
        mask = self._mask
        if not self.multi_channel and mask is not None and mask.shape[1] != 1:
            raise ValueError('single-channel PartialConv2d wants a [*,1,H,W] mask')
        cin = self.in_channels if self.multi_channel else 1
        extra = {} if act_slope is None and residual is None else dict(act_slope=act_slope, residual=residual)
        if late_bias:
            extra['raw_without_bias'] = True
        output, um = K.pconv_epilogue(raw, self.bias, mask, self.kernel_size[0], self.stride[0], self.padding[0],
                                                in_channels=cin, in_size=tuple(input.shape[2:]), **extra)
        self.update_mask = um
        if self.return_mask:
            return output, um.expand(-1, self.out_channels, -1, -1) if self.multi_channel else um
        return output

    forward.fuses_neighbours = True         # (a forward patched in from elsewhere -- bench.py's reference formulation -- does not carry this)
