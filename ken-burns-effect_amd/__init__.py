"""MI355X-native novel-view renderer for the 3D Ken Burns / dolly-zoom effect.

Scope: the per-frame point-cloud forward-warp + disocclusion
fill loop of the reference's ``utils/common.py`` (``process_kenburns`` and the
functions it drives), behind the same Python API, with every render stage running
as hand-written HIP for gfx950 through the C-ABI library declared in
``include/kbe.h``.

Module map (reference module -> this package):

=============================================  ==================================
``utils/common.py``                            :mod:`ken_burns_effect_amd.common`
``models/pointcloud_inpainting.py``            :mod:`ken_burns_effect_amd.pointcloud_inpainting`
``models/partial_inpainting.py``               :mod:`ken_burns_effect_amd.partial_inpainting`
``utils/partial_conv.py``                      :mod:`ken_burns_effect_amd.partial_conv`
``utils/pipeline.py``                          :mod:`ken_burns_effect_amd.pipeline`
``kbe.py``                                     :mod:`ken_burns_effect_amd.kbe`
=============================================  ==================================

The compute path has no CPU fallback: importing :mod:`common` is cheap, but the
first kernel call loads ``csrc/libkbe_hip.so`` and raises if it is missing.
"""

__version__ = '0.1.0'

__all__ = ['__version__']
