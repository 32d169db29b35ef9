"""ann3depth_amd — MI355X-native replacement for ann3depth's convolutional depth-regression training path.

Python here is only the host side (tensor allocation through PyTorch-ROCm, stream plumbing, the reference's
model/dataset plugin surface); every arithmetic op of the path is a HIP kernel in ``liba3d.so`` (``csrc/``),
reached through the C ABI declared in ``include/a3d.h``.  There is no CPU or eager-PyTorch fallback: importing an
op without the built library raises.
"""
__version__ = '0.1'
