"""segmentation_amd: MI355X-native sparse-convolution engine for Mosaic3D's SpUNet hot path
(voxelize -> kernel maps -> sparse conv U-Net -> CLIP-text argmax), hand-written gfx950 HIP kernels
behind a C ABI (include/m3d.h), with a host-side mirror of the spconv.pytorch surface.

The kernels live in libm3d_hip.so next to this file; operators raise if it is missing -- there is
no CPU fallback."""
from . import _lib  # noqa: F401

__all__ = ["ops", "sparse", "spunet", "engine", "compat", "scenes"]
__version__ = "0.1.0"
