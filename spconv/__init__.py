"""`spconv` as Mosaic3D imports it (mosaic3d/src/models/utils/structure.py:6,
mosaic3d/src/models/networks/spunet/spconv_unet_v1m3_pdnorm.py:10, spconv_unet_v1m1_base.py:11,
opensegment3d/model.py:5), served by segmentation_amd: with this directory's parent on sys.path the reference's
model files import and run unedited on the gfx950 kernels of libm3d_hip.so.  Only the surface the reference uses
exists; there is no CPU path."""
from . import pytorch  # noqa: F401

__version__ = "2.3.6+segmentation_amd"
