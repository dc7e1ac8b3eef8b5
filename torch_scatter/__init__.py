"""`torch_scatter` as Mosaic3D imports it (scatter: mosaic3d/src/models/utils/structure.py:10,
spconv_unet_v1m3_pdnorm.py:14; segment_csr: mosaic3d/src/models/losses/caption_loss.py:10,
mosaic3d/src/utils/caption_utils.py:9), served by segmentation_amd."""
from segmentation_amd.sparse import scatter, segment_csr

__all__ = ["scatter", "segment_csr"]
