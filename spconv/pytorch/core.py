"""spconv.pytorch.core (ImplicitGemmIndiceData: mosaic3d/src/models/networks/opensegment3d/model.py:258-310)."""
from segmentation_amd.sparse import ImplicitGemmIndiceData, SparseConvTensor

__all__ = ["SparseConvTensor", "ImplicitGemmIndiceData"]
