"""spconv.pytorch.modules (is_spconv_module: mosaic3d/src/models/utils/structure.py:300-327)."""
from segmentation_amd.sparse import SparseModule, SparseSequential, is_spconv_module

__all__ = ["is_spconv_module", "SparseModule", "SparseSequential"]
