"""spconv.pytorch: the names the reference's model code reaches (grep of `spconv\\.` over mosaic3d/src)."""
from segmentation_amd.sparse import (ConvAlgo, SparseConv3d, SparseConvTensor, SparseInverseConv3d, SparseModule,
                                     SparseSequential, SubMConv3d)

from . import core, modules, ops  # noqa: F401

__all__ = ["SparseConvTensor", "SubMConv3d", "SparseConv3d", "SparseInverseConv3d", "SparseSequential", "SparseModule",
           "ConvAlgo", "modules", "ops", "core"]
