"""spconv.pytorch.ops: indice_avgpool_implicit_gemm as OpenSegment3D's decoder calls it
(mosaic3d/src/models/networks/opensegment3d/model.py:274-276, :301-306)."""


def indice_avgpool_implicit_gemm(features, pair_fwd, num_out, calc_count=False):
    from segmentation_amd import ops
    return ops.avgpool(features, pair_fwd, num_out, calc_count)
