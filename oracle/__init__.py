"""CPU oracle for the Mosaic3D SpUNet hot path (voxelize -> kernel maps -> sparse-conv
U-Net -> CLIP-text argmax head).

TEST INFRASTRUCTURE ONLY.  Nothing in the shipped package may import this module: it is
used by tests/, by __graft_entry__.smoke() and by the cpu_baseline leg of bench.py as the
checker, never as the thing measured or shipped.

Pinning status
--------------
* voxelize (FNV hash, voxel order, v2p_map, spatial shape, offset helpers, ravel key):
  pinned against outputs of the reference's own functions (tests/golden/reference_voxelize.npz,
  produced by tests/golden/make_golden.py from /root/reference).
* the steps in front of the voxelizer (CenterShift, NormalizeColor, collate offsets, GridSample hash / unique /
  inverse / grid_coord / min_coord / displacement): pinned against outputs of the reference's own transform
  classes (tests/golden/reference_preprocess.npz, produced by tests/golden/make_golden_preprocess.py).  The
  Stray Scanner converter (convert_stray.py) needs open3d, absent here: its three arithmetic lines are restated
  and checked by known answers only.
* the open-vocabulary head (oracle/head.py) and the caption loss of the training step (oracle/caption_loss.py):
  pinned against outputs of the reference's own CLIPAlignmentEval.predict (on the features of the reference's own
  forward pass) and CaptionLoss.loss / extract_text_features with their gradients (tests/golden/reference_loss.npz,
  produced by tests/golden/make_golden_loss.py from /root/reference).
* sparse-conv kernel maps and convolution arithmetic: the reference delegates these to the
  un-vendored third-party wheel ``spconv-cu120`` (version unpinned in
  /root/reference/mosaic3d/requirements.txt:41), which is not installed anywhere and has no
  test or golden vector in the reference tree => **parity unpinned** at that boundary.  The
  restatement is instead anchored by known-answer tests against torch.nn.functional.conv3d on
  the densified tensor (tests/test_oracle_conv_kat.py).
"""
from .voxelize import (fnv_hash_vec, ravel_hash, offset2bincount, offset2batch, batch2offset,
                       grid_coords, mean_pooling, sparsify)
from .rulebook import subm_rulebook, strided_rulebook, kernel_offsets
from .sparse_conv import (subm_conv, strided_conv, inverse_conv, conv1x1, dense_equivalent_subm,
                          dense_equivalent_strided)
from .head import head_logits, head_predict
from . import preprocess
from . import caption_loss
from .spunet import spunet_forward, fold_pdnorm, make_spunet_params, SPUNET34C
