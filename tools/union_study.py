"""CPU study: distinct input rows a Morton-ordered output tile gathers over all 27 offsets (halo size)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle
from segmentation_amd import scenes
from tools.mask_study_helpers import morton_key

tv = int(sys.argv[1]) if len(sys.argv) > 1 else 60000
sc = scenes.make_scene(0, tv); n = len(sc['coord'])
sp = oracle.sparsify(sc['coord'], sc['feat'], np.array([0, n]), 0.02)
idx = sp['voxel_coords']; pair = oracle.subm_rulebook(idx, sp['spatial_shape'])
N = idx.shape[0]
mo = np.argsort(morton_key(idx), kind='stable')
p = pair[:, mo]
for BM in (128, 256, 512):
    u, tot, act = [], [], []
    for t0 in range(0, N - BM, BM * 7):  # sample tiles
        blk = p[:, t0:t0 + BM]
        v = blk[blk >= 0]
        u.append(len(np.unique(v))); tot.append(len(v)); act.append((blk >= 0).any(1).sum())
    print(f"BM={BM}: distinct rows {np.mean(u):.0f} (max {np.max(u)}), pairs {np.mean(tot):.0f}, active offsets {np.mean(act):.1f}")
