"""Scene sharding for multi-GPU inference: scenes are independent (voxel hash, kernel maps, eval-mode
norms and the head are all per scene), so every rank runs its own scenes and no data-path collective
exists.  The only cross-rank traffic is the timing reduction of the benchmark harness."""
from typing import List, Optional, Sequence


def assign_scenes(sizes: Sequence[int], world: int) -> List[List[int]]:
    """Longest-processing-time-first assignment of scene ids to ranks, by point count.
    Deterministic: ties go to the lower scene id / lower rank."""
    order = sorted(range(len(sizes)), key=lambda i: (-int(sizes[i]), i))
    load = [0] * world
    out: List[List[int]] = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda j: (load[j], j))
        out[r].append(i)
        load[r] += int(sizes[i])
    for r in range(world):
        out[r].sort()
    return out


def round_robin(n_scenes: int, world: int, rank: int) -> List[int]:
    """Scene i of the job goes to rank i mod world."""
    return list(range(rank, n_scenes, world))


def reduce_job(elapsed_s: float, units: float, dist=None, device=None):
    """Whole-job (max elapsed over ranks, total units over ranks).  `dist` is torch.distributed or None."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(elapsed_s), float(units)
    import torch
    t = torch.tensor([elapsed_s], dtype=torch.float64, device=device)
    u = torch.tensor([units], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(u, op=dist.ReduceOp.SUM)
    return float(t[0]), float(u[0])


def reduce_stats(value: float, dist=None, device=None):
    """(min, median, max) over the ranks of one number per rank (e.g. the convolution time per scene each rank
    measured on its own GPU).  `dist` is torch.distributed or None."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value), float(value), float(value)
    import torch
    world = dist.get_world_size()
    mine = torch.tensor([value], dtype=torch.float64, device=device)
    parts = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine)
    v = sorted(float(p[0]) for p in parts)
    mid = 0.5 * (v[(world - 1) // 2] + v[world // 2])
    return v[0], mid, v[-1]
