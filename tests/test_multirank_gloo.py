"""World-size-2 rehearsal of the scene-sharded path on the CPU (gloo): assignment covers every scene exactly
once, the job-level reduction takes the max time and the total units, and no collective is needed in between."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from segmentation_amd import sharding


def test_lpt_assignment_is_a_partition_and_balanced():
    sizes = [250_000, 150_000, 240_000, 160_000, 200_000, 199_000, 151_000]
    for world in (1, 2, 4, 8):
        parts = sharding.assign_scenes(sizes, world)
        flat = sorted(i for p in parts for i in p)
        assert flat == list(range(len(sizes)))
        loads = [sum(sizes[i] for i in p) for p in parts]
        if world <= 4:
            assert max(loads) - min(loads) <= max(sizes)
    assert sharding.round_robin(10, 4, 1) == [1, 5, 9]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sizes = [100 + 10 * i for i in range(7)]
        mine = sharding.assign_scenes(sizes, world)[rank]
        # stand-in for the per-scene GPU work: units = sum of the scene sizes this rank owns
        units = float(sum(sizes[i] for i in mine))
        elapsed = 1.0 + 0.5 * rank
        dist.barrier()
        t, u = sharding.reduce_job(elapsed, units, dist)
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)
        out[rank] = (t, u, gathered)
    finally:
        dist.destroy_process_group()


def test_two_ranks_gloo():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    sizes = [100 + 10 * i for i in range(7)]
    for rank in range(world):
        t, u, gathered = out[rank]
        assert t == 1.5                       # max over ranks
        assert u == float(sum(sizes))         # total units over ranks
        assert sorted(i for p in gathered for i in p) == list(range(7))


def _grad_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from segmentation_amd import train
        torch.manual_seed(0)  # same initial weights on every rank
        net = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.ReLU(), torch.nn.Linear(16, 4))
        x = torch.full((5, 8), float(rank + 1))
        net(x).square().sum().backward()
        local = [p.grad.clone() for p in net.parameters()]
        nb = train.allreduce_gradients(list(net.parameters()), dist, bucket_bytes=256)  # several small buckets
        out[rank] = (nb, local, [p.grad.clone() for p in net.parameters()])
    finally:
        dist.destroy_process_group()


def test_gradient_allreduce_two_ranks_gloo():
    """The data-parallel gradient exchange of the training step: bucketed all-reduce, mean over ranks."""
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_grad_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    nb0, local0, avg0 = out[0]
    nb1, local1, avg1 = out[1]
    assert nb0 == nb1 and nb0 >= 2
    for a, b, m0, m1 in zip(local0, local1, avg0, avg1):
        assert torch.allclose(m0, (a + b) / 2) and torch.equal(m0, m1)


def _uneven_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from segmentation_amd import train

        class Net(torch.nn.Module):
            """Shared trunk + one head per "condition" (all heads have the same shapes, like the per-dataset norms
            of PDBatchNorm) + a parameter no rank ever uses."""

            def __init__(self):
                super().__init__()
                self.trunk = torch.nn.Linear(6, 8)
                self.heads = torch.nn.ModuleList([torch.nn.Linear(8, 3) for _ in range(3)])
                self.unused = torch.nn.Parameter(torch.ones(5))

            def forward(self, x, cond):
                return self.heads[cond](torch.relu(self.trunk(x)))

        res = {}
        for overlap in (True, False):
            torch.manual_seed(0)
            net = Net()
            red = train.GradientReducer(list(net.parameters()), dist, bucket_bytes=128, overlap=overlap)
            for step in range(2):                      # two steps: the layout and the hooks are reused
                for p in net.parameters():
                    p.grad = None
                cond = (rank + step) % 3               # every rank trains another head, and it changes per step
                x = torch.full((4, 6), float(rank + 1 + step))
                # the overlapped run is told which heads idle (as TrainStep tells it about the other conditions' norms)
                red.begin(unused=[p for j in range(3) if j != cond for p in net.heads[j].parameters()] if overlap else ())
                net(x, cond).square().sum().backward()
                local = {n: (None if p.grad is None else p.grad.clone()) for n, p in net.named_parameters()}
                nb = red.finish()
                res[(overlap, step)] = (nb, cond, local, {n: (None if p.grad is None else p.grad.clone())
                                                          for n, p in net.named_parameters()})
            red.close()
        out[rank] = res
    finally:
        dist.destroy_process_group()


def test_gradient_allreduce_with_different_parameters_per_rank_gloo():
    """Ranks whose batches carry different conditions touch different parameters: the averaged gradient of a
    parameter is sum over the ranks that have one / world, parameters nobody used keep grad None, and the overlapped
    (hook-driven) exchange gives exactly what the one after backward gives."""
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_uneven_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    for step in range(2):
        for overlap in (True, False):
            (nb0, c0, loc0, avg0), (nb1, c1, loc1, avg1) = out[0][(overlap, step)], out[1][(overlap, step)]
            assert nb0 == nb1 >= 3 and c0 != c1
            for name in loc0:
                a, b = loc0[name], loc1[name]
                if a is None and b is None:
                    assert avg0[name] is None and avg1[name] is None, name     # used by nobody: untouched
                    continue
                want = ((0 if a is None else a) + (0 if b is None else b)) / world
                assert torch.allclose(avg0[name], want) and torch.equal(avg0[name], avg1[name]), name
            assert avg0["unused"] is None and avg0[f"heads.{c0}.weight"] is not None and avg0[f"heads.{c1}.weight"] is not None
        for name in out[0][(True, step)][3]:
            x, y = out[0][(True, step)][3][name], out[0][(False, step)][3][name]
            assert (x is None and y is None) or torch.equal(x, y), name


def test_bench_starts_its_own_ranks_dry_run():
    """`python bench.py --gpus 2` outside a launcher starts two ranks itself (before anything touches a GPU) and
    relays rank 0's JSON line; --dry-run rehearses sharding and the job reduction over gloo without kernels."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    for workload, scaling in (("cfg2", "weak"), ("cfg3", "strong")):
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run", "--workload",
                            workload], capture_output=True, text=True, timeout=300, env=env)
        assert r.returncode == 0, r.stderr[-2000:]
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1
        d = json.loads(lines[0])
        assert d["n_gpus"] == 2 and d["dry_run"] and d["scaling"] == scaling
        if workload == "cfg3":
            import bench
            targets = bench.cfg3_targets()
            assert len(targets) == 64 and abs(d["value"] * 1.25 - sum(targets)) < 1e-6 * sum(targets)
            parts = sharding.assign_scenes(targets, 2)
            assert d["scenes_rank0"] == parts[0] and len(parts[0]) + len(parts[1]) == 64


def test_largest_first_assignment_balances_the_64_scene_job_over_eight_ranks():
    """configs[2] on a whole node: 64 scenes of 150-250 k voxels over 8 ranks, heaviest rank within 5 % of the lightest."""
    import bench
    targets = bench.cfg3_targets()
    for world in (2, 4, 8):
        parts = sharding.assign_scenes(targets, world)
        assert sorted(i for p in parts for i in p) == list(range(64))
        loads = [sum(targets[i] for i in p) for p in parts]
        assert max(loads) / min(loads) <= 1.05, (world, loads)
        assert all(len(p) in (64 // world - 1, 64 // world, 64 // world + 1) for p in parts)


def test_bench_dry_run_with_eight_ranks():
    """The launch, the sharding and both reductions of bench.py with eight ranks on gloo (no kernels)."""
    import json
    import subprocess
    import sys
    import bench
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["OMP_NUM_THREADS"] = "1"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--dry-run", "--workload", "cfg3"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    targets = bench.cfg3_targets()
    parts = sharding.assign_scenes(targets, 8)
    loads = sorted(float(sum(targets[i] for i in p)) for p in parts)
    assert d["n_gpus"] == 8 and d["scaling"] == "strong" and d["scenes_rank0"] == parts[0]
    assert abs(d["value"] * 2.75 - sum(targets)) < 1e-6 * sum(targets)       # slowest rank: 1 + 0.25 * 7 seconds
    u = d["units_per_rank"]
    assert u["min"] == loads[0] and u["max"] == loads[-1] and u["median"] == 0.5 * (loads[3] + loads[4])


def test_bench_train_workload_dry_run_two_ranks():
    """`bench.py --workload train --gpus 2 --dry-run`: two ranks run the data-parallel step's control flow on gloo
    (hooks, bucketed asynchronous all-reduce, write-back, optimiser) and end with identical parameters."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run", "--workload", "train"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["dry_run"] and d["scaling"] == "weak" and d["buckets"] >= 2
    assert d["replicas_in_sync"] is True
    assert abs(d["value"] * 1.25 - 2 * 4 * 100_000) < 1e-6
