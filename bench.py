#!/usr/bin/env python
"""Headline benchmark: voxels/s of the SpUNet hot path (voxelize -> kernel maps -> spunet34c+ppt forward
-> CLIP-text logits -> argmax -> per-point labels) on synthetic ScanNet-shaped scenes, inputs resident
in HBM.  One process per GPU; scenes are sharded across ranks with no data-path collective.

    python bench.py                       # 1 GPU, configs[1] of BASELINE.json (one ~273 k-voxel scene per step)
    python bench.py --gpus N              # starts N ranks itself (torch.distributed.run) when not already under one
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W
    python bench.py --workload cfg3       # configs[2]: 64 scenes of 150-250 k voxels, largest-first over the ranks

Prints ONE JSON line on rank 0 with `roofline` (convolution kernel, HIP-event timed on the launch stream),
`stages` (voxelizer and kernel maps against the HBM bound), `variants` (the other precisions, same workload),
`parity` (every precision against a float64 run of the CPU oracle on the full-size scene) and `cpu_baseline`
(the CPU oracle in f32 on the host cores).
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3    # MI355X dense f32 matrix peak (v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD)
PEAK_BF16_MFMA_TFLOPS = 2500.0  # MI355X dense bf16 matrix peak
PEAK_HBM_GBS = 8000.0
# peak in algorithmic (f32-product) FLOP/s of the instruction each precision issues: the split paths spend
# 3 (6) bf16 MFMA products per f32 product, plain bf16 one
PEAK = {"f32": PEAK_F32_MFMA_TFLOPS, "bf16x6": PEAK_BF16_MFMA_TFLOPS / 6, "bf16x3": PEAK_BF16_MFMA_TFLOPS / 3,
        "bf16": PEAK_BF16_MFMA_TFLOPS, "f16x3": PEAK_BF16_MFMA_TFLOPS / 3}  # f16 and bf16 MFMA share one dense peak
DTYPE = {"f32": "f32 (exact f32 MFMA)",
         "bf16x6": "f32 (every product: 6 bf16 MFMA products of 3-term splits, f32 accumulate; f32-level accuracy)",
         "bf16x3": "f32 storage, products from 2-term bf16 splits (3 bf16 MFMA products, f32 accumulate)",
         "bf16": "bf16 storage, f32 accumulate",
         "f16x3": "f32 (every operand as an f16 pair exact to 2^-22: 3 f16 MFMA products, f32 accumulate; f32-level "
                  "accuracy; operands must lie inside the f16 range, checked on the device)"}
HEADLINE = "bf16x6"            # arithmetic not narrower than the reference's fp32 (see `parity` in the output)
GRID = 0.02
N_CLASSES = 200
TARGET_VOXELS = 290_000        # scene-generator target; lands at ~273 k occupied 2 cm voxels
CFG3_SCENES = 64


def build_engine(device, precision=HEADLINE, label_head=None):
    import torch
    from segmentation_amd import spunet
    from segmentation_amd.engine import SpUNetEngine
    from segmentation_amd import scenes
    cfg = spunet.SPUNET34C
    torch.manual_seed(0)
    model = spunet.PPT(spunet.SpUNet(**{k: cfg[k] for k in ("in_channels", "out_channels", "base_channels",
                                                             "context_channels", "channels", "layers",
                                                             "conditions")}), cfg["conditions"], 256).eval()
    eng = SpUNetEngine(model, cfg, "ScanNet", device, precision=precision, label_head=label_head)
    eng.set_text_embeddings(torch.from_numpy(scenes.text_embeddings(N_CLASSES, cfg["out_channels"], seed=1)))
    return eng, model, cfg


def cfg3_targets():
    """Scene-generator targets of the 64-scene job (seeds 0..63): occupied voxels ~ U(150 k, 250 k)."""
    import numpy as np
    rng = np.random.default_rng(2024)
    return [int(v * TARGET_VOXELS / 273_000) for v in rng.uniform(150_000, 250_000, size=CFG3_SCENES)]


def my_scene_specs(workload, rank, world, scenes_per_rank, target_voxels):
    """(seed, target_voxels) of the scenes this rank owns."""
    from segmentation_amd import sharding
    if workload == "cfg3":
        targets = cfg3_targets()
        # the point count of a generated scene is proportional to its target: largest first over the ranks
        mine = sharding.assign_scenes(targets, world)[rank]
        return [(i, targets[i]) for i in mine]
    # scene i of the job goes to rank i mod world (weak scaling: scenes_per_rank scenes on every GPU)
    return [(rank + j * world, target_voxels) for j in range(scenes_per_rank)]


def cpu_reference(model, cfg, scene, emb):
    """The reference path restated on the CPU (oracle/): timed in f32 on this host's cores (the baseline), and run
    once in float64 as the yardstick every GPU precision -- and the f32 CPU run itself -- is measured against."""
    import numpy as np
    import torch
    import oracle
    from oracle import spunet as ospunet
    cores = min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", os.cpu_count() or 1)))
    torch.set_num_threads(cores)
    P = {}
    for k, v in model.state_dict().items():
        P[k[len("backbone."):] if k.startswith("backbone.") else k] = v.detach().float().cpu()
    n = scene["coord"].shape[0]
    off = np.array([0, n], dtype=np.int64)
    passes = 3  # about 10 s on 16 cores: a bounded sample of the same workload
    t0 = time.perf_counter()
    for _ in range(passes):
        sp, out32 = ospunet.spunet_from_points(P, cfg, scene["coord"], scene["feat"], off, GRID)
        pred32 = oracle.head_predict(out32, sp["v2p_map"], emb)
    dt = time.perf_counter() - t0
    n0 = sp["voxel_coords"].shape[0]
    P64 = {k: v.double() for k, v in P.items()}
    _, out64 = ospunet.spunet_from_points(P64, cfg, scene["coord"], scene["feat"], off, GRID, dtype=torch.float64)
    logits64 = out64 @ emb.double().t()
    cb = dict(value=passes * n0 / dt, unit="voxels/s", cores=cores, kind="port",
              sample=f"{passes} f32 passes over one synthetic scene, N0={n0} voxels / {n} points, {dt:.1f} s wall")
    return cb, dict(sp=sp, feat64=out64, logits64=logits64, feat32=out32, pred32=pred32)


def feature_errors(f, ref64):
    """Per-voxel relative error of features f [N, D] against the float64 reference: max and mean over voxels of
    ||f - ref||_2 / ||ref||_2."""
    import torch
    d = (f.double() - ref64).norm(dim=1) / ref64.norm(dim=1).clamp_min(1e-30)
    return float(d.max()), float(d.mean())


def label_agreement(pred_points, ref, top2_margin=None):
    import torch
    v2p = torch.as_tensor(ref["sp"]["v2p_map"], dtype=torch.int64)
    ref_pred = ref["logits64"].max(1)[1][v2p]
    return float((pred_points.cpu() == ref_pred).double().mean())


def launch_ranks(args, argv):
    """`bench.py --gpus N` outside a distributed launcher: start N ranks as a child job (nothing in this process
    has touched a GPU yet) and pass its output through."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    return subprocess.call(cmd)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="cfg2", choices=["cfg2", "cfg3", "train"],
                    help="cfg2: one ScanNet-scale scene per GPU and step (weak scaling); cfg3: 64 scenes of "
                         "150-250 k voxels shared out largest-first (strong scaling); train: configs[3], one "
                         "optimisation step of train_spunet_multidata_ppt per step (4 scenes per GPU, 300 captions "
                         "per scene, forward + backward + gradient all-reduce + SGD)")
    ap.add_argument("--train-precision", default="bf16x3", choices=["f32", "bf16x3", "bf16x6"],
                    help="--workload train: how forward and input-gradient convolutions form their products")
    ap.add_argument("--train-voxels", type=int, default=100_000, help="--workload train: scene-generator target per scene")
    ap.add_argument("--force-reducer", action="store_true",
                    help="--workload train on one GPU: run the gradient exchange anyway (a process group of one rank "
                         "over RCCL), so that its packing, hooks, collective and write-back are part of the step")
    ap.add_argument("--target-voxels", type=int, default=TARGET_VOXELS)
    ap.add_argument("--scenes-per-rank", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--precision", default=HEADLINE, choices=["f32", "bf16x6", "bf16x3", "bf16", "f16x3"],
                    help="how products are formed: exact f32 MFMA, bf16 MFMA on 3-/2-term splits of f32, plain bf16")
    ap.add_argument("--fold-head", action="store_true", help="fold `final` into the text matrix (not the headline)")
    ap.add_argument("--no-variants", action="store_true", help="skip the measurements of the other precisions")
    ap.add_argument("--net-streams", type=int, default=3,
                    help="streams the networks of consecutive scenes are dealt to (predict_many); 1 = one network at a time")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="finish every scene before the next one starts (predict) instead of overlapping the map "
                         "building of the next scene with the network of the current one (predict_many)")
    ap.add_argument("--dry-run", action="store_true",
                    help="rehearse the launch, sharding and reduction on the CPU (gloo), no kernels")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        raise SystemExit(launch_ranks(args, sys.argv[1:]))

    import numpy as np
    import torch
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if world != args.gpus and rank == 0:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; reporting n_gpus={world}", file=sys.stderr)
    from segmentation_amd import sharding
    if args.workload == "train":
        return train_main(args, rank, world, local)
    specs = my_scene_specs(args.workload, rank, world, args.scenes_per_rank, args.target_voxels)
    scaling = "strong" if args.workload == "cfg3" else "weak"

    if world > 1:
        # one process per GPU on a host with about two cores per GPU: the few host-side torch operations of a scene
        # must not fan out into a thread pool per rank
        torch.set_num_threads(1)

    if args.dry_run:
        dist = None
        if world > 1:
            import torch.distributed as dist
            dist.init_process_group("gloo")
        units = float(sum(t for _, t in specs))
        elapsed, total = sharding.reduce_job(1.0 + 0.25 * rank, units, dist, None)
        lo, mid, hi = sharding.reduce_stats(units, dist, None)
        if rank == 0:
            print(json.dumps(dict(metric="voxels/sec SpUNet fwd", dry_run=True, n_gpus=world, value=total / elapsed,
                                  unit="voxels/s", scaling=scaling, scenes_rank0=[s for s, _ in specs],
                                  units_per_rank=dict(min=lo, median=mid, max=hi),
                                  config=dict(workload=args.workload))), flush=True)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an AMD GPU")
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist_
        dist = dist_
        dist.init_process_group("nccl", device_id=device)

    from segmentation_amd import ops, scenes
    eng, model, cfg = build_engine(device, args.precision)

    my = []
    for seed, target in specs:
        sc = scenes.make_scene(seed=seed, target_voxels=target)
        n = sc["coord"].shape[0]
        my.append(dict(coord=torch.from_numpy(sc["coord"]).to(device), feat=torch.from_numpy(sc["feat"]).to(device),
                       offset=np.array([0, n], dtype=np.int64), host=sc if len(my) == 0 else None))

    def step(e=eng, fold=args.fold_head):
        nv = 0
        for sc in my:
            out = e.predict(sc["coord"], sc["feat"], sc["offset"], GRID, fold_head=fold)
            nv += out["n_voxels"]
        return nv

    def seq(steps):
        return [(sc["coord"], sc["feat"], sc["offset"], GRID) for _ in range(steps) for sc in my]

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(fn, steps, warmup):
        for _ in range(warmup):
            fn()
        barrier()
        t0 = time.perf_counter()
        nv = 0
        for _ in range(steps):
            nv += fn()
        barrier()
        return time.perf_counter() - t0, nv

    def timed_many(e, steps, warmup, fold=False):
        """K passes over this rank's scenes as ONE sequence through predict_many: consecutive scenes overlap (the
        voxelizer and map builders of the next scene run on a side stream beside the network of the current one);
        every scene is still computed in full, results identical to one-at-a-time calls."""
        e.predict_many(seq(warmup), fold_head=fold, net_streams=args.net_streams)
        barrier()
        t0 = time.perf_counter()
        outs = e.predict_many(seq(steps), fold_head=fold, net_streams=args.net_streams)
        barrier()
        return time.perf_counter() - t0, sum(o["n_voxels"] for o in outs)

    with torch.no_grad():
        if args.precision == "f16x3":
            eng.f16_range_exceeded(clear=True)
        if args.no_pipeline:
            elapsed, nvox = timed(step, args.steps, args.warmup)
        else:
            elapsed, nvox = timed_many(eng, args.steps, args.warmup, args.fold_head)
        # precision f16x3: the timed passes count only if no operand left the f16 range (every rank checks its own)
        f16_bad = bool(eng.f16_range_exceeded(clear=True)) if args.precision == "f16x3" else None
        if f16_bad:
            raise SystemExit("bench.py: an activation left the f16 range; precision f16x3 is not valid for this model")

        # convolution-kernel roofline: HIP events around every launch of a few more steps (this rank's stream)
        psteps = min(args.steps, 5)
        ops.PROFILE = []
        step()                      # counts pairs per schedule (host syncs) -- not used for timing
        ops.PROFILE = []
        for _ in range(psteps):
            step()
        torch.cuda.synchronize()
        prof, ops.PROFILE = ops.PROFILE, None
        stage_stats = stage_rates(eng, my[0], ops) if rank == 0 else None

        # the other precisions on the same workload (secondary numbers; single rank only)
        variants = {}
        engines = {args.precision: eng}
        if world == 1 and not args.no_variants and not args.fold_head:
            for p in ("f32", "bf16x6", "f16x3", "bf16x3", "bf16"):
                if p == args.precision or p not in ops.PRECISION:
                    continue
                e2, _, _ = build_engine(device, p)
                engines[p] = e2
                if p == "f16x3":
                    e2.f16_range_exceeded(clear=True)
                steps_v = max(args.steps, 3)
                dt, nv = timed(lambda: step(e2, False), steps_v, 2) if args.no_pipeline else timed_many(e2, steps_v, 4)
                variants[p] = dict(ms_per_step=dt / steps_v * 1e3, voxels_per_s=nv / dt, dtype=DTYPE[p])
                if p == "f16x3":  # the timed passes are valid only if no operand left the f16 range
                    variants[p]["f16_range_exceeded"] = bool(e2.f16_range_exceeded(clear=True))
            steps_v = max(args.steps // 2, 3)
            if not args.no_pipeline:
                dt, nv = timed(step, steps_v, 2)
                variants["one_scene_at_a_time"] = dict(
                    ms_per_step=dt / steps_v * 1e3, voxels_per_s=nv / dt,
                    note=f"{args.precision}, every scene finished before the next one starts (the latency of a scene)")
            dt, nv = timed(lambda: step(eng, True), steps_v, 2) if args.no_pipeline else timed_many(eng, steps_v, 2, True)
            variants["fold_head"] = dict(ms_per_step=dt / max(args.steps // 2, 3) * 1e3, voxels_per_s=nv / dt,
                                         note=f"{args.precision} with final 96->768 folded into the text matrix: same "
                                              "labels, 768-d features never formed; not the headline")

    elapsed, nvox_all = sharding.reduce_job(elapsed, float(nvox), dist, device)
    # every rank measured the convolution launches of its own scenes on its own GPU: min / median / max over the ranks
    conv_secs = sum(e0.elapsed_time(e1) * 1e-3 for kind, _, e0, e1, _, _ in prof if kind in ("igemm", "dma"))
    conv_rank_ms = sharding.reduce_stats(conv_secs / psteps / max(len(my), 1) * 1e3, dist, device)

    if rank == 0:
        ig = [(f, e0.elapsed_time(e1) * 1e-3, b) for kind, f, e0, e1, _, b in prof if kind in ("igemm", "dma")]
        flops = sum(f for f, _, _ in ig)
        secs = sum(t for _, t, _ in ig)
        alg_bytes = sum(b for _, _, b in ig)
        nl = max(len(ig), 1)
        achieved = flops / secs / 1e12 if secs > 0 else 0.0
        # HBM bytes per launch of the same kernels from the committed rocprofv3 --pmc passes (separate runs)
        traffic = None
        for rnd in ("r03", "r02"):
            pmc = os.path.join(ROOT, "profiles", f"{rnd}_{args.precision}_conv_pmc.json")
            if os.path.exists(pmc):
                traffic = json.load(open(pmc))["all_conv_igemm"]["traffic_bytes_per_launch"]
                break
        peak = PEAK[args.precision]
        kname = "m3d::conv_igemm_dma_kernel<*>"
        roofline = dict(bound="mfma", achieved=achieved, peak=peak, unit="TFLOP/s", frac=achieved / peak, traffic=traffic,
                        kernel=kname + " (all tile shapes)", launches_per_step=nl // psteps,
                        flop_per_launch=flops / nl, avg_launch_us=secs / nl * 1e6,
                        algorithmic_bytes_per_launch=alg_bytes / nl,
                        conv_ms_per_scene=secs / psteps / max(len(my), 1) * 1e3,
                        conv_ms_per_scene_over_ranks=dict(min=conv_rank_ms[0], median=conv_rank_ms[1], max=conv_rank_ms[2]),
                        measured="HIP events around every launch of scenes processed one at a time (a kernel that "
                                 "has the GPU to itself); the timed region overlaps scenes, see config.parallelism",
                        peak_note="dense matrix peak of the instruction issued, in f32-product FLOP/s "
                                  "(bf16 MFMA peak / products per f32 product)")
        cb = parity = None
        if world == 1 and not args.no_cpu_baseline and my[0]["host"] is not None:
            emb = torch.from_numpy(scenes.text_embeddings(N_CLASSES, cfg["out_channels"], seed=1))
            cb, ref = cpu_reference(model, cfg, my[0]["host"], emb)
            e32 = feature_errors(ref["feat32"], ref["feat64"])
            v2p = torch.as_tensor(ref["sp"]["v2p_map"], dtype=torch.int64)
            parity = dict(reference="CPU oracle in float64 on the same scene; feature_rel_err = per-voxel "
                                    "||f - ref||_2 / ||ref||_2 (max and mean over voxels) of the 768-d features written "
                                    "out by forward_features; fused_logit_rel_err = the same measure on the per-voxel "
                                    "logits of the fused head the timed region runs; labels per point",
                          cpu_oracle_f32=dict(feature_rel_err_max=e32[0], feature_rel_err_mean=e32[1],
                                              label_agreement=float((ref["pred32"] == ref["logits64"].max(1)[1][v2p])
                                                                    .double().mean())))
            sc0 = my[0]
            with torch.no_grad():
                for p, e in engines.items():
                    f, vox = e.forward_features(sc0["coord"], sc0["feat"], sc0["offset"], GRID)
                    assert np.array_equal(vox["voxel_coords"].cpu().numpy(), ref["sp"]["voxel_coords"])
                    assert np.array_equal(vox["v2p_map"].cpu().numpy(), ref["sp"]["v2p_map"])
                    mx, mean = feature_errors(f.cpu(), ref["feat64"])
                    del f
                    # the path the timed region runs: last layer + logits + argmax in one kernel (the features
                    # stay on chip); its per-voxel logits against the float64 ones, its labels per point
                    out = e.predict(sc0["coord"], sc0["feat"], sc0["offset"], GRID, fold_head=False, return_logits=True)
                    lmx, lmean = feature_errors(out["voxel_logits"].cpu(), ref["logits64"])
                    parity[p] = dict(feature_rel_err_max=mx, feature_rel_err_mean=mean,
                                     fused_logit_rel_err_max=lmx, fused_logit_rel_err_mean=lmean,
                                     label_agreement=label_agreement(out["pred"], ref),
                                     voxel_order_and_maps="bit-exact")
                    del out
                    if p == "f16x3":  # no operand left the f16 range anywhere in this run (status word of the device)
                        parity[p]["f16_range_exceeded"] = bool(ops.f16_status(clear=True))
            cb["label_agreement_vs_gpu"] = parity[args.precision]["label_agreement"]
        st = eng.last_stats
        wl = ("configs[1]: synthetic ScanNet-scale scene @2 cm, spunet34c+ppt forward + 768-d CLIP-text argmax "
              "(C=200), voxelize+kernel maps included") if args.workload == "cfg2" else \
             ("configs[2]: 64 synthetic scenes of 150-250 k voxels @2 cm, shared out largest-first over the ranks, "
              "spunet34c+ppt forward + 768-d CLIP-text argmax (C=200), voxelize+kernel maps included")
        line = dict(metric="voxels/sec SpUNet fwd", value=nvox_all / elapsed, unit="voxels/s", n_gpus=world,
                    steps=args.steps, warmup=args.warmup, ms_per_step=elapsed / args.steps * 1e3,
                    higher_is_better=True, scaling=scaling, vs_baseline=None, dtype=DTYPE[args.precision],
                    data="synthetic",
                    config=dict(workload=wl, scenes_rank0=len(my), voxels_per_scene=st.get("n_voxels", [0])[0],
                                points_per_scene=st.get("n_points", 0), level_voxels=st.get("n_voxels"),
                                parallelism=f"scene-sharded x{world}, no collectives" + (
                                    "" if args.no_pipeline else "; consecutive scenes of a rank software-pipelined over "
                                    f"streams (voxelizer + maps of scene i+1 on a side stream, networks of consecutive scenes dealt to "
                                    f"{args.net_streams} streams)"),
                                fold_head=args.fold_head,
                                precision=args.precision),
                    roofline=roofline, stages=stage_stats, cpu_baseline=cb, parity=parity, variants=variants or None)
        if f16_bad is not None:
            line["config"]["f16_range_exceeded"] = f16_bad
        if variants and "f16x3" in variants and parity and "f16x3" in parity:
            # not the headline (22-bit, range-limited operands), but the fastest mode whose measured error is at f32
            # level: shown at the top so that it is not overlooked among the variants
            line["fastest_f32_level_variant"] = dict(
                precision="f16x3", ms_per_step=variants["f16x3"]["ms_per_step"],
                voxels_per_s=variants["f16x3"]["voxels_per_s"],
                feature_rel_err_max=parity["f16x3"]["feature_rel_err_max"],
                exact_f32_path_feature_rel_err_max=parity.get("f32", {}).get("feature_rel_err_max"),
                f16_range_exceeded=variants["f16x3"].get("f16_range_exceeded"))
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


TRAIN_SCENES_PER_GPU = 4
TRAIN_CAPTIONS = 300


def train_main(args, rank, world, local):
    """configs[3]: the optimisation step of train_spunet_multidata_ppt (language_module.py:209-277, CaptionLoss,
    SGD nesterov + OneCycleLR, data-parallel gradient averaging as configs/trainer/ddp.yaml:4-10) on one batch per
    rank: 4 synthetic scenes of one dataset condition, 300 captions per scene.  A step = forward + loss + backward
    (gradient buckets all-reduced over RCCL while backward runs) + optimiser.  Weak scaling: every rank its own batch."""
    import torch
    from segmentation_amd import sharding
    if world > 1:
        torch.set_num_threads(1)
    if args.dry_run:
        # the control flow of a data-parallel step on the CPU: hooks, bucketed asynchronous all-reduce, write-back
        from segmentation_amd import train
        dist = None
        if world > 1:
            import torch.distributed as dist
            dist.init_process_group("gloo")
        torch.manual_seed(0)
        net = torch.nn.Sequential(torch.nn.Linear(16, 64), torch.nn.ReLU(), torch.nn.Linear(64, 8))
        red = train.GradientReducer(list(net.parameters()), dist, bucket_bytes=2048, force=dist is not None)
        opt = torch.optim.SGD(net.parameters(), lr=0.05, momentum=0.9, nesterov=True)
        x = torch.randn(32, 16, generator=torch.Generator().manual_seed(rank))
        for _ in range(2):
            opt.zero_grad(set_to_none=True)
            red.begin()
            net(x).square().mean().backward()
            buckets = red.finish()
            opt.step()
        red.close()
        check = float(sum(p.detach().double().sum() for p in net.parameters()))
        lo, mid, hi = sharding.reduce_stats(check, dist, None)   # equal parameters on every rank after averaged steps
        elapsed, total = sharding.reduce_job(1.0 + 0.25 * rank, float(TRAIN_SCENES_PER_GPU * args.train_voxels), dist, None)
        if rank == 0:
            print(json.dumps(dict(metric="voxels/sec SpUNet train step", dry_run=True, n_gpus=world, value=total / elapsed,
                                  unit="voxels/s", scaling="weak", buckets=buckets, replicas_in_sync=abs(hi - lo) < 1e-9,
                                  config=dict(workload="train"))), flush=True)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an AMD GPU")
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    dist = None
    under_launcher = all(k in os.environ for k in ("RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"))
    if world > 1 or (args.force_reducer and under_launcher):
        # a launcher's rendezvous (its store answers on MASTER_PORT; a tcp:// address of our own would wait for a
        # server nobody starts)
        import torch.distributed as dist_
        dist = dist_
        dist.init_process_group("nccl", device_id=device)
    elif args.force_reducer:
        import socket
        import torch.distributed as dist_
        dist = dist_
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=device)
    from segmentation_amd import ops, scenes, sparse, spunet, train
    sparse.set_precision(args.train_precision)
    torch.manual_seed(0)
    cfg = spunet.SPUNET34C
    net = spunet.PPT(spunet.SpUNet(**{k: cfg[k] for k in ("in_channels", "out_channels", "base_channels",
                                                           "context_channels", "channels", "layers", "conditions")}),
                     cfg["conditions"], cfg["context_channels"]).to(device)
    B = TRAIN_SCENES_PER_GPU
    batch = scenes.make_caption_batch([1000 * rank + b for b in range(B)], args.train_voxels, device,
                                      captions_per_scene=TRAIN_CAPTIONS, text_dim=cfg["out_channels"],
                                      rng_seed=100 + rank)
    step = train.TrainStep(net, total_steps=max(1000, args.steps + args.warmup + 8), dist=dist,
                           force_reducer=args.force_reducer)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(max(args.warmup, 2)):
        loss = step(batch)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step(batch)
    barrier()
    elapsed = time.perf_counter() - t0
    with torch.no_grad():
        nvox = ops.voxelize(batch["coord"], batch["feat"], batch["offset"], GRID)["n_voxels"]
    elapsed, nvox_all = sharding.reduce_job(elapsed, float(nvox) * args.steps, dist, device)
    if rank == 0:
        nparam = sum(p.numel() for p in net.parameters())
        prec_note = {"f32": "f32 (exact f32 MFMA, forward / input gradient / weight gradient)",
                     "bf16x3": "f32 storage and accumulation, products from 2-term bf16 splits (3 bf16 MFMA products) in "
                               "forward, input-gradient and weight-gradient convolutions; norms, loss and optimiser f32",
                     "bf16x6": "f32 storage and accumulation, 3-term bf16 splits (6 products) in forward and input "
                               "gradient, 2-term in the weight gradient; norms, loss and optimiser f32"}[args.train_precision]
        line = dict(metric="voxels/sec SpUNet train step (fwd+bwd+all-reduce+SGD)", value=nvox_all / elapsed,
                    unit="voxels/s", n_gpus=world, steps=args.steps, warmup=max(args.warmup, 2),
                    ms_per_step=elapsed / args.steps * 1e3, higher_is_better=True, scaling="weak", vs_baseline=None,
                    dtype=prec_note, data="synthetic",
                    config=dict(workload="configs[3]: train_spunet_multidata_ppt step, spunet34c+ppt, caption loss on "
                                         "precomputed unit text rows, SGD(nesterov)+OneCycleLR",
                                scenes_per_gpu=B, captions_per_scene=TRAIN_CAPTIONS, voxels_per_gpu=int(nvox),
                                points_per_gpu=int(batch["coord"].shape[0]), parameters=nparam,
                                gradient_exchange=("bucketed all-reduce over RCCL overlapped with backward, "
                                                   f"{len(step.reducer.members)} buckets") if step.reducer else
                                "none (one rank)", precision=args.train_precision),
                    loss=float(loss), peak_memory_gib=torch.cuda.max_memory_allocated() / 2 ** 30)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def stage_rates(eng, sc, ops):
    """Voxelizer and kernel-map phases of one scene against the HBM bound: HIP events on the launch stream around
    each phase (median of 5), algorithmic bytes from the sizes of what each phase must read and write."""
    import numpy as np
    import torch
    ev = lambda: torch.cuda.Event(enable_timing=True)
    tv, tm = [], []
    for _ in range(5):
        e0, e1, e2 = ev(), ev(), ev()
        e0.record()
        vox = ops.voxelize(sc["coord"], sc["feat"], sc["offset"], GRID, pad=128)
        e1.record()
        levels = eng.build_levels(vox["voxel_coords"], vox["spatial_shape"], vox["batch_size"])
        e2.record()
        torch.cuda.synchronize()
        tv.append(e0.elapsed_time(e1) * 1e-3)
        tm.append(e1.elapsed_time(e2) * 1e-3)
    npts, c = sc["feat"].shape
    ns = [lv.n for lv in levels]
    # voxelizer: points in (coord, feat), v2p_map out, voxels out (coords, feats)
    vbytes = npts * (12 + 4 * c + 8) + ns[0] * (16 + 4 * c)
    # kernel maps per level: coordinates in, 27-column submanifold map out, both 8-column strided maps out
    mbytes = sum(n * 16 + 27 * n * 4 for n in ns) + sum(8 * 4 * (ns[l] + ns[l + 1]) for l in range(len(ns) - 1))
    out = {}
    for name, t, b in (("voxelize", float(np.median(tv)), vbytes), ("kernel_maps_and_schedules", float(np.median(tm)), mbytes)):
        out[name] = dict(ms=t * 1e3, algorithmic_bytes=b, achieved_gbs=b / t / 1e9, peak_gbs=PEAK_HBM_GBS,
                         frac=b / t / 1e9 / PEAK_HBM_GBS, bound="hbm")
    return out


if __name__ == "__main__":
    main()
