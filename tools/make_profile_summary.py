"""Copy the judged rocprofv3 summaries from gpurun_out/ into profiles/ (tracked).

usage: python tools/make_profile_summary.py <tag> <dir>
<dir> holds kernel_stats.csv (`rocprofv3 --kernel-trace --stats`), pmc_fetch.csv and pmc_write.csv (one `--pmc` run
each) and bench_line.json, as tools/profile_bench.sh leaves them.  Writes
  profiles/<tag>_kernel_stats.csv   the statistics as produced
  profiles/<tag>_bench_line.json    the bench line of the traced run (slower than an untraced run)
  profiles/<tag>_conv_pmc.json      HBM traffic of the convolution kernels (bench.py reads traffic_bytes_per_launch)
  profiles/<tag>_stage_pmc.json     per kernel of every stage: launches, average duration, traffic, GB/s
traffic = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 bytes -- FETCH_SIZE/WRITE_SIZE are in KiB, and on gfx950
FETCH_SIZE counts half the bytes of wide coalesced reads (128-B requests tallied at 64 B)."""
import collections
import csv
import json
import os
import shutil
import sys

tag, src = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)
shutil.copy(os.path.join(src, "kernel_stats.csv"), os.path.join(dst, f"{tag}_kernel_stats.csv"))
if os.path.exists(os.path.join(src, "bench_line.json")):
    shutil.copy(os.path.join(src, "bench_line.json"), os.path.join(dst, f"{tag}_bench_line.json"))


def agg(path, counter):
    d = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            a = d[r["Kernel_Name"]]
            a[0] += float(r["Counter_Value"])
            a[1] += 1
    return d


STAGES = [("voxelize", ("vox_", "unique_", "radix_", "pick_")),
          ("kernel_maps_and_schedules", ("subm_", "strided_", "bitmap_", "sched_", "tile_", "pair_", "inverse_", "hash_")),
          ("network", ("conv_", "split_", "pack_", "norm_", "affine", "bn_")),
          ("head", ("head_", "argmax", "gather_"))]


def stage_of(name):
    short = name.replace("void ", "").replace("m3d::", "")
    for stage, prefixes in STAGES:
        if short.startswith(prefixes):
            return stage
    return "other"


f = agg(os.path.join(src, "pmc_fetch.csv"), "FETCH_SIZE")
w = agg(os.path.join(src, "pmc_write.csv"), "WRITE_SIZE")
avg_ns = {r["Name"]: (float(r["AverageNs"]), int(r["Calls"])) for r in csv.DictReader(open(os.path.join(src, "kernel_stats.csv")))}

conv = {"unit": "KiB as reported by rocprofv3; traffic_bytes = (2*fetch + write) * 1024", "kernels": {}}
tf = tw = n = 0
for k in f:
    if "conv_igemm" not in k:
        continue
    conv["kernels"][k] = dict(launches=f[k][1], fetch_kib_per_launch=f[k][0] / f[k][1],
                              write_kib_per_launch=w[k][0] / max(w[k][1], 1))
    tf += f[k][0]
    tw += w[k][0]
    n += f[k][1]
conv["all_conv_igemm"] = dict(launches=n, fetch_kib_per_launch=tf / n, write_kib_per_launch=tw / n,
                              traffic_bytes_per_launch=(2 * tf + tw) * 1024 / n)
json.dump(conv, open(os.path.join(dst, f"{tag}_conv_pmc.json"), "w"), indent=1)

stages = {"note": "per launch; traffic_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024; avg_us from the kernel trace of a "
                  "separate run; gbs = traffic / duration (HBM peak 8000 GB/s)", "stages": collections.OrderedDict()}
for k in sorted(f, key=lambda k: -avg_ns.get(k, (0, 0))[0] * avg_ns.get(k, (0, 0))[1]):
    if k not in avg_ns:
        continue
    fetch, write = f[k][0] / f[k][1], w[k][0] / max(w[k][1], 1)
    traffic = (2 * fetch + write) * 1024
    us = avg_ns[k][0] / 1e3
    stages["stages"].setdefault(stage_of(k), []).append(dict(
        kernel=k[:160], launches_traced=avg_ns[k][1], avg_us=round(us, 2), fetch_kib=round(fetch, 1),
        write_kib=round(write, 1), traffic_bytes=int(traffic), gbs=round(traffic / us / 1e3, 1)))
json.dump(stages, open(os.path.join(dst, f"{tag}_stage_pmc.json"), "w"), indent=1)
print(json.dumps(conv["all_conv_igemm"]))
for stage, ks in stages["stages"].items():
    tot = sum(k["avg_us"] * k["launches_traced"] for k in ks)
    print(f"{stage}: {tot / 1e3:.2f} ms traced in total")
