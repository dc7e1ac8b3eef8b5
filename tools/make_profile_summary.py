"""Copy the judged rocprofv3 summaries from gpurun_out/ into profiles/ (tracked).

usage: python tools/make_profile_summary.py <tag> <kernel_stats.csv> [<pmc_fetch.csv> <pmc_write.csv>]
Writes profiles/<tag>_kernel_stats.csv (as produced by `rocprofv3 --kernel-trace --stats`) and, when the two
PMC passes are given, profiles/<tag>_conv_pmc.json with the HBM traffic of the convolution kernels:
traffic = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 bytes -- FETCH_SIZE/WRITE_SIZE are in KiB, and on gfx950
FETCH_SIZE counts half the bytes of wide coalesced reads (128-B requests tallied at 64 B)."""
import csv, json, os, shutil, sys, collections

tag, stats = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.makedirs(os.path.join(root, "profiles"), exist_ok=True)
shutil.copy(stats, os.path.join(root, "profiles", f"{tag}_kernel_stats.csv"))
if len(sys.argv) >= 5:
    def agg(path, counter):
        d = collections.defaultdict(lambda: [0.0, 0])
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] == counter:
                a = d[r["Kernel_Name"]]
                a[0] += float(r["Counter_Value"]); a[1] += 1
        return d
    f, w = agg(sys.argv[3], "FETCH_SIZE"), agg(sys.argv[4], "WRITE_SIZE")
    out = {"unit": "KiB as reported by rocprofv3; traffic_bytes = (2*fetch + write) * 1024", "kernels": {}}
    tf = tw = n = 0
    for k in f:
        if "conv_igemm" not in k:
            continue
        out["kernels"][k] = dict(launches=f[k][1], fetch_kib_per_launch=f[k][0] / f[k][1],
                                 write_kib_per_launch=w[k][0] / max(w[k][1], 1))
        tf += f[k][0]; tw += w[k][0]; n += f[k][1]
    out["all_conv_igemm"] = dict(launches=n, fetch_kib_per_launch=tf / n, write_kib_per_launch=tw / n,
                                 traffic_bytes_per_launch=(2 * tf + tw) * 1024 / n)
    json.dump(out, open(os.path.join(root, "profiles", f"{tag}_conv_pmc.json"), "w"), indent=1)
    print(json.dumps(out["all_conv_igemm"]))
