"""Print per-kernel sums of the counters in a rocprofv3 --pmc output directory, for kernels matching a substring."""
import csv, glob, sys, collections
d, pat = sys.argv[1], sys.argv[2]
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            acc[r["Kernel_Name"][:60]][r["Counter_Name"]] += float(r["Counter_Value"])
            calls[(r["Kernel_Name"][:60], r["Counter_Name"])] += 1
    for k, v in acc.items():
        print(k)
        for c, x in sorted(v.items()):
            print(f"   {c:32s} {x / calls[(k, c)]:16.0f} per launch")
