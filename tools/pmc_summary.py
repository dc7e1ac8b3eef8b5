"""Aggregate a rocprofv3 --pmc counter_collection CSV per kernel name: sum, calls, mean per launch."""
import csv, sys, collections
path = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for r in csv.DictReader(open(path)):
    name = r.get("Kernel_Name") or r.get("Name")
    c = r.get("Counter_Name")
    v = float(r.get("Counter_Value", 0))
    a = agg[name][c]
    a[0] += v; a[1] += 1
for name, cs in sorted(agg.items(), key=lambda kv: -sum(x[0] for x in kv[1].values())):
    for c, (tot, n) in cs.items():
        print(f"{name[:80]:80s} {c:12s} launches={n:5d} total={tot:14.1f} per_launch={tot/n:12.2f}")
