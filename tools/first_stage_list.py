"""List the kernels of the last first-stage pass in a rocprofv3 kernel-trace CSV: start offset, duration, gap."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the passes are separated by synchronisations: take the last quarter by count
per = len(rows) // 4
last = rows[-per:]
t0 = int(last[0]["Start_Timestamp"])
prev_end = t0
tot = 0
for r in last:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.1f}  gap {(s - prev_end) / 1e3:6.1f}  {r['Kernel_Name'][:70]}")
    prev_end = e
    tot += e - s
print(f"{len(last)} kernels, busy {tot / 1e3:.1f} us, span {(prev_end - t0) / 1e3:.1f} us")
