"""GPU busy time of a kernel trace: union of the kernel intervals against the traced span, per stream-agnostic
timeline, and the part of the span in which two or more kernels overlap.  usage: python3 tools/busy_union.py <kernel_trace.csv> [skip_fraction]"""
import csv, sys
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
t0 = rows[0][0] + int((rows[-1][1] - rows[0][0]) * skip)   # steady state: the last part of the run
rows = [r for r in rows if r[0] >= t0]
span = rows[-1][1] - rows[0][0]
ev = []
for s, e, _ in rows:
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
busy = over = 0
depth, last = 0, ev[0][0]
for t, d in ev:
    if depth >= 1: busy += t - last
    if depth >= 2: over += t - last
    depth += d; last = t
tot = sum(e - s for s, e, _ in rows)
print(f"span {span / 1e6:.2f} ms, kernels {len(rows)}, busy {busy / span:.3f} of span, >=2 kernels {over / span:.3f}, sum of durations {tot / span:.3f} x span")
