"""GPU occupancy of the pipelined sequence traced by tools/pipe_trace.py (rocprofv3 --kernel-trace, csv): the part of
the steady sequence (everything after the last gap of 100 ms or more) in which no kernel runs, exactly one class of
kernels runs, or several do.  usage: python3 tools/pipe_busy.py <kernel_trace.csv>"""
import collections, csv, sys
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1])))
cut = 0
for i in range(1, len(rows)):
    if rows[i][0] - max(r[1] for r in rows[max(0, i - 50):i]) > 100e6:
        cut = i
rows = rows[cut:]
# drop the first and last tenth (ramp up / drain)
t_lo = rows[0][0] + (rows[-1][1] - rows[0][0]) // 10
t_hi = rows[-1][1] - (rows[-1][1] - rows[0][0]) // 10
rows = [r for r in rows if r[0] >= t_lo and r[1] <= t_hi]
span = t_hi - t_lo


def cls(n):
    if "conv_igemm_dma_kernel<4, 1, 3" in n or "conv_igemm_dma_kernel<4, 1, 4" in n:
        return "large_conv"
    if "head_fused" in n:
        return "head"
    if "conv_" in n:
        return "small_conv"
    return "voxelize_maps_other"


tot = collections.Counter()
ev = []
for s, e, n in rows:
    c = cls(n)
    tot[c] += e - s
    ev += [(s, 1, c), (e, -1, c)]
ev.sort()
depth = collections.Counter()
last = t_lo
only = collections.Counter()
idle = multi = 0
for t, d, c in ev:
    act = [k for k, v in depth.items() if v > 0]
    if not act:
        idle += t - last
    elif len(act) == 1:
        only[act[0]] += t - last
    else:
        multi += t - last
    depth[c] += d
    last = t
print(f"steady span {span / 1e6:.1f} ms, {len(rows)} kernels; kernel time / span by class:",
      {k: round(v / span, 3) for k, v in tot.items()})
print(f"no kernel running {idle / span:.3f}; exactly one class running:", {k: round(v / span, 3) for k, v in only.items()},
      f"; several classes {multi / span:.3f}")
