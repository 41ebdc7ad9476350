#!/usr/bin/env python3
"""How much of a step runs with several kernels in flight?  From a rocprofv3 --kernel-trace CSV (one row per dispatch with
Start_Timestamp / End_Timestamp in ns): over the LAST training step of the trace (from the end of the second-to-last group of
adamw_multi_k launches to the end of the last group -- `bench.py --steps 1 --warmup 1`: the timed step) the wall time with exactly
0 / 1 / 2 / ... kernels in flight, the sum of the kernel durations and the number of queues that carried work.
usage: python scripts/overlap_stats.py <kernel_trace.csv>"""
import csv
import sys

path = sys.argv[1]
rows = list(csv.DictReader(open(path)))
col = lambda r, *names: next(r[n] for n in names if n in r)      # noqa: E731
ev = [(int(col(r, "Start_Timestamp", "Start")), int(col(r, "End_Timestamp", "End")), col(r, "Queue_Id", "Queue"), col(r, "Kernel_Name", "Name")) for r in rows]
ev.sort()
opt = [e for e in ev if "adamw_multi_k" in e[3]]
groups = []                                    # AdamW launch groups = optimizer steps (launches of one step are < 5 ms apart)
for e in opt:
    if groups and e[0] - groups[-1][1] < 5e6:
        groups[-1][1] = e[1]
    else:
        groups.append([e[0], e[1]])
assert len(groups) >= 2, "need two optimizer steps in the trace"
lo, hi = groups[-2][1], groups[-1][1]
ev = [e for e in ev if e[0] >= lo and e[1] <= hi]
pts = sorted([(e[0], 1) for e in ev] + [(e[1], -1) for e in ev])
depth, last, at = 0, pts[0][0], [0.0] * 8
for t, d in pts:
    at[min(depth, 7)] += t - last
    depth += d
    last = t
wall = pts[-1][0] - pts[0][0]
busy = sum(e[1] - e[0] for e in ev)
print(f"{path}: last training step: {len(ev)} dispatches on {len(set(e[2] for e in ev))} queues, window {wall/1e6:.1f} ms")
print(f"  sum of kernel durations      {busy/1e6:8.1f} ms  ({busy/wall:.2f} x the window)")
for k in range(0, 5):
    print(f"  exactly {k} kernel(s) in flight {at[k]/1e6:8.1f} ms  ({100*at[k]/wall:5.1f} %)")
print(f"  5 or more                    {sum(at[5:])/1e6:8.1f} ms  ({100*sum(at[5:])/wall:5.1f} %)")
