#!/bin/bash
# kernel-trace the training step with the two-stream view overlap on, then report how much kernels really overlap
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/ovl && mkdir -p /tmp/ovl
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/ovl -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /tmp/ovl/log.txt 2>&1
tail -1 /tmp/ovl/log.txt | cut -c1-300
python - <<'PY'
import csv, glob, collections
f = glob.glob('/tmp/ovl/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
print(len(rows), "kernels; columns:", list(rows[0].keys())[:14])
ev = []
byq = collections.Counter()
for r in rows:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    ev.append((s, e, r.get('Queue_Id'), r.get('Stream_Id'), r['Kernel_Name'][:40]))
    byq[(r.get('Queue_Id'), r.get('Stream_Id'))] += e - s
ev.sort()
t0, t1 = ev[0][0], max(e for _, e, *_ in ev)
# take last 40% of the trace (steady state)
cut = t0 + (t1 - t0) * 0.6
sel = [x for x in ev if x[0] >= cut]
tot = sum(e - s for s, e, *_ in sel)
# union
u, cs, ce = 0, None, None
for s, e, *_ in sel:
    if cs is None: cs, ce = s, e
    elif s <= ce: ce = max(ce, e)
    else: u += ce - cs; cs, ce = s, e
u += ce - cs
span = max(e for _, e, *_ in sel) - sel[0][0]
print(f"steady window {span/1e6:.1f} ms: sum of kernel durations {tot/1e6:.1f} ms, union busy {u/1e6:.1f} ms, overlap {(tot-u)/1e6:.1f} ms, idle {(span-u)/1e6:.1f} ms")
for k, v in byq.most_common(8): print("queue/stream", k, f"{v/1e6:.1f} ms")
PY
