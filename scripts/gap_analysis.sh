# GPU idle-gap analysis of one bench run from the rocprofv3 kernel trace (developer tool)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_gap -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /tmp/gap.log 2>&1
python - <<'PY'
import csv, glob, collections
f = glob.glob('/tmp/prof_gap/*/*kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows)
# last 2/3 of the timeline = the timed steps (warmup first)
t0, t1 = ev[0][0], ev[-1][1]
cut = t0 + (t1 - t0) // 3
ev = [e for e in ev if e[0] >= cut]
busy = 0; gaps = []; cur_end = ev[0][0]
for s, e, n in ev:
    if s > cur_end:
        gaps.append((s - cur_end, n))
    if e > cur_end:
        busy += e - max(s, cur_end); cur_end = e
span = ev[-1][1] - ev[0][0]
print(f"span {span/1e6:.1f} ms, busy {busy/1e6:.1f} ms, idle {(span-busy)/1e6:.1f} ms ({100*(span-busy)/span:.1f} %), kernels {len(ev)}")
h = collections.Counter()
for g, n in gaps:
    b = '<2us' if g < 2000 else '<5us' if g < 5000 else '<10us' if g < 10000 else '<20us' if g < 20000 else '<50us' if g < 50000 else '>=50us'
    h[b] += g
print({k: round(v/1e6, 2) for k, v in h.items()})
big = collections.Counter()
for g, n in gaps:
    if g >= 10000: big[n.split('(')[0][-40:]] += g
for k, v in big.most_common(12): print(f"  {v/1e6:7.2f} ms idle before {k}")
PY
