# usage: pmc_run.sh "<command>" <kernel regex> "<CTR CTR ...>" ["<CTR ...>" ...]
# one rocprofv3 pass per quoted counter group; prints per-kernel averages (raw counter units)
S=$1; R=$2; shift 2
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "$@"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $grp --kernel-include-regex "$R" --output-format csv -d /tmp/pmc_$i -- $S > /tmp/pmc_$i.log 2>&1
  f=$(ls /tmp/pmc_$i/*/*counter_collection.csv 2>/dev/null | head -1)
  if [ -z "$f" ]; then echo "pass $i ($grp) produced no counters"; tail -3 /tmp/pmc_$i.log; continue; fi
  python - "$f" <<'PY'
import csv,sys,collections,re
rows=list(csv.DictReader(open(sys.argv[1])))
agg=collections.OrderedDict()
for r in rows:
    m=re.search(r'(\w+<[^>]*>|\w+)\(', r['Kernel_Name'].replace('(anonymous namespace)::',''))
    name=m.group(1) if m else r['Kernel_Name'][:40]
    k=(r['Dispatch_Id'], name, r['Grid_Size'])
    agg.setdefault(k,collections.OrderedDict())[r['Counter_Name']]=float(r['Counter_Value'])
for k,v in agg.items():
    print(f"{k[0]:>5s} {k[1]:44s} grid={k[2]:>8s} " + ' '.join(f"{a}={b:.4g}" for a,b in v.items()))
PY
done
