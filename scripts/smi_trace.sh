#!/bin/bash
# usage: smi_trace.sh <out.txt> <period_s> -- <command...>    clock / socket power / junction temperature beside a command
OUT=$1; PER=$2; shift 3
( while true; do rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|Socket Graphics Package Power|Sensor junction" | sed -E 's/.*junction\) \(C\): ([0-9.]+).*/T=\1/; s/.*sclk clock level: [0-9S]+: \(([0-9]+)Mhz\).*/sclk=\1/; s/.*Power \(W\): ([0-9.]+).*/P=\1/' | tr "\n" " "; echo; sleep $PER; done ) > $OUT 2>&1 &
SMI=$!
"$@"
kill $SMI
python3 - $OUT <<'PY'
import re, sys
rows = [dict(re.findall(r'(\w+)=([\d.]+)', l)) for l in open(sys.argv[1])]
busy = [r for r in rows if 'P' in r and float(r['P']) > 600 and 'sclk' in r]
if busy:
    f = lambda k: [float(r[k]) for r in busy]
    print(f"under load ({len(busy)} samples): sclk mean {sum(f('sclk'))/len(busy):.0f} MHz [{min(f('sclk')):.0f}, {max(f('sclk')):.0f}], "
          f"power mean {sum(f('P'))/len(busy):.0f} W (max {max(f('P')):.0f}), junction {max(f('T')):.0f} C")
PY
