#!/bin/bash
# round 6: kept-graph sweep of the headline run (N = 1, 32 micro-batches) with the fused forward launch on (the re-forwarded
# micro-batch's graph no longer holds the expanded tensors of the narrow-input blocks: room for more kept mode-2 graphs)
mkdir -p gpurun_out
for kk in "$@"; do
  python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-n8-load --no-parity --roofline-in-timed-region --keep-kept $kk 2>gpurun_out/keep_$kk.err | tail -1 > gpurun_out/keep_$kk.json
  python - $kk <<'PY'
import json, sys
k = sys.argv[1]
try:
    r = json.load(open(f"gpurun_out/keep_{k}.json"))
    print(f"keep-kept {k}: {r['value']:.2f} pairs/s  {r['ms_per_step']:.1f} ms  peak {r['config']['peak_hbm_gb']} / {r['config']['peak_reserved_gb']} GB")
except Exception as e:
    print(k, "FAILED", e, open(f"gpurun_out/keep_{k}.err").read()[-400:])
PY
done
