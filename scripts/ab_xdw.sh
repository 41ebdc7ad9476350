#!/bin/bash
# A/B of the fused expand + depthwise forward launch (MC_XDW) on one box: cfg3 in recompute modes 0 / 1, cfg4 (headline), n8 share
mkdir -p gpurun_out
run() { # name, env..., -- args
  local name=$1; shift
  local envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" python bench.py --no-cpu-baseline --no-n8-load --roofline-in-timed-region "$@" 2>gpurun_out/ab_$name.err | tail -1 > gpurun_out/ab_$name.json
  python - "$name" <<'PY'
import json, sys
n = sys.argv[1]
try:
    r = json.load(open(f"gpurun_out/ab_{n}.json"))
    print(f"{n:28s} {r['value']:9.2f} pairs/s {r['ms_per_step']:10.1f} ms  peak {r['config']['peak_hbm_gb']} GB  loss {r['config']['loss']}")
except Exception as e:
    print(n, "FAILED", e)
PY
}
run cfg3_xdw0_rc0 MC_XDW=0 -- --workload cfg3 --steps 6 --warmup 2
run cfg3_xdw1_rc1 MC_XDW=1 -- --workload cfg3 --steps 6 --warmup 2 --recompute 1
run cfg3_xdw0_rc1 MC_XDW=0 -- --workload cfg3 --steps 6 --warmup 2 --recompute 1
run cfg4_xdw0 MC_XDW=0 -- --steps 2 --warmup 1
run cfg4_xdw1 MC_XDW=1 -- --steps 2 --warmup 1
run n8_xdw0 MC_XDW=0 -- --as-gpus 8 --steps 3 --warmup 1
run n8_xdw1 MC_XDW=1 -- --as-gpus 8 --steps 3 --warmup 1
