#!/bin/bash
# A/B of the E-free stride-1 3x3 blocks (MC_EFREE: fused forward + fused backward with e rows from the block input) on one box
mkdir -p gpurun_out
run() { local name=$1; shift; local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" python bench.py --no-cpu-baseline --no-n8-load --no-parity --roofline-in-timed-region "$@" 2>gpurun_out/abe_$name.err | tail -1 > gpurun_out/abe_$name.json
  python - "$name" <<'PY'
import json, sys
n = sys.argv[1]
try:
    r = json.load(open(f"gpurun_out/abe_{n}.json"))
    print(f"{n:24s} {r['value']:9.2f} pairs/s {r['ms_per_step']:10.1f} ms  peak {r['config']['peak_hbm_gb']} GB  loss {r['config']['loss']}  tagged GB/step {r['whole_step']['algorithmic_gb_per_step']}")
except Exception as e:
    print(n, "FAILED", e)
PY
}
run cfg3_e0 MC_EFREE=0 -- --workload cfg3 --steps 8 --warmup 2
run cfg3_e1 MC_EFREE=1 -- --workload cfg3 --steps 8 --warmup 2
run cfg2_e0 MC_EFREE=0 -- --workload cfg2 --steps 8 --warmup 2
run cfg2_e1 MC_EFREE=1 -- --workload cfg2 --steps 8 --warmup 2
run cfg4_e0 MC_EFREE=0 -- --steps 2 --warmup 1
run cfg4_e1 MC_EFREE=1 -- --steps 2 --warmup 1
run n8_e0 MC_EFREE=0 -- --as-gpus 8 --steps 3 --warmup 1
run n8_e1 MC_EFREE=1 -- --as-gpus 8 --steps 3 --warmup 1
