# round 5 final records, one box: full GPU suite, every workload, the f16 parity configuration, rank shares, profiles
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05; mkdir -p $O
export TMPDIR=/tmp
if [ "$1" != "nobench_tests" ]; then timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1; tail -2 $O/pytest_gpu.txt; fi
for wl in cfg1 cfg2 cfg3; do python bench.py --workload $wl --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_$wl.json; done
python bench.py 2>/dev/null | tail -1 > $O/bench_cfg4.json
python bench.py --storage f16 --no-cpu-baseline --no-n8-load 2>/dev/null | tail -1 > $O/bench_cfg4_f16.json
python bench.py --workload cfg3 --storage f16 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_cfg3_f16.json
python bench.py --workload cfg5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_cfg5.json
for n in 2 4; do python bench.py --as-gpus $n --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_as$n.json; done
for f in $O/bench_*.json; do python -c "import json,sys; d=json.loads(open('$f').read()); print('$f', d['value'], d['ms_per_step'], d['config'].get('peak_reserved_gb'), (d.get('roofline') or {}).get('frac'), (d.get('roofline') or {}).get('class'))"; done
# (profiles: bash scripts/profile_round5.sh stats | pmc -- run separately; the PMC table must exist BEFORE this script so that the
#  bench lines carry roofline.traffic from profiles/r05_roofline_traffic.json)
