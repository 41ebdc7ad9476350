mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/r04/pytest_gpu.txt 2>&1; tail -2 gpurun_out/r04/pytest_gpu.txt
for wl in cfg1 cfg2 cfg3; do python bench.py --workload $wl --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r04/bench_$wl.json; done
python bench.py 2>/dev/null | tail -1 > gpurun_out/r04/bench_cfg4.json
python bench.py --workload cfg5 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r04/bench_cfg5.json
for n in 2 4; do python bench.py --as-gpus $n --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r04/bench_as$n.json; done
for f in gpurun_out/r04/bench_*.json; do python -c "import json,sys; d=json.loads(open('$f').read()); print('$f', d['value'], d['ms_per_step'], d['config'].get('peak_reserved_gb'))"; done
