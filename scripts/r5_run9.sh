cd $GRAFT_REPO_ROOT; O=gpurun_out/r5i; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>$O/bench_cfg4.err | tail -1 > $O/bench_cfg4.json
python -c "import json;d=json.load(open('$O/bench_cfg4.json'));print('cfg4', d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['class'], d.get('n8_load'))" | tee -a $O/summary.txt
timeout 900 python bench.py --storage f16 --steps 2 --warmup 1 --no-cpu-baseline --no-n8-load 2>$O/bench_cfg4_f16.err | tail -1 > $O/bench_cfg4_f16.json
python -c "import json;d=json.load(open('$O/bench_cfg4_f16.json'));print('cfg4 f16', d['ms_per_step'], d['value'], d['dtype'])" | tee -a $O/summary.txt
cat $O/summary.txt
