cd $GRAFT_REPO_ROOT; O=gpurun_out/r5g; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "xbwd" > $O/kernels.log 2>&1; echo "kernels rc=$?" | tee -a $O/summary.txt
timeout 600 python scripts/xbwd_ab.py > $O/xbwd_ab.txt 2>&1
timeout 600 python scripts/tn_split_sweep.py > $O/tn.txt 2>&1
timeout 600 python bench.py --workload cfg3 --steps 6 --warmup 2 --no-cpu-baseline 2>$O/bench_cfg3.err | tail -1 > $O/bench_cfg3.json
python -c "import json;d=json.load(open('$O/bench_cfg3.json'));print('cfg3', d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['class'])" | tee -a $O/summary.txt
timeout 1800 python -m pytest tests/test_fullsize_gpu.py -m gpu -x -q -s > $O/fullsize.log 2>&1; echo "fullsize rc=$?" | tee -a $O/summary.txt
tail -n 3 $O/kernels.log; cat $O/xbwd_ab.txt; grep "512 x\|3072 x   512" $O/tn.txt; tail -n 4 $O/fullsize.log; cat $O/summary.txt
