cd $GRAFT_REPO_ROOT; O=gpurun_out/r5f; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "xbwd or wgrad or fused" > $O/kernels.log 2>&1; echo "kernels rc=$?" | tee -a $O/summary.txt
timeout 600 python scripts/xbwd_ab.py > $O/xbwd_ab.txt 2>&1; echo "xbwd_ab rc=$?" | tee -a $O/summary.txt
for v in 0 1 0 1; do
  MC_FUSE_XBWD=$v timeout 600 python bench.py --workload cfg3 --steps 6 --warmup 2 --no-cpu-baseline 2>$O/bench_cfg3_x$v.err | tail -1 > $O/bench_cfg3_x$v.json
  python -c "import json;d=json.load(open('$O/bench_cfg3_x$v.json'));print('cfg3 xbwd=$v', d['ms_per_step'], d['value'])" | tee -a $O/summary.txt
done
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_fullsize_gpu.py -m gpu -x -q -k "folded or e2e_vs_reference or trajectory or cfg3_shape" > $O/model.log 2>&1; echo "model rc=$?" | tee -a $O/summary.txt
tail -n 6 $O/kernels.log; cat $O/xbwd_ab.txt; tail -n 6 $O/model.log; cat $O/summary.txt
