cd $GRAFT_REPO_ROOT; O=gpurun_out/r5c; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python scripts/gemm_shapes.py > $O/gemm_shapes.txt 2>&1; echo "gemm_shapes rc=$?" | tee -a $O/summary.txt
for v in 0 1 0 1; do
  MC_FUSE_DW_BWD=$v timeout 600 python bench.py --workload cfg2 --steps 6 --warmup 2 --no-cpu-baseline 2>$O/bench_cfg2_fuse$v.err | tail -1 > $O/bench_cfg2_fuse$v.json
  python -c "import json;d=json.load(open('$O/bench_cfg2_fuse$v.json'));print('cfg2 fuse=$v', d['ms_per_step'], d['value'])" | tee -a $O/summary.txt
done
timeout 2400 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?" | tee -a $O/summary.txt
tail -n 8 $O/gpu_tests.log; cat $O/gemm_shapes.txt; cat $O/summary.txt
