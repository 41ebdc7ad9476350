# round 5, GPU call 1: correctness of the new pieces + first measurements (everything into gpurun_out/r5a/)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r5a; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "dwconv or adamw or unscale" > $O/kernels.log 2>&1; echo "kernels rc=$?" | tee -a $O/summary.txt
timeout 600 python scripts/dw_form_ab.py > $O/dw_form_ab.txt 2>&1; echo "dw_form_ab rc=$?" | tee -a $O/summary.txt
timeout 900 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/summary.txt
timeout 1500 python -m pytest tests/test_f16_storage_gpu.py -m gpu -x -q -s -k "dynamic_loss_scale or baseline_shapes" > $O/f16.log 2>&1; echo "f16 rc=$?" | tee -a $O/summary.txt
for v in 0 1 0 1; do
  MC_FUSE_DW_BWD=$v timeout 600 python bench.py --workload cfg3 --steps 6 --warmup 2 --no-cpu-baseline 2>$O/bench_cfg3_fuse$v.err | tail -1 > $O/bench_cfg3_fuse$v.json
  python -c "import json;d=json.load(open('$O/bench_cfg3_fuse$v.json'));print('cfg3 fuse=$v', d['ms_per_step'], d['value'])" | tee -a $O/summary.txt
done
tail -5 $O/kernels.log $O/smoke.log $O/f16.log; cat $O/dw_form_ab.txt; cat $O/summary.txt
