cd $GRAFT_REPO_ROOT; O=gpurun_out/r5b; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/summary.txt
timeout 1500 python -m pytest tests/test_f16_storage_gpu.py -m gpu -x -q -s -k "dynamic_loss_scale or baseline_shapes" > $O/f16.log 2>&1; echo "f16 rc=$?" | tee -a $O/summary.txt
for v in 0 1 0 1; do
  MC_FUSE_DW_BWD=$v timeout 600 python bench.py --workload cfg3 --steps 6 --warmup 2 --no-cpu-baseline 2>$O/bench_cfg3_fuse$v.err | tail -1 > $O/bench_cfg3_fuse$v.json
  python -c "import json;d=json.load(open('$O/bench_cfg3_fuse$v.json'));print('cfg3 fuse=$v', d['ms_per_step'], d['value'])" | tee -a $O/summary.txt
done
timeout 600 python bench.py --workload cfg3 --steps 4 --warmup 2 --no-cpu-baseline --op-profile > $O/opprof.json 2> $O/opprof.txt
tail -n 5 $O/smoke.log; tail -n 30 $O/f16.log; cat $O/summary.txt; head -60 $O/opprof.txt
