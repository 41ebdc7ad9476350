cd $GRAFT_REPO_ROOT; O=gpurun_out/r5n; mkdir -p $O
export TMPDIR=/tmp
for v in 0 1 0 1; do
  MC_LANE_XMAP=$v timeout 600 python bench.py --workload cfg3 --steps 6 --warmup 2 --no-cpu-baseline 2>$O/b$v.err | tail -1 > $O/b$v.json
  python -c "import json;d=json.load(open('$O/b$v.json'));print('cfg3 xmap=$v', d['ms_per_step'], d['value'])" | tee -a $O/summary.txt
done
