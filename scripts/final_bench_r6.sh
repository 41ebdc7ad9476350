# round 6 final records, one box: every workload, the single-view secondary metric, rank shares; the default line carries parity /
# parity_build (f16 child leg).  (GPU suite: run separately; profiles: scripts/profile_round6.sh -- the PMC table must exist before
# this script so that the bench lines carry roofline.traffic from profiles/r06_roofline_traffic.json)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06; mkdir -p $O
export TMPDIR=/tmp
python bench.py 2>$O/bench_cfg4.err | tail -1 > $O/bench_cfg4.json
for wl in cfg1 cfg2 cfg3; do python bench.py --workload $wl --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_$wl.json; done
python bench.py --workload cfg3 --recompute 1 --steps 10 --warmup 3 --no-cpu-baseline --no-parity 2>/dev/null | tail -1 > $O/bench_cfg3_rc1.json
python bench.py --loss breast_clip_contrastive --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_cfg4_single_view.json
python bench.py --workload cfg3 --loss breast_clip_contrastive --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_cfg3_single_view.json
MC_XDW=0 python bench.py --no-cpu-baseline --no-parity --no-n8-load 2>/dev/null | tail -1 > $O/bench_cfg4_xdw0.json
python bench.py --workload cfg5 --no-cpu-baseline --no-parity 2>/dev/null | tail -1 > $O/bench_cfg5.json
for n in 2 4; do python bench.py --as-gpus $n --no-cpu-baseline --no-parity 2>/dev/null | tail -1 > $O/bench_as$n.json; done
for f in $O/bench_*.json; do python -c "import json,sys; d=json.loads(open('$f').read()); print('$f', d['value'], d['ms_per_step'], d['config'].get('peak_reserved_gb'), (d.get('roofline') or {}).get('frac'), (d.get('roofline') or {}).get('class'), (d.get('parity') or {}).get('train_dloss'), ((d.get('parity_build') or {}).get('parity') or {}).get('train_dloss'), (d.get('parity_build') or {}).get('pairs_per_s'))"; done
