cd $GRAFT_REPO_ROOT; O=gpurun_out/r5e; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python scripts/tn_split_sweep.py > $O/tn_split_sweep.txt 2>&1; echo "rc=$?"
cat $O/tn_split_sweep.txt
